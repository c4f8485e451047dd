// slot_rate.hip -- micro-benchmark: what ONE wave per SIMD can issue beside a 32x32x16 f16 MFMA on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 scripts/ubench/slot_rate.hip -o /tmp/slot_rate && /tmp/slot_rate
// A "slot" = one MFMA (four accumulators in rotation: no dependent MFMA closer than four slots) followed by a fixed group of
// filler instructions, pinned with sched_barrier.  256 workgroups of 4 waves (one wave per SIMD, launch bound 256 x 1); cycles
// per slot from s_memtime (shader clock) of workgroup 0, effective clock from the wall time.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// KIND: 0 nothing | 1 NV x v_fma_f32 (independent) | 2 NV x v_exp_f32 | 3 NV x v_add_f32 in ONE dependent chain
//       4 NV x v_cvt_pk_f16_f32 | 5 NV x ds_read_b128 | 6 softmax chunk: 2 exp + 2 add (two chains) + 1 cvt_pk
//       7 chunk + one ds_read_b128 | 8 chunk without the adds (2 exp + cvt) | 9 NV x v_pk_add_f32 | 10 NV x s_nop 0 (SALU-ish issue)
//       11 chunk whose exponentials read MFMA RESULTS (the accumulator written two slots earlier)
//       12 chunk, MFMA accumulators pinned to AGPRs (asm)   13 = 11 + 12 alternating (the attention kernel's mix)
//       14 chunk + one ds_read_b128 per slot whose result is the A operand EIGHT slots later (counted waits, no stall on the read)
//       15 the same, one read every other slot   16 = 15 + exponentials on MFMA results + half the MFMAs on AGPR accumulators
template <int KIND, int NV, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(float* out, unsigned long long* clk, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
    const int lane = threadIdx.x & 63;
    h8 a = {1, 2, 3, 4, 5, 6, 7, (_Float16)lane}, b = {(_Float16)(lane & 7), 1, 1, 1, 1, 1, 1, 1};
    f16v c[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    float x[8], acc0 = 0.f, acc1 = 0.f;
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (lane + i);
    unsigned pk = 0;
    typedef const __attribute__((address_space(3))) h8* lp;
    h8 ld[2] = {a, b};
    const unsigned laddr = (unsigned)(size_t)&lds[0] + lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    h8 ring[8] = {a, a, a, a, a, a, a, a};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            constexpr bool RING = KIND == 14 || KIND == 15 || KIND == 16;
            const bool agpr = KIND == 12 || ((KIND == 13 || KIND == 16) && (s & 1));
            if (agpr) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[s & 3]) : "v"(RING ? ring[s] : a), "v"(b));
            else c[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(RING ? ring[s] : a, b, c[s & 3], 0, 0, 0);
            if (RING && (KIND == 14 || (s & 1))) ring[s] = *(lp)(laddr + 1024 * s);
            if (KIND >= 11) {
                const bool from_mfma = KIND == 11 || ((KIND == 13 || KIND == 16) && !(s & 1));
                float e0, e1;
                if (from_mfma) {
                    asm volatile("v_exp_f32 %0, %1" : "=v"(e0) : "v"(c[(s + 2) & 3][2 * s]));
                    asm volatile("v_exp_f32 %0, %1" : "=v"(e1) : "v"(c[(s + 2) & 3][2 * s + 1]));
                } else {
                    asm volatile("v_exp_f32 %0, %1" : "=v"(e0) : "v"(x[0]));
                    asm volatile("v_exp_f32 %0, %1" : "=v"(e1) : "v"(x[1]));
                }
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc0) : "v"(x[2]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc1) : "v"(x[3]));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(x[2]), "v"(x[3]));
                x[2] = e0; x[3] = e1;
            }
            if (KIND == 1) { for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[v & 7])); }
            if (KIND == 2) { for (int v = 0; v < NV; ++v) asm volatile("v_exp_f32 %0, %0" : "+v"(x[v & 7])); }
            if (KIND == 3) { for (int v = 0; v < NV; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc0) : "v"(x[v & 7])); }
            if (KIND == 4) { for (int v = 0; v < NV; ++v) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(x[v & 7]), "v"(x[(v + 1) & 7])); }
            if (KIND == 5) { for (int v = 0; v < NV; ++v) { ld[v & 1] = *(lp)(laddr + 1024 * (v & 7)); asm volatile("" : "+v"(ld[v & 1])); } }
            if (KIND == 6 || KIND == 7 || KIND == 8) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(x[0]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(x[1]));
                if (KIND != 8) {
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc0) : "v"(x[2]));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc1) : "v"(x[3]));
                }
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(x[2]), "v"(x[3]));
                if (KIND == 7) { ld[s & 1] = *(lp)(laddr + 1024 * (s & 7)); asm volatile("" : "+v"(ld[s & 1])); }
            }
            if (KIND == 9) { for (int v = 0; v < NV; ++v) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&x[2 * (v & 3)]) : "v"(*(double*)&x[2 * ((v + 1) & 3)])); }
            if (KIND == 10) { for (int v = 0; v < NV; ++v) asm volatile("s_nop 0"); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = acc0 + acc1 + (float)pk + (float)ld[0][0] + (float)ld[1][1];
    for (int i = 0; i < 8; ++i) r += x[i];
    for (int i = 0; i < 4; ++i) r += c[i][lane & 15];
    for (int i = 0; i < 8; ++i) r += (float)ring[i][lane & 7];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}

template <int KIND, int NV, int WAVES = 4>
int run(const char* name, float* d, unsigned long long* dclk) {
    const int iters = 4000, blocks = 256;   // one workgroup per CU
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<KIND, NV, WAVES><<<blocks, 64 * WAVES>>>(d, dclk, 50);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k<KIND, NV, WAVES><<<blocks, 64 * WAVES>>>(d, dclk, iters);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long cyc; CK(hipMemcpy(&cyc, dclk, 8, hipMemcpyDeviceToHost));
    const double slots = 8.0 * iters;
    printf("%-44s %6.1f cycles per slot of one wave   %.2f GHz   %.0f TFLOP/s\n", name, cyc / slots, cyc / (ms * 1e6),
           256.0 * WAVES * slots * 32768.0 / (ms * 1e-3) * 1e-12);
    return 0;
}

int main() {
    float* d; unsigned long long* dclk;
    CK(hipMalloc(&d, 256 * 512 * 4 * 4)); CK(hipMalloc(&dclk, 8));
    run<0, 0>("MFMA alone", d, dclk);
    run<1, 1>("+ 1 v_fma", d, dclk); run<1, 2>("+ 2 v_fma", d, dclk); run<1, 3>("+ 3 v_fma", d, dclk);
    run<1, 4>("+ 4 v_fma", d, dclk); run<1, 5>("+ 5 v_fma", d, dclk); run<1, 6>("+ 6 v_fma", d, dclk); run<1, 8>("+ 8 v_fma", d, dclk);
    run<2, 1>("+ 1 v_exp", d, dclk); run<2, 2>("+ 2 v_exp", d, dclk); run<2, 4>("+ 4 v_exp", d, dclk);
    run<3, 2>("+ 2 v_add (dependent chain)", d, dclk); run<3, 4>("+ 4 v_add (dependent chain)", d, dclk);
    run<4, 1>("+ 1 v_cvt_pk", d, dclk); run<4, 2>("+ 2 v_cvt_pk", d, dclk);
    run<9, 1>("+ 1 v_pk_add_f32", d, dclk); run<9, 2>("+ 2 v_pk_add_f32", d, dclk);
    run<5, 1>("+ 1 ds_read_b128", d, dclk); run<5, 2>("+ 2 ds_read_b128", d, dclk);
    run<10, 4>("+ 4 s_nop 0", d, dclk);
    run<8, 0>("+ 2 exp + cvt_pk", d, dclk);
    run<6, 0>("+ softmax chunk (2 exp, 2 add, cvt_pk)", d, dclk);
    run<7, 0>("+ softmax chunk + ds_read_b128", d, dclk);
    run<11, 0>("+ chunk, exp on MFMA results", d, dclk);
    run<12, 0>("+ chunk, AGPR accumulators", d, dclk);
    run<13, 0>("+ chunk, both alternating", d, dclk);
    run<14, 0>("+ chunk + ds_read per slot, used 8 slots later", d, dclk);
    run<15, 0>("+ chunk + ds_read every other slot, used later", d, dclk);
    run<16, 0>("+ all of it (attention4's mix)", d, dclk);
    run<0, 0, 8>("MFMA alone, 2 waves per SIMD", d, dclk);
    run<6, 0, 8>("+ softmax chunk, 2 waves per SIMD", d, dclk);
    run<7, 0, 8>("+ softmax chunk + ds_read, 2 waves per SIMD", d, dclk);
    return 0;
}
