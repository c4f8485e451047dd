// micro-benchmark (development aid): VALU issue rates and MFMA / VALU co-issue inside ONE wave per SIMD on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(S) S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
template <int KIND>
__global__ __launch_bounds__(256) void valu(float* out, int iters) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    const float a = 0.999f, b = 0.001f;
    const f2 pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#define S(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
            REP8(S)
#undef S
        } else if (KIND == 1) {
#define S(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
            REP8(S)
#undef S
        } else if (KIND == 2) {
#define S(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(pa), "v"(pb));
            S(p0) S(p1) S(p2) S(p3) S(p4) S(p5) S(p6) S(p7)
#undef S
        } else if (KIND == 3) {
#define S(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
            REP8(S)
#undef S
        } else if (KIND == 4) {
#define S(x) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(a));
            REP8(S)
#undef S
        } else if (KIND == 5) {
#define S(x) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
            REP8(S)
#undef S
        } else if (KIND == 6) {
#define S(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
            REP8(S)
#undef S
        } else if (KIND == 7) {
#define S(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(pa));
            S(p0) S(p1) S(p2) S(p3) S(p4) S(p5) S(p6) S(p7)
#undef S
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
}

// one wave per SIMD: per iteration 2 independent MFMA 32x32x16 chains + NV independent v_fma_f32
template <int NV, int LDS>
__global__ __launch_bounds__(256) void mix(float* out, int iters) {
    __shared__ float sm[4096];
    f16v c0, c1;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    bf8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (__bf16)(float)(threadIdx.x & 7); b[r] = (__bf16)1.f; }
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const float ka = 0.999f, kb = 0.001f;
    sm[threadIdx.x] = x0;
    __syncthreads();
    const unsigned la = (unsigned)(size_t)&sm[(threadIdx.x & 63) * 4];
    float4 l0 = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#define V(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(ka), "v"(kb));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
        if (LDS) asm volatile("ds_read_b128 %0, %1" : "=v"(l0) : "v"(la));
        if (NV > 0) V(x0) if (NV > 1) V(x1) if (NV > 2) V(x2) if (NV > 3) V(x3)
        if (NV > 4) V(x4) if (NV > 5) V(x5) if (NV > 6) V(x6) if (NV > 7) V(x7)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
        if (NV > 8) V(x0) if (NV > 9) V(x1) if (NV > 10) V(x2) if (NV > 11) V(x3)
        if (NV > 12) V(x4) if (NV > 13) V(x5) if (NV > 14) V(x6) if (NV > 15) V(x7)
#undef V
        if (LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + l0.x;
}

template <typename F>
float timeit(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f(100);
    hipEventRecord(a);
    f(20000);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    const char* names[] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "v_max3_f32", "v_cvt_pk_bf16_f32", "v_med3_i32", "v_rcp_f32", "v_pk_mul_f32"};
#define RUNV(K) { float ms = timeit([&](int it) { valu<K><<<256, 256>>>(d, it); }); \
    printf("%-20s %.3f ms -> %.2f ns per wave-instruction (1 wave/SIMD) = %.1f cycles @2.4GHz\n", names[K], ms, ms * 1e6 / (20000.0 * 8), ms * 1e6 / (20000.0 * 8) * 2.4); }
    RUNV(0) RUNV(1) RUNV(2) RUNV(3) RUNV(4) RUNV(5) RUNV(6) RUNV(7)
#define RUNM(NV, L) { float ms = timeit([&](int it) { mix<NV, L><<<256, 256>>>(d, it); }); \
    printf("2 MFMA 32x32x16 + %2d v_fma%s per iteration: %.3f ms -> %.1f ns = %.0f cycles @2.4GHz per iteration\n", NV, L ? " + ds_read_b128" : "", ms, ms * 1e6 / 20000.0, ms * 1e6 / 20000.0 * 2.4); }
    RUNM(0, 0) RUNM(4, 0) RUNM(8, 0) RUNM(12, 0) RUNM(16, 0) RUNM(0, 1) RUNM(8, 1) RUNM(16, 1)
    return 0;
}
