// micro-benchmark: issue rate of MFMA variants on gfx950 (development aid)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    h4 a4 = {(_Float16)threadIdx.x, 1, 2, 3}, b4 = {1, (_Float16)threadIdx.x, 1, 1};
    h8 a8 = {1, 2, 3, 4, 5, 6, 7, (_Float16)threadIdx.x}, b8 = a8;
    float fa = threadIdx.x, fb = 1.f;
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c3, 0, 0, 0);
        } else if (KIND == 1) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c3, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
template <int KIND>
void run(const char* name, float* d) {
    const int iters = 20000, blocks = 256 * 2;  // 2 WGs x 4 waves per CU = 2 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<blocks, 256>>>(d, 100);
    hipEventRecord(a);
    k<KIND><<<blocks, 256>>>(d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per SIMD: 2 waves x iters x 4 MFMAs
    double mf = 2.0 * iters * 4;
    printf("%s: %.3f ms -> %.1f ns per MFMA per SIMD (%.1f cycles @2.4GHz)\n", name, ms, ms * 1e6 / mf, ms * 1e6 / mf * 2.4);
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    run<0>("mfma_f32_16x16x16_f16", d);
    run<1>("mfma_f32_16x16x32_f16", d);
    run<2>("mfma_f32_16x16x4_f32 ", d);
    return 0;
}
