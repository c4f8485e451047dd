// Does v_mfma_f32_32x32x16_f16 / _bf16 / 16x16x32 flush SUBNORMAL 16-bit inputs?  (round 6: the split-operand kernels carry lo
// halves that are fp16 subnormals for |x| < 1/8.)   hipcc --offload-arch=gfx950 -O2 mfma_subnormal.hip -o mfma_subnormal
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k(float a_val, float b_val, float* out) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    a[0] = (_Float16)a_val;   // every lane: A[i][8 kg] = a_val, B[8 kg][j] = b_val -> D[i][j] = 2 a b
    b[0] = (_Float16)b_val;
    f16v c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = c[0];
        out[1] = d[0];
        out[2] = (float)a[0];                 // what the conversion itself made of a_val
        out[3] = (float)a[0] * (float)b[0];   // VALU product of the converted values
    }
}

int main() {
    float* d;
    hipMalloc(&d, 16);
    const float cases[][2] = {{1.0f, 1.0f}, {3.0e-5f, 1024.f}, {5.9604645e-8f, 1024.f}, {1024.f, 3.0e-5f}, {3.0e-5f, 3.0e-5f},
                              {6.2e-5f, 1024.f}};
    for (auto& c : cases) {
        k<<<1, 64>>>(c[0], c[1], d);
        float h[4];
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("a %.8g b %.8g : mfma32x32x16 %.8g (x2 lanes-k: expect %.8g)  mfma16x16x32 %.8g (expect %.8g)  cvt(a) %.8g\n", c[0], c[1], h[0],
               2 * h[3], h[1], 4 * h[3], h[2]);
    }
    return 0;
}
