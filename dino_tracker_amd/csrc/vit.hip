// vit.hip -- P1: DINOv2 ViT encoder as driven by VitExtractor (models/extractor.py:41-150, utils.py:33-72):
// overlapping patch embedding (14x14, stride 7) of ImageNet-normalised frames + interpolated position encoding,
// `depth` transformer blocks (LN -> QKV -> MHSA -> proj -> LayerScale -> +res; LN -> fc1 -> GELU -> fc2 -> LayerScale
// -> +res), output = residual stream after the last requested block (no final norm), CLS included.
//
//   patch_embed_kernel  exact fp32 implicit GEMM on the f32-input MFMA (K = 3*14*14 = 588), mean/std fused in the load
//   layernorm_kernel    one wave per token, fp32 statistics, 16-bit output
//   gemm_tiled_kernel   C = A W^T on MFMA 16x16x32 (fp32 accumulate), 128x128 tiles, LDS double-buffered, with
//                       fused epilogues: QKV split (Q pre-scaled by log2(e)/sqrt(d), V written transposed), GELU,
//                       LayerScale + residual add into the fp32 stream
//   attention4_kernel   (vit_attention4.h; round 4) flash attention, d_head = 64: S^T = K Q^T and O^T = V^T P^T on MFMA 32x32x16
//                       so that every per-query quantity is lane-local; K / V^T tiles by LDS-DMA into XOR-swizzled images;
//                       ONE wave per SIMD with 64 queries, its two query tiles half a key tile out of phase (the softmax of
//                       one runs under the MFMAs of the other), fragments held in registers for both; exp2-domain softmax
//                       with optimistic exponentials (guarded); XCD-aware grid.  attention2_kernel (vit_attention2.h: 16
//                       waves per CU, 32 queries each; rounds 2-3) stays selectable (DTK_VIT_ATTENTION_V2) as the cross-check.
// Residual stream fp32.  Matrix operands (LN output, Q / K / V^T / P, attention output, MLP hidden, the pending residual
// update, the weights) are a template parameter T: _Float16 by default since round 3 -- the same MFMA rate as bf16 with
// 8x less operand rounding (fp16 range: activations saturate at +-65504, FP16_OVFL mode, and set the model's overflow
// word) -- or __bf16 with DTK_VIT_BF16 (the reference runs fp32; parity is stated in DESIGN.md section 4).
#include <stdlib.h>
#include <utility>
#include "common.h"

#define ATT2_NS att2_f16
#define ATT2_T _Float16
#define ATT2_F16 1
#define ATT2_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#include "vit_attention2.h"
#include "vit_attention4.h"
#include "vit_attention5.h"
#include "vit_attention6.h"
#undef ATT2_NS
#undef ATT2_T
#undef ATT2_F16
#undef ATT2_MFMA
#define ATT2_NS att2_bf16
#define ATT2_T __bf16
#define ATT2_F16 0
#define ATT2_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#include "vit_attention2.h"
#include "vit_attention4.h"
#include "vit_attention5.h"
#include "vit_attention6.h"
#undef ATT2_NS
#undef ATT2_T
#undef ATT2_F16
#undef ATT2_MFMA

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

// operand-type plumbing of the templated kernels below
template <typename T> struct Vec {
    typedef T t8 __attribute__((ext_vector_type(8)));
    typedef T t4 __attribute__((ext_vector_type(4)));
};
template <typename T> struct IsF16 { static constexpr bool value = false; };
template <> struct IsF16<_Float16> { static constexpr bool value = true; };
__device__ __forceinline__ f4 mfma16(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f4 mfma16(bf8 a, bf8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f16v mfma32(h8 a, h8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f16v mfma32(bf8 a, bf8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
template <typename T> __device__ __forceinline__ void operand_mode() {
    if (IsF16<T>::value) att2c::fp16_saturate_mode();  // overflowing fp16 results clamp to +-65504 instead of +-inf
}

// ---------------------------------------------------------------------------------------------------------------
// patch embedding: tokens[f][1 + r*pw + c][:] = W . patch(r,c) + b + pos[r*pw + c];  tokens[f][0] = cls_pos
// 32 patches x 32 output features per wave (MFMA 32x32x2 f32), 4 waves = 64 patches x 64 features per workgroup
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patch_embed_kernel(const float* __restrict__ frames, const float* __restrict__ Wp,
                                                          const float* __restrict__ bp, const float* __restrict__ pos,
                                                          const float* __restrict__ cls_pos,
                                                          const float* __restrict__ mean_std, float* __restrict__ x,
                                                          int H, int W, int ph, int pw, int D, int patch, int stride,
                                                          int S) {
    const int HW = ph * pw;
    const int K = 3 * patch * patch;
    const int frame = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int p0 = blockIdx.x * 64 + (w >> 1) * 32;
    const int n0 = blockIdx.y * 64 + (w & 1) * 32;
    const int li = lane & 31, lk = lane >> 5;
    const float* img = frames + (size_t)frame * 3 * H * W;
    const int p = min(p0 + li, HW - 1);
    const int pr = p / pw, pc = p % pw;
    const int n = min(n0 + li, D - 1);
    const float* wrow = Wp + (size_t)n * K;  // [D][3][patch][patch]
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = lk; k < K; k += 2) {
        const int ch = k / (patch * patch), rem = k - ch * patch * patch;
        const int ky = rem / patch, kx = rem - ky * patch;
        const float px = img[((size_t)ch * H + pr * stride + ky) * W + pc * stride + kx];
        const float a = (px - mean_std[ch]) / mean_std[3 + ch];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wrow[k], acc, 0, 0, 0);
    }
    const int co = n0 + li;
    if (co < D) {
        const float b = bp[co];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * lk;
            const int pp = p0 + i;
            if (pp < HW) x[((size_t)frame * S + 1 + pp) * D + co] = acc[r] + b + pos[(size_t)pp * D + co];
        }
    }
    if (blockIdx.x == 0 && (w >> 1) == 0 && lk == 0 && co < D) x[(size_t)frame * S * D + co] = cls_pos[co];
}

// ---------------------------------------------------------------------------------------------------------------
// patch embedding on the split-fp16 MFMA (patch 14, stride 7): fp32-grade results from fp16 matrix instructions.
// Every fp32 operand is carried as hi + lo fp16 halves and a product is xh.wh + xh.wl + xl.wh accumulated in fp32 (each
// fp16 x fp16 product is exact in fp32; the dropped xl.wl term is 2^-22 relative).  The normalised frame is first
// rewritten as two NHWC planes with 4 channels per pixel (channel 3 = 0), so that with k = (ky * 14 + kx) * 4 + c an
// 8-wide A fragment is two neighbouring taps = two 8-byte LDS reads -- no im2col gather.  A workgroup owns 128
// consecutive tokens of one patch row (a wave 32 of them) x 96 output features and walks K = 784 in seven chunks of two
// kernel rows; per chunk the two pixel rows (903 px) and the 96 x 112 weight slab (both planes) are staged in LDS.
// ---------------------------------------------------------------------------------------------------------------
constexpr int PE_P = 14, PE_S = 7, PE_TOK = 128, PE_NF = 96;
constexpr int PE_ROWPX = (PE_TOK - 1) * PE_S + PE_P;  // 903 pixels of one kernel row for 128 tokens
constexpr int PE_K = PE_P * PE_P * 4;                 // 784
constexpr int PE_CK = 2 * PE_P * 4;                   // 112 k per chunk (two kernel rows)
constexpr int PE_WP = PE_CK + 8;                      // weight row pitch in halves (240 B: conflict-free fragments)
constexpr float PE_WSCALE = 256.f;                    // weights carry 2^8 so that their lo halves stay normal numbers

__global__ __launch_bounds__(256) void frame_split4_kernel(const float* __restrict__ frames, const float* __restrict__ mean_std,
                                                           h4v* __restrict__ hi_, h4v* __restrict__ lo_, int H, int W,
                                                           long long npix) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // over frames * H * W
    if (i >= npix) return;
    const long long f = i / ((long long)H * W), p = i - f * (long long)H * W;
    const float* fr = frames + f * 3 * (long long)H * W + p;
    h4v h, l;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = (fr[(long long)c * H * W] - mean_std[c]) / mean_std[3 + c];
        h[c] = (half_t)v;
        l[c] = (half_t)(v - (float)h[c]);
    }
    h[3] = (half_t)0.f;
    l[3] = (half_t)0.f;
    hi_[i] = h;
    lo_[i] = l;
}

// [D][3][14][14] fp32 -> split planes [D][784] of w * 2^8, k = (ky * 14 + kx) * 4 + c
__global__ void pack_patch_split_kernel(const float* __restrict__ w, int D, half_t* __restrict__ Wh, half_t* __restrict__ Wl) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)D * PE_K) return;
    const int n = (int)(idx / PE_K), k = (int)(idx - (long long)n * PE_K);
    const int tap = k >> 2, c = k & 3;
    const float v = c < 3 ? w[((size_t)n * 3 + c) * (PE_P * PE_P) + tap] * PE_WSCALE : 0.f;
    const half_t h = (half_t)v;
    Wh[idx] = h;
    Wl[idx] = (half_t)(v - (float)h);
}

__global__ __launch_bounds__(256) void patch_embed_split_kernel(const h4v* __restrict__ in_hi, const h4v* __restrict__ in_lo,
                                                                const half_t* __restrict__ Wh, const half_t* __restrict__ Wl,
                                                                const float* __restrict__ bp, const float* __restrict__ pos,
                                                                const float* __restrict__ cls_pos, float* __restrict__ x,
                                                                int H, int W, int ph, int pw, int D, int S, int groups_x,
                                                                int units, int slabs) {
    __shared__ h4v Xh[2 * PE_ROWPX], Xl[2 * PE_ROWPX];
    __shared__ __attribute__((aligned(16))) half_t Wsh[PE_NF * PE_WP], Wsl[PE_NF * PE_WP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // Block -> (frame, patch row, token group, feature slab).  Workgroup b runs on XCD b % 8 and every XCD has its own L2:
    // the feature slabs of a token group re-read the same pixel rows, and so does the patch row below it (stride 7, kernel
    // 14), so each XCD gets a contiguous range of (frame, patch row, group) units and runs a unit's slabs back to back --
    // with the plain 3-D grid those blocks sat on different XCDs and every re-read went to HBM (PMC: 7.5 GB fetched per
    // 90-frame launch for 0.6 GB of split frames).
    const int per_xcd = (units + 7) / 8;
    const int within = blockIdx.x >> 3;
    const int unit = (blockIdx.x & 7) * per_xcd + within / slabs;
    if (within / slabs >= per_xcd || unit >= units) return;
    const int n0 = (within % slabs) * PE_NF;
    const int pc0 = (unit % groups_x) * PE_TOK, pr = (unit / groups_x) % ph;
    const size_t frame = unit / (groups_x * ph);
    const int li = lane & 31, hh = lane >> 5;
    const h4v* fh = in_hi + frame * (size_t)H * W;
    const h4v* fl = in_lo + frame * (size_t)H * W;
    const int tokl = w * 32 + li;  // this lane's token (row of the A operand) inside the group
    f16v acc[3];
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    for (int c = 0; c < PE_P / 2; ++c) {
        __syncthreads();
        for (int idx = tid; idx < 2 * PE_ROWPX; idx += 256) {
            const int r = idx / PE_ROWPX, px = idx - r * PE_ROWPX;
            const size_t g = (size_t)(pr * PE_S + 2 * c + r) * W + min(pc0 * PE_S + px, W - 1);
            Xh[idx] = fh[g];
            Xl[idx] = fl[g];
        }
        for (int idx = tid; idx < PE_NF * (PE_CK / 8); idx += 256) {
            const int f = idx / (PE_CK / 8), piece = idx - f * (PE_CK / 8);
            const size_t src = (size_t)min(n0 + f, D - 1) * PE_K + c * PE_CK + piece * 8;
            *reinterpret_cast<uint4*>(Wsh + f * PE_WP + piece * 8) = *reinterpret_cast<const uint4*>(Wh + src);
            *reinterpret_cast<uint4*>(Wsl + f * PE_WP + piece * 8) = *reinterpret_cast<const uint4*>(Wl + src);
        }
        __syncthreads();
#pragma unroll
        for (int ls = 0; ls < PE_CK / 16; ++ls) {
            const int t0 = 2 * (2 * ls + hh);  // first of this lane's two taps inside the chunk (kx even: same row)
            const int p0 = (t0 / PE_P) * PE_ROWPX + tokl * PE_S + t0 % PE_P;
            const h4v a0 = Xh[p0], a1 = Xh[p0 + 1], b0 = Xl[p0], b1 = Xl[p0 + 1];
            const h8 xh = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            const h8 xl = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const int wo = (n * 32 + li) * PE_WP + ls * 16 + hh * 8;
                const h8 wh = *reinterpret_cast<const h8*>(Wsh + wo), wl = *reinterpret_cast<const h8*>(Wsl + wo);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, acc[n], 0, 0, 0);
            }
        }
    }
    // D[i][j]: j = lane & 31 (feature), i = (r&3) + 8 (r>>2) + 4 hh (token of the wave's 32)
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        const int co = n0 + n * 32 + li;
        if (co >= D) continue;
        const float b = bp[co];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pc = pc0 + w * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (pc < pw) {
                const int pp = pr * pw + pc;
                x[((size_t)frame * S + 1 + pp) * D + co] = acc[n][r] * (1.f / PE_WSCALE) + b + pos[(size_t)pp * D + co];
            }
        }
        if (pr == 0 && pc0 == 0 && w == 0 && hh == 0) x[(size_t)frame * S * D + co] = cls_pos[co];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm: fp32 row -> bf16 row (A operand of the next GEMM); one wave per token
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(float* __restrict__ x, const T* __restrict__ delta,
                                                        const float* __restrict__ gam, const float* __restrict__ bet,
                                                        T* __restrict__ y, long long rows, int D, float eps,
                                                        int* __restrict__ overflow) {
    typedef typename Vec<T>::t8 T8;
    typedef typename Vec<T>::t4 T4;
    (void)sizeof(T8); (void)sizeof(T4);
    operand_mode<T>();
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* p = x + row * D;
    // the row stays in registers (D <= 1024: four float4 per lane); a pending residual update (the LayerScale'd output of
    // the previous projection / MLP, written by the GEMM epilogue) is applied here: x += delta
    float4 v[4];
    float s = 0.f;
    bool sat = false;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
            v[it] = *reinterpret_cast<const float4*>(p + c);
            if (delta) {
                const T4 d = *reinterpret_cast<const T4*>(delta + row * D + c);
                const float d0 = (float)d[0], d1 = (float)d[1], d2 = (float)d[2], d3 = (float)d[3];
                // a saturated (or non-finite) residual update: the fp16 range was exceeded upstream
                if (IsF16<T>::value) sat |= !(fmaxf(fmaxf(fabsf(d0), fabsf(d1)), fmaxf(fabsf(d2), fabsf(d3))) < 65504.f);
                v[it].x += d0; v[it].y += d1; v[it].z += d2; v[it].w += d3;
                *reinterpret_cast<float4*>(p + c) = v[it];
            }
            s += (v[it].x + v[it].y) + (v[it].z + v[it].w);
        }
    }
    if (IsF16<T>::value && overflow && __any(sat) && lane == 0) atomicOr(overflow, 1);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (lane * 4 + it * 256 < D) {
            const float a = v[it].x - mean, b = v[it].y - mean, cc = v[it].z - mean, d = v[it].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    if (!y) return;  // final residual update only
    T* o = y + row * D;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
            const float4 g = *reinterpret_cast<const float4*>(gam + c), b = *reinterpret_cast<const float4*>(bet + c);
            T4 r = {(T)((v[it].x - mean) * rstd * g.x + b.x), (T)((v[it].y - mean) * rstd * g.y + b.y),
                     (T)((v[it].z - mean) * rstd * g.z + b.z), (T)((v[it].w - mean) * rstd * g.w + b.w)};
            *reinterpret_cast<T4*>(o + c) = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 GEMM  C[M][N] = A[M][K] . Wt[N][K]^T (+bias) with fused epilogues
// ---------------------------------------------------------------------------------------------------------------
constexpr int GM = 128, GN = 128, GK = 32;
enum { EPI_QKV = 0, EPI_GELU = 1, EPI_DELTA = 2, EPI_F32 = 3 };

template <typename T>
struct GemmEpi {
    const float* bias;   // [N]
    // EPI_QKV
    T* q;           // [F][heads][Sp][64]
    T* k;           // [F][heads][Sp][64]
    T* vt;          // [F][heads][64][Sp]
    int S, Sp, heads, D;
    float qscale;
    // EPI_GELU
    T* out;         // [M][N]
    // EPI_DELTA
    int no_store;        // DTK_DEV builds: skip the stores of the weight-stationary kernel (DTK_DEBUG & 65536)
    T* delta;       // [M][N] bf16: gamma * (A W^T + bias), added to the fp32 residual stream by the next LayerNorm
    const float* gamma;  // [N] LayerScale
    // EPI_F32 (tiled kernel only)
    float* out_f32;      // [M][N] fp32: A W^T + bias (the qkv facet output)
    // fp16 range (round 5): EPI_QKV / EPI_GELU kernels OR bit 2 / 4 into *ovf when a value they store reaches the fp16 limit --
    // every frame, inside the epilogue (no extra pass); NULL = not tracked (bf16 operands)
    int* ovf;
    // gemm_wide_delta_kernel<T, true, true> (round 6): the LayerNorm of the NEXT block inside fc2's epilogue -- x += delta where the
    // delta tile is staged, statistics and the 16-bit normalised row from the same registers (the arithmetic of layernorm_kernel)
    float* ln_x;             // [M][N] the fp32 residual stream
    const float *ln_w, *ln_b;
    T* ln_out;               // [M][N]
    float ln_eps;
    int* ln_ovf;             // overflow word: bit 1 = a saturated residual update (fp16 operands)
};

// running |max| of the values an epilogue stores (the saturation test of the fp16 range, see GemmEpi::ovf): two values per
// v_max3_f32 with |.| source modifiers.  NaN passes through fmaxf unnoticed: a non-finite activation surfaces in the next
// residual update, which the LayerNorm kernel checks for every token (overflow bit 1).
__device__ __forceinline__ float amax2(float m, float a, float b) { return fmaxf(fmaxf(m, fabsf(a)), fabsf(b)); }
template <typename T, int EPI>
__device__ __forceinline__ void amax_report(float amax, int* ovf) {
    if (IsF16<T>::value && (EPI == EPI_QKV || EPI == EPI_GELU) && ovf) {
        // (65488 = the midpoint below 65504: an fp32 value from there on is STORED as 65504, which the explicit scan of the tensor
        //  -- DTK_VIT_CHECK_RANGE -- reports)
        if (__any(amax >= 65488.f) && (threadIdx.x & 63) == 0) atomicOr(ovf, EPI == EPI_QKV ? 2 : 4);
    }
}

__device__ __forceinline__ int gswz(int row, int piece) {
    const int f = (0x1230 >> (((row >> 2) & 3) * 4)) & 3;  // {0,3,2,1}: conflict-free ds_read_b128 fragments
    return row * 4 + (piece ^ f);
}

// Epilogue of one 16x16 D tile: lane (fg, fj) holds rows mb .. mb+3 of column n (a[r]).  Shared by the tiled kernels.
template <typename T, int EPI>
__device__ __forceinline__ void gemm_store_tile(const f4& a, long long mb, int n, float bias, long long M, int N,
                                                const GemmEpi<T>& e, float& amax) {
    if (IsF16<T>::value && (EPI == EPI_QKV || EPI == EPI_GELU)) {   // rows past M hold the repeated last row: harmless
        const float sc = (EPI == EPI_QKV && n < e.D) ? e.qscale : 1.f;
        // what is STORED: Q already scaled; for the MLP hidden GELU(v) -- v for large positive v, ~0 for negative ones (ADVICE r5:
        // the pre-activation's |v| flagged a harmless v <= -65504): the positive part
        if (EPI == EPI_GELU) amax = fmaxf(fmaxf(fmaxf(amax, a[0] + bias), fmaxf(a[1] + bias, a[2] + bias)), a[3] + bias);
        else amax = amax2(amax2(amax, (a[0] + bias) * sc, (a[1] + bias) * sc), (a[2] + bias) * sc, (a[3] + bias) * sc);
    }
    if (EPI == EPI_QKV) {
        const int which = n / e.D, rem = n - which * e.D;
        const int head = rem >> 6, dh = rem & 63;
        const int frame = (int)(mb / e.S);  // the 4 rows of a fragment may straddle two frames: handle per row
        (void)frame;
        if (which == 2) {
            // V transposed: 4 consecutive tokens -> 8 contiguous bytes when they stay inside one frame
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long m = mb + r;
                if (m < M) {
                    const int f = (int)(m / e.S), s = (int)(m - (long long)f * e.S);
                    e.vt[(((size_t)f * e.heads + head) * 64 + dh) * e.Sp + s] = (T)(a[r] + bias);
                }
            }
        } else {
            T* dst = which == 0 ? e.q : e.k;
            const float sc = which == 0 ? e.qscale : 1.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long m = mb + r;
                if (m < M) {
                    const int f = (int)(m / e.S), s = (int)(m - (long long)f * e.S);
                    dst[(((size_t)f * e.heads + head) * e.Sp + s) * 64 + dh] = (T)((a[r] + bias) * sc);
                }
            }
        }
    } else if (EPI == EPI_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long m = mb + r;
            if (m < M) {
                const float v = a[r] + bias;
                e.out[m * N + n] = (T)(0.5f * v * (1.f + erff(v * 0.70710678118654752f)));
            }
        }
    } else if (EPI == EPI_F32) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long m = mb + r;
            if (m < M) e.out_f32[m * N + n] = a[r] + bias;
        }
    } else {
        const float gm = e.gamma[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long m = mb + r;
            if (m < M) e.delta[m * N + n] = (T)(gm * (a[r] + bias));
        }
    }
}

// (Round 2 measured two LDS-DMA forms of this main loop on fc2, K = 1536: 64-wide stages, two in flight, two barriers per
// stage: 17.9 ms; 32-wide stages in a ring of four, three in flight, one barrier per stage: 20.0 ms; this register-staged
// form: 17.1-17.5 ms.  Kept.  The SQ counters (profiles/r02_pmc_sq.md) show why it is slow -- 64 % of the wave cycles
// parked, MFMA pipe 27 % busy: one k-step of prefetch does not cover the HBM latency -- but TWO k-steps of register
// prefetch need 150 VGPRs = 3 waves per SIMD instead of 4 and measured 18.7 ms; forced to 128 VGPRs the loop spills.)
template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm_tiled_kernel(const T* __restrict__ A, const T* __restrict__ Wt,
                                                         long long M, int N, int K, GemmEpi<T> e) {
    typedef typename Vec<T>::t8 T8;
    typedef typename Vec<T>::t4 T4;
    (void)sizeof(T8); (void)sizeof(T4);
    operand_mode<T>();
    __shared__ uint4 As[2][GM * 4];
    __shared__ uint4 Bs[2][GN * 4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // block -> tile: workgroup b runs on XCD b % 8 (round-robin dispatch); the column tiles of one row block are given
    // to the same XCD back to back, so that A is fetched from HBM once and the other N/128 - 1 reads hit that XCD's L2
    // (with column-major block order fc2 re-read its 0.75 GB operand three times: 5.2 TB/s, HBM-bound)
    const int ncol = (N + GN - 1) / GN;
    const long long nrow = (M + GM - 1) / GM;
    const long long kb = blockIdx.x >> 3;
    const long long row_blk = (kb / ncol) * 8 + (blockIdx.x & 7);
    if (row_blk >= nrow) return;
    const long long m0 = row_blk * GM;
    const int n0 = (int)(kb % ncol) * GN;
    const int wr = w >> 1, wc = w & 1;  // wave tile 64 x 64
    const int fj = lane & 15, fg = lane >> 4;
    const int lrow = tid >> 2, lpiece = tid & 3;  // loader rows lrow, lrow + 64
    // clamp loader rows so that ragged M / N never read out of bounds (results of clamped rows are not stored)
    const long long ar0 = min(m0 + lrow, M - 1), ar1 = min(m0 + lrow + 64, M - 1);
    const int br0 = min(n0 + lrow, N - 1), br1 = min(n0 + lrow + 64, N - 1);
    const T* a0 = A + ar0 * K + lpiece * 8;
    const T* a1 = A + ar1 * K + lpiece * 8;
    const T* b0 = Wt + (size_t)br0 * K + lpiece * 8;
    const T* b1 = Wt + (size_t)br1 * K + lpiece * 8;
    f4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f4{0.f, 0.f, 0.f, 0.f};
    uint4 ra0 = *reinterpret_cast<const uint4*>(a0), ra1 = *reinterpret_cast<const uint4*>(a1);
    uint4 rb0 = *reinterpret_cast<const uint4*>(b0), rb1 = *reinterpret_cast<const uint4*>(b1);
    As[0][gswz(lrow, lpiece)] = ra0;
    As[0][gswz(lrow + 64, lpiece)] = ra1;
    Bs[0][gswz(lrow, lpiece)] = rb0;
    Bs[0][gswz(lrow + 64, lpiece)] = rb1;
    __syncthreads();
    const int nk = K / GK;
    int cur = 0;
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) {
            ra0 = *reinterpret_cast<const uint4*>(a0 + (ks + 1) * GK);
            ra1 = *reinterpret_cast<const uint4*>(a1 + (ks + 1) * GK);
            rb0 = *reinterpret_cast<const uint4*>(b0 + (ks + 1) * GK);
            rb1 = *reinterpret_cast<const uint4*>(b1 + (ks + 1) * GK);
        }
        T8 af[4], bfr[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const uint4 v = As[cur][gswz(wr * 64 + mi * 16 + fj, fg)];
            af[mi] = *reinterpret_cast<const T8*>(&v);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const uint4 v = Bs[cur][gswz(wc * 64 + ni * 16 + fj, fg)];
            bfr[ni] = *reinterpret_cast<const T8*>(&v);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = mfma16(af[mi], bfr[ni], acc[mi][ni]);
        if (ks + 1 < nk) {
            As[cur ^ 1][gswz(lrow, lpiece)] = ra0;
            As[cur ^ 1][gswz(lrow + 64, lpiece)] = ra1;
            Bs[cur ^ 1][gswz(lrow, lpiece)] = rb0;
            Bs[cur ^ 1][gswz(lrow + 64, lpiece)] = rb1;
        }
        __syncthreads();
        cur ^= 1;
    }
    // D fragment: lane (fg, fj) holds rows 4*fg + r (r = 0..3), column fj of each 16x16 tile
    float amax = 0.f;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + wc * 64 + ni * 16 + fj;
        if (n >= N) continue;
        const float bias = e.bias ? e.bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const long long mb = m0 + wr * 64 + mi * 16 + fg * 4;
            gemm_store_tile<T, EPI>(acc[mi][ni], mb, n, bias, M, N, e, amax);
        }
    }
    amax_report<T, EPI>(amax, e.ovf);
}

// ---------------------------------------------------------------------------------------------------------------
// weight-stationary GEMM for K = 384 (QKV, attention projection and fc1 of ViT-S): C[M][N] = A[M][384] . Wt[N][384]^T
//
// A workgroup of two waves owns 128 output features for a chunk of token tiles; a wave keeps its 64 weight rows in
// registers for the whole chunk (A operand of MFMA 32x32x16 bf16: 24 k-steps x 2 row tiles = 192 VGPRs) and the tokens
// stream through a triple-buffered LDS tile of 32 tokens (24 KB) filled by LDS-DMA two steps ahead, read once per wave
// as the B operand: one ds_read_b128 feeds two MFMAs and there is no K loop with barriers -- one barrier per 32 tokens.
// The weight rows are permuted over the MFMA rows so that in D a lane (token j, half h) holds 16 CONSECUTIVE output
// features (n0 + 32 t + 16 h + r): the epilogue of the previous tile (bias, GELU / Q scale / LayerScale, conversion,
// 16-byte stores) is issued between the MFMAs of the current one.  Two workgroups share a CU (one wave per SIMD).
// Measured alternatives (all slower on MI355X): staging the outputs through LDS for whole-line stores, one output
// feature per lane (2-byte stores), 4 tokens x 256 B per DMA request, register-staged tiles instead of LDS-DMA.
// ---------------------------------------------------------------------------------------------------------------
constexpr int WS_KS = 24, WS_K = 16 * WS_KS, WS_ROWS = 32, WS_COLS = 256;
constexpr int WS_TILE_BYTES = WS_ROWS * WS_K * 2;
constexpr int WS_LQ = WS_KS / 4;  // LDS-DMA requests per wave per tile
constexpr int WS_PITCH = 144;                    // staging row pitch: 128 B of payload + 16 (conflict-free 16-B accesses)
constexpr int WS_UNIT_BYTES = WS_ROWS * WS_PITCH;  // staging unit: 32 rows x 128 B (64 bf16 or 32 fp32 outputs)
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void ws_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
template <int N>
__device__ __forceinline__ void ws_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// One wave's share of a pipeline stage of the wide GEMMs by DESCRIPTOR LDS-DMA (round 6; common.h dtk_buffer_lds16: three issue
// slots per request where the global_load_lds form above costs ~13 -- 64-bit address arithmetic on the VALU, m0 saved and restored).
// Request i of wave w lands at stage + (w REQ + i) KB: the stages of gemm_wide_delta / gemm_wide / gemm_split_dma are laid out so.
// srd[i] = descriptor over the first row of the request's operand tile, voff[i] = the lane's constant byte offset in it (row, swizzled
// piece), soff = the k-step's byte offset.  A kernel that calls this must not use ws_glds16 (m0: tests/test_abi.py::test_m0_users).
template <int REQ, int I = 0>
__device__ __forceinline__ void wd_issue(const dtk_u4 (&srd)[REQ], const unsigned (&voff)[REQ], unsigned soff, unsigned lds_dst) {
    dtk_buffer_lds16<I * 1024>(srd[I], soff, voff[I], lds_dst);
    if constexpr (I + 1 < REQ) wd_issue<REQ, I + 1>(srd, voff, soff, lds_dst);
}

// GELU(x) = x Phi(x) with erf(s / sqrt 2) ~ s P(s^2) on |s| <= 4.25 (odd minimax polynomial, 9 coefficients, |err| < 2e-5;
// |GELU error| < 6e-5 everywhere, far below the bf16 rounding of the result); two values per packed fp32 instruction
__device__ __forceinline__ f2 gelu2(f2 x) {
    const f2 c = {4.25f, 4.25f};
    const f2 sx = __builtin_elementwise_min(__builtin_elementwise_max(x, -c), c);
    const f2 u = sx * sx;
    f2 p = {1.112979639e-10f, 1.112979639e-10f};
    p = __builtin_elementwise_fma(p, u, f2{-1.065557687e-08f, -1.065557687e-08f});
    p = __builtin_elementwise_fma(p, u, f2{4.510875158e-07f, 4.510875158e-07f});
    p = __builtin_elementwise_fma(p, u, f2{-1.125288873e-05f, -1.125288873e-05f});
    p = __builtin_elementwise_fma(p, u, f2{1.868377149e-04f, 1.868377149e-04f});
    p = __builtin_elementwise_fma(p, u, f2{-2.217123518e-03f, -2.217123518e-03f});
    p = __builtin_elementwise_fma(p, u, f2{1.963194646e-02f, 1.963194646e-02f});
    p = __builtin_elementwise_fma(p, u, f2{-1.326889843e-01f, -1.326889843e-01f});
    p = __builtin_elementwise_fma(p, u, f2{7.978046536e-01f, 7.978046536e-01f});
    const f2 one = {1.f, 1.f};
    const f2 e = __builtin_elementwise_min(__builtin_elementwise_max(sx * p, -one), one);
    const f2 hx = x * f2{0.5f, 0.5f};
    return __builtin_elementwise_fma(hx, e, hx);
}

// Epilogue of one TRANSPOSED 16x16 D tile (gemm_wide_kernel, round 5: its MFMAs run as (W tile) x (token tile)^T, so a lane
// holds four CONSECUTIVE output features nb .. nb+3 of ONE token m): one 8-byte store per tile and lane where the layout keeps
// features contiguous (GELU hidden, residual update, Q, K) instead of four 2-byte stores -- 32 store instructions per lane and
// 256 x 256 tile instead of 128 (the wide models' fc1 / qkv ran at 0.55 / 0.60 PF against fc2's 1.06 on the same main loop:
// profiles/r04_bench_width1024.json).  V^T keeps tokens contiguous, so its four features go out as 2-byte stores of 16 lanes = 32 B.
template <typename T, int EPI>
__device__ __forceinline__ void gemm_store_tile_t(const f4& a, long long m, int nb, long long M, int N, const GemmEpi<T>& e,
                                                  float& amax) {
    typedef typename Vec<T>::t4 T4;
    const float4 b4 = e.bias ? *reinterpret_cast<const float4*>(e.bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
    float v[4] = {a[0] + b4.x, a[1] + b4.y, a[2] + b4.z, a[3] + b4.w};
    if (EPI == EPI_QKV) {
        const int which = nb / e.D, rem = nb - which * e.D;
        const int head = rem >> 6, dh = rem & 63;
        const float sc = which == 0 ? e.qscale : 1.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= sc;
        if (IsF16<T>::value) amax = amax2(amax2(amax, v[0], v[1]), v[2], v[3]);
        if (m >= M || DTK_DBG(e.no_store, 4)) return;
        const int f = (int)(m / e.S), sp = (int)(m - (long long)f * e.S);
        if (which == 2) {
            T* vp = e.vt + (((size_t)f * e.heads + head) * 64 + dh) * e.Sp + sp;
#pragma unroll
            for (int r = 0; r < 4; ++r) vp[(size_t)r * e.Sp] = (T)v[r];
        } else {
            T* dst = (which == 0 ? e.q : e.k) + (((size_t)f * e.heads + head) * e.Sp + sp) * 64 + dh;
            *reinterpret_cast<T4*>(dst) = T4{(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
        }
    } else if (EPI == EPI_GELU) {
        if (IsF16<T>::value) amax = fmaxf(fmaxf(fmaxf(amax, v[0]), fmaxf(v[1], v[2])), v[3]);   // GELU(v) = v where it is large: the positive part
        if (m >= M || DTK_DBG(e.no_store, 4)) return;
        // (the packed polynomial GELU of the weight-stationary kernels: |error| < 6e-5, below the 16-bit rounding of the result;
        //  libm's erff costs ~10 x the instructions, and this epilogue runs for 4096 features of every token in fc1)
        const f2 g0 = gelu2(f2{v[0], v[1]}), g1 = gelu2(f2{v[2], v[3]});
        *reinterpret_cast<T4*>(e.out + m * N + nb) = T4{(T)g0[0], (T)g0[1], (T)g1[0], (T)g1[1]};
    } else {   // EPI_DELTA
        if (m >= M || DTK_DBG(e.no_store, 4)) return;
        const float4 g4 = *reinterpret_cast<const float4*>(e.gamma + nb);
        *reinterpret_cast<T4*>(e.delta + m * N + nb) = T4{(T)(g4.x * v[0]), (T)(g4.y * v[1]), (T)(g4.z * v[2]), (T)(g4.w * v[3])};
    }
}


// The same tile as a VALUE (EPI_GELU / EPI_DELTA; round 6: gemm_wide_kernel stages its output tile in LDS and writes whole rows)
template <typename T, int EPI>
__device__ __forceinline__ typename Vec<T>::t4 gemm_value_tile_t(const f4& a, const float4& b4, const float4& g4, float& amax) {
    typedef typename Vec<T>::t4 T4;
    const float v[4] = {a[0] + b4.x, a[1] + b4.y, a[2] + b4.z, a[3] + b4.w};
    if (EPI == EPI_GELU) {
        if (IsF16<T>::value) amax = fmaxf(fmaxf(fmaxf(amax, v[0]), fmaxf(v[1], v[2])), v[3]);
        const f2 g0 = gelu2(f2{v[0], v[1]}), g1 = gelu2(f2{v[2], v[3]});
        return T4{(T)g0[0], (T)g0[1], (T)g1[0], (T)g1[1]};
    }
    return T4{(T)(g4.x * v[0]), (T)(g4.y * v[1]), (T)(g4.z * v[2]), (T)(g4.w * v[3])};
}

// ---------------------------------------------------------------------------------------------------------------
// wide-tile GEMM for N = 384 and long K (fc2 of ViT-S: K = 1536):  C[M][384] = A[M][K] . Wt[384][K]^T, EPI_DELTA epilogue
//
// The 128x128 kernel above is bound by memory LATENCY, not bandwidth: a CU streams 16 KB per k-step and would need
// ~200 KB in flight to cover ~1.5 us, its 16 waves stage 64 KB (SQ counters: waves parked 64 %, MFMA pipe 27 % busy).
// Here one workgroup of 8 waves owns 256 rows x ALL 384 columns: A is read exactly once (no column tiles), a k-step
// moves 40 KB for 6.3 MFLOP (2.4x fewer bytes per flop), and the tiles arrive by LDS-DMA (global_load_lds_dwordx4, no
// staging registers) into a ring of three 40 KB stages, two of them in flight -- 80 KB per CU against the ~93 KB that
// Little's law asks for at this intensity.  LDS image = the 128x128 kernel's (row-major, four 16-byte pieces per row,
// piece XOR-swizzled by gswz), produced by giving every DMA lane the matching SOURCE address.  One barrier per k-step.
// Wave grid 2 x 4, wave tile 128 x 96 = 8 x 6 MFMA 16x16x32 tiles: 192 accumulator registers, two waves per SIMD.
// (For the attention projection, K = 384 = 12 k-steps, the pipeline's fill time dominates: 6.9 ms against 5.4 ms on the
// weight-stationary kernel.  fc2 only.)
// ---------------------------------------------------------------------------------------------------------------
constexpr int WD_M = 256, WD_N = 384, WD_STAGES = 3;
constexpr int WD_A_BYTES = WD_M * 64, WD_B_BYTES = WD_N * 64, WD_STAGE_BYTES = WD_A_BYTES + WD_B_BYTES;
constexpr int WD_REQ = (WD_M + WD_N) / 16 / 8;  // DMA requests per wave and stage (16 rows of 64 B each): 5

constexpr int WD_OPITCH = WD_N * 2 + 8;   // staged output rows (round 6): 768 B + 8

template <typename T, bool STAGED = true, bool FUSE_LN = false>
__global__ __launch_bounds__(512, 2) void gemm_wide_delta_kernel(const T* __restrict__ A, const T* __restrict__ Wt,
                                                                 long long M, int K, GemmEpi<T> e) {
    typedef typename Vec<T>::t8 T8;
    typedef typename Vec<T>::t4 T4;
    (void)sizeof(T8); (void)sizeof(T4);
    operand_mode<T>();
    __shared__ __attribute__((aligned(1024))) unsigned char stages[WD_STAGES * WD_STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long m0 = (long long)blockIdx.x * WD_M;
    const int wr = w >> 2, wc = w & 3;  // wave tile: rows wr*128.., columns wc*96..
    const int fj = lane & 15, fg = lane >> 4;
    // ---- DMA sources: request q = 5 w + i covers 16 rows (A rows 16q.. for q < 16, Wt rows 16(q-16).. otherwise); lane
    // (row 16q' + lane/4, slot lane%4) fetches the piece that gswz puts in that slot
    dtk_u4 srd[WD_REQ];
    unsigned voff[WD_REQ];
#pragma unroll
    for (int i = 0; i < WD_REQ; ++i) {
        const int q = w * WD_REQ + i;
        const bool isA = q < WD_M / 16;
        const int row = (isA ? q : q - WD_M / 16) * 16 + (lane >> 2);
        const int piece = (lane & 3) ^ ((0x1230 >> (((row >> 2) & 3) * 4)) & 3);
        const int trow = isA ? (int)(min(m0 + row, M - 1) - m0) : row;  // Wt has exactly WD_N rows; rows past M repeat the last one
        srd[i] = dtk_make_srd(isA ? A + m0 * K : Wt);
        voff[i] = (unsigned)(trow * K + piece * 8) * 2u;
    }
    const unsigned lds0 = (unsigned)(size_t)&stages[0] + (unsigned)w * (WD_REQ * 1024);
    const int nk = K / GK;
    auto issue = [&](int ks, int buf) {
        const int kk = min(ks, nk - 1);  // past the end: a harmless repeat keeps the request count per stage uniform
        wd_issue<WD_REQ>(srd, voff, (unsigned)kk * (GK * 2), __builtin_amdgcn_readfirstlane(lds0 + buf * WD_STAGE_BYTES));
    };
    f4 acc[8][6];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 6; ++ni) acc[mi][ni] = f4{0.f, 0.f, 0.f, 0.f};
    // fragment addresses: row (tile row 16 mi + fj) -> uint4 index row * 4 + (fg ^ f(row)); f depends on fj only
    const int fsw = (0x1230 >> (((fj >> 2) & 3) * 4)) & 3;
    const unsigned a_off = ((wr * 128 + fj) * 4 + (fg ^ fsw)) * 16;
    const unsigned b_off = WD_A_BYTES + ((wc * 96 + fj) * 4 + (fg ^ fsw)) * 16;
    issue(0, 0);
    issue(1, 1);
    ws_wait<WD_REQ>();  // stage 0 landed (stage 1 may still fly)
    __syncthreads();
    int buf = 0;
    for (int ks = 0; ks < nk; ++ks) {
        const int nxt2 = buf == 0 ? 2 : buf - 1;  // (buf + 2) % 3: the stage consumed in the previous iteration
        issue(ks + 2, nxt2);
        const unsigned char* sb = stages + buf * WD_STAGE_BYTES;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            T8 af[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                af[mi] = *reinterpret_cast<const T8*>(sb + a_off + (half * 4 + mi) * 1024);
#pragma unroll
            for (int ni = 0; ni < 6; ++ni) {
                const T8 bfr = *reinterpret_cast<const T8*>(sb + b_off + ni * 1024);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[half * 4 + mi][ni] =
                        mfma16(bfr, af[mi], acc[half * 4 + mi][ni]);   // (W tile) x (token tile)^T: D transposed (round 5), see the epilogue
            }
        }
        ws_wait<WD_REQ>();  // stage ks + 1 landed; the requests of ks + 2 stay in flight
        __syncthreads();
        buf = buf == 2 ? 0 : buf + 1;
    }
    ws_wait<0>();
    // D tiles are TRANSPOSED (round 5): lane (fg, fj) holds features 4 fg + r (r = 0..3) of token fj of each 16 x 16 tile -- four
    // consecutive features of one token: ONE 8-byte store per tile and lane (48 per lane and 256 x 384 tile) where the
    // token-major form needed four 2-byte stores (192)
    if (STAGED && FUSE_LN) {
        // Round 6: fc2's epilogue + the NEXT block's LayerNorm.  Two passes; pass p stages the token tiles mi = 4 p .. 4 p + 3 of EVERY
        // wave (rows wr 128 + 64 p .. + 63 of the tile: every wave frees half of its accumulators per pass, which is what leaves
        // registers for the rows of x below) as the same 16-bit delta the unfused path stores.  Then a wave takes 16 of the pass's 128
        // rows, eight at a time: x of the eight rows is requested first, then row by row x += delta, x written back, statistics and
        // the normalised 16-bit row by layernorm_kernel's own expressions and lane -> column map (lane c: columns 4 c .. 4 c + 3, lanes
        // 0-31 also 256 + 4 c ..): bit-identical rows.  The delta tile never reaches memory; one launch and one trip of x + delta per
        // block go away.
        static_assert(128 * WD_OPITCH <= WD_STAGES * WD_STAGE_BYTES, "staged half tile must fit the stages");
        const float4 ga = *reinterpret_cast<const float4*>(e.ln_w + lane * 4), ba = *reinterpret_cast<const float4*>(e.ln_b + lane * 4);
        const float4 gb = *reinterpret_cast<const float4*>(e.ln_w + 256 + (lane & 31) * 4), bb = *reinterpret_cast<const float4*>(e.ln_b + 256 + (lane & 31) * 4);
        bool sat = false;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __syncthreads();   // p = 0: every wave's requests have landed and the stages are dead; p = 1: pass 0 has been read
#pragma unroll
            for (int ni = 0; ni < 6; ++ni) {
                const int nb = wc * 96 + ni * 16 + fg * 4;
                const float4 b4 = e.bias ? *reinterpret_cast<const float4*>(e.bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 g4 = *reinterpret_cast<const float4*>(e.gamma + nb);
#pragma unroll
                for (int mq = 0; mq < 4; ++mq) {
                    const f4& a = acc[4 * p + mq][ni];
                    *reinterpret_cast<T4*>(stages + (wr * 64 + mq * 16 + fj) * WD_OPITCH + nb * 2) =
                        T4{(T)(g4.x * (a[0] + b4.x)), (T)(g4.y * (a[1] + b4.y)), (T)(g4.z * (a[2] + b4.z)), (T)(g4.w * (a[3] + b4.w))};
                }
            }
            __syncthreads();
            // local rows lr0 .. lr0 + NB - 1 of the pass <-> tile rows (lr >> 6) 128 + 64 p + (lr & 63); NB rows of x in flight per wave:
            // 8 while the second half of the accumulators is live (pass 0), 16 in pass 1
            auto ln_rows = [&](auto nb_c, int lr0) {
                constexpr int NB = decltype(nb_c)::value;
                const long long rb = m0 + (lr0 >> 6) * 128 + p * 64 + (lr0 & 63);
                float4 xa[NB], xb[NB];
#pragma unroll
                for (int rr = 0; rr < NB; ++rr) {
                    const float* xp = e.ln_x + min(rb + rr, M - 1) * WD_N + lane * 4;
                    xa[rr] = *reinterpret_cast<const float4*>(xp);
                    xb[rr] = lane < 32 ? *reinterpret_cast<const float4*>(xp + 256) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int rr = 0; rr < NB; ++rr) {
                    const long long row = rb + rr;
                    if (row >= M) continue;   // wave-uniform (the last tile's tail)
                    const unsigned char* sp = stages + (lr0 + rr) * WD_OPITCH + lane * 8;
                    float4 v[2] = {xa[rr], xb[rr]};
                    float s = 0.f;
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        if (it == 0 || lane < 32) {
                            const T4 d = *reinterpret_cast<const T4*>(sp + it * 512);
                            const float d0 = (float)d[0], d1 = (float)d[1], d2 = (float)d[2], d3 = (float)d[3];
                            if (IsF16<T>::value) sat |= !(fmaxf(fmaxf(fabsf(d0), fabsf(d1)), fmaxf(fabsf(d2), fabsf(d3))) < 65504.f);
                            v[it].x += d0; v[it].y += d1; v[it].z += d2; v[it].w += d3;
                            *reinterpret_cast<float4*>(e.ln_x + row * WD_N + it * 256 + lane * 4) = v[it];
                            s += (v[it].x + v[it].y) + (v[it].z + v[it].w);
                        }
                    }
                    const float mean = wave_sum(s) / (float)WD_N;
                    float q = 0.f;
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        if (it == 0 || lane < 32) {
                            const float a = v[it].x - mean, b = v[it].y - mean, cc = v[it].z - mean, d = v[it].w - mean;
                            q += (a * a + b * b) + (cc * cc + d * d);
                        }
                    }
                    const float rstd = rsqrtf(wave_sum(q) / (float)WD_N + e.ln_eps);
                    T* o = e.ln_out + row * WD_N;
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        if (it == 0 || lane < 32) {
                            const float4 g = it ? gb : ga, b = it ? bb : ba;
                            T4 r = {(T)((v[it].x - mean) * rstd * g.x + b.x), (T)((v[it].y - mean) * rstd * g.y + b.y),
                                     (T)((v[it].z - mean) * rstd * g.z + b.z), (T)((v[it].w - mean) * rstd * g.w + b.w)};
                            *reinterpret_cast<T4*>(o + it * 256 + lane * 4) = r;
                        }
                    }
                }
            };
            if (p == 0) {
                ln_rows(std::integral_constant<int, 8>{}, w * 16);
                ln_rows(std::integral_constant<int, 8>{}, w * 16 + 8);
            } else {
                ln_rows(std::integral_constant<int, 16>{}, w * 16);
            }
        }
        if (IsF16<T>::value && e.ln_ovf && __any(sat) && lane == 0) atomicOr(e.ln_ovf, 1);
        return;
    }
    if (STAGED) {
        // Round 6: through LDS (see gemm_wide_kernel's epilogue: a CU holds one workgroup of this kernel, nothing overlaps the epilogue,
        // and its 8-byte stores -- 16 tokens x 32 B per instruction -- ran at 1.5 TB/s).  The 256 x 384 tile is 192 KB of 16-bit values:
        // two passes of 128 rows (the waves of row half p stage, everybody writes out: the 128 rows are 96 KB of CONTIGUOUS memory).
        static_assert(128 * WD_OPITCH <= WD_STAGES * WD_STAGE_BYTES, "staged half tile must fit the stages");
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __syncthreads();   // p = 0: every wave's requests have landed and the stages are dead; p = 1: pass 0 has been read out
            if (wr == p) {
#pragma unroll
                for (int ni = 0; ni < 6; ++ni) {
                    const int nb = wc * 96 + ni * 16 + fg * 4;
                    const float4 b4 = e.bias ? *reinterpret_cast<const float4*>(e.bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 g4 = *reinterpret_cast<const float4*>(e.gamma + nb);
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi) {
                        const f4& a = acc[mi][ni];
                        *reinterpret_cast<T4*>(stages + (mi * 16 + fj) * WD_OPITCH + nb * 2) =
                            T4{(T)(g4.x * (a[0] + b4.x)), (T)(g4.y * (a[1] + b4.y)), (T)(g4.z * (a[2] + b4.z)), (T)(g4.w * (a[3] + b4.w))};
                    }
                }
            }
            __syncthreads();
            const long long mb = m0 + p * 128;
#pragma unroll 4
            for (int it = 0; it < 128 * (WD_N / 8) / 512; ++it) {   // 6144 16-byte pieces, 12 per thread
                const int idx = it * 512 + tid, row = idx / (WD_N / 8), piece = idx - row * (WD_N / 8);
                const unsigned char* sp = stages + row * WD_OPITCH + piece * 16;
                const uint2 lo = *reinterpret_cast<const uint2*>(sp), hi = *reinterpret_cast<const uint2*>(sp + 8);
                if (mb + row < M) *reinterpret_cast<uint4*>(e.delta + (mb + row) * WD_N + piece * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
        return;
    }
#pragma unroll
    for (int ni = 0; ni < 6; ++ni) {
        const int nb = wc * 96 + ni * 16 + fg * 4;
        const float4 b4 = e.bias ? *reinterpret_cast<const float4*>(e.bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 g4 = *reinterpret_cast<const float4*>(e.gamma + nb);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const long long m = m0 + wr * 128 + mi * 16 + fj;
            const f4& a = acc[mi][ni];
            if (m < M)
                *reinterpret_cast<T4*>(e.delta + m * WD_N + nb) =
                    T4{(T)(g4.x * (a[0] + b4.x)), (T)(g4.y * (a[1] + b4.y)), (T)(g4.z * (a[2] + b4.z)), (T)(g4.w * (a[3] + b4.w))};
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the same pipeline as a 256 x 256 tile for the GEMMs of wider models (N a multiple of 256: D = 768 / 1024 and their qkv
// / MLP widths), all epilogues.  Stage = 16 KB of A + 16 KB of B, ring of FOUR stages with three in flight (96 KB per
// CU); wave grid 2 x 4, wave tile 128 x 64 = 8 x 4 MFMA tiles (128 accumulator registers).  Block order as in
// gemm_tiled_kernel: the column tiles of a row block run back to back on one XCD.
// ---------------------------------------------------------------------------------------------------------------
constexpr int W2_M = 256, W2_N = 256, W2_STAGES = 4, W2_STAGE_BYTES = (W2_M + W2_N) * 64;
constexpr int W2_REQ = (W2_M + W2_N) / 16 / 8;  // 4 DMA requests per wave and stage
constexpr int W2_OPITCH = W2_N * 2 + 8;          // staged output rows: 512 B + 8 (the 8-byte writes of 16 tokens fall on 16 different bank pairs)
constexpr int W2_LDS_BYTES = W2_STAGES * W2_STAGE_BYTES > W2_M * W2_OPITCH ? W2_STAGES * W2_STAGE_BYTES : W2_M * W2_OPITCH;

inline unsigned gemm_wide_grid(int N, long long rows) {
    const long long ncol = N / W2_N, nrow = dtk_cdiv(rows, W2_M);
    return (unsigned)(dtk_cdiv(nrow, 8) * 8 * ncol);
}

template <typename T, int EPI, bool PIPE = true>
__global__ __launch_bounds__(512, 2) void gemm_wide_kernel(const T* __restrict__ A, const T* __restrict__ Wt,
                                                           long long M, int N, int K, GemmEpi<T> e) {
    typedef typename Vec<T>::t8 T8;
    typedef typename Vec<T>::t4 T4;
    (void)sizeof(T8); (void)sizeof(T4);
    operand_mode<T>();
    __shared__ __attribute__((aligned(1024))) unsigned char stages[W2_LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncol = N / W2_N;
    const long long nrow = (M + W2_M - 1) / W2_M;
    const long long kb = blockIdx.x >> 3;
    const long long row_blk = (kb / ncol) * 8 + (blockIdx.x & 7);
    if (row_blk >= nrow) return;
    const long long m0 = row_blk * W2_M;
    const int n0 = (int)(kb % ncol) * W2_N;
    const int wr = w >> 2, wc = w & 3;  // wave tile: rows wr*128.., columns wc*64..
    const int fj = lane & 15, fg = lane >> 4;
    dtk_u4 srd[W2_REQ];
    unsigned voff[W2_REQ];
#pragma unroll
    for (int i = 0; i < W2_REQ; ++i) {
        const int q = w * W2_REQ + i;  // 0..15: A rows 16q.., 16..31: Wt rows n0 + 16(q-16)..
        const bool isA = q < W2_M / 16;
        const int row = (isA ? q : q - W2_M / 16) * 16 + (lane >> 2);
        const int piece = (lane & 3) ^ ((0x1230 >> (((row >> 2) & 3) * 4)) & 3);
        const int trow = isA ? (int)(min(m0 + row, M - 1) - m0) : row;
        srd[i] = dtk_make_srd(isA ? A + m0 * K : Wt + (long long)n0 * K);
        voff[i] = (unsigned)(trow * K + piece * 8) * 2u;
    }
    const unsigned lds0 = (unsigned)(size_t)&stages[0] + (unsigned)w * (W2_REQ * 1024);
    const int nk = DTK_DBG(e.no_store, 8) ? 0 : K / GK;
    auto issue = [&](int ks, int buf) {
        const int kk = DTK_DBG(e.no_store, 16 | 8) ? 0 : min(ks, nk - 1);
        wd_issue<W2_REQ>(srd, voff, (unsigned)kk * (GK * 2), __builtin_amdgcn_readfirstlane(lds0 + buf * W2_STAGE_BYTES));
    };
    f4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f4{0.f, 0.f, 0.f, 0.f};
    const int fsw = (0x1230 >> (((fj >> 2) & 3) * 4)) & 3;
    const unsigned a_off = ((wr * 128 + fj) * 4 + (fg ^ fsw)) * 16;
    const unsigned b_off = W2_M * 64 + ((wc * 64 + fj) * 4 + (fg ^ fsw)) * 16;
    issue(0, 0);
    issue(1, 1);
    issue(2, 2);
    ws_wait<2 * W2_REQ>();  // stage 0 landed
    __syncthreads();
    int buf = 0;
    if (PIPE) {
        // Round 6: the fragment reads run ONE HALF-STEP AHEAD of the MFMAs that consume them.  In the form below every wave reads
        // its eight fragments right behind the barrier -- all eight waves of the CU at once, 64 ds_read_b128 = 256 LDS cycles plus
        // the latency, with the matrix pipes idle (SQ counters, fc2 of ViT-S on the same loop: waves parked 45 %, pipes 38 % busy).
        // Here a k-step is two halves of 16 MFMAs (token tiles 0-3 | 4-7 against the four W tiles); the A fragments of the second
        // half are requested in front of the first half's MFMAs, and A (first half) + W fragments of the NEXT stage behind the
        // barrier, in front of the second half's MFMAs: two fragment sets (fa / fb, alternating with the k-step: the loop is
        // unrolled by two so that the set is a compile-time index), 64 fragment registers + 128 accumulators.
        T8 fa[2][4], fb[2][4], ga[4];
        {
            const unsigned char* sb = stages;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[0][i] = *reinterpret_cast<const T8*>(sb + a_off + i * 1024);
                fb[0][i] = *reinterpret_cast<const T8*>(sb + b_off + i * 1024);
            }
        }
        for (int ks = 0; ks < nk; ks += 2) {   // (nk is even: K % 256 == 0 on this path)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                issue(ks + s + 3, (buf + 3) & 3);  // the stage consumed in the previous step
                const unsigned char* sb = stages + buf * W2_STAGE_BYTES;
                const unsigned char* sn = stages + ((buf + 1) & 3) * W2_STAGE_BYTES;
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) ga[mi] = *reinterpret_cast<const T8*>(sb + a_off + (4 + mi) * 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = mfma16(fb[s][ni], fa[s][mi], acc[mi][ni]);
                __builtin_amdgcn_sched_barrier(0);
                // (this wave's reads of the current stage have returned before it passes the barrier: the DMA requests of the next
                //  step overwrite the stage that was current one step earlier)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                ws_wait<2 * W2_REQ>();  // stage ks + 1 landed; ks + 2 and ks + 3 stay in flight
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 4; ++i) {   // (behind the last step: the repeated last stage, harmless)
                    fa[s ^ 1][i] = *reinterpret_cast<const T8*>(sn + a_off + i * 1024);
                    fb[s ^ 1][i] = *reinterpret_cast<const T8*>(sn + b_off + i * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) acc[4 + mi][ni] = mfma16(fb[s][ni], ga[mi], acc[4 + mi][ni]);
                __builtin_amdgcn_sched_barrier(0);
                buf = (buf + 1) & 3;
            }
        }
    } else {
    for (int ks = 0; ks < nk; ++ks) {
        issue(ks + 3, (buf + 3) & 3);  // the stage consumed in the previous iteration
        const unsigned char* sb = stages + buf * W2_STAGE_BYTES;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            T8 af[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                af[mi] = *reinterpret_cast<const T8*>(sb + a_off + (half * 4 + mi) * 1024);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const T8 bfr = *reinterpret_cast<const T8*>(sb + b_off + ni * 1024);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[half * 4 + mi][ni] =
                        mfma16(bfr, af[mi], acc[half * 4 + mi][ni]);   // (W tile) x (token tile)^T: D transposed, see the epilogue
            }
        }
        ws_wait<2 * W2_REQ>();  // stage ks + 1 landed; ks + 2 and ks + 3 stay in flight
        __syncthreads();
        buf = (buf + 1) & 3;
    }
    }
    ws_wait<0>();
    // D tiles are TRANSPOSED (the MFMAs above multiply (W tile) x (token tile)^T): lane (fg, fj) holds features 4 fg + r of token fj
    float amax = 0.f;
    if (PIPE && EPI != EPI_QKV) {
        // Round 6: the [M][N] epilogues leave through LDS.  A store instruction of the direct form below covers 16 tokens x 32 bytes --
        // sixteen quarter lines; with everything but the epilogue switched off (DTK_DEV ablation) the stores of fc1 at D = 1024 ran at
        // 1.5 TB/s and cost a third of the kernel, because a CU holds ONE workgroup of this kernel (128 KB of stages) and nothing
        // overlaps its epilogue.  The stages are dead here: the 256 x 256 tile is staged as 16-bit values (row pitch 520 B) and leaves
        // as whole 512-byte rows, 16 bytes per lane, two rows per wave and instruction.
        __syncthreads();   // every wave's requests have landed (the wait above) and every wave is done with the stages
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int cb = wc * 64 + ni * 16 + fg * 4;
            const float4 b4 = e.bias ? *reinterpret_cast<const float4*>(e.bias + n0 + cb) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 g4 = EPI == EPI_DELTA ? *reinterpret_cast<const float4*>(e.gamma + n0 + cb) : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
                *reinterpret_cast<T4*>(stages + (wr * 128 + mi * 16 + fj) * W2_OPITCH + cb * 2) =
                    gemm_value_tile_t<T, EPI>(acc[mi][ni], b4, g4, amax);
        }
        __syncthreads();
        T* const outp = (EPI == EPI_GELU ? e.out : e.delta) + n0 + (tid & 31) * 8;
#pragma unroll 4
        for (int it = 0; it < W2_M / 16; ++it) {
            const int row = it * 16 + (tid >> 5);
            const unsigned char* sp = stages + row * W2_OPITCH + (tid & 31) * 16;
            const uint2 lo = *reinterpret_cast<const uint2*>(sp), hi = *reinterpret_cast<const uint2*>(sp + 8);
            if (m0 + row < M && !DTK_DBG(e.no_store, 4)) *reinterpret_cast<uint4*>(outp + (m0 + row) * N) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    } else if (PIPE && EPI == EPI_QKV && n0 < 2 * e.D) {
        // Q and K tiles the same way (a 256-feature tile is four heads of ONE of q / k / v: D is a multiple of 256 on this path): a
        // token's 64 features of a head are 128 contiguous bytes of q / k [frame][head][position][64].  V^T: the next branch.
        __syncthreads();
        const int which = n0 / e.D, head0 = (n0 - which * e.D) >> 6;
        const float sc = which == 0 ? e.qscale : 1.f;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int cb = wc * 64 + ni * 16 + fg * 4;
            const float4 b4 = e.bias ? *reinterpret_cast<const float4*>(e.bias + n0 + cb) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const f4& a = acc[mi][ni];
                const float v0 = (a[0] + b4.x) * sc, v1 = (a[1] + b4.y) * sc, v2 = (a[2] + b4.z) * sc, v3 = (a[3] + b4.w) * sc;
                if (IsF16<T>::value) amax = amax2(amax2(amax, v0, v1), v2, v3);
                *reinterpret_cast<T4*>(stages + (wr * 128 + mi * 16 + fj) * W2_OPITCH + cb * 2) = T4{(T)v0, (T)v1, (T)v2, (T)v3};
            }
        }
        __syncthreads();
        T* const qk = (which == 0 ? e.q : e.k) + (tid & 7) * 8;
        const int hh = head0 + ((tid & 31) >> 3);
#pragma unroll 4
        for (int it = 0; it < W2_M / 16; ++it) {
            const int row = it * 16 + (tid >> 5);
            const unsigned char* sp = stages + row * W2_OPITCH + (tid & 31) * 16;
            const uint2 lo = *reinterpret_cast<const uint2*>(sp), hi = *reinterpret_cast<const uint2*>(sp + 8);
            const long long m = m0 + row;
            if (m < M && !DTK_DBG(e.no_store, 4)) {
                const unsigned f = (unsigned)m / (unsigned)e.S, pos = (unsigned)m - f * (unsigned)e.S;   // (M < 2^31 tokens)
                *reinterpret_cast<uint4*>(qk + (((size_t)f * e.heads + hh) * e.Sp + pos) * 64) = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    } else if (PIPE && EPI == EPI_QKV) {
        // V^T tiles: vt[frame][head][feature][position] keeps TOKENS contiguous, so the tile is staged transposed -- sT[feature][token],
        // 16-bit, the same pitch: a lane writes its four features of a token as four 2-byte pieces (the 16 tokens of a piece-write share
        // 8 dwords; the four feature groups of a wave fall on different banks) -- and leaves as 8-byte pieces of four tokens, 64 lanes =
        // 512 contiguous bytes of one feature row, when positions come in fours (S and Sp multiples of 4: 8108 / 8192 at 854 x 476);
        // otherwise (odd test sizes) element by element.  The direct form wrote 2-byte pieces, 16 tokens x 4 rows per instruction, and
        // made a V^T tile's epilogue 2.7 x a Q / K tile's (ViT-L qkv: 39.7 us per tile on average against proj's 34.4).
        __syncthreads();
        const int head0 = (n0 - 2 * e.D) >> 6;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int cb = wc * 64 + ni * 16 + fg * 4;
            const float4 b4 = e.bias ? *reinterpret_cast<const float4*>(e.bias + n0 + cb) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const f4& a = acc[mi][ni];
                const float v[4] = {a[0] + b4.x, a[1] + b4.y, a[2] + b4.z, a[3] + b4.w};
                if (IsF16<T>::value) amax = amax2(amax2(amax, v[0], v[1]), v[2], v[3]);
                unsigned char* sp = stages + cb * W2_OPITCH + (wr * 128 + mi * 16 + fj) * 2;
#pragma unroll
                for (int r = 0; r < 4; ++r) *reinterpret_cast<T*>(sp + r * W2_OPITCH) = (T)v[r];
            }
        }
        __syncthreads();
        if (((e.S | e.Sp) & 3) == 0) {
            const int g = tid & 63;   // tokens 4 g .. 4 g + 3 of the tile: one frame (frames start at multiples of 4), all below M or none
            const long long m = m0 + 4 * g;
            if (m < M && !DTK_DBG(e.no_store, 4)) {
                const unsigned f = (unsigned)m / (unsigned)e.S, pos = (unsigned)m - f * (unsigned)e.S;
                T* const base = e.vt + (((size_t)f * e.heads + head0) * 64) * e.Sp + pos;   // feature c of the tile: + c Sp
#pragma unroll 4
                for (int it = 0; it < W2_N / 8; ++it) {
                    const int c = it * 8 + (tid >> 6);
                    *reinterpret_cast<uint2*>(base + (size_t)c * e.Sp) = *reinterpret_cast<const uint2*>(stages + c * W2_OPITCH + g * 8);
                }
            }
        } else {
            const int t = tid & 255;
            const long long m = m0 + t;
            if (m < M && !DTK_DBG(e.no_store, 4)) {
                const unsigned f = (unsigned)m / (unsigned)e.S, pos = (unsigned)m - f * (unsigned)e.S;
                T* const base = e.vt + (((size_t)f * e.heads + head0) * 64) * e.Sp + pos;
#pragma unroll 4
                for (int it = 0; it < W2_N / 2; ++it) {
                    const int c = it * 2 + (tid >> 8);
                    base[(size_t)c * e.Sp] = *reinterpret_cast<const T*>(stages + c * W2_OPITCH + t * 2);
                }
            }
        }
    } else {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int nb = n0 + wc * 64 + ni * 16 + fg * 4;
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
                gemm_store_tile_t<T, EPI>(acc[mi][ni], m0 + wr * 128 + mi * 16 + fj, nb, M, N, e, amax);
        }
    }
    amax_report<T, EPI>(amax, e.ovf);
}

// V2M (round 4, default 3): bit 0 = weights in AGPRs (loaded there by asm; the builtin MFMA takes them from there as they are)
// and zero accumulators through the MFMA's C operand; bit 1 = token tiles by LDS-DMA through a buffer descriptor (scalar tile
// offset + constant per-lane offset: 3 issue slots per request instead of ~13).  V2M = 0 is the round 1-3 form, kept for the
// A / B measurement (DTK_VIT_GEMM_WS_V1).  Same-box A / B: proj 4.97 -> 3.62 ms, qkv 12.34 -> 12.10, fc1 14.68 -> 14.55 ms per step.
template <typename T, int EPI, int V2M = 3>
__global__ __launch_bounds__(256) void gemm_ws_kernel(const T* __restrict__ A, const T* __restrict__ Wt,
                                                      long long M, int N, GemmEpi<T> e, int tiles_per_chunk) {
    typedef typename Vec<T>::t8 T8;
    typedef typename Vec<T>::t4 T4;
    (void)sizeof(T8); (void)sizeof(T4);
    constexpr bool V2 = (V2M & 1) != 0;       // AGPR weights + zero C operand
    constexpr bool V2D = (V2M & 2) != 0;      // descriptor LDS-DMA
    operand_mode<T>();
    __shared__ __attribute__((aligned(1024))) unsigned char toks[3][WS_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char stage[4][2][WS_UNIT_BYTES];  // per wave: two staging units
    __shared__ __attribute__((aligned(16))) float s_bias[4][64], s_gamma[4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * WS_COLS + w * 64;
    const long long total_tiles = (M + WS_ROWS - 1) / WS_ROWS;
    const long long tile0 = (long long)blockIdx.y * tiles_per_chunk;
    if (tile0 >= total_tiles) return;
    const int NT = (int)((total_tiles - tile0) < tiles_per_chunk ? (total_tiles - tile0) : tiles_per_chunk);
    // weights: MFMA row i of row tile t carries output feature n0 + 32 t + nl(i), nl chosen so that D row
    // (r & 3) + 8 (r >> 2) + 4 h  <->  feature 16 h + r
    const int nl = (j & 3) + 4 * (j >> 3) + 16 * ((j >> 2) & 1);
    T8 wf[2][WS_KS];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = min(n0 + t * 32 + nl, N - 1);
        const T* wp = Wt + (size_t)row * WS_K + h * 8;
#pragma unroll
        for (int ks = 0; ks < WS_KS; ++ks) {
            if (V2) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wf[t][ks]) : "v"(wp + ks * 16) : "memory");
            else wf[t][ks] = *reinterpret_cast<const T8*>(wp + ks * 16);
        }
    }
    if (V2) {
        // the loads above are invisible to the compiler's own vmcnt bookkeeping: wait, then re-define every loaded register behind
        // the wait (an empty volatile asm with a "+a" operand emits nothing, but volatile asms keep their order and no use of
        // wf can be scheduled above its re-definition -- ADVICE r4: before, only scheduling luck kept uses behind the wait)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < WS_KS; ++ks) asm volatile("" : "+a"(wf[t][ks]));
    }
    {
        const int n = min(n0 + lane, N - 1);
        s_bias[w][lane] = e.bias ? e.bias[n] : 0.f;
        s_gamma[w][lane] = (EPI == EPI_DELTA) ? e.gamma[n] : 1.f;
    }
    // QKV: a 64-feature group is one head of q, k or v
    const int which = (EPI == EPI_QKV) ? min(n0, N - 1) / e.D : 0;  // (the last column group of N = 1152 / 384 has idle waves)
    const int head = (EPI == EPI_QKV) ? (min(n0, N - 1) - which * e.D) >> 6 : 0;
    const float qsc = (EPI == EPI_QKV && which == 0) ? e.qscale : 1.f;
    // this lane's token of the tile whose epilogue runs next, as (frame, position) for the QKV layouts
    long long m_ep = tile0 * WS_ROWS + j;
    int f_ep = 0, s_ep = 0;
    if (EPI == EPI_QKV) { f_ep = (int)(m_ep / e.S); s_ep = (int)(m_ep - (long long)f_ep * e.S); }
    // token tiles: a DMA request fetches 4 tokens x 256 contiguous bytes (8 cache lines; one token row per lane would be
    // 32).  Region (g, c) of 1 KB holds tokens 4g .. 4g+3, pieces 16c .. 16c+15 (piece = 8 k values); piece pp of token
    // tt sits at slot ((pp + g) & 15) * 4 + tt, which keeps the fragment reads of 16 consecutive lanes conflict-free
    const unsigned lds_base = (unsigned)(size_t)&toks[0][0];
    const int l_tt = lane & 3, l_pp = lane >> 2;
    // V2: descriptor over the chunk's first token row (32-bit offsets inside a chunk: tiles_per_chunk x 24 KB)
    const dtk_u4 srd = dtk_make_srd(A + tile0 * WS_ROWS * WS_K);
    // (rows of the LAST token tile past M - 1 read row M - 1, as the 64-bit form did: a second pair of offsets for that tile)
    unsigned a_voff[2], a_voff_last[2];
    const int last_rows = (int)(M - (total_tiles - 1) * WS_ROWS);   // valid rows of the last tile (1 .. 32)
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
        const int g = 2 * w + gg;
        a_voff[gg] = (unsigned)(((4 * g + l_tt) * WS_K + ((l_pp - g) & 15) * 8) * 2);
        a_voff_last[gg] = (unsigned)((min(4 * g + l_tt, last_rows - 1) * WS_K + ((l_pp - g) & 15) * 8) * 2);
    }
    const unsigned a_dst = __builtin_amdgcn_readfirstlane(lds_base + (2 * w) * 3 * 1024);
    auto issue = [&](int n, int buf) {
        if (V2D) {
            const unsigned toff = __builtin_amdgcn_readfirstlane((unsigned)n * (unsigned)WS_TILE_BYTES);
            const unsigned dst = __builtin_amdgcn_readfirstlane(a_dst + (unsigned)buf * (unsigned)WS_TILE_BYTES);
            const bool last = tile0 + n == total_tiles - 1;
            const unsigned v0 = last ? a_voff_last[0] : a_voff[0], v1 = last ? a_voff_last[1] : a_voff[1];
            dtk_buffer_lds16<0>(srd, toff, v0, dst);
            dtk_buffer_lds16<1024>(srd, toff + 256u, v0, dst);
            dtk_buffer_lds16<2048>(srd, toff + 512u, v0, dst);
            dtk_buffer_lds16<3072>(srd, toff, v1, dst);
            dtk_buffer_lds16<4096>(srd, toff + 256u, v1, dst);
            dtk_buffer_lds16<5120>(srd, toff + 512u, v1, dst);
            return;
        }
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const int g = 2 * w + gg;
            const long long m = min((tile0 + n) * WS_ROWS + 4 * g + l_tt, M - 1);
            const T* gp = A + m * WS_K + ((l_pp - g) & 15) * 8;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                ws_glds16(gp + c * 128, __builtin_amdgcn_readfirstlane(lds_base + buf * WS_TILE_BYTES + (g * 3 + c) * 1024));
        }
    };
    // B-operand fragment of k-step ks for lane (token j, half h): piece 2 ks + h -> region column c = ks >> 3
    unsigned frag_off[8];
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8)
        frag_off[k8] = (unsigned)((j >> 2) * 3 * 1024 + ((((2 * k8 + h + (j >> 2)) & 15) * 4 + (j & 3)) * 16));
    // epilogue of 8 values (row tile t, half hv of this lane's 16 features): features nb .. nb + 7
    float amax = 0.f;
    auto epi8 = [&](const f16v (&acc)[2], int t, int hv) {
        const int fl = 32 * t + 16 * h + 8 * hv;  // feature offset inside the wave's 64
        const int nb = n0 + fl;
        const float4 b0 = *reinterpret_cast<const float4*>(&s_bias[w][fl]), b1 = *reinterpret_cast<const float4*>(&s_bias[w][fl + 4]);
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = acc[t][8 * hv + r];
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        const bool ok = m_ep < M && nb < N && !DTK_DBG(e.no_store, 3);
        if (IsF16<T>::value && (EPI == EPI_QKV || EPI == EPI_GELU)) {   // the fp16 range, every value of every frame (GemmEpi::ovf)
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                if (EPI == EPI_GELU) amax = fmaxf(fmaxf(amax, v[r]), v[r + 1]);   // the stored GELU(v): the positive part of v
                else amax = amax2(amax, v[r] * qsc, v[r + 1] * qsc);
            }
        }
        if (EPI == EPI_GELU) {
            T8 o;
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                const f2 g = gelu2(f2{v[r], v[r + 1]});
                o[r] = (T)g[0];
                o[r + 1] = (T)g[1];
            }
            // (staging the GELU output for whole-line stores measured slower than these scattered 16-byte stores)
            if (ok) *reinterpret_cast<T8*>(e.out + m_ep * N + nb) = o;
        } else if (EPI == EPI_DELTA) {
            const float4 g0 = *reinterpret_cast<const float4*>(&s_gamma[w][fl]), g1 = *reinterpret_cast<const float4*>(&s_gamma[w][fl + 4]);
            T8 o = {(T)(v[0] * g0.x), (T)(v[1] * g0.y), (T)(v[2] * g0.z), (T)(v[3] * g0.w),
                     (T)(v[4] * g1.x), (T)(v[5] * g1.y), (T)(v[6] * g1.z), (T)(v[7] * g1.w)};
            *reinterpret_cast<T8*>(&stage[w][0][0] + j * WS_PITCH + fl * 2) = o;
        } else {
            const int dh = fl;  // 0..63 inside the head
            if (which == 2) {
                // V^T[f][head][dh][s]: consecutive lanes are consecutive tokens
                // (round 6, measured and dropped: staging the wave's 64 x 32 tile transposed in LDS and flushing 8-byte pieces of four
                //  tokens -- 8 store instructions instead of 32 -- made qkv 12.3 -> 15.7 ms per step: the 32 two-byte LDS writes sit in
                //  the lone wave's issue stream like the stores they replace; docs/NEGATIVE_RESULTS.md)
                if (ok) {
                    T* vp = e.vt + (((size_t)f_ep * e.heads + head) * 64 + dh) * e.Sp + s_ep;
#pragma unroll
                    for (int r = 0; r < 8; ++r) vp[(size_t)r * e.Sp] = (T)v[r];
                }
            } else {
                T8 o;
#pragma unroll
                for (int r = 0; r < 8; ++r) o[r] = (T)(v[r] * qsc);
                *reinterpret_cast<T8*>(&stage[w][0][0] + j * WS_PITCH + fl * 2) = o;
            }
        }
    };
    // staged rows -> global memory with 8 lanes per 128-byte output row (whole lines).  (mt_fl, f0_fl, s0_fl): first
    // token of the staged tile and its (frame, position)
    long long mt_fl = tile0 * WS_ROWS;
    int f0_fl = 0, s0_fl = 0;
    if (EPI == EPI_QKV) { f0_fl = (int)(mt_fl / e.S); s0_fl = (int)(mt_fl - (long long)f0_fl * e.S); }
    auto flush = [&]() {
        if (EPI != EPI_GELU && n0 < N && !DTK_DBG(e.no_store, 3) && !(EPI == EPI_QKV && which == 2)) {
#pragma unroll
            for (int u = 0; u < 1; ++u) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int row = 8 * p + (lane >> 3), piece = lane & 7;
                    const uint4 val = *reinterpret_cast<const uint4*>(&stage[w][u][0] + row * WS_PITCH + piece * 16);
                    if (mt_fl + row < M) {
                        if (EPI == EPI_DELTA) {
                            *reinterpret_cast<uint4*>(e.delta + (mt_fl + row) * N + n0 + piece * 8) = val;
                        } else {
                            int f = f0_fl, sp = s0_fl + row;  // a tile crosses at most one frame end
                            if (sp >= e.S) { sp -= e.S; ++f; }
                            T* dst = which == 0 ? e.q : e.k;
                            *reinterpret_cast<uint4*>(dst + (((size_t)f * e.heads + head) * e.Sp + sp) * 64 + piece * 8) = val;
                        }
                    }
                }
            }
        }
        mt_fl += WS_ROWS;
        if (EPI == EPI_QKV) {
            s0_fl += WS_ROWS;
            if (s0_fl >= e.S) { s0_fl -= e.S; ++f0_fl; }
        }
    };
    auto advance = [&]() {  // the epilogue moves on to the next tile
        m_ep += WS_ROWS;
        if (EPI == EPI_QKV) {
            s_ep += WS_ROWS;
            if (s_ep >= e.S) { s_ep -= e.S; ++f_ep; }
        }
    };
    // one step: MFMAs of the tile in `buf` into accN, with the four epilogue pieces of the previous tile (accP) in between
    auto step = [&](int buf, f16v (&accN)[2], const f16v (&accP)[2], bool have_prev, bool have_flush) {
        if (!V2) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) accN[t][r] = 0.f;
        }
        const unsigned char* base = &toks[0][0] + buf * WS_TILE_BYTES;
        const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // inline constant C
        T8 b[3];
        b[0] = *reinterpret_cast<const T8*>(base + frag_off[0]);
        b[1] = *reinterpret_cast<const T8*>(base + frag_off[1]);
#pragma unroll
        for (int ks = 0; ks < WS_KS; ++ks) {
            if (ks + 2 < WS_KS)
                b[(ks + 2) % 3] = *reinterpret_cast<const T8*>(base + frag_off[(ks + 2) & 7] + ((ks + 2) >> 3) * 1024);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // (V2: the builtin, NOT an asm MFMA: its A operand accepts the AGPR the weight was loaded into as it is, and the
                // compiler keeps managing the instruction's hazards -- an asm MFMA followed by compiler-scheduled code that
                // re-used its B registers for an LDS load at once gave wrong tokens, non-deterministically)
                if (V2) accN[t] = mfma32(wf[t][ks], b[ks % 3], ks == 0 ? zero16 : accN[t]);
                else accN[t] = mfma32(wf[t][ks], b[ks % 3], accN[t]);
            }
            // the tile staged during the previous step leaves first: its stores have the whole step to retire
            if (have_flush && ks == 0) flush();
            if (have_prev && ks % 6 == 2) epi8(accP, (ks / 6) >> 1, (ks / 6) & 1);
        }
        if (have_prev) advance();
    };
    f16v accA[2], accB[2];
    // three LDS buffers, tiles requested two steps ahead.  The wait at the end of a step must prove that the tile of the
    // next step has landed: loads retire in order among themselves and the stores of the epilogue only add to the
    // count, so vmcnt(LQ) -- the requests of the tile after next -- is sufficient, if conservative.
    issue(0, 0);
    issue(min(1, NT - 1), 1);
    ws_wait<WS_LQ>();
    __syncthreads();
    int n = 0, b0 = 0;
    for (; n + 1 < NT; n += 2) {
        const int b1 = b0 == 2 ? 0 : b0 + 1, b2 = b1 == 2 ? 0 : b1 + 1;
        issue(min(n + 2, NT - 1), b2);
        step(b0, accA, accB, n > 0, n > 1);
        ws_wait<WS_LQ>();
        __syncthreads();
        issue(min(n + 3, NT - 1), b0);
        step(b1, accB, accA, true, n > 0);
        ws_wait<WS_LQ>();
        __syncthreads();
        b0 = b2;
    }
    if (n < NT) {  // NT odd: one more tile
        step(b0, accA, accB, n > 0, n > 1);
        if (n > 0) flush();
#pragma unroll
        for (int u = 0; u < 4; ++u) epi8(accA, u >> 1, u & 1);
    } else {
        if (n > 1) flush();
#pragma unroll
        for (int u = 0; u < 4; ++u) epi8(accB, u >> 1, u & 1);
    }
    flush();
    amax_report<T, EPI>(amax, e.ovf);
    ws_wait<0>();
}

#include "vit_split.h"

__global__ __launch_bounds__(256) void zero_kernel(uint4* __restrict__ p, long long n16) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0, 0, 0, 0);
}

// The LAST residual update of a pass, fused with what leaves the encoder (round 5): out = x + delta goes straight to tokens_out
// [F][S][D] and / or feat_out [F][S-1][D] (CLS dropped); the fp32 stream x is dead after the last block and is not written back.
// Replaces a statistics-free LayerNorm launch + a device copy / drop_cls pass: 2.8 GB of traffic instead of 5.0 per 90 frames.
template <typename T>
__global__ __launch_bounds__(256) void final_update_kernel(const float* __restrict__ x, const T* __restrict__ delta,
                                                           float* __restrict__ tokens_out, float* __restrict__ feat_out, int S, int D,
                                                           long long total4, int* __restrict__ overflow) {
    typedef typename Vec<T>::t4 T4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    bool sat = false;
    if (i < total4) {
        const int d4 = D / 4;
        const long long row = i / d4;  // over F * S
        const int c = (int)(i - row * d4);
        const long long f = row / S, s = row - f * S;
        float4 v = reinterpret_cast<const float4*>(x)[i];
        const T4 d = reinterpret_cast<const T4*>(delta)[i];
        const float d0 = (float)d[0], d1 = (float)d[1], d2 = (float)d[2], d3 = (float)d[3];
        // a saturated (or non-finite) residual update: the fp16 range was exceeded upstream (the check layernorm_kernel makes)
        if (IsF16<T>::value) sat = !(fmaxf(fmaxf(fabsf(d0), fabsf(d1)), fmaxf(fabsf(d2), fabsf(d3))) < 65504.f);
        v.x += d0; v.y += d1; v.z += d2; v.w += d3;
        if (tokens_out) reinterpret_cast<float4*>(tokens_out)[i] = v;
        if (feat_out && s > 0) reinterpret_cast<float4*>(feat_out)[(f * (S - 1) + s - 1) * d4 + c] = v;
    }
    if (IsF16<T>::value && overflow && __any(sat) && (threadIdx.x & 63) == 0) atomicOr(overflow, 1);
}

// tap of the residual stream after a block (dtk_vit_model.tap_out): out += scale * (x + pending update)
template <typename T>
__global__ __launch_bounds__(256) void tap_kernel(const float* __restrict__ x, const T* __restrict__ delta, float* __restrict__ out,
                                                  float scale, long long total4) {
    typedef typename Vec<T>::t4 T4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    float4 v = reinterpret_cast<const float4*>(x)[i];
    if (delta) {
        const T4 d = reinterpret_cast<const T4*>(delta)[i];
        v.x += (float)d[0]; v.y += (float)d[1]; v.z += (float)d[2]; v.w += (float)d[3];
    }
    float4 o = reinterpret_cast<float4*>(out)[i];
    o.x += scale * v.x; o.y += scale * v.y; o.z += scale * v.z; o.w += scale * v.w;
    reinterpret_cast<float4*>(out)[i] = o;
}

// fp32 tokens [F][S][D] (CLS first) -> token-major features [F][HW][D] (drop CLS)
__global__ __launch_bounds__(256) void drop_cls_kernel(const float* __restrict__ x, float* __restrict__ out, int S, int D,
                                                       long long total4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int d4 = D / 4;
    const long long row = i / d4;  // over F*(S-1)
    const int c = (int)(i - row * d4);
    const long long f = row / (S - 1), s = row - f * (S - 1);
    reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(x)[(f * S + 1 + s) * d4 + c];
}

// frames per pass of the encoder: large batches give the weight-stationary GEMMs long token chunks per workgroup
// (their weights are loaded once per chunk); 30 frames of 67x121 tokens need ~2.6 GB of workspace
// Frames per pass of the encoder.  Round 4: 90 (a whole benchmark video: 7.8 GB of activations, 2.7 % of the 288 GB) instead of
// 30 -- fewer, longer launches: the attention's last partial round of workgroups weighs 0.7 % instead of 2.2 %, the
// weight-stationary GEMMs load their weights a third as often; 301.4 -> 294.8 ms per step (profiles/r04_frame_batch_sweep.txt).
constexpr int VIT_FRAME_BATCH = 90;

// 1-D grid of gemm_tiled_kernel: 8 row blocks (one per XCD) x all column tiles per group
inline unsigned gemm_grid(int N, long long rows) {
    const long long ncol = dtk_cdiv(N, GN), nrow = dtk_cdiv(rows, GM);
    return (unsigned)(dtk_cdiv(nrow, 8) * 8 * ncol);
}

struct VitPlan {
    int S, Sp, FB;
    size_t x, xn, q, k, vt, ao, hid, delta, total;
    bool split;                                    // some block runs on split operands (dtk_vit_layer.qkv_w_lo): the lo planes exist
    size_t xn_lo, q_lo, k_lo, vt_lo, ao_lo, hid_lo;
};

inline bool vit_layer_split(const dtk_vit_layer& L) { return L.qkv_w_lo != nullptr; }

VitPlan vit_plan(const dtk_vit_model* m, int ph, int pw, int frames) {
    VitPlan p;
    p.S = ph * pw + 1;
    p.Sp = (p.S + 127) / 128 * 128;
    const int fb = m->frame_batch > 0 ? m->frame_batch : VIT_FRAME_BATCH;
    p.FB = frames < fb ? frames : fb;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    const size_t rows = (size_t)p.FB * p.S;
    p.x = off; off = al(off + rows * m->D * 4);
    p.xn = off; off = al(off + rows * m->D * 2);
    p.q = off; off = al(off + (size_t)p.FB * m->heads * p.Sp * 64 * 2);
    p.k = off; off = al(off + (size_t)p.FB * m->heads * p.Sp * 64 * 2);
    p.vt = off; off = al(off + (size_t)p.FB * m->heads * p.Sp * 64 * 2);
    p.ao = off; off = al(off + rows * m->D * 2);
    p.hid = off; off = al(off + rows * 4 * m->D * 2);
    p.delta = off; off = al(off + rows * m->D * 2);
    p.split = false;
    for (int l = 0; l < m->depth; ++l) p.split |= vit_layer_split(m->layers[l]);
    p.xn_lo = p.q_lo = p.k_lo = p.vt_lo = p.ao_lo = p.hid_lo = 0;
    if (p.split) {   // the lo planes of a split block's operands (same shapes as the hi planes; q / k / vt contiguous: zeroed together)
        p.xn_lo = off; off = al(off + rows * m->D * 2);
        p.q_lo = off; off = al(off + (size_t)p.FB * m->heads * p.Sp * 64 * 2);
        p.k_lo = off; off = al(off + (size_t)p.FB * m->heads * p.Sp * 64 * 2);
        p.vt_lo = off; off = al(off + (size_t)p.FB * m->heads * p.Sp * 64 * 2);
        p.ao_lo = off; off = al(off + rows * m->D * 2);
        p.hid_lo = off; off = al(off + rows * 4 * m->D * 2);
    }
    p.total = off;
    return p;
}


// attention launch per operand type (the kernel lives in a per-type namespace of vit_attention2.h)
template <typename T> struct Att;
template <> struct Att<_Float16> {
    static int launch(const _Float16* q, const _Float16* k, const _Float16* vt, _Float16* o, int S, int Sp, int heads, int D,
                      int FH, int variant, hipStream_t st) {
        int QB;
        if (variant == 2) {
            const unsigned grid = att2c::attention2_grid(FH, S, 1, &QB);
            DTK_LAUNCH("vit_attention", (att2_f16::attention2_kernel<1>), dim3(grid), dim3(512), 0, st, q, k, vt, o, S, Sp,
                       heads, D, FH, QB);
        } else if (variant >= 5) {   // round-5 experiment (stand-alone stage only): two waves per SIMD; variant = 5 + its ablation bits
            const unsigned grid = att2_f16::attention5_grid(FH, S, &QB);
#define DTK_A5(V, ABLV)                                                                                                       \
    if (variant == V)                                                                                                         \
        DTK_LAUNCH("vit_attention", (att2_f16::attention5_kernel<ABLV>), dim3(grid), dim3(512), 0, st, q, k, vt, o, S, Sp, heads, D, \
                   FH, QB);
            DTK_A5(5, 0) DTK_A5(6, 32) DTK_A5(7, 8 | 64) DTK_A5(8, 8 | 32 | 64) DTK_A5(9, 8 | 128) DTK_A5(10, 8 | 32 | 128)
#undef DTK_A5
        } else if (variant == 4) {   // rounds 4-5: 64 queries per wave (DTK_VIT_ATTENTION_V4 / DTK_OPERAND_ATTENTION_V4)
            const unsigned grid = att2_f16::attention4_grid(FH, S, &QB);
            DTK_LAUNCH("vit_attention", (att2_f16::attention4_kernel<0>), dim3(grid), dim3(256), 0, st, q, k, vt, o, S, Sp, heads,
                       D, FH, QB);
        } else {
            const unsigned grid = att2_f16::attention6_grid(FH, S, &QB);
            DTK_LAUNCH("vit_attention", (att2_f16::attention6_kernel<0>), dim3(grid), dim3(256), 0, st, q, k, vt, o, S, Sp, heads,
                       D, FH, QB);
        }
        return DTK_OK;
    }
};
template <> struct Att<__bf16> {
    static int launch(const __bf16* q, const __bf16* k, const __bf16* vt, __bf16* o, int S, int Sp, int heads, int D, int FH,
                      int variant, hipStream_t st) {
        int QB;
        if (variant == 2) {
            const unsigned grid = att2c::attention2_grid(FH, S, 1, &QB);
            DTK_LAUNCH("vit_attention", (att2_bf16::attention2_kernel<1>), dim3(grid), dim3(512), 0, st, q, k, vt, o, S, Sp,
                       heads, D, FH, QB);
        } else if (variant >= 5) {
            const unsigned grid = att2_bf16::attention5_grid(FH, S, &QB);
            DTK_LAUNCH("vit_attention", (att2_bf16::attention5_kernel<0>), dim3(grid), dim3(512), 0, st, q, k, vt, o, S, Sp, heads,
                       D, FH, QB);
        } else if (variant == 4) {
            const unsigned grid = att2_bf16::attention4_grid(FH, S, &QB);
            DTK_LAUNCH("vit_attention", (att2_bf16::attention4_kernel<0>), dim3(grid), dim3(256), 0, st, q, k, vt, o, S, Sp,
                       heads, D, FH, QB);
        } else {
            const unsigned grid = att2_bf16::attention6_grid(FH, S, &QB);
            DTK_LAUNCH("vit_attention", (att2_bf16::attention6_kernel<0>), dim3(grid), dim3(256), 0, st, q, k, vt, o, S, Sp,
                       heads, D, FH, QB);
        }
        return DTK_OK;
    }
};

// |x| >= 65504 or non-finite anywhere in a 16-bit tensor -> *flag |= bit (DTK_VIT_CHECK_RANGE: every intermediate tensor)
template <typename T>
__global__ __launch_bounds__(256) void range_scan_kernel(const T* __restrict__ p, long long n8, int* __restrict__ flag, int bit) {
    typedef typename Vec<T>::t8 T8;
    bool bad = false;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const T8 v = reinterpret_cast<const T8*>(p)[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) bad |= !(fabsf((float)v[e]) < 65504.f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, bit);
}

template <typename T>
int vit_run(const dtk_vit_model* m, const float* frames, int nframes, int video_h, int video_w, float* tokens_out,
            float* feat_out, float* qkv_out, unsigned char* ws, const VitPlan& p, int ph, int pw, hipStream_t st) {
    const int HW = ph * pw, D = m->D;
    float* x = reinterpret_cast<float*>(ws + p.x);
    T* xn = reinterpret_cast<T*>(ws + p.xn);
    T* q = reinterpret_cast<T*>(ws + p.q);
    T* k = reinterpret_cast<T*>(ws + p.k);
    T* vt = reinterpret_cast<T*>(ws + p.vt);
    T* ao = reinterpret_cast<T*>(ws + p.ao);
    T* hid = reinterpret_cast<T*>(ws + p.hid);
    T* delta = reinterpret_cast<T*>(ws + p.delta);
    T* xn_lo = reinterpret_cast<T*>(ws + p.xn_lo);
    T* q_lo = reinterpret_cast<T*>(ws + p.q_lo);
    T* k_lo = reinterpret_cast<T*>(ws + p.k_lo);
    T* vt_lo = reinterpret_cast<T*>(ws + p.vt_lo);
    T* ao_lo = reinterpret_cast<T*>(ws + p.ao_lo);
    T* hid_lo = reinterpret_cast<T*>(ws + p.hid_lo);
    int* ovf = m->overflow;
    // Saturation of Q / K / V^T and of the MLP hidden (overflow bits 2 / 4): tracked for EVERY value of EVERY frame inside the
    // epilogues of the QKV and fc1 GEMMs (GemmEpi::ovf, round 5: four v_max3 per eight values, no extra pass; rounds 3-4 scanned
    // the first frame of a call only, so a later frame could saturate silently).  DTK_VIT_CHECK_RANGE additionally scans the
    // stored tensors of every block (a pass over 1.3 GB per block and 30 frames): the cross-check of the tests.
    const bool scan_all = ovf && (m->flags & DTK_VIT_CHECK_RANGE);
    int* const epi_ovf = IsF16<T>::value ? ovf : nullptr;
    const int S = p.S, Sp = p.Sp;
    // Q/K/V^T padding rows (s >= S) must be finite zeros: they are read by the last KV tile
    {
        const long long n16 = (long long)((p.ao - p.q) / 16);
        DTK_LAUNCH("vit_zero", zero_kernel, dim3(dtk_cdiv(n16, 256)), dim3(256), 0, st, reinterpret_cast<uint4*>(q), n16);
        if (p.split) {
            const long long m16 = (long long)((p.ao_lo - p.q_lo) / 16);
            DTK_LAUNCH("vit_zero", zero_kernel, dim3(dtk_cdiv(m16, 256)), dim3(256), 0, st, reinterpret_cast<uint4*>(q_lo), m16);
        }
    }
    for (int f0 = 0; f0 < nframes; f0 += p.FB) {
        const int nf = (nframes - f0) < p.FB ? (nframes - f0) : p.FB;
        const long long rows = (long long)nf * S;
        auto scan_range = [&](const T* t, long long n, int bit) -> int {
            const int blocks = (int)(dtk_cdiv(n / 8, 256) < 1024 ? dtk_cdiv(n / 8, 256) : 1024);
            DTK_LAUNCH("vit_range_scan", range_scan_kernel<T>, dim3(blocks), dim3(256), 0, st, t, n / 8, ovf, bit);
            return DTK_OK;
        };
        const bool pe_fits = (size_t)nf * video_h * video_w * 16 <= p.delta - p.hid &&
                             (size_t)D * PE_K * 4 <= p.total - p.delta;
        if (m->patch == PE_P && m->stride == PE_S && pe_fits) {
            // split planes of the normalised frames live in `hid`, the split weights in `delta` (both unused until the
            // first block)
            h4v* fh = reinterpret_cast<h4v*>(hid);
            h4v* fl = fh + (size_t)nf * video_h * video_w;
            half_t* wh = reinterpret_cast<half_t*>(delta);
            half_t* wl = wh + (size_t)D * PE_K;
            const long long npix = (long long)nf * video_h * video_w;
            DTK_LAUNCH("vit_frame_split", frame_split4_kernel, dim3(dtk_cdiv(npix, 256)), dim3(256), 0, st,
                       frames + (size_t)f0 * 3 * video_h * video_w, m->mean_std, fh, fl, video_h, video_w, npix);
            DTK_LAUNCH("vit_frame_split", pack_patch_split_kernel, dim3(dtk_cdiv((long long)D * PE_K, 256)), dim3(256), 0, st,
                       m->patch_w, D, wh, wl);
            const int groups_x = dtk_cdiv(pw, PE_TOK);
            const int units = groups_x * ph * nf, slabs = dtk_cdiv(D, PE_NF);
            DTK_LAUNCH("vit_patch_embed", patch_embed_split_kernel, dim3(8 * dtk_cdiv(units, 8) * slabs), dim3(256), 0, st, fh,
                       fl, wh, wl, m->patch_b, m->pos, m->cls_pos, x, video_h, video_w, ph, pw, D, S, groups_x, units, slabs);
        } else {
            DTK_LAUNCH("vit_patch_embed", patch_embed_kernel, dim3(dtk_cdiv(HW, 64), dtk_cdiv(D, 64), nf), dim3(256), 0, st,
                       frames + (size_t)f0 * 3 * video_h * video_w, m->patch_w, m->patch_b, m->pos, m->cls_pos, m->mean_std,
                       x, video_h, video_w, ph, pw, D, m->patch, m->stride, S);
        }
        // K = 384 GEMMs run weight-stationary; everything else (fc2, wider models) on the tiled kernel.  The residual
        // updates (LayerScale'd projection / MLP outputs) are written as `delta` and added by the next LayerNorm.
        // DTK_VIT_TILED_GEMMS: every GEMM on the tiled kernel (tests cross-check the weight-stationary one with it)
        const bool ws_ok = D == WS_K && !(m->flags & DTK_VIT_TILED_GEMMS);
        const bool ws_v1 = (m->flags & DTK_VIT_GEMM_WS_V1) != 0;
        const bool wide_v1 = (m->flags & DTK_VIT_GEMM_WIDE_V1) != 0;   // the 256 x 256 GEMMs without the half-step fragment prefetch (A / B)
#define DTK_SPLIT_DMA(NAME, EPI_, N_, ...)                                                                                            \
    do {                                                                                                                              \
        if (wide_v1) {                                                                                                                \
            DTK_LAUNCH(NAME, (gemm_split_dma_kernel<T, EPI_, false>), dim3(gemm_split_dma_grid(N_, rows)), dim3(512), 0, st, __VA_ARGS__); \
        } else {                                                                                                                      \
            DTK_LAUNCH(NAME, (gemm_split_dma_kernel<T, EPI_, true>), dim3(gemm_split_dma_grid(N_, rows)), dim3(512), 0, st, __VA_ARGS__);  \
        }                                                                                                                             \
    } while (0)
#define DTK_WIDE(NAME, EPI_, N_, ...)                                                                                          \
    do {                                                                                                                       \
        if (wide_v1) {                                                                                                         \
            DTK_LAUNCH(NAME, (gemm_wide_kernel<T, EPI_, false>), dim3(gemm_wide_grid(N_, rows)), dim3(512), 0, st, __VA_ARGS__); \
        } else {                                                                                                               \
            DTK_LAUNCH(NAME, (gemm_wide_kernel<T, EPI_, true>), dim3(gemm_wide_grid(N_, rows)), dim3(512), 0, st, __VA_ARGS__);  \
        }                                                                                                                      \
    } while (0)
        // DTK_DEV: bits 0-1 skip the weight-stationary kernel's stores; gemm_wide_kernel: 4 no stores, 8 no main loop (pipeline fill +
        // epilogue only), 16 every stage from k = 0 (cache-hot operands)
        const int dbg_ns = DTK_DBG(dtk_dev_flags() >> 16, 0x7f);
        auto ws_grid = [&](int N) {  // one resident round: one workgroup per CU
            const int colwg = dtk_cdiv(N, WS_COLS);
            const int chunks = 256 / colwg > 0 ? 256 / colwg : 1;
            const long long tiles = dtk_cdiv(rows, WS_ROWS);
            const int tpc = (int)dtk_cdiv(tiles, chunks);
            return std::make_pair(dim3(colwg, (unsigned)dtk_cdiv(tiles, tpc)), tpc);
        };
        // GEMMs of widths without a weight-stationary form: the 256 x 256 DMA kernel when the shape allows, else 128 x 128
        const bool wide_ok = !(m->flags & DTK_VIT_TILED_GEMMS) && D % W2_N == 0;
        bool pending = false;   // `delta` holds a residual update that the next LayerNorm (or the final update) has to apply
        bool ln1_done = false;  // the previous block's fc2 has already written this block's LayerNorm-1 rows to xn (round 6)
        const bool no_ln_fusion = (m->flags & DTK_VIT_NO_LN_FUSION) != 0;
        auto tap = [&](int l, bool with_delta) -> int {   // dtk_vit_model.tap_out: the block output of layer l joins the mean
            if (!m->tap_out || !((m->tap_mask >> l) & 1)) return DTK_OK;
            const long long t4 = rows * (D / 4);
            DTK_LAUNCH("vit_tap", tap_kernel<T>, dim3(dtk_cdiv(t4, 256)), dim3(256), 0, st, x, with_delta ? (const T*)delta : (const T*)nullptr,
                       m->tap_out + (size_t)f0 * S * D, m->tap_scale, t4);
            return DTK_OK;
        };
        for (int l = 0; l < m->depth; ++l) {
            const dtk_vit_layer& L = m->layers[l];
            const T* qkv_w = reinterpret_cast<const T*>(L.qkv_w);
            const T* proj_w = reinterpret_cast<const T*>(L.proj_w);
            const T* fc1_w = reinterpret_cast<const T*>(L.fc1_w);
            const T* fc2_w = reinterpret_cast<const T*>(L.fc2_w);
            if (vit_layer_split(L)) {
                // ---- the escalated precision (vit_split.h): every product on hi + lo operands, updates straight into x ----
                const T* qkv_wl = reinterpret_cast<const T*>(L.qkv_w_lo);
                const T* proj_wl = reinterpret_cast<const T*>(L.proj_w_lo);
                const T* fc1_wl = reinterpret_cast<const T*>(L.fc1_w_lo);
                const T* fc2_wl = reinterpret_cast<const T*>(L.fc2_w_lo);
                const float inv_ws = 1.f / (L.w_scale > 0.f ? L.w_scale : 1.f);
                // the LDS-DMA form of the split GEMM (256 x 128 tiles; round 6: fc2 38.3 -> 34.5, qkv 41.4 -> 39.3 ms per step, ViT-L
                // 1613 -> 1461 ms) unless DTK_VIT_TILED_GEMMS asks for the register-staged 128 x 128 kernel (the tests' cross-check)
                const bool split_dma = !(m->flags & DTK_VIT_TILED_GEMMS);
                DTK_LAUNCH("vit_layernorm_split", layernorm_split_kernel<T>, dim3(dtk_cdiv(rows, 4)), dim3(256), 0, st, x,
                           pending ? (const T*)delta : (const T*)nullptr, L.ln1_w, L.ln1_b, xn, xn_lo, rows, D, m->ln_eps, ovf);
                pending = false;
                SplitEpi<T> se{};
                if (qkv_out && l == m->depth - 1) {
                    se.bias = L.qkv_b; se.inv_wscale = inv_ws; se.out_f32 = qkv_out + (size_t)f0 * S * 3 * D;
                    if (split_dma && (3 * D) % SD_N == 0) {
                    DTK_SPLIT_DMA("vit_gemm_qkv_facet", SEPI_F32, 3 * D, xn, xn_lo, qkv_w, qkv_wl, rows, 3 * D, D, se);
                } else {
                    DTK_LAUNCH("vit_gemm_qkv_facet", (gemm_split_kernel<T, SEPI_F32>), dim3(gemm_split_grid(3 * D, rows)), dim3(256), 0, st, xn, xn_lo, qkv_w, qkv_wl, rows, 3 * D, D, se);
                }
                    se = SplitEpi<T>{};
                }
                se.bias = L.qkv_b; se.inv_wscale = inv_ws; se.q_hi = q; se.q_lo = q_lo; se.k_hi = k; se.k_lo = k_lo; se.vt_hi = vt;
                se.vt_lo = vt_lo; se.S = S; se.Sp = Sp; se.heads = m->heads; se.D = D; se.qscale = 0.125f * 1.4426950408889634f;
                se.ovf = epi_ovf;
                if (split_dma && (3 * D) % SD_N == 0) {
                    DTK_SPLIT_DMA("vit_gemm_qkv_split", SEPI_QKV, 3 * D, xn, xn_lo, qkv_w, qkv_wl, rows, 3 * D, D, se);
                } else {
                    DTK_LAUNCH("vit_gemm_qkv_split", (gemm_split_kernel<T, SEPI_QKV>), dim3(gemm_split_grid(3 * D, rows)), dim3(256), 0, st, xn, xn_lo, qkv_w, qkv_wl, rows, 3 * D, D, se);
                }
                {
                    const int nqb = dtk_cdiv(Sp, 128);
                    DTK_LAUNCH("vit_attention_split", attention_split_kernel<T>, dim3((unsigned)(nf * m->heads * nqb)), dim3(256), 0, st,
                               (const T*)q, (const T*)q_lo, (const T*)k, (const T*)k_lo, (const T*)vt, (const T*)vt_lo, ao, ao_lo, S, Sp,
                               m->heads, nqb);
                }
                se = SplitEpi<T>{};
                se.bias = L.proj_b; se.inv_wscale = inv_ws; se.x = x; se.gamma = L.ls1;
                if (split_dma && (D) % SD_N == 0) {
                    DTK_SPLIT_DMA("vit_gemm_proj_split", SEPI_RESID, D, (const T*)ao, (const T*)ao_lo, proj_w, proj_wl, rows, D, D, se);
                } else {
                    DTK_LAUNCH("vit_gemm_proj_split", (gemm_split_kernel<T, SEPI_RESID>), dim3(gemm_split_grid(D, rows)), dim3(256), 0, st, (const T*)ao, (const T*)ao_lo, proj_w, proj_wl, rows, D, D, se);
                }
                DTK_LAUNCH("vit_layernorm_split", layernorm_split_kernel<T>, dim3(dtk_cdiv(rows, 4)), dim3(256), 0, st, x,
                           (const T*)nullptr, L.ln2_w, L.ln2_b, xn, xn_lo, rows, D, m->ln_eps, ovf);
                se = SplitEpi<T>{};
                se.bias = L.fc1_b; se.inv_wscale = inv_ws; se.out_hi = hid; se.out_lo = hid_lo; se.ovf = epi_ovf;
                if (split_dma && (4 * D) % SD_N == 0) {
                    DTK_SPLIT_DMA("vit_gemm_fc1_split", SEPI_GELU, 4 * D, (const T*)xn, (const T*)xn_lo, fc1_w, fc1_wl, rows, 4 * D, D, se);
                } else {
                    DTK_LAUNCH("vit_gemm_fc1_split", (gemm_split_kernel<T, SEPI_GELU>), dim3(gemm_split_grid(4 * D, rows)), dim3(256), 0, st, (const T*)xn, (const T*)xn_lo, fc1_w, fc1_wl, rows, 4 * D, D, se);
                }
                se = SplitEpi<T>{};
                se.bias = L.fc2_b; se.inv_wscale = inv_ws; se.x = x; se.gamma = L.ls2;
                if (split_dma && (D) % SD_N == 0) {
                    DTK_SPLIT_DMA("vit_gemm_fc2_split", SEPI_RESID, D, (const T*)hid, (const T*)hid_lo, fc2_w, fc2_wl, rows, D, 4 * D, se);
                } else {
                    DTK_LAUNCH("vit_gemm_fc2_split", (gemm_split_kernel<T, SEPI_RESID>), dim3(gemm_split_grid(D, rows)), dim3(256), 0, st, (const T*)hid, (const T*)hid_lo, fc2_w, fc2_wl, rows, D, 4 * D, se);
                }
                if (tap(l, false)) return DTK_E_HIP;
                continue;
            }
            GemmEpi<T> e{};
            if (!ln1_done)
                DTK_LAUNCH("vit_layernorm", layernorm_kernel<T>, dim3(dtk_cdiv(rows, 4)), dim3(256), 0, st, x,
                           pending ? (const T*)delta : (const T*)nullptr, L.ln1_w, L.ln1_b, xn, rows, D, m->ln_eps, ovf);
            ln1_done = false;
            pending = true;
            if (qkv_out && l == m->depth - 1) {  // the qkv hook of the reference (models/extractor.py:107-118), fp32 out
                e.bias = L.qkv_b; e.out_f32 = qkv_out + (size_t)f0 * S * 3 * D;
                DTK_LAUNCH("vit_gemm_qkv_facet", (gemm_tiled_kernel<T, EPI_F32>), dim3(gemm_grid(3 * D, rows)), dim3(256), 0, st,
                           xn, qkv_w, rows, 3 * D, D, e);
                e = GemmEpi<T>{};
            }
            e.bias = L.qkv_b; e.q = q; e.k = k; e.vt = vt; e.S = S; e.Sp = Sp; e.heads = m->heads; e.D = D;
            e.qscale = 0.125f * 1.4426950408889634f;
            e.no_store = dbg_ns;
            e.ovf = epi_ovf;
            if (ws_ok) {
                const auto gr = ws_grid(3 * D);
                if (ws_v1) {
                    DTK_LAUNCH("vit_gemm_qkv", (gemm_ws_kernel<T, EPI_QKV, 0>), gr.first, dim3(256), 0, st, xn, qkv_w, rows, 3 * D, e,
                               gr.second);
                } else {
                    DTK_LAUNCH("vit_gemm_qkv", (gemm_ws_kernel<T, EPI_QKV>), gr.first, dim3(256), 0, st, xn, qkv_w, rows, 3 * D, e,
                               gr.second);
                }
            } else if (wide_ok) {
                DTK_WIDE("vit_gemm_qkv", EPI_QKV, 3 * D, xn, qkv_w, rows, 3 * D, D, e);
            } else {
                DTK_LAUNCH("vit_gemm_qkv", (gemm_tiled_kernel<T, EPI_QKV>), dim3(gemm_grid(3 * D, rows)), dim3(256), 0, st, xn,
                           qkv_w, rows, 3 * D, D, e);
            }
            if (scan_all && scan_range(q, (long long)(p.ao - p.q) / 2, 2)) return DTK_E_HIP;
            if (Att<T>::launch(q, k, vt, ao, S, Sp, m->heads, D, nf * m->heads, (m->flags & DTK_VIT_ATTENTION_V2) ? 2 : ((m->flags & DTK_VIT_ATTENTION_V4) ? 4 : 0), st)) return DTK_E_HIP;
            e = GemmEpi<T>{};
            e.bias = L.proj_b; e.delta = delta; e.gamma = L.ls1; e.no_store = dbg_ns;
            if (ws_ok) {
                const auto gr = ws_grid(D);
                if (ws_v1) {
                    DTK_LAUNCH("vit_gemm_proj", (gemm_ws_kernel<T, EPI_DELTA, 0>), gr.first, dim3(256), 0, st, ao, proj_w, rows, D, e,
                               gr.second);
                } else {
                    DTK_LAUNCH("vit_gemm_proj", (gemm_ws_kernel<T, EPI_DELTA>), gr.first, dim3(256), 0, st, ao, proj_w, rows, D, e,
                               gr.second);
                }
            } else if (wide_ok) {
                DTK_WIDE("vit_gemm_proj", EPI_DELTA, D, ao, proj_w, rows, D, D, e);
            } else {
                DTK_LAUNCH("vit_gemm_proj", (gemm_tiled_kernel<T, EPI_DELTA>), dim3(gemm_grid(D, rows)), dim3(256), 0, st, ao,
                           proj_w, rows, D, D, e);
            }
            DTK_LAUNCH("vit_layernorm", layernorm_kernel<T>, dim3(dtk_cdiv(rows, 4)), dim3(256), 0, st, x, (const T*)delta,
                       L.ln2_w, L.ln2_b, xn, rows, D, m->ln_eps, ovf);
            e = GemmEpi<T>{};
            e.bias = L.fc1_b; e.out = hid; e.no_store = dbg_ns; e.ovf = epi_ovf;
            if (ws_ok) {
                const auto gr = ws_grid(4 * D);
                if (ws_v1) {
                    DTK_LAUNCH("vit_gemm_fc1", (gemm_ws_kernel<T, EPI_GELU, 0>), gr.first, dim3(256), 0, st, xn, fc1_w, rows, 4 * D, e,
                               gr.second);
                } else {
                    DTK_LAUNCH("vit_gemm_fc1", (gemm_ws_kernel<T, EPI_GELU>), gr.first, dim3(256), 0, st, xn, fc1_w, rows, 4 * D, e,
                               gr.second);
                }
            } else if (wide_ok) {
                DTK_WIDE("vit_gemm_fc1", EPI_GELU, 4 * D, xn, fc1_w, rows, 4 * D, D, e);
            } else {
                DTK_LAUNCH("vit_gemm_fc1", (gemm_tiled_kernel<T, EPI_GELU>), dim3(gemm_grid(4 * D, rows)), dim3(256), 0, st, xn,
                           fc1_w, rows, 4 * D, D, e);
            }
            if (scan_all && scan_range(hid, rows * 4 * D, 4)) return DTK_E_HIP;
            e = GemmEpi<T>{};
            e.bias = L.fc2_b; e.delta = delta; e.gamma = L.ls2; e.no_store = dbg_ns;
            if (ws_ok && D == WD_N) {  // (ws_ok: the fast-path GEMMs are on, dtk_vit_model.flags)
                // round 6: the LayerNorm of the next block runs inside this epilogue when that block is a fast one (the split blocks
                // have their own LayerNorm: hi / lo planes); the last block's update goes to the outputs (below)
                const bool fuse_ln = !wide_v1 && !no_ln_fusion && l + 1 < m->depth && !vit_layer_split(m->layers[l + 1]);
                if (wide_v1) {
                    DTK_LAUNCH("vit_gemm_fc2", (gemm_wide_delta_kernel<T, false>), dim3(dtk_cdiv(rows, WD_M)), dim3(512), 0, st, hid, fc2_w,
                               rows, 4 * D, e);
                } else if (fuse_ln) {
                    const dtk_vit_layer& Ln = m->layers[l + 1];
                    e.ln_x = x; e.ln_w = Ln.ln1_w; e.ln_b = Ln.ln1_b; e.ln_out = xn; e.ln_eps = m->ln_eps; e.ln_ovf = ovf;
                    DTK_LAUNCH("vit_gemm_fc2", (gemm_wide_delta_kernel<T, true, true>), dim3(dtk_cdiv(rows, WD_M)), dim3(512), 0, st, hid, fc2_w,
                               rows, 4 * D, e);
                    pending = false;      // x is up to date ...
                    ln1_done = true;      // ... and xn holds the next block's normalised rows
                } else {
                    DTK_LAUNCH("vit_gemm_fc2", (gemm_wide_delta_kernel<T, true>), dim3(dtk_cdiv(rows, WD_M)), dim3(512), 0, st, hid, fc2_w,
                               rows, 4 * D, e);
                }
            } else if (wide_ok) {
                DTK_WIDE("vit_gemm_fc2", EPI_DELTA, D, hid, fc2_w, rows, D, 4 * D, e);
            } else {
                DTK_LAUNCH("vit_gemm_fc2", (gemm_tiled_kernel<T, EPI_DELTA>), dim3(gemm_grid(D, rows)), dim3(256), 0, st, hid,
                           fc2_w, rows, D, 4 * D, e);
            }
            if (tap(l, pending)) return DTK_E_HIP;
        }
        if (pending && (tokens_out || feat_out)) {
            // the last MLP's residual update, written straight to the outputs (x itself is dead after the last block)
            const long long t4 = rows * (D / 4);
            DTK_LAUNCH("vit_final_update", final_update_kernel<T>, dim3(dtk_cdiv(t4, 256)), dim3(256), 0, st, x, (const T*)delta,
                       tokens_out ? tokens_out + (size_t)f0 * S * D : (float*)nullptr,
                       feat_out ? feat_out + (size_t)f0 * HW * D : (float*)nullptr, S, D, t4, ovf);
        } else {   // depth 0 (embedding + position encoding only) or a split last block: no pending update
            if (pending)   // (qkv_out only: keep the residual-update check of the last block)
                DTK_LAUNCH("vit_layernorm", layernorm_kernel<T>, dim3(dtk_cdiv(rows, 4)), dim3(256), 0, st, x, (const T*)delta,
                           (const float*)nullptr, (const float*)nullptr, (T*)nullptr, rows, D, m->ln_eps, ovf);
            if (tokens_out)
                DTK_HIP(hipMemcpyAsync(tokens_out + (size_t)f0 * S * D, x, (size_t)rows * D * sizeof(float),
                                       hipMemcpyDeviceToDevice, st));
            if (feat_out) {
                const long long total4 = (long long)nf * HW * (D / 4);
                DTK_LAUNCH("vit_drop_cls", drop_cls_kernel, dim3(dtk_cdiv(total4, 256)), dim3(256), 0, st, x,
                           feat_out + (size_t)f0 * HW * D, S, D, total4);
            }
        }
    }
    return DTK_OK;
}

}  // namespace

extern "C" size_t dtk_vit_workspace_bytes(const dtk_vit_model* m, int video_h, int video_w, int frames) {
    if (!m || frames <= 0) return 0;
    const int ph = 1 + (video_h - m->patch) / m->stride, pw = 1 + (video_w - m->patch) / m->stride;
    return vit_plan(m, ph, pw, frames).total;
}

extern "C" int dtk_vit_forward(const dtk_vit_model* m, const float* frames, int nframes, int video_h, int video_w,
                               float* tokens_out, float* feat_out, float* qkv_out, void* workspace,
                               size_t workspace_bytes, void* stream) {
    DTK_REQUIRE(m && frames && workspace && (tokens_out || feat_out || qkv_out || m->tap_out), "dtk_vit_forward: null pointer");
    DTK_REQUIRE(!qkv_out || m->depth > 0, "dtk_vit_forward: qkv_out needs at least one block");
    DTK_REQUIRE(m->D > 0 && m->heads > 0 && m->D == m->heads * 64, "dtk_vit_forward: d_head must be 64 (D=%d heads=%d)",
                m->D, m->heads);
    DTK_REQUIRE(m->D % 32 == 0 && m->depth >= 0 && m->layers, "dtk_vit_forward: bad model");
    DTK_REQUIRE(video_h >= m->patch && video_w >= m->patch, "dtk_vit_forward: frame smaller than a patch");
    const int ph = 1 + (video_h - m->patch) / m->stride, pw = 1 + (video_w - m->patch) / m->stride;
    const VitPlan p = vit_plan(m, ph, pw, nframes);
    if (workspace_bytes < p.total) {
        dtk_set_error("dtk_vit_forward: workspace %zu B < required %zu B", workspace_bytes, p.total);
        return DTK_E_WORKSPACE;
    }
    hipStream_t st = dtk_stream(stream);
    unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
    if (m->flags & DTK_VIT_BF16)
        return vit_run<__bf16>(m, frames, nframes, video_h, video_w, tokens_out, feat_out, qkv_out, ws, p, ph, pw, st);
    return vit_run<_Float16>(m, frames, nframes, video_h, video_w, tokens_out, feat_out, qkv_out, ws, p, ph, pw, st);
}

// The attention stage on its own (tests drive its guard / safe-pass logic with crafted operands; layouts as inside
// dtk_vit_forward).
extern "C" int dtk_vit_attention(const void* q, const void* k, const void* vt, void* out, int frames, int heads, int S,
                                 int Sp, int operand_type, void* stream) {
    DTK_REQUIRE(q && k && vt && out, "dtk_vit_attention: null pointer");
    DTK_REQUIRE(frames > 0 && heads > 0 && S > 0 && Sp >= S && Sp % 64 == 0, "dtk_vit_attention: bad sizes (Sp %% 64 == 0, Sp >= S)");
    int variant = (operand_type & DTK_OPERAND_ATTENTION_V2) ? 2 : ((operand_type & DTK_OPERAND_ATTENTION_V4) ? 4 : 0);
    if (operand_type & DTK_OPERAND_ATTENTION_V5) {   // 5 full, 6 in phase; micro-benchmark ablations (results are garbage): 7 / 8 no vector work, 9 / 10 no matrix work
        const int inph = (operand_type & DTK_OPERAND_ATTENTION_V5_INPHASE) ? 1 : 0;
        variant = 5 + inph + ((operand_type & 0x800) ? 2 : 0) + ((operand_type & 0x1000) ? 4 : 0);
        DTK_REQUIRE(variant <= 10, "dtk_vit_attention: ablation bits 0x800 and 0x1000 exclude each other");
    }
    operand_type &= ~(DTK_OPERAND_ATTENTION_V2 | DTK_OPERAND_ATTENTION_V4 | DTK_OPERAND_ATTENTION_V5 | DTK_OPERAND_ATTENTION_V5_INPHASE | 0x800 | 0x1000);
    DTK_REQUIRE(operand_type == DTK_OPERAND_F16 || operand_type == DTK_OPERAND_BF16, "dtk_vit_attention: operand_type");
    if (operand_type == DTK_OPERAND_BF16)
        return Att<__bf16>::launch(reinterpret_cast<const __bf16*>(q), reinterpret_cast<const __bf16*>(k),
                                   reinterpret_cast<const __bf16*>(vt), reinterpret_cast<__bf16*>(out), S, Sp, heads,
                                   heads * 64, frames * heads, variant, dtk_stream(stream));
    return Att<_Float16>::launch(reinterpret_cast<const _Float16*>(q), reinterpret_cast<const _Float16*>(k),
                                 reinterpret_cast<const _Float16*>(vt), reinterpret_cast<_Float16*>(out), S, Sp, heads,
                                 heads * 64, frames * heads, variant, dtk_stream(stream));
}

// The same stage on split operands (vit_split.h: the escalated precision): hi / lo planes of every operand and of the output.
extern "C" int dtk_vit_attention_split(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* vt_hi,
                                       const void* vt_lo, void* out_hi, void* out_lo, int frames, int heads, int S, int Sp,
                                       int operand_type, void* stream) {
    DTK_REQUIRE(q_hi && q_lo && k_hi && k_lo && vt_hi && vt_lo && out_hi && out_lo, "dtk_vit_attention_split: null pointer");
    DTK_REQUIRE(frames > 0 && heads > 0 && S > 0 && Sp >= S && Sp % 64 == 0, "dtk_vit_attention_split: bad sizes (Sp %% 64 == 0, Sp >= S)");
    DTK_REQUIRE(operand_type == DTK_OPERAND_F16 || operand_type == DTK_OPERAND_BF16, "dtk_vit_attention_split: operand_type");
    hipStream_t st = dtk_stream(stream);
    const int nqb = dtk_cdiv(Sp, 128);
    const unsigned grid = (unsigned)(frames * heads * nqb);
    if (operand_type == DTK_OPERAND_BF16) {
        typedef __bf16 T;
        DTK_LAUNCH("vit_attention_split", attention_split_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)q_hi, (const T*)q_lo,
                   (const T*)k_hi, (const T*)k_lo, (const T*)vt_hi, (const T*)vt_lo, (T*)out_hi, (T*)out_lo, S, Sp, heads, nqb);
    } else {
        typedef _Float16 T;
        DTK_LAUNCH("vit_attention_split", attention_split_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)q_hi, (const T*)q_lo,
                   (const T*)k_hi, (const T*)k_lo, (const T*)vt_hi, (const T*)vt_lo, (T*)out_hi, (T*)out_lo, S, Sp, heads, nqb);
    }
    return DTK_OK;
}
