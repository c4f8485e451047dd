// delta_dino.hip -- P2: refined = dino + align(DeltaDINO(frame))   (models/tracker.py:113-135,
// models/networks/delta_dino.py:53-61, models/utils.py:7-45), fp32 end to end.
//
//   conv5x5_kernel   implicit-GEMM 5x5 convolution (reflect padding, dilation 1 or 2) on the f32-input MFMA
//                    (32x32x2: exact fmaf chains, 157 TFLOP/s peak), eval-mode BatchNorm folded into a per-channel
//                    scale/shift, optional ReLU.  Activations are NHWC fp32; one workgroup = 8x8 output pixels x 64
//                    output channels, K swept in chunks of 8 input channels x 25 taps staged in LDS.
//   blurpool_kernel  antialiased_cnns.BlurPool(stride 2, filt 4, reflect pad (1,2,1,2)), NHWC.
//   align_add_kernel bilinear (border, align_corners) resample of the stride-8 CNN map at the ViT token centres,
//                    added to the token-major DINO features; also emits the per-cell L2 norms.
#include "common.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int CK = 8;        // input channels per LDS chunk
constexpr int TP = 8;        // output tile is TP x TP pixels
constexpr int TNC = 64;      // output channels per workgroup
constexpr int PITCH = 40;    // LDS row pitch of the input patch: 4 tile rows land on 32 distinct banks

__device__ __forceinline__ int reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return min(max(i, 0), n - 1);
}

// weights: Wk[25][CinP][CoutP] (zero padded), scale/shift[CoutP]
template <int DIL, bool NCHW_IN>
__global__ __launch_bounds__(256) void conv5x5_kernel(const float* __restrict__ in, const float* __restrict__ Wk,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      float* __restrict__ out, int H, int W, int Cin, int CinP, int Cout,
                                                      int CoutP, int relu, int tiles_x) {
    constexpr int PH = TP + 4 * DIL;  // patch side
    __shared__ float Xs[CK][PH][PITCH];
    __shared__ float Ws[25][CK][TNC];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
    const int y0 = ty * TP, x0 = tx * TP;
    const int n0 = blockIdx.y * TNC;
    const size_t frame = blockIdx.z;
    const float* fin = in + frame * (size_t)H * W * Cin;
    const int msub = w >> 1, nsub = w & 1;
    const int li = lane & 31, lk = lane >> 5;
    const int ly = msub * 4 + (li >> 3), lx = li & 7;
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int c0 = 0; c0 < CinP; c0 += CK) {
        __syncthreads();
        // ---- stage the input patch (reflect padding resolved here) ----
        if (NCHW_IN) {
            for (int idx = tid; idx < CK * PH * PH; idx += 256) {
                const int c = idx / (PH * PH), p = idx - c * (PH * PH);
                const int py = p / PH, px = p - py * PH;
                const int gy = reflect(y0 - 2 * DIL + py, H), gx = reflect(x0 - 2 * DIL + px, W);
                Xs[c][py][px] = (c0 + c < Cin) ? fin[((size_t)(c0 + c) * H + gy) * W + gx] : 0.f;
            }
        } else {
            for (int idx = tid; idx < PH * PH * 2; idx += 256) {
                const int p = idx >> 1, hq = idx & 1;
                const int py = p / PH, px = p - py * PH;
                const int gy = reflect(y0 - 2 * DIL + py, H), gx = reflect(x0 - 2 * DIL + px, W);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c0 + hq * 4 < Cin) v = *reinterpret_cast<const float4*>(fin + ((size_t)gy * W + gx) * Cin + c0 + hq * 4);
                Xs[hq * 4 + 0][py][px] = v.x;
                Xs[hq * 4 + 1][py][px] = v.y;
                Xs[hq * 4 + 2][py][px] = v.z;
                Xs[hq * 4 + 3][py][px] = v.w;
            }
        }
        // ---- stage the weights of this channel chunk: 25 taps x 8 cin x 64 cout ----
        for (int idx = tid; idx < 25 * CK * (TNC / 4); idx += 256) {
            const int row = idx / (TNC / 4), q = idx - row * (TNC / 4);  // row = tap*CK + c
            const int tap = row / CK, c = row - tap * CK;
            const float4 v = *reinterpret_cast<const float4*>(Wk + ((size_t)tap * CinP + c0 + c) * CoutP + n0 + q * 4);
            *reinterpret_cast<float4*>(&Ws[tap][c][q * 4]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int ky = 0; ky < 5; ++ky)
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                const int tap = ky * 5 + kx;
#pragma unroll
                for (int s = 0; s < CK / 2; ++s) {
                    const float a = Xs[2 * s + lk][ly + ky * DIL][lx + kx * DIL];
                    const float b = Ws[tap][2 * s + lk][nsub * 32 + li];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                }
            }
    }
    // ---- epilogue: D[i][j], j = lane&31 (cout), i = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel of the 32-px sub-tile) ----
    const int co = n0 + nsub * 32 + li;
    if (co < Cout) {
        const float sc = scale[co], sh = shift[co];
        float* fout = out + frame * (size_t)H * W * Cout;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * lk;
            const int oy = y0 + msub * 4 + (i >> 3), ox = x0 + (i & 7);
            if (oy < H && ox < W) {
                float v = acc[r] * sc + sh;
                if (relu) v = fmaxf(v, 0.f);
                fout[((size_t)oy * W + ox) * Cout + co] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// split-fp16 convolution (layers 2-4): every fp32 operand is carried as hi + lo fp16 halves (x = xh + xl to 2^-22
// relative, or to 2^-25 absolute where the lo half of |x| < 1/8 is an fp16 subnormal), and a product is three matrix-core products  xh.wh + xh.wl + xl.wh  accumulated in fp32 -- fp32-grade
// results (each fp16 x fp16 product is exact in fp32; the dropped xl.wl term is 2^-22 relative) at a third of the fp16
// MFMA rate instead of the 1/16 of the f32-input MFMA.  Activations live as two NHWC fp16 planes, weights as two planes in
// [cout tile][cin chunk][tap][64 cout][16 cin] (pre-scaled by 2^8 so that the lo halves stay normal numbers).
// Workgroup = 16 x 16 output pixels x 64 output channels; wave = 4 x 16 pixels x 64 channels (2 M-tiles x 2 N-tiles of
// 32x32x16: 8 ds_read_b128 feed 12 MFMAs).
// LDS rows of 16 cin (32 B) sit at a 48-byte pitch: the ds_read_b128 fragment reads of 32 lanes are conflict-free.
// ---------------------------------------------------------------------------------------------------------------
typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr int SK = 16;             // cin per chunk
constexpr int SPT = 24;            // LDS pitch of one 16-cin row in halves (48 B)
constexpr int STY = 16, STX = 16;  // pixel tile
constexpr float WSCALE = 256.f;

template <int DIL>
struct SplitCfg {
    static constexpr int PY = STY + 4 * DIL, PX = STX + 4 * DIL;
    // Occupancy beats conflict-free LDS here (measured: the dilated layer went 17.0 -> 9.9 ms when a second workgroup
    // fitted a CU): dense 32-byte rows (2-way conflicts on the fragment reads) bring the stride-1 layers to three
    // workgroups per CU and the dilated one to two.  XPT/WPT = 24 would be the conflict-free 48-byte pitch.
    static constexpr int XPT = DIL == 1 ? 16 : SPT;
    static constexpr int PXP = DIL == 1 ? PX : 24;  // padded row: 24 px * 48 B = 128 mod 256 B (conflict-free rows)
    static constexpr int X_HALVES = PY * PXP * XPT;
    static constexpr int WPT = 16;
    static constexpr int W_HALVES = 5 * 64 * WPT;
    static constexpr size_t LDS_BYTES = (size_t)(2 * X_HALVES + 2 * W_HALVES) * sizeof(half_t);
    static constexpr size_t LDS_BYTES_SINGLE = (size_t)(X_HALVES + W_HALVES) * sizeof(half_t);   // hi planes only
    static_assert(PX <= PXP, "patch wider than the padded LDS row");
};

// (SINGLE: the hi halves only -- plain fp16 operands, one product per term instead of three: the "fp16" training mode.)
template <int DIL, bool OUT_SPLIT, bool SINGLE = false>
__global__ __launch_bounds__(256, 3) void conv5x5_split_kernel(const half_t* __restrict__ in_hi, const half_t* __restrict__ in_lo,
                                                            const half_t* __restrict__ Wh, const half_t* __restrict__ Wl,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            half_t* __restrict__ out_hi, half_t* __restrict__ out_lo,
                                                            float* __restrict__ out_f32, int H, int W, int Cin, int Cout,
                                                            int relu, int tiles_x, int zero_pad, int xcd_tiles) {
    typedef SplitCfg<DIL> Cfg;
    constexpr int PY = Cfg::PY, PX = Cfg::PX, PXP = Cfg::PXP;
    extern __shared__ __attribute__((aligned(16))) half_t smem_s[];
    // (SINGLE: the hi planes only -- half the LDS, the lo operands are neither staged nor read; in_lo / Wl / out_lo unused)
    half_t* Xh = smem_s;                       // [PY][PXP][SPT]
    half_t* Xl = Xh + Cfg::X_HALVES;
    half_t* Wsh = SINGLE ? Xl : Xl + Cfg::X_HALVES;   // [5 taps][64 cout][SPT]
    half_t* Wsl = Wsh + Cfg::W_HALVES;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int tile_id = blockIdx.x, ct = blockIdx.y;  // pixel tile of the frame, 64-cout tile
    size_t frame = blockIdx.z;
    if (xcd_tiles > 0) {
        // XCD-aware 1-D grid (the training step's launches): workgroup ids go round-robin over the 8 XCDs, so within one XCD
        // consecutive workgroups take the cout tiles of ONE pixel tile -- its input patch comes from HBM once and from that
        // XCD's L2 for the other cout tiles (measured: 1.6 GB fetched per launch with the plain grid, 10x the input).
        const int nct = (Cout + 63) / 64;
        const long long lin = blockIdx.x, j = lin >> 3;
        ct = (int)(j % nct);
        const long long t = (j / nct) * 8 + (lin & 7);   // over frames x tiles
        if (t >= (long long)xcd_tiles) return;           // xcd_tiles = frames * tiles per frame
        const int per_frame = tiles_x * ((H + STY - 1) / STY);
        frame = (size_t)(t / per_frame);
        tile_id = (int)(t % per_frame);
    }
    const int ty = tile_id / tiles_x, tx = tile_id % tiles_x;
    const int y0 = ty * STY, x0 = tx * STX;
    const half_t* fh = in_hi + frame * (size_t)H * W * Cin;
    const half_t* fl = in_lo + frame * (size_t)H * W * Cin;
    const int li = lane & 31, hi = lane >> 5;
    const int ly = w * 4 + (li >> 3), lx = li & 7;   // M-tile m covers columns m*8 .. m*8+7 of the wave's 4 rows
    const int nck = Cin / SK;
    f16v acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    // weights of one tap row (5 taps x 64 cout x 2 pieces, both planes = 5 x 16 bytes per thread) travel through
    // registers one stage ahead, so that their L2 latency hides behind the MFMAs of the current row
    // (macros with named registers: an array captured by a lambda ends up in scratch memory)
    // SINGLE: 5 taps x 64 cout x 2 pieces of the hi plane = 640 pieces of 16 bytes: three per thread (the third for tid < 128)
    uint4 wr0, wr1, wr2, wr3, wr4;
#define DD_WSRC(i) (((tid & 1) ? Wl : Wh) + wbase_ + (size_t)((tid + (i) * 256) >> 2) * SK + ((tid >> 1) & 1) * 8)
#define DD_WDST(i) (((tid & 1) ? Wsl : Wsh) + ((tid + (i) * 256) >> 2) * Cfg::WPT + ((tid >> 1) & 1) * 8)
#define DD_WSRC1(i) (Wh + wbase_ + (size_t)((tid + (i) * 256) >> 1) * SK + (tid & 1) * 8)
#define DD_WDST1(i) (Wsh + ((tid + (i) * 256) >> 1) * Cfg::WPT + (tid & 1) * 8)
#define wload(ck_, ky_)                                                                          \
    do {                                                                                         \
        const size_t wbase_ = ((((size_t)ct * nck + (ck_)) * 25) + (ky_) * 5) * 64 * SK;         \
        if (SINGLE) {                                                                            \
            wr0 = *reinterpret_cast<const uint4*>(DD_WSRC1(0));                                  \
            wr1 = *reinterpret_cast<const uint4*>(DD_WSRC1(1));                                  \
            if (tid < 128) wr2 = *reinterpret_cast<const uint4*>(DD_WSRC1(2));                   \
        } else {                                                                                 \
            wr0 = *reinterpret_cast<const uint4*>(DD_WSRC(0));                                   \
            wr1 = *reinterpret_cast<const uint4*>(DD_WSRC(1));                                   \
            wr2 = *reinterpret_cast<const uint4*>(DD_WSRC(2));                                   \
            wr3 = *reinterpret_cast<const uint4*>(DD_WSRC(3));                                   \
            wr4 = *reinterpret_cast<const uint4*>(DD_WSRC(4));                                   \
        }                                                                                        \
    } while (0)
#define wstore()                                                                                 \
    do {                                                                                         \
        if (SINGLE) {                                                                            \
            *reinterpret_cast<uint4*>(DD_WDST1(0)) = wr0;                                        \
            *reinterpret_cast<uint4*>(DD_WDST1(1)) = wr1;                                        \
            if (tid < 128) *reinterpret_cast<uint4*>(DD_WDST1(2)) = wr2;                         \
        } else {                                                                                 \
            *reinterpret_cast<uint4*>(DD_WDST(0)) = wr0;                                         \
            *reinterpret_cast<uint4*>(DD_WDST(1)) = wr1;                                         \
            *reinterpret_cast<uint4*>(DD_WDST(2)) = wr2;                                         \
            *reinterpret_cast<uint4*>(DD_WDST(3)) = wr3;                                         \
            *reinterpret_cast<uint4*>(DD_WDST(4)) = wr4;                                         \
        }                                                                                        \
    } while (0)
    wload(0, 0);
    for (int ck = 0; ck < nck; ++ck) {
        __syncthreads();
        // input patch, both planes: PY*PX pixels x 2 pieces of 8 cin
        for (int idx = tid; idx < PY * PX * (SINGLE ? 2 : 4); idx += 256) {
            const int plane = SINGLE ? 0 : idx & 1, piece = SINGLE ? idx & 1 : (idx >> 1) & 1, p = SINGLE ? idx >> 1 : idx >> 2;
            const int py = p / PX, px = p - py * PX;
            const int iy = y0 - 2 * DIL + py, ix = x0 - 2 * DIL + px;
            const int gy = reflect(iy, H), gx = reflect(ix, W);
            const half_t* src = (plane ? fl : fh) + ((size_t)gy * W + gx) * Cin + ck * SK + piece * 8;
            uint4 v = *reinterpret_cast<const uint4*>(src);
            // zero padding (the training step's data gradients): positions outside the image contribute nothing; rows /
            // columns past the far edge of a partial tile are never used by a stored output either way
            if (zero_pad && (iy < 0 || iy >= H || ix < 0 || ix >= W)) v = uint4{0u, 0u, 0u, 0u};
            *reinterpret_cast<uint4*>((plane ? Xl : Xh) + (py * PXP + px) * Cfg::XPT + piece * 8) = v;
        }
        for (int ky = 0; ky < 5; ++ky) {
            if (ky) __syncthreads();
            wstore();  // weights of this tap row (requested while the previous row's MFMAs ran)
            __syncthreads();
            if (ky < 4) wload(ck, ky + 1);
            else if (ck + 1 < nck) wload(ck + 1, 0);
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                h8 xh[2], xl[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int p = (ly + ky * DIL) * PXP + m * 8 + lx + kx * DIL;
                    xh[m] = *reinterpret_cast<const h8*>(Xh + p * Cfg::XPT + hi * 8);
                    if (!SINGLE) xl[m] = *reinterpret_cast<const h8*>(Xl + p * Cfg::XPT + hi * 8);
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int wr = (kx * 64 + n * 32 + li) * Cfg::WPT + hi * 8;
                    const h8 wh = *reinterpret_cast<const h8*>(Wsh + wr);
                    h8 wl = wh;
                    if (!SINGLE) wl = *reinterpret_cast<const h8*>(Wsl + wr);
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        if (!SINGLE) {
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[m], wh, acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[m], wl, acc[m][n], 0, 0, 0);
                        }
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[m], wh, acc[m][n], 0, 0, 0);
                    }
                }
            }
        }
    }
#undef wload
#undef wstore
#undef DD_WSRC
#undef DD_WDST
#undef DD_WSRC1
#undef DD_WDST1
    // D[i][j]: j = lane&31 (cout), i = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel of the M-tile's 4 x 8 block)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int co = ct * 64 + n * 32 + li;
        if (co >= Cout) continue;
        const float sc = scale ? scale[co] : 1.f, sh = shift ? shift[co] : 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int oy = y0 + w * 4 + (i >> 3), ox = x0 + m * 8 + (i & 7);
                if (oy < H && ox < W) {
                    float v = (acc[m][n][r] * (1.f / WSCALE)) * sc + sh;
                    if (relu) v = fmaxf(v, 0.f);
                    const size_t o = frame * (size_t)H * W * Cout + ((size_t)oy * W + ox) * Cout + co;
                    if (OUT_SPLIT) {
                        const half_t vh = (half_t)v;
                        out_hi[o] = vh;
                        if (!SINGLE) out_lo[o] = (half_t)(v - (float)vh);
                    } else {
                        out_f32[o] = v;
                    }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// first layer (3 -> 64 channels) on the split-fp16 MFMA as well.  The frame is first rewritten as two NHWC planes with
// 4 channels per pixel (hi / lo halves, channel 3 = 0): 8 bytes per pixel and plane, so that the K index
// k = tap * 4 + channel (25 taps -> K = 100, padded to 112 = 7 k-steps of 16) makes every 8-wide A fragment two
// neighbouring taps = two 8-byte LDS reads, with no im2col gather.  The 64 x 112 weights (two planes) live in
// registers for the lifetime of a workgroup.  A = pixels (rows), B = output channels (columns): in D a lane holds one
// channel of 16 pixels and a store instruction writes 128 contiguous bytes (32 channels) of two pixels.
// ---------------------------------------------------------------------------------------------------------------
constexpr int C1_KS = 7;             // k-steps of 16: K = 112 >= 25 taps x 4 channels
constexpr int C1_K = 16 * C1_KS;
constexpr int C1_TY = 16, C1_TX = 8;  // output tile; wave = 4 rows x 8 columns
typedef _Float16 h4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void video_split4_kernel(const float* __restrict__ video, h4v* __restrict__ hi_,
                                                           h4v* __restrict__ lo_, int H, int W, long long npix) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // over frames * H * W
    if (i >= npix) return;
    const long long f = i / ((long long)H * W), p = i - f * (long long)H * W;
    const float* fr = video + f * 3 * (long long)H * W + p;
    h4v h, l;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = fr[(long long)c * H * W];
        h[c] = (half_t)v;
        l[c] = (half_t)(v - (float)h[c]);
    }
    h[3] = (half_t)0.f;
    l[3] = (half_t)0.f;
    hi_[i] = h;
    lo_[i] = l;
}

// [64][3][5][5] fp32 -> split planes [64 cout][112 k] of w * 2^8, k = tap * 4 + channel
__global__ void pack_conv1_split_kernel(const float* __restrict__ w, half_t* __restrict__ Wh, half_t* __restrict__ Wl) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 64 * C1_K) return;
    const int co = idx / C1_K, k = idx - co * C1_K;
    const int tap = k >> 2, c = k & 3;
    const float v = (tap < 25 && c < 3) ? w[((size_t)co * 3 + c) * 25 + tap] * WSCALE : 0.f;
    const half_t h = (half_t)v;
    Wh[idx] = h;
    Wl[idx] = (half_t)(v - (float)h);
}

// (SINGLE: plain fp16 operands -- the hi planes only, one product per term -- and an fp16 NHWC output)
template <bool SINGLE>
__global__ __launch_bounds__(256) void conv1_split_kernel(const h4v* __restrict__ in_hi, const h4v* __restrict__ in_lo,
                                                          const half_t* __restrict__ Wh, const half_t* __restrict__ Wl,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* __restrict__ out, half_t* __restrict__ out16, int H, int W,
                                                          int tiles_x) {
    constexpr int PY = C1_TY + 4, PX = C1_TX + 4;
    __shared__ h4v Xh[PY * PX], Xl[PY * PX];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
    const int y0 = ty * C1_TY, x0 = tx * C1_TX;
    const size_t frame = blockIdx.z;
    const int li = lane & 31, hh = lane >> 5;
    // weights: B operand, lane (cout li of column tile n, half hh) holds k = 16 ks + 8 hh .. + 7
    h8 wh[2][C1_KS], wl[2][C1_KS];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int ks = 0; ks < C1_KS; ++ks) {
            const size_t o = (size_t)(n * 32 + li) * C1_K + ks * 16 + hh * 8;
            wh[n][ks] = *reinterpret_cast<const h8*>(Wh + o);
            if (!SINGLE) wl[n][ks] = *reinterpret_cast<const h8*>(Wl + o);
        }
    const h4v* fh = in_hi + frame * (size_t)H * W;
    const h4v* fl = in_lo + frame * (size_t)H * W;
    for (int idx = tid; idx < PY * PX; idx += 256) {
        const int py = idx / PX, px = idx - py * PX;
        const size_t g = (size_t)reflect(y0 - 2 + py, H) * W + reflect(x0 - 2 + px, W);
        Xh[idx] = fh[g];
        if (!SINGLE) Xl[idx] = fl[g];
    }
    __syncthreads();
    const int ly = w * 4 + (li >> 3), lx = li & 7;  // this lane's pixel (row of the A operand)
    f16v acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < C1_KS; ++ks) {
        // the 8 k values of this lane: taps t0 = 2 (2 ks + hh) and t0 + 1 (taps >= 25 carry zero weights: any valid pixel)
        const int t0 = min(4 * ks + 2 * hh, 24), t1 = min(4 * ks + 2 * hh + 1, 24);
        const int p0 = (ly + t0 / 5) * PX + lx + t0 % 5, p1 = (ly + t1 / 5) * PX + lx + t1 % 5;
        const h4v a0 = Xh[p0], a1 = Xh[p1];
        const h8 xh = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        h8 xl = xh;
        if (!SINGLE) {
            const h4v b0 = Xl[p0], b1 = Xl[p1];
            xl = h8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (!SINGLE) {
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh[n][ks], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl[n][ks], acc[n], 0, 0, 0);
            }
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh[n][ks], acc[n], 0, 0, 0);
        }
    }
    // D[i][j]: j = lane & 31 (cout), i = (r&3) + 8 (r>>2) + 4 hh (pixel of the wave's 4 x 8 block); BN fold + ReLU
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int co = n * 32 + li;
        const float sc = scale[co] * (1.f / WSCALE), sh = shift[co];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const int oy = y0 + w * 4 + (i >> 3), ox = x0 + (i & 7);
            if (oy < H && ox < W) {
                const size_t o = (frame * (size_t)H * W + (size_t)oy * W + ox) * 64 + co;
                const float v = fmaxf(acc[n][r] * sc + sh, 0.f);
                if (SINGLE) out16[o] = (half_t)v;
                else out[o] = v;
            }
        }
    }
}

// NHWC blur-pool on split planes: one thread per (output pixel, 8 channels)
__global__ __launch_bounds__(256) void blurpool_split_kernel(const half_t* __restrict__ in_hi, const half_t* __restrict__ in_lo,
                                                             half_t* __restrict__ out_hi, half_t* __restrict__ out_lo, int H,
                                                             int W, int Ho, int Wo, int C) {
    const size_t frame = blockIdx.y;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c8 = C / 8;
    if (idx >= (long long)Ho * Wo * c8) return;
    const int cq = (int)(idx % c8);
    const int p = (int)(idx / c8);
    const int oy = p / Wo, ox = p - oy * Wo;
    const half_t* fh = in_hi + frame * (size_t)H * W * C;
    const half_t* fl = in_lo + frame * (size_t)H * W * C;
    const float f[4] = {1.f, 3.f, 3.f, 1.f};
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gy = reflect(2 * oy - 1 + i, H);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gx = reflect(2 * ox - 1 + j, W);
            const float wgt = f[i] * f[j] * (1.f / 64.f);
            const size_t o = ((size_t)gy * W + gx) * C + cq * 8;
            const uint4 uh = *reinterpret_cast<const uint4*>(fh + o), ul = *reinterpret_cast<const uint4*>(fl + o);
            const h8 vh = *reinterpret_cast<const h8*>(&uh), vl = *reinterpret_cast<const h8*>(&ul);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(wgt, (float)vh[e] + (float)vl[e], acc[e]);
        }
    }
    h8 oh, ol;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        oh[e] = (half_t)acc[e];
        ol[e] = (half_t)(acc[e] - (float)oh[e]);
    }
    const size_t oo = frame * (size_t)Ho * Wo * C + ((size_t)oy * Wo + ox) * C + cq * 8;
    *reinterpret_cast<h8*>(out_hi + oo) = oh;
    *reinterpret_cast<h8*>(out_lo + oo) = ol;
}

// NHWC blur-pool on one fp16 plane (the fp16 operand mode: every activation is a single fp16 NHWC plane): one thread per
// (output pixel, 8 channels); fp32 accumulation of the 16 taps
__global__ __launch_bounds__(256) void blurpool_half_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int H, int W,
                                                            int Ho, int Wo, int C) {
    const size_t frame = blockIdx.y;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c8 = C / 8;
    if (idx >= (long long)Ho * Wo * c8) return;
    const int cq = (int)(idx % c8);
    const int p = (int)(idx / c8);
    const int oy = p / Wo, ox = p - oy * Wo;
    const half_t* fin = in + frame * (size_t)H * W * C;
    const float f[4] = {1.f, 3.f, 3.f, 1.f};
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gy = reflect(2 * oy - 1 + i, H);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gx = reflect(2 * ox - 1 + j, W);
            const float wgt = f[i] * f[j] * (1.f / 64.f);
            const uint4 u = *reinterpret_cast<const uint4*>(fin + ((size_t)gy * W + gx) * C + cq * 8);
            const h8 v = *reinterpret_cast<const h8*>(&u);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(wgt, (float)v[e], acc[e]);
        }
    }
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)acc[e];
    *reinterpret_cast<h8*>(out + frame * (size_t)Ho * Wo * C + ((size_t)oy * Wo + ox) * C + cq * 8) = o;
}

// NHWC blur-pool of the fp32 first-layer output, written as split planes: one thread per (output pixel, 4 channels)
__global__ __launch_bounds__(256) void blurpool_to_split_kernel(const float* __restrict__ in, half_t* __restrict__ out_hi,
                                                                half_t* __restrict__ out_lo, int H, int W, int Ho, int Wo,
                                                                int C) {
    const size_t frame = blockIdx.y;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    if (idx >= (long long)Ho * Wo * c4) return;
    const int cq = (int)(idx % c4);
    const int p = (int)(idx / c4);
    const int oy = p / Wo, ox = p - oy * Wo;
    const float* fin = in + frame * (size_t)H * W * C;
    const float f[4] = {1.f, 3.f, 3.f, 1.f};
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gy = reflect(2 * oy - 1 + i, H);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gx = reflect(2 * ox - 1 + j, W);
            const float wgt = f[i] * f[j] * (1.f / 64.f);
            const float4 v = *reinterpret_cast<const float4*>(fin + ((size_t)gy * W + gx) * C + cq * 4);
            acc[0] = fmaf(wgt, v.x, acc[0]);
            acc[1] = fmaf(wgt, v.y, acc[1]);
            acc[2] = fmaf(wgt, v.z, acc[2]);
            acc[3] = fmaf(wgt, v.w, acc[3]);
        }
    }
    typedef _Float16 h4v __attribute__((ext_vector_type(4)));
    h4v oh, ol;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        oh[e] = (half_t)acc[e];
        ol[e] = (half_t)(acc[e] - (float)oh[e]);
    }
    const size_t oo = frame * (size_t)Ho * Wo * C + ((size_t)oy * Wo + ox) * C + cq * 4;
    *reinterpret_cast<h4v*>(out_hi + oo) = oh;
    *reinterpret_cast<h4v*>(out_lo + oo) = ol;
}

// [Cout][Cin][5][5] fp32 -> split planes [cout tile][cin chunk][25][64][16] of w * 2^8
// (flip_transpose: the operator of the data gradient -- w is then [Cin][Cout][5][5] and tap t reads tap 24 - t)
__global__ void pack_split_kernel(const float* __restrict__ w, int Cin, int Cout, half_t* __restrict__ Wh, half_t* __restrict__ Wl,
                                  int flip_transpose) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int nck = Cin / SK, nct = (Cout + 63) / 64;
    const long long total = (long long)nct * nck * 25 * 64 * SK;
    if (idx >= total) return;
    const int ci_l = (int)(idx % SK);
    const int co_l = (int)((idx / SK) % 64);
    const int tap = (int)((idx / (SK * 64)) % 25);
    const int ck = (int)((idx / (SK * 64 * 25)) % nck);
    const int ct = (int)(idx / ((long long)SK * 64 * 25 * nck));
    const int co = ct * 64 + co_l, ci = ck * SK + ci_l;
    const size_t src = flip_transpose ? ((size_t)ci * Cout + co) * 25 + (24 - tap) : ((size_t)co * Cin + ci) * 25 + tap;
    const float v = (co < Cout) ? w[src] * WSCALE : 0.f;
    const half_t h = (half_t)v;
    Wh[idx] = h;
    Wl[idx] = (half_t)(v - (float)h);
}

// NHWC blur-pool: one thread per (output pixel, 4 channels)
__global__ __launch_bounds__(256) void blurpool_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                       int Ho, int Wo, int C) {
    const size_t frame = blockIdx.y;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    if (idx >= (long long)Ho * Wo * c4) return;
    const int cq = (int)(idx % c4);
    const int p = (int)(idx / c4);
    const int oy = p / Wo, ox = p - oy * Wo;
    const float* fin = in + frame * (size_t)H * W * C;
    const float f[4] = {1.f, 3.f, 3.f, 1.f};
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gy = reflect(2 * oy - 1 + i, H);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gx = reflect(2 * ox - 1 + j, W);
            const float wgt = f[i] * f[j] * (1.f / 64.f);
            const float4 v = *reinterpret_cast<const float4*>(fin + ((size_t)gy * W + gx) * C + cq * 4);
            acc.x = fmaf(wgt, v.x, acc.x);
            acc.y = fmaf(wgt, v.y, acc.y);
            acc.z = fmaf(wgt, v.z, acc.z);
            acc.w = fmaf(wgt, v.w, acc.w);
        }
    }
    *reinterpret_cast<float4*>(out + frame * (size_t)Ho * Wo * C + ((size_t)oy * Wo + ox) * C + cq * 4) = acc;
}

// one wave per ViT cell: refined = dino + bilinear(cnn), norm
__global__ __launch_bounds__(256) void align_add_kernel(const float* __restrict__ cnn, const float* __restrict__ dino,
                                                        float* __restrict__ out, float* __restrict__ norms, int ch, int cw,
                                                        int ph, int pw, int C, int patch, int vit_stride, int cnn_stride,
                                                        int nframes) {
    const long long cell = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int HW = ph * pw;
    if (cell >= (long long)nframes * HW) return;
    const int lane = threadIdx.x & 63;
    const int fr = (int)(cell / HW), k = (int)(cell % HW);
    const int r = k / pw, c = k % pw;
    // models/utils.py:30-41: grid = -1 - 1/c_br + 2*pos/c_br, then grid_sample(align_corners, border)
    const float brx = (float)((cw - 1) * cnn_stride), bry = (float)((ch - 1) * cnn_stride);
    const float vx = (float)c * (float)vit_stride + (float)patch / 2.f;
    const float vy = (float)r * (float)vit_stride + (float)patch / 2.f;
    const float gx = -1.f - (1.f / brx) + (2.f * vx / brx);
    const float gy = -1.f - (1.f / bry) + (2.f * vy / bry);
    float ix = ((gx + 1.f) / 2.f) * (float)(cw - 1);
    float iy = ((gy + 1.f) / 2.f) * (float)(ch - 1);
    ix = fminf(fmaxf(ix, 0.f), (float)(cw - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(ch - 1));
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float fx = ix - x0f, fy = iy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x1 = min(x0 + 1, cw - 1), y1 = min(y0 + 1, ch - 1);
    const float* base = cnn + (size_t)fr * ch * cw * C;
    const float* p00 = base + ((size_t)y0 * cw + x0) * C;
    const float* p01 = base + ((size_t)y0 * cw + x1) * C;
    const float* p10 = base + ((size_t)y1 * cw + x0) * C;
    const float* p11 = base + ((size_t)y1 * cw + x1) * C;
    const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
    const float* d = dino + cell * C;
    float* o = out + cell * C;
    float ss = 0.f;
    for (int q = lane * 4; q < C; q += 256) {
        const float4 a = *reinterpret_cast<const float4*>(p00 + q), b = *reinterpret_cast<const float4*>(p01 + q);
        const float4 e = *reinterpret_cast<const float4*>(p10 + q), f = *reinterpret_cast<const float4*>(p11 + q);
        const float4 dv = *reinterpret_cast<const float4*>(d + q);
        float4 v;
        v.x = dv.x + (a.x * w00 + b.x * w01 + e.x * w10 + f.x * w11);
        v.y = dv.y + (a.y * w00 + b.y * w01 + e.y * w10 + f.y * w11);
        v.z = dv.z + (a.z * w00 + b.z * w01 + e.z * w10 + f.z * w11);
        v.w = dv.w + (a.w * w00 + b.w * w01 + e.w * w10 + f.w * w11);
        *reinterpret_cast<float4*>(o + q) = v;
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss = wave_sum(ss);
    if (lane == 0 && norms) norms[cell] = sqrtf(ss);
}

// [Cout][Cin][5][5] -> [25][CinP][CoutP]; BN(eval) folded: y = conv*scale + shift
__global__ void pack_conv_kernel(const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ bn_w,
                                 const float* __restrict__ bn_b, const float* __restrict__ bn_mean,
                                 const float* __restrict__ bn_var, float eps, int Cin, int Cout, int CinP, int CoutP,
                                 float* __restrict__ Wk, float* __restrict__ scale, float* __restrict__ shift) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = 25LL * CinP * CoutP;
    if (idx < total) {
        const int co = (int)(idx % CoutP);
        const int ci = (int)((idx / CoutP) % CinP);
        const int tap = (int)(idx / ((long long)CoutP * CinP));
        Wk[idx] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 25 + tap] : 0.f;
    }
    if (idx < CoutP) {
        const int co = (int)idx;
        if (co < Cout) {
            const float s = bn_w[co] / sqrtf(bn_var[co] + eps);
            scale[co] = s;
            shift[co] = (bias[co] - bn_mean[co]) * s + bn_b[co];
        } else {
            scale[co] = 0.f;
            shift[co] = 0.f;
        }
    }
}

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }
inline int pool_out(int n) { return (n - 1) / 2 + 1; }

struct Plan {
    int H[4], W[4];       // conv input/output spatial size of layers 0..3
    int Cin[4], Cout[4];
    size_t act[4], pool[3];  // float offsets in the workspace (per batch)
    size_t vsplit;           // NHWC4 hi / lo planes of the input frames (4 floats' worth of bytes per pixel)
    size_t total_floats;
};

Plan make_plan(int video_h, int video_w, int C, int fb) {
    Plan p;
    const int chans[5] = {3, 64, 128, 256, C};
    int h = video_h, w = video_w;
    size_t off = 0;
    for (int l = 0; l < 4; ++l) {
        p.H[l] = h; p.W[l] = w; p.Cin[l] = chans[l]; p.Cout[l] = chans[l + 1];
        p.act[l] = off; off += (size_t)fb * h * w * chans[l + 1];
        if (l < 3) {
            h = pool_out(h); w = pool_out(w);
            p.pool[l] = off; off += (size_t)fb * h * w * chans[l + 1];
        }
    }
    p.vsplit = off; off += (size_t)fb * video_h * video_w * 4;
    p.total_floats = off;
    return p;
}

inline size_t f32_packed_floats(int cinp, int coutp) { return (size_t)25 * cinp * coutp + 2 * (size_t)coutp; }
// halves in ONE split weight plane == floats taken by the two planes together
inline size_t split_plane_halves(int cin, int cout) { return (size_t)((cout + 63) / 64) * (cin / SK) * 25 * 64 * SK; }

// DTK_DEV builds, DTK_DEBUG bit 1024: run layers 2-4 on the fp32-input MFMA kernel instead of the split-fp16 one
inline bool dd_force_f32() { return DTK_DBG(dtk_dev_flags(), 1024) != 0; }

}  // namespace

extern "C" size_t dtk_delta_dino_packed_floats(int layer, int C) {
    const int chans[5] = {3, 64, 128, 256, C};
    if (layer < 0 || layer > 3 || C <= 0) return 0;
    const int cinp = pad_to(chans[layer], CK), coutp = pad_to(chans[layer + 1], TNC);
    return f32_packed_floats(cinp, coutp) + (layer ? split_plane_halves(chans[layer], chans[layer + 1]) : (size_t)64 * C1_K);
}

extern "C" int dtk_delta_dino_pack(int layer, int C, const float* w, const float* bias, const float* bn_w,
                                   const float* bn_b, const float* bn_mean, const float* bn_var, float eps, float* packed,
                                   void* stream) {
    DTK_REQUIRE(layer >= 0 && layer <= 3 && C > 0 && w && bias && bn_w && bn_b && bn_mean && bn_var && packed,
                "dtk_delta_dino_pack: bad args");
    const int chans[5] = {3, 64, 128, 256, C};
    const int cin = chans[layer], cout = chans[layer + 1];
    const int cinp = pad_to(cin, CK), coutp = pad_to(cout, TNC);
    float* Wk = packed;
    float* scale = packed + (size_t)25 * cinp * coutp;
    float* shift = scale + coutp;
    const long long total = 25LL * cinp * coutp;
    DTK_LAUNCH("dd_pack", pack_conv_kernel, dim3(dtk_cdiv(total, 256)), dim3(256), 0, dtk_stream(stream), w, bias, bn_w,
               bn_b, bn_mean, bn_var, eps, cin, cout, cinp, coutp, Wk, scale, shift);
    if (layer == 0) {
        half_t* Wh = reinterpret_cast<half_t*>(packed + f32_packed_floats(cinp, coutp));
        DTK_LAUNCH("dd_pack", pack_conv1_split_kernel, dim3(dtk_cdiv(64 * C1_K, 256)), dim3(256), 0, dtk_stream(stream), w, Wh,
                   Wh + 64 * C1_K);
    } else {
        half_t* Wh = reinterpret_cast<half_t*>(packed + f32_packed_floats(cinp, coutp));
        const size_t nh = split_plane_halves(cin, cout);
        DTK_LAUNCH("dd_pack", pack_split_kernel, dim3(dtk_cdiv((long long)nh, 256)), dim3(256), 0, dtk_stream(stream), w, cin,
                   cout, Wh, Wh + nh, 0);
    }
    return DTK_OK;
}

constexpr int DD_FRAME_BATCH = 8;

extern "C" size_t dtk_delta_dino_workspace_bytes(const dtk_geom* g) {
    if (!g) return 0;
    const int fb = g->T < DD_FRAME_BATCH ? g->T : DD_FRAME_BATCH;
    return make_plan(g->video_h, g->video_w, g->C, fb).total_floats * sizeof(float);
}

extern "C" int dtk_delta_dino_refine(const dtk_geom* g, const float* video, const float* dino, const float* const* packed,
                                     float* out, float* norms, int t0, int nframes, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    return dtk_delta_dino_refine_mode(g, video, dino, packed, out, norms, t0, nframes, DTK_DD_SPLIT, workspace, workspace_bytes,
                                      stream);
}

extern "C" int dtk_delta_dino_refine_mode(const dtk_geom* g, const float* video, const float* dino,
                                          const float* const* packed, float* out, float* norms, int t0, int nframes,
                                          int operands, void* workspace, size_t workspace_bytes, void* stream) {
    DTK_REQUIRE(operands == DTK_DD_SPLIT || operands == DTK_DD_FP16, "dtk_delta_dino_refine_mode: unknown operand mode");
    const bool single = operands == DTK_DD_FP16;
    DTK_REQUIRE(g && video && dino && packed && out && workspace, "dtk_delta_dino_refine: null pointer");
    DTK_REQUIRE(g->C % 4 == 0, "dtk_delta_dino_refine: C must be a multiple of 4");
    DTK_REQUIRE(t0 >= 0 && nframes >= 0 && t0 + nframes <= g->T, "dtk_delta_dino_refine: frame range out of bounds");
    DTK_REQUIRE(g->video_h >= 16 && g->video_w >= 16, "dtk_delta_dino_refine: video too small");
    const int fb = g->T < DD_FRAME_BATCH ? g->T : DD_FRAME_BATCH;
    const Plan p = make_plan(g->video_h, g->video_w, g->C, fb);
    if (workspace_bytes < p.total_floats * sizeof(float)) {
        dtk_set_error("dtk_delta_dino_refine: workspace %zu B < required %zu B", workspace_bytes,
                      p.total_floats * sizeof(float));
        return DTK_E_WORKSPACE;
    }
    hipStream_t st = dtk_stream(stream);
    float* ws = reinterpret_cast<float*>(workspace);
    const int HW = g->ph * g->pw;
    static const bool lds_ok = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5x5_split_kernel<1, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)SplitCfg<1>::LDS_BYTES) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5x5_split_kernel<2, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)SplitCfg<2>::LDS_BYTES) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5x5_split_kernel<1, true, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)SplitCfg<1>::LDS_BYTES_SINGLE) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5x5_split_kernel<2, false, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)SplitCfg<2>::LDS_BYTES_SINGLE) == hipSuccess;
    }();
    DTK_REQUIRE(lds_ok, "dtk_delta_dino_refine: cannot reserve LDS for the split-fp16 convolution");
    for (int f0 = t0; f0 < t0 + nframes; f0 += fb) {
        const int nf = (t0 + nframes - f0) < fb ? (t0 + nframes - f0) : fb;
        const float* cur = video + (size_t)f0 * 3 * g->video_h * g->video_w;
        const bool split = !dd_force_f32();
        // DTK_DD_FP16 (`single`): every activation is ONE fp16 NHWC plane (the first half of its fp32-sized slot) and every
        // product one MFMA: the lo halves are neither written, staged nor multiplied.  The last layer's output stays fp32.
        const bool half_path = single && split;
        for (int l = 0; l < 4; ++l) {
            const int H = p.H[l], W = p.W[l], cin = p.Cin[l], cout = p.Cout[l];
            const int cinp = pad_to(cin, CK), coutp = pad_to(cout, TNC);
            const float* Wk = packed[l];
            const float* scale = Wk + (size_t)25 * cinp * coutp;
            const float* shift = scale + coutp;
            float* act = ws + p.act[l];
            // split activations: hi plane in the first half of the fp32-sized slot, lo plane in the second
            const size_t in_n = (size_t)nf * H * W * cin, out_n = (size_t)nf * H * W * cout;
            if (l == 0 && split) {
                h4v* vh = reinterpret_cast<h4v*>(ws + p.vsplit);
                h4v* vl = vh + (size_t)nf * H * W;
                const long long npix = (long long)nf * H * W;
                DTK_LAUNCH("dd_video_split", video_split4_kernel, dim3(dtk_cdiv(npix, 256)), dim3(256), 0, st, cur, vh, vl, H, W,
                           npix);
                const half_t* Wh = reinterpret_cast<const half_t*>(Wk + f32_packed_floats(cinp, coutp));
                const int tiles_x = dtk_cdiv(W, C1_TX), tiles_y = dtk_cdiv(H, C1_TY);
                if (half_path) {
                    DTK_LAUNCH("dd_conv1", conv1_split_kernel<true>, dim3(tiles_x * tiles_y, 1, nf), dim3(256), 0, st, vh, vl, Wh,
                               Wh + 64 * C1_K, scale, shift, (float*)nullptr, reinterpret_cast<half_t*>(act), H, W, tiles_x);
                } else {
                    DTK_LAUNCH("dd_conv1", conv1_split_kernel<false>, dim3(tiles_x * tiles_y, 1, nf), dim3(256), 0, st, vh, vl, Wh,
                               Wh + 64 * C1_K, scale, shift, act, (half_t*)nullptr, H, W, tiles_x);
                }
            } else if (l == 0 || !split) {
                const int tiles_x = dtk_cdiv(W, TP), tiles_y = dtk_cdiv(H, TP);
                dim3 grid(tiles_x * tiles_y, coutp / TNC, nf);
                if (l == 0) {
                    DTK_LAUNCH("dd_conv1", (conv5x5_kernel<1, true>), grid, dim3(256), 0, st, cur, Wk, scale, shift, act, H,
                               W, cin, cinp, cout, coutp, 1, tiles_x);
                } else if (l < 3) {
                    DTK_LAUNCH("dd_conv23", (conv5x5_kernel<1, false>), grid, dim3(256), 0, st, cur, Wk, scale, shift, act,
                               H, W, cin, cinp, cout, coutp, 1, tiles_x);
                } else {
                    DTK_LAUNCH("dd_conv4", (conv5x5_kernel<2, false>), grid, dim3(256), 0, st, cur, Wk, scale, shift, act, H,
                               W, cin, cinp, cout, coutp, 0, tiles_x);
                }
            } else {
                const half_t* Wh = reinterpret_cast<const half_t*>(Wk + f32_packed_floats(cinp, coutp));
                const half_t* Wl = Wh + split_plane_halves(cin, cout);
                const half_t* ih = reinterpret_cast<const half_t*>(cur);
                half_t* oh = reinterpret_cast<half_t*>(act);
                const int tiles_x = dtk_cdiv(W, STX), tiles_y = dtk_cdiv(H, STY);
                dim3 grid(tiles_x * tiles_y, (cout + 63) / 64, nf);
                if (l < 3 && single) {
                    DTK_LAUNCH("dd_conv23", (conv5x5_split_kernel<1, true, true>), grid, dim3(256), SplitCfg<1>::LDS_BYTES_SINGLE,
                               st, ih, ih, Wh, Wl, scale, shift, oh, oh, (float*)nullptr, H, W, cin, cout, 1, tiles_x, 0, 0);
                } else if (single) {
                    DTK_LAUNCH("dd_conv4", (conv5x5_split_kernel<2, false, true>), grid, dim3(256), SplitCfg<2>::LDS_BYTES_SINGLE,
                               st, ih, ih, Wh, Wl, scale, shift, (half_t*)nullptr, (half_t*)nullptr, act, H, W, cin, cout, 0,
                               tiles_x, 0, 0);
                } else if (l < 3) {
                    DTK_LAUNCH("dd_conv23", (conv5x5_split_kernel<1, true>), grid, dim3(256), SplitCfg<1>::LDS_BYTES, st, ih,
                               ih + in_n, Wh, Wl, scale, shift, oh, oh + out_n, (float*)nullptr, H, W, cin, cout, 1, tiles_x, 0, 0);
                } else {
                    DTK_LAUNCH("dd_conv4", (conv5x5_split_kernel<2, false>), grid, dim3(256), SplitCfg<2>::LDS_BYTES, st, ih,
                               ih + in_n, Wh, Wl, scale, shift, (half_t*)nullptr, (half_t*)nullptr, act, H, W, cin, cout, 0,
                               tiles_x, 0, 0);
                }
            }
            cur = act;
            if (l < 3) {
                const int Ho = pool_out(H), Wo = pool_out(W);
                float* pl = ws + p.pool[l];
                const size_t pool_n = (size_t)nf * Ho * Wo * cout;
                if (!split) {
                    const long long n = (long long)Ho * Wo * (cout / 4);
                    DTK_LAUNCH("dd_blurpool", blurpool_kernel, dim3(dtk_cdiv(n, 256), nf), dim3(256), 0, st, act, pl, H, W,
                               Ho, Wo, cout);
                } else if (half_path) {
                    const long long n = (long long)Ho * Wo * (cout / 8);
                    DTK_LAUNCH("dd_blurpool", blurpool_half_kernel, dim3(dtk_cdiv(n, 256), nf), dim3(256), 0, st,
                               reinterpret_cast<const half_t*>(act), reinterpret_cast<half_t*>(pl), H, W, Ho, Wo, cout);
                } else if (l == 0) {
                    const long long n = (long long)Ho * Wo * (cout / 4);
                    half_t* ph_ = reinterpret_cast<half_t*>(pl);
                    DTK_LAUNCH("dd_blurpool", blurpool_to_split_kernel, dim3(dtk_cdiv(n, 256), nf), dim3(256), 0, st, act,
                               ph_, ph_ + pool_n, H, W, Ho, Wo, cout);
                } else {
                    const long long n = (long long)Ho * Wo * (cout / 8);
                    const half_t* ah = reinterpret_cast<const half_t*>(act);
                    half_t* ph_ = reinterpret_cast<half_t*>(pl);
                    DTK_LAUNCH("dd_blurpool", blurpool_split_kernel, dim3(dtk_cdiv(n, 256), nf), dim3(256), 0, st, ah,
                               ah + out_n, ph_, ph_ + pool_n, H, W, Ho, Wo, cout);
                }
                cur = pl;
            }
        }
        const long long cells = (long long)nf * HW;
        DTK_LAUNCH("dd_align_add", align_add_kernel, dim3(dtk_cdiv(cells, 4)), dim3(256), 0, st, cur,
                   dino + (size_t)f0 * HW * g->C, out + (size_t)f0 * HW * g->C, norms ? norms + (size_t)f0 * HW : nullptr,
                   p.H[3], p.W[3], g->ph, g->pw, g->C, g->patch, g->stride, 8, nf);
    }
    return DTK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The split-fp16 implicit-GEMM convolution as a stand-alone operator for the TRAINING step (delta_dino.py:29-44 under
// autograd; dino_tracker_amd/train_ops.py _ConvMfma): forward and data gradient of the 5 x 5 layers run on
// conv5x5_split_kernel -- no unfolded operand in memory (round 3's first form wrote and re-read 10 GB of im2col columns per
// iteration) -- between two layout kernels, because the training pipeline around it (batch-statistics BatchNorm, blur-pool,
// autograd) is NCHW fp32:
//   nchw_to_split_kernel   x [N][C][H][W] fp32 (times a power-of-two device scalar)  ->  hi / lo fp16 planes [N][H+2b][W+2b][C],
//                          a ring of b zero pixels around the image (the data gradient of a reflect-padded convolution is a
//                          zero-padded convolution of dY over the PADDED domain);
//   nhwc_to_nchw_kernel    y [N][H+2b][W+2b][C] fp32  ->  [N][C][H][W], optionally folding the ring back with the adjoint of the
//                          reflect padding (a pixel within b of the border also collects what its mirror images received)
//                          and dividing by the scalar.
// Both move 64 pixels x 64 channels through an LDS tile so that global reads and writes are contiguous on both sides.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int LT = 64;  // tile: 64 pixels x 64 channels

__global__ __launch_bounds__(256) void nchw_to_split_kernel(const float* __restrict__ x, half_t* __restrict__ hi,
                                                            half_t* __restrict__ lo, int C, int H, int W, int b,
                                                            const float* __restrict__ scale) {
    __shared__ float tile[LT][LT + 1];  // [channel][pixel]
    const int He = H + 2 * b, We = W + 2 * b;
    const long long Le = (long long)He * We;
    const int n = blockIdx.z, c0 = blockIdx.y * LT;
    const long long p0 = (long long)blockIdx.x * LT;
    const int tid = threadIdx.x;
    const float sc = scale ? *scale : 1.f;
    {   // read: lanes along pixels (contiguous in a row of the source plane)
        const int px = tid & 63;
        const long long pe = p0 + px;
        const int ye = (int)(pe / We), xe = (int)(pe - (long long)ye * We);
        const int y = ye - b, xx = xe - b;
        const bool in = pe < Le && y >= 0 && y < H && xx >= 0 && xx < W;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int ch = i * 4 + (tid >> 6);
            const int c = c0 + ch;
            tile[ch][px] = (in && c < C) ? x[(((size_t)n * C + c) * H + y) * W + xx] * sc : 0.f;
        }
    }
    __syncthreads();
    // write: 8 lanes cover the 64 channels of one pixel (128 contiguous bytes per plane)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int px = i * 32 + (tid >> 3), cg = (tid & 7) * 8;
        const long long pe = p0 + px;
        if (pe >= Le || c0 + cg >= C) continue;
        h8 vh, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = tile[cg + e][px];
            vh[e] = (half_t)v;
            vl[e] = (half_t)(v - (float)vh[e]);
        }
        const size_t o = ((size_t)n * Le + pe) * C + c0 + cg;
        *reinterpret_cast<h8*>(hi + o) = vh;
        *reinterpret_cast<h8*>(lo + o) = vl;
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ y, float* __restrict__ out, int C, int H,
                                                           int W, int b, int fold, const float* __restrict__ scale) {
    __shared__ float tile[LT][LT + 1];  // [channel][pixel]
    const int He = H + 2 * b, We = W + 2 * b;
    const long long L = (long long)H * W;
    const int n = blockIdx.z, c0 = blockIdx.y * LT;
    const long long p0 = (long long)blockIdx.x * LT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float inv = scale ? 1.f / *scale : 1.f;
    const float* yn = y + (size_t)n * He * We * C;
    // read: lanes along channels; each wave gathers 16 pixels (+ their mirror images in the ring)
    for (int i = 0; i < 16; ++i) {
        const int px = w * 16 + i;
        const long long p = p0 + px;
        float a = 0.f;
        if (p < L && c0 + lane < C) {
            const int yy = (int)(p / W), xx = (int)(p - (long long)yy * W);
            int qy[2], qx[2], ny = 1, nx = 1;
            qy[0] = yy + b;
            qx[0] = xx + b;
            if (fold) {
                if (yy >= 1 && yy <= b) qy[ny++] = b - yy;
                else if (yy <= H - 2 && yy >= H - 1 - b) qy[ny++] = b + 2 * (H - 1) - yy;
                if (xx >= 1 && xx <= b) qx[nx++] = b - xx;
                else if (xx <= W - 2 && xx >= W - 1 - b) qx[nx++] = b + 2 * (W - 1) - xx;
            }
            for (int iy = 0; iy < ny; ++iy)
                for (int ix = 0; ix < nx; ++ix) a += yn[((size_t)qy[iy] * We + qx[ix]) * C + c0 + lane];
        }
        tile[lane][px] = a * inv;
    }
    __syncthreads();
    // write: lanes along pixels of one channel plane
    const long long p = p0 + lane;
    if (p < L) {
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int ch = i * 4 + w;
            if (c0 + ch < C) out[((size_t)n * C + c0 + ch) * L + p] = tile[ch][lane];
        }
    }
}

}  // namespace

extern "C" size_t dtk_conv_split_weight_halves(int Cin, int Cout) { return split_plane_halves(Cin, Cout); }

extern "C" int dtk_conv_split_pack(const float* w, int Cin, int Cout, int flip_transpose, void* Wh, void* Wl, void* stream) {
    DTK_REQUIRE(w && Wh && Wl, "dtk_conv_split_pack: null pointer");
    DTK_REQUIRE(Cin > 0 && Cin % SK == 0 && Cout > 0, "dtk_conv_split_pack: Cin=%d must be a positive multiple of %d", Cin, SK);
    const size_t nh = split_plane_halves(Cin, Cout);
    DTK_LAUNCH("train_conv_pack", pack_split_kernel, dim3(dtk_cdiv((long long)nh, 256)), dim3(256), 0, dtk_stream(stream), w, Cin,
               Cout, reinterpret_cast<half_t*>(Wh), reinterpret_cast<half_t*>(Wl), flip_transpose);
    return DTK_OK;
}

extern "C" int dtk_conv_split_input(const float* x, int N, int C, int H, int W, int border, const float* scale, void* hi,
                                    void* lo, void* stream) {
    DTK_REQUIRE(x && hi && lo, "dtk_conv_split_input: null pointer");
    DTK_REQUIRE(N > 0 && N <= 65535 && C > 0 && C % 8 == 0 && H > 0 && W > 0 && border >= 0, "dtk_conv_split_input: bad shape");
    const long long Le = (long long)(H + 2 * border) * (W + 2 * border);
    DTK_LAUNCH("train_conv_in", nchw_to_split_kernel, dim3(dtk_cdiv(Le, LT), dtk_cdiv(C, LT), N), dim3(256), 0,
               dtk_stream(stream), x, reinterpret_cast<half_t*>(hi), reinterpret_cast<half_t*>(lo), C, H, W, border, scale);
    return DTK_OK;
}

extern "C" int dtk_conv_split_run(const void* in_hi, const void* in_lo, const void* Wh, const void* Wl, float* out_nhwc, int N,
                                  int H, int W, int Cin, int Cout, int dilation, int zero_pad, int fp16_only, void* stream) {
    DTK_REQUIRE(in_hi && in_lo && Wh && Wl && out_nhwc, "dtk_conv_split_run: null pointer");
    DTK_REQUIRE(N > 0 && N <= 65535 && H > 4 * dilation && W > 4 * dilation, "dtk_conv_split_run: bad shape");
    DTK_REQUIRE(Cin % SK == 0 && Cout > 0, "dtk_conv_split_run: Cin=%d must be a multiple of %d", Cin, SK);
    DTK_REQUIRE(dilation == 1 || dilation == 2, "dtk_conv_split_run: dilation %d (1 or 2)", dilation);
    static const bool lds_ok = [] {
        bool ok = true;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5x5_split_kernel<1, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)SplitCfg<1>::LDS_BYTES) == hipSuccess;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5x5_split_kernel<2, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)SplitCfg<2>::LDS_BYTES) == hipSuccess;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5x5_split_kernel<1, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)SplitCfg<1>::LDS_BYTES) == hipSuccess;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5x5_split_kernel<2, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)SplitCfg<2>::LDS_BYTES) == hipSuccess;
        return ok;
    }();
    DTK_REQUIRE(lds_ok, "dtk_conv_split_run: cannot reserve LDS for the split-fp16 convolution");
    const int tiles_x = dtk_cdiv(W, STX), tiles_y = dtk_cdiv(H, STY);
    const int nct = (Cout + 63) / 64;
    const long long all_tiles = (long long)tiles_x * tiles_y * N;
    dim3 grid((unsigned)(((all_tiles + 7) / 8) * 8 * nct));   // XCD-aware 1-D grid (see the kernel)
    const int xcd_tiles = (int)all_tiles;
    const half_t* ih = reinterpret_cast<const half_t*>(in_hi);
    const half_t* il = reinterpret_cast<const half_t*>(in_lo);
    const half_t* wh = reinterpret_cast<const half_t*>(Wh);
    const half_t* wl = reinterpret_cast<const half_t*>(Wl);
#define DTK_CONV_RUN(NAME, DILV, SINGLEV)                                                                                      \
    DTK_LAUNCH(NAME, (conv5x5_split_kernel<DILV, false, SINGLEV>), grid, dim3(256), SplitCfg<DILV>::LDS_BYTES, dtk_stream(stream), \
               ih, il, wh, wl, (const float*)nullptr, (const float*)nullptr, (half_t*)nullptr, (half_t*)nullptr, out_nhwc, H, W, \
               Cin, Cout, 0, tiles_x, zero_pad, xcd_tiles)
    if (dilation == 1 && !fp16_only) { DTK_CONV_RUN("train_conv", 1, false); }
    else if (dilation == 1) { DTK_CONV_RUN("train_conv_h", 1, true); }
    else if (!fp16_only) { DTK_CONV_RUN("train_conv_d2", 2, false); }
    else { DTK_CONV_RUN("train_conv_d2_h", 2, true); }
#undef DTK_CONV_RUN
    return DTK_OK;
}

extern "C" int dtk_conv_split_output(const float* y_nhwc, int N, int C, int H, int W, int border, int reflect_fold,
                                     const float* scale, float* out_nchw, void* stream) {
    DTK_REQUIRE(y_nhwc && out_nchw, "dtk_conv_split_output: null pointer");
    DTK_REQUIRE(N > 0 && N <= 65535 && C > 0 && H > 0 && W > 0 && border >= 0, "dtk_conv_split_output: bad shape");
    DTK_REQUIRE(!reflect_fold || (H > 2 * border + 1 && W > 2 * border + 1), "dtk_conv_split_output: image smaller than the fold");
    const long long L = (long long)H * W;
    DTK_LAUNCH("train_conv_out", nhwc_to_nchw_kernel, dim3(dtk_cdiv(L, LT), dtk_cdiv(C, LT), N), dim3(256), 0,
               dtk_stream(stream), y_nhwc, out_nchw, C, H, W, border, reflect_fold, scale);
    return DTK_OK;
}
