// track_exact.hip -- DTK_TRACK_EXACT: the fp32 path of dtk_track.
//
//   corr_exact_kernel : rho[m][cell] = relu( <s_m, F[a_m][cell]> / max(|s_m| |F[a_m][cell]|, 1e-8) )
//                       (models/tracker.py:158-173), fp32 FMA chains over C, 64x64 LDS-tiled, staged through
//                       `workspace` in chunks that stay inside the 256 MiB Infinity Cache.
//   head_exact_kernel : TrackerHead.forward (models/networks/tracker_head.py:107-121) on one map per workgroup:
//                       first-max argmax, 3x3 conv(1->16) + ReLU + 3x3 conv(16->1) with zero padding, softmax over
//                       all cells, radius-35 disk around the argmax, zero-sum fallback, weighted mean.
//
// This path is the arithmetic ground truth on the device (fp32 everywhere) and the fallback of the fused MFMA path
// for the rare sources whose low-precision pass is inconclusive.
#include <limits.h>
#include "common.h"
#include "head_common.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16;
constexpr int HR = 6;  // output rows per conv block in head_exact_kernel

// ---- head parameter packing: W / sum(W) per (out,in) kernel, conv_norm.py:34-46 ---------------------------
__global__ void head_prepare_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                    const float* __restrict__ w2, const float* __restrict__ b2,
                                    float* __restrict__ head) {
    const int ch = threadIdx.x;  // 0..15
    if (ch >= DTK_HEAD_HIDDEN) return;
    float s1 = 0.f, s2 = 0.f;
    for (int t = 0; t < 9; ++t) {
        s1 += w1[ch * 9 + t];  // [16][1][3][3]
        s2 += w2[ch * 9 + t];  // [1][16][3][3]
    }
    auto fix = [](float s) {
        if (fabsf(s) < 1e-8f) s = (s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f)) * 1e-8f;
        return s;
    };
    s1 = fix(s1);
    s2 = fix(s2);
    for (int t = 0; t < 9; ++t) {
        head[ch * 9 + t] = w1[ch * 9 + t] / s1;
        head[144 + 16 + ch * 9 + t] = w2[ch * 9 + t] / s2;
    }
    head[144 + ch] = b1[ch];
    if (ch == 0) head[144 + 16 + 144] = b2[0];
}

// ---- |s_m| for the sources of one chunk; one wave per source ---------------------------------------------
__global__ __launch_bounds__(256) void row_norms_kernel(const float* __restrict__ emb,
                                                        const int32_t* __restrict__ src_row,
                                                        float* __restrict__ snorm, int m0, int count, int M,
                                                        const int32_t* __restrict__ dM, int C) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= count) return;
    const int lane = threadIdx.x & 63;
    const int m = m0 + i;
    float s = 0.f;
    if (m < dtk_active(M, dM)) {
        const float* p = emb + (size_t)(src_row ? src_row[m] : m) * C;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(p + c);
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        s = wave_sum(s);
    }
    if (lane == 0) snorm[i] = sqrtf(s);
}

// ---- correlation: 64 sources x 64 cells per workgroup, 4x4 per thread ------------------------------------
__global__ __launch_bounds__(256) void corr_exact_kernel(dtk_geom g, const float* __restrict__ feat,
                                                         const float* __restrict__ norms,
                                                         const float* __restrict__ emb,
                                                         const int32_t* __restrict__ src_row,
                                                         const int32_t* __restrict__ tgt,
                                                         const float* __restrict__ snorm, float* __restrict__ maps,
                                                         int m0, int count, int M, const int32_t* __restrict__ dM,
                                                         int HWs, int relu) {
    __shared__ __attribute__((aligned(16))) float As[TK][TM + 4];
    __shared__ __attribute__((aligned(16))) float Bs[TK][TN + 4];
    __shared__ int s_tgt[TM];
    __shared__ int s_row[TM];
    __shared__ int s_fr[2];
    const int HW = g.ph * g.pw;
    const int active = min(dtk_active(M, dM), m0 + count);
    const int tile_m0 = m0 + blockIdx.y * TM;
    if (tile_m0 >= active) return;
    const int cell0 = blockIdx.x * TN;
    const int tid = threadIdx.x;
    if (tid < TM) {
        const int m = tile_m0 + tid;
        const bool ok = m < active;
        int f = ok ? tgt[m] : -1;
        if (ok) f = min(max(f, 0), g.T - 1);
        s_tgt[tid] = f;
        s_row[tid] = ok ? (src_row ? src_row[m] : m) : (src_row ? src_row[tile_m0] : tile_m0);
        int lo = ok ? f : INT_MAX, hi = f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o, WAVE));
            hi = max(hi, __shfl_xor(hi, o, WAVE));
        }
        if (tid == 0) {
            s_fr[0] = lo;
            s_fr[1] = hi;
        }
    }
    __syncthreads();
    const int fmin = s_fr[0], fmax = s_fr[1];
    const int ty = tid >> 4, tx = tid & 15;
    const int lr = tid >> 2, lk = (tid & 3) * 4;  // loader: row / cell lr, k offset lk
    const float* arow = emb + (size_t)s_row[lr] * g.C;
    const int lcell = cell0 + lr;
    for (int f = fmin; f <= fmax; ++f) {
        bool mine = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) mine |= (s_tgt[ty * 4 + i] == f);
        if (!__syncthreads_or(mine)) continue;
        const float* brow = feat + ((size_t)f * HW + min(lcell, HW - 1)) * g.C;
        float acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        for (int k0 = 0; k0 < g.C; k0 += TK) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (k0 + lk < g.C) {
                a = *reinterpret_cast<const float4*>(arow + k0 + lk);
                if (lcell < HW) b = *reinterpret_cast<const float4*>(brow + k0 + lk);
            }
            As[lk + 0][lr] = a.x; As[lk + 1][lr] = a.y; As[lk + 2][lr] = a.z; As[lk + 3][lr] = a.w;
            Bs[lk + 0][lr] = b.x; Bs[lk + 1][lr] = b.y; Bs[lk + 2][lr] = b.z; Bs[lk + 3][lr] = b.w;
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < TK; ++kk) {
                const float4 av = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
                const float4 bv = *reinterpret_cast<const float4*>(&Bs[kk][tx * 4]);
                const float aa[4] = {av.x, av.y, av.z, av.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(aa[i], bb[j], acc[i][j]);
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = ty * 4 + i;
            if (s_tgt[r] != f) continue;
            const int ml = tile_m0 - m0 + r;
            const float sn = snorm[ml];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cell = cell0 + tx * 4 + j;
                if (cell < HW) {
                    const float den = fmaxf(sn * norms[(size_t)f * HW + cell], 1e-8f);
                    const float rho = acc[i][j] / den;
                    maps[(size_t)ml * HWs + cell] = relu ? fmaxf(rho, 0.f) : rho;
                }
            }
        }
    }
}

// ---- block reductions (256 threads = 4 waves); `red` is >= 8 floats of LDS ------------------------------
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

namespace {

__global__ __launch_bounds__(256) void head_exact_kernel(dtk_geom g, const float* __restrict__ head,
                                                         const float* __restrict__ maps, int HWs,
                                                         const int32_t* __restrict__ out_idx,
                                                         float* __restrict__ out_xy, int m0, int count, int M,
                                                         const int32_t* __restrict__ dM, int normalized,
                                                         float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HW = g.ph * g.pw, pw = g.pw, ph = g.ph;
    const int HWp = (HW + 3) & ~3;
    float* sx = smem;                 // relu'd cosine map
    float* sz = sx + HWp;             // refined logits
    float* sh = sz + HWp;             // hidden ring [16][HR+2][pw]
    float* red = sh + DTK_HEAD_HIDDEN * (HR + 2) * pw;  // 16 floats
    int* redi = reinterpret_cast<int*>(red + 8);
    const int i = blockIdx.x;
    const int m = m0 + i;
    if (i >= count || m >= dtk_active(M, dM)) return;
    const int tid = threadIdx.x;
    const float* map = maps + (size_t)i * HWs;
    for (int c = tid; c < HW; c += 256) sx[c] = map[c];
    __syncthreads();

    // first maximum (torch.argmax): larger value wins, ties -> lower flat index
    float best = -1.f;
    int bi = INT_MAX;
    for (int c = tid; c < HW; c += 256) {
        const float v = sx[c];
        if (v > best) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, WAVE);
        const int oi = __shfl_xor(bi, o, WAVE);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { red[tid >> 6] = best; redi[tid >> 6] = bi; }
    __syncthreads();
    best = red[0]; bi = redi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (red[w] > best || (red[w] == best && redi[w] < bi)) { best = red[w]; bi = redi[w]; }
    const int kstar = bi;
    __syncthreads();

    const float* w1 = head;              // [16][9]
    const float* b1 = head + 144;        // [16]
    const float* w2 = head + 160;        // [16][9]
    const float b2 = head[304];
    const int ring = HR + 2;
    for (int r0 = 0; r0 < ph; r0 += HR) {
        const int nout = min(HR, ph - r0);
        // hidden rows r0-1 .. r0+nout (zero outside the map: conv2's zero padding)
        for (int idx = tid; idx < (nout + 2) * pw; idx += 256) {
            const int hr = idx / pw, c = idx - hr * pw;
            const int row = r0 - 1 + hr;
            if (row < 0 || row >= ph) {
#pragma unroll
                for (int ch = 0; ch < DTK_HEAD_HIDDEN; ++ch) sh[(ch * ring + hr) * pw + c] = 0.f;
                continue;
            }
            float x9[9];
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int rr = row + dy, cc = c + dx;
                    x9[(dy + 1) * 3 + dx + 1] = (rr >= 0 && rr < ph && cc >= 0 && cc < pw) ? sx[rr * pw + cc] : 0.f;
                }
#pragma unroll
            for (int ch = 0; ch < DTK_HEAD_HIDDEN; ++ch) {
                float a = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) a = fmaf(w1[ch * 9 + t], x9[t], a);
                a += b1[ch];
                sh[(ch * ring + hr) * pw + c] = fmaxf(a, 0.f);
            }
        }
        __syncthreads();
        for (int idx = tid; idx < nout * pw; idx += 256) {
            const int ro = idx / pw, c = idx - ro * pw;
            float a = 0.f;
#pragma unroll
            for (int ch = 0; ch < DTK_HEAD_HIDDEN; ++ch) {
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int cc = c + dx;
                        const float hv = (cc >= 0 && cc < pw) ? sh[(ch * ring + ro + 1 + dy) * pw + cc] : 0.f;
                        a = fmaf(w2[ch * 9 + (dy + 1) * 3 + dx + 1], hv, a);
                    }
            }
            sz[(r0 + ro) * pw + c] = a + b2;
        }
        __syncthreads();
    }

    float zm = -INFINITY;
    for (int c = tid; c < HW; c += 256) zm = fmaxf(zm, sz[c]);
    zm = block_max(zm, red);
    float zs = 0.f;
    for (int c = tid; c < HW; c += 256) zs += expf(sz[c] - zm);
    zs = block_sum(zs, red);
    if (tid < 64) {
        auto zfun = [&](int r, int c) { return sz[r * pw + c]; };
        float sq = 0.f;
        dtk_disk_softargmax(g, kstar, zm, zs, zfun, normalized, out_xy + 2 * (size_t)(out_idx ? out_idx[m] : m), &sq);
        // what the backward of the training step needs again (dtk_head_backward): arg-max cell, softmax statistics, disk mass
        if (stats && tid == 0) {
            float* st = stats + 4 * (size_t)i;
            st[0] = __int_as_float(kstar);
            st[1] = zm;
            st[2] = zs;
            st[3] = sq;
        }
    }
}

}  // namespace

// first maximum of a raw cosine map (torch.argmax semantics: ties -> lowest flat index) and its value; one workgroup per map
__global__ __launch_bounds__(256) void argmax_exact_kernel(const float* __restrict__ maps, int HW, int HWs,
                                                           const int32_t* __restrict__ out_idx, int32_t* __restrict__ arg_cell,
                                                           float* __restrict__ arg_cos, int m0, int count) {
    __shared__ float red[4];
    __shared__ int redi[4];
    const int i = blockIdx.x;
    if (i >= count) return;
    const int tid = threadIdx.x;
    const float* map = maps + (size_t)i * HWs;
    float best = -INFINITY;
    int bi = INT_MAX;
    for (int c = tid; c < HW; c += 256) {
        const float v = map[c];
        if (v > best) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, WAVE);
        const int oi = __shfl_xor(bi, o, WAVE);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { red[tid >> 6] = best; redi[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        best = red[0]; bi = redi[0];
        for (int w = 1; w < 4; ++w)
            if (red[w] > best || (red[w] == best && redi[w] < bi)) { best = red[w]; bi = redi[w]; }
        const int oi = out_idx ? out_idx[m0 + i] : m0 + i;
        arg_cell[oi] = bi;
        arg_cos[oi] = best;
    }
}

extern "C" int dtk_head_prepare(const float* w1, const float* b1, const float* w2, const float* b2, float* head,
                                void* stream) {
    DTK_REQUIRE(w1 && b1 && w2 && b2 && head, "dtk_head_prepare: null pointer");
    DTK_LAUNCH("head_prepare", head_prepare_kernel, dim3(1), dim3(64), 0, dtk_stream(stream), w1, b1, w2, b2, head);
    return DTK_OK;
}

extern "C" int dtk_head_forward(const dtk_geom* g, const float* head, const float* maps, float* out_xy, int B,
                                int normalized, void* stream);

// ---- host driver of the exact path -----------------------------------------------------------------------
static inline int exact_hws(const dtk_geom* g) { return (g->ph * g->pw + 63) & ~63; }
static inline size_t exact_head_lds(const dtk_geom* g) {
    const int HWp = (g->ph * g->pw + 3) & ~3;
    return sizeof(float) * (size_t)(2 * HWp + DTK_HEAD_HIDDEN * (HR + 2) * g->pw + 16);
}
constexpr int EXACT_CHUNK = 4096;  // 4096 maps x 32 KB = 133 MB: stays in the 256 MiB Infinity Cache

size_t dtk_track_exact_workspace_bytes(const dtk_geom* g, int M) {
    const int chunk = M < EXACT_CHUNK ? (M < 64 ? 64 : M) : EXACT_CHUNK;
    return (size_t)chunk * (exact_hws(g) + 1) * sizeof(float);
}

int dtk_track_exact(const dtk_geom* g, const float* feat, const float* norms, const float* head, const float* emb,
                    const int32_t* src_row, const int32_t* tgt, const int32_t* out_idx, float* out_xy, int M,
                    const int32_t* dM, int normalized, void* workspace, size_t workspace_bytes, void* stream) {
    DTK_REQUIRE(g->C % TK == 0, "dtk_track(exact): C=%d must be a multiple of %d", g->C, TK);
    const int HWs = exact_hws(g);
    const size_t lds = exact_head_lds(g);
    DTK_REQUIRE(lds <= 160 * 1024, "dtk_track(exact): token grid %dx%d needs %zu B of LDS (> 160 KiB)", g->ph, g->pw, lds);
    long long chunk = (long long)(workspace_bytes / ((size_t)(HWs + 1) * sizeof(float)));
    if (chunk > M) chunk = M;
    if (chunk > 65535LL * TM) chunk = 65535LL * TM;
    if (chunk < 1) {
        dtk_set_error("dtk_track(exact): workspace of %zu B holds no map (need %zu B per source)", workspace_bytes,
                      (size_t)(HWs + 1) * sizeof(float));
        return DTK_E_WORKSPACE;
    }
    hipStream_t st = dtk_stream(stream);
    DTK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(head_exact_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    float* maps = reinterpret_cast<float*>(workspace);
    float* snorm = maps + (size_t)chunk * HWs;
    const int HW = g->ph * g->pw;
    for (long long m0 = 0; m0 < M; m0 += chunk) {
        const int cnt = (int)((M - m0) < chunk ? (M - m0) : chunk);
        DTK_LAUNCH("row_norms", row_norms_kernel, dim3(dtk_cdiv(cnt, 4)), dim3(256), 0, st, emb, src_row, snorm, (int)m0,
                           cnt, M, dM, g->C);
        DTK_LAUNCH("corr_exact", corr_exact_kernel, dim3(dtk_cdiv(HW, TN), dtk_cdiv(cnt, TM)), dim3(256), 0, st, *g, feat,
                           norms, emb, src_row, tgt, snorm, maps, (int)m0, cnt, M, dM, HWs, 1);
        DTK_LAUNCH("head_exact", head_exact_kernel, dim3(cnt), dim3(256), lds, st, *g, head, maps, HWs, out_idx, out_xy,
                           (int)m0, cnt, M, dM, normalized, (float*)nullptr);
    }
    return DTK_OK;
}

extern "C" int dtk_head_forward(const dtk_geom* g, const float* head, const float* maps, float* out_xy, int B,
                                int normalized, void* stream) {
    DTK_REQUIRE(g && head && maps && out_xy && B >= 0, "dtk_head_forward: null pointer");
    DTK_REQUIRE(g->ph > 0 && g->pw > 0 && g->stride > 0 && g->patch > 0, "dtk_head_forward: bad geometry");
    if (B == 0) return DTK_OK;
    const size_t lds = exact_head_lds(g);
    DTK_REQUIRE(lds <= 160 * 1024, "dtk_head_forward: token grid %dx%d needs %zu B of LDS (> 160 KiB)", g->ph, g->pw, lds);
    DTK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(head_exact_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DTK_LAUNCH("head_exact", head_exact_kernel, dim3(B), dim3(256), lds, dtk_stream(stream), *g, head, maps, g->ph * g->pw,
                       (const int32_t*)nullptr, out_xy, 0, B, B, (const int32_t*)nullptr, normalized, (float*)nullptr);
    return DTK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// TrackerHead.forward / backward of the TRAINING step (tracker_head.py:68-121 under autograd, dino_tracker.py:392-448).
// Forward = head_exact_kernel, which also leaves per-map statistics (arg-max cell, softmax maximum and partition sum, disk
// mass).  The backward is LOCAL: with x^ = sum_D q X / sum_D q over the disk D around the arg-max and q = softmax(z) on D,
//     dz_k = p_k (dq_k - sum_{j in D} p_j dq_j),   dq_k = [gx (X_k - x^) + gy (Y_k - y^)] / sum_D q   (k in D, else 0)
// and the subtracted mean vanishes identically when no zero-mass fallback fired (sum_D p_j (X_j - x^) = 0): logits outside the
// disk do not move the output.  So dz lives on the disk (<= 81 cells), the hidden gradient on its 13 x 13 neighbourhood, the
// input gradient on 15 x 15 -- the windows of the inference kernel refine_head.  One wave per map, windows in LDS:
// recompute hidden and logits of the window, then the two transposed 3 x 3 convolutions and the four parameter gradients
// (per-map partials, summed by the caller: deterministic).  A map whose fallback fired (disk mass s < 1e-8) has q = p + 1/|D|
// on the disk and the mean no longer vanishes: its WHOLE gradient is of size p_k |dq| <= 1e-8 |dq| (eight orders below an
// ordinary map's), split into the disk part, which this kernel returns, and -p_j cbar on every other cell of the map
// (sum over j <= 1e-8 max |dq|), which it drops.  Callers that want that remainder too read `stats` and send such batches
// through the traced path (train_ops.HEAD_FALLBACK_EXACT).
// ---------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int HB_RD = 5, HB_WX = 2 * HB_RD + 5, HB_WH = 2 * HB_RD + 3, HB_WZ = 2 * HB_RD + 1;  // 15, 13, 11

__global__ __launch_bounds__(256) void head_backward_kernel(dtk_geom g, const float* __restrict__ head,
                                                            const float* __restrict__ maps, const float* __restrict__ stats,
                                                            const float* __restrict__ gout, float* __restrict__ dmaps,
                                                            float* __restrict__ dhead, int B, int normalized) {
    constexpr int NH = HB_WH * HB_WH, NZ = HB_WZ * HB_WZ, XP = HB_WX + 1;
    __shared__ float s_x[4][HB_WX * XP];
    __shared__ float s_h[4][DTK_HEAD_HIDDEN * NH];   // hidden activations of the window, later their gradients
    __shared__ unsigned char s_m[4][DTK_HEAD_HIDDEN * NH];  // relu' of the hidden pre-activation (0 also outside the map)
    __shared__ float s_z[4][NZ], s_dz[4][NZ];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    if (b >= B) return;  // wave-uniform; only wave-level barriers below
    const int ph = g.ph, pw = g.pw, HW = ph * pw;
    const float* st = stats + 4 * (size_t)b;
    const int kstar = __float_as_int(st[0]);
    const float zmax = st[1], Z = st[2];
    const int kr = kstar / pw, kc = kstar % pw;
    const float* map = maps + (size_t)b * HW;
    const float* w1 = head;
    const float* b1 = head + 144;
    const float* w2 = head + 160;
    const float b2 = head[304];
    float* sx = s_x[w];
    float* sh = s_h[w];
    unsigned char* sm = s_m[w];
    float* sz = s_z[w];
    float* sdz = s_dz[w];
    auto wsync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // x window: rows kr - 7 .. kr + 7, columns kc - 7 .. kc + 7 (zero outside the map: conv1's padding)
    for (int i = lane; i < HB_WX * HB_WX; i += WAVE) {
        const int jr = i / HB_WX, jc = i - jr * HB_WX;
        const int r = kr - (HB_RD + 2) + jr, c = kc - (HB_RD + 2) + jc;
        sx[jr * XP + jc] = (r >= 0 && r < ph && c >= 0 && c < pw) ? map[r * pw + c] : 0.f;
    }
    wsync();
    // hidden window: cell (hy, hx) = map cell (kr - 6 + hy, kc - 6 + hx); outside the map it is conv2's zero padding
    for (int i = lane; i < NH; i += WAVE) {
        const int hy = i / HB_WH, hx = i - hy * HB_WH;
        const int r = kr - (HB_RD + 1) + hy, c = kc - (HB_RD + 1) + hx;
        const bool in = r >= 0 && r < ph && c >= 0 && c < pw;
        float x9[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) x9[dy * 3 + dx] = sx[(hy + dy) * XP + hx + dx];
#pragma unroll
        for (int ch = 0; ch < DTK_HEAD_HIDDEN; ++ch) {
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) a = fmaf(w1[ch * 9 + t], x9[t], a);
            a += b1[ch];
            const bool pos = in && a > 0.f;
            sh[ch * NH + i] = pos ? a : 0.f;
            sm[ch * NH + i] = pos ? 1 : 0;
        }
    }
    wsync();
    // logits of the 11 x 11 window: cell (zy, zx) = map cell (kr - 5 + zy, kc - 5 + zx) = hidden cell (zy + 1, zx + 1)
    for (int i = lane; i < NZ; i += WAVE) {
        const int zy = i / HB_WZ, zx = i - zy * HB_WZ;
        float a = 0.f;
#pragma unroll
        for (int ch = 0; ch < DTK_HEAD_HIDDEN; ++ch)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) a = fmaf(w2[ch * 9 + dy * 3 + dx], sh[ch * NH + (zy + dy) * HB_WH + zx + dx], a);
        sz[i] = a + b2;
    }
    wsync();
    // disk: masses, centre, dq, dz
    const float half = (float)(g.patch / 2);
    float p_[2], cx[2], cy[2];
    bool ok[2];
    float sq = 0.f, sqx = 0.f, sqy = 0.f;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int i = lane + WAVE * sl;
        const int zy = i / HB_WZ, zx = i - zy * HB_WZ;
        const int r = kr - HB_RD + zy, c = kc - HB_RD + zx;
        const float dx = (float)((c - kc) * g.stride), dy = (float)((r - kr) * g.stride);
        ok[sl] = i < NZ && r >= 0 && r < ph && c >= 0 && c < pw && sqrtf(dx * dx + dy * dy) <= g.radius;
        p_[sl] = ok[sl] ? expf(sz[min(i, NZ - 1)] - zmax) / Z : 0.f;
        cx[sl] = (float)(c * g.stride) + half;
        cy[sl] = (float)(r * g.stride) + half;
        sq += p_[sl]; sqx += p_[sl] * cx[sl]; sqy += p_[sl] * cy[sl];
    }
    sq = wave_sum(sq); sqx = wave_sum(sqx); sqy = wave_sum(sqy);
    // zero-mass fallback (tracker_head.py:86-94; the forward's own statistic decides): q = (p + 1 / |D|) on the disk
    const bool fallback = st[3] < 1e-8f;
    if (fallback) {
        float cnt = wave_sum((ok[0] ? 1.f : 0.f) + (ok[1] ? 1.f : 0.f));
        const float u = 1.f / cnt;
        float s1 = 0.f, sx1 = 0.f, sy1 = 0.f;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
            if (ok[sl]) { s1 += u; sx1 += u * cx[sl]; sy1 += u * cy[sl]; }
        sq += wave_sum(s1); sqx += wave_sum(sx1); sqy += wave_sum(sy1);
    }
    const float xh = sqx / sq, yh = sqy / sq;
    float gx = gout[2 * (size_t)b], gy = gout[2 * (size_t)b + 1];
    if (normalized) { gx *= 2.f / (float)(g.video_w - 1); gy *= 2.f / (float)(g.video_h - 1); }
    float dq[2], cbar = 0.f;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        dq[sl] = ok[sl] ? (gx * (cx[sl] - xh) + gy * (cy[sl] - yh)) / sq : 0.f;
        cbar += p_[sl] * dq[sl];
    }
    cbar = wave_sum(cbar);  // zero up to rounding unless the fallback fired (then |cbar| <= 1e-8 max |dq|)
    float db2 = 0.f;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int i = lane + WAVE * sl;
        const float dz = ok[sl] ? p_[sl] * (dq[sl] - cbar) : 0.f;
        if (i < NZ) sdz[i] = dz;
        db2 += dz;
    }
    db2 = wave_sum(db2);
    wsync();
    float* dh = dhead + 305 * (size_t)b;
    // dW2[ch][t] = sum_k dz_k hid[ch][k + t]
    for (int o = lane; o < DTK_HEAD_HIDDEN * 9; o += WAVE) {
        const int ch = o / 9, t = o - ch * 9, dy = t / 3, dx = t - dy * 3;
        float a = 0.f;
        for (int i = 0; i < NZ; ++i) {
            const int zy = i / HB_WZ, zx = i - zy * HB_WZ;
            a = fmaf(sdz[i], sh[ch * NH + (zy + dy) * HB_WH + zx + dx], a);
        }
        dh[160 + o] = a;
    }
    if (lane == 0) dh[304] = db2;
    wsync();
    // dhid_pre[ch][m] = relu'(.) sum_t w2[ch][t] dz[m - t]   (hidden cell m = (hy, hx); dz cell = (hy - dy, hx - dx) in z coordinates)
    for (int i = lane; i < NH; i += WAVE) {
        const int hy = i / HB_WH, hx = i - hy * HB_WH;
        float dzt[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int zy = hy - dy, zx = hx - dx;  // z cell whose tap (dy, dx) reads this hidden cell
                dzt[dy * 3 + dx] = (zy >= 0 && zy < HB_WZ && zx >= 0 && zx < HB_WZ) ? sdz[zy * HB_WZ + zx] : 0.f;
            }
#pragma unroll
        for (int ch = 0; ch < DTK_HEAD_HIDDEN; ++ch) {
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) a = fmaf(w2[ch * 9 + t], dzt[t], a);
            sh[ch * NH + i] = sm[ch * NH + i] ? a : 0.f;
        }
    }
    wsync();
    // db1[ch], dW1[ch][t] = sum_m dhid_pre[ch][m] x[m + t]
    for (int o = lane; o < DTK_HEAD_HIDDEN * 10; o += WAVE) {
        const int ch = o / 10, t = o - ch * 10;
        float a = 0.f;
        if (t == 9) {
            for (int i = 0; i < NH; ++i) a += sh[ch * NH + i];
            dh[144 + ch] = a;
        } else {
            const int dy = t / 3, dx = t - dy * 3;
            for (int i = 0; i < NH; ++i) {
                const int hy = i / HB_WH, hx = i - hy * HB_WH;
                a = fmaf(sh[ch * NH + i], sx[(hy + dy) * XP + hx + dx], a);
            }
            dh[ch * 9 + t] = a;
        }
    }
    // dx[j] = sum_ch sum_t w1[ch][t] dhid_pre[ch][j - t]   (x cell j = (jr, jc); hidden cell = (jr - dy, jc - dx))
    float* dm = dmaps + (size_t)b * HW;
    for (int i = lane; i < HB_WX * HB_WX; i += WAVE) {
        const int jr = i / HB_WX, jc = i - jr * HB_WX;
        const int r = kr - (HB_RD + 2) + jr, c = kc - (HB_RD + 2) + jc;
        if (r < 0 || r >= ph || c < 0 || c >= pw) continue;
        float a = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int hy = jr - dy, hx = jc - dx;
                if (hy < 0 || hy >= HB_WH || hx < 0 || hx >= HB_WH) continue;
#pragma unroll
                for (int ch = 0; ch < DTK_HEAD_HIDDEN; ++ch) a = fmaf(w1[ch * 9 + dy * 3 + dx], sh[ch * NH + hy * HB_WH + hx], a);
            }
        dm[r * pw + c] = a;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the cosine maps of the TRAINING step (models/tracker.py:158-173 under autograd) behind the local head backward:
// the map gradient of a source is non-zero only on the 15 x 15 window around its arg-max (head_backward_kernel), so
//     rho = <s, f> / max(|s| |f|, 1e-8),   relu'(rho) d_rho  ->  ds += g (f / den - rho s / |s|^2),  df = g (s / den - rho f / |f|^2)
// (ds += g f / 1e-8, df = g s / 1e-8 where the clamp is active) is evaluated for those <= 225 cells only -- autograd's form is
// two dense products over every cell of every frame of the batch with a gradient that is 99.7 % zeros.  One workgroup per
// source; df goes to the token-major gradient volume with atomic adds (several sources share cells).
// ---------------------------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void corr_window_backward_kernel(dtk_geom g, const float* __restrict__ feat,
                                                                   const float* __restrict__ norms, const float* __restrict__ emb,
                                                                   const int32_t* __restrict__ tgt, const float* __restrict__ maps,
                                                                   const float* __restrict__ dmaps, const float* __restrict__ stats,
                                                                   float* __restrict__ demb, float* __restrict__ dfeat, int B) {
    // one workgroup per source: its four waves take the window's rows round-robin (a single wave walking all 225 cells is a
    // chain of dependent global loads: 0.41 ms per call), their partial ds are summed through LDS
    constexpr int MAXJ = 16;  // C <= 1024
    __shared__ float s_ds[4][MAXJ * WAVE];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const int C = g.C, pw = g.pw, ph = g.ph, HW = ph * pw;
    const int nj = (C + WAVE - 1) / WAVE;
    const int f = min(max(tgt[b], 0), g.T - 1);
    const int kstar = __float_as_int(stats[4 * (size_t)b]);
    const int kr = kstar / pw, kc = kstar % pw;
    float s[MAXJ], ds[MAXJ];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        const int c = j * WAVE + lane;
        s[j] = (j < nj && c < C) ? emb[(size_t)b * C + c] : 0.f;
        ds[j] = 0.f;
        ss += s[j] * s[j];
    }
    ss = wave_sum(ss);
    const float sn = sqrtf(ss);
    const float inv_ss = ss > 0.f ? 1.f / ss : 0.f;
    const float* fbase = feat + (size_t)f * HW * C;
    float* dbase = dfeat + (size_t)f * HW * C;
    const float* mp = maps + (size_t)b * HW;
    const float* dm = dmaps + (size_t)b * HW;
    const int r0 = max(kr - (HB_RD + 2), 0), r1 = min(kr + (HB_RD + 2), ph - 1);
    const int c0 = max(kc - (HB_RD + 2), 0), c1 = min(kc + (HB_RD + 2), pw - 1);
    for (int r = r0 + w; r <= r1; r += 4)
        for (int c = c0; c <= c1; ++c) {
            const int cell = r * pw + c;
            const float gk = dm[cell], rho = mp[cell];
            if (gk == 0.f || !(rho > 0.f)) continue;  // wave-uniform
            const float fn = norms[(size_t)f * HW + cell];
            const float den = sn * fn;
            const bool clamped = !(den > 1e-8f);
            const float inv_den = clamped ? 1e8f : 1.f / den;
            const float a_s = clamped ? 0.f : rho * inv_ss;
            const float a_f = clamped ? 0.f : rho / (fn * fn);
            const float* fp = fbase + (size_t)cell * C;
            float* dp = dbase + (size_t)cell * C;
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                const int ch = j * WAVE + lane;
                if (j < nj && ch < C) {
                    const float fv = fp[ch];
                    ds[j] += gk * (fv * inv_den - a_s * s[j]);
                    atomicAdd(dp + ch, gk * (s[j] * inv_den - a_f * fv));
                }
            }
        }
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) s_ds[w][j * WAVE + lane] = ds[j];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) demb[(size_t)b * C + c] = (s_ds[0][c] + s_ds[1][c]) + (s_ds[2][c] + s_ds[3][c]);
}

}  // namespace

extern "C" int dtk_corr_window_backward(const dtk_geom* g, const float* feat, const float* norms, const float* emb,
                                        const int32_t* tgt, const float* maps, const float* dmaps, const float* stats,
                                        float* demb, float* dfeat, int B, void* stream) {
    DTK_REQUIRE(g && feat && norms && emb && tgt && maps && dmaps && stats && demb && dfeat && B >= 0,
                "dtk_corr_window_backward: null pointer");
    DTK_REQUIRE(g->C > 0 && g->C <= 1024, "dtk_corr_window_backward: C=%d outside 1..1024", g->C);
    DTK_REQUIRE(g->radius / (float)g->stride <= (float)HB_RD, "dtk_corr_window_backward: radius / stride > %d", HB_RD);
    if (B == 0) return DTK_OK;
    DTK_LAUNCH("train_corr_bwd", corr_window_backward_kernel, dim3(B), dim3(256), 0, dtk_stream(stream), *g, feat,
               norms, emb, tgt, maps, dmaps, stats, demb, dfeat, B);
    return DTK_OK;
}

extern "C" int dtk_head_forward_train(const dtk_geom* g, const float* head, const float* maps, float* out_xy, float* stats, int B,
                                      int normalized, void* stream) {
    DTK_REQUIRE(g && head && maps && out_xy && stats && B >= 0, "dtk_head_forward_train: null pointer");
    DTK_REQUIRE(g->ph > 0 && g->pw > 0 && g->stride > 0 && g->patch > 0, "dtk_head_forward_train: bad geometry");
    if (B == 0) return DTK_OK;
    const size_t lds = exact_head_lds(g);
    DTK_REQUIRE(lds <= 160 * 1024, "dtk_head_forward_train: token grid %dx%d needs %zu B of LDS (> 160 KiB)", g->ph, g->pw, lds);
    DTK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(head_exact_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DTK_LAUNCH("train_head_fwd", head_exact_kernel, dim3(B), dim3(256), lds, dtk_stream(stream), *g, head, maps, g->ph * g->pw,
               (const int32_t*)nullptr, out_xy, 0, B, B, (const int32_t*)nullptr, normalized, stats);
    return DTK_OK;
}

extern "C" int dtk_head_backward(const dtk_geom* g, const float* head, const float* maps, const float* stats, const float* grad_out,
                                 float* dmaps, float* dhead_partial, int B, int normalized, void* stream) {
    DTK_REQUIRE(g && head && maps && stats && grad_out && dmaps && dhead_partial && B >= 0, "dtk_head_backward: null pointer");
    DTK_REQUIRE(g->stride > 0 && g->radius / (float)g->stride <= (float)HB_RD,
                "dtk_head_backward: disk radius %g px exceeds %d cells of stride %d", (double)g->radius, HB_RD, g->stride);
    if (B == 0) return DTK_OK;
    DTK_LAUNCH("train_head_bwd", head_backward_kernel, dim3(dtk_cdiv(B, 4)), dim3(256), 0, dtk_stream(stream), *g, head, maps,
               stats, grad_out, dmaps, dhead_partial, B, normalized);
    return DTK_OK;
}

// Tracker.get_corr_maps_for_frame_set (models/tracker.py:158-169) on its own: the cosine maps of M sources, fp32,
// maps[m][cell] (cell = row * pw + col), WITHOUT the ReLU unless `relu` (the reference applies it afterwards, :173).
extern "C" int dtk_corr_maps(const dtk_geom* g, const float* feat, const float* norms, const float* emb,
                             const int32_t* src_row, const int32_t* tgt, float* maps, float* snorm_scratch, int M,
                             int relu, void* stream) {
    DTK_REQUIRE(g && feat && norms && emb && tgt && maps && snorm_scratch, "dtk_corr_maps: null pointer");
    DTK_REQUIRE(g->T > 0 && g->ph > 0 && g->pw > 0 && g->C > 0 && g->C % TK == 0, "dtk_corr_maps: bad geometry (C %% %d)", TK);
    DTK_REQUIRE(M >= 0, "dtk_corr_maps: negative M");
    const int HW = g->ph * g->pw;
    hipStream_t st = dtk_stream(stream);
    const long long step = 65535LL * TM;
    for (long long m0 = 0; m0 < M; m0 += step) {
        const int cnt = (int)((M - m0) < step ? (M - m0) : step);
        DTK_LAUNCH("row_norms", row_norms_kernel, dim3(dtk_cdiv(cnt, 4)), dim3(256), 0, st, emb, src_row, snorm_scratch + m0,
                   (int)m0, cnt, M, (const int32_t*)nullptr, g->C);
        DTK_LAUNCH("corr_exact", corr_exact_kernel, dim3(dtk_cdiv(HW, TN), dtk_cdiv(cnt, TM)), dim3(256), 0, st, *g, feat,
                   norms, emb, src_row, tgt, snorm_scratch + m0, maps + (size_t)m0 * HW, (int)m0, cnt, M,
                   (const int32_t*)nullptr, HW, relu);
    }
    return DTK_OK;
}

// exact fp32 arg-max of the RAW cosine maps (no ReLU: torch.argmax of the affinity row, extract_dino_best_buddies.py:38-39)
int dtk_argmax_exact(const dtk_geom* g, const float* feat, const float* norms, const float* emb, const int32_t* src_row,
                     const int32_t* tgt, const int32_t* out_idx, int32_t* arg_cell, float* arg_cos, int M, void* workspace,
                     size_t workspace_bytes, void* stream) {
    DTK_REQUIRE(g->C % TK == 0, "dtk_argmax_cells(exact): C=%d must be a multiple of %d", g->C, TK);
    const int HWs = exact_hws(g), HW = g->ph * g->pw;
    long long chunk = (long long)(workspace_bytes / ((size_t)(HWs + 1) * sizeof(float)));
    if (chunk > M) chunk = M;
    if (chunk > 65535LL * TM) chunk = 65535LL * TM;
    if (chunk < 1) {
        dtk_set_error("dtk_argmax_cells(exact): workspace of %zu B holds no map", workspace_bytes);
        return DTK_E_WORKSPACE;
    }
    hipStream_t st = dtk_stream(stream);
    float* maps = reinterpret_cast<float*>(workspace);
    float* snorm = maps + (size_t)chunk * HWs;
    for (long long m0 = 0; m0 < M; m0 += chunk) {
        const int cnt = (int)((M - m0) < chunk ? (M - m0) : chunk);
        DTK_LAUNCH("row_norms", row_norms_kernel, dim3(dtk_cdiv(cnt, 4)), dim3(256), 0, st, emb, src_row, snorm, (int)m0, cnt, M,
                   (const int32_t*)nullptr, g->C);
        DTK_LAUNCH("corr_exact", corr_exact_kernel, dim3(dtk_cdiv(HW, TN), dtk_cdiv(cnt, TM)), dim3(256), 0, st, *g, feat, norms,
                   emb, src_row, tgt, snorm, maps, (int)m0, cnt, M, (const int32_t*)nullptr, HWs, 0);
        DTK_LAUNCH("argmax_exact", argmax_exact_kernel, dim3(cnt), dim3(256), 0, st, maps, HW, HWs, out_idx, arg_cell, arg_cos,
                   (int)m0, cnt);
    }
    return DTK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// DINO best-buddy ambiguity ratio (second half): preprocessing_dino_bb/compute_dino_bb_nms.py:12-66.
// Per source (a best-buddy cell of frame sf) the reference takes the affinity row against frame tf, its top-k (400)
// entries, boxes of +-box_size px around their cell centres, torchvision batched_nms (greedy, descending score, a box is
// dropped when its IoU with an already KEPT box exceeds the threshold), zeroes the dropped affinities and returns the two
// largest entries of that masked list and r = second / first.
// Only the first two kept boxes matter, and greedy NMS decides them without the full sweep: the arg-max is always kept,
// and the next candidate in descending order that does not overlap IT is the second kept one (everything in between
// overlaps the arg-max, the only kept box so far).  So per source:
//   peak   = first maximum of the row (cell k*, value a)
//   K      = the topk-th largest value of the row (exact: bit-wise bisection on an order-preserving integer key)
//   b      = max { v[k] : v[k] >= K, k != k*, IoU(box_k, box_k*) <= thresh }      (second kept box, if any)
//   nsupp  = # { k : v[k] >= K, k != k*, IoU(box_k, box_k*) > thresh }             (zeroed entries)
//   top-2 of the multiset { a, b, 0 x min(nsupp, 2) }  ->  peak_affs[2], r = second / first.
// IoU in fp32 exactly as torchvision's nms kernel computes it (boxes = centre -+ box_size, centres = patch/2 + stride *
// cell: all integers, so also equal to the coordinate-offset form batched_nms uses for few boxes).
// One workgroup per source over its fp32 map (corr_exact_kernel, no ReLU), rows in registers.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int NMS_VPT = 40;  // values per thread: HW <= 256 * 40 = 10240 cells

__device__ __forceinline__ unsigned order_key(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(256) void bb_nms_kernel(dtk_geom g, const float* __restrict__ maps, int HWs, float box_half,
                                                     float iou_thresh, int topk, float* __restrict__ peak_affs,
                                                     float* __restrict__ r_out, int m0, int count) {
    __shared__ float red[4];
    __shared__ int redi[4];
    __shared__ int s_cnt;
    const int i = blockIdx.x;
    if (i >= count) return;
    const int tid = threadIdx.x, HW = g.ph * g.pw;
    const float* map = maps + (size_t)i * HWs;
    float v[NMS_VPT];
    float best = -INFINITY;
    int bi = INT_MAX;
#pragma unroll
    for (int j = 0; j < NMS_VPT; ++j) {
        const int c = tid + 256 * j;
        v[j] = c < HW ? map[c] : -INFINITY;
        if (c < HW && v[j] > best) { best = v[j]; bi = c; }
    }
    // ---- first maximum (torch semantics: lowest index among equals) ----
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, WAVE);
        const int oi = __shfl_xor(bi, o, WAVE);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { red[tid >> 6] = best; redi[tid >> 6] = bi; }
    __syncthreads();
    best = red[0]; bi = redi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (red[w] > best || (red[w] == best && redi[w] < bi)) { best = red[w]; bi = redi[w]; }
    // ---- K = key of the topk-th largest value: build it bit by bit, most significant first ----
    unsigned K = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = K | (1u << bit);
        int c = 0;
#pragma unroll
        for (int j = 0; j < NMS_VPT; ++j) c += (tid + 256 * j < HW && order_key(v[j]) >= cand) ? 1 : 0;
        c = wave_sum_i(c);
        __syncthreads();
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        if ((tid & 63) == 0) atomicAdd(&s_cnt, c);
        __syncthreads();
        if (s_cnt >= topk) K = cand;
    }
    // ---- second kept box / number of suppressed ones among the top-k ----
    const float half = (float)(g.patch / 2), st = (float)g.stride;
    const float pcx = half + st * (float)(bi % g.pw), pcy = half + st * (float)(bi / g.pw);
    const float px1 = pcx - box_half, px2 = pcx + box_half, py1 = pcy - box_half, py2 = pcy + box_half;
    const float parea = (px2 - px1) * (py2 - py1);
    float b = -INFINITY;
    int nsupp = 0;
#pragma unroll
    for (int j = 0; j < NMS_VPT; ++j) {
        const int c = tid + 256 * j;
        if (c < HW && c != bi && order_key(v[j]) >= K) {
            const float cx = half + st * (float)(c % g.pw), cy = half + st * (float)(c / g.pw);
            const float x1 = cx - box_half, x2 = cx + box_half, y1 = cy - box_half, y2 = cy + box_half;
            const float iw = fmaxf(fminf(x2, px2) - fmaxf(x1, px1), 0.f), ih = fmaxf(fminf(y2, py2) - fmaxf(y1, py1), 0.f);
            const float inter = iw * ih;
            const float iou = inter / (parea + (x2 - x1) * (y2 - y1) - inter);
            if (iou > iou_thresh) ++nsupp;
            else b = fmaxf(b, v[j]);
        }
    }
    b = wave_max(b);
    nsupp = wave_sum_i(nsupp);
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = b; redi[tid >> 6] = nsupp; }
    __syncthreads();
    if (tid == 0) {
        b = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        nsupp = redi[0] + redi[1] + redi[2] + redi[3];
        // two largest of { best, b, 0 (nsupp >= 1), 0 (nsupp >= 2) }
        float c4[4] = {best, b, nsupp >= 1 ? 0.f : -INFINITY, nsupp >= 2 ? 0.f : -INFINITY};
        float t1 = -INFINITY, t2 = -INFINITY;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (c4[q] > t1) { t2 = t1; t1 = c4[q]; }
            else if (c4[q] > t2) t2 = c4[q];
        }
        peak_affs[2 * (size_t)(m0 + i)] = t1;
        peak_affs[2 * (size_t)(m0 + i) + 1] = t2;
        r_out[m0 + i] = t2 / t1;
    }
}

}  // namespace

extern "C" size_t dtk_bb_nms_workspace_bytes(const dtk_geom* g, int M) { return dtk_track_exact_workspace_bytes(g, M); }

extern "C" int dtk_bb_nms(const dtk_geom* g, const float* feat, const float* norms, const float* emb, const int32_t* src_row,
                          const int32_t* tgt, float box_size, float iou_thresh, int topk, float* peak_affs, float* r, int M,
                          void* workspace, size_t workspace_bytes, void* stream) {
    DTK_REQUIRE(g && feat && norms && emb && tgt && peak_affs && r && workspace, "dtk_bb_nms: null pointer");
    DTK_REQUIRE(M >= 0, "dtk_bb_nms: negative M");
    if (M == 0) return DTK_OK;
    DTK_REQUIRE(g->C % TK == 0, "dtk_bb_nms: C=%d must be a multiple of %d", g->C, TK);
    const int HWs = exact_hws(g), HW = g->ph * g->pw;
    DTK_REQUIRE(HW <= 256 * NMS_VPT, "dtk_bb_nms: %d cells per frame (max %d)", HW, 256 * NMS_VPT);
    DTK_REQUIRE(topk >= 2 && topk <= HW, "dtk_bb_nms: topk=%d must be in [2, %d] (torch.topk)", topk, HW);
    long long chunk = (long long)(workspace_bytes / ((size_t)(HWs + 1) * sizeof(float)));
    if (chunk > M) chunk = M;
    if (chunk > 65535LL * TM) chunk = 65535LL * TM;
    if (chunk < 1) {
        dtk_set_error("dtk_bb_nms: workspace of %zu B holds no map", workspace_bytes);
        return DTK_E_WORKSPACE;
    }
    hipStream_t st = dtk_stream(stream);
    float* maps = reinterpret_cast<float*>(workspace);
    float* snorm = maps + (size_t)chunk * HWs;
    for (long long m0 = 0; m0 < M; m0 += chunk) {
        const int cnt = (int)((M - m0) < chunk ? (M - m0) : chunk);
        DTK_LAUNCH("row_norms", row_norms_kernel, dim3(dtk_cdiv(cnt, 4)), dim3(256), 0, st, emb, src_row, snorm, (int)m0, cnt, M,
                   (const int32_t*)nullptr, g->C);
        DTK_LAUNCH("corr_exact", corr_exact_kernel, dim3(dtk_cdiv(HW, TN), dtk_cdiv(cnt, TM)), dim3(256), 0, st, *g, feat, norms,
                   emb, src_row, tgt, snorm, maps, (int)m0, cnt, M, (const int32_t*)nullptr, HWs, 0);
        DTK_LAUNCH("bb_nms", bb_nms_kernel, dim3(cnt), dim3(256), 0, st, *g, maps, HWs, box_size, iou_thresh, topk, peak_affs, r,
                   (int)m0, cnt);
    }
    return DTK_OK;
}
