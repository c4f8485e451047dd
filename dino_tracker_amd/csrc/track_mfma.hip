// track_mfma.hip -- DTK_TRACK_MFMA: the fast path of dtk_track.
//
// Per ROUND of up to 4 M sources (sorted by target frame; MFMA_SUPER), the pipeline since round 3-4:
//   src16_kernel        the distinct source rows -> fp16 unit vectors s^ = 32 s/|s| (once per call when the caller passes a table)
//   corr_peaks_kernel   source-stationary correlation rho~ = <s^, F^>/1024 on fp16 MFMA 32x32x16 (fp32 accumulate): 64 sources per
//                       wave held in AGPRs, the frame's cells streamed by descriptor LDS-DMA; the maps are NEVER stored -- each lane
//                       keeps a running top-6 (value, cell) list in registers, merged per source at the end -> Rec {amax, <= 10
//                       candidate cells within EPS_C of the approximate maximum}
//   rescore_kernel      the candidates re-scored in fp32 from the fp32 master -> exact arg-max k* (first index on ties)
//   key_scan / scatter  counting sort of the round's sources by (frame, cell of k*): key-consecutive sources share window cells
//   refine_corr_dma     16 x 4 key-consecutive sources per workgroup: fp32-grade correlation of the union of their 15 x 15 windows
//                       around k*, split-fp16 MFMA 16x16x32 (hi hi + hi lo + lo hi) on cells streamed PRE-SPLIT by LDS-DMA
//                       (featsplit_kernel, once per volume) -> per-source fp32 windows
//   refine_head_kernel  one wave per source: the 3x3 / 3x3 refiner on the window as chained f32 MFMAs (exact fp32 products), a
//                       CERTIFICATE that the zero-mass fallback of tracker_head.py:86-94 cannot fire (then the softmax statistics
//                       of the whole map cancel out of the result), disk soft arg-max
//   tier 2 (whole-map statistics: corr16_tiled + head16 + refine) and tier 3 (track_exact.hip, fp32 everything) take the sources
//   the fast tier cannot certify or whose candidate list overflowed -- device-side lists and counts, no host synchronisation.
// Everything that decides the output (arg-max, logits inside the disk, soft arg-max) is fp32; the fp16 pass only proposes
// candidates, inside a proven band (tests/test_numeric_claims.py).
// Rounds 1-2 history (kept as tier 2): corr16 -> fp16 maps in an Infinity-Cache-sized chunk -> head16 (v_dot2 refiner over the whole
// map for (zmax, Z)) -> refine.
#include <limits.h>
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "head_common.h"

int dtk_track_exact(const dtk_geom* g, const float* feat, const float* norms, const float* head, const float* emb,
                    const int32_t* src_row, const int32_t* tgt, const int32_t* out_idx, float* out_xy, int M,
                    const int32_t* dM, int normalized, void* workspace, size_t workspace_bytes, void* stream);
size_t dtk_track_exact_workspace_bytes(const dtk_geom* g, int M);

namespace {

typedef _Float16 half_t;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr float FSCALE = 32.f;            // fp16 operands carry 32 x unit vectors (keeps small components normal)
constexpr float INV_SCALE2 = 1.f / 1024.f;
constexpr float EPS_C = 3e-3f;            // candidate window: 2 x (fp16 operand + fp16 storage error bound)
constexpr int KC = 10;                    // candidates kept per source
constexpr int CM = 64, CN = 128, CK = 32; // corr16 tile
constexpr int NB_MAX = 1024;              // refine: largest window-union box (cells) correlated as one group
constexpr int RD = 5;                     // disk radius in cells supported by refine32 (radius / stride <= 5)
constexpr int MFMA_CHUNK = 16384;
constexpr int MFMA_SUPER = 4194304;   // sources per round (round 4: 524 288 -> 4 M, 285.5 -> 279.8 ms per benchmark step: fewer, fuller launches
                                      // and denser key-sorted windows; 7.3 GB of workspace at full use)

struct Rec {  // per source, written by head16, read by refine32
    float amax;
    int ncand;
    int cand[KC];
    float zmax, Z;
};

// fp16 path geometry: every map row is padded to PWP = a multiple of the 128-cell GEMM tile, so that an N-tile is (part
// of) ONE map row and the output tile can be stored with 16-byte pieces; a stored map is [(ph+2) rows][XW cols] with
// 8 zero columns left of the cells (16-byte alignment) and >= 8 zero columns to their right.
__host__ __device__ inline int pw_pad(int pw) { return (pw + CN - 1) / CN * CN; }
__host__ __device__ inline int map_xw(int pw) { return pw_pad(pw) + 16; }
__host__ __device__ inline int hw_pad(int ph, int pw) { return ph * pw_pad(pw); }

// ---- fp16 unit-norm copy of the feature volume ---------------------------------------------------------------
__global__ __launch_bounds__(256) void feat16_kernel(const float* __restrict__ feat, const float* __restrict__ norms,
                                                     half_t* __restrict__ f16, int T, int ph, int pw, int PWP, int C) {
    const long long cell = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);  // over T * ph * PWP padded cells
    const int HWp = ph * PWP, HW = ph * pw;
    if (cell >= (long long)T * HWp) return;
    const int lane = threadIdx.x & 63;
    const int t = (int)(cell / HWp), pc = (int)(cell % HWp);
    const int r = pc / PWP, c = pc - r * PWP;
    half_t* o = f16 + cell * C;
    if (c >= pw) {
        for (int k = lane * 8; k < C; k += 512) *reinterpret_cast<uint4*>(o + k) = make_uint4(0, 0, 0, 0);
        return;
    }
    const float* p = feat + ((size_t)t * HW + r * pw + c) * C;
    const float nrm = norms[(size_t)t * HW + r * pw + c];
    const float sc = nrm > 1e-30f ? FSCALE / nrm : 0.f;
    for (int k = lane * 8; k < C; k += 512) {
        const float4 a = *reinterpret_cast<const float4*>(p + k), b = *reinterpret_cast<const float4*>(p + k + 4);
        h8 v = {(half_t)(a.x * sc), (half_t)(a.y * sc), (half_t)(a.z * sc), (half_t)(a.w * sc),
                (half_t)(b.x * sc), (half_t)(b.y * sc), (half_t)(b.z * sc), (half_t)(b.w * sc)};
        *reinterpret_cast<h8*>(o + k) = v;
    }
}

// widths served by corr_peaks_wide_kernel (K split over wave pairs): the ViT-B / ViT-L feature widths
__host__ __device__ inline bool peaks_wide_ok(int C) { return C == 768 || C == 1024; }

// ---- split-fp16 copy of the feature volume (C = 384), the streamed operand of refine_corr_dma ---------------------------
// fs[cell][chunk kc of 32 channels][hi 32 | lo 32]: x * 32 = hi + lo, both fp16 -- exactly the halves the window
// correlation otherwise makes while it stages fp32 rows into LDS, made ONCE per volume.  A (cell, chunk) is one 128-byte
// line, so an LDS-DMA request of 8 cells fetches 8 whole lines and a step needs no VALU and no ds_write at all.
// The scale (round 6): 2^5 as long as the volume's largest cell norm allows it, otherwise the largest power of two with
// scale x max norm <= 2^14 -- chosen once per volume on the device (rcscale_*_kernel) and kept in a slot at the end of the feat_f16
// buffer, which the window-correlation kernels read.  Rounds 2-5 had the constant 2^5: features with DINOv2-like outlier statistics
// (token norms ~4 000, components beyond 2 047) left the fp16 range in a few cells -- the window value became fmaxf(NaN, 0) = 0 and 15
// of 92 160 positions moved by 0.02 .. 1.3 px (profiles/r06_e2e_error_outlier_854x476x90_1024q.json, first run).  A power of two
// changes no bit of the results where the old scale did not overflow (benign features keep 2^5).
constexpr float RC_SCALE_MAX = 32.f;
__host__ __device__ inline size_t split_planes_offset(const dtk_geom* g) {
    return ((size_t)g->T * hw_pad(g->ph, g->pw) * g->C * 2 + 255) / 256 * 256;
}
__host__ __device__ inline bool has_split_planes(const dtk_geom* g) {
    // (32-bit offsets in the descriptor.  Round 6 measured the wide widths on this form too -- refine_corr_dma_kernel<32> at C = 1024:
    // 256 registers of stationary sources leave one workgroup per CU, 44.0 ms per benchmark step with a ring of 4 stages, 44.6 with 8,
    // against 37.3 ms for the generic refine_corr_kernel -- so C = 768 / 1024 stay on the generic kernel and carry no split planes)
    return g->C == 384 && (long long)g->T * g->ph * g->pw * g->C * 4 < (1LL << 32);
}
// slot [0]: the scale (float); [1]: bits of the largest cell norm (scratch of the reduction)
__host__ __device__ inline size_t rc_scale_offset(const dtk_geom* g) {
    const size_t unit = (size_t)g->T * hw_pad(g->ph, g->pw) * g->C * 2;
    const size_t end = has_split_planes(g) ? split_planes_offset(g) + (size_t)g->T * g->ph * g->pw * g->C * 4 : unit;
    return (end + 255) / 256 * 256;
}
__global__ __launch_bounds__(256) void rcscale_max_kernel(const float* __restrict__ norms, long long n, unsigned* __restrict__ slot) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = norms[i];
        m = fmaxf(m, v == v ? v : 3.0e38f);   // (a NaN norm: the smallest scale)
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(slot + 1, __float_as_uint(m));   // norms are >= 0: the bit patterns order like the values
}
__global__ void rcscale_final_kernel(unsigned* __restrict__ slot) {
    const float m = __uint_as_float(slot[1]);
    float sc = RC_SCALE_MAX;
    while (sc > 0x1p-40f && m * sc > 16384.f) sc *= 0.5f;
    reinterpret_cast<float*>(slot)[0] = sc;
}
__global__ __launch_bounds__(256) void featsplit_kernel(const float* __restrict__ feat, half_t* __restrict__ fs,
                                                        long long n8, const float* __restrict__ rc_scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // one 8-channel piece
    if (i >= n8) return;
    const float RC_SCALE = *rc_scale;
    const float4 a = *reinterpret_cast<const float4*>(feat + i * 8), b = *reinterpret_cast<const float4*>(feat + i * 8 + 4);
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float xs = x[e] * RC_SCALE;
        hi[e] = (half_t)xs;
        lo[e] = (half_t)(xs - (float)hi[e]);
    }
    const long long chunk = i >> 2;  // 32 channels = 4 pieces; C % 32 == 0, so chunks never straddle cells
    const int piece = (int)(i & 3);
    *reinterpret_cast<h8*>(fs + chunk * 64 + piece * 8) = hi;
    *reinterpret_cast<h8*>(fs + chunk * 64 + 32 + piece * 8) = lo;
}

// ---- sources of a chunk -> fp16 unit vectors; one wave per source ---------------------------------------------------
__global__ __launch_bounds__(256) void src16_kernel(const float* __restrict__ emb, const int32_t* __restrict__ src_row,
                                                    half_t* __restrict__ s16, int m0, int count, int M,
                                                    const int32_t* __restrict__ dM, int C, float scale,
                                                    float* __restrict__ rown) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= count) return;
    const int lane = threadIdx.x & 63;
    const int m = m0 + i;
    half_t* o = s16 + (size_t)i * C;
    if (m >= dtk_active(M, dM)) {
        for (int k = lane * 8; k < C; k += 512) *reinterpret_cast<uint4*>(o + k) = make_uint4(0, 0, 0, 0);
        return;
    }
    const float* p = emb + (size_t)(src_row ? src_row[m] : m) * C;
    float s = 0.f;
    for (int k = lane * 4; k < C; k += 256) {
        const float4 v = *reinterpret_cast<const float4*>(p + k);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = sqrtf(wave_sum(s));
    if (rown && lane == 0) rown[i] = s;   // the table form: |row|, the same sum in the same order as rescore_kernel's own (it reads this instead)
    const float sc = s > 1e-30f ? scale / s : 0.f;
    for (int k = lane * 8; k < C; k += 512) {
        const float4 a = *reinterpret_cast<const float4*>(p + k), b = *reinterpret_cast<const float4*>(p + k + 4);
        h8 v = {(half_t)(a.x * sc), (half_t)(a.y * sc), (half_t)(a.z * sc), (half_t)(a.w * sc),
                (half_t)(b.x * sc), (half_t)(b.y * sc), (half_t)(b.z * sc), (half_t)(b.w * sc)};
        *reinterpret_cast<h8*>(o + k) = v;
    }
}

// ---- corr16: fp16 MFMA GEMM, 64 sources x 128 cells per workgroup ---------------------------------------------------
// LDS tiles are [row][32 k] fp16 = 4 x 16-byte pieces per row; piece g of row r lives at r*4 + (g ^ SWZ[(r>>2)&3]),
// which makes the ds_read_b128 fragment reads (lane = 16*g + r%16) conflict-free.
__device__ __forceinline__ int swz(int row, int piece) {
    const int f = (0x1230 >> (((row >> 2) & 3) * 4)) & 3;  // {0,3,2,1}
    return row * 4 + (piece ^ f);
}

// PEAKS = false: the fp16 maps are stored (whole-map refiner statistics follow in head16).
// PEAKS = true : nothing of the correlation volume leaves the chip: per (source, N-tile) only a 12-byte record is kept --
//                the two largest values of the tile with their cells and the number of cells within EPS_C of the
//                tile maximum -- from which select_kernel derives the approximate maximum and the candidate cells.
struct TileRec {
    float v1, v2;
    uint32_t idx;  // i1 | i2 << 8 | count << 16  (columns inside the tile; count saturates at 255)
};

template <bool PEAKS>
__global__ __launch_bounds__(256) void corr16_tiled_kernel(dtk_geom g, const half_t* __restrict__ f16,
                                                     const half_t* __restrict__ s16, const int32_t* __restrict__ tgt,
                                                     half_t* __restrict__ maps, TileRec* __restrict__ trec, int m0,
                                                     int count, int M, const int32_t* __restrict__ dM, int HWp, int MP,
                                                     int dbg) {
    __shared__ uint4 smem_ab[2 * (CM + CN) * 4];  // 24 KB: A/B double buffers; reused as the 64 x 128 fp16 output tile
    uint4 (*As)[CM * 4] = reinterpret_cast<uint4 (*)[CM * 4]>(smem_ab);
    uint4 (*Bs)[CN * 4] = reinterpret_cast<uint4 (*)[CN * 4]>(smem_ab + 2 * CM * 4);
    half_t* Ts = reinterpret_cast<half_t*>(smem_ab);  // [CM][CN + 8]
    constexpr int TP16 = CN + 8;
    __shared__ int s_tgt[CM];
    __shared__ int s_fr[2];
    const int active = min(dtk_active(M, dM), m0 + count);
    // block -> (source tile, N-tile): workgroup L runs on XCD L % 8 (observed dispatch order; speed only).  Each XCD gets
    // the source tiles mt = 8*j + xcd and walks the N-tiles in panels of 8, so that its private L2 holds one B panel
    // (8 x 96 KB at C = 384) plus its share of the A tiles instead of streaming the whole frame per source tile.
    const int NT = HWp / CN, MT = (count + CM - 1) / CM;
    int bx, by;
    {
        const int L = blockIdx.x, xcd = L & 7, k = L >> 3;
        const int mt_per = (MT + 7) >> 3;                     // source tiles per XCD
        const int per_panel = mt_per * 8;                     // (source tile, N-tile-in-panel) pairs per panel per XCD
        const int p = k / per_panel, rem = k - p * per_panel;
        by = (rem >> 3) * 8 + xcd;
        bx = p * 8 + (rem & 7);
        if (by >= MT || bx >= NT) return;
    }
    const int tile_m0 = m0 + by * CM;
    if (tile_m0 >= active) return;
    const int cell0 = bx * CN;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < CM) {
        const int m = tile_m0 + tid;
        const bool ok = m < active;
        int f = ok ? min(max(tgt[m], 0), g.T - 1) : -1;
        s_tgt[tid] = f;
        int lo = ok ? f : INT_MAX, hi = f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o, WAVE));
            hi = max(hi, __shfl_xor(hi, o, WAVE));
        }
        if (tid == 0) { s_fr[0] = lo; s_fr[1] = hi; }
    }
    __syncthreads();
    const int fmin = s_fr[0], fmax = s_fr[1];
    const int wr = w >> 1, wc = w & 1;             // wave tile: rows wr*32.., cols wc*64..
    const int fj = lane & 15, fg = lane >> 4;      // fragment row / k-piece
    const int lrow = tid >> 2, lpiece = tid & 3;   // loader: row, 16-byte piece
    const int nk = g.C / CK;
    const half_t* arow = s16 + (size_t)(tile_m0 - m0 + lrow) * g.C + lpiece * 8;
    for (int f = fmin; f <= fmax; ++f) {
        bool mine = false;
#pragma unroll
        for (int i = 0; i < 8; ++i) mine |= (s_tgt[wr * 32 + fg * 4 + (i & 3) + (i >> 2) * 16] == f);
        if (!__syncthreads_or(mine)) continue;
        const half_t* b0 = f16 + ((size_t)f * HWp + cell0 + lrow) * g.C + lpiece * 8;
        const half_t* b1 = b0 + (size_t)64 * g.C;
        f4 acc[2][4];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f4{0.f, 0.f, 0.f, 0.f};
        uint4 ra = *reinterpret_cast<const uint4*>(arow);
        uint4 rb0 = *reinterpret_cast<const uint4*>(b0), rb1 = *reinterpret_cast<const uint4*>(b1);
        As[0][swz(lrow, lpiece)] = ra;
        Bs[0][swz(lrow, lpiece)] = rb0;
        Bs[0][swz(lrow + 64, lpiece)] = rb1;
        __syncthreads();
        int cur = 0;
        for (int ks = 0; ks < nk; ++ks) {
            if (ks + 1 < nk) {
                ra = *reinterpret_cast<const uint4*>(arow + (ks + 1) * CK);
                rb0 = *reinterpret_cast<const uint4*>(b0 + (ks + 1) * CK);
                rb1 = *reinterpret_cast<const uint4*>(b1 + (ks + 1) * CK);
            }
            h8 af[2], bf[4];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const uint4 v = As[cur][swz(wr * 32 + mi * 16 + fj, fg)];
                af[mi] = *reinterpret_cast<const h8*>(&v);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const uint4 v = Bs[cur][swz(wc * 64 + ni * 16 + fj, fg)];
                bf[ni] = *reinterpret_cast<const h8*>(&v);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
            if (ks + 1 < nk) {
                As[cur ^ 1][swz(lrow, lpiece)] = ra;
                Bs[cur ^ 1][swz(lrow, lpiece)] = rb0;
                Bs[cur ^ 1][swz(lrow + 64, lpiece)] = rb1;
            }
            __syncthreads();
            cur ^= 1;
        }
        // D: lane (fg, fj) holds rows 4*fg + r, column fj of each 16x16 tile.  The tile is transposed through LDS and
        // stored as 16-byte pieces: an N-tile is 128 consecutive columns of ONE map row (padded cells are zero), and maps
        // are written in the layout head16 copies into LDS verbatim (zero border, written once per call by a memset).
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr * 32 + mi * 16 + fg * 4 + r;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    Ts[row * TP16 + wc * 64 + ni * 16 + fj] = (half_t)fmaxf(acc[mi][ni][r] * INV_SCALE2, 0.f);
            }
        __syncthreads();
        const int PWP = pw_pad(g.pw);
        const int mr = cell0 / PWP, mc = cell0 - mr * PWP;
        if (!PEAKS) {
            const size_t po = (size_t)(mr + 1) * map_xw(g.pw) + 8 + mc;
            const int piece = tid & 15, r0 = tid >> 4;
#pragma unroll
            for (int rr = 0; rr < CM; rr += 16) {
                const int row = rr + r0;
                if (s_tgt[row] == f && !DTK_DBG(dbg, 256))
                    *reinterpret_cast<uint4*>(maps + (size_t)(tile_m0 - m0 + row) * MP + po + piece * 8) =
                        *reinterpret_cast<const uint4*>(Ts + row * TP16 + piece * 8);
            }
        } else {
            // four threads per source row, 32 cells each: top-2 with positions, then the band count
            const int row = tid >> 2, qd = tid & 3;
            const half_t* tr = Ts + row * TP16 + qd * 32;
            float v1 = -1.f, v2 = -1.f;
            int i1 = 0, i2 = 0;
            const int ncol = g.pw - mc - qd * 32;  // cells of this quarter that exist (padded columns are not cells)
#pragma unroll
            for (int p8 = 0; p8 < 4; ++p8) {
                const uint4 u = *reinterpret_cast<const uint4*>(tr + p8 * 8);
                const h8 hv = *reinterpret_cast<const h8*>(&u);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = (p8 * 8 + e < ncol) ? (float)hv[e] : -1.f;
                    const int ci = qd * 32 + p8 * 8 + e;
                    if (v > v1) { v2 = v1; i2 = i1; v1 = v; i1 = ci; }
                    else if (v > v2) { v2 = v; i2 = ci; }
                }
            }
            // merge the four quarters (lanes tid^1, tid^2 hold the other quarters of the same row)
#pragma unroll
            for (int o = 1; o <= 2; o <<= 1) {
                const float w1 = __shfl_xor(v1, o, WAVE), w2 = __shfl_xor(v2, o, WAVE);
                const int j1 = __shfl_xor(i1, o, WAVE), j2 = __shfl_xor(i2, o, WAVE);
                // top-2 of {v1, v2, w1, w2}; ties resolved towards the lower column
                const bool a_first = v1 > w1 || (v1 == w1 && i1 < j1);
                const float n1 = a_first ? v1 : w1;
                const int m1 = a_first ? i1 : j1;
                const float c1 = a_first ? v2 : v1, c2 = a_first ? w1 : w2;  // runners-up of either side
                const int d1 = a_first ? i2 : i1, d2 = a_first ? j1 : j2;
                const bool c_first = c1 > c2 || (c1 == c2 && d1 < d2);
                v1 = n1; i1 = m1;
                v2 = c_first ? c1 : c2; i2 = c_first ? d1 : d2;
            }
            int cnt = 0;
            const float band = v1 - EPS_C;
#pragma unroll
            for (int p8 = 0; p8 < 4; ++p8) {
                const uint4 u = *reinterpret_cast<const uint4*>(tr + p8 * 8);
                const h8 hv = *reinterpret_cast<const h8*>(&u);
#pragma unroll
                for (int e = 0; e < 8; ++e) cnt += (p8 * 8 + e < ncol && (float)hv[e] >= band) ? 1 : 0;
            }
            cnt += __shfl_xor(cnt, 1, WAVE);
            cnt += __shfl_xor(cnt, 2, WAVE);
            if (qd == 0 && s_tgt[row] == f) {
                TileRec r;
                r.v1 = v1; r.v2 = v2;
                r.idx = (uint32_t)i1 | ((uint32_t)i2 << 8) | ((uint32_t)min(cnt, 255) << 16);
                trec[(size_t)(tile_m0 - m0 + row) * (HWp / CN) + bx] = r;
            }
        }
        __syncthreads();  // the next frame of a mixed tile restages As/Bs
    }
}

// ---- head16: approximate statistics of one map per workgroup ---------------------------------------------------------
// wpk: packed fp16 weights built by head16_pack_kernel: [0..79] conv1 tap pairs per channel (16 x 5 h2),
// [80..151] conv2 channel pairs per tap (9 x 8 h2); biases stay fp32 in `head`.
__global__ void head16_pack_kernel(const float* __restrict__ head, uint32_t* __restrict__ wpk) {
    const int i = threadIdx.x;
    auto pack = [](float a, float b) {
        h2 v = {(half_t)a, (half_t)b};
        return *reinterpret_cast<uint32_t*>(&v);
    };
    if (i < 80) {
        const int ch = i / 5, p = i % 5;
        const float a = head[ch * 9 + 2 * p], b = (2 * p + 1 < 9) ? head[ch * 9 + 2 * p + 1] : 0.f;
        wpk[i] = pack(a, b);
    } else if (i < 152) {
        const int tap = (i - 80) / 8, cp = (i - 80) % 8;
        wpk[i] = pack(head[160 + (2 * cp) * 9 + tap], head[160 + (2 * cp + 1) * 9 + tap]);
    } else if (i == 152) {
        // bound on |z_fp16pass - z_exact| for any cell: the fp16 pass sees x within DX of the exact relu'd cosine and
        // rounds weights, x and the hidden activations to fp16 (relative 2^-11 each); x <= 1.
        const float DX = 2e-3f, U = 1.f / 1024.f;
        float e = 0.f;
        for (int ch = 0; ch < 16; ++ch) {
            float s1 = 0.f, s2 = 0.f;
            for (int t = 0; t < 9; ++t) { s1 += fabsf(head[ch * 9 + t]); s2 += fabsf(head[160 + ch * 9 + t]); }
            e += s2 * (s1 * DX + 2.f * U * (fabsf(head[144 + ch]) + s1));
        }
        reinterpret_cast<float*>(wpk)[152] = e;
    } else if (i >= 160 && i < 176) {
        // constants of the upper bound z_ub(a) = b2 + sum_ch [ W2p relu(b1 + P1 a) + W2n relu(b1 + N1 a) ] of the refiner
        // output over ANY map with values in [0, a]: P1/N1 = positive / negative tap sums of conv1, W2p/W2n of conv2
        const int ch = i - 160;
        float p1 = 0.f, n1 = 0.f, p2 = 0.f, n2 = 0.f;
        for (int t = 0; t < 9; ++t) {
            const float a = head[ch * 9 + t], b = head[160 + ch * 9 + t];
            p1 += fmaxf(a, 0.f); n1 += fminf(a, 0.f);
            p2 += fmaxf(b, 0.f); n2 += fminf(b, 0.f);
        }
        float* cf = reinterpret_cast<float*>(wpk) + 160;
        cf[ch] = p1; cf[16 + ch] = n1; cf[32 + ch] = p2; cf[48 + ch] = n2;
    }
}

__device__ __forceinline__ h2 as_h2(uint32_t u) { return *reinterpret_cast<h2*>(&u); }

// select_kernel (fast mode): one wave per source -- approximate maximum and candidate cells from the per-tile records of
// corr16<PEAKS>.  A tile whose maximum lies within EPS_C of the global one but that holds more than two cells in its own
// band may hide further candidates: such a source is marked as overflowing and is re-done by the exact path.
__global__ __launch_bounds__(256) void select_kernel(dtk_geom g, const TileRec* __restrict__ trec, int ntiles,
                                                     Rec* __restrict__ rec, int m0, int count, int M,
                                                     const int32_t* __restrict__ dM) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= count || m0 + i >= dtk_active(M, dM)) return;
    const int PWP = pw_pad(g.pw);
    const TileRec* tr = trec + (size_t)i * ntiles;
    float amax = 0.f;
    for (int t = lane; t < ntiles; t += WAVE) amax = fmaxf(amax, tr[t].v1);
    amax = wave_max(amax);
    const float thr = amax - EPS_C;
    int ncand = 0, mine[4];
    bool overflow = false;
    for (int t = lane; t < ntiles; t += WAVE) {
        const TileRec r = tr[t];
        if (r.v1 < thr) continue;
        const int cell0 = t * CN, mr = cell0 / PWP, mc = cell0 - mr * PWP;
        const int c1 = r.idx & 255, c2 = (r.idx >> 8) & 255, cnt = r.idx >> 16;
        if (ncand < 4) mine[ncand] = mr * g.pw + mc + c1;
        ++ncand;
        if (r.v2 >= thr) {
            if (ncand < 4) mine[ncand] = mr * g.pw + mc + c2;
            ++ncand;
        }
        if (cnt > 2) overflow = true;
    }
    int total = ncand;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(total, o, WAVE);
        if (lane >= o) total += y;
    }
    const int base = total - ncand;
    int all = __shfl(total, 63, WAVE);
    if (__any(overflow)) all = KC + 1;
    Rec* r = rec + i;
    for (int k = 0; k < ncand && k < 4; ++k)
        if (base + k < KC) r->cand[base + k] = mine[k];
    if (lane == 0) { r->amax = amax; r->ncand = all; r->zmax = 0.f; r->Z = -1.f; }
}

// ---- corr_peaks (fast mode, C = 384): source-stationary correlation + candidate selection in one kernel ---------------
// A workgroup owns 256 sources of one target frame (a wave 64 of them) for the whole frame: the sources sit in registers
// as the B operand of the 32x32x16 MFMA (24 k-steps x 2 column tiles = 192 VGPRs), the frame's cells stream through a
// triple-buffered 24 KB LDS tile of 32 cells filled by LDS-DMA (global_load_lds_dwordx4, no staging registers) and are
// read once per wave as the A operand: one ds_read_b128 feeds two MFMAs, half the LDS traffic per flop of the tiled
// kernel, and no correlation value ever leaves the registers.  In D a lane holds ONE source and 16 cells, so the
// running top-4 of a source is lane-local: each value gets its 13-bit (step, register) position OR-ed into the low
// mantissa bits (costs 2^-10 relative, covered by TRUNC_C in the candidate band) and is pushed through three v_med3 and
// one v_max.  The cell <-> MFMA row assignment puts the cells on a checkerboard over the two lane halves, so a peak and
// its neighbours split evenly between the two top-4 lists of a source.  The epilogue of step n-1 is
// independent of the MFMAs of step n and is interleaved with them by the scheduler (one wave per SIMD: 512 VGPRs).
constexpr int PK_SRC = 256;               // sources per workgroup
constexpr int PK_CB = 1;                  // 32-cell blocks per step in production (2 = the four-accumulator variant: measured slower)
constexpr int PK_IDX_BITS = 13;           // position tag: step << 5 | cell block << 4 | accumulator register
constexpr int PK_VAL_BITS = 17;           // sources carry 2^12 x, cells 2^5 x unit vectors: the accumulator is 2^17 rho
constexpr float PK_SRC_SCALE = 4096.f;
constexpr int PK_TOP = 6;                 // list length per lane half
// candidate band of the fused pass: two fp16 operand roundings bound |rho16 - rho| by 2^-10 (Cauchy-Schwarz over the
// relative errors), so the exact arg-max cell is within 2 x 2^-10 of the fp16 maximum; + fp32 accumulation and the
// truncation to PK_VAL_BITS
constexpr float EPS_PK = 2.1e-3f;
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr size_t peaks_lds_bytes(int cb) { return 3 * (size_t)(32 * cb) * 384 * 2 + 64; }   // three tiles + the frame range (dynamic LDS)

// (LDS-DMA requests go through dtk_buffer_lds16, common.h; issued from asm so that the compiler does not serialise them against
// the ds_reads of the OTHER buffer with a vmcnt(0) of its own: completion is awaited explicitly -- glds_wait -- before the
// barrier that publishes the tile.)
template <int N>
__device__ __forceinline__ void glds_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// Sorted top-6 insertion of one accumulator value as a program of eight VALU instructions: value -> key (fixed point -- the
// accumulator already is 2^17 rho; negative values stay negative and never enter a list that starts at zero -- shifted above
// the 13-bit position tag, which is wave-uniform: an SGPR operand), then one v_med3_i32 per list entry from the tail up and a
// v_max_i32 for the head.  top_push_op<OP> issues instruction OP of it, so that the caller can spread the 32 x 8 instructions of
// a tile EVENLY over its 48 MFMA slots (5 or 6 per slot): a lone wave hides about five VALU instructions under a 32x32x16 MFMA
// and pays ~6 cycles for each further one (profiles/r04_slot_rate_one_wave_per_simd.txt), so eight in two slots and none in the
// third -- the round-1 arrangement -- cost ~45 cycles per slot where 5.3 in every slot cost ~38.
__device__ __forceinline__ int med3_i32(int a, int b, int c) {
    int d;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
template <int OP>
__device__ __forceinline__ void top_push_op(int (&v)[PK_TOP], float x, int tag, int& key) {
    static_assert(PK_TOP == 6, "the insertion program is written out for six entries");
    if constexpr (OP == 0) key = (int)x;                                   // v_cvt_i32_f32
    else if constexpr (OP == 1) {   // v_lshl_or_b32 with the tag in an SGPR (opaque to the compiler, which otherwise spends a second
        int t2;                     // VALU instruction -- v_lshlrev + v_or3 -- on folding the register index into it)
        asm("s_mov_b32 %0, %1" : "=s"(t2) : "s"(tag));
        key = (key << PK_IDX_BITS) | t2;
    }
    else if constexpr (OP < 7) v[7 - OP] = med3_i32(v[6 - OP], v[7 - OP], key);   // OP 2..6: entries 5, 4, 3, 2, 1
    else v[0] = max(v[0], key);
}
__device__ __forceinline__ void top_push_value(int (&v)[PK_TOP], float x, int tag) {
    int key;
    top_push_op<0>(v, x, tag, key); top_push_op<1>(v, x, tag, key); top_push_op<2>(v, x, tag, key); top_push_op<3>(v, x, tag, key);
    top_push_op<4>(v, x, tag, key); top_push_op<5>(v, x, tag, key); top_push_op<6>(v, x, tag, key); top_push_op<7>(v, x, tag, key);
}

// (The 192 source registers of a wave are only ever MFMA operands: loaded STRAIGHT INTO AGPRs by the asm that defines them, they
// stay there -- the builtin MFMA takes a B operand from an AGPR as it is -- and everything the VALU touches (accumulators, lists,
// fragment ring, addresses) fits the architectural VGPRs; left to the register allocator the kernel sat at exactly 256 VGPRs and
// shuttled ~34 values per tile through v_accvgpr_read.)

// VAR: development variants (DTK_DEBUG bits 8192 / 16384 / 32768): 1 = no top-N updates, 2 = no tile requests after the
// first two, 4 = no LDS reads.  0 in production.  (Measured with them: MFMAs alone 1.63 PF; + LDS reads or + DMA alone
// unchanged; both 1.22 PF; + list updates 0.90 PF.  Staging the tiles through registers instead of LDS-DMA: 0.39 PF.)
template <int I, int N, class F>
__device__ __forceinline__ void peaks_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        peaks_static_for<I + 1, N>(f);
    }
}

// LDS-DMA requests I .. N-1 of a tile for one wave: request I = (cell block I / NL, cache line I % NL of a cell's 768-byte row)
template <int I, int N, int NL, int C>
__device__ __forceinline__ void peaks_dma(dtk_u4 srd, unsigned toff, unsigned voff, unsigned dst) {
    if constexpr (I < N) {
        dtk_buffer_lds16<I * 4096>(srd, toff + (unsigned)((I % NL) * 128 + (I / NL) * 32 * C * 2), voff, dst);
        peaks_dma<I + 1, N, NL, C>(srd, toff, voff, dst);
    }
}

// CB = 32-cell blocks per step.  CB = 2 (round 4 experiment): a step is 64 cells, so that FOUR accumulators (2 source tiles x 2
// cell blocks) take turns instead of two and the barriers per cell halve -- measured SLOWER than CB = 1 (816 against 955 TFLOP/s,
// profiles/r04_corr_peaks_ablations.txt), kept as a development variant.
__device__ __forceinline__ int cell_key(int cell, int pw);
// Round 5: a source whose list holds ONE candidate with a clearly positive approximate maximum is finished here -- that cell is
// the exact arg-max (rescore_kernel's single-candidate case: the band proof, tests/test_numeric_claims.py), |source| comes from the
// row table's norms -- so the record is marked done (ncand = -1), k*, |s| and the (frame, cell key) histogram are written from this
// kernel's epilogue, and rescore_kernel's wave for the source returns at once.  kstar == nullptr: off (no row table, arg-max-only calls).
struct PeaksDone {
    int32_t* kstar;
    float* snorm;
    int32_t* hist;
    const float* rown;
    int HWk;
};
template <int KS, int VAR, int CB>
__global__ __launch_bounds__(256) void corr_peaks_kernel(dtk_geom g, const half_t* __restrict__ f16,
                                                         const half_t* __restrict__ s16, const int32_t* __restrict__ tgt,
                                                         Rec* __restrict__ rec, int m0, int count, int HWp,
                                                         const int32_t* __restrict__ row_of, PeaksDone done) {
    constexpr int C = KS * 16;
    constexpr int PK_CELLS = 32 * CB;            // cells per step: CB 32-cell blocks
    constexpr int TSH = CB > 1 ? 5 : 4;          // position tag: step << TSH | cell block << 4 | accumulator register
    constexpr int TILE_BYTES = PK_CELLS * C * 2;
    extern __shared__ __attribute__((aligned(1024))) unsigned char cells_dyn[];   // [3][TILE_BYTES] + s_fr[8]
    unsigned char* cells = cells_dyn;
    int* s_fr = reinterpret_cast<int*>(cells_dyn + 3 * TILE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int src0 = blockIdx.x * PK_SRC + w * 64;  // first source of this wave (index inside the launch)
    // target frames of this lane's two sources; frame range of the workgroup
    int tf[2];
    {
        int lo = INT_MAX, hi = -1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int i = src0 + t * 32 + j;
            tf[t] = i < count ? min(max(tgt[m0 + i], 0), g.T - 1) : -1;
            if (tf[t] >= 0) { lo = min(lo, tf[t]); hi = max(hi, tf[t]); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o, WAVE));
            hi = max(hi, __shfl_xor(hi, o, WAVE));
        }
        if (lane == 0) { s_fr[w] = lo; s_fr[4 + w] = hi; }
    }
    // the sources: B operand, lane (j, h) holds source j, k = 16 ks + 8 h .. + 7.  Loaded STRAIGHT INTO AGPRs (the asm below is
    // what defines them, so the register allocator keeps them there: a value defined in a VGPR and merely used through an "a"
    // constraint is copied at every use); sources past `count` read the last valid row -- their lists are never written out.
    h8 bs[2][KS];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int i = src0 + t * 32 + j;
        // row_of != NULL: s16 is the fp16 table of ALL rows of emb, made once per call (a source is row row_of[m]); else the
        // round's sources converted in order by src16_kernel
        const int ic = min(i, count - 1);
        const half_t* sp = s16 + (size_t)(row_of ? row_of[m0 + ic] : ic) * C + h * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(bs[t][ks]) : "v"(sp + ks * 16) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (the asm loads are invisible to the compiler's vmcnt bookkeeping: re-define every loaded register behind the wait -- empty
    //  volatile asms keep their order, and no use of bs can be scheduled above its re-definition)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(bs[t][ks]));
    __syncthreads();
    const int fmin = min(min(s_fr[0], s_fr[1]), min(s_fr[2], s_fr[3]));
    const int fmax = max(max(s_fr[4], s_fr[5]), max(s_fr[6], s_fr[7]));
    // The LDS image of a 32-cell block, chosen so that ONE LDS-DMA request touches 8 cache lines instead of 32 (a VMEM
    // instruction costs the issuing wave one address-unit slot per cache line: DESIGN section 3; round 1-3's image
    // [k-step][k-half][cell] made every request gather 32 B from each of 32 cells and cost a lone wave ~136 cycles of issue):
    //   [line L of a cell's row: 64 k-values = 4 k-steps][position pos of the cell][8 pieces of 16 B]      (NL x 32 x 128 B)
    // a cell's 128-byte line stays contiguous, so the 64 lanes of a request fetch 8 whole lines (lane l: line of position
    // 8 w + (l >> 3), piece slot l & 7).  For the fragment reads to be conflict-free (ds_read_b128 serves lane groups of 16:
    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32; the 16 pieces of a group must fall into 16 distinct 16-byte
    // bank slots) MFMA row j sits at position pi(j) = (j & 16) | (j & 7) << 1 | (j >> 3) & 1 and its pieces are XOR-swizzled by
    // j & 7: slot = (piece & 7) ^ (j & 7) -- through the SOURCE address of the DMA, whose LDS side is lane-linear.
    // MFMA row i = 8a + 4p + b carries cell 8a + 2b + p of the block (on odd map rows the two lane halves swap: a peak's
    // vertical neighbours split between the lists as well).
    constexpr int NL = KS / 4;                   // 128-byte lines per cell
    const int tiles_per_row = pw_pad(g.pw) / PK_CELLS;
    const unsigned lds_base = (unsigned)(size_t)cells;
    const int NT = HWp / PK_CELLS;
    const int band = (int)(EPS_PK * (float)(1 << PK_VAL_BITS)) << PK_IDX_BITS;
    const dtk_u4 srd = dtk_make_srd(f16);
    unsigned voff_even, voff_odd;
    {
        const int pos = 8 * w + (lane >> 3), slot = lane & 7;
        const int jr = (pos & 16) | ((pos >> 1) & 7) | ((pos & 1) << 3);      // the MFMA row whose cell sits at `pos`
        const int cell = 8 * (jr >> 3) + 2 * (jr & 3) + ((jr >> 2) & 1);
        const int piece = slot ^ (jr & 7);                                      // piece (within the line) that lands in `slot`
        voff_even = (unsigned)(cell * C * 2 + piece * 16);
        voff_odd = (unsigned)((cell ^ 1) * C * 2 + piece * 16);
    }
    const unsigned dma_dst = __builtin_amdgcn_readfirstlane(lds_base + w * 1024);
    // fragment addresses of lane (j, h): k-step ks -> line ks >> 2 (an immediate), piece 2 (ks & 3) + h inside it
    unsigned frag_off[4];
    {
        const int pi = (j & 16) | ((j & 7) << 1) | ((j >> 3) & 1);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) frag_off[k4] = (unsigned)(pi * 128 + (((2 * k4 + h) ^ (j & 7)) << 4));
    }
    for (int f = fmin; f <= fmax; ++f) {
        // tiles are requested in order (0, 1, 2, ...; past the end the last one again): the scalar offset of the next tile and
        // the parity of its map row advance incrementally (a division by the runtime tiles-per-row costs ~15 SALU instructions)
        unsigned next_off = __builtin_amdgcn_readfirstlane((unsigned)f * (unsigned)HWp * (unsigned)(C * 2));   // (volume < 4 GB: host check)
        int next_n = 0, next_in_row = 0, next_odd = 0;
        auto issue = [&](int n, int buf) {
            if ((VAR & 2) && n > 1) return;
            const unsigned voff = next_odd ? voff_odd : voff_even;
            const unsigned dst = __builtin_amdgcn_readfirstlane(dma_dst + (unsigned)buf * (unsigned)TILE_BYTES);
            peaks_dma<0, NL * CB, NL, C>(srd, next_off, voff, dst);
            if (next_n + 1 < NT) {   // (n == next_n except for the clamped repeats of the last tile)
                ++next_n;
                next_off += (unsigned)(PK_CELLS * C * 2);
                if (++next_in_row == tiles_per_row) { next_in_row = 0; next_odd ^= 1; }
            }
        };
        int v[2][PK_TOP];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int k = 0; k < PK_TOP; ++k) v[t][k] = 0;
        // one step: the MFMAs of the tile in `buf` into accN[cell block][source tile], interleaved with the list updates of
        // the previous tile's accP (step index np)
        auto step = [&](int buf, f16v (&accN)[CB][2], const f16v (&accP)[CB][2], int np) {
            const unsigned char* base = cells + (size_t)buf * TILE_BYTES;
            // fragment fr = (k-step fr / CB, cell block fr % CB)
            auto frag = [&](int fr) {
                const int ks = fr / CB, cb = fr % CB;
                return *reinterpret_cast<const h8*>(base + frag_off[ks & 3] + (cb * NL + (ks >> 2)) * 4096);
            };
            const int ib = np << TSH;
            const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // inline constant C
            int keys[2] = {0, 0};   // the key of the value whose insertion is in flight (one per source tile at most)
            h8 a[3];
            a[0] = frag(0);
            a[1] = frag(1);
            constexpr int NF = KS * CB;          // A fragments of the tile: fragment fr = k-step fr / CB, cell block fr % CB
            constexpr int NSLOT = 2 * NF, NINS = 32 * CB * 8;   // MFMA slots; list instructions (32 CB values x 8)
            // (compile-time recursion instead of `#pragma unroll`: at 96 slots the unroller gives up and the accumulators'
            // indices become run-time -- scratch memory)
            peaks_static_for<0, NSLOT>([&](auto qc) {
                constexpr int q = decltype(qc)::value, fr = q / 2, t = q % 2, ks = fr / CB, cb = fr % CB;
                if (t == 0 && fr + 2 < NF && !(VAR & 4)) a[(fr + 2) % 3] = frag(fr + 2);
                accN[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[fr % 3], bs[t][ks], ks == 0 ? zero16 : accN[cb][t], 0, 0, 0);
                // the list instructions of the previous tile, spread evenly over the MFMA slots: slot q issues instructions
                // [NINS q / NSLOT, NINS (q + 1) / NSLOT) of the stream; value e = index / 8 is accumulator register e / (2 CB) of
                // (source tile e & 1, cell block (e >> 1) % CB); position tag = step << 5 | cell block << 4 | register
                peaks_static_for<q * NINS / NSLOT, (q + 1) * NINS / NSLOT>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, e = i >> 3, lt = e & 1, lcb = (e >> 1) % CB, lr = e / (2 * CB);
                    if (!(VAR & 1) || e == 0) top_push_op<(i & 7)>(v[lt], accP[lcb][lt][lr], ib | (lcb << 4) | lr, keys[lt]);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        auto epi = [&](const f16v (&acc)[CB][2], int np) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int t = 0; t < 2; ++t) top_push_value(v[t], acc[cb][t][r], (np << TSH) | (cb << 4) | r);
        };
        f16v accA[CB][2], accB[CB][2];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) accB[cb][t][r] = 0.f;
        // three LDS buffers, tiles requested two steps ahead (an L2 miss takes longer than one step): at the end of step
        // n the tile of step n+1 must have landed while the requests of step n+2 stay in flight -> vmcnt(LQ CB).  Tiles past
        // the end are clamped to the last one (never read), which keeps that count uniform.
        issue(0, 0);
        issue(min(1, NT - 1), 1);
        glds_wait<NL * CB>();
        __syncthreads();
        int n = 0, b0 = 0;  // b0 = n % 3
        for (; n + 1 < NT; n += 2) {
            const int b1 = b0 == 2 ? 0 : b0 + 1, b2 = b1 == 2 ? 0 : b1 + 1;
            issue(min(n + 2, NT - 1), b2);
            step(b0, accA, accB, max(n - 1, 0));  // first step: accB = 0, pushes zeros
            glds_wait<NL * CB>();
            __syncthreads();
            issue(min(n + 3, NT - 1), b0);
            step(b1, accB, accA, n);
            glds_wait<NL * CB>();
            __syncthreads();
            b0 = b2;
        }
        if (n < NT) {  // NT odd: one more tile
            step(b0, accA, accB, max(n - 1, 0));
            epi(accA, n);
        } else {
            epi(accB, n - 1);
        }
        glds_wait<0>();
        __syncthreads();  // every wave is done with the buffers before the next frame restages them
        // merge the two lane halves of each source and write its record
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int o[PK_TOP];
#pragma unroll
            for (int k = 0; k < PK_TOP; ++k) o[k] = __shfl(v[t][k], lane ^ 32, WAVE);
            const int i = src0 + t * 32 + j;
            if (h == 0 && tf[t] == f) {
                const int amax_i = max(v[t][0], o[0]);
                const int thr = amax_i - band;
                int nc = 0, c0 = 0;
                Rec* rr = rec + i;
                const int PWP = pw_pad(g.pw);
#pragma unroll
                for (int k = 0; k < 2 * PK_TOP; ++k) {
                    const int xi = k < PK_TOP ? v[t][k] : o[k - PK_TOP];
                    if (xi >= thr) {
                        const int tag = xi & ((1 << PK_IDX_BITS) - 1);
                        const int r = tag & 15, cb = CB > 1 ? (tag >> 4) & 1 : 0, st = tag >> TSH;
                        const int pc = st * PK_CELLS + cb * 32 + 8 * (r >> 2) + 2 * (r & 3) +
                                       ((k < PK_TOP ? 0 : 1) ^ ((st / tiles_per_row) & 1));
                        const int row = pc / PWP, col = pc - row * PWP;
                        const int cellc = row * g.pw + min(col, g.pw - 1);
                        if (nc < KC) rr->cand[nc] = cellc;
                        if (nc == 0) c0 = cellc;
                        ++nc;
                    }
                }
                // a last list entry inside the band may hide a further one; a maximum inside the band of zero decides nothing
                if (v[t][PK_TOP - 1] >= thr || o[PK_TOP - 1] >= thr || thr <= 0) nc = KC + 1;
                const float amax = (float)(amax_i >> PK_IDX_BITS) * (1.f / (float)(1 << PK_VAL_BITS));
                rr->amax = amax;
                rr->zmax = 0.f;
                rr->Z = -1.f;
                if (done.kstar != nullptr && nc == 1 && amax > 2.f * EPS_C) {
                    rr->ncand = -1;   // finished: rescore_kernel skips it
                    done.kstar[i] = c0;
                    done.snorm[i] = done.rown[row_of[m0 + i]];
                    atomicAdd(&done.hist[(size_t)f * done.HWk + cell_key(c0, g.pw)], 1);
                } else {
                    rr->ncand = nc;
                }
            }
        }
    }
}

// ---- corr_peaks for WIDE features (round 6: C = 768 / 1024, the reference's shipped ViT-L configuration) -----------------------
// At C = 1024 the 64 sources of a wave would need 512 operand registers.  Here the K dimension is split over a PAIR of waves:
//   workgroup = 4 waves = 2 source groups (sg = w >> 1, 64 sources each) x 2 K halves (kh = w & 1, C / 2 channels each);
//   a wave keeps ITS K half of its group's 64 sources in the AGPR half of the register file (2 tiles x KSH k-steps x 4 = 256
//   registers at C = 1024) and reads ITS K half of the cell tile from LDS -- one ds_read_b128 still feeds two MFMAs, as in the
//   C = 384 kernel, and every wave reads only half of the tile;
//   the partial sums meet through LDS: of the two accumulators (source tiles t = 0, 1) a wave GIVES the one it does not own
//   (t = 1 - kh) to its partner and TAKES the partner's partial of the one it owns (t = kh): 4 KB out, 4 KB in per wave and
//   32-cell tile against 32 KB of fragment reads.  The owner then runs the top-6 list program of corr_peaks_kernel on its 32
//   sources x 16 cells per lane -- half the list work per MFMA of the C = 384 kernel (128 VALU instructions per 64 MFMA slots).
// Pipeline per tile n (one wave):  [request tile n+1]  MFMAs of tile n into accN, and between them: partial of tile n-1 out ->
// barrier B -> partner's partial in -> sums -> list updates of tile n-1;  wait for tile n+1 -> barrier A.
// The cell tile (32 cells x C x 2 B = 64 KB at C = 1024) sits in a ring of TWO buffers (128 KB) + 16 KB of exchange slots; the
// LDS image, the DMA requests (8 whole 128-byte lines each), the cell <-> MFMA row checkerboard, the position tags, the candidate
// band and the record format are those of corr_peaks_kernel, so rescore / refine see no difference.
constexpr int PKW_SRC = 128;              // sources per workgroup
constexpr size_t peaks_wide_lds_bytes(int C) { return 2 * (size_t)32 * C * 2 + 4 * 4096 + 64; }

template <int KSH>
__global__ __launch_bounds__(256) void corr_peaks_wide_kernel(dtk_geom g, const half_t* __restrict__ f16,
                                                              const half_t* __restrict__ s16, const int32_t* __restrict__ tgt,
                                                              Rec* __restrict__ rec, int m0, int count, int HWp,
                                                              const int32_t* __restrict__ row_of, PeaksDone done) {
    constexpr int C = 32 * KSH;                  // both K halves
    constexpr int NL = C / 64, NLH = NL / 2;     // 128-byte lines per cell; per K half
    constexpr int TILE_BYTES = 32 * C * 2;
    constexpr int TSH = 4;
    extern __shared__ __attribute__((aligned(1024))) unsigned char cells_dyn[];   // [2][TILE_BYTES] | [4][4096] exchange | s_fr[8]
    unsigned char* cells = cells_dyn;
    unsigned char* xch = cells_dyn + 2 * TILE_BYTES;
    int* s_fr = reinterpret_cast<int*>(cells_dyn + 2 * TILE_BYTES + 4 * 4096);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = w & 1, sg = w >> 1;
    const int j = lane & 31, h = lane >> 5;
    const int src0 = blockIdx.x * PKW_SRC + sg * 64;   // first source of this wave's group (index inside the launch)
    int tf[2];
    {
        int lo = INT_MAX, hi = -1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int i = src0 + t * 32 + j;
            tf[t] = i < count ? min(max(tgt[m0 + i], 0), g.T - 1) : -1;
            if (tf[t] >= 0) { lo = min(lo, tf[t]); hi = max(hi, tf[t]); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o, WAVE));
            hi = max(hi, __shfl_xor(hi, o, WAVE));
        }
        if (lane == 0) { s_fr[w] = lo; s_fr[4 + w] = hi; }
    }
    // the sources: B operand, lane (j, h) holds source j of tile t, channels kh C/2 + 16 ks + 8 h .. + 7 -- straight into AGPRs
    h8 bs[2][KSH];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int i = src0 + t * 32 + j;
        const int ic = min(i, count - 1);
        const half_t* sp = s16 + (size_t)(row_of ? row_of[m0 + ic] : ic) * C + kh * (C / 2) + h * 8;
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(bs[t][ks]) : "v"(sp + ks * 16) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) asm volatile("" : "+a"(bs[t][ks]));
    __syncthreads();
    const int fmin = min(min(s_fr[0], s_fr[1]), min(s_fr[2], s_fr[3]));
    const int fmax = max(max(s_fr[4], s_fr[5]), max(s_fr[6], s_fr[7]));
    const int tiles_per_row = pw_pad(g.pw) / 32;
    const unsigned lds_base = (unsigned)(size_t)cells;
    const int NT = HWp / 32;
    const int band = (int)(EPS_PK * (float)(1 << PK_VAL_BITS)) << PK_IDX_BITS;
    const dtk_u4 srd = dtk_make_srd(f16);
    unsigned voff_even, voff_odd;
    {
        const int pos = 8 * w + (lane >> 3), slot = lane & 7;
        const int jr = (pos & 16) | ((pos >> 1) & 7) | ((pos & 1) << 3);      // the MFMA row whose cell sits at `pos`
        const int cell = 8 * (jr >> 3) + 2 * (jr & 3) + ((jr >> 2) & 1);
        const int piece = slot ^ (jr & 7);
        voff_even = (unsigned)(cell * C * 2 + piece * 16);
        voff_odd = (unsigned)((cell ^ 1) * C * 2 + piece * 16);
    }
    const unsigned dma_dst = __builtin_amdgcn_readfirstlane(lds_base + w * 1024);
    unsigned frag_off[4];
    {
        const int pi = (j & 16) | ((j & 7) << 1) | ((j >> 3) & 1);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) frag_off[k4] = (unsigned)(kh * NLH * 4096 + pi * 128 + (((2 * k4 + h) ^ (j & 7)) << 4));
    }
    // exchange slots: [wave][quad of accumulator registers][lane] x 16 B; both waves of a pair map lanes to (source, cells) alike
    unsigned char* x_mine = xch + w * 4096 + lane * 16;
    const unsigned char* x_partner = xch + (w ^ 1) * 4096 + lane * 16;
    for (int f = fmin; f <= fmax; ++f) {
        unsigned next_off = __builtin_amdgcn_readfirstlane((unsigned)f * (unsigned)HWp * (unsigned)(C * 2));   // (volume < 4 GB: host check)
        int next_in_row = 0, next_odd = 0;
        auto issue = [&](int buf) {   // the tiles are requested in order: 0, 1, 2, ...
            const unsigned voff = next_odd ? voff_odd : voff_even;
            const unsigned dst = __builtin_amdgcn_readfirstlane(dma_dst + (unsigned)buf * (unsigned)TILE_BYTES);
            peaks_dma<0, NL, NL, C>(srd, next_off, voff, dst);
            next_off += (unsigned)(32 * C * 2);
            if (++next_in_row == tiles_per_row) { next_in_row = 0; next_odd ^= 1; }
        };
        int v[PK_TOP];
#pragma unroll
        for (int k = 0; k < PK_TOP; ++k) v[k] = 0;
        // partial of the tile this wave does NOT own -> its exchange slot
        auto give = [&](const f16v (&acc)[2]) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<f4*>(x_mine + i * 1024) = f4{acc[1 - kh][4 * i], acc[1 - kh][4 * i + 1], acc[1 - kh][4 * i + 2], acc[1 - kh][4 * i + 3]};
        };
        // one step: the MFMAs of the tile in `buf` into accN; between them the exchange and the list updates of the previous tile
        // (accP, step index np).  `more`: a further tile exists and is requested into the other buffer first.
        auto step = [&](int buf, f16v (&accN)[2], const f16v (&accP)[2], bool have_prev, int np, bool more) {
            if (more) issue(buf ^ 1);
            const unsigned char* base = cells + (size_t)buf * TILE_BYTES;
            auto frag = [&](int ks) { return *reinterpret_cast<const h8*>(base + frag_off[ks & 3] + (ks >> 2) * 4096); };
            const int ib = np << TSH;
            const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int key = 0;
            f4 xr[4];
            float val[16];
            h8 a[3];
            a[0] = frag(0);
            a[1] = frag(1);
            constexpr int NSLOT = 2 * KSH, L0 = 8, NINS = 16 * 8;   // list instructions run in slots L0 .. NSLOT-1
            peaks_static_for<0, NSLOT>([&](auto qc) {
                constexpr int q = decltype(qc)::value, ks = q / 2, t = q % 2;
                if (t == 0 && ks + 2 < KSH) a[(ks + 2) % 3] = frag(ks + 2);
                accN[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks % 3], bs[t][ks], ks == 0 ? zero16 : accN[t], 0, 0, 0);
                if (have_prev) {
                    if constexpr (q == 1) give(accP);
                    if constexpr (q == 2) __syncthreads();   // barrier B: every partial of the previous tile is in its slot
                    if constexpr (q == 3) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) xr[i] = *reinterpret_cast<const f4*>(x_partner + i * 1024);
                    }
                    if constexpr (q == 6) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) val[r] = accP[kh][r] + xr[r >> 2][r & 3];
                    }
                    if constexpr (q >= L0) {
                        peaks_static_for<(q - L0) * NINS / (NSLOT - L0), (q - L0 + 1) * NINS / (NSLOT - L0)>([&](auto ic) {
                            constexpr int i = decltype(ic)::value, e = i >> 3;
                            top_push_op<(i & 7)>(v, val[e], ib | e, key);
                        });
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            glds_wait<0>();      // the next tile has landed (this wave's requests; the barrier covers the others')
            __syncthreads();     // barrier A: the tile is published, `buf` and the exchange slots are free
        };
        // the last tile's partials: no MFMAs to hide under
        auto finish = [&](const f16v (&acc)[2], int np) {
            give(acc);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f4 x = *reinterpret_cast<const f4*>(x_partner + i * 1024);
#pragma unroll
                for (int c = 0; c < 4; ++c) top_push_value(v, acc[kh][4 * i + c] + x[c], (np << TSH) | (4 * i + c));
            }
        };
        f16v accA[2], accB[2];
        issue(0);
        glds_wait<0>();
        __syncthreads();
        int n = 0;
        for (; n + 1 < NT; n += 2) {
            step(0, accA, accB, n > 0, n - 1, true);
            step(1, accB, accA, true, n, n + 2 < NT);
        }
        if (n < NT) {   // NT odd: one more tile
            step(0, accA, accB, n > 0, n - 1, false);
            finish(accA, n);
        } else {
            finish(accB, n - 1);
        }
        __syncthreads();   // every wave is done with the buffers and the slots before the next frame restages them
        // merge the two lane halves of each source of the OWNED tile and write its record (as corr_peaks_kernel)
        {
            const int t = kh;
            int o[PK_TOP];
#pragma unroll
            for (int k = 0; k < PK_TOP; ++k) o[k] = __shfl(v[k], lane ^ 32, WAVE);
            const int i = src0 + t * 32 + j;
            if (h == 0 && tf[t] == f) {
                const int amax_i = max(v[0], o[0]);
                const int thr = amax_i - band;
                int nc = 0, c0 = 0;
                Rec* rr = rec + i;
                const int PWP = pw_pad(g.pw);
#pragma unroll
                for (int k = 0; k < 2 * PK_TOP; ++k) {
                    const int xi = k < PK_TOP ? v[k] : o[k - PK_TOP];
                    if (xi >= thr) {
                        const int tag = xi & ((1 << PK_IDX_BITS) - 1);
                        const int r = tag & 15, st = tag >> TSH;
                        const int pc = st * 32 + 8 * (r >> 2) + 2 * (r & 3) + ((k < PK_TOP ? 0 : 1) ^ ((st / tiles_per_row) & 1));
                        const int row = pc / PWP, col = pc - row * PWP;
                        const int cellc = row * g.pw + min(col, g.pw - 1);
                        if (nc < KC) rr->cand[nc] = cellc;
                        if (nc == 0) c0 = cellc;
                        ++nc;
                    }
                }
                if (v[PK_TOP - 1] >= thr || o[PK_TOP - 1] >= thr || thr <= 0) nc = KC + 1;
                const float amax = (float)(amax_i >> PK_IDX_BITS) * (1.f / (float)(1 << PK_VAL_BITS));
                rr->amax = amax;
                rr->zmax = 0.f;
                rr->Z = -1.f;
                if (done.kstar != nullptr && nc == 1 && amax > 2.f * EPS_C) {
                    rr->ncand = -1;   // finished: rescore_kernel skips it
                    done.kstar[i] = c0;
                    done.snorm[i] = done.rown[row_of[m0 + i]];
                    atomicAdd(&done.hist[(size_t)f * done.HWk + cell_key(c0, g.pw)], 1);
                } else {
                    rr->ncand = nc;
                }
            }
        }
    }
}

// head16_kernel: one workgroup per map.  The map sits in LDS as fp16 with a zero border; besides the approximate maximum
// and the candidate cells, the whole refiner runs on the matrix cores without ever staging the hidden activations:
//   conv1 as GEMM1 (MFMA 16x16x16 f16):  h^T[16 ch][16 px] = W1[16 ch][K = 3 rows x (3 taps + pad) | bias] . X[K][16 px]
//   D of GEMM1 (lane (g, j): channels 4g..4g+3 of pixel j) IS the B fragment of
//   conv2 as GEMM2:  P[(dy, dx)][16 px] = W2[taps][16 ch] . relu(h)^T   (per-pixel 16 -> 9 projection)
//   z[row r+dy'][col c] = sum_dx P[(dy, dx)][c + dx]: the dx sum is two DPP row shifts, the dy sum is carried in one
//   accumulator per 16-lane group: the group <-> output-row assignment rotates with the row (three static A2 variants),
//   so each lane group finishes one output row every third step and no data ever moves between groups.
// Work item = (14-column segment, quarter of the rows); 36 items over the 4 waves.
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

template <int DPP>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), DPP, 0xF, 0xF, true));
}

struct HeadLane {  // per-lane constants of the chained GEMMs
    h4 a1;         // W1 fragment
    h4 a2[3];      // W2 fragment per phase
    float keep[3]; // 0 where this lane's group starts a new output row at that phase, else 1
    int done[3];   // 1 where this lane's group completes an output row at that phase
};

// OFF: byte offset of the segment relative to xp (the second segment of a pair sits 14 columns = 28 bytes to the right);
// `edge`: the segment touches a map border, so hidden activations of out-of-map columns must be zeroed (hmask)
template <int OFF>
__device__ __forceinline__ void head_step(const HeadLane& L, int v, bool row_ok, bool can_complete, bool edge,
                                          const half_t* xp, unsigned hmask, int done_v, float& acc, float& zst) {
    float V = 0.f;
    if (row_ok) {
        // B1: k = 4g + i: rows r'-1+g (g < 3), taps dx = i-1 (i < 3); g = 3: the constant (1,0,0,0) for the bias.
        // Three 2-byte LDS reads straight into packed halves (the address is only 2-byte aligned: a merged ds_read_b32
        // would be a misaligned access); loads and their wait live in one asm statement (the compiler does not count
        // asm loads)
        unsigned xlo, xhi;
        asm volatile(
            "ds_read_u16_d16 %0, %2 offset:%3\n\t"
            "ds_read_u16_d16_hi %0, %2 offset:%4\n\t"
            "ds_read_u16 %1, %2 offset:%5\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(xlo), "=&v"(xhi)
            : "v"((unsigned)(size_t)xp), "i"(OFF), "i"(OFF + 2), "i"(OFF + 4)
            : "memory");
        const h2 x01 = __builtin_bit_cast(h2, xlo), x2 = __builtin_bit_cast(h2, xhi);
        const h4 b1 = {x01[0], x01[1], x2[0], x2[1]};
        f4 d1 = {0.f, 0.f, 0.f, 0.f};
        d1 = __builtin_amdgcn_mfma_f32_16x16x16f16(L.a1, b1, d1, 0, 0, 0);
        // fp16 (round-to-zero pack never overflows to inf) + packed relu; hidden activations outside the map are zero
        // (conv2's zero padding)
        const h2 zero2 = {(half_t)0.f, (half_t)0.f};
        h2 lo = __builtin_elementwise_max(__builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(d1[0], d1[1])), zero2);
        h2 hi = __builtin_elementwise_max(__builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(d1[2], d1[3])), zero2);
        if (edge) {
            lo = __builtin_bit_cast(h2, __builtin_bit_cast(unsigned, lo) & hmask);
            hi = __builtin_bit_cast(h2, __builtin_bit_cast(unsigned, hi) & hmask);
        }
        const h4 b2 = {lo[0], lo[1], hi[0], hi[1]};
        f4 d2 = {0.f, 0.f, 0.f, 0.f};
        d2 = __builtin_amdgcn_mfma_f32_16x16x16f16(L.a2[v], b2, d2, 0, 0, 0);
        // sum over dx: P[dx=-1] of the pixel to the left, P[dx=0] here, P[dx=+1] of the pixel to the right
        V = dpp_f<0x111>(d2[0]) + d2[1] + dpp_f<0x101>(d2[2]);
    }
    acc = fmaf(acc, L.keep[v], V);
    if (can_complete) zst = done_v ? acc : zst;
}

__global__ __launch_bounds__(256) void head16_kernel(dtk_geom g, const float* __restrict__ head,
                                                     const uint32_t* __restrict__ wpk,
                                                     const half_t* __restrict__ maps, int HWp, Rec* __restrict__ rec,
                                                     int m0, int count, int M, const int32_t* __restrict__ dM, int dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int ph = g.ph, pw = g.pw;
    const int xw = map_xw(pw);                              // 8 zero columns left, >= 8 right
    const int xs_elems = (ph + 2) * xw + 32;                // + slack: edge segments read a little past a row
    half_t* xs = reinterpret_cast<half_t*>(smem_raw);       // [(ph+2)][(pw+4)], row -1 and row ph are zero
    half_t* cst = xs + ((xs_elems + 7) & ~7);               // {1,0,0,..} at +0 and +14: the bias column of B1
    float* red = reinterpret_cast<float*>(cst + 32);        // 16 floats
    int* s_cnt = reinterpret_cast<int*>(red + 16);
    int* s_cand = s_cnt + 1;                                // KC ints
    const int i = blockIdx.x;
    const int m = m0 + i;
    if (i >= count || m >= dtk_active(M, dM)) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // the map arrives in the padded layout (HWp here is its pitch): a straight 16-byte copy into LDS
    const int n16 = ((xs_elems + 7) & ~7) / 8;
    const uint4* map16 = reinterpret_cast<const uint4*>(maps + (size_t)i * HWp);
    uint4* xs16 = reinterpret_cast<uint4*>(xs);
    if (tid < 32) cst[tid] = (half_t)((tid == 0 || tid == 14) ? 1.f : 0.f);
    if (tid == 0) *s_cnt = 0;
    h2 mx2 = {(half_t)0.f, (half_t)0.f};
    for (int c = tid; c < n16; c += 256) {
        const uint4 v = map16[c];
        xs16[c] = v;
        mx2 = __builtin_elementwise_max(mx2, __builtin_elementwise_max(
                  __builtin_elementwise_max(as_h2(v.x), as_h2(v.y)), __builtin_elementwise_max(as_h2(v.z), as_h2(v.w))));
    }
    float amax = fmaxf((float)mx2[0], (float)mx2[1]);
    amax = wave_max(amax);
    if (lane == 0) red[w] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float thr = amax - EPS_C;
    const half_t thr_h = (half_t)fmaxf(thr - 1e-3f, -1.f);  // coarse fp16 pre-filter, exact test below
    for (int c = tid; c < (DTK_DBG(dbg, 64) ? 0 : n16); c += 256) {
        const uint4 v = xs16[c];
        const h2 m4 = __builtin_elementwise_max(__builtin_elementwise_max(as_h2(v.x), as_h2(v.y)),
                                                __builtin_elementwise_max(as_h2(v.z), as_h2(v.w)));
        if (m4[0] < thr_h && m4[1] < thr_h) continue;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int pidx = c * 8 + e;
            const int pr = pidx / xw, pc = pidx - pr * xw;
            if (pr < 1 || pr > ph || pc < 8 || pc >= pw + 8) continue;  // border zeros are not cells
            if ((float)xs[pidx] >= thr) {
                const int slot = atomicAdd(s_cnt, 1);
                if (slot < KC) s_cand[slot] = (pr - 1) * pw + (pc - 8);
            }
        }
    }

    // ---- per-lane constants ----
    const int lg = lane >> 4, lj = lane & 15;
    constexpr float LOG2E = 1.4426950408889634f;
    HeadLane L;
    {
        // A1[m = ch lj][k = 4*lg + i]
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        if (lg < 3) {
#pragma unroll
            for (int q = 0; q < 3; ++q) a[q] = head[lj * 9 + lg * 3 + q];  // tap (dy = lg-1, dx = q-1)
        } else {
            a[0] = head[144 + lj];
        }
        L.a1 = h4{(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3]};
        // A2_v[m = lj][k = ch 4*lg + i]: m = 4*gm + t <-> (dy of group gm at phase v, dx = t-1)
        const int gm = lj >> 2, t = lj & 3;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            float b[4] = {0.f, 0.f, 0.f, 0.f};
            if (gm < 3 && t < 3) {
                const int dy = (gm == (v + 1) % 3) ? -1 : ((gm == v) ? 0 : 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) b[q] = LOG2E * head[160 + (4 * lg + q) * 9 + (dy + 1) * 3 + t];
            }
            L.a2[v] = h4{(half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
            L.keep[v] = (lg == (v + 1) % 3) ? 0.f : 1.f;
            L.done[v] = (lg == (v + 2) % 3) ? 1 : 0;
        }
    }
    const float LOG2E_ = 1.4426950408889634f;
    const float b2 = LOG2E_ * head[304];   // logits are carried in log2 units: exp2 is one instruction
    const int nq = (pw + 13) / 14;
    float rm = -1e30f, rs = 0.f;  // this lane's running (max, sum of exp) over the logits it completed
    __syncthreads();
    const int wu = __builtin_amdgcn_readfirstlane(w);  // provably wave-uniform: the row loop below becomes scalar control flow
    // work item = (pair of 14-column segments, quarter of the rows); the two segments of a pair are advanced in lock-step
    const int npair = (nq + 1) >> 1;
    for (int item = wu; item < (DTK_DBG(dbg, 32) ? 0 : npair * 4); item += 4) {
        const int qa = (item >> 2) * 2, qb = qa + 1, rq = item & 3;
        const bool has_b = qb < nq;  // wave-uniform
        const bool edge_a = qa == 0 || qa == nq - 1, edge_b = qb == nq - 1;
        const int ra = (ph * rq) >> 2, rb = (ph * (rq + 1)) >> 2;
        const int cja = 14 * qa - 1 + lj, cjb = cja + 14;             // this lane's pixel column in either segment
        const unsigned hma = (cja >= 0 && cja < pw) ? 0xFFFFFFFFu : 0u, hmb = (cjb < pw) ? 0xFFFFFFFFu : 0u;
        const bool zoka = lg < 3 && lj >= 1 && lj <= 14 && cja < pw, zokb = has_b && lg < 3 && lj >= 1 && lj <= 14 && cjb < pw;
        int dna[3], dnb[3];  // completion masks: only valid output pixels ever reach the statistics
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            dna[v] = (L.done[v] && zoka) ? 1 : 0;
            dnb[v] = (L.done[v] && zokb) ? 1 : 0;
        }
        // B1 source: lanes of group g < 3 read row r'-1+g at columns cj-1..cj+1 (segment b: +28 bytes); group 3 reads
        // the constant
        const half_t* xp = (lg < 3) ? xs + (ra - 1 - 1 + lg + 1) * xw + (cja - 1 + 8) : cst;
        const int xstep = (lg < 3) ? xw : 0;
        float acca = 0.f, zsta = -1e30f, accb = 0.f, zstb = -1e30f;
        const int nsteps = rb - ra + 2;
        for (int u0 = 0; u0 < nsteps; u0 += 3) {
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int u = u0 + v;
                if (u < nsteps) {
                    const int r = ra - 1 + u;
                    head_step<0>(L, v, r >= 0 && r < ph, u >= 2, edge_a, xp, hma, dna[v], acca, zsta);
                    if (has_b) head_step<28>(L, v, r >= 0 && r < ph, u >= 2, edge_b, xp, hmb, dnb[v], accb, zstb);
                    xp += xstep;
                }
            }
            // fold the output rows completed in this round into the running statistics (log2 domain, deferred maximum:
            // the running reference rm only moves when a logit exceeds it by more than 8, so the common path is
            // two subtractions, two exp2 and two adds; an untouched zst of -1e30 contributes exp2(-inf) = 0)
            const float za = zsta + b2, zb_ = zstb + b2;
            const float zmx = fmaxf(za, zb_);
            if (__any(zmx > rm + 8.f)) {
                asm volatile("; rescale (rare)" ::: "memory");  // keeps this a real branch instead of selects
                const float mn = fmaxf(rm, zmx);
                rs *= __builtin_amdgcn_exp2f(rm - mn);
                rm = mn;
            }
            rs += __builtin_amdgcn_exp2f(za - rm) + __builtin_amdgcn_exp2f(zb_ - rm);  // raw v_exp_f32: exp2(-huge) = 0
            zsta = -1e30f;
            zstb = -1e30f;
        }
    }
    // back to natural units: zmax = rm * ln 2; the sum is unit-free
    rm = (rm > -1e29f) ? rm * 0.6931471805599453f : rm;
    // merge the per-lane (max, sum) pairs
    float zm = wave_max(rm);
    float zs = rs * expf(rm - zm);
    zs = wave_sum(zs);
    __syncthreads();
    if (lane == 0) { red[w] = zm; red[4 + w] = zs; }
    __syncthreads();
    if (tid == 0) {
        const float zmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float Z = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) Z += red[4 + k] * expf(red[k] - zmax);
        Rec r;
        r.amax = amax;
        r.ncand = *s_cnt;
#pragma unroll
        for (int k = 0; k < KC; ++k) r.cand[k] = (k < r.ncand) ? s_cand[k] : 0;
        r.zmax = zmax;
        r.Z = Z;
        rec[i] = r;
    }
}

// ---- refine: exact fp32 finish ------------------------------------------------------------------------------------
// refine_corr_kernel (16 consecutive sources per workgroup): candidates re-scored in fp32 -> exact argmax k*;
//   fp32 correlation of every source with the union box of the tile's (2*RD+5)^2 windows on the f32-input MFMA
//   (16x16x4: bit-exact fmaf chains), scattered into per-source window buffers xwin[m][WX*WX] in global memory.
//   No LDS tile: occupancy is register-limited and four N-tiles are in flight per wave.
// refine_head_kernel (one wave per source): fp32 refiner on the window, disk soft-argmax, output or redo.
constexpr int WX = 2 * RD + 5;  // x window side (15)
constexpr int WH = 2 * RD + 3;  // hidden window side (13)
constexpr int WZ = 2 * RD + 1;  // logit window side (11)


struct Redo {
    int32_t* count;
    int32_t* src_row;
    int32_t* tgt;
    int32_t* out_idx;
};

// spatial sort key of an argmax cell: 8-row bands, column-major inside a band, so that sources that are consecutive in
// key order have nearly coincident windows
__device__ __forceinline__ int cell_key(int cell, int pw) {
    const int r = cell / pw, c = cell - r * pw;
    return (r >> 3) * (8 * pw) + c * 8 + (r & 7);
}

__device__ unsigned long long g_dbg[4];  // counters of DTK_DEV builds (DTK_DEBUG & 16: refine_corr boxes; & 4096: redo reasons)

// one wave per source: |s|, exact fp32 re-scoring of the candidates -> k*, histogram of (frame, cell key)
__global__ __launch_bounds__(256) void rescore_kernel(dtk_geom g, const float* __restrict__ feat,
                                                      const float* __restrict__ norms, const float* __restrict__ emb,
                                                      const int32_t* __restrict__ src_row, const int32_t* __restrict__ tgt,
                                                      const int32_t* __restrict__ out_idx, const Rec* __restrict__ rec,
                                                      int32_t* __restrict__ kstar, float* __restrict__ snorm,
                                                      int32_t* __restrict__ hist, int HWk, Redo redo, int m0, int count,
                                                      int M, const int32_t* __restrict__ dM, int dbg,
                                                      int32_t* __restrict__ arg_cell, float* __restrict__ arg_cos,
                                                      const float* __restrict__ rown) {
    const int HW = g.ph * g.pw, C = g.C;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int m = m0 + i;
    if (i >= count) return;
    if (m >= dtk_active(M, dM)) {
        if (lane == 0) kstar[i] = -1;
        return;
    }
    if (rec[i].ncand == -1) return;   // finished by corr_peaks' epilogue (PeaksDone): k*, |s| and the histogram are already written
    const int row = src_row ? src_row[m] : m;
    const int f = min(max(tgt[m], 0), g.T - 1);
    const float* sp = emb + (size_t)row * C;
    // |source|: from the row table's norms when the sources are rows of one (round 5: the anchor stage tracks 92 k distinct rows
    // into 90 frames each -- 8.3 M wave-wide sums of the same 92 k rows, and for a single-candidate source the only reason to touch the
    // row at all), else summed here; both forms add the same terms in the same order
    float sn;
    if (rown) {
        sn = rown[row];
    } else {
        float ss = 0.f;
        for (int k = lane * 4; k < C; k += 256) {
            const float4 v = *reinterpret_cast<const float4*>(sp + k);
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        sn = sqrtf(wave_sum(ss));
    }
    const Rec rc = rec[i];
    bool redo_it = rc.ncand > KC || rc.ncand < 1;
    float best = -1.f;
    int bi = INT_MAX;
    // One candidate whose approximate maximum is clearly positive IS the answer (round 5): the band proof puts the exact arg-max among
    // the candidates, and amax - EPS_C > 0 makes the exact maximum positive -- the row of that cell need not be fetched at all (one
    // dependent trip to memory less for the majority of the sources; the kernel is latency-bound).  dtk_argmax_cells wants the exact
    // cosine and takes the full path.
    const bool single = !redo_it && rc.ncand == 1 && rc.amax > 2.f * EPS_C && arg_cell == nullptr && !DTK_DBG(dbg, 2);
    if (single) {
        bi = min(max(rc.cand[0], 0), HW - 1);
        best = rc.amax;
    } else if (!redo_it) {
        const int ncd = DTK_DBG(dbg, 2) ? 1 : rc.ncand;
        for (int k = 0; k < ncd; ++k) {
            const int cell = min(max(rc.cand[k], 0), HW - 1);
            const float* fp = feat + ((size_t)f * HW + cell) * C;
            float d = 0.f;
            for (int c = lane * 4; c < C; c += 256) {
                const float4 a = *reinterpret_cast<const float4*>(sp + c), b = *reinterpret_cast<const float4*>(fp + c);
                d += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
            }
            d = wave_sum(d);
            const float v = fmaxf(d / fmaxf(sn * norms[(size_t)f * HW + cell], 1e-8f), 0.f);
            if (v > best || (v == best && cell < bi)) { best = v; bi = cell; }
        }
        // a non-positive exact maximum means the relu'd map may be all-zero (argmax 0): let the exact path decide
        if (!(best > 0.f)) redo_it = true;
    }
    if (DTK_DBG(dbg, 4096) && lane == 0 && redo_it)
        atomicAdd(&g_dbg[rc.ncand > KC ? 0 : (rc.ncand < 1 ? 1 : 2)], 1ULL);
    if (lane == 0) {
        snorm[i] = sn;
        if (redo_it) {
            const int slot = atomicAdd(redo.count, 1);
            redo.src_row[slot] = row;
            redo.tgt[slot] = f;
            redo.out_idx[slot] = out_idx ? out_idx[m] : m;
            kstar[i] = -1;
        } else if (arg_cell) {  // dtk_argmax_cells: the exact arg-max cell and its cosine are the result
            const int oi = out_idx ? out_idx[m] : m;
            arg_cell[oi] = bi;
            arg_cos[oi] = best;
        } else {
            kstar[i] = bi;
            atomicAdd(&hist[(size_t)f * HWk + cell_key(bi, g.pw)], 1);
        }
    }
}

// ---- exclusive scan of the key histogram (n up to T * HWk): block sums -> scan of block sums -> final -------------
constexpr int SCAN_PER_BLOCK = 1024;
__global__ __launch_bounds__(256) void scan_blocksum_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ bsum, int n) {
    __shared__ int red[4];
    const int base = blockIdx.x * SCAN_PER_BLOCK;
    int s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = base + j * 256 + threadIdx.x;
        s += idx < n ? in[idx] : 0;
    }
    s = wave_sum_i(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ int block_incl_scan(int v, int* lds4, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, WAVE);
        if (lane >= o) x += y;
    }
    __syncthreads();
    if (lane == 63) lds4[w] = x;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; ++k) base += lds4[k];
    *total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    return x + base;
}
__global__ __launch_bounds__(256) void scan_top_kernel(int32_t* __restrict__ bsum, int nb, int32_t* __restrict__ total_out) {
    __shared__ int lds4[4];
    int carry = 0;
    for (int i0 = 0; i0 < nb; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const int v = i < nb ? bsum[i] : 0;
        int total;
        const int incl = block_incl_scan(v, lds4, &total);
        if (i < nb) bsum[i] = carry + incl - v;
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = carry;
}
__global__ __launch_bounds__(256) void scan_final_kernel(const int32_t* __restrict__ in, const int32_t* __restrict__ boff,
                                                         int32_t* __restrict__ out, int n) {
    __shared__ int lds4[4];
    const int base = blockIdx.x * SCAN_PER_BLOCK;
    int carry = boff[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = base + j * 256 + threadIdx.x;
        const int v = idx < n ? in[idx] : 0;
        int total;
        const int incl = block_incl_scan(v, lds4, &total);
        if (idx < n) out[idx] = carry + incl - v;
        carry += total;
    }
}
// perm[offset[key] + arrival] = source; arrival order inside a key is irrelevant (results do not depend on tile mates)
__global__ __launch_bounds__(256) void scatter_kernel(dtk_geom g, const int32_t* __restrict__ tgt,
                                                      const int32_t* __restrict__ kstar, const int32_t* __restrict__ off,
                                                      int32_t* __restrict__ cursor, int32_t* __restrict__ perm, int HWk,
                                                      int m0, int count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const int k = kstar[i];
    if (k < 0) return;
    const int f = min(max(tgt[m0 + i], 0), g.T - 1);
    const size_t key = (size_t)f * HWk + cell_key(k, g.pw);
    perm[off[key] + atomicAdd(&cursor[key], 1)] = i;
}

// Window correlations, fp32-grade on the fp16 matrix cores.
//   * A workgroup takes RC_SRC = 64 key-consecutive sources (after the counting sort they sit on the same or neighbouring
//     arg-max cells of one frame) and correlates them with the cells of their windows' union box: one pass over the box's
//     feature rows (1.5 KB per cell at C = 384) serves up to 64 windows.  Round 1 took 16 sources per workgroup, moved 9 KB
//     of features per source through the fabric and sat at 3.9 TB/s (PMC) -- memory-bound.
//   * Workgroup b runs on XCD b % 8; every XCD gets a contiguous range of the key-sorted tiles, so that the overlapping
//     boxes of neighbouring tiles are re-read from that XCD's L2.
//   * The arithmetic: each fp32 operand is split while it is staged into LDS, x = hi + lo (both fp16, 2^5 x so that the
//     lo halves stay normal), and a product is hi.hi + hi.lo + lo.hi on MFMA 16x16x32 f16 with fp32 accumulation -- the
//     scheme of delta_dino.hip: error <= 2^-22 relative per operand (tests/test_numeric_claims.py), three instructions at
//     16x the f32-input MFMA rate instead of one (64 sources x 256 cells x 384 channels on the f32 MFMA alone cost
//     more than the whole round-1 kernel).
constexpr int RC_SRC = 64;
constexpr int RCD_NW = 4, RCD_NS = 4, RCD_NBM = NB_MAX;   // refine_corr_dma: waves (= M tiles) per workgroup, ring depth, largest group box.
// Measured against this form (10.1 ms per step): a 6-deep ring with 384-cell groups 15.1 ms; 8 waves / 128 sources with a 12-deep ring 19.5 ms
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void refine_corr_kernel(dtk_geom g, const float* __restrict__ feat,
                                                          const float* __restrict__ norms,
                                                          const float* __restrict__ emb,
                                                          const int32_t* __restrict__ src_row,
                                                          const int32_t* __restrict__ tgt,
                                                          const int32_t* __restrict__ kstar,
                                                          const float* __restrict__ snorm,
                                                          const int32_t* __restrict__ perm,
                                                          const int32_t* __restrict__ nvalid,
                                                          float* __restrict__ xwin, int m0, int ntiles, int dbg,
                                                          const float* __restrict__ rc_scale) {
    const float RC_SCALE = *rc_scale;
    __shared__ float s_sn[RC_SRC];
    __shared__ int s_row[RC_SRC], s_f[RC_SRC], s_k[RC_SRC], s_m[RC_SRC], s_grp[RC_SRC], s_box[RC_SRC * 4], s_first[RC_SRC],
        s_last[RC_SRC], s_ng;
    const int ph = g.ph, pw = g.pw, HW = ph * pw, C = g.C;
    const int nv = *nvalid;
    const int per = (ntiles + 7) / 8;
    const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int t0 = tile * RC_SRC;
    if ((int)(blockIdx.x >> 3) >= per || tile >= ntiles || t0 >= nv) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < RC_SRC) {
        const bool ok = t0 + tid < nv;
        const int i = perm[ok ? t0 + tid : t0];  // index inside the round
        const int m = m0 + i;
        s_m[tid] = m;
        s_row[tid] = src_row ? src_row[m] : m;
        s_f[tid] = ok ? min(max(tgt[m], 0), g.T - 1) : -1;
        s_k[tid] = kstar[i];
        s_sn[tid] = snorm[i];
    }
    __syncthreads();

    // greedy grouping: consecutive (key-sorted) sources with the same frame whose window union stays small
    if (tid == 0) {
        int ng = 0, cf = -2, r0 = 0, r1 = 0, c0 = 0, c1 = 0;
        for (int s = 0; s < RC_SRC; ++s) {
            s_grp[s] = -1;
            if (s_f[s] < 0) continue;
            const int kr = s_k[s] / pw, kc = s_k[s] % pw;
            const int a0 = max(kr - (RD + 2), 0), a1 = min(kr + (RD + 2), ph - 1);
            const int b0 = max(kc - (RD + 2), 0), b1 = min(kc + (RD + 2), pw - 1);
            bool fits = false;
            if (ng > 0 && s_f[s] == cf) {
                const int n0 = min(r0, a0), n1 = max(r1, a1), e0 = min(c0, b0), e1 = max(c1, b1);
                if ((n1 - n0 + 1) * (e1 - e0 + 1) <= NB_MAX) { r0 = n0; r1 = n1; c0 = e0; c1 = e1; fits = true; }
            }
            if (!fits) { ++ng; cf = s_f[s]; r0 = a0; r1 = a1; c0 = b0; c1 = b1; s_first[ng - 1] = s; }
            s_grp[s] = ng - 1;
            s_last[ng - 1] = s;
            s_box[(ng - 1) * 4 + 0] = r0; s_box[(ng - 1) * 4 + 1] = r1;
            s_box[(ng - 1) * 4 + 2] = c0; s_box[(ng - 1) * 4 + 3] = c1;
        }
        s_ng = ng;
    }
    __syncthreads();
    const int ng = s_ng;
    if (DTK_DBG(dbg, 16) && tid == 0) {
        atomicAdd(&g_dbg[0], 1ULL);
        atomicAdd(&g_dbg[1], (unsigned long long)ng);
        for (int gi = 0; gi < ng; ++gi) {
            const int ncl = (s_box[gi * 4 + 1] - s_box[gi * 4] + 1) * (s_box[gi * 4 + 3] - s_box[gi * 4 + 2] + 1);
            atomicAdd(&g_dbg[2], (unsigned long long)ncl);
            atomicAdd(&g_dbg[3], (unsigned long long)((ncl + 63) / 64));
        }
    }

    // ---- correlation of the group's sources with the cells of its union box ------------------------------------------
    // LDS: hi and lo planes of A (64 sources = four 16-row M tiles) and B (64 cells per block, one 16-cell N tile per wave),
    // 32-channel K chunks, double-buffered.  Row pitch 40 halves (80 B): a fragment read (lane (fg, fj): 8 halves at
    // [row fj][8 fg]) is a conflict-free ds_read_b128.  A B fragment pair feeds up to 12 MFMAs (4 M tiles x 3 products).
    constexpr int RP = 40;
    __shared__ __attribute__((aligned(16))) half_t Ah[2][RC_SRC * RP], Al[2][RC_SRC * RP], Bh[2][64 * RP], Bl[2][64 * RP];
    const int fj = lane & 15, fg = lane >> 4;
    const int lrow = tid >> 3, lk4 = (tid & 7) * 4;  // loader: rows lrow, lrow + 32 (cells or sources), 4-float piece
    const float* arow0 = emb + (size_t)s_row[lrow] * C + lk4;
    const float* arow1 = emb + (size_t)s_row[lrow + 32] * C + lk4;
    const int nkc = C / 32;
    for (int gi = 0; gi < (DTK_DBG(dbg, 1) ? 0 : ng); ++gi) {
        const int rmin = s_box[gi * 4], rmax = s_box[gi * 4 + 1], cmin = s_box[gi * 4 + 2], cmax = s_box[gi * 4 + 3];
        const int nc = cmax - cmin + 1, ncells = (rmax - rmin + 1) * nc;
        const int mt0 = s_first[gi] >> 4, mt1 = s_last[gi] >> 4;  // M tiles that hold members (members are consecutive)
        const int gf = s_f[s_first[gi]];
        const float* fbase = feat + (size_t)gf * HW * C;
        for (int blk = 0; blk < ncells; blk += 64) {
            // loader cells of this thread (clamped: results of padded cells are never stored)
            const int l0 = min(blk + lrow, ncells - 1), l1 = min(blk + lrow + 32, ncells - 1);
            const float* b0 = fbase + (size_t)((rmin + l0 / nc) * pw + cmin + l0 % nc) * C + lk4;
            const float* b1 = fbase + (size_t)((rmin + l1 / nc) * pw + cmin + l1 % nc) * C + lk4;
            float4 ra0, ra1, rb0, rb1;
            ra0 = *reinterpret_cast<const float4*>(arow0);
            ra1 = *reinterpret_cast<const float4*>(arow1);
            rb0 = *reinterpret_cast<const float4*>(b0);
            rb1 = *reinterpret_cast<const float4*>(b1);
            __syncthreads();  // previous block / group finished reading LDS
            // split x (scaled) into fp16 hi + lo and store 8 bytes into each plane
            auto split_store = [&](half_t* hp, half_t* lp, int row, const float4& v) {
                const float x[4] = {v.x * RC_SCALE, v.y * RC_SCALE, v.z * RC_SCALE, v.w * RC_SCALE};
                h4v hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hi[e] = (half_t)x[e];
                    lo[e] = (half_t)(x[e] - (float)hi[e]);
                }
                *reinterpret_cast<h4v*>(hp + row * RP + lk4) = hi;
                *reinterpret_cast<h4v*>(lp + row * RP + lk4) = lo;
            };
#define RC_STORE(buf)                                        \
    do {                                                     \
        split_store(Ah[buf], Al[buf], lrow, ra0);            \
        split_store(Ah[buf], Al[buf], lrow + 32, ra1);       \
        split_store(Bh[buf], Bl[buf], lrow, rb0);            \
        split_store(Bh[buf], Bl[buf], lrow + 32, rb1);       \
    } while (0)
            RC_STORE(0);
            __syncthreads();
            f4 acc[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = f4{0.f, 0.f, 0.f, 0.f};
            int cur = 0;
            for (int kc = 0; kc < nkc; ++kc) {
                if (kc + 1 < nkc && !DTK_DBG(dbg, 8)) {
                    ra0 = *reinterpret_cast<const float4*>(arow0 + (kc + 1) * 32);
                    ra1 = *reinterpret_cast<const float4*>(arow1 + (kc + 1) * 32);
                    rb0 = *reinterpret_cast<const float4*>(b0 + (kc + 1) * 32);
                    rb1 = *reinterpret_cast<const float4*>(b1 + (kc + 1) * 32);
                }
                if (!DTK_DBG(dbg, 4)) {
                    const h8 bh = *reinterpret_cast<const h8*>(&Bh[cur][(w * 16 + fj) * RP + fg * 8]);
                    const h8 bl = *reinterpret_cast<const h8*>(&Bl[cur][(w * 16 + fj) * RP + fg * 8]);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        if (mt < mt0 || mt > mt1) continue;  // workgroup-uniform
                        const h8 ah = *reinterpret_cast<const h8*>(&Ah[cur][(mt * 16 + fj) * RP + fg * 8]);
                        const h8 al = *reinterpret_cast<const h8*>(&Al[cur][(mt * 16 + fj) * RP + fg * 8]);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[mt], 0, 0, 0);
                    }
                }
                if (kc + 1 < nkc) RC_STORE(cur ^ 1);
                __syncthreads();
                cur ^= 1;
            }
#undef RC_STORE
            const int ci = blk + w * 16 + fj;
            if (ci < ncells) {
                const int cr = rmin + ci / nc, ccol = cmin + ci % nc;
                const float fn = norms[(size_t)gf * HW + cr * pw + ccol];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    if (mt < mt0 || mt > mt1) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int sidx = mt * 16 + fg * 4 + r;  // rows 4*fg + r of the D fragment of M tile mt
                        if (s_grp[sidx] != gi) continue;
                        const int dr = cr - (s_k[sidx] / pw - (RD + 2)), dc = ccol - (s_k[sidx] % pw - (RD + 2));
                        if (dr >= 0 && dr < WX && dc >= 0 && dc < WX)
                            xwin[((size_t)(s_m[sidx] - m0) * WX + dr) * WX + dc] =
                                fmaxf(acc[mt][r] * (1.f / (RC_SCALE * RC_SCALE)) / fmaxf(s_sn[sidx] * fn, 1e-8f), 0.f);
                    }
                }
            }
        }
    }
}

// s_waitcnt vmcnt(n) for a count that is a compile-time constant after unrolling (n <= 40)
__device__ __forceinline__ void vm_wait_n(int n) {
    switch (n) {
#define DTK_VMW(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        DTK_VMW(0) DTK_VMW(1) DTK_VMW(2) DTK_VMW(3) DTK_VMW(4) DTK_VMW(5) DTK_VMW(6) DTK_VMW(7) DTK_VMW(8) DTK_VMW(9) DTK_VMW(10)
        DTK_VMW(11) DTK_VMW(12) DTK_VMW(13) DTK_VMW(14) DTK_VMW(15) DTK_VMW(16) DTK_VMW(17) DTK_VMW(18) DTK_VMW(19) DTK_VMW(20)
        DTK_VMW(21) DTK_VMW(22) DTK_VMW(23) DTK_VMW(24) DTK_VMW(25) DTK_VMW(26) DTK_VMW(27) DTK_VMW(28) DTK_VMW(29) DTK_VMW(30)
        DTK_VMW(31) DTK_VMW(32) DTK_VMW(33) DTK_VMW(34) DTK_VMW(35) DTK_VMW(36) DTK_VMW(37) DTK_VMW(38) DTK_VMW(39) DTK_VMW(40)
#undef DTK_VMW
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// The same correlations for C = 32 NKC known at compile time (C = 384: NKC = 12), restructured around what the counters
// showed (profiles/r04_pmc_sq.md, refine_corr: waves parked 61 %, 25 % issuing -- two thirds of it the hi / lo split of every
// staged element --, MFMA pipe 13 % busy; neither a deeper register pipeline nor source-stationary operands alone moved it):
//   * the CELLS come pre-split from the volume's split-plane copy (featsplit_kernel) by LDS-DMA, 8 whole 128-byte lines per
//     request, into a ring of NS stages requested NS - 1 steps ahead: a step costs a wave two requests, eight fragment reads
//     and twelve MFMAs -- no VALU, no ds_write;
//   * the SOURCES are stationary: a wave owns ONE M tile, reads its 16 source rows once per tile and keeps them as split
//     fp16 A fragments (NKC x (hi, lo) x 4 = 96 registers);
//   * the (64-cell block, 32-channel chunk) steps of a group form one sequence, so the pipeline never restarts inside a box.
// LDS image of a stage: [64 cells][8 pieces of 16 B] (pieces 0-3: hi, 4-7: lo), piece p of cell c in slot p ^ ((c >> 1) & 7)
// (through the SOURCE address of the DMA, whose LDS side is lane-linear): the fragment reads are conflict-free ds_read_b128.
// Values and summation order are those of refine_corr_kernel: the results are bit-identical.
// vmcnt bookkeeping (gfx9: loads and stores share the counter and retire in order): every wave issues the SAME VMEM stream --
// two requests per step, 16 window stores per block as buffer stores whose switched-off lanes carry an out-of-range offset
// (no branch, so the count is static); the norms of the box come from LDS (a compiler-managed global load here would be
// waited for with vmcnt(0) and drain the ring).
template <int NKC, int NS, int NBM, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 3) void refine_corr_dma_kernel(dtk_geom g, const half_t* __restrict__ fs,
                                                              const float* __restrict__ norms,
                                                              const float* __restrict__ emb,
                                                              const int32_t* __restrict__ src_row,
                                                              const int32_t* __restrict__ tgt,
                                                              const int32_t* __restrict__ kstar,
                                                              const float* __restrict__ snorm,
                                                              const int32_t* __restrict__ perm,
                                                              const int32_t* __restrict__ nvalid,
                                                              float* __restrict__ xwin, unsigned xwin_bytes, int m0,
                                                              int ntiles, const float* __restrict__ rc_scale) {
    const float RC_SCALE = *rc_scale;   // (a scalar load, before the ring starts: the vmcnt bookkeeping below never sees it)
    // NS = stages of the LDS ring, NBM = largest union box (cells) correlated as one group
    constexpr int STAGE = 64 * 128;    // bytes: 64 cells x (32 hi + 32 lo) halves
    static_assert(NKC % NS == 0 && NKC >= NS, "the stage of a step is chosen at compile time");
    constexpr int C = NKC * 32;
    // NW waves per workgroup, each ONE M tile: SRC = 16 NW key-consecutive sources share every streamed cell (NW = 8: half the
    // stream per source; one workgroup of 8 waves per CU, hence the deeper ring)
    constexpr int SRC = 16 * NW, NT = 64 * NW, REQ = 8 / NW, WCELLS = 64 / NW;
    static_assert(NW == 4 || NW == 8, "a step is 8 DMA requests of 8 cells");
    __shared__ float s_sn[SRC];
    __shared__ int s_row[SRC], s_f[SRC], s_k[SRC], s_m[SRC], s_grp[SRC], s_box[SRC * 4], s_first[SRC], s_last[SRC], s_ng;
    __shared__ float s_fn[NBM];
    __shared__ __attribute__((aligned(1024))) unsigned char ring[NS * STAGE];
    const int ph = g.ph, pw = g.pw, HW = ph * pw;
    const int nv = *nvalid;
    const int per = (ntiles + 7) / 8;
    const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int t0 = tile * SRC;
    if ((int)(blockIdx.x >> 3) >= per || tile >= ntiles || t0 >= nv) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < SRC) {
        const bool ok = t0 + tid < nv;
        const int i = perm[ok ? t0 + tid : t0];  // index inside the round
        const int m = m0 + i;
        s_m[tid] = m;
        s_row[tid] = src_row ? src_row[m] : m;
        s_f[tid] = ok ? min(max(tgt[m], 0), g.T - 1) : -1;
        s_k[tid] = kstar[i];
        s_sn[tid] = snorm[i];
    }
    __syncthreads();
    // greedy grouping: consecutive (key-sorted) sources with the same frame whose window union stays small
    if (tid == 0) {
        int ng = 0, cf = -2, r0 = 0, r1 = 0, c0 = 0, c1 = 0;
        for (int s = 0; s < SRC; ++s) {
            s_grp[s] = -1;
            if (s_f[s] < 0) continue;
            const int kr = s_k[s] / pw, kc = s_k[s] % pw;
            const int a0 = max(kr - (RD + 2), 0), a1 = min(kr + (RD + 2), ph - 1);
            const int b0 = max(kc - (RD + 2), 0), b1 = min(kc + (RD + 2), pw - 1);
            bool fits = false;
            if (ng > 0 && s_f[s] == cf) {
                const int n0 = min(r0, a0), n1 = max(r1, a1), e0 = min(c0, b0), e1 = max(c1, b1);
                if ((n1 - n0 + 1) * (e1 - e0 + 1) <= NBM) { r0 = n0; r1 = n1; c0 = e0; c1 = e1; fits = true; }
            }
            if (!fits) { ++ng; cf = s_f[s]; r0 = a0; r1 = a1; c0 = b0; c1 = b1; s_first[ng - 1] = s; }
            s_grp[s] = ng - 1;
            s_last[ng - 1] = s;
            s_box[(ng - 1) * 4 + 0] = r0; s_box[(ng - 1) * 4 + 1] = r1;
            s_box[(ng - 1) * 4 + 2] = c0; s_box[(ng - 1) * 4 + 3] = c1;
        }
        s_ng = ng;
    }
    __syncthreads();
    const int ng = s_ng;

    const int fj = lane & 15, fg = lane >> 4;
    // A fragments of this wave's sources: lane (fj, fg) holds k = 32 kc + 8 fg .. + 7 of source 16 w + fj
    h8 ah[NKC], al[NKC];
    {
        const float* ap = emb + (size_t)s_row[16 * w + fj] * C + fg * 8;
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const float4 u = *reinterpret_cast<const float4*>(ap + kc * 32), v = *reinterpret_cast<const float4*>(ap + kc * 32 + 4);
            const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xs = x[e] * RC_SCALE;
                ah[kc][e] = (half_t)xs;
                al[kc][e] = (half_t)(xs - (float)ah[kc][e]);
            }
        }
    }
    const dtk_u4 srd = dtk_make_srd(fs);
    dtk_u4 srd_x;  // window stores: range-checked, so that an offset of ~0 switches a lane off
    {
        const unsigned long long xa = (unsigned long long)(size_t)xwin;
        srd_x = dtk_u4{(unsigned)xa, (unsigned)(xa >> 32) & 0xffffu, xwin_bytes, 0x00020000u};
#pragma unroll
        for (int k = 0; k < 4; ++k) srd_x[k] = __builtin_amdgcn_readfirstlane(srd_x[k]);
    }
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)ring + (unsigned)(w * WCELLS * 128));  // this wave's cells of a stage
    // fragment reads: cell nt * 16 + fj, hi piece fg / lo piece 4 + fg ((c >> 1) & 7 does not depend on nt)
    const int rd_hi = fj * 128 + ((fg ^ ((fj >> 1) & 7)) << 4), rd_lo = fj * 128 + (((4 + fg) ^ ((fj >> 1) & 7)) << 4);
    // requests: lane l of request q fetches, for cell WCELLS w + 8 q + (l >> 3) of the block, the piece that lands in slot l & 7
    const int dq = lane >> 3;
    for (int gi = 0; gi < ng; ++gi) {
        const int rmin = s_box[gi * 4], rmax = s_box[gi * 4 + 1], cmin = s_box[gi * 4 + 2], cmax = s_box[gi * 4 + 3];
        const int nc = cmax - cmin + 1, ncells = (rmax - rmin + 1) * nc;
        const bool active = 16 * w <= s_last[gi] && 16 * w + 15 >= s_first[gi];  // wave-uniform: members are consecutive
        const int gf = s_f[s_first[gi]];
        auto req_off = [&](int blk, int q) {  // clamped: results of padded cells are never stored
            const int c = WCELLS * w + 8 * q + dq;
            const int l = min(blk + c, ncells - 1);
            const unsigned cell = (unsigned)(gf * HW + (rmin + l / nc) * pw + cmin + l % nc);
            return cell * (unsigned)(C * 4) + (unsigned)((((lane & 7) ^ ((c >> 1) & 7))) << 4);
        };
        // this lane's four sources (rows 4 fg + r of the D fragments): window origin, output offset, norm, membership
        int krow[4], kcol[4];
        unsigned xoff[4];
        float sn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int sidx = 16 * w + 4 * fg + r;
            krow[r] = s_grp[sidx] == gi ? s_k[sidx] / pw - (RD + 2) : -100000;  // a non-member never matches a window
            kcol[r] = s_k[sidx] % pw - (RD + 2);
            xoff[r] = (unsigned)(s_m[sidx] - m0) * (unsigned)(WX * WX);
            sn[r] = s_sn[sidx];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // the previous group is done with the ring and with s_fn
        for (int c = tid; c < ncells; c += NT) s_fn[c] = norms[(size_t)gf * HW + (rmin + c / nc) * pw + cmin + c % nc];
        unsigned vc[2] = {req_off(0, 0), req_off(0, REQ - 1)}, vn[2] = {0u, 0u};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the norms: the counting below starts from an empty queue)
#define RC_ISSUE(STG, V, KC)                                                                  \
    do {                                                                                      \
        dtk_buffer_lds16<0>(srd, (unsigned)((KC) * 128), V[0], lds_w + (unsigned)((STG) * STAGE));                       \
        if (REQ == 2) dtk_buffer_lds16<0>(srd, (unsigned)((KC) * 128), V[1], lds_w + (unsigned)((STG) * STAGE + 1024));  \
    } while (0)
#pragma unroll
        for (int d = 0; d < NS - 1; ++d) RC_ISSUE(d, vc, d);
        for (int blk = 0; blk < ncells; blk += 64) {
            const bool last_blk = blk + 64 >= ncells;
            if (!last_blk) { vn[0] = req_off(blk + 64, 0); vn[1] = req_off(blk + 64, REQ - 1); }
            f4 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                const int stg = kc % NS;
                // the requests of this step have landed when at most those issued after them are outstanding: the requests of
                // the next two steps (where those exist) and, in the first three steps of a later block, the 16 window stores
                // of the previous block (issued between the request of step 2 and that of step 3)
                // (requests younger than this step's: two per existing step among the next NS - 2; in the last block those
                //  past the end do not exist)
                const int young = REQ * (last_blk ? min(NKC - 1 - kc, NS - 2) : NS - 2) + ((blk > 0 && kc < NS - 1) ? 16 : 0);
                vm_wait_n(young);
                __syncthreads();  // every wave's part of this stage has landed; stage (kc - 1) % NS is free again
                if (kc + NS - 1 < NKC) RC_ISSUE((kc + NS - 1) % NS, vc, kc + NS - 1);
                else if (!last_blk) RC_ISSUE((kc + NS - 1) % NS, vn, kc + NS - 1 - NKC);
                if (active) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const h8 bh = *reinterpret_cast<const h8*>(&ring[stg * STAGE + nt * 2048 + rd_hi]);
                        const h8 bl = *reinterpret_cast<const h8*>(&ring[stg * STAGE + nt * 2048 + rd_lo]);
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kc], bh, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kc], bl, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kc], bh, acc[nt], 0, 0, 0);
                    }
                }
            }
            vc[0] = vn[0]; vc[1] = vn[1];
            // windows of the block: D fragment, lane (fj, fg): sources 16 w + 4 fg + r (rows) of cell nt * 16 + fj (column)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int ci = blk + nt * 16 + fj;
                const int cic = min(ci, ncells - 1);
                const int cr = rmin + cic / nc, ccol = cmin + cic % nc;
                const float fn = s_fn[cic];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int dr = cr - krow[r], dc = ccol - kcol[r];
                    const bool ok = active && ci < ncells && (unsigned)dr < (unsigned)WX && (unsigned)dc < (unsigned)WX;
                    const float val = fmaxf(acc[nt][r] * (1.f / (RC_SCALE * RC_SCALE)) / fmaxf(sn[r] * fn, 1e-8f), 0.f);
                    const unsigned off = ok ? (xoff[r] + (unsigned)(dr * WX + dc)) * 4u : 0xfffffff0u;
                    asm volatile("buffer_store_dword %0, %1, %2, 0 offen" : : "v"(val), "v"(off), "s"(srd_x) : "memory");
                }
            }
        }
#undef RC_ISSUE
    }
}

// one wave per source
__global__ __launch_bounds__(256) void refine_head_kernel(dtk_geom g, const float* __restrict__ head,
                                                          const int32_t* __restrict__ src_row,
                                                          const int32_t* __restrict__ tgt,
                                                          const int32_t* __restrict__ out_idx,
                                                          float* __restrict__ out_xy, const Rec* __restrict__ rec,
                                                          const int32_t* __restrict__ kstar,
                                                          const float* __restrict__ xwin,
                                                          const float* __restrict__ zerr, Redo redo, Redo uncert,
                                                          int fast, int m0, int count, int M,
                                                          const int32_t* __restrict__ dM, int normalized) {
    constexpr int XP = 16;  // pitch of the x window in LDS: 15 columns + a column of ones (the bias row of GEMM1)
    __shared__ float s_x[4][WX * XP + 16];
    constexpr int PN = (WH * WH + 15) / 16 * 16;  // cells per tap plane
    __shared__ float s_p[4][9 * PN];
    __shared__ float s_z[4][WZ * WZ + 7];
    __shared__ float s_out[4][2];
    const int ph = g.ph, pw = g.pw;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + w;
    const int m = m0 + i;
    if (i >= count || m >= dtk_active(M, dM)) return;  // wave-uniform; no block-level barrier below
    const int k = kstar[i];
    if (k < 0) return;  // already queued for the exact path
    const int kr = k / pw, kc = k % pw;
    const float* xg = xwin + (size_t)i * WX * WX;
    float* sx = s_x[w];
    const int gq = lane >> 4, jq = lane & 15;
    // This kernel is bound by INSTRUCTION ISSUE (the round-1 form: ~1900 instructions per source, 77 of them MFMAs; at 4
    // cycles per wave64 instruction that is its whole run time), so the code below is organised to be short: no integer
    // divisions, no per-group index arithmetic, no predicated stores.
    // bit i of inside(k0, n, len): window index i <-> map coordinate k0 + i lies inside [0, n)
    auto inside = [](int k0, int n, int len) -> unsigned {
        const int lo = max(0, -k0), hi = min(len - 1, n - 1 - k0);
        return hi >= lo ? ((2u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
    };
    // x window: lane (gq, jq) loads column jq of rows gq, gq + 4, gq + 8, gq + 12; column 15 holds 1.0 in every row
    {
        const unsigned xr = inside(kr - (RD + 2), ph, WX), xc = inside(kc - (RD + 2), pw, WX);
        const bool colin = (xc >> jq) & 1u;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const int jr = gq + 4 * sl;
            if (jr < WX) {
                float v = jq == WX ? 1.f : 0.f;  // zero padding of conv1 outside the map
                if (colin && ((xr >> jr) & 1u)) v = xg[jr * WX + jq];
                sx[jr * XP + jq] = v;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- the refiner on the matrix cores (f32-input MFMA 16x16x4: fp32 products, fp32 accumulation), one ROW of the
    // 13 x 13 hidden window per pass (lanes jq = 13..15 of a 16-lane group compute values nobody reads):
    //   GEMM1  H^T[16 ch][16 px] = W1e[16 ch][12] . Xe[12][16 px]     k = tap 0..8, k = 9: bias (the column of ones), 10, 11: 0
    //   GEMM2  P[tap][16 px]     = W2[tap][16 ch] . relu(H^T)         rows 9..15 repeat taps 0..6
    // D of GEMM1 -- lane (g, j): channels 4g + r of cell j -- feeds GEMM2 as its B operand when k-step kp is given the
    // channels {4g' + kp}: lane group g' then simply supplies its register kp.  The per-tap planes go to LDS and
    //   z[zy][zx] = b2 + sum_tap P[tap][(zy + dy) * WH + zx + dx]
    // is nine LDS reads per logit instead of 144.  With a row per pass every LDS address is a per-lane constant plus an
    // immediate: a pass is 3 reads, 7 MFMAs, 4 relu + 4 selects (hidden cells outside the map are conv2's zero padding)
    // and 4 stores.  The repeated rows of GEMM2 store the same values to the same addresses as the originals (no
    // predicate); the stores of lanes 13..15 land on the first cells of the next row and are overwritten by that row's
    // pass (same wave, program order), or in the plane's padding after the last row.
    const float b2 = head[304];
    typedef __attribute__((address_space(3))) float lds_f32;
    float a1[3], a2[4];
    const lds_f32* xptr[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        const int kk = 4 * ks + gq;
        const float wv = head[kk < 9 ? jq * 9 + kk : 144 + jq];
        a1[ks] = kk <= 9 ? wv : 0.f;
        xptr[ks] = (const lds_f32*)sx + (kk < 9 ? (kk / 3) * XP + (kk % 3) + jq : WX);  // a tap, or the column of ones
    }
    float* Pb = s_p[w];
    lds_f32* pptr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int tap = 4 * gq + r;
        pptr[r] = (lds_f32*)Pb + (tap < 9 ? tap : tap - 9) * PN + jq;
    }
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) a2[kp] = head[160 + (4 * gq + kp) * 9 + (jq < 9 ? jq : jq - 9)];
    // hidden cell (hy, hx) of the window is cell (kr - (RD+1) + hy, kc - (RD+1) + hx) of the map
    const unsigned rmask = __builtin_amdgcn_readfirstlane(inside(kr - (RD + 1), ph, WH));
    const bool colok = (inside(kc - (RD + 1), pw, WH) >> jq) & 1u;
    // Two rows of the hidden window per pass, their MFMA chains INTERLEAVED (round 5): a row is a chain of 3 + 4 dependent f32
    // MFMAs, and the compiler, reusing the same two accumulators for every row, emitted the 13 rows strictly one after the other
    // with `s_nop 9` in front of every dependent read (ISA of round 4).  With two independent chains in flight every dependent
    // MFMA has another row's MFMA between it and its producer.  Same arithmetic per row: results are bit-identical.
#pragma unroll
    for (int hy = 0; hy < WH; hy += 2) {
        constexpr bool dummy = false; (void)dummy;
        const bool two = hy + 1 < WH;
        const bool inA = colok & (((rmask >> hy) & 1u) != 0);
        const bool inB = two && (colok & (((rmask >> (hy + 1)) & 1u) != 0));
        f4 d1a = {0.f, 0.f, 0.f, 0.f}, d1b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            d1a = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ks], xptr[ks][hy * XP], d1a, 0, 0, 0);
            if (two) d1b = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ks], xptr[ks][(hy + 1) * XP], d1b, 0, 0, 0);
        }
        f4 d2a = {0.f, 0.f, 0.f, 0.f}, d2b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) {
            float ra, rb;  // one v_max_f32 each (fmaxf compiles to a canonicalising v_max x, x first); NaN -> 0 like fmaxf
            asm("v_max_f32 %0, 0, %1" : "=v"(ra) : "v"(d1a[kp]));
            d2a = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[kp], inA ? ra : 0.f, d2a, 0, 0, 0);
            if (two) {
                asm("v_max_f32 %0, 0, %1" : "=v"(rb) : "v"(d1b[kp]));
                d2b = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[kp], inB ? rb : 0.f, d2b, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) pptr[r][hy * WH] = d2a[r];
        if (two) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pptr[r][(hy + 1) * WH] = d2b[r];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* zb = s_z[w];
    for (int j = lane; j < WZ * WZ; j += WAVE) {
        const int zy = j / WZ, zx = j % WZ;  // centre in the hidden window: (zy+1, zx+1)
        float a = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) a += Pb[(dy * 3 + dx) * PN + (zy + dy) * WH + zx + dx];
        zb[j] = a + b2;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const Rec rc = rec[i];
    auto zfun = [&](int r, int c) { return zb[(r - (kr - RD)) * WZ + (c - (kc - RD))]; };
    if (fast) {
        // Certificate that the zero-mass fallback (tracker_head.py:86-94) cannot fire, without the map's softmax
        // statistics: the masked mass is at least exp(zw) / Z_ub, zw = largest exact logit inside the disk (the arg-max
        // cell is always inside) and Z_ub >= sum over all cells of exp(z) for ANY map with values in [0, amax]:
        //   interior cells (all nine conv2 taps inside the map):  z <= zi = b2 + sum_ch [W2+ hmax + W2- hmin]
        //   border cells (conv2 sees zero-padded hidden cells):   z <= zb = b2 + sum_ch  W2+ hmax        (>= zi)
        // with hmax = relu(b1 + P1 amax), hmin = relu(b1 + N1 amax) per channel (hmin must not be used where a hidden
        // tap can be padding: 0 < hmin there).  If exp(zw) / Z_ub exceeds the 1e-8 threshold the result is the plain
        // ratio, in which the statistics cancel.
        // the disk cells of this lane (two slots of the WZ x WZ window), shared by the certificate and the soft arg-max
        float zv[2], cx[2], cy[2];
        bool okc[2];
        float zw = -INFINITY;
        const float half = (float)(g.patch / 2);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int j = lane + WAVE * sl;
            const int jr = j / WZ, jc = j - jr * WZ;
            const int r = kr - RD + jr, c = kc - RD + jc;
            const float dx = (float)((c - kc) * g.stride), dy = (float)((r - kr) * g.stride);
            okc[sl] = j < WZ * WZ && r >= 0 && r < ph && c >= 0 && c < pw && sqrtf(dx * dx + dy * dy) <= g.radius;
            zv[sl] = okc[sl] ? zb[min(j, WZ * WZ - 1)] : -INFINITY;
            cx[sl] = (float)(c * g.stride) + half;
            cy[sl] = (float)(r * g.stride) + half;
            zw = fmaxf(zw, zv[sl]);
        }
        zw = wave_max(zw);
        const float* cf = zerr + 8;  // = wpk + 160: P1, N1, W2p, W2n per channel
        const float amx = fminf(rc.amax + 2e-3f, 1.f);
        float zub_b = 0.f, zub_i = 0.f;
        if (lane < 16) {
            const float bb = head[144 + lane];
            zub_b = cf[32 + lane] * fmaxf(bb + cf[lane] * amx, 0.f);
            zub_i = zub_b + cf[48 + lane] * fmaxf(bb + cf[16 + lane] * amx, 0.f);
        }
        zub_b = wave_sum(zub_b) + head[304];
        zub_i = wave_sum(zub_i) + head[304];
        const float n_int = (float)(max(ph - 2, 0) * max(pw - 2, 0)), n_brd = (float)(ph * pw) - n_int;
        const float log_zsum = zub_b + logf(n_brd + n_int * expf(zub_i - zub_b));  // zub_i <= zub_b
        const bool certified = (log_zsum - zw) < (18.42f - 0.1f);
        if (!certified) {
            if (lane == 0) {
                const int slot = atomicAdd(uncert.count, 1);
                uncert.src_row[slot] = src_row ? src_row[m] : m;
                uncert.tgt[slot] = min(max(tgt[m], 0), g.T - 1);
                uncert.out_idx[slot] = out_idx ? out_idx[m] : m;
            }
            return;
        }
        // softmax statistics (zw, 1): the ratio is independent of them, and sq >= 1 (the cell that attains zw contributes exp(0)) --
        // the zero-mass branch of dtk_softargmax_finish is dead, so the three geometry sums it would need (count and coordinate
        // sums of the disk) are not formed here (round 5: three wave reductions, ~6 % of this kernel's instructions)
        float sq = 0.f, sqx = 0.f, sqy = 0.f;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
            if (okc[sl]) {
                const float q = expf(zv[sl] - zw);
                sq += q; sqx += q * cx[sl]; sqy += q * cy[sl];
            }
        sq = wave_sum(sq); sqx = wave_sum(sqx); sqy = wave_sum(sqy);
        if (lane == 0) {
            dtk_softargmax_finish(g, sq, sqx, sqy, 1.f, 0.f, 0.f, normalized, s_out[w]);
            const int oi = out_idx ? out_idx[m] : m;
            out_xy[2 * (size_t)oi] = s_out[w][0];
            out_xy[2 * (size_t)oi + 1] = s_out[w][1];
        }
        return;
    }
    float sq = 0.f;
    dtk_disk_softargmax(g, k, rc.zmax, rc.Z, zfun, normalized, s_out[w], &sq);
    if (lane == 0) {
        // the fallback test sq < 1e-8 uses the fp16 pass's (zmax, Z), i.e. sq is known up to a factor exp(+-E):
        // when the test is not clear-cut the exact path decides
        const float band = expf(fminf(*zerr, 80.f)) * 1.5f;
        const bool unclear = (sq > 1e-8f / band && sq < 1e-8f * band) || !(rc.Z > 0.f) || !(sq == sq);
        const int oi = out_idx ? out_idx[m] : m;
        if (unclear) {
            const int slot = atomicAdd(redo.count, 1);
            redo.src_row[slot] = src_row ? src_row[m] : m;
            redo.tgt[slot] = min(max(tgt[m], 0), g.T - 1);
            redo.out_idx[slot] = oi;
        } else {
            out_xy[2 * (size_t)oi] = s_out[w][0];
            out_xy[2 * (size_t)oi + 1] = s_out[w][1];
        }
    }
}

struct MfmaLayout {
    size_t s16, maps, rec, wpk, kstar, xwin, snorm, rown, perm, hist, off, cursor, bsum, nvalid, redo_cnt, redo_lists, exact, total;
    int HWk, nkeys, nblocks, cap;
    size_t unc_lists, trec;
    int MP;      // pitch (in halves) of one padded fp16 map: (ph+2) x (pw+4) + slack, multiple of 8
    int chunk;   // sources per corr16/head16 launch: their fp16 maps stay Infinity-Cache resident
    int super;   // sources per refine / redo round: large, so that uneven tiles balance across the chip
    int HWp;
};

MfmaLayout mfma_layout(const dtk_geom* g, int M, int round_sources) {
    MfmaLayout L;
    L.chunk = M < MFMA_CHUNK ? ((M + CM - 1) / CM * CM) : MFMA_CHUNK;
    // sources per round: MFMA_SUPER unless the caller asks for smaller rounds (dtk_track_opts.round_sources; tests use
    // it to run several rounds on small inputs).  Multiples of 256 (corr_peaks owns 256 sources per workgroup).
    const int round = round_sources > 0 ? (round_sources + 255) / 256 * 256 : MFMA_SUPER;
    L.super = M < round ? ((M + CM - 1) / CM * CM) : round;
    if (L.chunk > L.super) L.chunk = L.super;
    L.HWp = hw_pad(g->ph, g->pw);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    L.s16 = off; off = al(off + (size_t)L.super * g->C * 2);  // corr_peaks converts a whole super-chunk at once
    L.MP = ((g->ph + 2) * map_xw(g->pw) + 32 + 7) & ~7;
    L.maps = off; off = al(off + (size_t)L.chunk * L.MP * 2);
    L.rec = off; off = al(off + (size_t)L.super * sizeof(Rec));
    L.trec = off; off = al(off + (size_t)L.chunk * (L.HWp / CN) * sizeof(TileRec));
    L.wpk = off; off = al(off + 256 * 4);
    L.kstar = off; off = al(off + (size_t)L.super * 4);
    L.xwin = off; off = al(off + (size_t)L.super * WX * WX * 4);
    L.HWk = (g->ph + 7) / 8 * 8 * g->pw;
    L.nkeys = g->T * L.HWk;
    L.nblocks = (L.nkeys + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    L.snorm = off; off = al(off + (size_t)L.super * 4);
    L.rown = off; off = al(off + (size_t)L.super * 4);      // |row| of the source-row table (written by src16 in its table form)
    L.perm = off; off = al(off + (size_t)L.super * 4);
    L.hist = off; off = al(off + (size_t)L.nkeys * 4);     // hist and cursor are contiguous: one memset
    L.cursor = off; off = al(off + (size_t)L.nkeys * 4);
    L.off = off; off = al(off + (size_t)(L.nkeys + 1) * 4);
    L.bsum = off; off = al(off + (size_t)(L.nblocks + 1) * 4);
    L.nvalid = off; off = al(off + 16);
    L.cap = (M + CM - 1) / CM * CM;                            // redo / uncertified lists can hold every source
    L.redo_cnt = off; off = al(off + 16);                     // [0] redo count, [1] uncertified count
    L.redo_lists = off; off = al(off + (size_t)3 * L.cap * 4);
    L.unc_lists = off; off = al(off + (size_t)3 * L.cap * 4);
    L.exact = off;
    // staging of the exact path for re-done sources: large, so that a refine round costs a handful of (mostly empty)
    // exact-path launches instead of hundreds
    const int redo_rows = L.super < 65536 ? L.super : 65536;
    L.total = off + (size_t)redo_rows * (((size_t)g->ph * g->pw + 63) / 64 * 64 + 1) * sizeof(float);
    return L;
}

}  // namespace

size_t dtk_track_mfma_workspace_bytes(const dtk_geom* g, int M, int round_sources) {
    return mfma_layout(g, M, round_sources).total;
}

#ifdef DTK_DEV
// development aid (DTK_DEV builds only, not part of dtk.h): read and reset the counters enabled by DTK_DEBUG & 16 / 4096
extern "C" int dtk_debug_counters(unsigned long long* out4) {
    unsigned long long z[4] = {0, 0, 0, 0};
    DTK_HIP(hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_dbg), sizeof(z)));
    DTK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof(z)));
    return DTK_OK;
}
#endif

extern "C" size_t dtk_feat_f16_bytes(const dtk_geom* g) {
    if (!g || g->T <= 0 || g->C <= 0) return 0;
    const size_t unit = (size_t)g->T * hw_pad(g->ph, g->pw) * g->C * 2;
    (void)unit;
    return rc_scale_offset(g) + 256;   // unit-norm copy [+ split planes] + the scale slot of the window correlations
}

extern "C" int dtk_make_feat_f16(const dtk_geom* g, const float* feat, const float* norms, void* feat_f16, void* stream) {
    DTK_REQUIRE(g && feat && norms && feat_f16, "dtk_make_feat_f16: null pointer");
    DTK_REQUIRE(g->C % CK == 0, "dtk_make_feat_f16: C=%d must be a multiple of %d for the MFMA path", g->C, CK);
    const int HWp = hw_pad(g->ph, g->pw);
    const long long cells = (long long)g->T * HWp;
    DTK_LAUNCH("feat16", feat16_kernel, dim3(dtk_cdiv(cells, 4)), dim3(256), 0, dtk_stream(stream), feat, norms,
               reinterpret_cast<half_t*>(feat_f16), g->T, g->ph, g->pw, pw_pad(g->pw), g->C);
    // the scale of the window correlations' fp16 halves: from the largest cell norm of THIS volume (see RC_SCALE_MAX)
    unsigned* slot = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(feat_f16) + rc_scale_offset(g));
    DTK_HIP(dtk_zero_async(slot, 256, dtk_stream(stream)));   // (dtk_make_feat_f16 runs inside the captured training iteration: common.h)
    {
        const long long nn = (long long)g->T * g->ph * g->pw;
        const int blocks = (int)(dtk_cdiv(nn, 256) < 512 ? dtk_cdiv(nn, 256) : 512);
        DTK_LAUNCH("rcscale", rcscale_max_kernel, dim3(blocks), dim3(256), 0, dtk_stream(stream), norms, nn, slot);
        DTK_LAUNCH("rcscale", rcscale_final_kernel, dim3(1), dim3(1), 0, dtk_stream(stream), slot);
    }
    if (has_split_planes(g)) {
        const long long n8 = (long long)g->T * g->ph * g->pw * g->C / 8;
        DTK_LAUNCH("featsplit", featsplit_kernel, dim3(dtk_cdiv(n8, 256)), dim3(256), 0, dtk_stream(stream), feat,
                   reinterpret_cast<half_t*>(reinterpret_cast<unsigned char*>(feat_f16) + split_planes_offset(g)), n8,
                   reinterpret_cast<const float*>(slot));
    }
    return DTK_OK;
}

namespace {

struct SrcLists {
    const int32_t *src_row, *tgt, *out_idx;
};

// One pass of the MFMA pipeline over `count` sources (exact count, known on the host).
//   fast = true : corr16 -> peak16 -> rescore/sort/refine_corr -> refine_head with the no-fallback certificate;
//                 sources it cannot certify go to `uncert`, sources the fp16 pass cannot decide go to `redo`
//   fast = false: corr16 -> head16 (whole-map refiner statistics on the matrix cores) -> ... -> refine_head with the
//                 statistics; undecidable sources go to `redo`
int mfma_phase(const dtk_geom* g, const MfmaLayout& L, unsigned char* ws, const float* feat, const float* norms,
               const half_t* f16, const float* head, const float* emb, SrcLists in, float* out_xy, int count,
               int normalized, bool fast, Redo redo, Redo uncert, size_t lds_head, hipStream_t st, int dbg,
               int32_t* arg_cell = nullptr, float* arg_cos = nullptr, bool row_table = false) {
    half_t* s16 = reinterpret_cast<half_t*>(ws + L.s16);
    half_t* maps = reinterpret_cast<half_t*>(ws + L.maps);
    Rec* rec = reinterpret_cast<Rec*>(ws + L.rec);
    uint32_t* wpk = reinterpret_cast<uint32_t*>(ws + L.wpk);
    int32_t* kstar = reinterpret_cast<int32_t*>(ws + L.kstar);
    float* xwin = reinterpret_cast<float*>(ws + L.xwin);
    int32_t* hist = reinterpret_cast<int32_t*>(ws + L.hist);
    int32_t* cursor = reinterpret_cast<int32_t*>(ws + L.cursor);
    int32_t* koff = reinterpret_cast<int32_t*>(ws + L.off);
    int32_t* bsum = reinterpret_cast<int32_t*>(ws + L.bsum);
    int32_t* nvalid = reinterpret_cast<int32_t*>(ws + L.nvalid);
    float* snorm = reinterpret_cast<float*>(ws + L.snorm);
    int32_t* perm = reinterpret_cast<int32_t*>(ws + L.perm);
    const int32_t* nodm = nullptr;
    const int M = count;
    for (long long s0 = 0; s0 < M; s0 += L.super) {
        const int scnt = (int)((M - s0) < L.super ? (M - s0) : L.super);
        // source-stationary fused correlation + selection (C = 384, position tags of 13 bits)
        const int pk_cb = DTK_DBG(dbg, 65536) ? 2 : PK_CB, pk_cells = 32 * pk_cb;
        const bool peaks_shape = L.HWp / pk_cells <= (1 << (PK_IDX_BITS - (pk_cb > 1 ? 5 : 4))) &&
                                 L.HWp % pk_cells == 0 && pw_pad(g->pw) % pk_cells == 0 && !DTK_DBG(dbg, 2048) &&
                                 (long long)g->T * L.HWp * g->C * 2 < (1LL << 32);   // (32-bit tile offsets of the LDS-DMA descriptor)
        // round 6: C = 768 / 1024 on the K-split form (corr_peaks_wide_kernel: wave pairs share 64 sources, one K half each)
        const bool peaks_wide = fast && peaks_wide_ok(g->C) && pk_cb == 1 && peaks_shape;
        const bool peaks = (fast && g->C == 384 && peaks_shape) || peaks_wide;
        // (histogram and cursors of the round's counting sort: zeroed in front of corr_peaks, whose epilogue already counts)
        DTK_HIP(hipMemsetAsync(ws + L.hist, 0, L.cursor + (size_t)L.nkeys * 4 - L.hist, st));
        if (peaks) {
            const int32_t* row_of = row_table ? in.src_row : nullptr;   // (the table: dtk_track_mfma)
            if (!row_of)
                DTK_LAUNCH("src16", src16_kernel, dim3(dtk_cdiv(scnt, 4)), dim3(256), 0, st, emb, in.src_row, s16, (int)s0, scnt, M,
                           nodm, g->C, PK_SRC_SCALE, (float*)nullptr);
            const dim3 pgrid(dtk_cdiv(scnt, peaks_wide ? PKW_SRC : PK_SRC));
            // single-candidate sources are finished in corr_peaks' epilogue when the row table (and its norms) exist
            const bool fuse_done = row_table && arg_cell == nullptr && !DTK_DBG(dbg, 2);
            const PeaksDone pdone = fuse_done ? PeaksDone{kstar, snorm, hist, reinterpret_cast<const float*>(ws + L.rown), L.HWk}
                                              : PeaksDone{nullptr, nullptr, nullptr, nullptr, 0};
#define DTK_PEAKS(V, CBV)                                                                                                    \
    do {                                                                                                                     \
        static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_peaks_kernel<24, V, CBV>),  \
                                                            hipFuncAttributeMaxDynamicSharedMemorySize,                      \
                                                            (int)peaks_lds_bytes(CBV));                                      \
        DTK_HIP(attr_);                                                                                                      \
        DTK_LAUNCH("corr_peaks", (corr_peaks_kernel<24, V, CBV>), pgrid, dim3(256), peaks_lds_bytes(CBV), st, *g, f16, s16,  \
                   in.tgt, rec, (int)s0, scnt, L.HWp, row_of, pdone);                                                        \
    } while (0)
#define DTK_PEAKS_WIDE(KSHV)                                                                                                  \
    do {                                                                                                                     \
        static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_peaks_wide_kernel<KSHV>),   \
                                                            hipFuncAttributeMaxDynamicSharedMemorySize,                      \
                                                            (int)peaks_wide_lds_bytes(32 * KSHV));                           \
        DTK_HIP(attr_);                                                                                                      \
        DTK_LAUNCH("corr_peaks", (corr_peaks_wide_kernel<KSHV>), pgrid, dim3(256), peaks_wide_lds_bytes(32 * KSHV), st, *g,  \
                   f16, s16, in.tgt, rec, (int)s0, scnt, L.HWp, row_of, pdone);                                              \
    } while (0)
            if (peaks_wide) {
                if (g->C == 1024) DTK_PEAKS_WIDE(32);
                else DTK_PEAKS_WIDE(24);
            } else
#ifdef DTK_DEV
            if (pk_cb == 2) {
                switch ((dbg >> 13) & 7) {
                    case 0: DTK_PEAKS(0, 2); break;
                    case 1: DTK_PEAKS(1, 2); break;
                    case 2: DTK_PEAKS(2, 2); break;
                    case 4: DTK_PEAKS(4, 2); break;
                    default: DTK_PEAKS(7, 2); break;
                }
            } else {
                switch ((dbg >> 13) & 7) {
                    case 0: DTK_PEAKS(0, PK_CB); break;
                    case 1: DTK_PEAKS(1, PK_CB); break;
                    case 2: DTK_PEAKS(2, PK_CB); break;
                    case 3: DTK_PEAKS(3, PK_CB); break;
                    case 4: DTK_PEAKS(4, PK_CB); break;
                    case 5: DTK_PEAKS(5, PK_CB); break;
                    case 6: DTK_PEAKS(6, PK_CB); break;
                    default: DTK_PEAKS(7, PK_CB); break;
                }
            }
#else
            DTK_PEAKS(0, PK_CB);
#endif
#undef DTK_PEAKS
#undef DTK_PEAKS_WIDE
        }
        for (long long m0 = s0; m0 < s0 + scnt && !peaks; m0 += L.chunk) {
            const int cnt = (int)((s0 + scnt - m0) < L.chunk ? (s0 + scnt - m0) : L.chunk);
            DTK_LAUNCH("src16", src16_kernel, dim3(dtk_cdiv(cnt, 4)), dim3(256), 0, st, emb, in.src_row, s16, (int)m0, cnt, M,
                       nodm, g->C, FSCALE, (float*)nullptr);
            {
                const int MT = dtk_cdiv(cnt, CM), NT = L.HWp / CN;
                const int blocks = 8 * ((MT + 7) / 8) * 8 * ((NT + 7) / 8);
                TileRec* trec = reinterpret_cast<TileRec*>(ws + L.trec);
                if (fast) {
                    DTK_LAUNCH("corr16_peaks", corr16_tiled_kernel<true>, dim3(blocks), dim3(256), 0, st, *g, f16, s16, in.tgt,
                               maps, trec, (int)m0, cnt, M, nodm, L.HWp, L.MP, dbg);
                    DTK_LAUNCH("select", select_kernel, dim3(dtk_cdiv(cnt, 4)), dim3(256), 0, st, *g, trec, NT, rec + (m0 - s0),
                               (int)m0, cnt, M, nodm);
                } else {
                    DTK_LAUNCH("corr16", corr16_tiled_kernel<false>, dim3(blocks), dim3(256), 0, st, *g, f16, s16, in.tgt,
                               maps, trec, (int)m0, cnt, M, nodm, L.HWp, L.MP, dbg);
                }
            }
            if (!fast) {
                DTK_LAUNCH("head16", head16_kernel, dim3(cnt), dim3(256), lds_head, st, *g, head, wpk, maps, L.MP,
                           rec + (m0 - s0), (int)m0, cnt, M, nodm, dbg);
            }
        }
        DTK_LAUNCH("rescore", rescore_kernel, dim3(dtk_cdiv(scnt, 4)), dim3(256), 0, st, *g, feat, norms, emb, in.src_row,
                   in.tgt, in.out_idx, rec, kstar, snorm, hist, L.HWk, redo, (int)s0, scnt, M, nodm, dbg, arg_cell, arg_cos,
                   row_table ? reinterpret_cast<const float*>(ws + L.rown) : (const float*)nullptr);
        if (arg_cell) continue;  // arg-max only: no window refinement
        DTK_LAUNCH("key_scan", scan_blocksum_kernel, dim3(L.nblocks), dim3(256), 0, st, hist, bsum, L.nkeys);
        DTK_LAUNCH("key_scan", scan_top_kernel, dim3(1), dim3(256), 0, st, bsum, L.nblocks, nvalid);
        DTK_LAUNCH("key_scan", scan_final_kernel, dim3(L.nblocks), dim3(256), 0, st, hist, bsum, koff, L.nkeys);
        DTK_LAUNCH("key_scatter", scatter_kernel, dim3(dtk_cdiv(scnt, 256)), dim3(256), 0, st, *g, in.tgt, kstar, koff, cursor,
                   perm, L.HWk, (int)s0, scnt);
        {
            const int rtiles = dtk_cdiv(scnt, RC_SRC);
            const float* rc_scale = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(f16) + rc_scale_offset(g));
            if (has_split_planes(g) && !DTK_DBG(dbg, 131072) && (size_t)L.super * WX * WX * 4 < (1ull << 32)) {
                const int dtiles = dtk_cdiv(scnt, 16 * RCD_NW);
#define DTK_RCD(NKCV)                                                                                                          \
    DTK_LAUNCH("refine_corr", (refine_corr_dma_kernel<NKCV, RCD_NS, RCD_NBM, RCD_NW>), dim3(8 * dtk_cdiv(dtiles, 8)),              \
               dim3(64 * RCD_NW), 0, st, *g,                                                                                       \
               reinterpret_cast<const half_t*>(reinterpret_cast<const unsigned char*>(f16) + split_planes_offset(g)), norms, emb,   \
               in.src_row, in.tgt, kstar, snorm, perm, nvalid, xwin, (unsigned)((size_t)scnt * WX * WX * 4), (int)s0, dtiles, \
               rc_scale)
                DTK_RCD(12);
#undef DTK_RCD
            }
            else
                DTK_LAUNCH("refine_corr", refine_corr_kernel, dim3(8 * dtk_cdiv(rtiles, 8)), dim3(256), 0, st, *g, feat, norms,
                           emb, in.src_row, in.tgt, kstar, snorm, perm, nvalid, xwin, (int)s0, rtiles, dbg, rc_scale);
        }
        DTK_LAUNCH("refine_head", refine_head_kernel, dim3(dtk_cdiv(scnt, 4)), dim3(256), 0, st, *g, head, in.src_row, in.tgt,
                   in.out_idx, out_xy, rec, kstar, xwin, reinterpret_cast<const float*>(wpk) + 152, redo, uncert,
                   fast ? 1 : 0, (int)s0, scnt, M, nodm, normalized);
    }
    return DTK_OK;
}

}  // namespace

// NOTE: unlike the rest of the library this entry synchronises `stream`: once after the first phase (the sizes of the
// whole-map tier and of the exact tier are only known on the device, and launching worst-case grids for them costs
// more than the round trip), once more if the whole-map tier ran, and once up front if `dM` is given.
int dtk_track_mfma(const dtk_geom* g, const float* feat, const float* norms, const void* feat_f16, const float* head,
                   const float* emb, const int32_t* src_row, const int32_t* tgt, const int32_t* out_idx, float* out_xy,
                   int M, const int32_t* dM, const dtk_track_opts* opts, dtk_track_stats* stats, void* workspace,
                   size_t workspace_bytes, void* stream) {
    DTK_REQUIRE(g->C % CK == 0, "dtk_track(mfma): C=%d must be a multiple of %d", g->C, CK);
    DTK_REQUIRE((int)(g->radius / (float)g->stride) <= RD, "dtk_track(mfma): disk radius %g px > %d cells", g->radius, RD);
    const int normalized = opts->normalized;
    const MfmaLayout L = mfma_layout(g, M, opts->round_sources);
    if (workspace_bytes < L.total) {
        dtk_set_error("dtk_track(mfma): workspace %zu B < required %zu B", workspace_bytes, L.total);
        return DTK_E_WORKSPACE;
    }
    hipStream_t st = dtk_stream(stream);
    const int dbg = dtk_dev_flags();
    unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
    int count = M;
    if (dM) {  // data-dependent number of sources: read it instead of launching worst-case grids
        int32_t v = 0;
        DTK_HIP(hipMemcpyAsync(&v, dM, sizeof(v), hipMemcpyDeviceToHost, st));
        DTK_HIP(hipStreamSynchronize(st));
        if (stats) stats->syncs++;
        count = v < M ? (v < 0 ? 0 : v) : M;
    }
    if (stats) stats->sources = count;
    if (count == 0) return DTK_OK;
    int32_t* counters = reinterpret_cast<int32_t*>(ws + L.redo_cnt);
    Redo redo, uncert;
    redo.count = counters;
    redo.src_row = reinterpret_cast<int32_t*>(ws + L.redo_lists);
    redo.tgt = redo.src_row + L.cap;
    redo.out_idx = redo.tgt + L.cap;
    uncert.count = counters + 1;
    uncert.src_row = reinterpret_cast<int32_t*>(ws + L.unc_lists);
    uncert.tgt = uncert.src_row + L.cap;
    uncert.out_idx = uncert.tgt + L.cap;
    const int ph = g->ph, pw = g->pw;
    const size_t lds_head = (size_t)((((ph + 2) * map_xw(pw) + 32 + 7) & ~7)) * 2 + 64 + 16 * 4 + (1 + KC) * 4 + 16;
    DTK_REQUIRE(lds_head <= 160 * 1024, "dtk_track(mfma): token grid %dx%d too large for head16 (%zu B LDS)", ph, pw, lds_head);
    DTK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(head16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_head));
    DTK_HIP(hipMemsetAsync(counters, 0, 4 * sizeof(int32_t), st));
    DTK_LAUNCH("head16_pack", head16_pack_kernel, dim3(1), dim3(256), 0, st, head, reinterpret_cast<uint32_t*>(ws + L.wpk));
    DTK_HIP(hipMemsetAsync(ws + L.maps, 0, (size_t)L.chunk * L.MP * 2, st));  // zero borders of the padded maps
    const half_t* f16 = reinterpret_cast<const half_t*>(feat_f16);
    const bool fast_ok = opts->tier != DTK_TIER_WHOLE_MAP && !DTK_DBG(dbg, 512);
    // Sources that are rows of a small matrix (the anchor stage: N T rows for N T (T + 1) sources): convert the ROWS to fp16 unit
    // vectors once per call instead of every round's sources in order (2.9 ms per benchmark step, 0.4 GB of copies per round);
    // corr_peaks then gathers its 64 sources per wave through src_row.  Needs the row count from the caller (opts->emb_rows).
    const bool row_table = fast_ok && src_row != nullptr && opts->emb_rows > 0 && opts->emb_rows <= L.super &&
                           (g->C == 384 || peaks_wide_ok(g->C)) && !DTK_DBG(dbg, 262144);
    if (row_table)
        DTK_LAUNCH("src16", src16_kernel, dim3(dtk_cdiv(opts->emb_rows, 4)), dim3(256), 0, st, emb, (const int32_t*)nullptr,
                   reinterpret_cast<half_t*>(ws + L.s16), 0, opts->emb_rows, opts->emb_rows, (const int32_t*)nullptr, g->C,
                   PK_SRC_SCALE, reinterpret_cast<float*>(ws + L.rown));
    int rc = mfma_phase(g, L, ws, feat, norms, f16, head, emb, SrcLists{src_row, tgt, out_idx}, out_xy, count, normalized,
                        fast_ok, redo, uncert, lds_head, st, dbg, nullptr, nullptr, row_table);
    if (rc) return rc;
    int32_t hc[2] = {0, 0};
    DTK_HIP(hipMemcpyAsync(hc, counters, sizeof(hc), hipMemcpyDeviceToHost, st));
    DTK_HIP(hipStreamSynchronize(st));
    if (stats) { stats->syncs++; stats->whole_map_tier = fast_ok ? hc[1] : count; }
    if (hc[1] > 0) {  // sources without a no-fallback certificate: whole-map statistics
        rc = mfma_phase(g, L, ws, feat, norms, f16, head, emb, SrcLists{uncert.src_row, uncert.tgt, uncert.out_idx}, out_xy,
                        hc[1], normalized, false, redo, uncert, lds_head, st, dbg);
        if (rc) return rc;
        DTK_HIP(hipMemcpyAsync(hc, counters, sizeof(hc), hipMemcpyDeviceToHost, st));
        DTK_HIP(hipStreamSynchronize(st));
        if (stats) stats->syncs++;
    }
    if (stats) stats->exact_tier = hc[0];
    if (hc[0] > 0) {  // sources the fp16 pass could not decide: the exact fp32 path
        rc = dtk_track_exact(g, feat, norms, head, emb, redo.src_row, redo.tgt, redo.out_idx, out_xy, hc[0], nullptr,
                             normalized, ws + L.exact, workspace_bytes - L.exact, stream);
        if (rc) return rc;
    }
    return DTK_OK;
}

int dtk_argmax_exact(const dtk_geom* g, const float* feat, const float* norms, const float* emb, const int32_t* src_row,
                     const int32_t* tgt, const int32_t* out_idx, int32_t* arg_cell, float* arg_cos, int M, void* workspace,
                     size_t workspace_bytes, void* stream);

// dtk_argmax_cells on the MFMA path: fp16 candidate search + fp32 re-scoring (the first two stages of dtk_track); sources
// the fp16 pass cannot decide (more than 10 cells in the band, non-positive maximum) take the exact fp32 path.
int dtk_argmax_mfma(const dtk_geom* g, const float* feat, const float* norms, const void* feat_f16, const float* emb,
                    const int32_t* src_row, const int32_t* tgt, int32_t* arg_cell, float* arg_cos, int M, void* workspace,
                    size_t workspace_bytes, void* stream) {
    DTK_REQUIRE(g->C % CK == 0, "dtk_argmax_cells(mfma): C=%d must be a multiple of %d", g->C, CK);
    const MfmaLayout L = mfma_layout(g, M, 0);
    if (workspace_bytes < L.total) {
        dtk_set_error("dtk_argmax_cells(mfma): workspace %zu B < required %zu B", workspace_bytes, L.total);
        return DTK_E_WORKSPACE;
    }
    hipStream_t st = dtk_stream(stream);
    unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
    int32_t* counters = reinterpret_cast<int32_t*>(ws + L.redo_cnt);
    Redo redo, uncert;
    redo.count = counters;
    redo.src_row = reinterpret_cast<int32_t*>(ws + L.redo_lists);
    redo.tgt = redo.src_row + L.cap;
    redo.out_idx = redo.tgt + L.cap;
    uncert = redo;
    DTK_HIP(hipMemsetAsync(counters, 0, 4 * sizeof(int32_t), st));
    int rc = mfma_phase(g, L, ws, feat, norms, reinterpret_cast<const half_t*>(feat_f16), nullptr, emb,
                        SrcLists{src_row, tgt, nullptr}, nullptr, M, 0, true, redo, uncert, 0, st, dtk_dev_flags(), arg_cell,
                        arg_cos);
    if (rc) return rc;
    int32_t hc = 0;
    DTK_HIP(hipMemcpyAsync(&hc, counters, sizeof(hc), hipMemcpyDeviceToHost, st));
    DTK_HIP(hipStreamSynchronize(st));
    if (hc > 0)
        return dtk_argmax_exact(g, feat, norms, emb, redo.src_row, redo.tgt, redo.out_idx, arg_cell, arg_cos, hc, ws + L.exact,
                                workspace_bytes - L.exact, stream);
    return DTK_OK;
}
