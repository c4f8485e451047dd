// anchors.hip -- anchor bookkeeping and occlusion (models/model_inference.py:130-200), device-side, no host sync.
//
//   dtk_build_anchor_sources : A_n = {a : cs[n][a] >= th}; pairs (n,a) numbered n-major (the layout of the
//                              reference's dict n -> [A_n, T, 2]); source list of the anchor stage sorted by
//                              anchor frame so that consecutive sources share their target feature frame.
//   dtk_occlusion            : d[k][t] = |G[k][t] - traj[a_k]|, lower median over anchors, threshold tau.
#include "common.h"

namespace {

__device__ __forceinline__ bool is_anchor(const float* cs, float th, int n, int T, int t) {
    return cs[(size_t)n * T + t] >= th;
}

__global__ __launch_bounds__(256) void anchor_count_kernel(const float* __restrict__ cs, float th, int N, int T,
                                                           int32_t* __restrict__ n_anchors) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    int c = 0;
    for (int t = 0; t < T; ++t) c += is_anchor(cs, th, n, T, t) ? 1 : 0;
    n_anchors[n] = c;
}

__global__ __launch_bounds__(256) void frame_count_kernel(const float* __restrict__ cs, float th, int N, int T,
                                                          int32_t* __restrict__ frame_cnt) {
    __shared__ int red[4];
    const int a = blockIdx.x;
    int c = 0;
    for (int n = threadIdx.x; n < N; n += 256) c += is_anchor(cs, th, n, T, a) ? 1 : 0;
    c = wave_sum_i(c);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) frame_cnt[a] = red[0] + red[1] + red[2] + red[3];
}

// block-wide inclusive scan of one int per thread (256 threads); returns inclusive prefix, *total = block sum
__device__ __forceinline__ int block_scan_incl(int v, int* lds4, int* total) {
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
    for (int i = 0; i < w; ++i) base += lds4[i];
    *total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    return x + base;
}

// single block: out[0..n] = exclusive scan of in[0..n-1]; optionally count zeros
__global__ __launch_bounds__(256) void exclusive_scan_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                             int n, int32_t* __restrict__ zero_count) {
    __shared__ int lds4[4];
    int carry = 0, zeros = 0;
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const int v = i < n ? in[i] : 0;
        if (i < n && v == 0) ++zeros;
        int total;
        const int incl = block_scan_incl(v, lds4, &total);
        if (i < n) out[i] = carry + incl - v;
        carry += total;
    }
    if (threadIdx.x == 0) out[n] = carry;
    if (zero_count) {
        zeros = wave_sum_i(zeros);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = zeros;
        __syncthreads();
        if (threadIdx.x == 0) *zero_count = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    }
}

__global__ void finalize_counts_kernel(const int32_t* __restrict__ pair_off, int N, int T, int32_t* __restrict__ counts) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        counts[0] = pair_off[N];
        counts[1] = pair_off[N] * T;
    }
}

// block a: walk the queries in order, place the pairs of frame a at frame_off[a] + (rank among queries)
__global__ __launch_bounds__(256) void emit_sources_kernel(const float* __restrict__ cs, float th, int N, int T,
                                                           const int32_t* __restrict__ pair_off,
                                                           const int32_t* __restrict__ frame_off,
                                                           int32_t* __restrict__ pair_frame,
                                                           int32_t* __restrict__ src_row, int32_t* __restrict__ tgt,
                                                           int32_t* __restrict__ out_idx) {
    __shared__ int lds4[4];
    const int a = blockIdx.x;
    int carry = frame_off[a];
    for (int n0 = 0; n0 < N; n0 += 256) {
        const int n = n0 + threadIdx.x;
        const int flag = (n < N && is_anchor(cs, th, n, T, a)) ? 1 : 0;
        int total;
        const int incl = block_scan_incl(flag, lds4, &total);
        if (flag) {
            const int j = carry + incl - 1;  // position among the pairs sorted by anchor frame
            int rank = 0;
            for (int t = 0; t < a; ++t) rank += is_anchor(cs, th, n, T, t) ? 1 : 0;
            const int p = pair_off[n] + rank;
            pair_frame[p] = a;
            const size_t mb = (size_t)j * T;
            for (int t = 0; t < T; ++t) {
                src_row[mb + t] = n * T + t;
                tgt[mb + t] = a;
                out_idx[mb + t] = p * T + t;
            }
        }
        carry += total;
    }
}

// |G[p][t] - traj[a]| without fma contraction (the reference evaluates sqrt(dx*dx + dy*dy) with separate roundings)
__device__ __forceinline__ float anchor_dist(const float* __restrict__ green, const float* __restrict__ tr, int p, int a,
                                             int T, int t) {
    const float dx = green[((size_t)p * T + t) * 2] - tr[2 * a];
    const float dy = green[((size_t)p * T + t) * 2 + 1] - tr[2 * a + 1];
    return sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
}

// one block per query, one wave per frame t at a time: every anchor distance is computed ONCE into LDS (so the rank
// selection compares stored values: an element can never compare "less than itself"), then lane k ranks d[k] against
// all A values; the lane whose rank is the lower median's publishes it (torch.median semantics, ties by index).
__global__ __launch_bounds__(256) void occlusion_kernel(const float* __restrict__ green,
                                                        const int32_t* __restrict__ pair_off,
                                                        const int32_t* __restrict__ pair_frame,
                                                        const float* __restrict__ traj, const float* __restrict__ cs,
                                                        float anchor_th, float cos_th, uint8_t* __restrict__ occ, int N,
                                                        int T) {
    extern __shared__ float smem_occ[];  // med[T] | d[4][T]
    float* med = smem_occ;
    __shared__ float red[4];
    const int n = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* d = smem_occ + T + w * T;
    const int p0 = pair_off[n], A = pair_off[n + 1] - p0;
    const float* tr = traj + (size_t)n * T * 2;
    if (A <= 0) {  // the reference raises here (torch.stack of an empty list); host reports it via counts[2]
        for (int t = threadIdx.x; t < T; t += 256) occ[(size_t)n * T + t] = 1;
        return;
    }
    const int want = (A - 1) / 2;  // torch.median: lower median
    for (int t = w; t < T; t += 4) {
        for (int k = lane; k < A; k += WAVE) d[k] = anchor_dist(green, tr, p0 + k, pair_frame[p0 + k], T, t);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < A; k += WAVE) {
            const float dk = d[k];
            int rank = 0;
            for (int k2 = 0; k2 < A; ++k2) {
                const float d2 = d[k2];
                rank += (d2 < dk || (d2 == dk && k2 < k)) ? 1 : 0;
            }
            if (rank == want) med[t] = dk;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // d is rewritten for the next frame
    }
    __syncthreads();
    float tau = -INFINITY;
    for (int t = threadIdx.x; t < T; t += 256)
        if (cs[(size_t)n * T + t] >= anchor_th) tau = fmaxf(tau, med[t]);
    tau = wave_max(tau);
    if (lane == 0) red[w] = tau;
    __syncthreads();
    tau = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    for (int t = threadIdx.x; t < T; t += 256)
        occ[(size_t)n * T + t] = (med[t] > tau || cs[(size_t)n * T + t] < cos_th) ? 1 : 0;
}

}  // namespace

extern "C" int dtk_build_anchor_sources(const float* cs, float anchor_th, int N, int T, int32_t* n_anchors,
                                        int32_t* pair_off, int32_t* pair_frame, int32_t* src_row, int32_t* tgt,
                                        int32_t* out_idx, int32_t* counts, int32_t* scratch, void* stream) {
    DTK_REQUIRE(cs && n_anchors && pair_off && pair_frame && src_row && tgt && out_idx && counts && scratch,
                "dtk_build_anchor_sources: null pointer");
    DTK_REQUIRE(N > 0 && T > 0 && (long long)N * T * T < 2147483647LL, "dtk_build_anchor_sources: N*T*T out of range");
    hipStream_t st = dtk_stream(stream);
    int32_t* frame_cnt = scratch;          // [T]
    int32_t* frame_off = scratch + T;      // [T+1]
    DTK_LAUNCH("anchor_count", anchor_count_kernel, dim3(dtk_cdiv(N, 256)), dim3(256), 0, st, cs, anchor_th, N, T, n_anchors);
    DTK_LAUNCH("exclusive_scan", exclusive_scan_kernel, dim3(1), dim3(256), 0, st, n_anchors, pair_off, N, counts + 2);
    DTK_LAUNCH("frame_count", frame_count_kernel, dim3(T), dim3(256), 0, st, cs, anchor_th, N, T, frame_cnt);
    DTK_LAUNCH("exclusive_scan", exclusive_scan_kernel, dim3(1), dim3(256), 0, st, frame_cnt, frame_off, T, (int32_t*)nullptr);
    DTK_LAUNCH("finalize_counts", finalize_counts_kernel, dim3(1), dim3(64), 0, st, pair_off, N, T, counts);
    DTK_LAUNCH("emit_sources", emit_sources_kernel, dim3(T), dim3(256), 0, st, cs, anchor_th, N, T, pair_off, frame_off,
                       pair_frame, src_row, tgt, out_idx);
    return DTK_OK;
}

extern "C" int dtk_occlusion(const float* green, const int32_t* pair_off, const int32_t* pair_frame, const float* traj,
                             const float* cs, float anchor_th, float cos_th, uint8_t* occ, int N, int T, void* stream) {
    DTK_REQUIRE(green && pair_off && pair_frame && traj && cs && occ && N >= 0 && T > 0, "dtk_occlusion: bad args");
    DTK_REQUIRE((size_t)5 * T * sizeof(float) <= 60 * 1024, "dtk_occlusion: T=%d too large", T);
    if (N == 0) return DTK_OK;
    DTK_LAUNCH("occlusion", occlusion_kernel, dim3(N), dim3(256), (size_t)5 * T * sizeof(float), dtk_stream(stream), green,
                       pair_off, pair_frame, traj, cs, anchor_th, cos_th, occ, N, T);
    return DTK_OK;
}


// ------------------------------------------------------------------------------------------------------------
// TAP-Vid metric counts on the device (eval/metrics.py:7-147, compute_tapvid_metrics for ONE video), so that the
// trajectories of inference_benchmark.py never have to leave the GPU as .npy files to be scored.
// One thread per (query, frame); counts[18] (uint64):
//   [0] evaluated points  [1] occlusion prediction == ground truth  [2] visible in the ground truth
//   [3+3i] within threshold 2^i px & visible  [4+3i] ... & predicted visible (true positives)  [5+3i] false positives
// all restricted to the evaluated frames of the query (strided: every frame but the query frame; first: later frames).
// The distance test replicates the float32 arithmetic of the numpy code: (px*sp - gx*sg)^2 summed, < thresh^2.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tapvid_counts_kernel(const float* __restrict__ pred, const uint8_t* __restrict__ pred_occ,
                                                            const float* __restrict__ gt, const uint8_t* __restrict__ gt_occ,
                                                            const int32_t* __restrict__ qframe, float spx, float spy,
                                                            float sgx, float sgy, int first_mode, int N, int T,
                                                            unsigned long long* __restrict__ counts) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    int c[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) c[k] = 0;
    if (i < (long long)N * T) {
        const int n = (int)(i / T), t = (int)(i - (long long)n * T);
        const int qf = qframe[n];
        const bool ev = first_mode ? (t > qf) : (t != qf);
        if (ev) {
            const bool vis = gt_occ[i] == 0, pvis = pred_occ[i] == 0;
            const float dx = __fsub_rn(__fmul_rn(pred[2 * i], spx), __fmul_rn(gt[2 * i], sgx));
            const float dy = __fsub_rn(__fmul_rn(pred[2 * i + 1], spy), __fmul_rn(gt[2 * i + 1], sgy));
            const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
            c[0] = 1;
            c[1] = (pvis == vis);
            c[2] = vis;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float th = (float)(1 << k);
                const bool within = d2 < th * th;
                c[3 + 3 * k] = within && vis;
                c[4 + 3 * k] = within && vis && pvis;
                c[5 + 3 * k] = ((!vis) && pvis) || ((!within) && pvis);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        const int s = wave_sum_i(c[k]);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&counts[k], (unsigned long long)s);
    }
}

extern "C" int dtk_tapvid_counts(const float* pred_tracks, const uint8_t* pred_occluded, const float* gt_tracks,
                                 const uint8_t* gt_occluded, const int32_t* query_frame, float pred_scale_x,
                                 float pred_scale_y, float gt_scale_x, float gt_scale_y, int first_mode, int N, int T,
                                 unsigned long long* counts18, void* stream) {
    DTK_REQUIRE(pred_tracks && pred_occluded && gt_tracks && gt_occluded && query_frame && counts18, "dtk_tapvid_counts: null pointer");
    DTK_REQUIRE(N >= 0 && T > 0, "dtk_tapvid_counts: bad sizes");
    hipStream_t st = dtk_stream(stream);
    DTK_HIP(hipMemsetAsync(counts18, 0, 18 * sizeof(unsigned long long), st));
    if (N == 0) return DTK_OK;
    DTK_LAUNCH("tapvid_counts", tapvid_counts_kernel, dim3(dtk_cdiv((long long)N * T, 256)), dim3(256), 0, st, pred_tracks,
               pred_occluded, gt_tracks, gt_occluded, query_frame, pred_scale_x, pred_scale_y, gt_scale_x, gt_scale_y,
               first_mode, N, T, counts18);
    return DTK_OK;
}
