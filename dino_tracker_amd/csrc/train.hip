// train.hip -- kernels of the per-video test-time training step.
//
// Train-mode BatchNorm2d of the Delta-DINO CNN (models/networks/delta_dino.py:38: nn.BatchNorm2d after every conv, batches
// of <= 8 frames, tracker.py:118-124), forward and backward, with the ReLU that follows three of the four layers fused in.
// HBM-bound: forward reads x twice and writes y once, backward reads x and dy twice and writes dx once.
//
// Why hand-written: the batch statistics decide every activation of the layer, and the channels of this CNN have means
// that are large against their spread.  A one-pass E[x^2] - E[x]^2 in float32 loses the variance there (measured on this
// box with the library BatchNorm behind torch.nn.BatchNorm2d: 5e-3 relative error in the layer's OUTPUT, 9e-2 in its
// weight gradient, profiles/r02_train_grad_check.txt).  Here every thread reduces groups of eight values exactly
// (mean, then centred squares) and groups, lanes, waves and workgroups are merged pairwise with Chan's update
//     n = na + nb,  d = mb - ma,  m = ma + d nb / n,  M2 = M2a + M2b + d^2 na nb / n
// so the variance never is a difference of large numbers.
//
// Layout: x, y, dy, dx are [N][C][HW] float32 (NCHW, contiguous).  Grid (S, C): workgroup (s, c) owns slice s of channel
// c's N * HW values; S is chosen so that S * C fills the chip, and is at most 64 (one partial per lane in the merge).
#include "common.h"

namespace {

struct Moments {
    float n, mean, m2;
};

__device__ __forceinline__ Moments merge(Moments a, Moments b) {
    const float n = a.n + b.n;
    if (n == 0.f) return a;
    const float d = b.mean - a.mean;
    const float fb = b.n / n;
    Moments r;
    r.n = n;
    r.mean = a.mean + d * fb;
    r.m2 = a.m2 + b.m2 + d * d * a.n * fb;
    return r;
}

__device__ __forceinline__ Moments wave_merge(Moments m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Moments other;
        other.n = __shfl_xor(m.n, o, WAVE);
        other.mean = __shfl_xor(m.mean, o, WAVE);
        other.m2 = __shfl_xor(m.m2, o, WAVE);
        m = merge(m, other);
    }
    // merge(a, b) is not symmetric in rounding, so the lanes of a butterfly can end a few ulps apart: every lane takes
    // lane 0's result, so that the forward's normalisation, the saved statistics and the backward's recomputed ReLU mask
    // all use bit-identical moments
    m.n = __shfl(m.n, 0, WAVE);
    m.mean = __shfl(m.mean, 0, WAVE);
    m.m2 = __shfl(m.m2, 0, WAVE);
    return m;
}

// slice s of channel c: elements [lo, hi) of the channel's N * HW values, in groups of G
constexpr int G = 8;
__device__ __forceinline__ void slice_bounds(long long L, int S, int s, long long* lo, long long* hi) {
    const long long per = ((L + S - 1) / S + G - 1) / G * G;
    *lo = min(L, per * s);
    *hi = min(L, per * (s + 1));
}
// walks the values of channel c: element e of the channel = frame e / HW, pixel e % HW (one division at the start, then
// additions)
struct Cursor {
    size_t addr;  // offset of the current element in the NCHW tensor
    int r;        // pixel inside the frame
    int HW;
    size_t jump;  // (C - 1) * HW: from the end of one frame's plane to the start of the next frame's
    __device__ __forceinline__ Cursor(long long e, int c, int C, int HW_) : HW(HW_) {
        const long long n = e / HW_;
        r = (int)(e - n * HW_);
        addr = ((size_t)n * C + c) * HW_ + r;
        jump = (size_t)(C - 1) * HW_;
    }
    __device__ __forceinline__ void advance(int k) {  // k < HW is not required
        r += k;
        addr += k;
        while (r >= HW) { r -= HW; addr += jump; }
    }
};

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int N, int C,
                                                       int HW, int S) {
    __shared__ Moments sm[4];
    const int c = blockIdx.y, s = blockIdx.x;
    const long long L = (long long)N * HW;
    long long lo, hi;
    slice_bounds(L, S, s, &lo, &hi);
    Moments acc = {0.f, 0.f, 0.f};
    // a thread's group of eight = eight values 256 apart: every load of the wave is one contiguous 256-byte row
    Cursor cur(min(lo + threadIdx.x, L - 1), c, C, HW);
    for (long long e0 = lo + threadIdx.x; e0 < hi; e0 += 256 * G) {
        float v[G];
        int cnt = 0;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const bool ok = e0 + 256 * k < hi;
            v[k] = ok ? x[cur.addr] : 0.f;
            sum += v[k];
            cnt += ok;
            cur.advance(256);
        }
        Moments g;
        g.n = (float)cnt;
        g.mean = sum / g.n;
        g.m2 = 0.f;
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const float d = (e0 + 256 * k < hi) ? v[k] - g.mean : 0.f;
            g.m2 += d * d;
        }
        acc = merge(acc, g);
    }
    acc = wave_merge(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        Moments t = merge(merge(sm[0], sm[1]), merge(sm[2], sm[3]));
        float* p = part + ((size_t)c * S + s) * 3;
        p[0] = t.n; p[1] = t.mean; p[2] = t.m2;
    }
}

// merges the S partials of channel c (S <= 64): every lane of wave 0 ends with the channel's moments
__device__ __forceinline__ Moments channel_moments(const float* __restrict__ part, int c, int S) {
    const int lane = threadIdx.x & 63;
    Moments m = {0.f, 0.f, 0.f};
    if (lane < S) {
        const float* p = part + ((size_t)c * S + lane) * 3;
        m.n = p[0]; m.mean = p[1]; m.m2 = p[2];
    }
    return wave_merge(m);
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ running_mean, float* __restrict__ running_var,
                                                       const float* __restrict__ pre_bias, float momentum, float eps, int relu,
                                                       float* __restrict__ y, float* __restrict__ save_mean,
                                                       float* __restrict__ save_rstd, int N, int C, int HW, int S) {
    const int c = blockIdx.y, s = blockIdx.x;
    const Moments m = channel_moments(part, c, S);  // every wave recomputes it: S <= 64 loads, no barrier
    const float var = m.m2 / m.n;
    const float rstd = 1.f / sqrtf(var + eps);
    if (s == 0 && threadIdx.x == 0) {
        save_mean[c] = m.mean;
        save_rstd[c] = rstd;
        // a per-channel bias added in front of the layer (the conv's) shifts the batch mean and nothing else
        const float shift = pre_bias ? pre_bias[c] : 0.f;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (m.mean + shift);
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (m.m2 / fmaxf(m.n - 1.f, 1.f));
    }
    const float a = rstd * gamma[c], b = beta[c], mu = m.mean;  // y = (x - mu) a + b: the subtraction first (exact for x near mu)
    const long long L = (long long)N * HW;
    long long lo, hi;
    slice_bounds(L, S, s, &lo, &hi);
    Cursor cur(min(lo + threadIdx.x, L - 1), c, C, HW);
    for (long long e = lo + threadIdx.x; e < hi; e += 256, cur.advance(256)) {
        const float v = fmaf(x[cur.addr] - mu, a, b);
        y[cur.addr] = relu ? fmaxf(v, 0.f) : v;
    }
}

// backward, pass 1: per slice  sum(dy'), sum(dy' xhat)  with dy' = dy [y > 0] when the ReLU is fused
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_rstd, int relu,
                                                          float* __restrict__ part, int N, int C, int HW, int S) {
    __shared__ float sm[4][3];
    const int c = blockIdx.y, s = blockIdx.x;
    const float mean = save_mean[c], rstd = save_rstd[c], ga = gamma[c], be = beta[c];
    const float fa = rstd * ga;  // y = (x - mean) fa + beta in bn_apply_kernel
    const long long L = (long long)N * HW;
    long long lo, hi;
    slice_bounds(L, S, s, &lo, &hi);
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    Cursor cur(min(lo + threadIdx.x, L - 1), c, C, HW);
    for (long long e = lo + threadIdx.x; e < hi; e += 256, cur.advance(256)) {
        const size_t ad = cur.addr;
        const float xv = x[ad];
        const float xh = (xv - mean) * rstd;
        float g = dy[ad];
        if (relu && !(fmaf(xv - mean, fa, be) > 0.f)) g = 0.f;  // the forward's own expression: same mask to the bit
        s1 += g;
        s2 = fmaf(g, xh, s2);
        s3 += xh;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    s3 = wave_sum(s3);
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = s1; sm[threadIdx.x >> 6][1] = s2; sm[threadIdx.x >> 6][2] = s3; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* p = part + ((size_t)c * S + s) * 3;
        p[0] = (sm[0][0] + sm[1][0]) + (sm[2][0] + sm[3][0]);
        p[1] = (sm[0][1] + sm[1][1]) + (sm[2][1] + sm[3][1]);
        p[2] = (sm[0][2] + sm[1][2]) + (sm[2][2] + sm[3][2]);
    }
}

// backward, pass 2:  dx = gamma rstd (dy' - mean(dy') - xhat mean(dy' xhat));  dgamma = sum(dy' xhat), dbeta = sum(dy')
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_rstd, int relu,
                                                           const float* __restrict__ part, float* __restrict__ dx,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ dpre_bias, int N, int C, int HW, int S) {
    const int c = blockIdx.y, s = blockIdx.x;
    const int lane = threadIdx.x & 63;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (lane < S) {
        const float* p = part + ((size_t)c * S + lane) * 3;
        s1 = p[0]; s2 = p[1]; s3 = p[2];
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    s3 = wave_sum(s3);
    if (s == 0 && threadIdx.x == 0) {
        dbeta[c] = s1;
        dgamma[c] = s2;
        // gradient of a bias added in front of the layer = sum of dx = gamma rstd (s1 - L s1/L - s2/L sum(xhat)): zero in
        // exact arithmetic (the batch mean is subtracted), in float32 the rounding residue of sum(xhat)
        if (dpre_bias) dpre_bias[c] = -gamma[c] * save_rstd[c] * (s2 / (float)((long long)N * HW)) * s3;
    }
    const float mean = save_mean[c], rstd = save_rstd[c], ga = gamma[c], be = beta[c];
    const float fa = rstd * ga;
    const long long L = (long long)N * HW;
    const float inv = 1.f / (float)L;
    const float k = ga * rstd, m1 = s1 * inv, m2 = s2 * inv;
    long long lo, hi;
    slice_bounds(L, S, s, &lo, &hi);
    Cursor cur(min(lo + threadIdx.x, L - 1), c, C, HW);
    for (long long e = lo + threadIdx.x; e < hi; e += 256, cur.advance(256)) {
        const size_t ad = cur.addr;
        const float xv = x[ad];
        const float xh = (xv - mean) * rstd;
        float g = dy[ad];
        if (relu && !(fmaf(xv - mean, fa, be) > 0.f)) g = 0.f;
        dx[ad] = k * (g - m1 - xh * m2);
    }
}

// ---- BlurPool (antialiased_cnns.BlurPool, filt_size 4, stride 2, reflect padding (1, 2, 1, 2)) ----------------------------
// y[o] = sum_i f[i] x[refl(2 o + i - 1)] per axis, f = [1, 3, 3, 1] / 8, refl(-1) = 1, refl(n) = n - 2, refl(n + 1) = n - 3.
// One thread per output (forward) / per input (backward: the transposed operator as a gather -- an input position is read
// by at most two outputs directly and, within three cells of the far border or at index 1, by the reflected taps too).
__device__ __forceinline__ int refl(int p, int n) { return p < 0 ? -p : (p >= n ? 2 * n - 2 - p : p); }
__device__ __forceinline__ float blur_tap(int i) { return (i == 0 || i == 3) ? 0.125f : 0.375f; }

__global__ __launch_bounds__(256) void blurpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes,
                                                           int H, int W, int Ho, int Wo) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * Ho * Wo) return;
    const int ox = (int)(idx % Wo);
    const long long t = idx / Wo;
    const int oy = (int)(t % Ho);
    const float* xp = x + (t / Ho) * (long long)H * W;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float* row = xp + (long long)refl(2 * oy + i - 1, H) * W;
        float r = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) r = fmaf(blur_tap(j), row[refl(2 * ox + j - 1, W)], r);
        acc = fmaf(blur_tap(i), r, acc);
    }
    y[idx] = acc;
}

// the outputs o and weights through which input position p (of n, output length no) is read: at most 6 entries
__device__ __forceinline__ int blur_adjoint(int p, int n, int no, int (&o)[6], float (&wgt)[6]) {
    int cnt = 0;
    auto add = [&](int q) {  // padded-axis position q in [-1, n + 1] that maps onto p
        // q = 2 o + i - 1, i in 0..3  ->  o in [(q - 2) / 2, (q + 1) / 2]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int num = q + 1 - i;
            if (num >= 0 && !(num & 1) && (num >> 1) < no) { o[cnt] = num >> 1; wgt[cnt] = blur_tap(i); ++cnt; }
        }
    };
    add(p);
    if (p == 1) add(-1);
    if (p == n - 2) add(n);
    if (p == n - 3) add(n + 1);
    return cnt;
}

__global__ __launch_bounds__(256) void blurpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long long planes,
                                                           int H, int W, int Ho, int Wo) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * H * W) return;
    const int px = (int)(idx % W);
    const long long t = idx / W;
    const int py = (int)(t % H);
    const float* gp = dy + (t / H) * (long long)Ho * Wo;
    int oy[6], ox[6];
    float wy[6], wx[6];
    const int ny = blur_adjoint(py, H, Ho, oy, wy), nx = blur_adjoint(px, W, Wo, ox, wx);
    float acc = 0.f;
    for (int a = 0; a < ny; ++a) {
        float r = 0.f;
        for (int b = 0; b < nx; ++b) r = fmaf(wx[b], gp[(long long)oy[a] * Wo + ox[b]], r);
        acc = fmaf(wy[a], r, acc);
    }
    dx[idx] = acc;
}

int slices(int N, int C, int HW) {
    const long long L = (long long)N * HW;
    long long S = 4096 / (C > 0 ? C : 1);  // ~16 workgroups per CU over all channels
    const long long cap = (L + 2047) / 2048;  // at least 2048 values per workgroup
    if (S > cap) S = cap;
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    return (int)S;
}

}  // namespace

extern "C" size_t dtk_batchnorm_workspace_bytes(int32_t C) { return (size_t)(C > 0 ? C : 0) * 64 * 3 * sizeof(float); }

extern "C" int dtk_batchnorm_train_forward(const float* x, const float* gamma, const float* beta, const float* pre_bias,
                                           float* running_mean, float* running_var, float momentum, float eps, int32_t relu,
                                           float* y,
                                           float* save_mean, float* save_rstd, int32_t N, int32_t C, int32_t HW,
                                           void* workspace, size_t workspace_bytes, void* stream) {
    DTK_REQUIRE(x && gamma && beta && y && save_mean && save_rstd && workspace, "dtk_batchnorm_train_forward: null pointer");
    DTK_REQUIRE(N > 0 && C > 0 && HW > 0 && (long long)N * HW > 1, "dtk_batchnorm_train_forward: bad shape %d x %d x %d", N, C, HW);
    DTK_REQUIRE(workspace_bytes >= dtk_batchnorm_workspace_bytes(C), "dtk_batchnorm_train_forward: workspace too small");
    hipStream_t st = dtk_stream(stream);
    const int S = slices(N, C, HW);
    float* part = static_cast<float*>(workspace);
    DTK_LAUNCH("bn_stats", bn_stats_kernel, dim3(S, C), dim3(256), 0, st, x, part, N, C, HW, S);
    DTK_LAUNCH("bn_apply", bn_apply_kernel, dim3(S, C), dim3(256), 0, st, x, part, gamma, beta, running_mean, running_var,
               pre_bias, momentum, eps, relu, y, save_mean, save_rstd, N, C, HW, S);
    return DTK_OK;
}

extern "C" int dtk_batchnorm_train_backward(const float* x, const float* dy, const float* gamma, const float* beta,
                                            const float* save_mean, const float* save_rstd, int32_t relu, float* dx,
                                            float* dgamma, float* dbeta, float* dpre_bias, int32_t N, int32_t C, int32_t HW,
                                            void* workspace, size_t workspace_bytes, void* stream) {
    DTK_REQUIRE(x && dy && gamma && beta && save_mean && save_rstd && dx && dgamma && dbeta && workspace,
                "dtk_batchnorm_train_backward: null pointer");
    DTK_REQUIRE(N > 0 && C > 0 && HW > 0, "dtk_batchnorm_train_backward: bad shape %d x %d x %d", N, C, HW);
    DTK_REQUIRE(workspace_bytes >= dtk_batchnorm_workspace_bytes(C), "dtk_batchnorm_train_backward: workspace too small");
    hipStream_t st = dtk_stream(stream);
    const int S = slices(N, C, HW);
    float* part = static_cast<float*>(workspace);
    DTK_LAUNCH("bn_bwd_sums", bn_bwd_sums_kernel, dim3(S, C), dim3(256), 0, st, x, dy, gamma, beta, save_mean, save_rstd, relu,
               part, N, C, HW, S);
    DTK_LAUNCH("bn_bwd_apply", bn_bwd_apply_kernel, dim3(S, C), dim3(256), 0, st, x, dy, gamma, beta, save_mean, save_rstd, relu,
               part, dx, dgamma, dbeta, dpre_bias, N, C, HW, S);
    return DTK_OK;
}

extern "C" int dtk_blurpool_forward(const float* x, float* y, int64_t planes, int32_t H, int32_t W, void* stream) {
    DTK_REQUIRE(x && y, "dtk_blurpool_forward: null pointer");
    DTK_REQUIRE(planes > 0 && H >= 4 && W >= 4, "dtk_blurpool_forward: bad shape %lld x %d x %d (reflection needs >= 4)", (long long)planes, H, W);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)planes * Ho * Wo;
    DTK_LAUNCH("blurpool_fwd", blurpool_fwd_kernel, dim3(dtk_cdiv(total, 256)), dim3(256), 0, dtk_stream(stream), x, y,
               (long long)planes, H, W, Ho, Wo);
    return DTK_OK;
}

extern "C" int dtk_blurpool_backward(const float* dy, float* dx, int64_t planes, int32_t H, int32_t W, void* stream) {
    DTK_REQUIRE(dy && dx, "dtk_blurpool_backward: null pointer");
    DTK_REQUIRE(planes > 0 && H >= 4 && W >= 4, "dtk_blurpool_backward: bad shape %lld x %d x %d", (long long)planes, H, W);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)planes * H * W;
    DTK_LAUNCH("blurpool_bwd", blurpool_bwd_kernel, dim3(dtk_cdiv(total, 256)), dim3(256), 0, dtk_stream(stream), dy, dx,
               (long long)planes, H, W, Ho, Wo);
    return DTK_OK;
}
