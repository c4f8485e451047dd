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
#include <cstdlib>
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

__device__ __forceinline__ float blur_fwd_one(const float* __restrict__ xp, int oy, int ox, int H, int W) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float* row = xp + (long long)refl(2 * oy + i - 1, H) * W;
        float r = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) r = fmaf(blur_tap(j), row[refl(2 * ox + j - 1, W)], r);
        acc = fmaf(blur_tap(i), r, acc);
    }
    return acc;
}

__global__ __launch_bounds__(256) void blurpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes,
                                                           int H, int W, int Ho, int Wo) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * Ho * Wo) return;
    const int ox = (int)(idx % Wo);
    const long long t = idx / Wo;
    const int oy = (int)(t % Ho);
    y[idx] = blur_fwd_one(x + (t / Ho) * (long long)H * W, oy, ox, H, W);
}

// Round 6: a thread owns output column ox of four consecutive output rows 4 i .. 4 i + 3, INTERIOR tiles only (no reflected tap): the
// ten input rows are combined along x once (four loads each) and every output is a four-term combination of four of those row values
// -- 10 loads per output where the per-output form issues 16 behind two 64-bit divisions (it ran at 2.0 TB/s: 0.51 ms for the first
// layer).  Same products and the same order of additions as blur_fwd_one.  The outputs outside the interior rectangle go to
// blurpool_fwd_border_kernel.
__global__ __launch_bounds__(256) void blurpool_fwd4_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int Ho, int Wo,
                                                            int oxhi, int ihi) {
    const int ox = 1 + blockIdx.x * 64 + (threadIdx.x & 63), i = 1 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= oxhi || i >= ihi) return;
    const float* xp = x + (long long)blockIdx.z * H * W + 2 * ox - 1;
    float* yp = y + (long long)blockIdx.z * Ho * Wo + ox;
    float r[10];
#pragma unroll
    for (int a = 0; a < 10; ++a) {
        const float* row = xp + (long long)(8 * i - 1 + a) * W;
        r[a] = fmaf(0.125f, row[3], fmaf(0.375f, row[2], fmaf(0.375f, row[1], 0.125f * row[0])));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        yp[(long long)(4 * i + k) * Wo] = fmaf(0.125f, r[2 * k + 3], fmaf(0.375f, r[2 * k + 2], fmaf(0.375f, r[2 * k + 1], 0.125f * r[2 * k])));
}

// the outputs OUTSIDE the rectangle ox in [x0, x1), oy in [y0, y1): left strip, right strip, then top and bottom between them
__global__ __launch_bounds__(256) void blurpool_fwd_border_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int Ho,
                                                                  int Wo, int x0, int x1, int y0, int y1) {
    int t = blockIdx.x * 256 + threadIdx.x;
    const int nl = x0 * Ho, nr = (Wo - x1) * Ho, nt = (x1 - x0) * y0, nb = (x1 - x0) * (Ho - y1);
    int ox, oy;
    if (t < nl) { ox = t % x0; oy = t / x0; }
    else if ((t -= nl) < nr) { ox = x1 + t % (Wo - x1); oy = t / (Wo - x1); }
    else if ((t -= nr) < nt) { ox = x0 + t % (x1 - x0); oy = t / (x1 - x0); }
    else if ((t -= nt) < nb) { ox = x0 + t % (x1 - x0); oy = y1 + t / (x1 - x0); }
    else return;
    y[(long long)blockIdx.z * Ho * Wo + (long long)oy * Wo + ox] = blur_fwd_one(x + (long long)blockIdx.z * H * W, oy, ox, H, W);
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

// The same operator with one thread per 2 x 2 block of inputs (positions 2m, 2m + 1 per axis): away from the borders
//     dx[2m] = (3 dy[m] + dy[m - 1]) / 8,   dx[2m + 1] = (dy[m + 1] + 3 dy[m]) / 8     per axis,
// i.e. nine loads and four stores without the tap lists and the 64-bit index divisions of the per-element form (which made the
// kernel 4-7x slower than its traffic: 1.8 ms for the 832 MB of the first layer's gradient); blocks that touch index 1 or the
// last three indices of an axis take the per-element path.
// one 2 x 2 block of inputs (positions 2 mx, 2 mx + 1 and 2 my, 2 my + 1 of plane `gp` -> `dp`)
__device__ __forceinline__ void blur_bwd_block(const float* __restrict__ gp, float* __restrict__ dp, int mx, int my, int H, int W, int Ho,
                                               int Wo, int vec) {
    if (2 * mx >= W || 2 * my >= H) return;
    const bool inx = mx >= 1 && 2 * mx + 1 <= W - 4, iny = my >= 1 && 2 * my + 1 <= H - 4;
    if (inx && iny) {
        float re[3], ro[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float* row = gp + (long long)(my - 1 + a) * Wo + mx - 1;
            const float g0 = row[0], g1 = row[1], g2 = row[2];
            re[a] = fmaf(0.375f, g1, 0.125f * g0);
            ro[a] = fmaf(0.125f, g2, 0.375f * g1);
        }
        const float e0 = fmaf(0.375f, re[1], 0.125f * re[0]), o0 = fmaf(0.375f, ro[1], 0.125f * ro[0]);
        const float e1 = fmaf(0.125f, re[2], 0.375f * re[1]), o1 = fmaf(0.125f, ro[2], 0.375f * ro[1]);
        float* r0 = dp + (long long)(2 * my) * W + 2 * mx;
        if (vec) {
            *reinterpret_cast<float2*>(r0) = float2{e0, o0};
            *reinterpret_cast<float2*>(r0 + W) = float2{e1, o1};
        } else {
            r0[0] = e0; r0[1] = o0;
            r0[W] = e1; r0[W + 1] = o1;
        }
        return;
    }
    for (int sy = 0; sy < 2; ++sy)
        for (int sx = 0; sx < 2; ++sx) {
            const int py = 2 * my + sy, px = 2 * mx + sx;
            if (py >= H || px >= W) continue;
            int oy[6], ox[6];
            float wy[6], wx[6];
            const int ny = blur_adjoint(py, H, Ho, oy, wy), nx = blur_adjoint(px, W, Wo, ox, wx);
            float acc = 0.f;
            for (int a = 0; a < ny; ++a) {
                float r = 0.f;
                for (int b = 0; b < nx; ++b) r = fmaf(wx[b], gp[(long long)oy[a] * Wo + ox[b]], r);
                acc = fmaf(wy[a], r, acc);
            }
            dp[(long long)py * W + px] = acc;
        }
}

__global__ __launch_bounds__(256) void blurpool_bwd2_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W,
                                                            int Ho, int Wo, int vec) {
    const int mx = blockIdx.x * 64 + (threadIdx.x & 63), my = blockIdx.y * 4 + (threadIdx.x >> 6);
    blur_bwd_block(dy + (long long)blockIdx.z * Ho * Wo, dx + (long long)blockIdx.z * H * W, mx, my, H, W, Ho, Wo, vec);
}

// Round 6: a thread owns 2 x 4 such blocks (inputs 4 j .. 4 j + 3 of rows 8 i .. 8 i + 7), INTERIOR tiles only (j in [jlo, jhi), i in
// [ilo, ihi): every block of the tile is away from the borders).  The six rows of dy it needs are combined along x once -- four values
// per row, (even, odd) of both blocks -- and every row of dx is a two-term combination of two neighbouring row results: 24 loads and
// 16 8-byte stores for 32 outputs where the per-block form issues 72 and 16 (the one-block kernel ran at a fifth of its traffic: 0.98
// ms for the 1.04 GB of the first layer's gradient).  The blocks outside the interior rectangle go to blurpool_bwd_border_kernel (the
// per-block routine over the four border strips): with both in one kernel the lane that owns a border tile held its whole wave for
// eight slow blocks (measured: 2.6 ms).  Needs W even (8-byte aligned row pairs: `vec` of the caller).
template <bool VEC>   // VEC: W and H W even, every row pair starts 8-byte aligned; otherwise 4-byte stores
__global__ __launch_bounds__(256) void blurpool_bwd8_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W,
                                                            int Ho, int Wo, int jlo, int jhi, int ilo, int ihi) {
    const int j = jlo + blockIdx.x * 64 + (threadIdx.x & 63), i = ilo + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (j >= jhi || i >= ihi) return;
    const int mx0 = 2 * j, my0 = 4 * i;
    const float* gp = dy + (long long)blockIdx.z * Ho * Wo;
    float* dp = dx + (long long)blockIdx.z * H * W;
    float4 prev, cur, nxt;   // x-combined rows my - 1, my, my + 1: (even, odd) of block mx0, (even, odd) of block mx0 + 1
    auto xrow = [&](int a) {
        const float* row = gp + (long long)a * Wo + mx0 - 1;
        const float g0 = row[0], g1 = row[1], g2 = row[2], g3 = row[3];
        return float4{fmaf(0.375f, g1, 0.125f * g0), fmaf(0.125f, g2, 0.375f * g1), fmaf(0.375f, g2, 0.125f * g1), fmaf(0.125f, g3, 0.375f * g2)};
    };
    prev = xrow(my0 - 1);
    cur = xrow(my0);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        nxt = xrow(my0 + b + 1);
        float* r0 = dp + (long long)(2 * (my0 + b)) * W + 4 * j;
        const float4 e = {fmaf(0.375f, cur.x, 0.125f * prev.x), fmaf(0.375f, cur.y, 0.125f * prev.y), fmaf(0.375f, cur.z, 0.125f * prev.z),
                          fmaf(0.375f, cur.w, 0.125f * prev.w)};
        const float4 o = {fmaf(0.125f, nxt.x, 0.375f * cur.x), fmaf(0.125f, nxt.y, 0.375f * cur.y), fmaf(0.125f, nxt.z, 0.375f * cur.z),
                          fmaf(0.125f, nxt.w, 0.375f * cur.w)};
        if (VEC) {
            *reinterpret_cast<float2*>(r0) = float2{e.x, e.y};
            *reinterpret_cast<float2*>(r0 + 2) = float2{e.z, e.w};
            *reinterpret_cast<float2*>(r0 + W) = float2{o.x, o.y};
            *reinterpret_cast<float2*>(r0 + W + 2) = float2{o.z, o.w};
        } else {
            r0[0] = e.x; r0[1] = e.y; r0[2] = e.z; r0[3] = e.w;
            r0[W] = o.x; r0[W + 1] = o.y; r0[W + 2] = o.z; r0[W + 3] = o.w;
        }
        prev = cur;
        cur = nxt;
    }
}

// the 2 x 2 blocks OUTSIDE the rectangle mx in [x0, x1), my in [y0, y1) of an nbx x nby grid of blocks: left strip, right strip, then the
// top and bottom strips between them; one thread per block
__global__ __launch_bounds__(256) void blurpool_bwd_border_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W,
                                                                  int Ho, int Wo, int nbx, int nby, int x0, int x1, int y0, int y1, int vec) {
    int t = blockIdx.x * 256 + threadIdx.x;
    const int nl = x0 * nby, nr = (nbx - x1) * nby, nt = (x1 - x0) * y0, nb = (x1 - x0) * (nby - y1);
    int mx, my;
    if (t < nl) { mx = t % x0; my = t / x0; }
    else if ((t -= nl) < nr) { mx = x1 + t % (nbx - x1); my = t / (nbx - x1); }
    else if ((t -= nr) < nt) { mx = x0 + t % (x1 - x0); my = t / (x1 - x0); }
    else if ((t -= nt) < nb) { mx = x0 + t % (x1 - x0); my = y1 + t / (x1 - x0); }
    else return;
    blur_bwd_block(dy + (long long)blockIdx.z * Ho * Wo, dx + (long long)blockIdx.z * H * W, mx, my, H, W, Ho, Wo, vec);
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
    const int oxhi = (W - 3) / 2 + 1, ihi = (H - 9) / 8 + 1;   // interior: ox in [1, oxhi), row groups i in [1, ihi) (rows 4 i .. 4 i + 3)
    if (planes <= 65535 && oxhi > 1 && ihi > 1) {
        DTK_LAUNCH("blurpool_fwd", blurpool_fwd4_kernel, dim3(dtk_cdiv(oxhi - 1, 64), dtk_cdiv(ihi - 1, 4), (unsigned)planes), dim3(256), 0,
                   dtk_stream(stream), x, y, H, W, Ho, Wo, oxhi, ihi);
        const int x0 = 1, x1 = oxhi, y0 = 4, y1 = 4 * ihi;
        const int nborder = x0 * Ho + (Wo - x1) * Ho + (x1 - x0) * y0 + (x1 - x0) * (Ho - y1);
        DTK_LAUNCH("blurpool_fwd_border", blurpool_fwd_border_kernel, dim3(dtk_cdiv(nborder, 256), 1, (unsigned)planes), dim3(256), 0,
                   dtk_stream(stream), x, y, H, W, Ho, Wo, x0, x1, y0, y1);
        return DTK_OK;
    }
    DTK_LAUNCH("blurpool_fwd", blurpool_fwd_kernel, dim3(dtk_cdiv(total, 256)), dim3(256), 0, dtk_stream(stream), x, y,
               (long long)planes, H, W, Ho, Wo);
    return DTK_OK;
}

extern "C" int dtk_blurpool_backward(const float* dy, float* dx, int64_t planes, int32_t H, int32_t W, void* stream) {
    DTK_REQUIRE(dy && dx, "dtk_blurpool_backward: null pointer");
    DTK_REQUIRE(planes > 0 && H >= 4 && W >= 4, "dtk_blurpool_backward: bad shape %lld x %d x %d", (long long)planes, H, W);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (planes <= 65535) {
        const int vec = (((long long)H * W) % 2 == 0 && W % 2 == 0) ? 1 : 0;  // every row pair starts 8-byte aligned
        const int jhi = (W - 7) / 4 + 1, ihi = (H - 11) / 8 + 1;   // interior tiles: j in [1, jhi), i in [1, ihi)
        if (jhi > 1 && ihi > 1) {
            const int nbx = (W + 1) / 2, nby = (H + 1) / 2;
            const dim3 grid(dtk_cdiv(jhi - 1, 64), dtk_cdiv(ihi - 1, 4), (unsigned)planes);
            if (vec) DTK_LAUNCH("blurpool_bwd", blurpool_bwd8_kernel<true>, grid, dim3(256), 0, dtk_stream(stream), dy, dx, H, W, Ho, Wo, 1, jhi, 1, ihi);
            else DTK_LAUNCH("blurpool_bwd", blurpool_bwd8_kernel<false>, grid, dim3(256), 0, dtk_stream(stream), dy, dx, H, W, Ho, Wo, 1, jhi, 1, ihi);
            const int x0 = 2, x1 = 2 * jhi, y0 = 4, y1 = 4 * ihi;
            const int nborder = x0 * nby + (nbx - x1) * nby + (x1 - x0) * y0 + (x1 - x0) * (nby - y1);
            DTK_LAUNCH("blurpool_bwd_border", blurpool_bwd_border_kernel, dim3(dtk_cdiv(nborder, 256), 1, (unsigned)planes), dim3(256), 0,
                       dtk_stream(stream), dy, dx, H, W, Ho, Wo, nbx, nby, x0, x1, y0, y1, vec);
            return DTK_OK;
        }
        DTK_LAUNCH("blurpool_bwd", blurpool_bwd2_kernel, dim3(dtk_cdiv((W + 1) / 2, 64), dtk_cdiv((H + 1) / 2, 4), (unsigned)planes),
                   dim3(256), 0, dtk_stream(stream), dy, dx, H, W, Ho, Wo, vec);
        return DTK_OK;
    }
    const long long total = (long long)planes * H * W;
    DTK_LAUNCH("blurpool_bwd", blurpool_bwd_kernel, dim3(dtk_cdiv(total, 256)), dim3(256), 0, dtk_stream(stream), dy, dx,
               (long long)planes, H, W, Ho, Wo);
    return DTK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Convolutions of the training step as matrix products on the fp16 matrix cores, fp32-grade (round 3).
//
// Round 2 ran every convolution of the Delta-DINO CNN (forward, data gradient, weight gradient; delta_dino.py:29-31 under
// autograd, dino_tracker.py:392-448) as ATen im2col / col2im + an fp32 library GEMM: 40 % of the iteration's kernel time at
// fp32 vector rates.  Here:
//   gemm_nt_split_kernel   C[m][n] (+)= sum_k A[m][k] B[n][k], both operands fp32 with the reduction index contiguous, split
//                          x = hi + lo into fp16 halves WHILE they are staged into LDS (after a power-of-two operand scale
//                          that keeps the halves normal numbers), product = hi.hi + hi.lo + lo.hi on MFMA 16x16x32 f16 with
//                          fp32 accumulation -- the scheme of delta_dino.hip / refine_corr (2^-22 relative per operand,
//                          tests/test_numeric_claims.py) -- 128 x 128 tiles, batched, optional split of the reduction over
//                          workgroups (atomic accumulation) for the weight gradient whose reduction runs over all pixels.
//   im2col_kernel          unfolded operand of a stride-1 k x k convolution with reflect (or zero) padding and dilation, in
//                          either of the two layouts the three products need (below); col2im_kernel its adjoint as a GATHER
//                          (incl. the adjoint of the reflect padding: a pixel near the border also collects what its mirror
//                          images in the padding received), transpose_kernel for the operands that are reduction-major.
//   forward        Y[cout][l]   = sum_k W[cout][k]  cols[l][k]          cols  = im2col layout 0  [L][Kp]
//   weight grad    dW[cout][k]  = sum_l dY[cout][l] colsT[k][l]         colsT = im2col layout 1  [Kp][Lp], all frames and
//                                                                       pixel chunks accumulate into one dW
//   data grad      dcols[k][l]  = sum_c Wt[k][c]    dYt[l][c]  -> col2im   (Wt, dYt: transposes)
// ---------------------------------------------------------------------------------------------------------------------
namespace {

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int GT = 128, GKS = 32, GRP = 40;  // tile side, k-step, LDS row pitch in halves (80 B: conflict-free fragments)

template <bool VEC>
__global__ __launch_bounds__(256) void gemm_nt_split_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                            float* __restrict__ C, int M, int N, int K, long long lda,
                                                            long long ldb, long long ldc, long long sA, long long sB,
                                                            long long sC, int split_k, int k_chunk, int accumulate,
                                                            const float* __restrict__ scale_a, const float* __restrict__ scale_b,
                                                            const int32_t* __restrict__ idx_b, const int32_t* __restrict__ idx_c) {
    __shared__ __attribute__((aligned(16))) half_t Ah[GT * GRP], Al[GT * GRP], Bh[GT * GRP], Bl[GT * GRP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int bz = blockIdx.z / split_k, ks = blockIdx.z - bz * split_k;
    A += bz * sA;
    B += (idx_b ? idx_b[bz] : bz) * sB;   // (optional: batch b reads operand B / writes output C number idx[b])
    C += (idx_c ? idx_c[bz] : bz) * sC;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const int k0 = ks * k_chunk, k1 = min(K, k0 + k_chunk);
    if (k0 >= k1) return;
    const float sa = scale_a ? *scale_a : 1.f, sb = scale_b ? *scale_b : 1.f;
    const int wr = w >> 1, wc = w & 1;  // wave tile 64 x 64
    const int fj = lane & 15, fg = lane >> 4;
    // loader: piece p = tid + 256 i (i < 4): row p >> 3 of the tile, floats 4 (p & 7) .. + 3 of the k-step
    float4 ra[4], rb[4];
    auto load = [&](int kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = tid + 256 * i, row = p >> 3, kq = kk + (p & 7) * 4;
            const int am = m0 + row, bn = n0 + row;
            if (VEC) {  // rows 16-byte aligned and K a multiple of 4: whole pieces are inside or outside
                ra[i] = (am < M && kq < k1) ? *reinterpret_cast<const float4*>(A + am * lda + kq) : float4{0.f, 0.f, 0.f, 0.f};
                rb[i] = (bn < N && kq < k1) ? *reinterpret_cast<const float4*>(B + bn * ldb + kq) : float4{0.f, 0.f, 0.f, 0.f};
            } else {
                float a[4], b[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = (am < M && kq + e < k1) ? A[am * lda + kq + e] : 0.f;
                    b[e] = (bn < N && kq + e < k1) ? B[bn * ldb + kq + e] : 0.f;
                }
                ra[i] = float4{a[0], a[1], a[2], a[3]};
                rb[i] = float4{b[0], b[1], b[2], b[3]};
            }
        }
    };
    auto split_store = [&](half_t* hp, half_t* lp, int row, int kq, const float4& v, float s) {
        const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
        h4v hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = (half_t)x[e];
            lo[e] = (half_t)(x[e] - (float)hi[e]);
        }
        *reinterpret_cast<h4v*>(hp + row * GRP + kq) = hi;
        *reinterpret_cast<h4v*>(lp + row * GRP + kq) = lo;
    };
    f4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f4{0.f, 0.f, 0.f, 0.f};
    load(k0);
    for (int kk = k0; kk < k1; kk += GKS) {
        __syncthreads();  // the previous step's fragment reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = tid + 256 * i;
            split_store(Ah, Al, p >> 3, (p & 7) * 4, ra[i], sa);
            split_store(Bh, Bl, p >> 3, (p & 7) * 4, rb[i], sb);
        }
        __syncthreads();
        if (kk + GKS < k1) load(kk + GKS);  // in flight under the MFMAs
        h8 bh[4], bl[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            bh[ni] = *reinterpret_cast<const h8*>(&Bh[(wc * 64 + ni * 16 + fj) * GRP + fg * 8]);
            bl[ni] = *reinterpret_cast<const h8*>(&Bl[(wc * 64 + ni * 16 + fj) * GRP + fg * 8]);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const h8 ah = *reinterpret_cast<const h8*>(&Ah[(wr * 64 + mi * 16 + fj) * GRP + fg * 8]);
            const h8 al = *reinterpret_cast<const h8*>(&Al[(wr * 64 + mi * 16 + fj) * GRP + fg * 8]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                // D[i][j] = sum_k A_op[i][k] B_op[j][k]: first operand = rows m, second = rows n
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[ni], acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ni], acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ni], acc[mi][ni], 0, 0, 0);
            }
        }
    }
    // D fragment: lane (fg, fj) holds rows 4 fg + r (r = 0..3) of column fj of each 16 x 16 tile
    const float inv = 1.f / (sa * sb);
    const bool atomic = split_k > 1 || accumulate == 2;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wc * 64 + ni * 16 + fj;
            if (n >= N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * 64 + mi * 16 + fg * 4 + r;
                if (m >= M) continue;
                float* dst = C + m * ldc + n;
                const float v = acc[mi][ni][r] * inv;
                if (atomic) atomicAdd(dst, v);
                else *dst = accumulate ? *dst + v : v;
            }
        }
}

// reflect-padded source index of padded coordinate q (torch 'reflect': no edge repeat)
__device__ __forceinline__ int reflect_idx(int q, int pad, int n) {
    int s = q - pad;
    if (s < 0) s = -s;
    if (s >= n) s = 2 * (n - 1) - s;
    return s;
}

// one element of the unfolded operand: output pixel l, patch index k (k >= K: zero fill)
__device__ __forceinline__ float im2col_value(const float* __restrict__ xf, long long l, int k, int K, int H, int W, int ks, int pad,
                                              int dil, int reflect) {
    if (k >= K) return 0.f;
    const long long L = (long long)H * W;
    const int oy = (int)(l / W), ox = (int)(l - (long long)oy * W);
    const int c = k / (ks * ks), rem = k - c * ks * ks, ky = rem / ks, kx = rem - ky * ks;
    int sy = oy + ky * dil, sx = ox + kx * dil;  // padded coordinates
    if (reflect) { sy = reflect_idx(sy, pad, H); sx = reflect_idx(sx, pad, W); }
    else {
        sy -= pad; sx -= pad;
        if (sy < 0 || sy >= H || sx < 0 || sx >= W) return 0.f;
    }
    return xf[(long long)c * L + (long long)sy * W + sx];
}

// layout 0: cols[f][l][Kp] -- a workgroup owns 32 consecutive pixels and walks ALL patch indices in tiles of 32 through an
// LDS transpose tile: the frame is READ along the pixels (consecutive lanes = consecutive x of one tap) and the operand is
// WRITTEN along k, 128 bytes per pixel and tile, the 32 rows of a workgroup forming one contiguous 32 x Kp x 4 byte range.
// (First form: one workgroup per 32 x 32 tile = 5 M workgroups of 1 K elements for the second layer: 1.2 TB/s.)
// layout 1: cols[f][Kp][Lp], pixels contiguous on both sides: a workgroup owns 1024 pixels of 8 patch indices, 16-byte stores
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ cols, int C, int H, int W,
                                                     int ks, int pad, int dil, int reflect, int layout, int Kp, long long Lp) {
    const int f = blockIdx.z;
    const long long L = (long long)H * W;
    const float* xf = x + (long long)f * C * L;
    const int K = C * ks * ks;
    if (layout == 0) {
        __shared__ float t[32][33];
        float* cf = cols + (long long)f * L * Kp;
        const long long l0 = (long long)blockIdx.x * 32;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        const long long lr = l0 + tx;
        for (int k0 = 0; k0 < Kp; k0 += 32) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kk = ty + 8 * i;
                t[kk][tx] = lr < L ? im2col_value(xf, lr, k0 + kk, K, H, W, ks, pad, dil, reflect) : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long long l = l0 + ty + 8 * i;
                if (l < L) cf[l * Kp + k0 + tx] = t[tx][ty + 8 * i];
            }
        }
    } else {
        float* cf = cols + (long long)f * Kp * Lp;
        const long long l = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
        if (l >= Lp) return;  // Lp is a multiple of 4
        for (int kk = 0; kk < 8; ++kk) {
            const int k = blockIdx.y * 8 + kk;
            if (k >= Kp) break;
            float4 v;
            v.x = l + 0 < L ? im2col_value(xf, l + 0, k, K, H, W, ks, pad, dil, reflect) : 0.f;
            v.y = l + 1 < L ? im2col_value(xf, l + 1, k, K, H, W, ks, pad, dil, reflect) : 0.f;
            v.z = l + 2 < L ? im2col_value(xf, l + 2, k, K, H, W, ks, pad, dil, reflect) : 0.f;
            v.w = l + 3 < L ? im2col_value(xf, l + 3, k, K, H, W, ks, pad, dil, reflect) : 0.f;
            *reinterpret_cast<float4*>(cf + (long long)k * Lp + l) = v;
        }
    }
}

// adjoint of im2col (layout 1 of the gradient: dcols[f][Kp][L]) as a gather: input pixel (c, y, x) sums, over its pre-images
// (qy, qx) in the padded frame (itself and, within `pad` of a border, its mirror image) and over the k x k taps, the entry of
// the output pixel (qy - ky dil, qx - kx dil) that read it.
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcols, float* __restrict__ dx, int C, int H, int W,
                                                     int ks, int pad, int dil, int reflect, int Kp) {
    const int f = blockIdx.z, c = blockIdx.y;
    const long long L = (long long)H * W;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= L) return;
    const int y = (int)(p / W), xx = (int)(p - (long long)y * W);
    const float* df = dcols + (long long)f * Kp * L;
    int qy[2], qx[2], ny = 1, nx = 1;
    qy[0] = y + pad;
    qx[0] = xx + pad;
    if (reflect) {
        if (y >= 1 && y <= pad) qy[ny++] = pad - y;
        else if (y <= H - 2 && y >= H - 1 - pad) qy[ny++] = pad + 2 * (H - 1) - y;
        if (xx >= 1 && xx <= pad) qx[nx++] = pad - xx;
        else if (xx <= W - 2 && xx >= W - 1 - pad) qx[nx++] = pad + 2 * (W - 1) - xx;
    }
    float s = 0.f;
    for (int iy = 0; iy < ny; ++iy)
        for (int ix = 0; ix < nx; ++ix)
            for (int ky = 0; ky < ks; ++ky) {
                const int oy = qy[iy] - ky * dil;
                if (oy < 0 || oy >= H) continue;  // stride 1, "same" size: the output has H x W pixels
                for (int kx = 0; kx < ks; ++kx) {
                    const int ox = qx[ix] - kx * dil;
                    if (ox < 0 || ox >= W) continue;
                    s += df[(long long)((c * ks + ky) * ks + kx) * L + (long long)oy * W + ox];
                }
            }
    dx[((long long)f * C + c) * L + p] = s;
}

// dst[b][c][r] = src[b][r][c]  (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, long long R, long long Cc) {
    __shared__ float t[32][33];
    const long long b = blockIdx.z;
    const float* s = src + b * R * Cc;
    float* d = dst + b * R * Cc;
    const long long r0 = (long long)blockIdx.y * 32, c0 = (long long)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long r = r0 + ty + 8 * i, c = c0 + tx;
        if (r < R && c < Cc) t[ty + 8 * i][tx] = s[r * Cc + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < R && c < Cc) d[c * R + r] = t[tx][ty + 8 * i];
    }
}

}  // namespace

extern "C" int dtk_gemm_nt_f32_indexed(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, int64_t lda,
                                       int64_t ldb, int64_t ldc, int32_t batch, int64_t stride_a, int64_t stride_b, int64_t stride_c,
                                       int32_t split_k, int32_t accumulate, const float* scale_a, const float* scale_b,
                                       const int32_t* index_b, const int32_t* index_c, void* stream) {
    DTK_REQUIRE(A && B && C, "dtk_gemm_nt_f32: null pointer");
    DTK_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0 && split_k > 0, "dtk_gemm_nt_f32: bad sizes");
    DTK_REQUIRE(lda >= K && ldb >= K && ldc >= N, "dtk_gemm_nt_f32: leading dimensions");
    DTK_REQUIRE((long long)batch * split_k <= 65535, "dtk_gemm_nt_f32: batch * split_k > 65535");
    int k_chunk = dtk_cdiv(dtk_cdiv(K, split_k), GKS) * GKS;
    const bool vec = lda % 4 == 0 && ldb % 4 == 0 && K % 4 == 0 && stride_a % 4 == 0 && stride_b % 4 == 0 &&
                     ((size_t)A & 15) == 0 && ((size_t)B & 15) == 0;
    const dim3 grid(dtk_cdiv(N, GT), dtk_cdiv(M, GT), batch * split_k);
    if (vec)
        DTK_LAUNCH("train_gemm", gemm_nt_split_kernel<true>, grid, dim3(256), 0, dtk_stream(stream), A, B, C, M, N, K, (long long)lda,
                   (long long)ldb, (long long)ldc, (long long)stride_a, (long long)stride_b, (long long)stride_c, split_k, k_chunk,
                   accumulate, scale_a, scale_b, index_b, index_c);
    else
        DTK_LAUNCH("train_gemm", gemm_nt_split_kernel<false>, grid, dim3(256), 0, dtk_stream(stream), A, B, C, M, N, K, (long long)lda,
                   (long long)ldb, (long long)ldc, (long long)stride_a, (long long)stride_b, (long long)stride_c, split_k, k_chunk,
                   accumulate, scale_a, scale_b, index_b, index_c);
    return DTK_OK;
}

extern "C" int dtk_gemm_nt_f32(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb,
                               int64_t ldc, int32_t batch, int64_t stride_a, int64_t stride_b, int64_t stride_c, int32_t split_k,
                               int32_t accumulate, const float* scale_a, const float* scale_b, void* stream) {
    return dtk_gemm_nt_f32_indexed(A, B, C, M, N, K, lda, ldb, ldc, batch, stride_a, stride_b, stride_c, split_k, accumulate, scale_a,
                                   scale_b, nullptr, nullptr, stream);
}

extern "C" int dtk_im2col(const float* x, float* cols, int32_t n, int32_t C, int32_t H, int32_t W, int32_t ksize, int32_t pad,
                          int32_t dil, int32_t reflect, int32_t layout, int32_t Kp, int64_t Lp, void* stream) {
    DTK_REQUIRE(x && cols && n > 0 && C > 0 && H > 0 && W > 0 && ksize > 0, "dtk_im2col: bad arguments");
    DTK_REQUIRE(2 * pad == dil * (ksize - 1), "dtk_im2col: only 'same' convolutions (2 pad == dil (k - 1))");
    DTK_REQUIRE(!reflect || (pad < H && pad < W), "dtk_im2col: reflect padding needs pad < H, W");
    DTK_REQUIRE(Kp >= C * ksize * ksize && (layout == 0 || Lp >= (long long)H * W), "dtk_im2col: Kp / Lp too small");
    DTK_REQUIRE(layout != 0 || Kp % 32 == 0, "dtk_im2col: layout 0 needs Kp %% 32 == 0");
    DTK_REQUIRE(layout == 0 || (Lp % 4 == 0 && ((size_t)cols & 15) == 0), "dtk_im2col: layout 1 needs Lp %% 4 == 0 and 16-byte alignment");
    const long long L = (long long)H * W;
    if (layout == 0)
        DTK_LAUNCH("train_im2col", im2col_kernel, dim3(dtk_cdiv(L, 32), 1, n), dim3(256), 0, dtk_stream(stream), x, cols, C, H, W,
                   ksize, pad, dil, reflect, 0, Kp, (long long)0);
    else
        DTK_LAUNCH("train_im2col", im2col_kernel, dim3(dtk_cdiv(Lp, 1024), dtk_cdiv(Kp, 8), n), dim3(256), 0, dtk_stream(stream), x,
                   cols, C, H, W, ksize, pad, dil, reflect, 1, Kp, (long long)Lp);
    return DTK_OK;
}

extern "C" int dtk_col2im(const float* dcols, float* dx, int32_t n, int32_t C, int32_t H, int32_t W, int32_t ksize, int32_t pad,
                          int32_t dil, int32_t reflect, int32_t Kp, void* stream) {
    DTK_REQUIRE(dcols && dx && n > 0 && C > 0 && H > 0 && W > 0 && ksize > 0, "dtk_col2im: bad arguments");
    DTK_REQUIRE(2 * pad == dil * (ksize - 1), "dtk_col2im: only 'same' convolutions (2 pad == dil (k - 1))");
    DTK_REQUIRE(!reflect || (2 * pad < H && 2 * pad < W), "dtk_col2im: reflect padding needs 2 pad < H, W");
    DTK_REQUIRE(Kp >= C * ksize * ksize && C <= 65535, "dtk_col2im: Kp too small / too many channels");
    const long long L = (long long)H * W;
    DTK_LAUNCH("train_col2im", col2im_kernel, dim3(dtk_cdiv(L, 256), C, n), dim3(256), 0, dtk_stream(stream), dcols, dx, C, H, W,
               ksize, pad, dil, reflect, Kp);
    return DTK_OK;
}

extern "C" int dtk_transpose_f32(const float* src, float* dst, int64_t rows, int64_t cols, int32_t batch, void* stream) {
    DTK_REQUIRE(src && dst && rows > 0 && cols > 0 && batch > 0, "dtk_transpose_f32: bad arguments");
    DTK_REQUIRE(dtk_cdiv(rows, 32) <= 65535 && batch <= 65535, "dtk_transpose_f32: too many row tiles / batches");
    DTK_LAUNCH("train_transpose", transpose_kernel, dim3(dtk_cdiv(cols, 32), dtk_cdiv(rows, 32), batch), dim3(256), 0,
               dtk_stream(stream), src, dst, (long long)rows, (long long)cols);
    return DTK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// CNN -> ViT grid alignment of the training step (models/utils.py:7-45: a bilinear grid_sample with border clamp and
// align_corners of the stride-8 CNN map at the ViT token centres), forward and backward.  The sampling positions depend on
// the geometry only, so they come as per-axis tables (built once on the host in the reference's float32 order of
// operations): destination index i reads source cells lo[i] and hi[i] = min(lo[i] + 1, n - 1) with weights 1 - whi[i] and
// whi[i].  Round 2 applied the two interpolation matrices as library GEMMs over 60 x 107 planes: six launches of 2.3 ms per
// iteration for what is 4 reads per output.  The backward is a GATHER: lo / hi are non-decreasing, so the destination
// indices that read source cell a form two contiguous ranges (as lo, as hi), tabulated per axis.
//   fwd  dst[p][i][j] = sum_{a in {lo_y[i], hi_y[i]}} sum_{b in {lo_x[j], hi_x[j]}} wy wx src[p][a][b]
//   bwd  dsrc[p][a][b] = sum over (i, j) that read (a, b) of wy wx ddst[p][i][j]
// ---------------------------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void resample_fwd_kernel(const float* __restrict__ src, float* __restrict__ dst, int hs, int ws,
                                                           int hd, int wd, const int32_t* __restrict__ ylo,
                                                           const float* __restrict__ ywhi, const int32_t* __restrict__ xlo,
                                                           const float* __restrict__ xwhi) {
    const long long p = blockIdx.y;
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= hd * wd) return;
    const int i = o / wd, j = o - i * wd;
    const int a0 = ylo[i], a1 = min(a0 + 1, hs - 1), b0 = xlo[j], b1 = min(b0 + 1, ws - 1);
    const float wy = ywhi[i], wx = xwhi[j];
    const float* s = src + p * hs * ws;
    // the order of the reference's two products: columns first (cnn . Mx^T), then rows (My . that)
    const float r0 = s[a0 * ws + b0] * (1.f - wx) + s[a0 * ws + b1] * wx;
    const float r1 = s[a1 * ws + b0] * (1.f - wx) + s[a1 * ws + b1] * wx;
    dst[p * hd * wd + o] = r0 * (1.f - wy) + r1 * wy;
}

// ranges: r[0..n) start and r[n..2n) end of the destination indices whose LO is a; r[2n..3n), r[3n..4n) of those whose HI is a
__global__ __launch_bounds__(256) void resample_bwd_kernel(const float* __restrict__ ddst, float* __restrict__ dsrc, int hs, int ws,
                                                           int hd, int wd, const int32_t* __restrict__ yr,
                                                           const float* __restrict__ ywhi, const int32_t* __restrict__ xr,
                                                           const float* __restrict__ xwhi) {
    const long long p = blockIdx.y;
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= hs * ws) return;
    const int a = o / ws, b = o - a * ws;
    const float* d = ddst + p * hd * wd;
    float acc = 0.f;
#pragma unroll
    for (int sy = 0; sy < 2; ++sy) {
        const int i0 = yr[(2 * sy) * hs + a], i1 = yr[(2 * sy + 1) * hs + a];
        for (int i = i0; i < i1; ++i) {
            const float wy = sy ? ywhi[i] : 1.f - ywhi[i];
            float row = 0.f;
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                const int j0 = xr[(2 * sx) * ws + b], j1 = xr[(2 * sx + 1) * ws + b];
                for (int j = j0; j < j1; ++j) row += (sx ? xwhi[j] : 1.f - xwhi[j]) * d[i * wd + j];
            }
            acc += wy * row;
        }
    }
    dsrc[p * hs * ws + o] = acc;
}

}  // namespace

extern "C" int dtk_resample2d_forward(const float* src, float* dst, int64_t planes, int32_t hs, int32_t ws, int32_t hd, int32_t wd,
                                      const int32_t* ylo, const float* ywhi, const int32_t* xlo, const float* xwhi, void* stream) {
    DTK_REQUIRE(src && dst && ylo && ywhi && xlo && xwhi, "dtk_resample2d_forward: null pointer");
    DTK_REQUIRE(planes > 0 && planes <= 65535 && hs > 0 && ws > 0 && hd > 0 && wd > 0, "dtk_resample2d_forward: bad sizes");
    DTK_LAUNCH("train_align_fwd", resample_fwd_kernel, dim3(dtk_cdiv((long long)hd * wd, 256), (unsigned)planes), dim3(256), 0,
               dtk_stream(stream), src, dst, hs, ws, hd, wd, ylo, ywhi, xlo, xwhi);
    return DTK_OK;
}

extern "C" int dtk_resample2d_backward(const float* ddst, float* dsrc, int64_t planes, int32_t hs, int32_t ws, int32_t hd, int32_t wd,
                                       const int32_t* yranges, const float* ywhi, const int32_t* xranges, const float* xwhi,
                                       void* stream) {
    DTK_REQUIRE(ddst && dsrc && yranges && ywhi && xranges && xwhi, "dtk_resample2d_backward: null pointer");
    DTK_REQUIRE(planes > 0 && planes <= 65535 && hs > 0 && ws > 0 && hd > 0 && wd > 0, "dtk_resample2d_backward: bad sizes");
    DTK_LAUNCH("train_align_bwd", resample_bwd_kernel, dim3(dtk_cdiv((long long)hs * ws, 256), (unsigned)planes), dim3(256), 0,
               dtk_stream(stream), ddst, dsrc, hs, ws, hd, wd, yranges, ywhi, xranges, xwhi);
    return DTK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of a 5 x 5 convolution WITHOUT the unfolded operand (round 3, second form):
//     dW[co][ci][ky][kx] = sum_{frames, y, x} dY[co][y][x] Xpad[ci][y + ky d][x + kx d]
// as 25 small matrix products per workgroup whose reduction index is the PIXEL: both operands are read where autograd leaves
// them (NCHW fp32: pixels contiguous), split hi + lo into fp16 while they are staged into LDS as [channel][row][x], and the
// MFMA 16x16x32 fragments are 8 consecutive pixels of one channel -- a 16-byte LDS read.  The x shift of a tap (kx d pixels)
// is applied IN REGISTERS: a lane reads the two aligned 8-pixel groups that cover its shifted window and selects / funnel-
// shifts (v_alignbit) the 16-bit elements, so that no tap needs an unaligned LDS access or its own copy of the tile.
// A workgroup (4 waves) owns a 32 x 32 block of (cout, cin) for all 25 taps -- 25 accumulator tiles of 16 x 16 per wave -- and
// walks its share of 4-row x 32-pixel tiles of all frames; the partial sums go to dW with atomic adds (S-way split over the
// pixels).  The im2col form wrote and re-read 10 GB of columns per iteration for the same products.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int WG_R = 4, WG_TW = 32, WG_XP = 40;  // tile rows, tile width, LDS row pitch of X in halves (width + 8)
constexpr int WG_PART = 4 * 25 * 4 * 64;          // floats of one workgroup's partial block: 32 cout x 32 cin x 25 taps

template <int DIL>
struct WgradCfg {
    static constexpr int XR = WG_R + 4 * DIL;               // X rows incl. halo
    static constexpr int XC = XR * WG_XP + 8;               // halves per cin plane (+16 B: conflict-free across channels)
    static constexpr int YC = WG_R * WG_TW + 8;             // halves per cout plane
    static constexpr int X_HALVES = 32 * XC, Y_HALVES = 32 * YC;
    static constexpr size_t LDS_BYTES = (size_t)(2 * X_HALVES + 2 * Y_HALVES) * sizeof(half_t);
};

typedef unsigned int u4v __attribute__((ext_vector_type(4)));

// halves [S .. S + 7] of the 16 halves (a, b)
template <int S>
__device__ __forceinline__ h8 shift_window(const u4v a, const u4v b) {
    const unsigned int u[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    u4v r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (S % 2 == 0) r[j] = u[S / 2 + j];
        else r[j] = __builtin_amdgcn_alignbit(u[(S - 1) / 2 + j + 1], u[(S - 1) / 2 + j], 16);
    }
    return __builtin_bit_cast(h8, r);
}

template <int DIL, bool SINGLE>
__global__ __launch_bounds__(256, 2) void conv_wgrad_split_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   float* __restrict__ partial, int N, int Cin, int Cout, int H,
                                                                   int W, int reflect_pad, const float* __restrict__ scale_dy,
                                                                   int tiles_x, int tiles_y, int splits) {
    typedef WgradCfg<DIL> Cfg;
    extern __shared__ __attribute__((aligned(16))) half_t wg_smem[];
    half_t* Xh = wg_smem;
    half_t* Xl = Xh + Cfg::X_HALVES;
    half_t* Yh = Xl + Cfg::X_HALVES;
    half_t* Yl = Yh + Cfg::Y_HALVES;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fj = lane & 15, fg = lane >> 4;
    const int cot = w & 1, cit = w >> 1;
    const int nci = (Cin + 31) / 32;
    // XCD-aware 1-D grid: workgroup ids go round-robin over the 8 XCDs; within one XCD consecutive workgroups take the weight
    // blocks of ONE pixel split, so the X / dY tiles they all read come from HBM once and from that XCD's L2 afterwards
    // (plain (split, block) grid: 2.9 GB fetched per launch, 8x the operands)
    const int nblocks = nci * ((Cout + 31) / 32);
    const long long lin = blockIdx.x, jj = lin >> 3;
    const int wblock = (int)(jj % nblocks);
    const int split = (int)((jj / nblocks) * 8 + (lin & 7));
    if (split >= splits) return;
    const int co0 = (wblock / nci) * 32, ci0 = (wblock % nci) * 32;
    const float sdy = scale_dy ? *scale_dy : 1.f;
    f4 acc[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
    const long long tiles = (long long)N * tiles_y * tiles_x;
    const size_t plane = (size_t)H * W;
    // Operand tiles travel through REGISTERS one tile ahead (requested before the MFMA loop of the current tile, converted and
    // written to LDS after it): X = 32 cin x XR rows x 5 groups of 8 pixels, dY = 32 cout x 4 rows x 4 groups.
    constexpr int XITEMS = 32 * Cfg::XR * (WG_XP / 8), XPER = (XITEMS + 255) / 256;
    constexpr int YITEMS = 32 * WG_R * (WG_TW / 8), YPER = YITEMS / 256;
    static_assert(YITEMS % 256 == 0, "dY items per thread");
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // rows of an NCHW plane are only 4-byte aligned
    f4 xr[XPER][2], yr[YPER][2];
    auto load_tile = [&](long long tile) {
        const int n = (int)(tile / (tiles_y * tiles_x));
        const int rem = (int)(tile - (long long)n * tiles_y * tiles_x);
        const int y0 = (rem / tiles_x) * WG_R, x0 = (rem % tiles_x) * WG_TW;
        // every X position of the tile inside the image (then so is every dY position): no reflection, no bounds
        const bool interior = y0 - 2 * DIL >= 0 && y0 + WG_R - 1 + 2 * DIL < H && x0 - 2 * DIL >= 0 && x0 - 2 * DIL + WG_XP - 1 < W;
#pragma unroll
        for (int i = 0; i < XPER; ++i) {
            const int it = tid + 256 * i;
            const int xg = it % (WG_XP / 8), rr = (it / (WG_XP / 8)) % Cfg::XR, c = it / ((WG_XP / 8) * Cfg::XR);
            const int ci = ci0 + c;
            const bool live = it < XITEMS && ci < Cin;
            const int iy = y0 - 2 * DIL + rr, ix0 = x0 - 2 * DIL + xg * 8;
            const float* pl = x + ((size_t)n * Cin + min(ci, Cin - 1)) * plane;
            if (interior) {
                const float* src = pl + (size_t)iy * W + ix0;
                xr[i][0] = live ? (f4)*reinterpret_cast<const f4u*>(src) : f4{0.f, 0.f, 0.f, 0.f};
                xr[i][1] = live ? (f4)*reinterpret_cast<const f4u*>(src + 4) : f4{0.f, 0.f, 0.f, 0.f};
            } else {
                const bool row_in = reflect_pad || (iy >= 0 && iy < H);
                int gy = iy < 0 ? -iy : (iy >= H ? 2 * (H - 1) - iy : iy);
                gy = min(max(gy, 0), H - 1);
                const float* src = pl + (size_t)gy * W;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ix = ix0 + e;
                    int gx = ix < 0 ? -ix : (ix >= W ? 2 * (W - 1) - ix : ix);
                    gx = min(max(gx, 0), W - 1);
                    const bool in = live && row_in && (reflect_pad || (ix >= 0 && ix < W));
                    xr[i][e >> 2][e & 3] = in ? src[gx] : 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < YPER; ++i) {
            const int it = tid + 256 * i;
            const int xg = it % (WG_TW / 8), rr = (it / (WG_TW / 8)) % WG_R, c = it / ((WG_TW / 8) * WG_R);
            const int co = co0 + c, iy = y0 + rr, ix0 = x0 + xg * 8;
            const float* src = dy + ((size_t)n * Cout + min(co, Cout - 1)) * plane + (size_t)min(iy, H - 1) * W;
            if (interior) {
                yr[i][0] = co < Cout ? (f4)*reinterpret_cast<const f4u*>(src + ix0) : f4{0.f, 0.f, 0.f, 0.f};
                yr[i][1] = co < Cout ? (f4)*reinterpret_cast<const f4u*>(src + ix0 + 4) : f4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ix = ix0 + e;
                    yr[i][e >> 2][e & 3] = (co < Cout && iy < H && ix < W) ? src[ix] : 0.f;  // zero outside: no contribution
                }
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < XPER; ++i) {
            const int it = tid + 256 * i;
            if (it >= XITEMS) continue;
            const int xg = it % (WG_XP / 8), rr = (it / (WG_XP / 8)) % Cfg::XR, c = it / ((WG_XP / 8) * Cfg::XR);
            h8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = xr[i][e >> 2][e & 3];
                hi[e] = (half_t)v;
                lo[e] = (half_t)(v - (float)hi[e]);
            }
            const int o = c * Cfg::XC + rr * WG_XP + xg * 8;
            *reinterpret_cast<h8*>(Xh + o) = hi;
            *reinterpret_cast<h8*>(Xl + o) = lo;
        }
#pragma unroll
        for (int i = 0; i < YPER; ++i) {
            const int it = tid + 256 * i;
            const int xg = it % (WG_TW / 8), rr = (it / (WG_TW / 8)) % WG_R, c = it / ((WG_TW / 8) * WG_R);
            h8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = yr[i][e >> 2][e & 3] * sdy;
                hi[e] = (half_t)v;
                lo[e] = (half_t)(v - (float)hi[e]);
            }
            const int o = c * Cfg::YC + rr * WG_TW + xg * 8;
            *reinterpret_cast<h8*>(Yh + o) = hi;
            *reinterpret_cast<h8*>(Yl + o) = lo;
        }
    };
    if ((long long)split < tiles) load_tile(split);
    for (long long tile = split; tile < tiles; tile += splits) {
        __syncthreads();  // the previous tile's fragment reads are done
        store_tile();
        __syncthreads();
        if (tile + splits < tiles) load_tile(tile + splits);  // in flight under this tile's MFMAs
        const half_t* ybase_h = Yh + (cot * 16 + fj) * Cfg::YC + fg * 8;
        const half_t* ybase_l = Yl + (cot * 16 + fj) * Cfg::YC + fg * 8;
        const half_t* xbase_h = Xh + (cit * 16 + fj) * Cfg::XC + fg * 8;
        const half_t* xbase_l = Xl + (cit * 16 + fj) * Cfg::XC + fg * 8;
#pragma unroll 1
        for (int r = 0; r < WG_R; ++r) {
            const h8 ah = *reinterpret_cast<const h8*>(ybase_h + r * WG_TW);
            const h8 al = *reinterpret_cast<const h8*>(ybase_l + r * WG_TW);
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const int ro = (r + ky * DIL) * WG_XP;
                const u4v h0 = *reinterpret_cast<const u4v*>(xbase_h + ro), h1 = *reinterpret_cast<const u4v*>(xbase_h + ro + 8);
                const u4v l0 = *reinterpret_cast<const u4v*>(xbase_l + ro), l1 = *reinterpret_cast<const u4v*>(xbase_l + ro + 8);
#define WG_TAP(KX)                                                                                              \
    {                                                                                                           \
        const h8 bh = shift_window<(KX) * DIL>(h0, h1);                                                         \
        f4& a_ = acc[ky * 5 + (KX)];                                                                            \
        if (!SINGLE) {                                                                                          \
            const h8 bl = shift_window<(KX) * DIL>(l0, l1);                                                     \
            a_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, a_, 0, 0, 0);                                   \
            a_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, a_, 0, 0, 0);                                   \
        }                                                                                                       \
        a_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, a_, 0, 0, 0);                                       \
    }
                WG_TAP(0) WG_TAP(1) WG_TAP(2) WG_TAP(3) WG_TAP(4)
#undef WG_TAP
            }
        }
    }
    // Partial sums of this workgroup, in FRAGMENT order (wave, tap, r, lane): every store instruction writes 64 consecutive
    // floats.  (Atomic adds straight into dW serialise: 1024 workgroups x 25 600 adds on the same 2e5 addresses took as long
    // as the products.)  conv_wgrad_reduce_kernel sums over the pixel splits and un-permutes.
    float* part = partial + ((size_t)wblock * splits + split) * WG_PART + (size_t)w * 25 * 4 * 64 + lane;
#pragma unroll
    for (int t = 0; t < 25; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(t * 4 + r) * 64] = acc[t][r];
}

// dw[co][ci][t] = (1 / scale) sum_s partial[block(co, ci)][s][fragment index of (co, ci, t)]
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int Cin,
                                                                int Cout, int splits, const float* __restrict__ scale_dy) {
    const int idx = blockIdx.x * 256 + threadIdx.x;  // position inside one workgroup's partial block
    if (idx >= WG_PART) return;
    const int lane = idx & 63, r = (idx >> 6) & 3, t = (idx >> 8) % 25, w = idx / (25 * 4 * 64);
    const int fj = lane & 15, fg = lane >> 4, cot = w & 1, cit = w >> 1;
    const int nci = (Cin + 31) / 32;
    const int co = (blockIdx.y / nci) * 32 + cot * 16 + 4 * fg + r, ci = (blockIdx.y % nci) * 32 + cit * 16 + fj;
    if (co >= Cout || ci >= Cin) return;
    const float* p = partial + (size_t)blockIdx.y * splits * WG_PART + idx;
    float a = 0.f;
    for (int sp = 0; sp < splits; ++sp) a += p[(size_t)sp * WG_PART];
    dw[((size_t)co * Cin + ci) * 25 + t] = a / (scale_dy ? *scale_dy : 1.f);
}

}  // namespace

static long long wgrad_splits(int N, int Cin, int Cout, int H, int W, int dilation) {
    const int blocks = dtk_cdiv(Cout, 32) * dtk_cdiv(Cin, 32);
    const long long tiles = (long long)N * dtk_cdiv(W, WG_TW) * dtk_cdiv(H, WG_R);
    // measured (scripts/wgrad_bench.py, 8 frames): two resident workgroups per CU are best at dilation 1 (512: 1.14 / 1.12 ms for
    // layers 2 / 3, 1024: 1.25 / 1.22); the dilated layer's kernel holds one workgroup per CU and wants shorter ones (2048: 1.44 ms,
    // 512: 1.79)
    long long splits = dtk_cdiv(dilation == 1 ? 512 : 2048, blocks);
#ifdef DTK_DEV
    if (const char* e = getenv("DTK_WGRAD_WORKGROUPS")) splits = dtk_cdiv(atoi(e) > 0 ? atoi(e) : 1024, blocks);
#endif
    if (splits > tiles) splits = tiles;
    if (splits < 1) splits = 1;
    return splits;
}

extern "C" size_t dtk_conv_wgrad_split_workspace_bytes(int N, int Cin, int Cout, int H, int W, int dilation) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)dtk_cdiv(Cout, 32) * dtk_cdiv(Cin, 32) * wgrad_splits(N, Cin, Cout, H, W, dilation) * WG_PART * sizeof(float);
}

extern "C" int dtk_conv_wgrad_split(const float* x, const float* dy, float* dw, int N, int Cin, int Cout, int H, int W,
                                    int dilation, int reflect_pad, const float* scale_dy, int fp16_only,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    DTK_REQUIRE(x && dy && dw && workspace, "dtk_conv_wgrad_split: null pointer");
    DTK_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 4 * dilation && W > 4 * dilation, "dtk_conv_wgrad_split: bad shape");
    DTK_REQUIRE(dilation == 1 || dilation == 2, "dtk_conv_wgrad_split: dilation %d (1 or 2)", dilation);
    if (workspace_bytes < dtk_conv_wgrad_split_workspace_bytes(N, Cin, Cout, H, W, dilation)) {
        dtk_set_error("dtk_conv_wgrad_split: workspace %zu B < required %zu B", workspace_bytes,
                      dtk_conv_wgrad_split_workspace_bytes(N, Cin, Cout, H, W, dilation));
        return DTK_E_WORKSPACE;
    }
    static const bool lds_ok = [] {
        bool ok = true;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_split_kernel<1, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)WgradCfg<1>::LDS_BYTES) == hipSuccess;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_split_kernel<2, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)WgradCfg<2>::LDS_BYTES) == hipSuccess;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_split_kernel<1, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)WgradCfg<1>::LDS_BYTES) == hipSuccess;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_split_kernel<2, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)WgradCfg<2>::LDS_BYTES) == hipSuccess;
        return ok;
    }();
    DTK_REQUIRE(lds_ok, "dtk_conv_wgrad_split: cannot reserve LDS");
    const int tiles_x = dtk_cdiv(W, WG_TW), tiles_y = dtk_cdiv(H, WG_R);
    const int blocks = dtk_cdiv(Cout, 32) * dtk_cdiv(Cin, 32);
    const long long splits = wgrad_splits(N, Cin, Cout, H, W, dilation);
    float* partial = static_cast<float*>(workspace);
    dim3 grid((unsigned)(((splits + 7) / 8) * 8 * blocks));
#define DTK_WGRAD_RUN(NAME, DILV, SINGLEV)                                                                                        \
    DTK_LAUNCH(NAME, (conv_wgrad_split_kernel<DILV, SINGLEV>), grid, dim3(256), WgradCfg<DILV>::LDS_BYTES, dtk_stream(stream), x, dy, \
               partial, N, Cin, Cout, H, W, reflect_pad, scale_dy, tiles_x, tiles_y, (int)splits)
    if (dilation == 1 && !fp16_only) { DTK_WGRAD_RUN("train_conv_wgrad", 1, false); }
    else if (dilation == 1) { DTK_WGRAD_RUN("train_conv_wgrad_h", 1, true); }
    else if (!fp16_only) { DTK_WGRAD_RUN("train_conv_wgrad_d2", 2, false); }
    else { DTK_WGRAD_RUN("train_conv_wgrad_d2_h", 2, true); }
#undef DTK_WGRAD_RUN
    DTK_LAUNCH("train_conv_wgrad_reduce", conv_wgrad_reduce_kernel, dim3(dtk_cdiv(WG_PART, 256), blocks), dim3(256), 0,
               dtk_stream(stream), partial, dw, Cin, Cout, (int)splits, scale_dy);
    return DTK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The two embedding regularisers of the training loss (dino_tracker.py:128-139) in one pass each way:
//     norm term  = mean over frames and cells of | |x| / |raw| - 1 |,     angle term = mean of | <x, raw> / (|x| |raw|) - 1 |
// with x, raw [F][C][n] (NCHW: a cell's channels are n floats apart, adjacent cells adjacent -- one thread per cell reads
// coalesced).  Forward leaves the per-cell sums (|x|^2, |raw|^2, <x, raw>) for the backward, which is closed form per element:
//     dx = gn sgn(|x|/|raw| - 1) x / (|x| |raw|)  +  ga sgn(cos - 1) (raw / (|x| |raw|) - <x, raw> x / (|x|^3 |raw|)).
// The traced form is ~25 element-wise / reduce kernels over 100 MB tensors.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void emb_reg_forward_kernel(const float* __restrict__ x, const float* __restrict__ raw,
                                                              float* __restrict__ cell_sums, float* __restrict__ out, int C,
                                                              long long n, long long cells) {
    __shared__ float red[2][4];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float tn = 0.f, ta = 0.f;
    if (i < cells) {
        const long long f = i / n, c0 = i - f * n;
        const float* xp = x + f * C * n + c0;
        const float* rp = raw + f * C * n + c0;
        float r = 0.f, d = 0.f, dot = 0.f;
        for (int c = 0; c < C; ++c) {
            const float a = xp[(long long)c * n], b = rp[(long long)c * n];
            r = fmaf(a, a, r);
            d = fmaf(b, b, d);
            dot = fmaf(a, b, dot);
        }
        cell_sums[i] = r;
        cell_sums[cells + i] = d;
        cell_sums[2 * cells + i] = dot;
        const float nr = sqrtf(r), nd = sqrtf(d);
        tn = fabsf(nr / nd - 1.f);
        ta = fabsf(dot / (nr * nd) - 1.f);
    }
    tn = wave_sum(tn);
    ta = wave_sum(ta);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = tn; red[1][w] = ta; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float inv = 1.f / (float)cells;
        atomicAdd(out, (red[0][0] + red[0][1] + red[0][2] + red[0][3]) * inv);
        atomicAdd(out + 1, (red[1][0] + red[1][1] + red[1][2] + red[1][3]) * inv);
    }
}

__global__ __launch_bounds__(256) void emb_reg_backward_kernel(const float* __restrict__ x, const float* __restrict__ raw,
                                                               const float* __restrict__ cell_sums, const float* __restrict__ gout,
                                                               float* __restrict__ dx, int C, long long n, long long cells) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= cells) return;
    const long long f = i / n, c0 = i - f * n;
    const float r = cell_sums[i], d = cell_sums[cells + i], dot = cell_sums[2 * cells + i];
    const float nr = sqrtf(r), nd = sqrtf(d);
    const float inv = 1.f / (float)cells;
    auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
    const float gn = gout[0] * inv * sgn(nr / nd - 1.f), ga = gout[1] * inv * sgn(dot / (nr * nd) - 1.f);
    const float kx = gn / (nr * nd) - ga * dot / (nr * nr * nr * nd);   // coefficient of x
    const float kr = ga / (nr * nd);                                    // coefficient of raw
    const float* xp = x + f * C * n + c0;
    const float* rp = raw + f * C * n + c0;
    float* dp = dx + f * C * n + c0;
    for (int c = 0; c < C; ++c) dp[(long long)c * n] = fmaf(kx, xp[(long long)c * n], kr * rp[(long long)c * n]);
}

}  // namespace

extern "C" int dtk_emb_reg_forward(const float* x, const float* raw, int F, int C, int n, float* cell_sums, float* out2,
                                   void* stream) {
    DTK_REQUIRE(x && raw && cell_sums && out2 && F > 0 && C > 0 && n > 0, "dtk_emb_reg_forward: bad arguments");
    const long long cells = (long long)F * n;
    hipStream_t st = dtk_stream(stream);
    DTK_HIP(dtk_zero_async(out2, 2 * sizeof(float), st));   // (a kernel, not a memset node: common.h)
    DTK_LAUNCH("train_emb_reg", emb_reg_forward_kernel, dim3(dtk_cdiv(cells, 256)), dim3(256), 0, st, x, raw, cell_sums, out2, C,
               (long long)n, cells);
    return DTK_OK;
}

extern "C" int dtk_emb_reg_backward(const float* x, const float* raw, const float* cell_sums, const float* grad_out2, int F, int C,
                                    int n, float* dx, void* stream) {
    DTK_REQUIRE(x && raw && cell_sums && grad_out2 && dx && F > 0 && C > 0 && n > 0, "dtk_emb_reg_backward: bad arguments");
    const long long cells = (long long)F * n;
    DTK_LAUNCH("train_emb_reg_bwd", emb_reg_backward_kernel, dim3(dtk_cdiv(cells, 256)), dim3(256), 0, dtk_stream(stream), x, raw,
               cell_sums, grad_out2, dx, C, (long long)n, cells);
    return DTK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// N1: the contrastive (InfoNCE) terms of the training loss (dino_tracker.py:141-243, :327-343) without the affinity tensors
// of torch.bmm + ATen.  One "problem" q = B anchor embeddings a[q][i] against ALL n cells of frame fidx[q] of the batch's
// frame embeddings fe[F][C][n] (the layout the model keeps them in):
//     s[i][j] = a_i . f_j / max(|a_i| |f_j|, eps),      lse[i] = log sum_j exp(s[i][j] / temp)
// (the reference's term is  -log( exp(cos(a_i, b_i) / temp) / sum_j exp(s[i][j] / temp) ) = lse[i] - cos(a_i, b_i) / temp;
//  |s / temp| <= 1 / temp, so the plain sum the reference takes is safe in fp32).  Both directions of every frame pair are
// problems of ONE call.  The products run on dtk_gemm_nt_f32 (fp32-grade split-fp16 MFMA, batch -> frame through an index):
//   forward   S[q]  [B][n] = a[q] [B][C] . feT[fidx[q]] [n][C]          feT = token-major copy (dtk_pack_features, + |f_j|)
//   backward  with G = dL/ds = g_i exp(s / temp - lse_i) / temp and G1 = G / (|a_i| |f_j|):
//             da_i  = sum_j G1[i][j] f_j - (sum_j G[i][j] s[i][j]) a_i / |a_i|^2         dA[q] = G1[q] [B][n] . fe[fidx[q]] [C][n]
//             dfe_j = sum_i G1[i][j] a_i - (sum_i G[i][j] s[i][j]) f_j / |f_j|^2         dfe[fidx[q]] [C][n] += aT[q] [C][B] . G1T[q] [n][B]
//   (the clamp at eps is treated as inactive in the gradient, as it is for every vector the model produces).
// ---------------------------------------------------------------------------------------------------------------------
namespace {

constexpr float CL_EPS = 1e-8f;

// one workgroup per row (q, i): cosines in place, lse
__global__ __launch_bounds__(256) void cl_lse_kernel(float* __restrict__ S, const float* __restrict__ na, const float* __restrict__ nf,
                                                     const int32_t* __restrict__ fidx, float* __restrict__ lse, int B, int n, int np,
                                                     float inv_temp) {
    __shared__ float red[4];
    const long long row = blockIdx.x;
    const int q = (int)(row / B);
    float* sp = S + row * np;
    const float* nfp = nf + (long long)fidx[q] * n;
    const float a = na[row];
    float acc = 0.f;
    for (int j = threadIdx.x; j < np; j += 256) {
        float v = 0.f;
        if (j < n) {
            v = sp[j] / fmaxf(a * nfp[j], CL_EPS);
            acc += __expf(v * inv_temp);
        }
        sp[j] = v;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) lse[row] = logf(red[0] + red[1] + red[2] + red[3]);
}

// 64 x 64 tile of problem q: G1 (row-major and transposed), row sums r_i = sum_j G s, column sums c_j = sum_i G s, max |G1|
__global__ __launch_bounds__(256) void cl_grad_tile_kernel(const float* __restrict__ S, const float* __restrict__ lse,
                                                           const float* __restrict__ g, const float* __restrict__ na,
                                                           const float* __restrict__ nf, const int32_t* __restrict__ fidx,
                                                           float* __restrict__ G1, float* __restrict__ G1T, float* __restrict__ rsum,
                                                           float* __restrict__ csum, unsigned* __restrict__ gmax, int B, int n, int np,
                                                           float inv_temp) {
    __shared__ float t[64][65];
    const int q = blockIdx.z, i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = j0 + tx;
    const float* nfp = nf + (long long)fidx[q] * n;
    const float fnorm = j < n ? nfp[j] : 1.f;
    float cs = 0.f, mx = 0.f;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int il = ty + 4 * k, i = i0 + il;
        float g1 = 0.f, gs = 0.f;
        if (i < B && j < n) {
            const long long row = (long long)q * B + i;
            const float s = S[row * np + j];
            const float G = g[row] * __expf(s * inv_temp - lse[row]) * inv_temp;
            g1 = G / fmaxf(na[row] * fnorm, CL_EPS);
            gs = G * s;
        }
        if (i < B && j < np) G1[((long long)q * B + i) * np + j] = g1;   // (columns n .. np-1: zero padding of the reduction)
        t[il][tx] = g1;
        cs += gs;
        mx = fmaxf(mx, fabsf(g1));
        const float rs = wave_sum(gs);   // a wave = the 64 columns of one row
        if (tx == 0 && i < B) atomicAdd(&rsum[(long long)q * B + i], rs);
    }
    if (j < n) atomicAdd(&csum[(long long)q * n + j], cs);
    mx = wave_max(mx);
    if (tx == 0) atomicMax(gmax, __float_as_uint(mx));   // (non-negative floats order like their bit patterns)
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int jl = ty + 4 * k, jj = j0 + jl, i = i0 + tx;
        if (jj < n && i < B) G1T[((long long)q * n + jj) * B + i] = t[tx][jl];
    }
}

// power-of-two operand scale that brings max |x| to ~2^13 (fp16 hi halves well inside the normal range)
__global__ void cl_scale_kernel(const unsigned* __restrict__ gmax, float* __restrict__ scale) {
    const float m = __uint_as_float(*gmax);
    *scale = m > 0.f ? exp2f(fminf(fmaxf(13.f - ceilf(log2f(m)), -60.f), 60.f)) : 1.f;
}

// da[row][c] -= (r_row / |a_row|^2) a[row][c]
__global__ __launch_bounds__(256) void cl_da_fix_kernel(float* __restrict__ da, const float* __restrict__ a, const float* __restrict__ rsum,
                                                        const float* __restrict__ na, long long rows, int C) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C) return;
    const long long row = idx / C;
    const float nn = na[row] * na[row];
    if (nn > 1e-30f) da[idx] -= rsum[row] / nn * a[idx];
}

// dfe[f][c][j] -= (sum over the problems q on frame f of c_q[j]) / |f_j|^2 * fe[f][c][j];  one thread per (f, j)
__global__ __launch_bounds__(256) void cl_dfe_fix_kernel(float* __restrict__ dfe, const float* __restrict__ fe, const float* __restrict__ csum,
                                                         const float* __restrict__ nf, const int32_t* __restrict__ fidx, int Q, int C, int n,
                                                         int F) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)F * n) return;
    const int f = (int)(idx / n), j = (int)(idx - (long long)f * n);
    float cs = 0.f;
    for (int q = 0; q < Q; ++q)
        if (fidx[q] == f) cs += csum[(long long)q * n + j];
    const float nn = nf[idx] * nf[idx];
    if (cs == 0.f || !(nn > 1e-30f)) return;
    const float k = cs / nn;
    const float* fp = fe + (long long)f * C * n + j;
    float* dp = dfe + (long long)f * C * n + j;
    for (int c = 0; c < C; ++c) dp[(long long)c * n] -= k * fp[(long long)c * n];
}

inline size_t cl_al(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

extern "C" size_t dtk_contrastive_workspace_bytes(int Q, int B, int C, int n) {
    if (Q <= 0 || B <= 0 || C <= 0 || n <= 0) return 0;
    const size_t np = ((size_t)n + 3) / 4 * 4;
    return cl_al((size_t)Q * B * np * 4) + cl_al((size_t)Q * n * B * 4) + cl_al((size_t)Q * C * B * 4) + cl_al((size_t)Q * B * 4) +
           cl_al((size_t)Q * n * 4) + 512;
}

extern "C" int dtk_contrastive_forward(const float* fe, const float* a, const int32_t* fidx, float temp, int Q, int B, int C, int n,
                                       int F, float* fet, float* nf, float* na, float* S, float* lse, void* stream) {
    DTK_REQUIRE(fe && a && fidx && fet && nf && na && S && lse, "dtk_contrastive_forward: null pointer");
    DTK_REQUIRE(Q > 0 && B > 0 && C > 0 && n > 0 && F > 0 && temp > 0.f, "dtk_contrastive_forward: bad sizes");
    const int np = (n + 3) / 4 * 4;
    int rc = dtk_pack_features(fe, fet, nf, F, C, n, stream);                 // token-major copy of the frames + |f_j|
    if (rc != DTK_OK) return rc;
    rc = dtk_feature_norms(a, na, Q, C, B, stream);                            // |a_i|
    if (rc != DTK_OK) return rc;
    rc = dtk_gemm_nt_f32_indexed(a, fet, S, B, n, C, C, C, np, Q, (int64_t)B * C, (int64_t)n * C, (int64_t)B * np, 1, 0, nullptr,
                                 nullptr, fidx, nullptr, stream);
    if (rc != DTK_OK) return rc;
    DTK_LAUNCH("train_cl_lse", cl_lse_kernel, dim3((unsigned)(Q * B)), dim3(256), 0, dtk_stream(stream), S, na, nf, fidx, lse, B, n, np,
               1.f / temp);
    return DTK_OK;
}

extern "C" int dtk_contrastive_backward(const float* fe, const float* a, const int32_t* fidx, float temp, int Q, int B, int C, int n,
                                        int F, const float* nf, const float* na, const float* S, const float* lse, const float* g,
                                        float* da, float* dfe, void* workspace, size_t workspace_bytes, void* stream) {
    DTK_REQUIRE(fe && a && fidx && nf && na && S && lse && g && da && dfe && workspace, "dtk_contrastive_backward: null pointer");
    DTK_REQUIRE(Q > 0 && B > 0 && C > 0 && n > 0 && F > 0 && temp > 0.f, "dtk_contrastive_backward: bad sizes");
    DTK_REQUIRE(workspace_bytes >= dtk_contrastive_workspace_bytes(Q, B, C, n), "dtk_contrastive_backward: workspace too small");
    const int np = (n + 3) / 4 * 4;
    hipStream_t st = dtk_stream(stream);
    unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
    float* G1 = reinterpret_cast<float*>(w);   w += cl_al((size_t)Q * B * np * 4);
    float* G1T = reinterpret_cast<float*>(w);  w += cl_al((size_t)Q * n * B * 4);
    float* aT = reinterpret_cast<float*>(w);   w += cl_al((size_t)Q * C * B * 4);
    float* rsum = reinterpret_cast<float*>(w); w += cl_al((size_t)Q * B * 4);
    float* csum = reinterpret_cast<float*>(w); w += cl_al((size_t)Q * n * 4);
    unsigned* gmax = reinterpret_cast<unsigned*>(w);
    float* scale = reinterpret_cast<float*>(w + 256);
    DTK_HIP(dtk_zero_async(rsum, cl_al((size_t)Q * B * 4) + cl_al((size_t)Q * n * 4) + 512, st));   // rsum, csum, gmax, scale
    DTK_HIP(dtk_zero_async(dfe, (size_t)F * C * n * 4, st));
    DTK_LAUNCH("train_cl_grad", cl_grad_tile_kernel, dim3(dtk_cdiv(np, 64), dtk_cdiv(B, 64), Q), dim3(256), 0, st, S, lse, g, na, nf, fidx,
               G1, G1T, rsum, csum, gmax, B, n, np, 1.f / temp);
    DTK_LAUNCH("train_cl_scale", cl_scale_kernel, dim3(1), dim3(1), 0, st, gmax, scale);
    // dA[q] [B][C] = G1[q] [B][n] . fe[fidx[q]] [C][n]
    int rc = dtk_gemm_nt_f32_indexed(G1, fe, da, B, C, n, np, n, C, Q, (int64_t)B * np, (int64_t)C * n, (int64_t)B * C, 1, 0, scale, nullptr,
                                     fidx, nullptr, stream);
    if (rc != DTK_OK) return rc;
    DTK_LAUNCH("train_cl_da", cl_da_fix_kernel, dim3(dtk_cdiv((long long)Q * B * C, 256)), dim3(256), 0, st, da, a, rsum, na,
               (long long)Q * B, C);
    // dfe[fidx[q]] [C][n] += aT[q] [C][B] . G1T[q] [n][B]   (atomic: several problems may share a frame)
    rc = dtk_transpose_f32(a, aT, B, C, Q, stream);
    if (rc != DTK_OK) return rc;
    rc = dtk_gemm_nt_f32_indexed(aT, G1T, dfe, C, n, B, B, B, n, Q, (int64_t)C * B, (int64_t)n * B, (int64_t)C * n, 1, 2, nullptr, scale,
                                 nullptr, fidx, stream);
    if (rc != DTK_OK) return rc;
    DTK_LAUNCH("train_cl_dfe", cl_dfe_fix_kernel, dim3(dtk_cdiv((long long)F * n, 256)), dim3(256), 0, st, dfe, fe, csum, nf, fidx, Q, C, n, F);
    return DTK_OK;
}


// ---- bilinear reads of the batch's frame embeddings, both ways (include/dtk.h: dtk_sample_bilinear_forward / _backward) -------------
// Tracker.sample_embeddings of the training step (tracker.py:96-111 -> utils.py:65-109 with t an exact frame index): out[b] = the four
// cells around (x, y) of frame t, weighted -- align_corners, border clamp -- read from the token-major copy [n][h w][C] the tracker
// passes keep anyway.  The traced form (train_ops._bilinear_corners / _bilinear_read) is ~45 small library launches forward and ~20
// backward, four times per iteration; here one launch each way.  Same arithmetic, same order: fx = clamp((x + 1) / 2 (w - 1), 0, w - 1),
// x0 = floor(fx), wx = fx - x0, x1 = min(x0 + 1, w - 1); corners in the order (y0, x0), (y0, x1), (y1, x0), (y1, x1).
namespace {
struct BilinearTaps { long long cell[4]; float wgt[4]; };
__device__ __forceinline__ BilinearTaps bilinear_taps(const float* __restrict__ p, int n, int h, int w) {
    const float fx = fminf(fmaxf((p[0] + 1.f) * 0.5f * (float)(w - 1), 0.f), (float)(w - 1));
    const float fy = fminf(fmaxf((p[1] + 1.f) * 0.5f * (float)(h - 1), 0.f), (float)(h - 1));
    const int t = min(max((int)rintf(p[2]), 0), n - 1);
    const float x0f = fminf(floorf(fx), (float)(w - 1)), y0f = fminf(floorf(fy), (float)(h - 1));
    const float wx = fx - x0f, wy = fy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
    BilinearTaps r;
    const long long base = (long long)t * h * w;
    r.cell[0] = base + (long long)y0 * w + x0; r.wgt[0] = (1.f - wx) * (1.f - wy);
    r.cell[1] = base + (long long)y0 * w + x1; r.wgt[1] = wx * (1.f - wy);
    r.cell[2] = base + (long long)y1 * w + x0; r.wgt[2] = (1.f - wx) * wy;
    r.cell[3] = base + (long long)y1 * w + x1; r.wgt[3] = wx * wy;
    return r;
}
__global__ __launch_bounds__(128) void sample_bilinear_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ pts,
                                                                  float* __restrict__ out, int n, int h, int w, int C) {
    const BilinearTaps tp = bilinear_taps(pts + (long long)blockIdx.x * 3, n, h, w);
    for (int c = threadIdx.x; c < C; c += 128) {
        float acc = feat[tp.cell[0] * C + c] * tp.wgt[0];
        acc = acc + feat[tp.cell[1] * C + c] * tp.wgt[1];
        acc = acc + feat[tp.cell[2] * C + c] * tp.wgt[2];
        acc = acc + feat[tp.cell[3] * C + c] * tp.wgt[3];
        out[(long long)blockIdx.x * C + c] = acc;
    }
}
__global__ __launch_bounds__(128) void sample_bilinear_bwd_kernel(const float* __restrict__ g, const float* __restrict__ pts,
                                                                  float* __restrict__ dfeat, int n, int h, int w, int C) {
    const BilinearTaps tp = bilinear_taps(pts + (long long)blockIdx.x * 3, n, h, w);
    for (int c = threadIdx.x; c < C; c += 128) {
        const float gv = g[(long long)blockIdx.x * C + c];
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(&dfeat[tp.cell[k] * C + c], gv * tp.wgt[k]);
    }
}
}  // namespace

extern "C" int dtk_sample_bilinear_forward(const float* feat, const float* pts, float* out, int32_t B, int32_t n, int32_t h, int32_t w,
                                           int32_t C, void* stream) {
    DTK_REQUIRE(feat && pts && out, "dtk_sample_bilinear_forward: null pointer");
    DTK_REQUIRE(B >= 0 && n > 0 && h > 0 && w > 0 && C > 0, "dtk_sample_bilinear_forward: bad sizes");
    if (B == 0) return DTK_OK;
    DTK_LAUNCH("train_sample_fwd", sample_bilinear_fwd_kernel, dim3((unsigned)B), dim3(128), 0, dtk_stream(stream), feat, pts, out, n, h, w, C);
    return DTK_OK;
}

extern "C" int dtk_sample_bilinear_backward(const float* g, const float* pts, float* dfeat, int32_t B, int32_t n, int32_t h, int32_t w,
                                            int32_t C, void* stream) {
    DTK_REQUIRE(g && pts && dfeat, "dtk_sample_bilinear_backward: null pointer");
    DTK_REQUIRE(B >= 0 && n > 0 && h > 0 && w > 0 && C > 0, "dtk_sample_bilinear_backward: bad sizes");
    if (B == 0) return DTK_OK;
    DTK_LAUNCH("train_sample_bwd", sample_bilinear_bwd_kernel, dim3((unsigned)B), dim3(128), 0, dtk_stream(stream), g, pts, dfeat, n, h, w, C);
    return DTK_OK;
}


// ---- operand scale of a gradient tensor (include/dtk.h: dtk_pow2_scale) ---------------------------------------------------------------
// out[0] = 2^e with max |x| * 2^e in [2^9, 2^10] (max |x| clamped below at 1e-30), what train_ops._pow2_scale computed with a library
// reduction: torch.exp2(torch.floor(10 - torch.log2(vector_norm(x, inf).clamp_min(1e-30)))).  Why a kernel of its own (round 6): the
// library's multi-block reduction zeroes its semaphores with a memset, and a memset captured into a graph does not hold its place in
// the stream on this stack (common.h: dtk_zero_async) -- inside the captured training iteration the norm came back as 0 and every data
// gradient behind it as NaN.  scratch: one 32-bit word.  NaN inputs propagate (a NaN maximum gives a NaN scale, as before).
namespace {
__global__ __launch_bounds__(256) void absmax_bits_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ bits) {
    float m = 0.f;
    bool bad = false;
    const long long n4 = (((size_t)x & 15) == 0) ? n / 4 : 0;          // 16-byte pieces (torch allocations are aligned), then the tail
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    auto take = [&](const float4 v) {
        const float a = fabsf(v.x), b = fabsf(v.y), c = fabsf(v.z), d = fabsf(v.w);
        bad |= (a != a) | (b != b) | (c != c) | (d != d);
        m = fmaxf(fmaxf(m, fmaxf(a, b)), fmaxf(c, d));
    };
    for (; i + 3 * stride < n4; i += 4 * stride) {      // four 16-byte loads in flight per lane
        const float4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
        take(v0); take(v1); take(v2); take(v3);
    }
    for (; i < n4; i += stride) take(x4[i]);
    for (i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float v = fabsf(x[i]);
        bad |= v != v;
        m = fmaxf(m, v);
    }
    unsigned u = bad ? 0x7fc00000u : __float_as_uint(m);      // (non-negative floats and the quiet NaN above them order as integers)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, o, 64));
    __shared__ unsigned part[4];     // one atomic per workgroup: thousands of waves on one address serialise in the L2
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = u;
    __syncthreads();
    if (threadIdx.x == 0) {
        u = max(max(part[0], part[1]), max(part[2], part[3]));
        if (u != 0u) atomicMax(bits, u);
    }
}
__global__ void pow2_scale_kernel(const unsigned* __restrict__ bits, float* __restrict__ out) {
    const float amax = fmaxf(__uint_as_float(*bits), 1e-30f);   // (fmaxf drops a NaN: restored below)
    out[0] = (*bits > 0x7f800000u) ? __uint_as_float(0x7fc00000u) : exp2f(floorf(10.f - log2f(amax)));
}
}  // namespace

extern "C" int dtk_pow2_scale(const float* x, int64_t n, float* out, void* scratch, void* stream) {
    DTK_REQUIRE(x && out && scratch && n > 0, "dtk_pow2_scale: null pointer or empty tensor");
    hipStream_t st = dtk_stream(stream);
    unsigned* bits = reinterpret_cast<unsigned*>(scratch);
    DTK_HIP(dtk_zero_async(bits, 4, st));
    const long long blocks = (n + 256 * 16 - 1) / (256 * 16);
    DTK_LAUNCH("train_absmax", absmax_bits_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, x, (long long)n, bits);
    DTK_LAUNCH("train_pow2_scale", pow2_scale_kernel, dim3(1), dim3(1), 0, st, bits, out);
    return DTK_OK;
}


// ---- fused Adam (round 5; include/dtk.h: dtk_adam_step) ------------------------------------------------------------------------
// One launch over every parameter tensor of the training step.  Block b works on chunk b of the concatenation of the tensors in
// chunks of ADAM_CHUNK elements (a tensor's last chunk is short); it finds its tensor by walking the <= 32 sizes in the argument block.
namespace {
constexpr int ADAM_CHUNK = 4096;   // elements per block (256 threads x 4 float4 ... tails handled per element)
struct AdamScalars { float step_size[DTK_ADAM_MAX_TENSORS]; float inv_bc2_sqrt[DTK_ADAM_MAX_TENSORS]; float omb1, beta2, omb2, eps; };

// DEV: the two per-tensor scalars come from device memory (dev[t] = step size, dev[DTK_ADAM_MAX_TENSORS + t] = 1 / sqrt(bias
// correction 2)) instead of the argument block -- the form a captured graph replays: pointers and sizes are baked into the launch,
// the learning rate and the step count change every iteration.
template <bool DEV>
__global__ __launch_bounds__(256) void adam_multi_kernel(dtk_adam_args a, AdamScalars sc, const float* __restrict__ dev) {
    long long chunk = blockIdx.x;
    int t = 0;
    for (; t < a.n_tensors; ++t) {
        const long long nc = (a.numel[t] + ADAM_CHUNK - 1) / ADAM_CHUNK;
        if (chunk < nc) break;
        chunk -= nc;
    }
    if (t >= a.n_tensors) return;
    float* __restrict__ p = a.param[t];
    const float* __restrict__ g = a.grad[t];
    float* __restrict__ m = a.exp_avg[t];
    float* __restrict__ v = a.exp_avg_sq[t];
    const long long n = a.numel[t], base = chunk * ADAM_CHUNK;
    const float step_size = DEV ? dev[t] : sc.step_size[t], inv_bc2_sqrt = DEV ? dev[DTK_ADAM_MAX_TENSORS + t] : sc.inv_bc2_sqrt[t];
    for (long long i = base + threadIdx.x; i < n && i < base + ADAM_CHUNK; i += 256) {
        const float gi = g[i];
        const float mi = m[i] + sc.omb1 * (gi - m[i]);                   // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = sc.beta2 * v[i] + sc.omb2 * gi * gi;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_bc2_sqrt + sc.eps;        // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
        p[i] = p[i] - step_size * (mi / denom);                          // param.addcdiv_(exp_avg, denom, value = -step_size)
    }
}
}  // namespace

static int adam_check(const dtk_adam_args* a, long long* chunks_out, bool need_tensors = true) {
    DTK_REQUIRE(a && a->n_tensors > 0 && a->n_tensors <= DTK_ADAM_MAX_TENSORS, "dtk_adam_step: bad arguments");
    DTK_REQUIRE(a->beta1 >= 0. && a->beta1 < 1. && a->beta2 >= 0. && a->beta2 < 1. && a->eps >= 0., "dtk_adam_step: bad betas / eps");
    long long chunks = 0;
    for (int t = 0; t < a->n_tensors; ++t) {
        DTK_REQUIRE((!need_tensors || (a->param[t] && a->grad[t] && a->exp_avg[t] && a->exp_avg_sq[t] && a->numel[t] > 0)) &&
                        a->group[t] >= 0 && a->group[t] < DTK_ADAM_MAX_GROUPS && a->step[t] >= 1,
                    "dtk_adam_step: tensor %d: null pointer, empty tensor, bad group or step < 1", t);
        chunks += (a->numel[t] + ADAM_CHUNK - 1) / ADAM_CHUNK;
    }
    *chunks_out = chunks;
    return DTK_OK;
}

static void adam_scalars(const dtk_adam_args* a, AdamScalars* sc) {
    for (int t = 0; t < DTK_ADAM_MAX_TENSORS; ++t) {
        const int st_ = t < a->n_tensors ? a->step[t] : 1;
        const double bc1 = 1.0 - pow(a->beta1, (double)st_), bc2 = 1.0 - pow(a->beta2, (double)st_);
        sc->step_size[t] = t < a->n_tensors ? (float)(a->lr[a->group[t]] / bc1) : 0.f;
        sc->inv_bc2_sqrt[t] = (float)(1.0 / sqrt(bc2));
    }
    // torch forms its scalar coefficients in double and rounds them to fp32 once (1 - 0.999 = 0.001 there; 1.f - 0.999f = 0.00100004673)
    sc->omb1 = (float)(1.0 - a->beta1); sc->beta2 = (float)a->beta2; sc->omb2 = (float)(1.0 - a->beta2); sc->eps = (float)a->eps;
}

extern "C" int dtk_adam_step(const dtk_adam_args* a, void* stream) {
    long long chunks = 0;
    if (int rc = adam_check(a, &chunks)) return rc;
    AdamScalars sc;
    adam_scalars(a, &sc);
    hipStream_t st = dtk_stream(stream);
    DTK_LAUNCH("adam_step", adam_multi_kernel<false>, dim3((unsigned)chunks), dim3(256), 0, st, *a, sc, (const float*)nullptr);
    return DTK_OK;
}

// The per-tensor scalars of dtk_adam_step (step[] and lr[] of `a`), as that call forms them: out_host[t] = lr[group[t]] / (1 -
// beta1^step[t]), out_host[DTK_ADAM_MAX_TENSORS + t] = 1 / sqrt(1 - beta2^step[t]).  Host arithmetic only.
extern "C" int dtk_adam_scalars(const dtk_adam_args* a, float* out_host) {
    long long chunks = 0;
    if (int rc = adam_check(a, &chunks, false)) return rc;   // (only n_tensors, betas, eps, group[], step[] and lr[] are read)
    DTK_REQUIRE(out_host != nullptr, "dtk_adam_scalars: null output");
    AdamScalars sc;
    adam_scalars(a, &sc);
    for (int t = 0; t < DTK_ADAM_MAX_TENSORS; ++t) {
        out_host[t] = sc.step_size[t];
        out_host[DTK_ADAM_MAX_TENSORS + t] = sc.inv_bc2_sqrt[t];
    }
    return DTK_OK;
}

// dtk_adam_step with the per-tensor scalars read from DEVICE memory (2 * DTK_ADAM_MAX_TENSORS floats in dtk_adam_scalars' layout;
// step[] / lr[] of `a` are only validated): the launch a captured iteration replays while the host refreshes the scalars.
extern "C" int dtk_adam_step_dev(const dtk_adam_args* a, const float* scalars_dev, void* stream) {
    long long chunks = 0;
    if (int rc = adam_check(a, &chunks)) return rc;
    DTK_REQUIRE(scalars_dev != nullptr, "dtk_adam_step_dev: null scalars");
    AdamScalars sc;
    adam_scalars(a, &sc);
    hipStream_t st = dtk_stream(stream);
    DTK_LAUNCH("adam_step", adam_multi_kernel<true>, dim3((unsigned)chunks), dim3(256), 0, st, *a, sc, scalars_dev);
    return DTK_OK;
}
