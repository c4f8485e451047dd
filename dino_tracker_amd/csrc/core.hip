// core.hip -- error plumbing, layout conversion, norms, bilinear point sampling, trajectory cos-sims.
#include <stdarg.h>
#include <stdio.h>
#include "common.h"

static thread_local char g_err[512] = "";

void dtk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int dtk_version(void) { return 1; }

// ------------------------------------------------------------------------------------------------------------
// per-kernel timing with hipEvents on the launch stream (bench.py's `roofline` figures come from here).
// Off by default: no events are recorded and launches are untouched.
// ------------------------------------------------------------------------------------------------------------
#include <string.h>
#include <mutex>
#include <string>
#include <vector>
namespace {
struct ProfRec { const char* name; hipEvent_t a, b; };
struct ProfAgg { std::string name; double ms; long long launches; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof_open;
std::vector<ProfAgg> g_prof_agg;
}  // namespace

void dtk_prof_begin(const char* name, hipStream_t st) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{name, nullptr, nullptr};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, st);
    g_prof_open.push_back(r);
}
void dtk_prof_end(const char* name, hipStream_t st) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_open.empty() && g_prof_open.back().name == name) (void)hipEventRecord(g_prof_open.back().b, st);
}

extern "C" int dtk_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof_open) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof_open.clear();
    g_prof_agg.clear();
    g_prof_on = on != 0;
    return DTK_OK;
}
// waits for all recorded kernels, folds them into per-name totals; returns the number of distinct kernels
extern "C" int dtk_profile_collect(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof_open) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            ProfAgg* slot = nullptr;
            for (auto& a : g_prof_agg) if (a.name == r.name) slot = &a;
            if (!slot) { g_prof_agg.push_back(ProfAgg{r.name, 0.0, 0}); slot = &g_prof_agg.back(); }
            slot->ms += ms;
            slot->launches += 1;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof_open.clear();
    return (int)g_prof_agg.size();
}
extern "C" const char* dtk_profile_name(int i) { return (i >= 0 && i < (int)g_prof_agg.size()) ? g_prof_agg[i].name.c_str() : ""; }
extern "C" double dtk_profile_ms(int i) { return (i >= 0 && i < (int)g_prof_agg.size()) ? g_prof_agg[i].ms : 0.0; }
extern "C" long long dtk_profile_launches(int i) { return (i >= 0 && i < (int)g_prof_agg.size()) ? g_prof_agg[i].launches : 0; }
extern "C" const char* dtk_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------------------
// [T][C][HW] <-> [T][HW][C] through a 32x33 LDS tile: both global sides coalesced.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        int R, int Cn) {
    // src: [batch][R][Cn] -> dst: [batch][Cn][R]
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.z * R * Cn;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        int r = r0 + ty + j, c = c0 + tx;
        if (r < R && c < Cn) tile[ty + j][tx] = src[base + (size_t)r * Cn + c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        int c = c0 + ty + j, r = r0 + tx;
        if (r < R && c < Cn) dst[base + (size_t)c * R + r] = tile[tx][ty + j];
    }
}

// one wave per cell: |F[t][cell][:]|_2
__global__ __launch_bounds__(256) void norms_kernel(const float* __restrict__ thwc, float* __restrict__ norms,
                                                    long long cells, int C) {
    const long long cell = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= cells) return;
    const int lane = threadIdx.x & 63;
    const float* p = thwc + cell * C;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        float4 v = *reinterpret_cast<const float4*>(p + c);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = wave_sum(s);
    if (lane == 0) norms[cell] = sqrtf(s);
}

static int launch_transpose(const float* src, float* dst, int batch, int R, int Cn, hipStream_t st) {
    dim3 grid(dtk_cdiv(Cn, 32), dtk_cdiv(R, 32), batch);
    DTK_LAUNCH("transpose", transpose_kernel, grid, dim3(256), 0, st, src, dst, R, Cn);
    return DTK_OK;
}

extern "C" int dtk_feature_norms(const float* thwc, float* norms, int T, int C, int HW, void* stream) {
    DTK_REQUIRE(thwc && norms && T > 0 && HW > 0 && C > 0 && C % 4 == 0, "dtk_feature_norms: bad args (C %% 4)");
    long long cells = (long long)T * HW;
    DTK_LAUNCH("norms", norms_kernel, dim3(dtk_cdiv(cells, 4)), dim3(256), 0, dtk_stream(stream), thwc, norms, cells, C);
    return DTK_OK;
}

extern "C" int dtk_pack_features(const float* chw, float* thwc, float* norms, int T, int C, int HW, void* stream) {
    DTK_REQUIRE(chw && thwc && T > 0 && HW > 0 && C > 0 && C % 4 == 0, "dtk_pack_features: bad args (C %% 4)");
    int rc = launch_transpose(chw, thwc, T, C, HW, dtk_stream(stream));
    if (rc) return rc;
    if (norms) return dtk_feature_norms(thwc, norms, T, C, HW, stream);
    return DTK_OK;
}

extern "C" int dtk_unpack_features(const float* thwc, float* chw, int T, int C, int HW, void* stream) {
    DTK_REQUIRE(chw && thwc && T > 0 && HW > 0 && C > 0, "dtk_unpack_features: bad args");
    return launch_transpose(thwc, chw, T, HW, C, dtk_stream(stream));
}

// ------------------------------------------------------------------------------------------------------------
// K8: one wave per point, lanes over channels (float4).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_kernel(dtk_geom g, const float* __restrict__ feat,
                                                     const float* __restrict__ xy, const int32_t* __restrict__ t_idx,
                                                     const int32_t* __restrict__ out_row, float* __restrict__ out,
                                                     int B) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const float half = 0.5f * (float)g.patch;
    float u = (xy[2 * b] - half) / (float)g.stride;
    float v = (xy[2 * b + 1] - half) / (float)g.stride;
    u = fminf(fmaxf(u, 0.f), (float)(g.pw - 1));
    v = fminf(fmaxf(v, 0.f), (float)(g.ph - 1));
    const float u0f = floorf(u), v0f = floorf(v);
    const float fu = u - u0f, fv = v - v0f;
    const int u0 = (int)u0f, v0 = (int)v0f;
    const int u1 = min(u0 + 1, g.pw - 1), v1 = min(v0 + 1, g.ph - 1);
    int t = t_idx[b];
    t = min(max(t, 0), g.T - 1);
    const size_t fb = (size_t)t * g.ph * g.pw;
    const float* p00 = feat + (fb + (size_t)v0 * g.pw + u0) * g.C;
    const float* p01 = feat + (fb + (size_t)v0 * g.pw + u1) * g.C;
    const float* p10 = feat + (fb + (size_t)v1 * g.pw + u0) * g.C;
    const float* p11 = feat + (fb + (size_t)v1 * g.pw + u1) * g.C;
    float* o = out + (size_t)(out_row ? out_row[b] : b) * g.C;
    for (int c = lane * 4; c < g.C; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(p00 + c), bq = *reinterpret_cast<const float4*>(p01 + c);
        const float4 cq = *reinterpret_cast<const float4*>(p10 + c), d = *reinterpret_cast<const float4*>(p11 + c);
        float4 r;
        r.x = (a.x * (1.f - fu) + bq.x * fu) * (1.f - fv) + (cq.x * (1.f - fu) + d.x * fu) * fv;
        r.y = (a.y * (1.f - fu) + bq.y * fu) * (1.f - fv) + (cq.y * (1.f - fu) + d.y * fu) * fv;
        r.z = (a.z * (1.f - fu) + bq.z * fu) * (1.f - fv) + (cq.z * (1.f - fu) + d.z * fu) * fv;
        r.w = (a.w * (1.f - fu) + bq.w * fu) * (1.f - fv) + (cq.w * (1.f - fu) + d.w * fu) * fv;
        *reinterpret_cast<float4*>(o + c) = r;
    }
}

static int check_geom(const dtk_geom* g, const char* who) {
    DTK_REQUIRE(g != nullptr, "%s: null geometry", who);
    DTK_REQUIRE(g->T > 0 && g->C > 0 && g->ph > 0 && g->pw > 0 && g->patch > 0 && g->stride > 0, "%s: bad geometry", who);
    DTK_REQUIRE(g->C % 4 == 0, "%s: C must be a multiple of 4", who);
    DTK_REQUIRE(g->ph == 1 + (g->video_h - g->patch) / g->stride && g->pw == 1 + (g->video_w - g->patch) / g->stride,
                "%s: token grid %dx%d inconsistent with video %dx%d patch %d stride %d", who, g->ph, g->pw, g->video_h,
                g->video_w, g->patch, g->stride);
    return DTK_OK;
}

extern "C" int dtk_sample_points(const dtk_geom* g, const float* feat, const float* xy, const int32_t* t_idx,
                                 const int32_t* out_row, float* out, int B, void* stream) {
    int rc = check_geom(g, "dtk_sample_points");
    if (rc) return rc;
    DTK_REQUIRE(feat && xy && t_idx && out && B >= 0, "dtk_sample_points: null pointer");
    if (B == 0) return DTK_OK;
    DTK_LAUNCH("sample", sample_kernel, dim3(dtk_cdiv(B, 4)), dim3(256), 0, dtk_stream(stream), *g, feat, xy, t_idx,
                       out_row, out, B);
    return DTK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Literal utils.bilinear_interpolate_video (utils.py:75-101): F.grid_sample of the 1 x C x T x h x w volume at
// (x, y, t) in [-1, 1], trilinear, align_corners=True, padding_mode='border'.  One wave per point.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_grid_kernel(const float* __restrict__ feat, const float* __restrict__ pts,
                                                          float* __restrict__ out, int T, int C, int ph, int pw, int B) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    // grid_sample: unnormalise ((x + 1) / 2 * (size - 1)), then clip to the border
    auto axis = [](float xn, int size, int& i0, int& i1, float& f) {
        float u = (xn + 1.f) * 0.5f * (float)(size - 1);
        u = fminf(fmaxf(u, 0.f), (float)(size - 1));
        const float u0 = floorf(u);
        f = u - u0;
        i0 = (int)u0;
        i1 = min(i0 + 1, size - 1);
    };
    int u0, u1, v0, v1, t0, t1;
    float fu, fv, ft;
    axis(pts[3 * b], pw, u0, u1, fu);
    axis(pts[3 * b + 1], ph, v0, v1, fv);
    axis(pts[3 * b + 2], T, t0, t1, ft);
    const float w00 = (1.f - fu) * (1.f - fv), w01 = fu * (1.f - fv), w10 = (1.f - fu) * fv, w11 = fu * fv;
    float* o = out + (size_t)b * C;
    for (int c = lane * 4; c < C; c += 256) {
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float wt = k ? ft : 1.f - ft;
            if (wt == 0.f) continue;  // integral t (the tracker's case): one frame is read
            const size_t fb = (size_t)(k ? t1 : t0) * ph * pw;
            const float4 a = *reinterpret_cast<const float4*>(feat + (fb + (size_t)v0 * pw + u0) * C + c);
            const float4 bq = *reinterpret_cast<const float4*>(feat + (fb + (size_t)v0 * pw + u1) * C + c);
            const float4 cq = *reinterpret_cast<const float4*>(feat + (fb + (size_t)v1 * pw + u0) * C + c);
            const float4 d = *reinterpret_cast<const float4*>(feat + (fb + (size_t)v1 * pw + u1) * C + c);
            r.x += wt * (a.x * w00 + bq.x * w01 + cq.x * w10 + d.x * w11);
            r.y += wt * (a.y * w00 + bq.y * w01 + cq.y * w10 + d.y * w11);
            r.z += wt * (a.z * w00 + bq.z * w01 + cq.z * w10 + d.z * w11);
            r.w += wt * (a.w * w00 + bq.w * w01 + cq.w * w10 + d.w * w11);
        }
        *reinterpret_cast<float4*>(o + c) = r;
    }
}

extern "C" int dtk_sample_grid(const float* feat, int T, int C, int ph, int pw, const float* pts, float* out, int B,
                               void* stream) {
    DTK_REQUIRE(feat && pts && out && B >= 0, "dtk_sample_grid: null pointer");
    DTK_REQUIRE(T > 0 && C > 0 && C % 4 == 0 && ph > 0 && pw > 0, "dtk_sample_grid: bad sizes (C must be a multiple of 4)");
    if (B == 0) return DTK_OK;
    DTK_LAUNCH("sample_grid", sample_grid_kernel, dim3(dtk_cdiv(B, 4)), dim3(256), 0, dtk_stream(stream), feat, pts, out, T,
               C, ph, pw, B);
    return DTK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// NormalizedConv2d.forward (models/networks/conv_norm.py:34-46) as a stand-alone layer: every (out, in) k x k kernel
// divided by its own sum (|sum| < 1e-8 -> sign * 1e-8), stride 1, zero padding k/2.  The tracker path never calls this
// (the two layers of TrackerHead.cnn_refiner live inside the fused head kernels); it exists so that the module's
// forward() is the same arithmetic on the device.  One thread per output element.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void normalized_conv2d_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                int B, int Cin, int Cout, int H, int W, int k) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * Cout * H * W;
    if (idx >= total) return;
    const int col = (int)(idx % W), row = (int)((idx / W) % H), co = (int)((idx / ((long long)W * H)) % Cout);
    const int b = (int)(idx / ((long long)W * H * Cout));
    const int pad = k / 2;
    float acc = bias ? bias[co] : 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
        const float* wk = w + ((size_t)co * Cin + ci) * k * k;
        float s = 0.f;
        for (int t = 0; t < k * k; ++t) s += wk[t];
        if (fabsf(s) < 1e-8f) s = (s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f)) * 1e-8f;
        const float* xp = x + ((size_t)b * Cin + ci) * H * W;
        float a = 0.f;
        for (int dy = 0; dy < k; ++dy) {
            const int rr = row + dy - pad;
            if (rr < 0 || rr >= H) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int cc = col + dx - pad;
                if (cc >= 0 && cc < W) a = fmaf(wk[dy * k + dx] / s, xp[(size_t)rr * W + cc], a);
            }
        }
        acc += a;
    }
    y[idx] = acc;
}

extern "C" int dtk_normalized_conv2d(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout,
                                     int H, int W, int k, void* stream) {
    DTK_REQUIRE(x && w && y, "dtk_normalized_conv2d: null pointer");
    DTK_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && k > 0 && (k & 1), "dtk_normalized_conv2d: bad sizes");
    const long long total = (long long)B * Cout * H * W;
    if (total == 0) return DTK_OK;
    DTK_LAUNCH("normalized_conv2d", normalized_conv2d_kernel, dim3(dtk_cdiv(total, 256)), dim3(256), 0, dtk_stream(stream), x,
               w, bias, y, B, Cin, Cout, H, W, k);
    return DTK_OK;
}

// ------------------------------------------------------------------------------------------------------------
// K14: cs[n][t] = <a,b> / (max(|a|,eps) max(|b|,eps)), a = S[n][tq[n]], b = S[n][t]; one wave per (n,t).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cos_sims_kernel(const float* __restrict__ S, const int32_t* __restrict__ tq,
                                                       float* __restrict__ cs, int N, int T, int C) {
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (long long)N * T) return;
    const int lane = threadIdx.x & 63;
    const int n = (int)(w / T);
    int q = tq[n];
    q = min(max(q, 0), T - 1);
    const float* a = S + ((size_t)n * T + q) * C;
    const float* b = S + (size_t)w * C;
    float ab = 0.f, aa = 0.f, bb = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 x = *reinterpret_cast<const float4*>(a + c), y = *reinterpret_cast<const float4*>(b + c);
        ab += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        aa += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        bb += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
    }
    ab = wave_sum(ab);
    aa = wave_sum(aa);
    bb = wave_sum(bb);
    if (lane == 0) cs[w] = ab / (fmaxf(sqrtf(aa), 1e-8f) * fmaxf(sqrtf(bb), 1e-8f));
}

extern "C" int dtk_traj_cos_sims(const float* S, const int32_t* tq, float* cs, int N, int T, int C, void* stream) {
    DTK_REQUIRE(S && tq && cs && N >= 0 && T > 0 && C > 0 && C % 4 == 0, "dtk_traj_cos_sims: bad args");
    if (N == 0) return DTK_OK;
    DTK_LAUNCH("cos_sims", cos_sims_kernel, dim3(dtk_cdiv((long long)N * T, 4)), dim3(256), 0, dtk_stream(stream), S, tq,
                       cs, N, T, C);
    return DTK_OK;
}
