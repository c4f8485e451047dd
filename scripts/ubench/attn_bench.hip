// attn_bench.hip -- stand-alone timing + correctness harness for the ViT attention kernels (d_head = 64).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I dino_tracker_amd/csrc \
//         scripts/ubench/attn_bench.hip -o scripts/ubench/attn_bench && scripts/ubench/attn_bench [frames] [S]
// Shapes of the benchmark: 30 frames x 6 heads, S = 8108 tokens.  Random Q / K / V^T (bf16), Q pre-scaled like the QKV
// epilogue does.  Every variant is checked against a host fp64 softmax(QK^T)V on a sample of (frame, head, query) rows.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../dino_tracker_amd/csrc/vit.hip"
#include "attention_v1.h"

void dtk_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
void dtk_prof_begin(const char*, hipStream_t) {}
void dtk_prof_end(const char*, hipStream_t) {}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float gauss(uint64_t& s) {
    auto u = [&]() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (float)((s >> 33) + 1) / 2147483649.0f; };
    return sqrtf(-2.f * logf(u())) * cosf(6.2831853f * u());
}

int main(int argc, char** argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 30, S = argc > 2 ? atoi(argv[2]) : 8108, heads = 6, D = 384;
    const int Sp = (S + 127) / 128 * 128, FH = F * heads;
    const size_t nqk = (size_t)FH * Sp * 64;
    std::vector<uint16_t> hq(nqk, 0), hk(nqk, 0), hv(nqk, 0);
    uint64_t seed = 1234;
    const float qs = 0.125f * 1.4426950408889634f;
    for (int fh = 0; fh < FH; ++fh)
        for (int s = 0; s < S; ++s)
            for (int d = 0; d < 64; ++d) {
                hq[((size_t)fh * Sp + s) * 64 + d] = f2bf(gauss(seed) * qs * 1.5f);
                hk[((size_t)fh * Sp + s) * 64 + d] = f2bf(gauss(seed));
                hv[((size_t)fh * 64 + d) * Sp + s] = f2bf(gauss(seed));
            }
    // stress rows for the guard / safe-pass logic of the optimistic softmax (frame 0, head 0): every key gets +50 in
    // dimension 0; query 5 is scaled so that its scores reach ~+-60 (row sums beyond 2^40: power-of-two rescale), query 6 so
    // that they reach ~+-300 (inf: poisoned, safe pass), query 7 sits at about -300 for EVERY key (all-zero row: safe pass)
    for (int s2 = 0; s2 < S; ++s2) hk[((size_t)0 * Sp + s2) * 64 + 0] = f2bf(bf2f(hk[((size_t)0 * Sp + s2) * 64 + 0]) + 50.f);
    for (int d = 0; d < 64; ++d) {
        hq[((size_t)0 * Sp + 5) * 64 + d] = f2bf(bf2f(hq[((size_t)0 * Sp + 5) * 64 + d]) * 25.f);
        hq[((size_t)0 * Sp + 6) * 64 + d] = f2bf(bf2f(hq[((size_t)0 * Sp + 6) * 64 + d]) * 120.f);
        hq[((size_t)0 * Sp + 7) * 64 + d] = f2bf(d == 0 ? -6.f : 0.01f * bf2f(hq[((size_t)0 * Sp + 7) * 64 + d]));
    }
    bf16_t *q, *k, *vt, *o1, *o2;
    CK(hipMalloc(&q, nqk * 2)); CK(hipMalloc(&k, nqk * 2)); CK(hipMalloc(&vt, nqk * 2));
    const size_t no = (size_t)F * S * D;
    CK(hipMalloc(&o1, no * 2)); CK(hipMalloc(&o2, no * 2));
    CK(hipMemcpy(q, hq.data(), nqk * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(k, hk.data(), nqk * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(vt, hv.data(), nqk * 2, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double flop = 4.0 * (double)S * S * 64 * FH;
    std::vector<uint16_t> ho(no);
    // host reference rows
    struct Row { int fh, qi; double o[64]; };
    std::vector<Row> rows;
    for (int i = 0; i < 48; ++i) {
        Row r; r.fh = (i * 37) % FH; r.qi = i == 0 ? 0 : (i == 1 ? S - 1 : (int)((uint64_t)(i * 2654435761u) % S));
        if (i >= 2 && i <= 6) { r.fh = 0; r.qi = i + 2; }  // the stress rows 5, 6, 7 and their neighbours 4, 8
        std::vector<double> p(S);
        double mx = -1e300;
        for (int s = 0; s < S; ++s) {
            double acc = 0;
            for (int d = 0; d < 64; ++d) acc += (double)bf2f(hq[((size_t)r.fh * Sp + r.qi) * 64 + d]) * bf2f(hk[((size_t)r.fh * Sp + s) * 64 + d]);
            p[s] = acc; mx = fmax(mx, acc);
        }
        double l = 0;
        for (int s = 0; s < S; ++s) { p[s] = exp2(p[s] - mx); l += p[s]; }
        for (int d = 0; d < 64; ++d) {
            double acc = 0;
            for (int s = 0; s < S; ++s) acc += p[s] * bf2f(hv[((size_t)r.fh * 64 + d) * Sp + s]);
            r.o[d] = acc / l;
        }
        rows.push_back(r);
    }
    auto check = [&](bf16_t* o, const char* name) {
        CK(hipMemcpy(ho.data(), o, no * 2, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0;
        for (auto& r : rows) {
            const int frame = r.fh / heads, head = r.fh % heads;
            double rw = 0, rs = 0;
            for (int d = 0; d < 64; ++d) {
                const double got = bf2f(ho[((size_t)frame * S + r.qi) * D + head * 64 + d]);
                rw = fmax(rw, fabs(got - r.o[d])); rs = fmax(rs, fabs(r.o[d]));
            }
            if (rw > 2e-3) printf("    row fh=%d q=%d: err %.3e (|ref| %.3e)\n", r.fh, r.qi, rw, rs);
            worst = fmax(worst, rw); scale = fmax(scale, rs);
        }
        printf("  %-22s max |err| vs fp64 on %zu rows: %.3e (max |ref| %.3e)\n", name, rows.size(), worst, scale);
    };
    auto timeit = [&](const char* name, auto launch, bf16_t* o) {
        CK(hipMemset(o, 0xff, no * 2));
        for (int i = 0; i < 2; ++i) launch();
        CK(hipDeviceSynchronize());
        const int reps = 10;
        CK(hipEventRecord(a));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
        printf("%-22s %8.3f ms  %7.1f TFLOP/s  (%.3f of 2.5 PF)\n", name, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 2500.0);
        check(o, name);
    };
    printf("attention d=64: %d frames x %d heads, S = %d (Sp %d), %.2f TFLOP per launch\n", F, heads, S, Sp, flop * 1e-12);
    timeit("v1 (round 1)", [&] { hipLaunchKernelGGL(attention_kernel, dim3(dtk_cdiv(S, 128 * ATT_QT), FH), dim3(256), 0, 0, q, k, vt, o1, S, Sp, heads, D); }, o1);
#define V2_RUN(QT_, MODE_, PIN_, NAME) do { int QB; const unsigned g = att2::attention2_grid(FH, S, QT_, &QB); \
        timeit(NAME, [&] { hipLaunchKernelGGL((att2::attention2_kernel<QT_, 0, MODE_, PIN_>), dim3(g), dim3(512), 0, 0, q, k, vt, o2, S, Sp, heads, D, FH, QB); }, o2); } while (0)
    V2_RUN(1, 0, true, "v2 QT1 max pin");
    V2_RUN(1, 1, true, "v2 QT1 opt pin");
    V2_RUN(1, 1, false, "v2 QT1 opt nopin");
    V2_RUN(2, 1, true, "v2 QT2 opt pin");
    { int QB; const unsigned g = att2::attention2_grid(FH, S, 1, &QB);
      timeit("v2 QT1 opt pin prio", [&] { hipLaunchKernelGGL((att2::attention2_kernel<1, 0, 1, true, true>), dim3(g), dim3(512), 0, 0, q, k, vt, o2, S, Sp, heads, D, FH, QB); }, o2);
      timeit("v2 QT1 opt nopin prio", [&] { hipLaunchKernelGGL((att2::attention2_kernel<1, 0, 1, false, true>), dim3(g), dim3(512), 0, 0, q, k, vt, o2, S, Sp, heads, D, FH, QB); }, o2); }
    if (argc > 3) {  // ablations of v2 QT=1 (results are wrong by construction; timing only)
        int QB; const unsigned g = att2::attention2_grid(FH, S, 1, &QB);
#define ABL_RUN(A, NAME) timeit(NAME, [&] { hipLaunchKernelGGL((att2::attention2_kernel<1, A, 1, true>), dim3(g), dim3(512), 0, 0, q, k, vt, o2, S, Sp, heads, D, FH, QB); }, o2)
        ABL_RUN(1, "abl: no exp");
        ABL_RUN(2, "abl: no DMA");
        ABL_RUN(4, "abl: one LDS address");
        ABL_RUN(8, "abl: no max logic");
        ABL_RUN(16, "abl: no barrier");
        ABL_RUN(1 | 8, "abl: no exp, no max");
        ABL_RUN(2 | 4, "abl: no DMA, one LDS");
        ABL_RUN(1 | 2 | 4 | 8 | 16, "abl: MFMA + cvt only");
    }
    return 0;
}
