// attn_bench.hip -- stand-alone timing + correctness harness for the ViT attention kernels (d_head = 64).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I dino_tracker_amd/csrc \
//         -I scripts/ubench scripts/ubench/attn_bench.hip -o /tmp/attn_bench && /tmp/attn_bench [frames] [S] [abl]
// Shapes of the benchmark: 30 frames x 6 heads, S = 8108 tokens.  Random Q / K / V^T in the operand type (fp16 and bf16
// are both run), Q pre-scaled like the QKV epilogue does.  Every variant is checked against a host fp64 softmax(QK^T)V on a
// sample of (frame, head, query) rows.  v2 = attention2_kernel (rounds 2-3), v3 = attention3_kernel (pipelined across
// key tiles inside a wave).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

#define ATT2_NS att2_f16
#define ATT2_T _Float16
#define ATT2_F16 1
#define ATT2_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#include "vit_attention2.h"
#ifndef ATTN_NO_V3
#include "attention3.h"
#endif
#include "vit_attention4.h"
#include "attention6.h"
#undef ATT2_NS
#undef ATT2_T
#undef ATT2_F16
#undef ATT2_MFMA
#define ATT2_NS att2_bf16
#define ATT2_T __bf16
#define ATT2_F16 0
#define ATT2_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#include "vit_attention2.h"
#ifndef ATTN_NO_V3
#include "attention3.h"
#endif
#include "vit_attention4.h"
#include "attention6.h"
#undef ATT2_NS
#undef ATT2_T
#undef ATT2_F16
#undef ATT2_MFMA

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float h2f(uint16_t v) { _Float16 h; memcpy(&h, &v, 2); return (float)h; }
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t v; memcpy(&v, &h, 2); return v; }
static float gauss(uint64_t& s) {
    auto u = [&]() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (float)((s >> 33) + 1) / 2147483649.0f; };
    return sqrtf(-2.f * logf(u())) * cosf(6.2831853f * u());
}

template <bool F16>
struct Run {
    typedef typename std::conditional<F16, _Float16, __bf16>::type T;
    static float dec(uint16_t v) { return F16 ? h2f(v) : bf2f(v); }
    static uint16_t enc(float f) { return F16 ? f2h(f) : f2bf(f); }

    static void go(int F, int S, bool abl) {
        const int heads = 6, D = 384;
        const int Sp = (S + 127) / 128 * 128, FH = F * heads;
        const size_t nqk = (size_t)FH * Sp * 64;
        std::vector<uint16_t> hq(nqk, 0), hk(nqk, 0), hv(nqk, 0);
        uint64_t seed = 1234;
        const float qs = 0.125f * 1.4426950408889634f;
        for (int fh = 0; fh < FH; ++fh)
            for (int s = 0; s < S; ++s)
                for (int d = 0; d < 64; ++d) {
                    hq[((size_t)fh * Sp + s) * 64 + d] = enc(gauss(seed) * qs * 1.5f);
                    hk[((size_t)fh * Sp + s) * 64 + d] = enc(gauss(seed));
                    hv[((size_t)fh * 64 + d) * Sp + s] = enc(gauss(seed));
                }
        // stress rows for the guard / safe-pass logic (frame 0, head 0): every key gets +50 in dimension 0; query 5 is scaled
        // so that its scores reach ~+-60, query 6 ~+-300, query 7 sits at about -300 for EVERY key
        for (int s2 = 0; s2 < S; ++s2) hk[((size_t)0 * Sp + s2) * 64 + 0] = enc(dec(hk[((size_t)0 * Sp + s2) * 64 + 0]) + 50.f);
        for (int d = 0; d < 64; ++d) {
            hq[((size_t)0 * Sp + 5) * 64 + d] = enc(dec(hq[((size_t)0 * Sp + 5) * 64 + d]) * 25.f);
            hq[((size_t)0 * Sp + 6) * 64 + d] = enc(dec(hq[((size_t)0 * Sp + 6) * 64 + d]) * 120.f);
            hq[((size_t)0 * Sp + 7) * 64 + d] = enc(d == 0 ? -6.f : 0.01f * dec(hq[((size_t)0 * Sp + 7) * 64 + d]));
        }
        T *q, *k, *vt, *o2;
        CK(hipMalloc(&q, nqk * 2)); CK(hipMalloc(&k, nqk * 2)); CK(hipMalloc(&vt, nqk * 2));
        const size_t no = (size_t)F * S * D;
        CK(hipMalloc(&o2, no * 2));
        CK(hipMemcpy(q, hq.data(), nqk * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(k, hk.data(), nqk * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(vt, hv.data(), nqk * 2, hipMemcpyHostToDevice));
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        const double flop = 4.0 * (double)S * S * 64 * FH;
        std::vector<uint16_t> ho(no);
        struct Row { int fh, qi; double o[64]; };
        std::vector<Row> rows;
        for (int i = 0; i < 48; ++i) {
            Row r; r.fh = (i * 37) % FH; r.qi = i == 0 ? 0 : (i == 1 ? S - 1 : (int)((uint64_t)(i * 2654435761u) % S));
            if (i >= 2 && i <= 6) { r.fh = 0; r.qi = i + 2; }  // the stress rows 5, 6, 7 and their neighbours 4, 8
            std::vector<double> p(S);
            double mx = -1e300;
            for (int s = 0; s < S; ++s) {
                double acc = 0;
                for (int d = 0; d < 64; ++d) acc += (double)dec(hq[((size_t)r.fh * Sp + r.qi) * 64 + d]) * dec(hk[((size_t)r.fh * Sp + s) * 64 + d]);
                p[s] = acc; mx = fmax(mx, acc);
            }
            double l = 0;
            for (int s = 0; s < S; ++s) { p[s] = exp2(p[s] - mx); l += p[s]; }
            for (int d = 0; d < 64; ++d) {
                double acc = 0;
                for (int s = 0; s < S; ++s) acc += p[s] * dec(hv[((size_t)r.fh * 64 + d) * Sp + s]);
                r.o[d] = acc / l;
            }
            rows.push_back(r);
        }
        auto check = [&](const char* name) {
            CK(hipMemcpy(ho.data(), o2, no * 2, hipMemcpyDeviceToHost));
            double worst = 0, scale = 0;
            int bad = 0;
            for (auto& r : rows) {
                const int frame = r.fh / heads, head = r.fh % heads;
                double rw = 0, rs = 0;
                for (int d = 0; d < 64; ++d) {
                    const double got = dec(ho[((size_t)frame * S + r.qi) * D + head * 64 + d]);
                    rw = fmax(rw, fabs(got - r.o[d])); rs = fmax(rs, fabs(r.o[d]));
                    if (!(got == got)) rw = 1e9;
                }
                if (rw > (F16 ? 1e-3 : 8e-3) * fmax(rs, 0.05) && bad++ < 6) printf("    row fh=%d q=%d: err %.3e (|ref| %.3e)\n", r.fh, r.qi, rw, rs);
                worst = fmax(worst, rw); scale = fmax(scale, rs);
            }
            printf("  %-24s max |err| vs fp64 on %zu rows: %.3e (max |ref| %.3e), %d rows beyond tolerance\n", name, rows.size(), worst, scale, bad);
        };
        auto timeit = [&](const char* name, auto launch) {
            CK(hipMemset(o2, 0xff, no * 2));
            for (int i = 0; i < 2; ++i) launch();
            CK(hipDeviceSynchronize());
            const int reps = 10;
            CK(hipEventRecord(a));
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
            printf("%-26s %8.3f ms  %7.1f TFLOP/s  (%.3f of 2.5 PF)\n", name, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 2500.0);
            check(name);
        };
        printf("== %s operands: %d frames x %d heads, S = %d (Sp %d), %.2f TFLOP per launch\n", F16 ? "fp16" : "bf16", F, heads, S, Sp, flop * 1e-12);
        using namespace att2c;
#define NSK(x) std::conditional<F16, std::true_type, std::false_type>::type::value ? 0 : 0
        auto v2 = [&](auto kern, int QT) { int QB; const unsigned g = attention2_grid(FH, S, QT, &QB);
            hipLaunchKernelGGL(kern, dim3(g), dim3(512), 0, 0, (const T*)q, (const T*)k, (const T*)vt, o2, S, Sp, heads, D, FH, QB); };
        auto v3 = [&](auto kern, int NW) { const int QB = (S + 32 * NW - 1) / (32 * NW); const unsigned g = (unsigned)(((FH + 7) / 8) * 8 * QB);
            hipLaunchKernelGGL(kern, dim3(g), dim3(64 * NW), 0, 0, (const T*)q, (const T*)k, (const T*)vt, o2, S, Sp, heads, D, FH, QB); };
        auto v4 = [&](auto kern) { int QB; const unsigned g = att2_f16::attention4_grid(FH, S, &QB);
            hipLaunchKernelGGL(kern, dim3(g), dim3(256), 0, 0, (const T*)q, (const T*)k, (const T*)vt, o2, S, Sp, heads, D, FH, QB); };
        auto v6 = [&](auto kern, auto nq_tag) { constexpr int NQ = decltype(nq_tag)::value; int QB; const unsigned g = att2_f16::attention6_grid<NQ>(FH, S, &QB);
            hipLaunchKernelGGL(kern, dim3(g), dim3(256), 0, 0, (const T*)q, (const T*)k, (const T*)vt, o2, S, Sp, heads, D, FH, QB); };
        if constexpr (F16) {
            if (getenv("ATTN_V6")) {   // round 6: NQ query tiles per wave (attention6.h: no guards -- the stress rows 5 .. 7 of (frame 0, head 0) are expected to fail)
                timeit("v4 (library form)", [&] { v4(att2_f16::attention4_kernel<0>); });
                timeit("v4 without guards (ABL 8)", [&] { v4(att2_f16::attention4_kernel<8>); });
                timeit("v6 NQ = 2 (64 q/wave)", [&] { v6(att2_f16::attention6_kernel<2>, std::integral_constant<int, 2>{}); });
                timeit("v6 NQ = 3 (96 q/wave)", [&] { v6(att2_f16::attention6_kernel<3>, std::integral_constant<int, 3>{}); });
                timeit("v6 NQ = 4 (128 q/wave)", [&] { v6(att2_f16::attention6_kernel<4>, std::integral_constant<int, 4>{}); });
                timeit("v4 (library form), again", [&] { v4(att2_f16::attention4_kernel<0>); });
                timeit("v6 NQ = 4, again", [&] { v6(att2_f16::attention6_kernel<4>, std::integral_constant<int, 4>{}); });
                timeit("v6 NQ = 4: no LDS reads", [&] { v6(att2_f16::attention6_kernel<4, 128>, std::integral_constant<int, 4>{}); });
                timeit("v6 NQ = 4: no exp", [&] { v6(att2_f16::attention6_kernel<4, 1>, std::integral_constant<int, 4>{}); });
                CK(hipFree(q)); CK(hipFree(k)); CK(hipFree(vt)); CK(hipFree(o2));
                return;
            }
            timeit("v4 (1 wave/SIMD, 64 q/wave)", [&] { v4(att2_f16::attention4_kernel<0>); });
            timeit("v4 row sums on the matrix pipe", [&] { v4(att2_f16::attention4_kernel<0, true>); });
            if (abl) {
                timeit("v4 abl: no exp", [&] { v4(att2_f16::attention4_kernel<1>); });
                timeit("v4 abl: no barrier", [&] { v4(att2_f16::attention4_kernel<16>); });
                timeit("v4 abl: no DMA", [&] { v4(att2_f16::attention4_kernel<2>); });
                timeit("v4 abl: no LDS reads", [&] { v4(att2_f16::attention4_kernel<128>); });
                timeit("v4 abl: MFMA + cvt only", [&] { v4(att2_f16::attention4_kernel<1 | 2 | 8 | 16 | 128>); });
                auto clocked = [&](const char* name, auto kern) {
                    // shader-clock cycles (s_memtime) of one workgroup's three phases; the effective clock follows from the key
                    // loop's slot count and the wall time.  The stamps land in the first words of the output, which workgroup
                    // 0 (a different one) also writes: read them from a second launch into a scratch copy of O.
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(a)); v4(kern); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    unsigned w3[3]; CK(hipMemcpy(w3, o2, 12, hipMemcpyDeviceToHost));
                    int QB; const unsigned g = att2_f16::attention4_grid(FH, S, &QB);
                    const double rounds = g / 256.0, ntiles = (S + 63) / 64, tot = (double)w3[0] + w3[1] + w3[2];
                    printf("%-26s prologue %u, key loop %u (%.1f per MFMA slot), epilogue %u cycles; %.3f ms -> %.2f GHz if every round took as long (%.1f rounds)\n",
                           name, w3[0], w3[1], w3[1] / (ntiles * 32), w3[2], ms, tot * rounds / (ms * 1e6), rounds);
                };
                clocked("v4 clock", att2_f16::attention4_kernel<256>);
                clocked("v4 clock, MFMA + cvt only", att2_f16::attention4_kernel<256 | 1 | 2 | 8 | 16 | 128>);
            }
            timeit("v2 QT1 max", [&] { v2(att2_f16::attention2_kernel<1, 0, 0, true>, 1); });
            timeit("v2 QT1 opt (library)", [&] { v2(att2_f16::attention2_kernel<1, 0, 1, true>, 1); });
            timeit("v2 K block 1 from global (ABL 32)", [&] { v2(att2_f16::attention2_kernel<1, 32, 1, true>, 1); });
            timeit("v2 K odd d steps from global (ABL 64)", [&] { v2(att2_f16::attention2_kernel<1, 64, 1, true>, 1); });
            timeit("v2 QT1 opt (library), again", [&] { v2(att2_f16::attention2_kernel<1, 0, 1, true>, 1); });
            // (-DATTN_NO_V3: without the v3 instantiations.  The out-of-line safe pass is ONE function per translation unit: next
            // to attention3 (<= 256 registers) it is compiled with 178 and drags every attention2 variant here to 2 waves per
            // SIMD, while the library's (attention2 only) has 128 registers / 4 waves -- 3.5 ms here against 3.0 ms there.)
#ifndef ATTN_NO_V3
            if (abl > 1) {
                timeit("v3 8 waves, ring 5", [&] { v3(att2_f16::attention3_kernel<8, 5>, 8); });
                timeit("v3 8 waves, ring 4", [&] { v3(att2_f16::attention3_kernel<8, 4>, 8); });
                timeit("v3 4 waves, ring 4", [&] { v3(att2_f16::attention3_kernel<4, 4>, 4); });
            }
#endif
            if (abl) {
                timeit("v2 abl: no exp", [&] { v2(att2_f16::attention2_kernel<1, 1, 1, true>, 1); });
                timeit("v2 abl: MFMA + cvt only", [&] { v2(att2_f16::attention2_kernel<1, 1 | 2 | 4 | 8 | 16, 1, true>, 1); });
#ifndef ATTN_NO_V3
                timeit("v3 abl: no exp", [&] { v3(att2_f16::attention3_kernel<8, 5, 1>, 8); });
                timeit("v3 abl: no barrier", [&] { v3(att2_f16::attention3_kernel<8, 5, 16>, 8); });
#endif
            }
        } else {
            timeit("v2 QT1 opt (library)", [&] { v2(att2_bf16::attention2_kernel<1, 0, 1, true>, 1); });
            timeit("v4 (1 wave/SIMD, 64 q/wave)", [&] { v4(att2_bf16::attention4_kernel<0>); });
            timeit("v4 row sums on the matrix pipe", [&] { v4(att2_bf16::attention4_kernel<0, true>); });
#ifndef ATTN_NO_V3
            if (abl > 1) timeit("v3 8 waves, ring 5", [&] { v3(att2_bf16::attention3_kernel<8, 5>, 8); });
#endif
        }
        CK(hipFree(q)); CK(hipFree(k)); CK(hipFree(vt)); CK(hipFree(o2));
    }
};

int main(int argc, char** argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 30, S = argc > 2 ? atoi(argv[2]) : 8108;
    const bool abl = argc > 3;
    Run<true>::go(F, S, abl);
    Run<false>::go(F, S, false);
    return 0;
}
