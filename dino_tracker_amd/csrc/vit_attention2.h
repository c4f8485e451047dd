// vit_attention2.h -- flash attention for d_head = 64, second generation (round 2).
//
// What changed against attention_kernel (vit.hip), and why (profiles/r01_bench_kernel_trace.md: 0.37 of the MFMA peak):
//   * 8 (or 16) waves per CU instead of 4: a workgroup is 512 threads = 8 waves x 32 queries, compiled for <= 128 VGPRs so
//     that TWO workgroups share a CU (4 waves per SIMD).  With d_head = 64 the softmax costs ~3.5 VALU instructions per
//     score against 0.5 MFMA: one wave per SIMD can hide ~4-5 issue slots under a 32x32x16 MFMA (MI355X_MICROARCH.md,
//     "one wave per SIMD"), i.e. it is VALU-issue bound by construction; with several waves per SIMD one wave's exponentials
//     run under another wave's MFMAs (the pipes are separate).
//   * K / V^T tiles arrive by LDS-DMA (global_load_lds_dwordx4), three buffers, requested two tiles ahead, ONE barrier per
//     64-key tile; no staging registers, no ds_write pass.  The LDS image is the DMA's lane-linear one with the 16-byte
//     pieces of a row XOR-swizzled through the per-lane SOURCE address (piece ^ ((row >> 1) & 7)): every ds_read_b128 of a
//     fragment then touches 16 distinct 16-byte slots (conflict-free), which the padded pitches of round 1 achieved with
//     register staging only.
//   * The K fragment is read with its rows permuted (quads 1 and 2 of every 16 rows swapped) so that, in the accumulator
//     of S^T = K Q^T, a lane holds 8 CONSECUTIVE keys per 16-key group: the P^T fragment is still a plain repack of the
//     accumulator and the V^T fragment becomes ONE ds_read_b128 (round 1: two ds_read_b64 at a key stride).
//   * workgroup -> (frame, head, query block) mapping keeps all query blocks of one (frame, head) on one XCD (blocks are
//     dispatched round-robin over the 8 XCDs), so that its 2 MB of K / V^T are fetched from HBM once and re-read from that
//     XCD's 4 MB L2 by the other blocks (round 1: 4.5x the algorithmic HBM traffic, PMC).
// The arithmetic is the one of attention_kernel: exp2-domain online softmax, deferred maximum (threshold 8), raw v_exp_f32,
// scores arriving as s - m through the C operand of the first MFMA, bf16 P, fp32 accumulation.
#pragma once

namespace att2 {

typedef __bf16 bf16_t;
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int TILE_KEYS = 64;
constexpr int TILE_BYTES = 2 * TILE_KEYS * 64 * 2;  // K tile (64 keys x 64 d) + V^T tile (64 d x 64 keys), bf16
constexpr int NBUF = 3;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
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
// Combine a per-lane value with the one of lane ^ 32 (the two halves of a wave hold the two key halves of a query) without
// an LDS round trip: v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second.
// NB the two results are copied into scalars BEFORE the bit cast: `__builtin_bit_cast(float, sw[1])` applied to the vector
// element directly reads element 0 under hipcc / ROCm 7.2 (seen in the ISA: both uses came from the first register), which
// silently drops the other half.
__device__ __forceinline__ void halves(float x, float& lo_all, float& hi_all) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const unsigned r0 = sw[0], r1 = sw[1];
    lo_all = __uint_as_float(r0);  // the value held by the lower-half lane of the pair (lanes 0..31), in both lanes
    hi_all = __uint_as_float(r1);  // the value held by the upper-half lane
}
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// Safe pass of ONE wave (rare: only after a poisoned row sum, MODE 1): queries q0 .. q0+nq-1 of one (frame, head) again,
// 32 at a time, with a running maximum per 64-key tile; fragments come straight from global memory in the layouts of the
// main loop (K rows permuted so that a lane holds 8 consecutive keys per 16-key group), no LDS, no barriers.
__device__ __noinline__ void safe_pass(const bf16_t* Qb, const bf16_t* Kb, const bf16_t* Vb, bf16_t* Ob, int q0, int nq, int S,
                                       int Sp, int D) {
    const int lane = threadIdx.x & 63, lq = lane & 31, hi = lane >> 5;
    const int krow = (lq & 19) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    const int ntiles = (S + 63) / 64;
#pragma unroll 1
    for (int qq = q0; qq < q0 + nq; qq += 32) {
        bf8 qf[4];
        const int qrow = min(qq + lq, Sp - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf8*>(Qb + (size_t)qrow * 64 + ks * 16 + hi * 8);
        f16v oa[2];
        float m = -3e38f;
        f2 l = {0.f, 0.f};
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) oa[db][r] = 0.f;
#pragma unroll 1
        for (int t = 0; t < ntiles; ++t) {
            f16v s2[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s2[b][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf8 kf = *reinterpret_cast<const bf8*>(Kb + (size_t)(t * 64 + b * 32 + krow) * 64 + ks * 16 + hi * 8);
                    s2[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s2[b], 0, 0, 0);
                }
            }
            float tm = -3e38f;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + b * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                    if (key >= S) s2[b][r] = -1e30f;
                    tm = fmaxf(tm, s2[b][r]);
                }
            tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
            const float mn = fmaxf(m, tm);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            l *= f2{alpha, alpha};
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oa[db][r] *= alpha;
#pragma unroll
            for (int bj = 0; bj < 4; ++bj) {
                bf8 pfr;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(s2[bj >> 1][8 * (bj & 1) + e] - m);
                    l[e & 1] += pv;
                    pfr[e] = (bf16_t)pv;
                }
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf8 vf = *reinterpret_cast<const bf8*>(Vb + (size_t)(db * 32 + lq) * Sp + t * 64 + bj * 16 + hi * 8);
                    oa[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pfr, oa[db], 0, 0, 0);
                }
            }
        }
        const float lh = l[0] + l[1];
        const float inv = 1.f / (lh + __shfl_xor(lh, 32, 64));
        const int qi = qq + lq;
        if (qi < S) {
            bf16_t* orow = Ob + (size_t)qi * D;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int d = db * 32 + 8 * rq + 4 * hi;
                    bf4 v = {(bf16_t)(oa[db][4 * rq + 0] * inv), (bf16_t)(oa[db][4 * rq + 1] * inv),
                             (bf16_t)(oa[db][4 * rq + 2] * inv), (bf16_t)(oa[db][4 * rq + 3] * inv)};
                    *reinterpret_cast<bf4*>(orow + d) = v;
                }
        }
    }
}

// QT: 32-query tiles per wave (1: 256 queries per workgroup, <= 128 VGPRs; 2: 512 queries, <= 256 VGPRs)
// grid: 1-D, 8 * ceil(F*heads / 8) * QB blocks, QB = ceil(S / (256 * QT))
// ABL: ablation switches of the micro-benchmark (scripts/ubench/attn_bench.hip); 0 in the library.
//   1: no exponentials (p = score * c)   2: no LDS-DMA after the first two tiles   4: every fragment read hits ONE LDS address
//   8: no row maximum / deferred-max logic  16: no barrier
// MODE 0: running maximum per tile (deferred rescale, threshold 8), as in round 1.
// MODE 1: OPTIMISTIC exponentials.  With d_head = 64 the softmax costs as many VALU cycles as the tile costs MFMA cycles
//   (PMC: SQ_ACTIVE_INST_VALU 3.5 M cycles per SIMD against 2.9 M of SQ_VALU_MFMA_BUSY_CYCLES per launch), and a third of
//   them only maintain the running maximum.  Any reference m gives the same softmax as long as 2^(s - m) neither overflows
//   nor underflows to an all-zero row (P is bf16 = fp32's exponent range, O and l accumulate in fp32).  So the reference
//   starts at 0 -- p = exp2(s), no subtraction at all -- and a guard on the tile's row sum (one compare per tile:
//   not (sum < 2^40), which also catches inf / NaN) makes the wave move the reference up by an exact power of two AFTER the
//   tile's PV product (nothing has overflowed yet: 2^40 is 87 binades below fp32's limit); from then on that wave
//   subtracts its reference like MODE 0 does.  A score more than 120 above the reference in ONE step (83 nats), or a row
//   whose scores all sit 100 binades below 0, poisons the row sum (NaN / 0) instead; a wave that finds such a sum at the end
//   redoes its queries in a safe pass (running maximum per tile, operands read straight from global memory, no barriers).
//   tests/test_gpu_p1.py forces all three events.
template <int QT, int ABL = 0, int MODE = 1, bool PIN = true, bool PRIO = false>
__global__ __launch_bounds__(512, QT == 1 ? 4 : 2) void attention2_kernel(const bf16_t* __restrict__ Q,
                                                                          const bf16_t* __restrict__ Kg,
                                                                          const bf16_t* __restrict__ Vt,
                                                                          bf16_t* __restrict__ O, int S, int Sp, int heads,
                                                                          int D, int FH, int QB) {
    __shared__ __attribute__((aligned(1024))) unsigned char tiles[NBUF][TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware mapping: block b runs on XCD b % 8; within an XCD the sequence b / 8 walks (fh of that XCD, query block)
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int fh = (seq / QB) * 8 + xcd;
    const int qb = seq % QB;
    if (fh >= FH) return;
    const int frame = fh / heads, head = fh - frame * heads;
    const int q0 = qb * (256 * QT) + w * (32 * QT);
    const int lq = lane & 31, hi = lane >> 5;
    const bf16_t* Qb = Q + (size_t)fh * Sp * 64;
    const bf16_t* Kb = Kg + (size_t)fh * Sp * 64;
    const bf16_t* Vb = Vt + (size_t)fh * 64 * Sp;

    // ---- DMA source of this lane: wave w fills rows 8w .. 8w+7 of the K tile and of the V^T tile (one request each);
    // LDS slot (row, piece') holds global piece  piece' ^ ((row >> 1) & 7)
    const int lrow = w * 8 + (lane >> 3), lpc = (lane & 7) ^ ((lrow >> 1) & 7);
    const bf16_t* ksrc = Kb + (size_t)lrow * 64 + lpc * 8;        // + t * 64 * 64
    const bf16_t* vsrc = Vb + (size_t)lrow * Sp + lpc * 8;        // + t * 64
    const unsigned lds_base = (unsigned)(size_t)&tiles[0][0];
    auto issue = [&](int t, int buf) {
        if ((ABL & 2) && t > 1) return;
        glds16(ksrc + (size_t)t * 64 * 64, lds_base + buf * TILE_BYTES + w * 1024);
        glds16(vsrc + (size_t)t * 64, lds_base + buf * TILE_BYTES + 8192 + w * 1024);
    };

    // Q^T fragments (B operand): lane (query lq, hi) holds d = 16*ks + 8*hi .. +7 for ks = 0..3
    bf8 qf[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qrow = min(q0 + qt * 32 + lq, Sp - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[qt][ks] = *reinterpret_cast<const bf8*>(Qb + (size_t)qrow * 64 + ks * 16 + hi * 8);
    }
    const int ntiles = (S + 63) / 64;
    issue(0, 0);
    issue(min(1, ntiles - 1), 1);

    f16v o[QT][2];  // O^T accumulators: d-block db: rows d = 32*db + (r&3) + 8*(r>>2) + 4*hi, column = query lq
    float m_run[QT];
    f2 l_run[QT];
    bool has_m = false;  // MODE 1, wave-uniform: a guard has tripped, the reference of some query is not 0 any more
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = 0.f;
        l_run[qt] = f2{0.f, 0.f};
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qt][db][r] = 0.f;
    }
    // fragment addresses inside a buffer.  K: MFMA row lq carries key row pi(lq) (quads 1 <-> 2 of each 16 swapped), so
    // that accumulator registers r = 8j .. 8j+7 of lane-half hi are the consecutive keys 16j + 8hi .. + 7 of the block.
    const int krow = (lq & 19) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    int koff[2], voff[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int r = b * 32 + krow;
        koff[b] = r * 128 + ((hi ^ ((r >> 1) & 7)) << 4);   // piece 2*ks + hi: the ks part is XORed in below (bits 1..2)
        const int d = b * 32 + lq;                          // here b = d-block
        voff[b] = 8192 + d * 128 + ((hi ^ ((d >> 1) & 7)) << 4);
    }
    // touch the Q fragments before the loop (their vmcnt wait must not end up inside it, behind the DMA requests)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[qt][ks]));
    vm_wait<2>();   // tile 0 landed (this wave's two requests of tile 1 may still fly)
    __syncthreads();

    int buf = 0;
    for (int t = 0; t < ntiles; ++t) {
        {
            const int nb = buf >= 1 ? buf - 1 : 2;  // (t + 2) % 3
            issue(min(t + 2, ntiles - 1), nb);
        }
        const unsigned char* tb = &tiles[buf][0];
        // Fragment streams are software-pipelined BY HAND (ring of four 16-byte fragments, three reads ahead, order pinned
        // with sched_barrier): left alone, hipcc under the 128-VGPR cap issues ds_read -> s_waitcnt lgkmcnt(0) -> MFMA one
        // fragment at a time and every MFMA eats a full LDS latency (measured: -21 % when the reads are taken away).
        auto ldk = [&](int f) {  // K fragment f: d step ks = f >> 1 of key block b = f & 1 (accumulators alternate)
            return *reinterpret_cast<const bf8*>((ABL & 4) ? tb + lane * 16 : tb + (koff[f & 1] ^ ((f >> 1) << 5)));
        };
        auto ldv = [&](int g) {  // V^T fragment g: key group bj = g >> 1 (b = bj >> 1, j = bj & 1) of d-block db = g & 1
            return *reinterpret_cast<const bf8*>((ABL & 4) ? tb + 8192 + lane * 16 : tb + (voff[g & 1] ^ ((g >> 1) << 5)));
        };
        // ---- S^T = K Q^T : two 32-key blocks x four 16-wide d steps ----
        f16v sc[QT][2];
        bf8 kr[4], vr[4];
        auto scores = [&]() {
            const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // inline constant C
            kr[0] = ldk(0); kr[1] = ldk(1); kr[2] = ldk(2);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                if (f + 3 < 8) kr[(f + 3) & 3] = ldk(f + 3);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    sc[qt][f & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[f & 3], qf[qt][f >> 1],
                                                                            f < 2 ? zero16 : sc[qt][f & 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            // keys beyond S (last tile only) are masked out; register r of lane-half hi = key 32b + 16(r>>3) + 8hi + (r&7)
            if (t == ntiles - 1 && (S & 63) != 0) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = t * 64 + b * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                            if (key >= S) sc[qt][b][r] = -1e30f;
                        }
            }
        };
        scores();
        // ---- online softmax (exp2 domain; Q carries log2(e)/sqrt(d)) : everything per query is lane-local ----
        bf8 pf[QT][4];  // P^T fragments: [query tile][16-key group bj]
        float lsum[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            auto tile_max = [&]() {
                float tm = -3e38f;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tm = fmaxf(tm, sc[qt][b][r]);
                // the other half of the query's keys sits in lane ^ 32: one v_permlane32_swap (no LDS round trip)
                float a, b;
                halves(tm, a, b);
                return fmaxf(a, b);
            };
            auto rescale_to = [&](float mn) {
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - mn);  // (harmless on the first tile: l = o = 0)
                m_run[qt] = mn;
                l_run[qt] *= f2{alpha, alpha};
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qt][db][r] *= alpha;
            };
            // exponentials of the tile against the current reference; returns this lane's part of the row sum
            auto exps = [&](bool sub) {
                const f2 nm = {-m_run[qt], -m_run[qt]};
                f2 lt = {0.f, 0.f};
#pragma unroll
                for (int bj = 0; bj < 4; ++bj)
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        f2 sv = {sc[qt][bj >> 1][8 * (bj & 1) + e], sc[qt][bj >> 1][8 * (bj & 1) + e + 1]};
                        if (sub) sv += nm;
                        const f2 p = (ABL & 1) ? sv * f2{0.01f, 0.01f} : f2{__builtin_amdgcn_exp2f(sv[0]), __builtin_amdgcn_exp2f(sv[1])};
                        lt += p;
                        pf[qt][bj][e] = (bf16_t)p[0];
                        pf[qt][bj][e + 1] = (bf16_t)p[1];
                    }
                return lt;
            };
            if (MODE == 0) {
                // deferred maximum: m_run only moves (and O, l are only rescaled) when some query of the wave sees a score
                // more than 8 above it, so P stays <= 2^8.  The first tile always takes the branch.
                if (!(ABL & 8)) {
                    const float tm = tile_max();
                    if (t == 0 || !__all(tm - m_run[qt] <= 8.f)) {
                        asm volatile("; rescale" ::: "memory");
                        rescale_to(t == 0 ? tm : fmaxf(tm, m_run[qt]));
                    }
                }
                l_run[qt] += exps(true);
            } else {
                const f2 lt = has_m ? exps(true) : exps(false);  // wave-uniform branch: no subtraction while m == 0
                lsum[qt] = lt[0] + lt[1];
                l_run[qt] += lt;
            }
        }
        bool resc = false;
        if (MODE == 1 && !(ABL & 8)) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) resc |= __any(!(lsum[qt] < 0x1p40f));
        }
        // ---- O^T += V^T P^T : A fragment of (d-block db, keys 16 bj + 8hi .. +7) = one 16-byte read ----
        vr[0] = ldv(0); vr[1] = ldv(1); vr[2] = ldv(2);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + 3 < 8) vr[(g + 3) & 3] = ldv(g + 3);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
                o[qt][g & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[g & 3], pf[qt][g >> 1], o[qt][g & 1], 0, 0, 0);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (MODE == 1 && resc) {  // rare: a row sum of this tile passed 2^40 -> move the reference up by a power of two
            asm volatile("; guard tripped" ::: "memory");
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float a, b;
                halves(lsum[qt], a, b);
                const float tot = a + b;  // both key halves
                if (!(tot < 0x1p120f)) {
                    l_run[qt] = f2{__builtin_nanf(""), __builtin_nanf("")};  // beyond repair here: safe pass at the end
                } else if (tot >= 0x1p40f) {
                    const float k = floorf(__builtin_amdgcn_logf(tot));  // v_log_f32 = log2
                    const float alpha = __builtin_amdgcn_exp2f(-k);      // exact power of two
                    m_run[qt] += k;
                    l_run[qt] *= f2{alpha, alpha};
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[qt][db][r] *= alpha;
                    has_m = true;
                }
            }
        }
        vm_wait<2>();   // this wave's requests of tile t+1 have landed; those of tile t+2 stay in flight
        if (!(ABL & 16)) __syncthreads();
        buf = buf == 2 ? 0 : buf + 1;
    }
    vm_wait<0>();
    float l_tot[QT];
    bool redo = false;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float l_half = l_run[qt][0] + l_run[qt][1];
        float a, b;
        halves(l_half, a, b);
        l_tot[qt] = a + b;
        if (MODE == 1) redo |= __any(!(l_tot[qt] > 0x1p-100f && l_tot[qt] < 0x1p120f));
    }
    if (MODE == 1 && redo && !(ABL & 8)) {
        // rare: see safe_pass (kept out of line so that it costs the main loop no registers)
        safe_pass(Qb, Kb, Vb, O + (size_t)frame * S * D + head * 64, q0, 32 * QT, S, Sp, D);
        return;
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float inv = 1.f / l_tot[qt];
        const int qi = q0 + qt * 32 + lq;
        if (qi < S) {
            bf16_t* orow = O + ((size_t)frame * S + qi) * D + head * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int d = db * 32 + 8 * rq + 4 * hi;
                    bf4 v = {(bf16_t)(o[qt][db][4 * rq + 0] * inv), (bf16_t)(o[qt][db][4 * rq + 1] * inv),
                             (bf16_t)(o[qt][db][4 * rq + 2] * inv), (bf16_t)(o[qt][db][4 * rq + 3] * inv)};
                    *reinterpret_cast<bf4*>(orow + d) = v;
                }
        }
    }
}

inline unsigned attention2_grid(int FH, int S, int QT, int* qb_out) {
    const int QB = (S + 256 * QT - 1) / (256 * QT);
    *qb_out = QB;
    return (unsigned)(((FH + 7) / 8) * 8 * QB);
}

}  // namespace att2
