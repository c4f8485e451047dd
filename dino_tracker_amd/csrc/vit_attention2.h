// vit_attention2.h -- flash attention for d_head = 64, second generation (round 2; operand type a parameter since round 3).
//
// Included once per operand type (no include guard): the includer defines
//   ATT2_NS    namespace of this instantiation            (att2_f16 / att2_bf16)
//   ATT2_T     operand type of Q / K / V^T / P / O        (_Float16 / __bf16)
//   ATT2_F16   1 when ATT2_T is _Float16
//   ATT2_MFMA  the 32x32x16 MFMA builtin for that type
// fp16 is the default of the library since round 3: the same MFMA rate as bf16 with 8x less operand rounding (round 2's
// end-to-end error from the VIDEO, p99 1.4e-3 px, was the bf16 operands of P1).  fp16's narrow exponent range is what the
// guards of MODE 1 below are sized for.
//
// What changed against attention_kernel (round 1), and why (profiles/r01_bench_kernel_trace.md: 0.37 of the MFMA peak):
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
// Arithmetic: exp2-domain online softmax, raw v_exp_f32, 16-bit P (RTN pack: v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32), fp32
// accumulation of O and of the row sums.
#ifndef ATT2_NS
#error "define ATT2_NS, ATT2_T, ATT2_F16 and ATT2_MFMA before including vit_attention2.h"
#endif

#ifndef DTK_ATT2_COMMON
#define DTK_ATT2_COMMON
namespace att2c {
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

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
// MODE register bit 23 (FP16_OVFL): an fp16 result that overflows is clamped to +-65504 instead of becoming +-inf
// (conversions included), so that a rare out-of-range activation saturates instead of poisoning everything downstream.
__device__ __forceinline__ void fp16_saturate_mode() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }

inline unsigned attention2_grid(int FH, int S, int QT, int* qb_out) {
    const int QB = (S + 256 * QT - 1) / (256 * QT);
    *qb_out = QB;
    return (unsigned)(((FH + 7) / 8) * 8 * QB);
}
}  // namespace att2c
#endif

namespace ATT2_NS {
using namespace att2c;

typedef ATT2_T op_t;
typedef ATT2_T op8 __attribute__((ext_vector_type(8)));
typedef ATT2_T op4 __attribute__((ext_vector_type(4)));
constexpr bool F16 = ATT2_F16 != 0;
// MODE 1 thresholds on a lane's part (32 keys) of a tile's row sum: at RESC_T the reference moves up after the tile; at
// POISON_T a P entry may have left the operand type's range (fp16: 65504), the row is redone by the safe pass
constexpr float RESC_T = F16 ? 0x1p9f : 0x1p40f;
constexpr float POISON_T = F16 ? 0x1p15f : 0x1p120f;
constexpr float LOW_T = F16 ? 0x1p-7f : 0x1p-100f;  // a final row sum below this has lost P's precision (fp16 subnormals)

constexpr int TILE_KEYS = 64;
constexpr int TILE_BYTES = 2 * TILE_KEYS * 64 * 2;  // K tile (64 keys x 64 d) + V^T tile (64 d x 64 keys), bf16
constexpr int NBUF = 3;

// Safe pass of ONE wave (rare: only after a poisoned row sum, MODE 1): queries q0 .. q0+nq-1 of one (frame, head) again,
// 32 at a time, with a running maximum per 64-key tile; fragments come straight from global memory in the layouts of the
// main loop (K rows permuted so that a lane holds 8 consecutive keys per 16-key group), no LDS, no barriers.
// TAG: one instantiation per kernel family -- an out-of-line device function is compiled ONCE per instantiation with the
// register budget of its most generous caller, and every caller then inherits that allocation (round 3: next to a
// 256-register kernel the shared safe pass cost the 128-register kernel half its occupancy).
// (safe_pass_impl: the body, inlined where the caller wants the pass inside its OWN register budget -- attention5 runs two waves
//  per SIMD and an out-of-line callee is compiled to the caller's VGPR budget without knowing about its AGPRs)
__device__ __forceinline__ void safe_pass_impl(const op_t* Qb, const op_t* Kb, const op_t* Vb, op_t* Ob, int q0, int nq, int S,
                                               int Sp, int D) {
    const int lane = threadIdx.x & 63, lq = lane & 31, hi = lane >> 5;
    const int krow = (lq & 19) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    const int ntiles = (S + 63) / 64;
#pragma unroll 1
    for (int qq = q0; qq < q0 + nq; qq += 32) {
        op8 qf[4];
        const int qrow = min(qq + lq, Sp - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const op8*>(Qb + (size_t)qrow * 64 + ks * 16 + hi * 8);
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
                    const op8 kf = *reinterpret_cast<const op8*>(Kb + (size_t)(t * 64 + b * 32 + krow) * 64 + ks * 16 + hi * 8);
                    s2[b] = ATT2_MFMA(kf, qf[ks], s2[b], 0, 0, 0);
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
                op8 pfr;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(s2[bj >> 1][8 * (bj & 1) + e] - m);
                    l[e & 1] += pv;
                    pfr[e] = (op_t)pv;
                }
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const op8 vf = *reinterpret_cast<const op8*>(Vb + (size_t)(db * 32 + lq) * Sp + t * 64 + bj * 16 + hi * 8);
                    oa[db] = ATT2_MFMA(vf, pfr, oa[db], 0, 0, 0);
                }
            }
        }
        const float lh = l[0] + l[1];
        const float inv = 1.f / (lh + __shfl_xor(lh, 32, 64));
        const int qi = qq + lq;
        if (qi < S) {
            op_t* orow = Ob + (size_t)qi * D;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int d = db * 32 + 8 * rq + 4 * hi;
                    op4 v = {(op_t)(oa[db][4 * rq + 0] * inv), (op_t)(oa[db][4 * rq + 1] * inv),
                             (op_t)(oa[db][4 * rq + 2] * inv), (op_t)(oa[db][4 * rq + 3] * inv)};
                    *reinterpret_cast<op4*>(orow + d) = v;
                }
        }
    }
}
template <int TAG>
__device__ __noinline__ void safe_pass(const op_t* Qb, const op_t* Kb, const op_t* Vb, op_t* Ob, int q0, int nq, int S,
                                       int Sp, int D) {
    safe_pass_impl(Qb, Kb, Vb, Ob, q0, nq, S, Sp, D);
}

// QT: 32-query tiles per wave (1: 256 queries per workgroup, <= 128 VGPRs; 2: 512 queries, <= 256 VGPRs)
// grid: 1-D, 8 * ceil(F*heads / 8) * QB blocks, QB = ceil(S / (256 * QT))
// ABL: ablation switches of the micro-benchmark (scripts/ubench/attn_bench.hip); 0 in the library.
//   1: no exponentials (p = score * c)   2: no LDS-DMA after the first two tiles   4: every fragment read hits ONE LDS address
//   8: no row maximum / deferred-max logic  16: no barrier
// MODE 0: running maximum per tile (deferred rescale, threshold 8), as in round 1.
// MODE 1: OPTIMISTIC exponentials.  With d_head = 64 the softmax costs as many VALU cycles as the tile costs MFMA cycles
//   (PMC: SQ_ACTIVE_INST_VALU 3.5 M cycles per SIMD against 2.9 M of SQ_VALU_MFMA_BUSY_CYCLES per launch), and a third of
//   them only maintain the running maximum.  Any reference m gives the same softmax as long as 2^(s - m) stays inside the
//   range in which P keeps its precision: bf16 has fp32's exponent range; fp16 (the default) ends at 65504 above and has
//   full precision down to 2^-14 only.  So the reference is ESTIMATED ONCE per query -- the maximum of its scores against
//   the first 64 keys (which hold the CLS token) and against the 32 keys of its own wave (which hold its own key: the
//   self score is the usual row maximum of a trained ViT), i.e. a lower bound of the row maximum, hence p_max >= 1 -- and
//   is kept at exactly 0 (p = exp2(s), no subtraction at all) while every estimate of the wave is within +-3.  After that
//   only a guard on the tile's row sum runs (one compare per tile: not (a lane's 32-key part < RESC_T), which also catches
//   inf / NaN): the wave moves the reference up by an exact power of two AFTER the tile's PV product -- nothing has
//   overflowed yet while the part stays below POISON_T (fp16: RESC_T = 2^9, POISON_T = 2^15: every p < 65504) -- and from
//   then on subtracts its reference.  A part beyond POISON_T in ONE step (fp16: a score 15 binades = 10 nats above
//   everything the row has seen; bf16: 120 binades) poisons the row sum instead; a wave that finds a poisoned or a tiny
//   (< LOW_T) sum at the end redoes its queries in a safe pass (running maximum per tile, operands read straight from
//   global memory, no barriers).  The reference never exceeds the row maximum by more than 6 binades (a rescale sets it to
//   floor(log2(tile sum)) <= log2(64 p_max)), so p_max >= 2^-6 and the entries that matter at 11 bits are normal numbers.
//   tests/test_gpu_p1.py forces all of these events.
template <int QT, int ABL = 0, int MODE = 1, bool PIN = true, bool PRIO = false>
__global__ __launch_bounds__(512, QT == 1 ? 4 : 2) void attention2_kernel(const op_t* __restrict__ Q,
                                                                          const op_t* __restrict__ Kg,
                                                                          const op_t* __restrict__ Vt,
                                                                          op_t* __restrict__ O, int S, int Sp, int heads,
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
    const op_t* Qb = Q + (size_t)fh * Sp * 64;
    const op_t* Kb = Kg + (size_t)fh * Sp * 64;
    const op_t* Vb = Vt + (size_t)fh * 64 * Sp;

    // ---- DMA source of this lane: wave w fills rows 8w .. 8w+7 of the K tile and of the V^T tile (one request each);
    // LDS slot (row, piece') holds global piece  piece' ^ ((row >> 1) & 7)
    const int lrow = w * 8 + (lane >> 3), lpc = (lane & 7) ^ ((lrow >> 1) & 7);
    const op_t* ksrc = Kb + (size_t)lrow * 64 + lpc * 8;        // + t * 64 * 64
    const op_t* vsrc = Vb + (size_t)lrow * Sp + lpc * 8;        // + t * 64
    const unsigned lds_base = (unsigned)(size_t)&tiles[0][0];
    auto issue = [&](int t, int buf) {
        if ((ABL & 2) && t > 1) return;
        glds16(ksrc + (size_t)t * 64 * 64, lds_base + buf * TILE_BYTES + w * 1024);
        glds16(vsrc + (size_t)t * 64, lds_base + buf * TILE_BYTES + 8192 + w * 1024);
    };

    // Q^T fragments (B operand): lane (query lq, hi) holds d = 16*ks + 8*hi .. +7 for ks = 0..3
    op8 qf[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qrow = min(q0 + qt * 32 + lq, Sp - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[qt][ks] = *reinterpret_cast<const op8*>(Qb + (size_t)qrow * 64 + ks * 16 + hi * 8);
    }
    const int ntiles = (S + 63) / 64;
    issue(0, 0);
    issue(min(1, ntiles - 1), 1);

    f16v o[QT][2];  // O^T accumulators: d-block db: rows d = 32*db + (r&3) + 8*(r>>2) + 4*hi, column = query lq
    float m_run[QT];
    f2 l_run[QT];
    bool has_m = false;  // MODE 1, wave-uniform: the reference of some query of the wave is not 0
    if (F16) fp16_saturate_mode();
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
    // MODE 1: the reference estimate (see above), before the main loop so that the loop itself carries none of it: scores
    // of the wave's queries against keys 0..63 and against the wave's OWN 32 keys (rows q0 + 32 qt ..), K fragments straight
    // from global memory (any row order: only the maximum is used)
    if (MODE == 1 && !(ABL & 8)) {
        bool far = false;
        float est[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float tm = -3e38f;
#pragma unroll
            for (int blk = 0; blk < 3; ++blk) {
                const int kr0 = blk < 2 ? blk * 32 : q0 + qt * 32;
                const op_t* kp = Kb + (size_t)min(kr0 + lq, Sp - 1) * 64 + hi * 8;
                op8 kf[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const op8*>(kp + ks * 16);
                f16v so = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) so = ATT2_MFMA(kf[ks], qf[qt][ks], so, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kr0 + (r & 3) + 8 * (r >> 2) + 4 * hi;  // D row of the MFMA = key row of the block
                    tm = fmaxf(tm, key < S ? so[r] : -3e38f);
                }
            }
            float a, b;
            halves(tm, a, b);
            est[qt] = fmaxf(a, b);
            far |= __any(!(fabsf(est[qt]) <= 3.f));
        }
        if (far) {
            has_m = true;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) m_run[qt] = est[qt];
        }
    }
    vm_wait<0>();   // tiles 0 and 1 landed
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
            // experiment (ABL & 32 / 64, micro-benchmark only): the fragments of key block 1 (32) or of both blocks' odd d steps
            // (64) come straight from global memory through the texture path instead of LDS -- a quarter of the fragment reads
            // of the scores, an eighth of all: is the LDS read rate the wall?
            if (((ABL & 32) && (f & 1)) || ((ABL & 64) && ((f >> 1) & 1)))
                return *reinterpret_cast<const op8*>(Kb + (size_t)min(t * 64 + (f & 1) * 32 + krow, Sp - 1) * 64 + (f >> 1) * 16 + hi * 8);
            return *reinterpret_cast<const op8*>((ABL & 4) ? tb + lane * 16 : tb + (koff[f & 1] ^ ((f >> 1) << 5)));
        };
        auto ldv = [&](int g) {  // V^T fragment g: key group bj = g >> 1 (b = bj >> 1, j = bj & 1) of d-block db = g & 1
            return *reinterpret_cast<const op8*>((ABL & 4) ? tb + 8192 + lane * 16 : tb + (voff[g & 1] ^ ((g >> 1) << 5)));
        };
        // ---- S^T = K Q^T : two 32-key blocks x four 16-wide d steps ----
        f16v sc[QT][2];
        op8 kr[4], vr[4];
        auto scores = [&]() {
            const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // inline constant C
            kr[0] = ldk(0); kr[1] = ldk(1); kr[2] = ldk(2);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                if (f + 3 < 8) kr[(f + 3) & 3] = ldk(f + 3);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    sc[qt][f & 1] = ATT2_MFMA(kr[f & 3], qf[qt][f >> 1],
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
        op8 pf[QT][4];  // P^T fragments: [query tile][16-key group bj]
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
                        pf[qt][bj][e] = (op_t)p[0];
                        pf[qt][bj][e + 1] = (op_t)p[1];
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
            for (int qt = 0; qt < QT; ++qt) resc |= __any(!(lsum[qt] < RESC_T));
        }
        // ---- O^T += V^T P^T : A fragment of (d-block db, keys 16 bj + 8hi .. +7) = one 16-byte read ----
        vr[0] = ldv(0); vr[1] = ldv(1); vr[2] = ldv(2);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + 3 < 8) vr[(g + 3) & 3] = ldv(g + 3);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
                o[qt][g & 1] = ATT2_MFMA(vr[g & 3], pf[qt][g >> 1], o[qt][g & 1], 0, 0, 0);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (MODE == 1 && resc) {  // rare: a row sum of this tile passed RESC_T -> move the reference up by a power of two
            asm volatile("; guard tripped" ::: "memory");
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float a, b;
                halves(lsum[qt], a, b);
                const float tot = a + b;  // both key halves
                if (!(a < POISON_T && b < POISON_T)) {
                    l_run[qt] = f2{__builtin_nanf(""), __builtin_nanf("")};  // beyond repair here: safe pass at the end
                } else if (tot >= RESC_T) {
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
        if (MODE == 1) redo |= __any(!(l_tot[qt] > LOW_T && l_tot[qt] < 0x1p120f));
    }
    if (MODE == 1 && redo && !(ABL & 8)) {
        // rare: see safe_pass (kept out of line so that it costs the main loop no registers)
        safe_pass<2>(Qb, Kb, Vb, O + (size_t)frame * S * D + head * 64, q0, 32 * QT, S, Sp, D);
        return;
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float inv = 1.f / l_tot[qt];
        const int qi = q0 + qt * 32 + lq;
        if (qi < S) {
            op_t* orow = O + ((size_t)frame * S + qi) * D + head * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int d = db * 32 + 8 * rq + 4 * hi;
                    op4 v = {(op_t)(o[qt][db][4 * rq + 0] * inv), (op_t)(o[qt][db][4 * rq + 1] * inv),
                             (op_t)(o[qt][db][4 * rq + 2] * inv), (op_t)(o[qt][db][4 * rq + 3] * inv)};
                    *reinterpret_cast<op4*>(orow + d) = v;
                }
        }
    }
}


}  // namespace ATT2_NS
