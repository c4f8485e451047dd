// vit_attention4.h -- flash attention for d_head = 64, third generation (round 4): ONE wave per SIMD, 64 queries per wave.
//
// Include AFTER vit_attention2.h while its ATT2_* macros are still defined: this header re-opens the same per-type namespace
// and reuses the tile layout (K tile + V^T tile of 64 keys, 16-byte pieces XOR-swizzled through the DMA's source address),
// the fp16 range guards (MODE 1 of attention2_kernel: estimated reference, RESC_T / POISON_T, safe pass) and the output
// layout.  What changes is the shape of the work inside a CU.
//
// Why: attention2's wave owns 32 queries, so every 32x32x16 MFMA needs its own 1 KB
// A fragment from LDS -- at the full MFMA rate that is 128 B per clock per CU, the LDS peak -- and its 16 waves per CU run
// QK^T -> softmax -> PV in sequence behind one barrier per tile: its ablation with NO softmax and NO DMA tops out at 1.30 PF.
// Here a workgroup is 4 waves, one per SIMD, with the whole register file (launch bound 256 threads, 1 wave per SIMD),
// and a wave owns TWO 32-query tiles:
//   * a K fragment and a V^T fragment are read from LDS ONCE per key tile and kept in registers for both query tiles
//     (kf[8], vf[8]: 64 registers): LDS reads per MFMA halve;
//   * the two query tiles run half a key tile out of phase ("ping-pong").  A half-step issues 16 MFMAs that belong to one
//     query tile (8 of S^T = K Q^T, 8 of O^T += V^T P^T) and the 16 VALU chunks of the OTHER tile's softmax (2 exponentials,
//     row-sum adds, 1 packed convert each): inside a half-step the matrix stream and the vector stream have no dependence on
//     each other at all, so one MFMA is followed by one chunk, pinned with sched_barrier -- the ~5 issue slots a lone wave
//     hides under a 32x32x16 MFMA (MI355X_MICROARCH.md, "one wave per SIMD").  Only ONE score set per query tile is live
//     (attention3's pipelining across key tiles needed two);
//        E(t):  MFMA  S(1,t) [K(t)], PV(1,t-1) [V(t-1)]    VALU  softmax(0,t)     LDS: kf <- K(t+1), vf <- V(t) after last use
//        O(t):  MFMA  PV(0,t) [V(t)], S(0,t+1) [K(t+1)]    VALU  softmax(1,t)
//   * K / V^T tiles by LDS-DMA into a ring of four 16 KB buffers, requested three tiles ahead, one barrier per key tile.
// Arithmetic is attention2's MODE 1 (optimistic exponentials against an estimated reference, guarded).
#ifndef ATT2_NS
#error "include vit_attention2.h first and keep ATT2_NS, ATT2_T, ATT2_F16, ATT2_MFMA defined"
#endif

namespace ATT2_NS {

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef ATT2_T op2 __attribute__((ext_vector_type(2)));
constexpr int A4_NB = 4;        // LDS ring (buffers of TILE_BYTES)
constexpr int A4_AHEAD = 3;     // tiles requested ahead of the one whose V^T part is being read

// O^T += A B with the accumulator PINNED to the AGPR half of the register file.  The O accumulators (64 registers) are only
// ever touched by these MFMAs (and by the rare rescale / the epilogue), so with them in AGPRs everything the VALU works on
// -- scores, P, the fragment registers -- fits the 256 architectural VGPRs; left to the register allocator (builtin MFMA,
// -amdgpu-mfma-vgpr-form) the accumulators wandered between the two halves through v_accvgpr copies inside the key loop.
// The compiler's hazard recognizer does not see through an asm statement: the only instructions that read these registers
// outside the MFMAs themselves are behind agpr_settle() below.
__device__ __forceinline__ void mfma_acc_agpr(f16v& c, op8 a, op8 b) {
    if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// an MFMA result is read by a non-MFMA instruction: 16-pass MFMA -> up to 18 wait states (and more for a dependent chain in flight)
__device__ __forceinline__ void agpr_settle() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); }

inline unsigned attention4_grid(int FH, int S, int* qb_out) {
    const int QB = (S + 255) / 256;
    *qb_out = QB;
    return (unsigned)(((FH + 7) / 8) * 8 * QB);
}

// ABL (micro-benchmark only; 0 in the library): 1 no exponentials, 2 no LDS-DMA after the prologue, 8 no guards / estimate,
// 16 no barrier, 128 no LDS fragment reads after the prologue.  PKADD: row sums as packed adds (v_pk_add_f32) instead of a
// scalar chain.
template <int ABL = 0, bool PKADD = false>
__global__ __launch_bounds__(256, 1) void attention4_kernel(const op_t* __restrict__ Q, const op_t* __restrict__ Kg,
                                                            const op_t* __restrict__ Vt, op_t* __restrict__ O, int S, int Sp,
                                                            int heads, int D, int FH, int QB) {
    __shared__ __attribute__((aligned(1024))) unsigned char tiles[A4_NB][TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int fh = (seq / QB) * 8 + xcd;
    const int qb = seq % QB;
    if (fh >= FH) return;
    const int frame = fh / heads, head = fh - frame * heads;
    const int q0 = qb * 256 + w * 64;
    const int lq = lane & 31, hi = lane >> 5;
    const op_t* Qb = Q + (size_t)fh * Sp * 64;
    const op_t* Kb = Kg + (size_t)fh * Sp * 64;
    const op_t* Vb = Vt + (size_t)fh * 64 * Sp;
    if (F16) fp16_saturate_mode();

    // DMA sources of this lane: wave w fills rows 16w .. 16w+15 of the K part and of the V^T part (two requests each)
    const op_t* ksrc[2];
    const op_t* vsrc[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int lrow = (w * 2 + r) * 8 + (lane >> 3), lpc = (lane & 7) ^ ((lrow >> 1) & 7);
        ksrc[r] = Kb + (size_t)lrow * 64 + lpc * 8;
        vsrc[r] = Vb + (size_t)lrow * Sp + lpc * 8;
    }
    const unsigned lds_base = (unsigned)(size_t)&tiles[0][0];
    const int ntiles = (S + 63) / 64;
    auto issue = [&](int t, int buf) {
        const int tt = min(t, ntiles - 1);  // past the end: a harmless repeat keeps the request count per tile uniform
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            glds16(ksrc[r] + (size_t)tt * 64 * 64,
                   __builtin_amdgcn_readfirstlane(lds_base + buf * TILE_BYTES + (w * 2 + r) * 1024));
            glds16(vsrc[r] + (size_t)tt * 64,
                   __builtin_amdgcn_readfirstlane(lds_base + buf * TILE_BYTES + 8192 + (w * 2 + r) * 1024));
        }
    };
    // Q^T fragments (B operand): lane (query lq, hi) holds d = 16 ks + 8 hi .. + 7
    op8 qf[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qrow = min(q0 + qt * 32 + lq, Sp - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[qt][ks] = *reinterpret_cast<const op8*>(Qb + (size_t)qrow * 64 + ks * 16 + hi * 8);
    }
#pragma unroll
    for (int i = 0; i < A4_AHEAD; ++i) issue(i, i);

    f16v o[2][2];   // O^T accumulators [query tile][d block]: rows d = 32 db + (r & 3) + 8 (r >> 2) + 4 hi, column = query lq
    f16v sc[2][2];  // S^T accumulators [query tile][key block]: register r of lane-half hi = key 32 b + 16 (r >> 3) + 8 hi + (r & 7)
    u4v pf[2][4];   // P^T fragments [query tile][16-key group], as the four 32-bit words of an op8
    op8 kf[8], vf[8];
    float m_run[2] = {0.f, 0.f};
    float l_run[2] = {0.f, 0.f};
    int has_m = 0;   // wave-uniform (kept scalar): the reference of some query of the wave is not 0
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qt][db][r] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) pf[0][g][e] = pf[1][g][e] = 0u;
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) vf[g][e] = (op_t)0.f;   // E(0) multiplies them with P = 0

    const int krow = (lq & 19) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    int koff[2], voff[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int r = b * 32 + krow;
        koff[b] = r * 128 + ((hi ^ ((r >> 1) & 7)) << 4);
        const int d = b * 32 + lq;
        voff[b] = 8192 + d * 128 + ((hi ^ ((d >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[qt][ks]));
    // reference estimate (attention2 MODE 1): keys 0..63 and the query tile's own 32 keys
    if (!(ABL & 8)) {
        int far = 0;
        float est[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float tm = -3e38f;
#pragma unroll
            for (int blk = 0; blk < 3; ++blk) {
                const int kr0 = blk < 2 ? blk * 32 : q0 + qt * 32;
                const op_t* kp = Kb + (size_t)min(kr0 + lq, Sp - 1) * 64 + hi * 8;
                op8 kq[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kq[ks] = *reinterpret_cast<const op8*>(kp + ks * 16);
                f16v so = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) so = ATT2_MFMA(kq[ks], qf[qt][ks], so, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kr0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    tm = fmaxf(tm, key < S ? so[r] : -3e38f);
                }
            }
            float a, b;
            halves(tm, a, b);
            est[qt] = fmaxf(a, b);
            far |= __any(!(fabsf(est[qt]) <= 3.f)) ? 1 : 0;
        }
        if (__builtin_amdgcn_readfirstlane(far)) {
            has_m = 1;
            m_run[0] = est[0];
            m_run[1] = est[1];
        }
    }
    vm_wait<0>();  // tiles 0 .. A4_AHEAD-1 have landed
    __syncthreads();

    const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto ldk = [&](const unsigned char* tb, int f) { return *reinterpret_cast<const op8*>(tb + (koff[f & 1] ^ ((f >> 1) << 5))); };
    auto ldv = [&](const unsigned char* tb, int g) { return *reinterpret_cast<const op8*>(tb + (voff[g & 1] ^ ((g >> 1) << 5))); };
    auto mask_tail = [&](f16v (&s2)[2], int t) {  // keys beyond S (last tile only)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 64 + b * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                if (key >= S) s2[b][r] = -1e30f;
            }
    };
    // rare: a lane's 32-key part of the tile's row sum of query tile qt passed RESC_T (attention2 MODE 1, same arithmetic);
    // called after the PV product of that tile has been issued
    auto guard_tripped = [&](int qt, float lsum) {
        asm volatile("; guard tripped" ::: "memory");
        agpr_settle();
        has_m = 1;   // (unconditionally: with m_run = 0 the subtracting body computes the same values)
        float a, b;
        halves(lsum, a, b);
        const float tot = a + b;
        if (!(a < POISON_T && b < POISON_T)) {
            l_run[qt] = __builtin_nanf("");
        } else if (tot >= RESC_T) {
            const float k = floorf(__builtin_amdgcn_logf(tot));
            const float alpha = __builtin_amdgcn_exp2f(-k);
            m_run[qt] += k;
            l_run[qt] *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qt][db][r] *= alpha;
            has_m = true;
        }
    };

    // prologue: K(0) fragments, S(0,0)
#pragma unroll
    for (int f = 0; f < 8; ++f) kf[f] = ldk(&tiles[0][0], f);
#pragma unroll
    for (int f = 0; f < 8; ++f) sc[0][f & 1] = ATT2_MFMA(kf[f], qf[0][f >> 1], f < 2 ? zero16 : sc[0][f & 1], 0, 0, 0);
    if (ntiles == 1 && (S & 63) != 0) mask_tail(sc[0], 0);

    // One half-step.  SQ = the query tile whose softmax runs on the VALU; the MFMAs belong to the other one (MQ): its scores
    // against the key tile held in kf and its PV product with the V^T tile held in vf.  RELOAD (E half-steps): every fragment
    // register is refilled from LDS (tk: K part of the next key tile, tv: V^T part of this one) right after its last use.
    // Returns this lane's part of the row sum of query tile SQ over the key tile.
    auto half_step = [&](auto sq_tag, auto sub_tag, auto reload_tag, const unsigned char* tk, const unsigned char* tv) -> float {
        constexpr int SQ = decltype(sq_tag)::value, MQ = 1 - SQ;
        constexpr bool SUB = decltype(sub_tag)::value, RELOAD = decltype(reload_tag)::value;
        const float nm = -m_run[SQ];
        float lt = 0.f;
        f2 lt2 = {0.f, 0.f};
        // VALU chunk c = values 2c, 2c+1 of the 32 per lane: 16-key group bj = c >> 2, element pair e = 2 (c & 3)
        auto chunk = [&](int c) {
            const int bj = c >> 2, e = 2 * (c & 3);
            float s0 = sc[SQ][bj >> 1][8 * (bj & 1) + e], s1 = sc[SQ][bj >> 1][8 * (bj & 1) + e + 1];
            if (SUB) { s0 += nm; s1 += nm; }
            const float p0 = (ABL & 1) ? s0 * 0.01f : __builtin_amdgcn_exp2f(s0);
            const float p1 = (ABL & 1) ? s1 * 0.01f : __builtin_amdgcn_exp2f(s1);
            if (PKADD) lt2 += f2{p0, p1};
            else { lt += p0; lt += p1; }
            const op2 pk = {(op_t)p0, (op_t)p1};
            unsigned wv = __builtin_bit_cast(unsigned, pk);
            // anchors: the chunk's results are demanded HERE, in front of the slot's sched_barrier (pure arithmetic is otherwise
            // free to sink to the end of the half-step, where nothing hides it)
            if (PKADD) asm volatile("" : "+v"(wv), "+v"(lt2));
            else asm volatile("" : "+v"(wv), "+v"(lt));
            pf[SQ][bj][e >> 1] = wv;
        };
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = i >> 1;
            if ((i & 1) == 0) {
                sc[MQ][j & 1] = ATT2_MFMA(kf[j], qf[MQ][j >> 1], j < 2 ? zero16 : sc[MQ][j & 1], 0, 0, 0);
                if (RELOAD && !(ABL & 128)) kf[j] = ldk(tk, j);
            } else {
                mfma_acc_agpr(o[MQ][j & 1], vf[j], __builtin_bit_cast(op8, pf[MQ][j >> 1]));
                if (RELOAD && !(ABL & 128)) vf[j] = ldv(tv, j);
            }
            chunk(i);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (PKADD) lt = lt2[0] + lt2[1];
        l_run[SQ] += lt;
        return lt;
    };

    int pend1 = 0;   // wave-uniform: query tile 1's guard tripped in O(t-1); handled after PV(1,t-1), i.e. after E(t)
    float lsum1 = 0.f;
    // one key tile; returns (wave-uniform) whether a guard tripped in it
    auto key_tile = [&](auto sub_tag, int t) -> int {
        const int cb = t & (A4_NB - 1);
        vm_wait<4>();   // this wave's requests of tile t+1 have landed (those of tile t+2 stay in flight)
        if (!(ABL & 16)) __syncthreads();   // every wave is past E(t-1): buffer (t-1) % NB is free, tile t+1 is visible
        if (!((ABL & 2) && t > 0)) issue(t + A4_AHEAD, (cb + A4_AHEAD) & (A4_NB - 1));
        const unsigned char* tv = &tiles[cb][0];
        const unsigned char* tk = &tiles[(cb + 1) & (A4_NB - 1)][0];
        // ---- E(t): S(1,t), PV(1,t-1) || softmax(0,t); fragment registers refilled with K(t+1) / V(t)
        const float lsum0 = half_step(std::integral_constant<int, 0>{}, sub_tag, std::true_type{}, tk, tv);
        if (t == ntiles - 1 && (S & 63) != 0) mask_tail(sc[1], t);
        if (pend1) { guard_tripped(1, lsum1); pend1 = 0; }
        // ---- O(t): PV(0,t), S(0,t+1) || softmax(1,t)
        lsum1 = half_step(std::integral_constant<int, 1>{}, sub_tag, std::false_type{}, tk, tv);
        if (t + 1 == ntiles - 1 && (S & 63) != 0) mask_tail(sc[0], t + 1);
        if (ABL & 8) return 0;
        const int trip0 = __builtin_amdgcn_readfirstlane(__any(!(lsum0 < RESC_T)));
        pend1 = __builtin_amdgcn_readfirstlane(__any(!(lsum1 < RESC_T)));
        if (trip0) guard_tripped(0, lsum0);
        return trip0 | pend1;
    };
    // Two loops instead of a branch per half-step (a branch inside the loop body made the register allocator shuttle the
    // accumulators between the two bodies' assignments): while every reference is 0 the exponentials take the raw scores
    // (no subtraction); the first guard event -- or a far estimate -- moves on to the subtracting loop for good.
    int t = 0;
    if (!has_m) {
        for (; t < ntiles; ++t)
            if (key_tile(std::false_type{}, t)) { ++t; break; }
    }
    for (; t < ntiles; ++t) key_tile(std::true_type{}, t);
    // epilogue: PV(1, last)
#pragma unroll
    for (int g = 0; g < 8; ++g) mfma_acc_agpr(o[1][g & 1], vf[g], __builtin_bit_cast(op8, pf[1][g >> 1]));
    if (pend1) guard_tripped(1, lsum1);
    agpr_settle();
    vm_wait<0>();
    float l_tot[2];
    bool redo = false;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float a, b;
        halves(l_run[qt], a, b);
        l_tot[qt] = a + b;
        redo |= __any(!(l_tot[qt] > LOW_T && l_tot[qt] < 0x1p120f));
    }
    if (redo && !(ABL & 8)) {
        safe_pass<4>(Qb, Kb, Vb, O + (size_t)frame * S * D + head * 64, q0, 64, S, Sp, D);
        return;
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
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
