// vit_attention5.h -- flash attention for d_head = 64, EXPERIMENT of round 5: TWO waves per SIMD (8 waves per workgroup), 64
// queries per wave, the two waves of a SIMD alternating a matrix phase and a vector phase.
//
// Include AFTER vit_attention4.h (same per-type namespace: tile layout, swizzle, guards, safe pass, helpers are reused).
// attention4 overlaps the MFMAs of one query tile with the softmax of the other INSIDE a wave (one wave per SIMD, 512 registers).
// Here a wave has 256 registers and does one thing at a time; the overlap comes from its partner on the same SIMD:
//     step      M_a(t): PV(1, t-1), S(0, t)      V_a(t): softmax(0, t); Vt fragments <- V(t);   request K(t+3)
//               M_b(t): PV(0, t),   S(1, t)      V_b(t): softmax(1, t); K fragments  <- K(t+1); request V(t+3)
// every step ends in a workgroup barrier; waves 4..7 run the same program ONE STEP LATE, so on every SIMD one wave issues 16 MFMAs
// while the other issues the 16 softmax chunks (2 exponentials, 2 row-sum adds, 1 packed convert each) + 8 fragment reads + a DMA
// request.  What makes it fit 256 registers: ONE score set and ONE P set per wave (a P set is consumed by the matrix phase that
// follows the vector phase that made it), and the reference enters the scores by initialising the accumulators with -m (32 v_mov
// in the matrix phase, where the VALU is idle) instead of through a 16-register C operand per query tile.
// Tile t+1 is complete in LDS for everybody before anybody reads it: a wave waits for its own two requests of tile t+1 at the end
// of V_a(t) (vmcnt(3): K(t+2), V(t+2), K(t+3) may stay in flight), one barrier before the first reader (a wave of the early half in
// V_b(t), two steps after the late half's V_a(t)... see the step table in DESIGN / NEGATIVE_RESULTS).
// Guards as attention4 (RESC_T / POISON_T / LOW_T, safe pass); a tripped guard of query tile q is handled at the start of the wave's
// next vector phase, when PV(q, t) has been issued and no score of q computed against the old reference is waiting.
#ifndef ATT2_NS
#error "include vit_attention2.h / vit_attention4.h first and keep ATT2_NS, ATT2_T, ATT2_F16, ATT2_MFMA defined"
#endif

namespace ATT2_NS {

inline unsigned attention5_grid(int FH, int S, int* qb_out) {
    const int QB = (S + 511) / 512;
    *qb_out = QB;
    return (unsigned)(((FH + 7) / 8) * 8 * QB);
}

// ABL (micro-benchmark): 8 = no guards / estimate, 32 = both halves in phase (no one-step delay of waves 4..7),
// 64 = no vector work (exponentials, row sums, converts; refills and requests stay), 128 = no matrix work (the 16 MFMAs of a step)
template <int ABL = 0>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention5_kernel(const op_t* __restrict__ Q, const op_t* __restrict__ Kg,
                                                            const op_t* __restrict__ Vt, op_t* __restrict__ O, int S, int Sp,
                                                            int heads, int D, int FH, int QB) {
    __shared__ __attribute__((aligned(1024))) unsigned char tiles[A4_NB][TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..7
    const int late = (ABL & 32) ? 0 : (w >> 2);                      // waves 4..7 run one step behind
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int fh = (seq / QB) * 8 + xcd;
    const int qb = seq % QB;
    if (fh >= FH) return;
    const int frame = fh / heads, head = fh - frame * heads;
    const int q0 = qb * 512 + w * 64;
    const int lq = lane & 31, hi = lane >> 5;
    const op_t* Qb = Q + (size_t)fh * Sp * 64;
    const op_t* Kb = Kg + (size_t)fh * Sp * 64;
    const op_t* Vb = Vt + (size_t)fh * 64 * Sp;
    if (F16) fp16_saturate_mode();

    // DMA: wave w fills rows 8w .. 8w+7 of the K part and of the V^T part of a tile (one request each)
    const int lrow = w * 8 + (lane >> 3), lpc = (lane & 7) ^ ((lrow >> 1) & 7);
    const unsigned kvo = (unsigned)(lrow * 64 + lpc * 8) * 2u, vvo = (unsigned)(lrow * Sp + lpc * 8) * 2u;
    const unsigned lds_base = (unsigned)(size_t)&tiles[0][0];
    const int ntiles = (S + 63) / 64;
    const u4v srd_k = make_srd(Kb), srd_v = make_srd(Vb);
    const unsigned dst_k = __builtin_amdgcn_readfirstlane(lds_base + w * 1024);
    const unsigned dst_v = __builtin_amdgcn_readfirstlane(lds_base + 8192 + w * 1024);
    auto issue_k = [&](int t, auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        buffer_lds16<BUF * TILE_BYTES>(srd_k, (unsigned)min(t, ntiles - 1) * 8192u, kvo, dst_k);
    };
    auto issue_v = [&](int t, auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        buffer_lds16<BUF * TILE_BYTES>(srd_v, (unsigned)min(t, ntiles - 1) * 128u, vvo, dst_v);
    };
    op8 qf[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qrow = min(q0 + qt * 32 + lq, Sp - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[qt][ks] = *reinterpret_cast<const op8*>(Qb + (size_t)qrow * 64 + ks * 16 + hi * 8);
    }
    op8 kq[4][4];   // key rows of the reference estimate, requested with Q (see attention4)
    if (!(ABL & 8)) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const int kr0 = blk < 2 ? blk * 32 : q0 + (blk - 2) * 32;
            const op_t* kp = Kb + (size_t)min(kr0 + lq, Sp - 1) * 64 + hi * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kq[blk][ks] = *reinterpret_cast<const op8*>(kp + ks * 16);
        }
    }
    issue_k(0, std::integral_constant<int, 0>{}); issue_v(0, std::integral_constant<int, 0>{});
    issue_k(1, std::integral_constant<int, 1>{}); issue_v(1, std::integral_constant<int, 1>{});
    issue_k(2, std::integral_constant<int, 2>{}); issue_v(2, std::integral_constant<int, 2>{});
    static_assert(A4_AHEAD == 3 && A4_NB == 4, "prologue requests tiles 0..2 of a ring of four");

    f16v o[2][2];   // O^T accumulators (AGPRs through mfma_acc_agpr)
    f16v sc[2];     // ONE score set: [key block]
    u4v pf[4];      // ONE P^T set: [16-key group]
    op8 kf[8], vf[8];
    float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qt][db][r] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) pf[g][e] = 0u;

    const int krow = (lq & 19) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    unsigned ka[4], va[4];
    {
        const unsigned koff0 = krow * 128 + ((hi ^ ((krow >> 1) & 7)) << 4);
        const unsigned voff0 = 8192 + lq * 128 + ((hi ^ ((lq >> 1) & 7)) << 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ka[j] = lds_base + (koff0 ^ (j << 5));
            va[j] = lds_base + (voff0 ^ (j << 5));
        }
    }
    if (!(ABL & 8)) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float tm = -3e38f;
#pragma unroll
            for (int blk = 0; blk < 3; ++blk) {
                const int kr0 = blk < 2 ? blk * 32 : q0 + qt * 32;
                const int kb = blk < 2 ? blk : 2 + qt;
                f16v so = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) so = ATT2_MFMA(kq[kb][ks], qf[qt][ks], so, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kr0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    tm = fmaxf(tm, key < S ? so[r] : -3e38f);
                }
            }
            float a, b;
            halves(tm, a, b);
            m_run[qt] = fmaxf(a, b);
        }
    }
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) vf[g][e] = (op_t)0.f;   // M_a(0) multiplies them with P = 0
    vm_wait<0>();  // tiles 0 .. 2 have landed
    __syncthreads();

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
    typedef const __attribute__((address_space(3))) op8* lds_frag_ptr;
    auto ldk = [&](int buf, int f) { return *(lds_frag_ptr)(ka[f >> 1] + (unsigned)(buf * TILE_BYTES + (f & 1) * 4096)); };
    auto ldv = [&](int buf, int g) { return *(lds_frag_ptr)(va[g >> 1] + (unsigned)(buf * TILE_BYTES + (g & 1) * 4096)); };
#pragma clang diagnostic pop
#pragma unroll
    for (int f = 0; f < 8; ++f) kf[f] = ldk(0, f);   // K(0)

    // rare: see attention4's guard_tripped; here no score of the tile's query tile is in flight when it runs
    auto guard_tripped = [&](int qt, float lsum) {
        asm volatile("; guard tripped (attention5)" ::: "memory");
        agpr_settle();
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
        }
    };
    // matrix step: PV of query tile PQ with the P set and the V^T fragments in registers, scores of query tile 1 - PQ against the K
    // fragments in registers (accumulators initialised with -m: the reference costs no register set)
    auto m_step = [&](auto pq_tag) {
        constexpr int PQ = decltype(pq_tag)::value, SQ = 1 - PQ;
        const float nm = -m_run[SQ];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[b][r] = nm;
        if (ABL & 128) {   // (ablation: the operands stay live, the matrix pipe stays idle)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(kf[j]), "v"(vf[j]), "v"(pf[j >> 1]));
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j & 1] = ATT2_MFMA(kf[j], qf[SQ][j >> 1], sc[j & 1], 0, 0, 0);
            mfma_acc_agpr(o[PQ][j & 1], vf[j], __builtin_bit_cast(op8, pf[j >> 1]));
        }
    };
    // vector step: softmax of the score set -> P set, this lane's part of the row sum; fragment refills and one DMA request ride along
    auto v_step = [&](auto sq_tag, auto buf_tag, int t) -> float {
        constexpr int SQ = decltype(sq_tag)::value, VB = decltype(buf_tag)::value;
        constexpr int KB = (VB + 1) & (A4_NB - 1), DB = (VB + A4_AHEAD) & (A4_NB - 1);
        if (__builtin_amdgcn_readfirstlane(t == ntiles - 1 && (S & 63) != 0)) {
            asm volatile("; masked tail (last key tile only)" ::: "memory");   // (a volatile asm keeps this a BRANCH: if-converted, the 32 selects ran in every step)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + b * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                    if (key >= S) sc[b][r] = -1e30f;
                }
        }
        // SQ = 0 (V_a): V^T fragments <- V(t), request K(t+3);  SQ = 1 (V_b): K fragments <- K(t+1), request V(t+3)
        if (SQ == 0) issue_k(t + A4_AHEAD, std::integral_constant<int, DB>{});
        else issue_v(t + A4_AHEAD, std::integral_constant<int, DB>{});
        float lta = 0.f, ltb = 0.f;
        if (ABL & 64) asm volatile("" ::"v"(sc[0]), "v"(sc[1]));   // (ablation: the scores stay demanded, so their MFMAs stay)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int bj = c >> 2, e = 2 * (c & 3);
            if (!(ABL & 64)) {
                const float p0 = __builtin_amdgcn_exp2f(sc[bj >> 1][8 * (bj & 1) + e]);
                const float p1 = __builtin_amdgcn_exp2f(sc[bj >> 1][8 * (bj & 1) + e + 1]);
                lta += p0;
                ltb += p1;
                const op2 pk = {(op_t)p0, (op_t)p1};
                pf[bj][e >> 1] = __builtin_bit_cast(unsigned, pk);
            }
            if (c & 1) {   // eight fragment refills spread over the sixteen chunks
                const int f = c >> 1;
                if (SQ == 0) vf[f] = ldv(VB, f);
                else kf[f] = ldk(KB, f);
            }
        }
        const float lt = lta + ltb;
        l_run[SQ] += lt;
        return lt;
    };

    int pend0 = 0, pend1 = 0;   // wave-uniform
    float lsum0 = 0.f, lsum1 = 0.f;
    auto key_tile = [&](auto vb_tag, int t) {
        // M_a(t): PV(1, t-1), S(0, t)
        m_step(std::integral_constant<int, 1>{});
        __syncthreads();
        // V_a(t)
        if (pend1) { guard_tripped(1, lsum1); pend1 = 0; }   // PV(1, t-1) has been issued
        lsum0 = v_step(std::integral_constant<int, 0>{}, vb_tag, t);
        if (!(ABL & 8)) pend0 = __builtin_amdgcn_readfirstlane(__any(!(lsum0 < RESC_T)));
        vm_wait<3>();   // this wave's K(t+1), V(t+1) have landed; K(t+2), V(t+2), K(t+3) may stay in flight
        __syncthreads();
        // M_b(t): PV(0, t), S(1, t)
        m_step(std::integral_constant<int, 0>{});
        __syncthreads();
        // V_b(t)
        if (pend0) { guard_tripped(0, lsum0); pend0 = 0; }   // PV(0, t) has been issued
        lsum1 = v_step(std::integral_constant<int, 1>{}, vb_tag, t);
        if (!(ABL & 8)) pend1 = __builtin_amdgcn_readfirstlane(__any(!(lsum1 < RESC_T)));
        __syncthreads();
    };
    if (late) __syncthreads();   // waves 4..7: one step behind
    for (int t = 0; t < ntiles; t += A4_NB) {
        key_tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntiles) key_tile(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < ntiles) key_tile(std::integral_constant<int, 2>{}, t + 2);
        if (t + 3 < ntiles) key_tile(std::integral_constant<int, 3>{}, t + 3);
    }
    // PV(1, last)
#pragma unroll
    for (int g = 0; g < 8; ++g) mfma_acc_agpr(o[1][g & 1], vf[g], __builtin_bit_cast(op8, pf[g >> 1]));
    if (!late) __syncthreads();  // the early half's extra barrier: both halves have executed the same number
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
        safe_pass_impl(Qb, Kb, Vb, O + (size_t)frame * S * D + head * 64, q0, 64, S, Sp, D);   // inlined: see vit_attention2.h
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
                for (int pr = 0; pr < 2; ++pr) {
                    typedef unsigned u2v __attribute__((ext_vector_type(2)));
                    const op4 ve = {(op_t)(o[qt][db][8 * pr + 0] * inv), (op_t)(o[qt][db][8 * pr + 1] * inv),
                                    (op_t)(o[qt][db][8 * pr + 2] * inv), (op_t)(o[qt][db][8 * pr + 3] * inv)};
                    const op4 vo = {(op_t)(o[qt][db][8 * pr + 4] * inv), (op_t)(o[qt][db][8 * pr + 5] * inv),
                                    (op_t)(o[qt][db][8 * pr + 6] * inv), (op_t)(o[qt][db][8 * pr + 7] * inv)};
                    const u2v e = __builtin_bit_cast(u2v, ve), od = __builtin_bit_cast(u2v, vo);
                    u4v out;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(e[k], od[k], false, false);
                        const unsigned first = sw[0], second = sw[1];
                        out[k] = first;
                        out[2 + k] = second;
                    }
                    *reinterpret_cast<u4v*>(orow + db * 32 + 16 * pr + 8 * hi) = out;
                }
        }
    }
}

}  // namespace ATT2_NS
