// attention3.h -- round-3 experiment (NEGATIVE, kept for the micro-benchmark only; not part of libdtk.so).
// Include AFTER vit_attention2.h while its ATT2_* macros are still defined (it re-opens the same namespace and reuses the
// tile layout, the guards and the safe pass).
namespace ATT2_NS {
// ---------------------------------------------------------------------------------------------------------------------
// attention3_kernel (round 3): the same tiles, layouts, guards and safe pass, but SOFTWARE-PIPELINED ACROSS KEY TILES inside
// each wave.  Round 2's SQ counters say why the kernel above stops at 0.43 of the MFMA peak: per launch a SIMD spends 2.9 M
// cycles with the MFMA pipe busy and 3.5 M issuing VALU instructions -- and their sum (6.4 M) is about the kernel's duration
// (5.9 M): the two pipes hardly overlap, because a wave runs QK^T -> softmax -> PV strictly in sequence and the waves of a
// workgroup pass the same barrier every tile, so they sit in the same phase.  Here the scores of tile t+1 are computed WHILE
// tile t is exponentiated: one iteration issues 16 MFMAs (8 of S^T(t+1) = K(t+1) Q^T, 8 of O^T += V^T(t) P^T(t)) and the 16
// VALU chunks of softmax(t) (2 exponentials + 1 packed add + 1 packed convert each) alternately, one MFMA per chunk, pinned
// with sched_barrier: a chunk's ~30 VALU cycles run under the 32 cycles of the MFMA issued just before it, inside ONE wave.
// Costs a second set of score registers (~210 VGPRs -> 2 waves per SIMD: one 512-thread workgroup per CU, or two
// 256-thread ones) and a deeper LDS ring (K of tile t+1 and V^T of tile t are live at once; NB buffers, NB - 2 tiles in
// flight).  MODE 1 arithmetic only.  NW = waves per workgroup (8 or 4), 32 queries per wave.
// ---------------------------------------------------------------------------------------------------------------------
template <int NW, int NB, int ABL = 0>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void attention3_kernel(const op_t* __restrict__ Q,
                                                                              const op_t* __restrict__ Kg,
                                                                              const op_t* __restrict__ Vt,
                                                                              op_t* __restrict__ O, int S, int Sp, int heads,
                                                                              int D, int FH, int QB) {
    __shared__ __attribute__((aligned(1024))) unsigned char tiles[NB][TILE_BYTES];
    constexpr int RPW = 8 / NW;  // DMA requests per wave for the K part of a tile (and as many for the V^T part)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int fh = (seq / QB) * 8 + xcd;
    const int qb = seq % QB;
    if (fh >= FH) return;
    const int frame = fh / heads, head = fh - frame * heads;
    const int q0 = qb * (32 * NW) + w * 32;
    const int lq = lane & 31, hi = lane >> 5;
    const op_t* Qb = Q + (size_t)fh * Sp * 64;
    const op_t* Kb = Kg + (size_t)fh * Sp * 64;
    const op_t* Vb = Vt + (size_t)fh * 64 * Sp;
    if (F16) fp16_saturate_mode();

    const op_t* ksrc[RPW];
    const op_t* vsrc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int lrow = (w * RPW + r) * 8 + (lane >> 3), lpc = (lane & 7) ^ ((lrow >> 1) & 7);
        ksrc[r] = Kb + (size_t)lrow * 64 + lpc * 8;
        vsrc[r] = Vb + (size_t)lrow * Sp + lpc * 8;
    }
    const unsigned lds_base = (unsigned)(size_t)&tiles[0][0];
    const int ntiles = (S + 63) / 64;
    auto issue = [&](int t, int buf) {
        const int tt = min(t, ntiles - 1);  // past the end: a harmless repeat keeps the request count per tile uniform
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            glds16(ksrc[r] + (size_t)tt * 64 * 64,
                   __builtin_amdgcn_readfirstlane(lds_base + buf * TILE_BYTES + (w * RPW + r) * 1024));
            glds16(vsrc[r] + (size_t)tt * 64,
                   __builtin_amdgcn_readfirstlane(lds_base + buf * TILE_BYTES + 8192 + (w * RPW + r) * 1024));
        }
    };
    op8 qf[4];
    {
        const int qrow = min(q0 + lq, Sp - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const op8*>(Qb + (size_t)qrow * 64 + ks * 16 + hi * 8);
    }
#pragma unroll
    for (int i = 0; i < NB - 1; ++i) issue(i, i);

    f16v o[2];
    float m_run = 0.f;
    f2 l_run = {0.f, 0.f};
    bool has_m = false;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    const int krow = (lq & 19) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    int koff[2], voff[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int r = b * 32 + krow;
        koff[b] = r * 128 + ((hi ^ ((r >> 1) & 7)) << 4);
        const int d = b * 32 + lq;
        voff[b] = 8192 + d * 128 + ((hi ^ ((d >> 1) & 7)) << 4);
    }
    // reference estimate: keys 0..63 and the wave's own 32 keys (see MODE 1 above)
    if (!(ABL & 8)) {
        float tm = -3e38f;
#pragma unroll
        for (int blk = 0; blk < 3; ++blk) {
            const int kr0 = blk < 2 ? blk * 32 : q0;
            const op_t* kp = Kb + (size_t)min(kr0 + lq, Sp - 1) * 64 + hi * 8;
            op8 kf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const op8*>(kp + ks * 16);
            f16v so = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) so = ATT2_MFMA(kf[ks], qf[ks], so, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kr0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                tm = fmaxf(tm, key < S ? so[r] : -3e38f);
            }
        }
        float a, b;
        halves(tm, a, b);
        const float est = fmaxf(a, b);
        if (__builtin_amdgcn_readfirstlane(__any(!(fabsf(est) <= 3.f)))) {
            has_m = true;
            m_run = est;
        }
    }
    vm_wait<0>();  // the first NB - 1 tiles have landed
    __syncthreads();

    const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto mask_tail = [&](f16v (&sc)[2], int t) {  // keys beyond S (last tile only)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 64 + b * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                if (key >= S) sc[b][r] = -1e30f;
            }
    };
    f16v scA[2], scB[2];
    {   // prologue: scores of tile 0
        const unsigned char* tk = &tiles[0][0];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const op8 kr = *reinterpret_cast<const op8*>(tk + (koff[f & 1] ^ ((f >> 1) << 5)));
            scA[f & 1] = ATT2_MFMA(kr, qf[f >> 1], f < 2 ? zero16 : scA[f & 1], 0, 0, 0);
        }
        if (ntiles == 1 && (S & 63) != 0) mask_tail(scA, 0);
    }
    int vbuf = 0;  // buffer of tile t (V^T part read here); K of tile t+1 in the next one; tile t+NB-1 goes to the previous one
    // one iteration: softmax + PV of tile t (scores `cur`), scores of tile t+1 into `nxt`
    auto body = [&](f16v (&cur)[2], f16v (&nxt)[2], int t, auto sub_tag) {
        constexpr bool SUB = decltype(sub_tag)::value;
        const int kbuf = vbuf == NB - 1 ? 0 : vbuf + 1;
        issue(t + NB - 1, vbuf == 0 ? NB - 1 : vbuf - 1);
        const unsigned char* tk = &tiles[kbuf][0];
        const unsigned char* tv = &tiles[vbuf][0];
        auto ldk = [&](int f) { return *reinterpret_cast<const op8*>(tk + (koff[f & 1] ^ ((f >> 1) << 5))); };
        auto ldv = [&](int g) { return *reinterpret_cast<const op8*>(tv + (voff[g & 1] ^ ((g >> 1) << 5))); };
        op8 kr[4], vr[4], pf[4];
        f2 lt = {0.f, 0.f};
        const f2 nm = {-m_run, -m_run};
        // VALU chunk c = values 2c, 2c+1 of the 32 per lane: group bj = c >> 2, element pair e = 2 (c & 3)
        auto chunk = [&](int c) {
            const int bj = c >> 2, e = 2 * (c & 3);
            f2 sv = {cur[bj >> 1][8 * (bj & 1) + e], cur[bj >> 1][8 * (bj & 1) + e + 1]};
            if (SUB) sv += nm;
            const f2 p = (ABL & 1) ? sv * f2{0.01f, 0.01f} : f2{__builtin_amdgcn_exp2f(sv[0]), __builtin_amdgcn_exp2f(sv[1])};
            lt += p;
            pf[bj][e] = (op_t)p[0];
            pf[bj][e + 1] = (op_t)p[1];
        };
        // MFMA of slot i (issued in front of VALU chunk i):   Q0 Q1 Q2 Q3 | P0 Q4 P1 Q5 | P2 Q6 P3 Q7 | P4 P5 - - | tail P6 P7
        // (Qf = K fragment f of the NEXT tile; Pg = V^T fragment g of THIS tile, g = 2 bj + d-block: needs P group bj = chunks
        // 4 bj .. 4 bj + 3, all issued before its slot)
        kr[0] = ldk(0); kr[1] = ldk(1); kr[2] = ldk(2);
        vr[0] = ldv(0); vr[1] = ldv(1);
        auto qk = [&](int f) {
            if (f + 3 < 8) kr[(f + 3) & 3] = ldk(f + 3);
            nxt[f & 1] = ATT2_MFMA(kr[f & 3], qf[f >> 1], f < 2 ? zero16 : nxt[f & 1], 0, 0, 0);
        };
        auto pv = [&](int g) {
            if (g + 2 < 8) vr[(g + 2) & 3] = ldv(g + 2);
            o[g & 1] = ATT2_MFMA(vr[g & 3], pf[g >> 1], o[g & 1], 0, 0, 0);
        };
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i < 4) qk(i);
            else if (i < 12) { if (i & 1) qk(4 + ((i - 5) >> 1)); else pv((i - 4) >> 1); }
            else if (i < 14) pv(i - 8);
            chunk(i);
            __builtin_amdgcn_sched_barrier(0);
        }
        pv(6);
        __builtin_amdgcn_sched_barrier(0);
        pv(7);
        __builtin_amdgcn_sched_barrier(0);
        const float lsum = lt[0] + lt[1];
        l_run += lt;
        const bool resc = !(ABL & 8) && __builtin_amdgcn_readfirstlane(__any(!(lsum < RESC_T)));
        if (t + 1 == ntiles - 1 && (S & 63) != 0) mask_tail(nxt, t + 1);
        if (resc) {
            asm volatile("; guard tripped" ::: "memory");
            float a, b;
            halves(lsum, a, b);
            const float tot = a + b;
            if (!(a < POISON_T && b < POISON_T)) {
                l_run = f2{__builtin_nanf(""), __builtin_nanf("")};
            } else if (tot >= RESC_T) {
                const float k = floorf(__builtin_amdgcn_logf(tot));
                const float alpha = __builtin_amdgcn_exp2f(-k);
                // the scores of the NEXT tile are raw (the reference is subtracted when they are exponentiated): only O, l move
                m_run += k;
                l_run *= f2{alpha, alpha};
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
                has_m = true;
            }
        }
        vm_wait<2 * RPW * (NB - 3)>();  // tile t+2 has landed (K read by the next iteration); t+3 .. stay in flight
        if (!(ABL & 16)) __syncthreads();
        vbuf = kbuf;
    };
    // two bodies (with / without the subtraction of the reference) behind a wave-uniform branch; the trip count does not depend
    // on has_m (so that t and the buffer indices stay scalar), two tiles per trip so that the score sets swap roles by name
    int t = 0;
    for (; t + 1 < ntiles; t += 2) {
        if (has_m) body(scA, scB, t, std::true_type{}); else body(scA, scB, t, std::false_type{});
        if (has_m) body(scB, scA, t + 1, std::true_type{}); else body(scB, scA, t + 1, std::false_type{});
    }
    if (t < ntiles) {
        if (has_m) body(scA, scB, t, std::true_type{}); else body(scA, scB, t, std::false_type{});
    }
    vm_wait<0>();
    const float l_half = l_run[0] + l_run[1];
    float la, lb;
    halves(l_half, la, lb);
    const float l_tot = la + lb;
    if (!(ABL & 8) && __any(!(l_tot > LOW_T && l_tot < 0x1p120f))) {
        safe_pass<3>(Qb, Kb, Vb, O + (size_t)frame * S * D + head * 64, q0, 32, S, Sp, D);
        return;
    }
    const float inv = 1.f / l_tot;
    const int qi = q0 + lq;
    if (qi < S) {
        op_t* orow = O + ((size_t)frame * S + qi) * D + head * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int d = db * 32 + 8 * rq + 4 * hi;
                op4 v = {(op_t)(o[db][4 * rq + 0] * inv), (op_t)(o[db][4 * rq + 1] * inv),
                         (op_t)(o[db][4 * rq + 2] * inv), (op_t)(o[db][4 * rq + 3] * inv)};
                *reinterpret_cast<op4*>(orow + d) = v;
            }
    }
}

}  // namespace ATT2_NS
