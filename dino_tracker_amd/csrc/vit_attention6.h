// vit_attention6.h -- flash attention for d_head = 64, fourth generation (round 6): ONE wave per SIMD, 128 queries per wave.
//
// Include AFTER vit_attention4.h while the ATT2_* macros are still defined: same per-type namespace, tile layout (K tile + V^T tile of 64
// keys, 16-byte pieces XOR-swizzled through the DMA's source address), LDS-DMA helpers, fp16 range guards and output layout.
//
// Why.  attention4's wave owns two 32-query tiles: a K fragment and a V^T fragment read from LDS serve two tiles --
// 16 ds_read_b128 and 4 LDS-DMA requests per 32 MFMAs, one barrier per 32 MFMAs; its own ablations price the reads at 5.5 % and the
// requests at 6 % of the kernel, and its prologue (Q, the reference estimate, the first tiles) at 4 %.  Here a wave owns FOUR tiles: the
// same reads, requests and barrier serve 64 MFMAs, a workgroup's prologue 512 queries.  scripts/ubench/attention6.h measured the
// schedule without guards first (30 frames, same box): attention4 3.25 ms (3.03 without its guards), two tiles on this schedule 2.96,
// three 2.84, four 2.73 ms.  What pays for it is registers, and all of them are spoken for:
//     AGPRs 256:  O accumulators 4 x 2 x 16 = 128 (asm MFMA, as in attention4), Q fragments 4 x 4 x 4 = 64 (loaded there by asm),
//                 K and V^T fragments 8 x 4 + 8 x 4 = 64 (ds_read_b128 straight into AGPRs, by asm)
//     VGPRs:      two score sets 64, P fragments 4 x 16 = 64, the references' C operands 4 x 16 = 64, addresses and sums
// so every matrix instruction is an asm statement with its operands' register files spelled out ("a" / "v").
//
// Schedule of key tile t (one barrier at its start), sub-steps u = 0 .. 3, 16 slots each; a slot = one MFMA + one softmax chunk:
//     MFMA   slot 2j:      S(u, t)  (+)= K(t) fragment j x Q(u)           (8; the first of each key block takes C = -m(u))
//            slot 2j + 1:  O(u)      += V(t-1) fragment j x P(u, t-1)      (8)
//     VALU   chunk i of tile u - 1's scores of THIS key tile (u = 0: tile 3's scores of key tile t - 1): two exponentials, the adds and
//            the packed convert of the chunk before; MFMA first, then the adds (volatile asm keeps that order: left to the scheduler
//            the asm MFMA of the odd slots sank below its chunk and MFMAs issued in pairs), the exponentials anywhere in the slot
//     LDS    sub-step 3 only: every fragment register is refilled right after its last use (kf[j] <- K(t+1) after slot 2j, vf[j] <-
//            V(t) after slot 2j + 1), 16 slots ahead of its next use; the asm reads are invisible to the compiler's lgkmcnt
//            bookkeeping: counted waits (lgkmcnt(8) in front of the barrier, lgkmcnt(0) in slot 7 of sub-step 0)
//     DMA    sub-step 0, slots 2, 6, 10, 14: the four requests of tile t + 3 (ring of four buffers, as attention4)
// Score sets alternate between two register sets with every sub-step (four tiles: tile u uses set u & 1).
// Arithmetic is attention4's (attention2 MODE 1): optimistic exponentials against a reference estimated once per query, a guard on
// every tile's row sum, the rescale AFTER the PV product of the tile that tripped it (here: after sub-step u of the next key tile, where
// PV(u, t) and S(u, t + 1) are issued -- the scores of key tile t + 1 move with the reference), poison and the safe pass.
#ifndef ATT2_NS
#error "include vit_attention2.h / vit_attention4.h first and keep ATT2_NS, ATT2_T, ATT2_F16, ATT2_MFMA defined"
#endif

namespace ATT2_NS {

constexpr int A6_NQ = 4;

inline unsigned attention6_grid(int FH, int S, int* qb_out) {
    const int QB = (S + 128 * A6_NQ - 1) / (128 * A6_NQ);
    *qb_out = QB;
    return (unsigned)(((FH + 7) / 8) * 8 * QB);
}

// S^T accumulate: A = K fragment (AGPR), B = Q fragment (AGPR), accumulator in VGPRs; the first product of a key block takes the
// reference as its C operand (-m in all 16 registers of a lane: every accumulator register of a lane belongs to ONE query)
__device__ __forceinline__ void a6_s_first(f16v& acc, const op8& a, const op8& b, const f16v& c) {
    if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc) : "a"(a), "a"(b), "v"(c));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "a"(a), "a"(b), "v"(c));
}
__device__ __forceinline__ void a6_s_next(f16v& acc, const op8& a, const op8& b) {
    if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "a"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "a"(b));
}
// O^T accumulate: A = V^T fragment (AGPR), B = P fragment (VGPR), accumulator in AGPRs
__device__ __forceinline__ void a6_pv(f16v& c, const op8& a, const op8& b) {
    if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "a"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "a"(a), "v"(b));
}
// a fragment from LDS straight into AGPRs (base address in a VGPR + immediate)
template <int OFF>
__device__ __forceinline__ void a6_lds_frag(op8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(addr), "i"(OFF) : "memory");
}

template <int ABL = 0>
__global__ __launch_bounds__(256, 1) void attention6_kernel(const op_t* __restrict__ Q, const op_t* __restrict__ Kg,
                                                            const op_t* __restrict__ Vt, op_t* __restrict__ O, int S, int Sp,
                                                            int heads, int D, int FH, int QB) {
    constexpr int NQ = A6_NQ;
    __shared__ __attribute__((aligned(1024))) unsigned char tiles[A4_NB][TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int fh = (seq / QB) * 8 + xcd;
    const int qb = seq % QB;
    if (fh >= FH) return;
    const int frame = fh / heads, head = fh - frame * heads;
    const int q0 = qb * (128 * NQ) + w * (32 * NQ);
    const int lq = lane & 31, hi = lane >> 5;
    const op_t* Qb = Q + (size_t)fh * Sp * 64;
    const op_t* Kb = Kg + (size_t)fh * Sp * 64;
    const op_t* Vb = Vt + (size_t)fh * 64 * Sp;
    if (F16) fp16_saturate_mode();

    // Q^T fragments (B operand) straight into AGPRs: lane (query lq, hi) holds d = 16 ks + 8 hi .. + 7 -- and, in the same trip to
    // memory, the key rows of the reference estimate: keys 0..31, 32..63 and every tile's own 32 keys
    op8 qf[NQ][4];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
        const int qrow = min(q0 + qt * 32 + lq, Sp - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(qf[qt][ks]) : "v"(Qb + (size_t)qrow * 64 + ks * 16 + hi * 8) : "memory");
    }
    op8 kq[2 + NQ][4];
    if (!(ABL & 8)) {
#pragma unroll
        for (int blk = 0; blk < 2 + NQ; ++blk) {
            const int kr0 = blk < 2 ? blk * 32 : q0 + (blk - 2) * 32;
            const op_t* kp = Kb + (size_t)min(kr0 + lq, Sp - 1) * 64 + hi * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kq[blk][ks] = *reinterpret_cast<const op8*>(kp + ks * 16);
        }
    }
    unsigned kvo[2], vvo[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int lrow = (w * 2 + r) * 8 + (lane >> 3), lpc = (lane & 7) ^ ((lrow >> 1) & 7);
        kvo[r] = (unsigned)(lrow * 64 + lpc * 8) * 2u;
        vvo[r] = (unsigned)(lrow * Sp + lpc * 8) * 2u;
    }
    const unsigned lds_base = (unsigned)(size_t)&tiles[0][0];
    const int ntiles = (S + 63) / 64;
    const u4v srd_k = make_srd(Kb), srd_v = make_srd(Vb);
    unsigned dma_dst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_dst[i] = __builtin_amdgcn_readfirstlane(lds_base + (i >> 1) * 8192 + (w * 2 + (i & 1)) * 1024);
    auto issue_one = [&](int t, auto buf_tag, int i) {
        constexpr int BUF = decltype(buf_tag)::value;
        const unsigned tt = (unsigned)min(t, ntiles - 1);  // past the end: a harmless repeat keeps the request count per tile uniform
        if (i < 2) buffer_lds16<BUF * TILE_BYTES>(srd_k, tt * 8192u, kvo[i & 1], dma_dst[i]);
        else buffer_lds16<BUF * TILE_BYTES>(srd_v, tt * 128u, vvo[i & 1], dma_dst[i]);
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        issue_one(0, std::integral_constant<int, 0>{}, j);
        issue_one(1, std::integral_constant<int, 1>{}, j);
        issue_one(2, std::integral_constant<int, 2>{}, j);
    }
    static_assert(A4_AHEAD == 3 && A4_NB == 4, "prologue requests tiles 0..2 into a ring of four");
    // ONE trip to memory for the Q fragments, the estimate's key rows and the first three tiles (the asm loads of Q are invisible to the
    // compiler's vmcnt bookkeeping: wait, then re-define every loaded register behind the wait, as gemm_ws does for its weights)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+a"(qf[qt][ks]));
    __syncthreads();   // tiles 0 .. 2 of every wave have landed

    // reference estimate (attention2 MODE 1): keys 0..63 and the query tile's own 32 keys, while the first tiles are in flight
    float m_run[NQ], l_run[NQ];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) { m_run[qt] = 0.f; l_run[qt] = 0.f; }
    if (!(ABL & 8)) {
#pragma unroll
        for (int qt = 0; qt < NQ; ++qt) {
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
    f16v negm[NQ];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[qt][r] = -m_run[qt];

    f16v o[NQ][2];   // O^T accumulators (AGPRs): rows d = 32 db + (r & 3) + 8 (r >> 2) + 4 hi, column = query lq
    f16v sc[2][2];   // two score sets [set][key block]: register r of lane-half hi = key 32 b + 16 (r >> 3) + 8 hi + (r & 7)
    u4v pf[NQ][4];   // P^T fragments [query tile][16-key group]
    op8 kf[8], vf[8];  // K / V^T fragments (AGPRs)
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qt][db][r] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) pf[qt][g][e] = 0u;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[s][b][r] = -1e30f;   // (tile 3's first softmax runs on these: P = 0)
#pragma unroll
    for (int g = 0; g < 8; ++g) {
#pragma unroll
        for (int e = 0; e < 8; ++e) vf[g][e] = (op_t)0.f;   // key tile 0 multiplies them with P = 0
        asm volatile("" : "+a"(vf[g]));
    }
    // fragment addresses inside a ring buffer: K fragment f = (key block f & 1, d step f >> 1) at ka[f >> 1] + 4096 (f & 1), V^T fragment
    // g = (d block g & 1, 16-key group g >> 1) at va[g >> 1] + 4096 (g & 1); + 16384 * buffer: all immediates
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
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        if (f & 1) a6_lds_frag<4096>(kf[f], ka[f >> 1]);
        else a6_lds_frag<0>(kf[f], ka[f >> 1]);
    }
    // (asm reads: nobody but this statement waits for them -- the counted wait that opens a key tile assumes the SIXTEEN reads of a
    //  preceding last sub-step and would let key tile 0 start on fragments that have not landed)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    auto mask_tail = [&](f16v (&s2)[2], int t) {  // keys beyond S (last tile only)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 64 + b * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                if (key >= S) s2[b][r] = -1e30f;
            }
    };
    // rare: a lane's 32-key part of a tile's row sum of query tile qt passed RESC_T (attention2 MODE 1, same arithmetic); called after
    // the PV product of that key tile has been issued.  The scores of the NEXT key tile, computed against the old reference, are in
    // score set `set`: they move with it.
    auto guard_tripped = [&](auto qt_tag, auto set_tag, float lsum, bool scores_live) {
        constexpr int qt = decltype(qt_tag)::value, set = decltype(set_tag)::value;
        asm volatile("; guard tripped" ::: "memory");
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
            if (scores_live) {
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[set][b2][r] -= k;
            }
            l_run[qt] *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qt][db][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[qt][r] = -m_run[qt];
        }
    };

    const int tail_tile = (S & 63) != 0 ? ntiles - 1 : -1;
    int pend[NQ];        // wave-uniform: tile q's guard tripped in its last softmax; handled after its next PV product
    float lsum[NQ];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) { pend[qt] = 0; lsum[qt] = 0.f; }

    // sub-step U of the key tile in ring buffer VB: MFMAs of tile U into score set U & 1, softmax of tile (U - 1) mod 4 out of the other set
    auto sub_step = [&](auto u_tag, auto vb_tag, int t) {
        constexpr int U = decltype(u_tag)::value, VB = decltype(vb_tag)::value;
        constexpr int SET = U & 1, SQ = (U + NQ - 1) % NQ;
        constexpr int KB = (VB + 1) & (A4_NB - 1), DB = (VB + A4_AHEAD) & (A4_NB - 1);
        constexpr bool REFILL = U == NQ - 1, DMA = U == 0;
        float lta = 0.f, ltb = 0.f, q0e = 0.f, q1e = 0.f;   // two row-sum chains; the exponentials of the previous slot's chunk
        auto chunk_exp = [&](int c, float& p0, float& p1) {
            const int bj = c >> 2, e = 2 * (c & 3);
            const float s0 = sc[1 - SET][bj >> 1][8 * (bj & 1) + e], s1 = sc[1 - SET][bj >> 1][8 * (bj & 1) + e + 1];
            p0 = (ABL & 1) ? s0 * 0.01f : __builtin_amdgcn_exp2f(s0);
            p1 = (ABL & 1) ? s1 * 0.01f : __builtin_amdgcn_exp2f(s1);
        };
        auto chunk_fin = [&](int c, float p0, float p1) {
            const int bj = c >> 2, e = 2 * (c & 3);
            if (c == 0) {   // the chains start from the first chunk's values (no zero-initialised registers, no adds)
                lta = p0;
                ltb = p1;
            } else {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(lta) : "v"(p0));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(ltb) : "v"(p1));
            }
            unsigned wv;
            if constexpr (F16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(wv) : "v"(p0), "v"(p1));
            else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(wv) : "v"(p0), "v"(p1));
            pf[SQ][bj][e >> 1] = wv;
        };
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = i >> 1;
            if ((i & 1) == 0) {
                if (j < 2) a6_s_first(sc[SET][j & 1], kf[j], qf[U][j >> 1], negm[U]);
                else a6_s_next(sc[SET][j & 1], kf[j], qf[U][j >> 1]);
            } else {
                a6_pv(o[U][j & 1], vf[j], __builtin_bit_cast(op8, pf[U][j >> 1]));
            }
            if (!(ABL & 128) && REFILL) {   // the register's last use was the MFMA just issued
                if (i & 1) {
                    if (j & 1) a6_lds_frag<VB * TILE_BYTES + 4096>(vf[j], va[j >> 1]);
                    else a6_lds_frag<VB * TILE_BYTES>(vf[j], va[j >> 1]);
                } else {
                    if (j & 1) a6_lds_frag<KB * TILE_BYTES + 4096>(kf[j], ka[j >> 1]);
                    else a6_lds_frag<KB * TILE_BYTES>(kf[j], ka[j >> 1]);
                }
            }
            if (DMA && (i & 3) == 2 && !((ABL & 2) && t > 0)) issue_one(t + A4_AHEAD, std::integral_constant<int, DB>{}, i >> 2);
            // the second half of the previous key tile's fragment reads (kf[4..7], vf[4..7]: first used in slots 8 .. 15 of this sub-step)
            if (U == 0 && i == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float p0, p1;
            chunk_exp(i, p0, p1);
            if (i > 0) chunk_fin(i - 1, q0e, q1e);
            asm volatile("" : "+v"(p0), "+v"(p1));   // the exponentials stay in this slot
            q0e = p0;
            q1e = p1;
            __builtin_amdgcn_sched_barrier(0);
        }
        chunk_fin(15, q0e, q1e);
        const float lt = lta + ltb;
        l_run[SQ] += lt;
        if (t == tail_tile) {   // (one scalar compare: the key tile with keys beyond S, or -1)
            asm volatile("s_nop 15" ::: "memory");   // the scores were written by asm MFMAs two slots ago: the compiler pads nothing for them
            mask_tail(sc[SET], t);
        }
        if (ABL & 8) return;
        // tile U: its PV product of the previous key tile and its scores of this one are issued -- a pending rescale can run now
        if (pend[U]) {
            guard_tripped(u_tag, std::integral_constant<int, SET>{}, lsum[U], true);
            pend[U] = 0;
        }
        // tile SQ: the guard on the row sums just formed (handled after ITS next PV product, i.e. after sub-step SQ of the next key tile)
        lsum[SQ] = lt;
        pend[SQ] = __builtin_amdgcn_ballot_w64(!(lt < RESC_T)) != 0ull;   // (a compare into VCC and one scalar test)
    };
    auto key_tile = [&](auto vb_tag, int t) {
        vm_wait<4>();   // this wave's requests of tile t + 1 have landed (those of tile t + 2 stay in flight)
        // The fragment reads of the previous key tile's last sub-step are asm statements: not the compiler's to wait for.  They were
        // issued one per slot in the order kf[0], vf[0], kf[1], ... and LDS operations return in order: the first eight (kf[0..3],
        // vf[0..3], issued 9 .. 16 slots ago) cover slots 0 .. 7 of sub-step 0; the other eight are waited for in its slot 7.
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        if (!(ABL & 16)) __syncthreads();   // every wave is past key tile t - 1: buffer (t - 1) % 4 is free, tile t + 1 is visible
        sub_step(std::integral_constant<int, 0>{}, vb_tag, t);
        sub_step(std::integral_constant<int, 1>{}, vb_tag, t);
        sub_step(std::integral_constant<int, 2>{}, vb_tag, t);
        sub_step(std::integral_constant<int, 3>{}, vb_tag, t);
    };
    static_assert(A6_NQ == 4, "score sets alternate by U & 1: an even number of tiles");
    for (int t = 0; t < ntiles; t += A4_NB) {
        key_tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntiles) key_tile(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < ntiles) key_tile(std::integral_constant<int, 2>{}, t + 2);
        if (t + 3 < ntiles) key_tile(std::integral_constant<int, 3>{}, t + 3);
    }
    // epilogue: the softmax of tile 3 over the last key tile (its scores are in set 1), then PV(u, last) for every tile and the guards
    // that are still pending
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
        float lta = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int bj = c >> 2, e = 2 * (c & 3);
            const float p0 = __builtin_amdgcn_exp2f(sc[1][bj >> 1][8 * (bj & 1) + e]), p1 = __builtin_amdgcn_exp2f(sc[1][bj >> 1][8 * (bj & 1) + e + 1]);
            lta += p0 + p1;
            const op2 pk = {(op_t)p0, (op_t)p1};
            pf[NQ - 1][bj][e >> 1] = __builtin_bit_cast(unsigned, pk);
        }
        l_run[NQ - 1] += lta;
        lsum[NQ - 1] = lta;
        pend[NQ - 1] = (ABL & 8) ? 0 : (__builtin_amdgcn_ballot_w64(!(lta < RESC_T)) != 0ull);
        // (the converts above are compiler-scheduled VALU writes of registers an asm MFMA reads next: keep them apart)
        asm volatile("s_nop 4" ::: "memory");
#pragma unroll
        for (int g = 0; g < 8; ++g) a6_pv(o[0][g & 1], vf[g], __builtin_bit_cast(op8, pf[0][g >> 1]));
#pragma unroll
        for (int g = 0; g < 8; ++g) a6_pv(o[1][g & 1], vf[g], __builtin_bit_cast(op8, pf[1][g >> 1]));
#pragma unroll
        for (int g = 0; g < 8; ++g) a6_pv(o[2][g & 1], vf[g], __builtin_bit_cast(op8, pf[2][g >> 1]));
#pragma unroll
        for (int g = 0; g < 8; ++g) a6_pv(o[3][g & 1], vf[g], __builtin_bit_cast(op8, pf[3][g >> 1]));
    }
    if (pend[0]) guard_tripped(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, lsum[0], false);
    if (pend[1]) guard_tripped(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, lsum[1], false);
    if (pend[2]) guard_tripped(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, lsum[2], false);
    if (pend[3]) guard_tripped(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{}, lsum[3], false);
    agpr_settle();
    vm_wait<0>();
    float l_tot[NQ];
    bool redo = false;
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
        float a, b;
        halves(l_run[qt], a, b);
        l_tot[qt] = a + b;
        redo |= __any(!(l_tot[qt] > LOW_T && l_tot[qt] < 0x1p120f));
    }
    if (redo && !(ABL & 8)) {
        safe_pass<6>(Qb, Kb, Vb, O + (size_t)frame * S * D + head * 64, q0, 32 * NQ, S, Sp, D);
        return;
    }
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
        const float inv = 1.f / l_tot[qt];
        const int qi = q0 + qt * 32 + lq;
        if (qi < S) {
            op_t* orow = O + ((size_t)frame * S + qi) * D + head * 64;
            // (attention4's store: the two halves of a query trade 8-byte pieces by v_permlane32_swap, 16-byte stores)
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
                    const int d = db * 32 + 16 * pr + 8 * hi;
                    *reinterpret_cast<u4v*>(orow + d) = out;
                }
        }
    }
}

}  // namespace ATT2_NS
