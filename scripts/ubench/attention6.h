// attention6.h -- EXPERIMENT (round 6): attention4's structure with NQ query tiles per wave instead of two.
//
// Include after vit_attention4.h while the ATT2_* macros are defined (same per-type namespace, tile layout and DMA helpers).
//
// attention4: one wave per SIMD, 64 queries per wave; a K fragment and a V^T fragment are read from LDS once per key tile and serve two
// 32-query tiles: 16 ds_read_b128 and 4 LDS-DMA requests per 32 MFMAs.  Its own ablations price the LDS reads at 5.5 % and the DMA at 6 %
// of the kernel.  Here a wave owns NQ tiles (NQ = 3: 96 queries, NQ = 4: 128): the same 16 reads and 4 requests serve 16 NQ MFMAs, the
// barrier comes once per 16 NQ MFMAs, the prologue once per 128 NQ queries of a workgroup.  What pays for it is registers: O accumulators
// 32 NQ (AGPRs, asm MFMA as in attention4), Q fragments 16 NQ (AGPRs: loaded there by asm, the builtin MFMA takes them as they are, like
// gemm_ws's weights), P fragments 16 NQ, two score sets (64).
//
// Schedule of key tile t (one barrier at its start), sub-steps u = 0 .. NQ - 1, 16 slots each:
//     MFMA   slot 2j:      S(u, t)     += K(t) fragment j x Q(u)            (8 MFMAs, first of each key block on a zero C operand)
//            slot 2j + 1:  O(u)        += V(t-1) fragment j x P(u, t-1)      (8 MFMAs)
//     VALU   the 16 softmax chunks of tile u - 1's scores of THIS key tile (u = 0: tile NQ - 1's scores of key tile t - 1)
//     LDS    sub-step NQ - 1 only: every fragment register is refilled right after its last use (kf[j] <- K(t+1) after slot 2j,
//            vf[j] <- V(t) after slot 2j + 1), i.e. 16 slots ahead of its next use in sub-step 0 of key tile t + 1
//     DMA    sub-step 0, slots 2, 6, 10, 14: the four requests of tile t + 3
// Score sets alternate between two register sets by the parity of the global sub-step count.
//
// THIS FILE MEASURES THE SCHEDULE ONLY: no reference estimate, no guards, no safe pass (scores enter the exponential as they are;
// fine for the benchmark's random operands, wrong for the harness's stress rows).  If it pays, the guards of attention4 move in.
#ifndef ATT2_NS
#error "include vit_attention2.h / vit_attention4.h first and keep ATT2_NS, ATT2_T, ATT2_F16, ATT2_MFMA defined"
#endif

namespace ATT2_NS {

template <int NQ>
inline unsigned attention6_grid(int FH, int S, int* qb_out) {
    const int QB = (S + 128 * NQ - 1) / (128 * NQ);
    *qb_out = QB;
    return (unsigned)(((FH + 7) / 8) * 8 * QB);
}

template <int NQ, int ABL = 0>
__global__ __launch_bounds__(256, 1) void attention6_kernel(const op_t* __restrict__ Q, const op_t* __restrict__ Kg,
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
    const int q0 = qb * (128 * NQ) + w * (32 * NQ);
    const int lq = lane & 31, hi = lane >> 5;
    const op_t* Qb = Q + (size_t)fh * Sp * 64;
    const op_t* Kb = Kg + (size_t)fh * Sp * 64;
    const op_t* Vb = Vt + (size_t)fh * 64 * Sp;
    if (F16) fp16_saturate_mode();

    // Q^T fragments (B operand) straight into AGPRs: lane (query lq, hi) holds d = 16 ks + 8 hi .. + 7
    op8 qf[NQ][4];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
        const int qrow = min(q0 + qt * 32 + lq, Sp - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(qf[qt][ks]) : "v"(Qb + (size_t)qrow * 64 + ks * 16 + hi * 8) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+a"(qf[qt][ks]));

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
        const unsigned tt = (unsigned)min(t, ntiles - 1);
        if (i < 2) buffer_lds16<BUF * TILE_BYTES>(srd_k, tt * 8192u, kvo[i & 1], dma_dst[i]);
        else buffer_lds16<BUF * TILE_BYTES>(srd_v, tt * 128u, vvo[i & 1], dma_dst[i]);
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        issue_one(0, std::integral_constant<int, 0>{}, j);
        issue_one(1, std::integral_constant<int, 1>{}, j);
        issue_one(2, std::integral_constant<int, 2>{}, j);
    }

    f16v o[NQ][2];   // O^T accumulators (AGPRs)
    f16v sc[2][2];   // two score sets, [set][key block]
    u4v pf[NQ][4];   // P^T fragments [query tile][16-key group]
    op8 kf[8], vf[8];
    float l_run[NQ];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
        l_run[qt] = 0.f;
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
            for (int r = 0; r < 16; ++r) sc[s][b][r] = -1e30f;   // (the first softmax of tile NQ - 1 runs on these: P = 0)
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
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) vf[g][e] = (op_t)0.f;
    vm_wait<0>();
    __syncthreads();

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
    typedef const __attribute__((address_space(3))) op8* lds_frag_ptr;
    auto ldk = [&](int buf, int f) { return *(lds_frag_ptr)(ka[f >> 1] + (unsigned)(buf * TILE_BYTES + (f & 1) * 4096)); };
    auto ldv = [&](int buf, int g) { return *(lds_frag_ptr)(va[g >> 1] + (unsigned)(buf * TILE_BYTES + (g & 1) * 4096)); };
#pragma clang diagnostic pop
    auto mask_tail = [&](f16v (&s2)[2], int t) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 64 + b * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                if (key >= S) s2[b][r] = -1e30f;
            }
    };
#pragma unroll
    for (int f = 0; f < 8; ++f) kf[f] = ldk(0, f);

    const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // sub-step U of the key tile in ring buffer VB: MFMAs of tile U into score set SET, softmax of tile (U - 1) mod NQ out of set 1 - SET
    auto sub_step = [&](auto u_tag, auto vb_tag, auto set_tag, int t) {
        constexpr int U = decltype(u_tag)::value, VB = decltype(vb_tag)::value, SET = decltype(set_tag)::value;
        constexpr int SQ = (U + NQ - 1) % NQ;
        constexpr int KB = (VB + 1) & (A4_NB - 1), DB = (VB + A4_AHEAD) & (A4_NB - 1);
        constexpr bool REFILL = U == NQ - 1, DMA = U == 0;
        float lta = 0.f, ltb = 0.f, q0e = 0.f, q1e = 0.f;
        // Every instruction of a slot is a volatile asm statement, in the order written: MFMA first, then the two exponentials of chunk i,
        // then the adds and the packed convert of chunk i - 1.  (First form of this file: builtins + anchors, as attention4 -- the
        // scheduler sank the asm PV MFMA of the odd slots below their chunk, MFMAs issued in back-to-back pairs with ten VALU
        // instructions between the pairs.)
        auto chunk_exp = [&](int c, float& p0, float& p1) {
            const int bj = c >> 2, e = 2 * (c & 3);
            if (ABL & 1) {
                p0 = sc[1 - SET][bj >> 1][8 * (bj & 1) + e] * 0.01f;
                p1 = sc[1 - SET][bj >> 1][8 * (bj & 1) + e + 1] * 0.01f;
                return;
            }
            // (builtins, not asm: behind an opaque asm the compiler pads a wait state for the transcendental-result hazard; where in the
            // slot the two exponentials land does not matter -- the MFMA cannot sink below the volatile adds that follow it)
            p0 = __builtin_amdgcn_exp2f(sc[1 - SET][bj >> 1][8 * (bj & 1) + e]);
            p1 = __builtin_amdgcn_exp2f(sc[1 - SET][bj >> 1][8 * (bj & 1) + e + 1]);
        };
        auto chunk_fin = [&](int c, float p0, float p1) {
            const int bj = c >> 2, e = 2 * (c & 3);
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(lta) : "v"(p0));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(ltb) : "v"(p1));
            unsigned wv;
            if constexpr (F16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(wv) : "v"(p0), "v"(p1));
            else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(wv) : "v"(p0), "v"(p1));
            pf[SQ][bj][e >> 1] = wv;
        };
        auto s_mfma = [&](f16v& acc, op8 a, op8 b, bool first) {
            if constexpr (F16) {
                if (first) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
            } else {
                if (first) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
            }
        };
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = i >> 1;
            if ((i & 1) == 0) s_mfma(sc[SET][j & 1], kf[j], qf[U][j >> 1], j < 2);
            else mfma_acc_agpr(o[U][j & 1], vf[j], __builtin_bit_cast(op8, pf[U][j >> 1]));
            if (!(ABL & 128) && REFILL) {
                if (i & 1) vf[j] = ldv(VB, j);
                else kf[j] = ldk(KB, j);
            }
            if (DMA && (i & 3) == 2 && !((ABL & 2) && t > 0)) issue_one(t + A4_AHEAD, std::integral_constant<int, DB>{}, i >> 2);
            float p0, p1;
            chunk_exp(i, p0, p1);
            if (i > 0) chunk_fin(i - 1, q0e, q1e);
            asm volatile("" : "+v"(p0), "+v"(p1));   // the exponentials stay in this slot
            q0e = p0;
            q1e = p1;
            __builtin_amdgcn_sched_barrier(0);
        }
        chunk_fin(15, q0e, q1e);
        l_run[SQ] += lta + ltb;
        if (t == ntiles - 1 && (S & 63) != 0) mask_tail(sc[SET], t);
    };
    // one key tile: NQ sub-steps; the score set of sub-step u is (PAR0 + u) & 1
    auto key_tile = [&](auto vb_tag, auto par_tag, int t) {
        constexpr int PAR0 = decltype(par_tag)::value;
        vm_wait<4>();
        if (!(ABL & 16)) __syncthreads();
        sub_step(std::integral_constant<int, 0>{}, vb_tag, std::integral_constant<int, PAR0 & 1>{}, t);
        sub_step(std::integral_constant<int, 1>{}, vb_tag, std::integral_constant<int, (PAR0 + 1) & 1>{}, t);
        if constexpr (NQ > 2) sub_step(std::integral_constant<int, 2>{}, vb_tag, std::integral_constant<int, (PAR0 + 2) & 1>{}, t);
        if constexpr (NQ > 3) sub_step(std::integral_constant<int, 3>{}, vb_tag, std::integral_constant<int, (PAR0 + 3) & 1>{}, t);
    };
    // four key tiles per trip (ring of four buffers); the parity of the global sub-step count at the start of key tile k is (k NQ) & 1
    for (int t = 0; t < ntiles; t += A4_NB) {
        key_tile(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntiles) key_tile(std::integral_constant<int, 1>{}, std::integral_constant<int, NQ & 1>{}, t + 1);
        if (t + 2 < ntiles) key_tile(std::integral_constant<int, 2>{}, std::integral_constant<int, (2 * NQ) & 1>{}, t + 2);
        if (t + 3 < ntiles) key_tile(std::integral_constant<int, 3>{}, std::integral_constant<int, (3 * NQ) & 1>{}, t + 3);
    }
    // epilogue: the softmax of tile NQ - 1 over the last key tile, then PV(u, last) for every tile.  Which score set holds those scores
    // depends on the number of key tiles: the last sub-step of key tile k wrote set (k NQ + NQ - 1) & 1
    {
        const int last_set = ((ntiles - 1) * NQ + NQ - 1) & 1;
        float lta = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int bj = c >> 2, e = 2 * (c & 3);
            const float s0 = last_set ? sc[1][bj >> 1][8 * (bj & 1) + e] : sc[0][bj >> 1][8 * (bj & 1) + e];
            const float s1 = last_set ? sc[1][bj >> 1][8 * (bj & 1) + e + 1] : sc[0][bj >> 1][8 * (bj & 1) + e + 1];
            const float p0 = __builtin_amdgcn_exp2f(s0), p1 = __builtin_amdgcn_exp2f(s1);
            lta += p0 + p1;
            const op2 pk = {(op_t)p0, (op_t)p1};
            pf[NQ - 1][bj][e >> 1] = __builtin_bit_cast(unsigned, pk);
        }
        l_run[NQ - 1] += lta;
#pragma unroll
        for (int qt = 0; qt < NQ; ++qt)
#pragma unroll
            for (int g = 0; g < 8; ++g) mfma_acc_agpr(o[qt][g & 1], vf[g], __builtin_bit_cast(op8, pf[qt][g >> 1]));
    }
    agpr_settle();
    vm_wait<0>();
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
        float a, b;
        halves(l_run[qt], a, b);
        const float inv = 1.f / (a + b);
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
                    const int d = db * 32 + 16 * pr + 8 * hi;
                    *reinterpret_cast<u4v*>(orow + d) = out;
                }
        }
    }
}

}  // namespace ATT2_NS
