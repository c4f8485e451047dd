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

// Row sums of P on the matrix pipe: v_mfma_f32_4x4x4 (16 blocks of 4 x 4, K = 4) with A = ones makes every output row of a
// lane's column the sum of the lane's OWN four B values, accumulated in fp32 -- a lane-local sum of four 16-bit P entries for
// one issue slot and 8 cycles of the matrix pipe, against four v_add_f32.  Why it matters: a lone wave hides about four VALU
// instructions under a 32x32x16 MFMA and pays 6-8 cycles for every further one; the softmax chunk of a slot
// is 2 exponentials + 1 packed convert + 2 adds = five.
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4v rowsum4(unsigned w0, unsigned w1, f4v acc) {
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    const u2v b = {w0, w1};
    if constexpr (F16) {
        typedef _Float16 h4v __attribute__((ext_vector_type(4)));
        const h4v ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
        return __builtin_amdgcn_mfma_f32_4x4x4f16(ones, __builtin_bit_cast(h4v, b), acc, 0, 0, 0);
    } else {
        typedef short s4v __attribute__((ext_vector_type(4)));
        const s4v ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80};
        return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones, __builtin_bit_cast(s4v, b), acc, 0, 0, 0);
    }
}

// LDS-DMA request through a buffer descriptor: address = descriptor base + scalar offset (the tile) + 32-bit per-lane offset
// (constant for the whole kernel: one VGPR per request stream); the LDS destination is a per-wave scalar + an immediate (the
// ring buffer).  Three issue slots per request -- the global_load_lds form of attention2 cost a lone wave ~13 (64-bit address
// arithmetic per request, m0 saved and restored), which showed as 9 % of this kernel's time.
// (m0 is not otherwise used in this kernel: no indirect register indexing, no GWS / sendmsg.)
template <int LDS_IMM>
__device__ __forceinline__ void buffer_lds16(u4v srd, unsigned soff, unsigned voff, unsigned lds_dst) {
    asm volatile(
        "s_add_u32 m0, %3, %4\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %0, %1, %2 offen lds"
        :
        : "v"(voff), "s"(srd), "s"(soff), "s"(lds_dst), "i"(LDS_IMM)
        : "memory", "scc");   // (s_add_u32 writes SCC: without the clobber the compiler keeps a compare result live across the asm --
                              //  round 4 found this as wrong tokens in ONE instantiation of the weight-stationary GEMM)
}
__device__ __forceinline__ u4v make_srd(const void* p) {
    const unsigned long long a = (unsigned long long)(size_t)p;
    u4v r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};   // raw buffer, stride 0, no bounds in the way (common.h: dtk_make_srd)
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = __builtin_amdgcn_readfirstlane(r[k]);
    return r;
}

inline unsigned attention4_grid(int FH, int S, int* qb_out) {
    const int QB = (S + 255) / 256;
    *qb_out = QB;
    return (unsigned)(((FH + 7) / 8) * 8 * QB);
}

// ABL (micro-benchmark only; 0 in the library): 1 no exponentials, 2 no LDS-DMA after the prologue, 8 no guards / estimate,
// 16 no barrier, 128 no LDS fragment reads after the prologue, 256 shader-clock count of workgroup 0 into the first output words.
// RSM: row sums on the matrix pipe (rowsum4) instead of VALU adds -- measured equal (fp16) to 2.6 % faster (bf16), off in the
// library (profiles/r04_attention_v4.txt; packed v_pk_add_f32 row sums were 6-9 % SLOWER and are gone).
template <int ABL = 0, bool RSM = false>
__global__ __launch_bounds__(256, 1) void attention4_kernel(const op_t* __restrict__ Q, const op_t* __restrict__ Kg,
                                                            const op_t* __restrict__ Vt, op_t* __restrict__ O, int S, int Sp,
                                                            int heads, int D, int FH, int QB) {
    __shared__ __attribute__((aligned(1024))) unsigned char tiles[A4_NB][TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long clk0 = (ABL & 256) ? __builtin_amdgcn_s_memtime() : 0ull;   // (micro-benchmark: shader clock)
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

    // DMA sources of this lane: wave w fills rows 16w .. 16w+15 of the K part and of the V^T part (two requests each);
    // byte offsets from the tile's K base (Kb + t * 64 keys * 128 B) and V^T base (Vb + t * 64 keys * 2 B)
    unsigned kvo[2], vvo[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int lrow = (w * 2 + r) * 8 + (lane >> 3), lpc = (lane & 7) ^ ((lrow >> 1) & 7);
        kvo[r] = (unsigned)(lrow * 64 + lpc * 8) * 2u;
        vvo[r] = (unsigned)(lrow * Sp + lpc * 8) * 2u;
    }
    const unsigned lds_base = (unsigned)(size_t)&tiles[0][0];
    const int ntiles = (S + 63) / 64;
    // request i (0..3) of tile t into ring buffer BUF: K rows 16w.., 16w+8.., then the V^T rows
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
    // Q^T fragments (B operand): lane (query lq, hi) holds d = 16 ks + 8 hi .. + 7
    op8 qf[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qrow = min(q0 + qt * 32 + lq, Sp - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[qt][ks] = *reinterpret_cast<const op8*>(Qb + (size_t)qrow * 64 + ks * 16 + hi * 8);
    }
    // The key rows of the reference estimate (below): keys 0..31, 32..63 and the two query tiles' own 32 keys -- requested HERE, in
    // ONE batch with the Q fragments and in front of the tile requests (round 5).  The prologue used to be three dependent trips to
    // memory -- Q + the twelve tile requests drained (the compiler's wait for Q is a vmcnt(0) once asm-issued requests follow its
    // loads), then the estimate's first three key blocks, then the fourth -- 10.2 k cycles of a workgroup's 247 k with nothing to
    // overlap them (one workgroup per CU); batched it is one trip.  Same values, same arithmetic.
    op8 kq[4][4];
    if (!(ABL & 8)) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const int kr0 = blk < 2 ? blk * 32 : q0 + (blk - 2) * 32;
            const op_t* kp = Kb + (size_t)min(kr0 + lq, Sp - 1) * 64 + hi * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kq[blk][ks] = *reinterpret_cast<const op8*>(kp + ks * 16);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        issue_one(0, std::integral_constant<int, 0>{}, j);
        issue_one(1, std::integral_constant<int, 1>{}, j);
        issue_one(2, std::integral_constant<int, 2>{}, j);
    }
    static_assert(A4_AHEAD == 3, "prologue requests tiles 0..2");

    f16v o[2][2];   // O^T accumulators [query tile][d block]: rows d = 32 db + (r & 3) + 8 (r >> 2) + 4 hi, column = query lq
    f16v sc[2][2];  // S^T accumulators [query tile][key block]: register r of lane-half hi = key 32 b + 16 (r >> 3) + 8 hi + (r & 7)
    u4v pf[2][4];   // P^T fragments [query tile][16-key group], as the four 32-bit words of an op8
    op8 kf[8], vf[8];
    // The reference of a query enters its scores through the C operand of the first MFMA of every score block: every
    // accumulator register of a lane belongs to ONE query (the column), so C = (-m, ..., -m) makes the matrix pipe deliver
    // s - m at no VALU cost (attention2 needed a subtracting and a non-subtracting copy of its loop for this).
    float m_run[2] = {0.f, 0.f};
    f16v negm[2];
    float l_run[2] = {0.f, 0.f};
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
    // fragment addresses inside a ring buffer: K fragment f = (key block b = f & 1, d step ks = f >> 1) at ka[ks] + 4096 b,
    // V^T fragment g = (d block db = g & 1, 16-key group bj = g >> 1) at va[bj] + 4096 db; + 16384 * buffer: all immediates
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
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[qt][ks]));
    // reference estimate (attention2 MODE 1): keys 0..63 and the query tile's own 32 keys
    if (!(ABL & 8)) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float tm = -3e38f;
#pragma unroll
            for (int blk = 0; blk < 3; ++blk) {
                const int kr0 = blk < 2 ? blk * 32 : q0 + qt * 32;
                const int kb = blk < 2 ? blk : 2 + qt;   // the batch loaded in front of the tile requests
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
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[qt][r] = -m_run[qt];
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) vf[g][e] = (op_t)0.f;   // E(0) multiplies them with P = 0 (initialised here: the registers held the estimate's key rows until now)
    vm_wait<0>();  // tiles 0 .. A4_AHEAD-1 have landed
    __syncthreads();

    // (a 32-bit LDS address -> ds_read_b128 base + immediate; the host pass of the compiler, for which an LDS pointer has no
    // meaning, would warn about the integer's width)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
    typedef const __attribute__((address_space(3))) op8* lds_frag_ptr;
    auto ldk = [&](int buf, int f) { return *(lds_frag_ptr)(ka[f >> 1] + (unsigned)(buf * TILE_BYTES + (f & 1) * 4096)); };
    auto ldv = [&](int buf, int g) { return *(lds_frag_ptr)(va[g >> 1] + (unsigned)(buf * TILE_BYTES + (g & 1) * 4096)); };
#pragma clang diagnostic pop
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
    // called after the PV product of that tile has been issued.  The scores of the NEXT key tile are already in sc[qt]
    // (computed against the old reference): they move with it.
    auto guard_tripped = [&](int qt, float lsum) {
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
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[qt][b2][r] -= k;
            l_run[qt] *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qt][db][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[qt][r] = -m_run[qt];
        }
    };

    // prologue: K(0) fragments, S(0,0)
#pragma unroll
    for (int f = 0; f < 8; ++f) kf[f] = ldk(0, f);
#pragma unroll
    for (int f = 0; f < 8; ++f) sc[0][f & 1] = ATT2_MFMA(kf[f], qf[0][f >> 1], f < 2 ? negm[0] : sc[0][f & 1], 0, 0, 0);
    if (ntiles == 1 && (S & 63) != 0) mask_tail(sc[0], 0);

    // One half-step.  SQ = the query tile whose softmax runs on the VALU; the MFMAs belong to the other one (MQ): its scores
    // against the key tile held in kf and its PV product with the V^T tile held in vf.  E half-steps (SQ = 0): every fragment
    // register is refilled from LDS (K part of buffer KB = next key tile, V^T part of buffer VB = this one) right after its
    // last use.  O half-steps (SQ = 1) carry the four LDS-DMA requests of tile t + A4_AHEAD (buffer DB) in four of their slots.
    // Returns this lane's part of the row sum of query tile SQ over the key tile.
    auto half_step = [&](auto sq_tag, auto vb_tag, int t) -> float {
        constexpr int SQ = decltype(sq_tag)::value, MQ = 1 - SQ;
        constexpr int VB = decltype(vb_tag)::value, KB = (VB + 1) & (A4_NB - 1), DB = (VB + A4_AHEAD) & (A4_NB - 1);
        constexpr bool EVEN = SQ == 0;
        constexpr bool ADDS = !RSM;
        float lta = 0.f, ltb = 0.f;   // two independent row-sum chains (a dependent add right behind an add stalls a lone wave)
        f4v lacc = {0.f, 0.f, 0.f, 0.f};   // RSM: this lane's part of the row sum of query tile MQ over the tile being multiplied
        float q0 = 0.f, q1 = 0.f;     // the exponentials of the previous slot's chunk: summed and packed one slot later
        // VALU chunk c = values 2c, 2c+1 of the 32 per lane: 16-key group bj = c >> 2, element pair e = 2 (c & 3).
        // Software-pipelined by one slot: slot i issues the two exponentials of chunk i and the packed convert (and the adds) of
        // chunk i-1 -- a transcendental's result is not demanded in the slot that issues it.
        auto chunk_exp = [&](int c, float& p0, float& p1) {
            const int bj = c >> 2, e = 2 * (c & 3);
            const float s0 = sc[SQ][bj >> 1][8 * (bj & 1) + e], s1 = sc[SQ][bj >> 1][8 * (bj & 1) + e + 1];
            p0 = (ABL & 1) ? s0 * 0.01f : __builtin_amdgcn_exp2f(s0);
            p1 = (ABL & 1) ? s1 * 0.01f : __builtin_amdgcn_exp2f(s1);
        };
        auto chunk_fin = [&](int c, float p0, float p1) {
            const int bj = c >> 2, e = 2 * (c & 3);
            if (ADDS) { lta += p0; ltb += p1; }
            const op2 pk = {(op_t)p0, (op_t)p1};
            unsigned wv = __builtin_bit_cast(unsigned, pk);
            // anchors: the results are demanded HERE, in front of the slot's sched_barrier (pure arithmetic is otherwise free to
            // sink to the end of the half-step, where nothing hides it)
            if (!ADDS) asm volatile("" : "+v"(wv));
            else asm volatile("" : "+v"(wv), "+v"(lta), "+v"(ltb));
            pf[SQ][bj][e >> 1] = wv;
        };
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = i >> 1;
            if ((i & 1) == 0) {
                sc[MQ][j & 1] = ATT2_MFMA(kf[j], qf[MQ][j >> 1], j < 2 ? negm[MQ] : sc[MQ][j & 1], 0, 0, 0);
                // RSM: the row sum of P fragment (MQ, j >> 1), half j & 1 = the four values the PV product of slots 4 (j >> 1) + 1, + 3 multiplies
                if (RSM) lacc = rowsum4(pf[MQ][j >> 1][2 * (j & 1)], pf[MQ][j >> 1][2 * (j & 1) + 1], lacc);
            } else {
                mfma_acc_agpr(o[MQ][j & 1], vf[j], __builtin_bit_cast(op8, pf[MQ][j >> 1]));
            }
            // Fragment refills and DMA requests, ten per half-step.  A fragment register is refilled between its last use
            // (E: S(1,t) in slot 2j for kf[j], PV(1,t-1) in slot 2j+1 for vf[j]) and its next one in O (PV(0,t) in slot 2j,
            // S(0,t+1) in slot 2j+1), at least three slots ahead of it:
            //   E: vf[j] right after its use (odd slots), kf[0] in slot 8, kf[1] in slot 12
            //   O: kf[j], j = 2..7, in slot 2j-3 (odd slots 1..11); the four LDS-DMA requests of tile t + A4_AHEAD in slots 2, 6, 10, 14
            if (!(ABL & 128)) {
                if (EVEN && (i & 1)) vf[j] = ldv(VB, j);
                if (EVEN && i == 8) kf[0] = ldk(KB, 0);
                if (EVEN && i == 12) kf[1] = ldk(KB, 1);
                if (!EVEN && (i & 1) && i <= 11) kf[(i + 3) >> 1] = ldk(KB, (i + 3) >> 1);
            }
            if (!EVEN && (i & 3) == 2 && !((ABL & 2) && t > 0)) issue_one(t + A4_AHEAD, std::integral_constant<int, DB>{}, i >> 2);
            float p0, p1;
            chunk_exp(i, p0, p1);
            if (i > 0) chunk_fin(i - 1, q0, q1);
            asm volatile("" : "+v"(p0), "+v"(p1));
            q0 = p0;
            q1 = p1;
            __builtin_amdgcn_sched_barrier(0);
        }
        chunk_fin(15, q0, q1);
        if (RSM) {   // the sum belongs to query tile MQ (the tile whose PV product this half-step issued)
            l_run[MQ] += lacc[0];
            return lacc[0];
        }
        const float lt = lta + ltb;
        l_run[SQ] += lt;
        return lt;
    };

    int pend1 = 0;   // wave-uniform: query tile 1's guard tripped in O(t-1); handled after PV(1,t-1), i.e. after E(t)
    float lsum1 = 0.f;
    // one key tile t whose ring buffer is VB = t % A4_NB (a compile-time constant: the loop below is unrolled over the ring so
    // that every LDS address is a per-lane constant + an immediate)
    auto key_tile = [&](auto vb_tag, int t) {
        vm_wait<4>();   // this wave's requests of tile t+1 have landed (those of tile t+2 stay in flight)
        if (!(ABL & 16)) __syncthreads();   // every wave is past E(t-1): buffer (t-1) % NB is free, tile t+1 is visible
        // ---- E(t): S(1,t), PV(1,t-1) || softmax(0,t); fragment registers refilled with K(t+1) / V(t)
        const float ls_e = half_step(std::integral_constant<int, 0>{}, vb_tag, t);
        if (t == ntiles - 1 && (S & 63) != 0) mask_tail(sc[1], t);
        if (RSM) {   // ls_e = row sum of P(1,t-1), whose PV product was just issued
            if (!(ABL & 8) && __builtin_amdgcn_readfirstlane(__any(!(ls_e < RESC_T)))) guard_tripped(1, ls_e);
        } else if (pend1) { guard_tripped(1, lsum1); pend1 = 0; }
        // ---- O(t): PV(0,t), S(0,t+1) || softmax(1,t); LDS-DMA of tile t + A4_AHEAD
        const float ls_o = half_step(std::integral_constant<int, 1>{}, vb_tag, t);
        if (t + 1 == ntiles - 1 && (S & 63) != 0) mask_tail(sc[0], t + 1);
        if (ABL & 8) return;
        if (RSM) {   // ls_o = row sum of P(0,t)
            if (__builtin_amdgcn_readfirstlane(__any(!(ls_o < RESC_T)))) guard_tripped(0, ls_o);
            return;
        }
        lsum1 = ls_o;
        const int trip0 = __builtin_amdgcn_readfirstlane(__any(!(ls_e < RESC_T)));
        pend1 = __builtin_amdgcn_readfirstlane(__any(!(lsum1 < RESC_T)));
        if (trip0) guard_tripped(0, ls_e);
    };
    const unsigned long long clk1 = (ABL & 256) ? __builtin_amdgcn_s_memtime() : 0ull;
    for (int t = 0; t < ntiles; t += A4_NB) {
        key_tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntiles) key_tile(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < ntiles) key_tile(std::integral_constant<int, 2>{}, t + 2);
        if (t + 3 < ntiles) key_tile(std::integral_constant<int, 3>{}, t + 3);
    }
    const unsigned long long clk2 = (ABL & 256) ? __builtin_amdgcn_s_memtime() : 0ull;
    // epilogue: PV(1, last) (and, RSM, its row sum)
    {
        f4v lacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            mfma_acc_agpr(o[1][g & 1], vf[g], __builtin_bit_cast(op8, pf[1][g >> 1]));
            if (RSM) lacc = rowsum4(pf[1][g >> 1][2 * (g & 1)], pf[1][g >> 1][2 * (g & 1) + 1], lacc);
        }
        if (RSM) {
            l_run[1] += lacc[0];
            if (!(ABL & 8) && __builtin_amdgcn_readfirstlane(__any(!(lacc[0] < RESC_T)))) guard_tripped(1, lacc[0]);
        }
    }
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
            // a lane (query lq, half hi) holds d = 8 rq + 4 hi .. + 3 of every row quad rq: 8-byte pieces.  The two halves of a query
            // trade pieces (v_permlane32_swap: the upper half of the even quad's register <-> the lower half of the odd quad's), after
            // which the lower lane owns all 8 values of the even quad and the upper lane all 8 of the odd one: 16-byte stores, half
            // as many (round 5; the store tail of a workgroup is issue-bound, MI355X_MICROARCH.md T21)
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
                        const unsigned first = sw[0], second = sw[1];   // (copied to scalars first: see common.h, wave_allreduce_bits)
                        out[k] = first;
                        out[2 + k] = second;
                    }
                    const int d = db * 32 + 16 * pr + 8 * hi;   // lower lanes: quad 2 pr, upper lanes: quad 2 pr + 1
                    *reinterpret_cast<u4v*>(orow + d) = out;
                }
        }
    }
    if ((ABL & 256) && blockIdx.x == gridDim.x - 8 && tid == 0) {   // cycles of one LATE workgroup (prologue | key loop | epilogue)
        const unsigned long long now = __builtin_amdgcn_s_memtime();   // into output words whose owner (workgroup 0) finished long ago
        unsigned* dst = reinterpret_cast<unsigned*>(O);
        dst[0] = (unsigned)(clk1 - clk0);
        dst[1] = (unsigned)(clk2 - clk1);
        dst[2] = (unsigned)(now - clk2);
    }
}

}  // namespace ATT2_NS
