// attention_v1.h -- the round-1 attention kernel, kept for A/B timing in attn_bench.hip only (the library runs
// dino_tracker_amd/csrc/vit_attention2.h).  4 waves x 64 queries per workgroup, register-staged K / V^T tiles with padded
// LDS pitches, running maximum per tile through the C operand of the first score MFMA.
#pragma once
namespace {
// ---------------------------------------------------------------------------------------------------------------
// flash attention, d_head = 64.  One workgroup = 128 queries (4 waves x 32) of one (frame, head); KV tiles of 64 keys.
// ---------------------------------------------------------------------------------------------------------------
constexpr int KP = 72;   // K tile row pitch in bf16 (144 B): ds_read_b128 of 16 rows hits 16 distinct 16-B slots
constexpr int VP = 68;   // V^T tile row pitch in bf16 (136 B): ds_read_b64 of 32 rows hits 32 distinct 8-B slots
constexpr int ATT_QT = 2;  // 32-query tiles per wave

__global__ __launch_bounds__(256) void attention_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kg,
                                                        const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O, int S,
                                                        int Sp, int heads, int D) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[2][64 * KP];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[2][64 * VP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fh = blockIdx.y;  // frame * heads + head
    const int frame = fh / heads, head = fh - frame * heads;
    // a wave owns ATT_QT x 32 queries: every K / V^T fragment read from LDS feeds ATT_QT MFMAs (the kernel is bound by
    // LDS bandwidth, not by the matrix cores, when each fragment is used once)
    const int q0 = blockIdx.x * (128 * ATT_QT) + w * (32 * ATT_QT);
    const int lq = lane & 31, hi = lane >> 5;
    const bf16_t* Qb = Q + (size_t)fh * Sp * 64;
    const bf16_t* Kb = Kg + (size_t)fh * Sp * 64;
    const bf16_t* Vb = Vt + (size_t)fh * 64 * Sp;
    // Q^T fragments (B operand): lane (query lq, hi) holds d = 16*ks + 8*hi .. +7 for ks = 0..3
    bf8 qf[ATT_QT][4];
#pragma unroll
    for (int qt = 0; qt < ATT_QT; ++qt) {
        const int qrow = min(q0 + qt * 32 + lq, Sp - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[qt][ks] = *reinterpret_cast<const bf8*>(Qb + (size_t)qrow * 64 + ks * 16 + hi * 8);
    }
    f16v o[ATT_QT][2];  // O^T accumulators: d-block db: rows d = 32*db + (r&3) + 8*(r>>2) + 4*hi, column = query lq
    // m_run: softmax reference of this lane's query; negm = -m_run broadcast over an accumulator-shaped vector, used as
    // the C operand of the first score MFMA, so that the scores arrive as s - m_run and the softmax needs no subtraction
    // (a SIMD hides only ~4 VALU instructions under one 32x32x16 MFMA; everything beyond that is exposed)
    float m_run[ATT_QT];
    f2 l_run[ATT_QT];
    f16v negm[ATT_QT];
#pragma unroll
    for (int qt = 0; qt < ATT_QT; ++qt) {
        m_run[qt] = 0.f;
        l_run[qt] = f2{0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[qt][r] = 0.f;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qt][db][r] = 0.f;
    }
    const int ntiles = (S + 63) / 64;
    // loader: thread -> (row, 16-byte piece) x 2 for each of K and V^T
    const int lr = tid >> 3, lp = tid & 7;  // rows lr, lr + 32; piece lp (8 bf16)
    uint4 rk0, rk1, rv0, rv1;
#define ATT_LOAD_TILE(t)                                                                          \
    do {                                                                                          \
        const int k0_ = (t) * 64;                                                                 \
        rk0 = *reinterpret_cast<const uint4*>(Kb + (size_t)(k0_ + lr) * 64 + lp * 8);             \
        rk1 = *reinterpret_cast<const uint4*>(Kb + (size_t)(k0_ + lr + 32) * 64 + lp * 8);        \
        rv0 = *reinterpret_cast<const uint4*>(Vb + (size_t)lr * Sp + k0_ + lp * 8);               \
        rv1 = *reinterpret_cast<const uint4*>(Vb + (size_t)(lr + 32) * Sp + k0_ + lp * 8);        \
    } while (0)
#define ATT_STORE_TILE(buf)                                                                       \
    do {                                                                                          \
        *reinterpret_cast<uint4*>(&Ks[buf][lr * KP + lp * 8]) = rk0;                              \
        *reinterpret_cast<uint4*>(&Ks[buf][(lr + 32) * KP + lp * 8]) = rk1;                       \
        uint2* v0_ = reinterpret_cast<uint2*>(&Vs[buf][lr * VP + lp * 8]);                        \
        v0_[0] = make_uint2(rv0.x, rv0.y);                                                        \
        v0_[1] = make_uint2(rv0.z, rv0.w);                                                        \
        uint2* v1_ = reinterpret_cast<uint2*>(&Vs[buf][(lr + 32) * VP + lp * 8]);                 \
        v1_[0] = make_uint2(rv1.x, rv1.y);                                                        \
        v1_[1] = make_uint2(rv1.z, rv1.w);                                                        \
    } while (0)
    ATT_LOAD_TILE(0);
    ATT_STORE_TILE(0);
    // touch the Q fragments here: otherwise the compiler places their (first-iteration) vmcnt waits in front of the
    // score MFMAs INSIDE the loop, where in steady state they wait for the next tile's K / V prefetch instead
#pragma unroll
    for (int qt = 0; qt < ATT_QT; ++qt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[qt][ks]));
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) ATT_LOAD_TILE(t + 1);
        // ---- S^T = K Q^T : two 32-key blocks x four 16-wide d steps ----
        f16v sc[ATT_QT][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf8 kf = *reinterpret_cast<const bf8*>(&Ks[cur][(b * 32 + lq) * KP + ks * 16 + hi * 8]);
#pragma unroll
                for (int qt = 0; qt < ATT_QT; ++qt)
                    sc[qt][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qt][ks], ks == 0 ? negm[qt] : sc[qt][b], 0, 0, 0);
            }
        }
        // keys beyond S (last tile only) are masked out
        if (t == ntiles - 1 && (S & 63) != 0) {
#pragma unroll
            for (int qt = 0; qt < ATT_QT; ++qt)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * 64 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (key >= S) sc[qt][b][r] = -1e30f;
                    }
        }
        // ---- online softmax (exp2 domain; Q carries log2(e)/sqrt(d)) : everything per query is lane-local ----
        bf8 pf[ATT_QT][2][2];  // P^T fragments: [query tile][key block][16-slot group]
#pragma unroll
        for (int qt = 0; qt < ATT_QT; ++qt) {
            float tm = -3e38f;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) tm = fmaxf(tm, sc[qt][b][r]);
            tm = fmaxf(tm, __shfl_xor(tm, 32, WAVE));  // largest score of the tile relative to m_run
            // deferred maximum: the reference only moves (and O, l are only rescaled) when some query of the wave sees a
            // score more than 8 above it, so P stays <= 2^8 and the common path has no rescale.  The first tile always
            // takes this path (m_run = 0 is not a reference yet).
            if (t == 0 || !__all(tm <= 8.f)) {
                asm volatile("; rescale" ::: "memory");
                const float up = t == 0 ? tm : fmaxf(tm, 0.f);     // m_new - m_run
                const float alpha = __builtin_amdgcn_exp2f(-up);   // raw v_exp_f32 (never used on the first tile: l = o = 0)
                m_run[qt] += up;
                l_run[qt] *= f2{alpha, alpha};
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qt][db][r] *= alpha;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[qt][b][r] -= up;
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[qt][r] = -m_run[qt];
            }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f2 p = {__builtin_amdgcn_exp2f(sc[qt][b][8 * j + e]), __builtin_amdgcn_exp2f(sc[qt][b][8 * j + e + 1])};
                        l_run[qt] += p;
                        pf[qt][b][j][e] = (bf16_t)p[0];
                        pf[qt][b][j][e + 1] = (bf16_t)p[1];
                    }
        }
        // ---- O^T += V^T P^T : slot (hi, e) of group (b, j) is key 32b + 16j + 8(e>>2) + 4hi + (e&3) ----
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16_t* vp = &Vs[cur][(db * 32 + lq) * VP + b * 32 + j * 16 + 4 * hi];
                    const bf4 v0 = *reinterpret_cast<const bf4*>(vp), v1 = *reinterpret_cast<const bf4*>(vp + 8);
                    const bf8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                    for (int qt = 0; qt < ATT_QT; ++qt)
                        o[qt][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qt][b][j], o[qt][db], 0, 0, 0);
                }
        if (t + 1 < ntiles) ATT_STORE_TILE(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int qt = 0; qt < ATT_QT; ++qt) {
        const float l_half = l_run[qt][0] + l_run[qt][1];
        const float l_tot = l_half + __shfl_xor(l_half, 32, WAVE);
        const float inv = 1.f / l_tot;
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


}  // namespace
