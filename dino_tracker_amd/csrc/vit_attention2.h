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
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// QT: 32-query tiles per wave (1: 256 queries per workgroup, <= 128 VGPRs; 2: 512 queries, <= 256 VGPRs)
// grid: 1-D, 8 * ceil(F*heads / 8) * QB blocks, QB = ceil(S / (256 * QT))
template <int QT>
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
    f16v negm[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = 0.f;
        l_run[qt] = f2{0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[qt][r] = 0.f;
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
        // ---- S^T = K Q^T : two 32-key blocks x four 16-wide d steps ----
        f16v sc[QT][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf8 kf = *reinterpret_cast<const bf8*>(tb + (koff[b] ^ (ks << 5)));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    sc[qt][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qt][ks], ks == 0 ? negm[qt] : sc[qt][b], 0, 0, 0);
            }
        }
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
        // ---- online softmax (exp2 domain; Q carries log2(e)/sqrt(d)) : everything per query is lane-local ----
        bf8 pf[QT][2][2];  // P^T fragments: [query tile][key block][16-key group]
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float tm = -3e38f;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) tm = fmaxf(tm, sc[qt][b][r]);
            tm = fmaxf(tm, __shfl_xor(tm, 32, 64));  // largest score of the tile relative to m_run
            if (t == 0 || !__all(tm <= 8.f)) {
                asm volatile("; rescale" ::: "memory");
                const float up = t == 0 ? tm : fmaxf(tm, 0.f);     // m_new - m_run
                const float alpha = __builtin_amdgcn_exp2f(-up);
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
        // ---- O^T += V^T P^T : A fragment of (d-block db, keys 32b + 16j + 8hi .. +7) = one 16-byte read ----
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf8 vf = *reinterpret_cast<const bf8*>(tb + (voff[db] ^ ((4 * b + 2 * j) << 4)));
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        o[qt][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qt][b][j], o[qt][db], 0, 0, 0);
                }
        vm_wait<2>();   // this wave's requests of tile t+1 have landed; those of tile t+2 stay in flight
        __syncthreads();
        buf = buf == 2 ? 0 : buf + 1;
    }
    vm_wait<0>();
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float l_half = l_run[qt][0] + l_run[qt][1];
        const float l_tot = l_half + __shfl_xor(l_half, 32, 64);
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

inline unsigned attention2_grid(int FH, int S, int QT, int* qb_out) {
    const int QB = (S + 256 * QT - 1) / (256 * QT);
    *qb_out = QB;
    return (unsigned)(((FH + 7) / 8) * 8 * QB);
}

}  // namespace att2
