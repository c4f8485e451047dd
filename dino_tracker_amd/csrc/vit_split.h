// vit_split.h -- the ESCALATED precision of the ViT encoder (round 6): every matrix product of a block on split 16-bit operands.
//
// Included by vit.hip inside its anonymous namespace (it uses Vec / IsF16 / mfma16 / mfma32 / gswz / operand_mode / wave_sum).
//
// Why.  With plain fp16 operands every stored activation and weight is rounded to 11 significant bits; an emulation of exactly
// these roundings in float64 (scripts/p1_error_budget.py) reproduces the measured feature error of the fast path -- 1.3e-4
// relative on the benchmark weights, 5.8e-4 with LayerScale 1.0, 2.1e-3 with DINOv2-like outlier statistics -- and shows that
// no single tensor dominates it (LN output, weights, Q, K, V, attention output, MLP hidden, the pending update: 3e-4 .. 1.3e-3
// each under the outlier weights).  So the escalation is not "one more bit somewhere" but the whole block:
//   x = hi + lo,  hi = T(x),  lo = T(x - hi)        (T = _Float16: exact to max(2^-22 |x|, 2^-25); T = __bf16: 2^-16 |x|, fp32's range)
//   a . b  ~  a_hi b_hi + a_hi b_lo + a_lo b_hi     (three MFMAs, fp32 accumulate; the dropped lo.lo term is 2^-22 / 2^-16 relative)
// on LN output x W_qkv, Q K^T, P V, attention output x W_proj, LN output x W_fc1, GELU hidden x W_fc2; the residual updates are
// added to the fp32 stream inside the GEMM epilogue (no 16-bit pending update); GELU in fp32 through erfc (4.7e-7 absolute).  The same emulation with
// hi + lo operands lands on 3.6e-6 (outlier weights) / 2.1e-7 (benchmark weights) relative -- what the fp32 ORACLE itself is from
// float64 (3.0e-6 / 4.8e-7).  T = __bf16 is the range escalation: a value beyond 65504 needs fp32's exponent, and bf16 hi + lo keeps
// 16 significant bits there (plain bf16 operands: 8).
//
// Kernels (simple, LDS-tiled, ~3x the matrix work of the fast path by construction; this is the strict mode, not the headline):
//   layernorm_split_kernel   LayerNorm (+ a pending 16-bit update of a preceding FAST block) -> hi / lo planes
//   gemm_split_dma_kernel    256 x 128 x 32 tiles by LDS-DMA (ring of three 48 KB stages), the production form;
//   gemm_split_kernel        128 x 128 x 32 tiles, register-prefetched (cross-check, DTK_VIT_TILED_GEMMS): hi / lo planes of A and W in LDS, MFMA 16x16x32 as
//                            (W tile) x (token tile)^T so that a lane owns 4 consecutive features of one token (8-byte / 16-byte
//                            stores); epilogues: Q / K / V^T planes, GELU planes, fp32 residual add, fp32 out (the qkv facet)
//   attention_split_kernel   flash attention, d_head 64: S^T = K Q^T and O^T = V^T P^T on MFMA 32x32x16 (per-query statistics are
//                            lane-local), 4 waves x 32 queries per workgroup, 64-key tiles staged through LDS, plain online
//                            softmax in fp32 (running maximum, rescale every tile), P carried as hi + lo with a 2^10 scale so
//                            that small probabilities stay normal numbers.

constexpr int SP_M = 128, SP_N = 128, SP_K = 32;
enum { SEPI_QKV = 0, SEPI_GELU = 1, SEPI_RESID = 2, SEPI_F32 = 3 };

template <typename T>
struct SplitEpi {
    const float* bias;     // [N]
    float inv_wscale;      // 1 / (what the weight planes were multiplied by: 2^8 for fp16 so that their lo halves stay normal)
    // SEPI_QKV
    T *q_hi, *q_lo, *k_hi, *k_lo, *vt_hi, *vt_lo;   // [F][heads][Sp][64] / [F][heads][64][Sp]
    int S, Sp, heads, D;
    float qscale;
    // SEPI_GELU
    T *out_hi, *out_lo;    // [M][N]
    // SEPI_RESID: x[m][n] += gamma[n] * (acc + bias[n])   (fp32 residual stream, read-modify-write by the lane that owns it)
    float* x;
    const float* gamma;
    // SEPI_F32
    float* out_f32;
    int* ovf;              // fp16: OR 2 (Q / K / V) or 4 (hidden) when a stored value reaches the fp16 limit
};

// hi = T(v), lo = T(v - hi).  `v` is pinned as an fp32 VALUE first: with HIP's default -ffp-contract=fast the compiler otherwise
// fuses the caller's last multiply into the conversions (v_fma_mixlo_f16: T(a * b) rounded ONCE from the exact product) for the lo
// half while the stored hi half is T(fp32(a * b)) -- two different hi's wherever a * b sits within fp32 rounding of a 16-bit
// midpoint (one value in 2^13), and there hi + lo is off by a whole 16-bit ulp.  Found by the stand-alone attention test: median
// error 1e-7, five outputs of 115 200 off by 2^-13 .. 2^-10.
template <typename T>
__device__ __forceinline__ void split2(float v, T& hi, T& lo) {
    asm volatile("" : "+v"(v));
    hi = (T)v;
    lo = (T)(v - (float)hi);
}

// GELU(x) = x Phi(x) through erfc(|z|) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2), t = 1 / (1 + p |z|), z = x / sqrt 2
// (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7): x >= 0: x - (x / 2) erfc(z); x < 0: (x / 2) erfc(|z|) -- no cancellation in the
// negative tail.  Measured against float64 over [-12, 12]: |error| <= 4.7e-7 absolute, <= 2.9e-7 |x| (torch's fp32 gelu: 1.2e-6);
// 14 VALU instructions, two of them transcendental, where libm's erff is ~50 with its two branches taken by different lanes -- the
// erff epilogue was 40 % of the split fc1 GEMM (58.8 ms per step against fc2's 34.5 for the same matrix work).
__device__ __forceinline__ float gelu_erfc(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float c = p * t * __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);   // erfc(|z|)
    const float h = 0.5f * x * c;
    return x >= 0.f ? x - h : h;
}

// ---- LayerNorm -> hi / lo planes ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_split_kernel(float* __restrict__ x, const T* __restrict__ delta,
                                                              const float* __restrict__ gam, const float* __restrict__ bet,
                                                              T* __restrict__ y_hi, T* __restrict__ y_lo, long long rows, int D,
                                                              float eps, int* __restrict__ overflow) {
    typedef typename Vec<T>::t4 T4;
    operand_mode<T>();
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* p = x + row * D;
    float4 v[4];
    float s = 0.f;
    bool sat = false;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
            v[it] = *reinterpret_cast<const float4*>(p + c);
            if (delta) {   // the pending update of a preceding fast block (a split block adds its updates in its own epilogues)
                const T4 d = *reinterpret_cast<const T4*>(delta + row * D + c);
                const float d0 = (float)d[0], d1 = (float)d[1], d2 = (float)d[2], d3 = (float)d[3];
                if (IsF16<T>::value) sat |= !(fmaxf(fmaxf(fabsf(d0), fabsf(d1)), fmaxf(fabsf(d2), fabsf(d3))) < 65504.f);
                v[it].x += d0; v[it].y += d1; v[it].z += d2; v[it].w += d3;
                *reinterpret_cast<float4*>(p + c) = v[it];
            }
            s += (v[it].x + v[it].y) + (v[it].z + v[it].w);
        }
    }
    if (IsF16<T>::value && overflow && __any(sat) && lane == 0) atomicOr(overflow, 1);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (lane * 4 + it * 256 < D) {
            const float a = v[it].x - mean, b = v[it].y - mean, cc = v[it].z - mean, d = v[it].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
            const float4 g = *reinterpret_cast<const float4*>(gam + c), b = *reinterpret_cast<const float4*>(bet + c);
            const float r[4] = {(v[it].x - mean) * rstd * g.x + b.x, (v[it].y - mean) * rstd * g.y + b.y,
                                (v[it].z - mean) * rstd * g.z + b.z, (v[it].w - mean) * rstd * g.w + b.w};
            T4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { T a, b2; split2<T>(r[e], a, b2); h[e] = a; l[e] = b2; }
            *reinterpret_cast<T4*>(y_hi + row * D + c) = h;
            *reinterpret_cast<T4*>(y_lo + row * D + c) = l;
        }
    }
}

// ---- epilogue of one TRANSPOSED 16 x 16 D tile: the lane holds features nb .. nb+3 of token m -------------------------------
template <typename T, int EPI>
__device__ __forceinline__ void split_store_tile(const f4& a, long long m, int nb, long long M, int N, const SplitEpi<T>& e,
                                                 float& amax) {
    typedef typename Vec<T>::t4 T4;
    const float4 b4 = e.bias ? *reinterpret_cast<const float4*>(e.bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
    float v[4] = {a[0] * e.inv_wscale + b4.x, a[1] * e.inv_wscale + b4.y, a[2] * e.inv_wscale + b4.z, a[3] * e.inv_wscale + b4.w};
    if (EPI == SEPI_QKV) {
        const int which = nb / e.D, rem = nb - which * e.D;
        const int head = rem >> 6, dh = rem & 63;
        const float sc = which == 0 ? e.qscale : 1.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= sc;
        if (IsF16<T>::value) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        if (m >= M) return;
        T4 h, l;
#pragma unroll
        for (int r = 0; r < 4; ++r) { T x0, x1; split2<T>(v[r], x0, x1); h[r] = x0; l[r] = x1; }
        const int f = (int)(m / e.S), sp = (int)(m - (long long)f * e.S);
        if (which == 2) {
            const size_t o = (((size_t)f * e.heads + head) * 64 + dh) * e.Sp + sp;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e.vt_hi[o + (size_t)r * e.Sp] = h[r];
                e.vt_lo[o + (size_t)r * e.Sp] = l[r];
            }
        } else {
            const size_t o = (((size_t)f * e.heads + head) * e.Sp + sp) * 64 + dh;
            *reinterpret_cast<T4*>((which == 0 ? e.q_hi : e.k_hi) + o) = h;
            *reinterpret_cast<T4*>((which == 0 ? e.q_lo : e.k_lo) + o) = l;
        }
    } else if (EPI == SEPI_GELU) {
        if (m >= M) return;
        T4 h, l;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float g = gelu_erfc(v[r]);
            if (IsF16<T>::value) amax = fmaxf(amax, fabsf(g));
            T x0, x1;
            split2<T>(g, x0, x1);
            h[r] = x0; l[r] = x1;
        }
        *reinterpret_cast<T4*>(e.out_hi + m * N + nb) = h;
        *reinterpret_cast<T4*>(e.out_lo + m * N + nb) = l;
    } else if (EPI == SEPI_RESID) {
        if (m >= M) return;
        const float4 g4 = *reinterpret_cast<const float4*>(e.gamma + nb);
        float4* xp = reinterpret_cast<float4*>(e.x + m * N + nb);
        float4 xv = *xp;
        xv.x += g4.x * v[0]; xv.y += g4.y * v[1]; xv.z += g4.z * v[2]; xv.w += g4.w * v[3];
        *xp = xv;
    } else {
        if (m >= M) return;
        *reinterpret_cast<float4*>(e.out_f32 + m * N + nb) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---- C[M][N] = (A_hi + A_lo)[M][K] . (W_hi + W_lo)[N][K]^T on three MFMAs per product -----------------------------------------
// gemm_tiled_kernel's structure (128 x 128 x 32 tiles, next k-step prefetched into registers while the current one is multiplied
// out of LDS, XCD-aware block order) with four operand planes: 2 x 4 x 8 KB of LDS, two workgroups per CU.
template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const T* __restrict__ Ah, const T* __restrict__ Al,
                                                            const T* __restrict__ Wh, const T* __restrict__ Wl, long long M, int N,
                                                            int K, SplitEpi<T> e) {
    typedef typename Vec<T>::t8 T8;
    operand_mode<T>();
    __shared__ uint4 Sm[2][4][SP_M * 4];   // [buffer][A_hi, A_lo, W_hi, W_lo][row * 4 + swizzled piece]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ncol = (N + SP_N - 1) / SP_N;
    const long long nrow = (M + SP_M - 1) / SP_M;
    const long long kb = blockIdx.x >> 3;
    const long long row_blk = (kb / ncol) * 8 + (blockIdx.x & 7);
    if (row_blk >= nrow) return;
    const long long m0 = row_blk * SP_M;
    const int n0 = (int)(kb % ncol) * SP_N;
    const int wr = w >> 1, wc = w & 1;   // wave tile 64 tokens x 64 features
    const int fj = lane & 15, fg = lane >> 4;
    const int lrow = tid >> 2, lpiece = tid & 3;
    const long long ar0 = min(m0 + lrow, M - 1), ar1 = min(m0 + lrow + 64, M - 1);
    const int br0 = min(n0 + lrow, N - 1), br1 = min(n0 + lrow + 64, N - 1);
    const size_t ao0 = (size_t)ar0 * K + lpiece * 8, ao1 = (size_t)ar1 * K + lpiece * 8;
    const size_t bo0 = (size_t)br0 * K + lpiece * 8, bo1 = (size_t)br1 * K + lpiece * 8;
    f4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f4{0.f, 0.f, 0.f, 0.f};
    // (named registers, not an array captured by a lambda: that form was placed in scratch memory)
    uint4 r0, r1, r2, r3, r4, r5, r6, r7;
#define SP_FETCH(ks_)                                                                                               \
    do {                                                                                                            \
        const size_t k_ = (size_t)(ks_) * SP_K;                                                                     \
        r0 = *reinterpret_cast<const uint4*>(Ah + ao0 + k_); r1 = *reinterpret_cast<const uint4*>(Ah + ao1 + k_);   \
        r2 = *reinterpret_cast<const uint4*>(Al + ao0 + k_); r3 = *reinterpret_cast<const uint4*>(Al + ao1 + k_);   \
        r4 = *reinterpret_cast<const uint4*>(Wh + bo0 + k_); r5 = *reinterpret_cast<const uint4*>(Wh + bo1 + k_);   \
        r6 = *reinterpret_cast<const uint4*>(Wl + bo0 + k_); r7 = *reinterpret_cast<const uint4*>(Wl + bo1 + k_);   \
    } while (0)
#define SP_STASH(buf_)                                                                          \
    do {                                                                                        \
        const int i0_ = gswz(lrow, lpiece), i1_ = gswz(lrow + 64, lpiece);                      \
        Sm[buf_][0][i0_] = r0; Sm[buf_][0][i1_] = r1; Sm[buf_][1][i0_] = r2; Sm[buf_][1][i1_] = r3; \
        Sm[buf_][2][i0_] = r4; Sm[buf_][2][i1_] = r5; Sm[buf_][3][i0_] = r6; Sm[buf_][3][i1_] = r7; \
    } while (0)
    SP_FETCH(0);
    SP_STASH(0);
    __syncthreads();
    const int nk = K / SP_K;
    int cur = 0;
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) SP_FETCH(ks + 1);
        T8 ah[4], al[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int idx = gswz(wr * 64 + mi * 16 + fj, fg);
            const uint4 vh = Sm[cur][0][idx], vl = Sm[cur][1][idx];
            ah[mi] = *reinterpret_cast<const T8*>(&vh);
            al[mi] = *reinterpret_cast<const T8*>(&vl);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int idx = gswz(wc * 64 + ni * 16 + fj, fg);
            const uint4 vh = Sm[cur][2][idx], vl = Sm[cur][3][idx];
            const T8 bh = *reinterpret_cast<const T8*>(&vh), bl = *reinterpret_cast<const T8*>(&vl);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {   // (W tile) x (token tile)^T: D transposed -- the small terms first
                acc[mi][ni] = mfma16(bl, ah[mi], acc[mi][ni]);
                acc[mi][ni] = mfma16(bh, al[mi], acc[mi][ni]);
                acc[mi][ni] = mfma16(bh, ah[mi], acc[mi][ni]);
            }
        }
        if (ks + 1 < nk) SP_STASH(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // D tile (ni, mi) transposed: lane (fg, fj) holds features n0 + wc*64 + ni*16 + 4 fg + r of token m0 + wr*64 + mi*16 + fj
    float amax = 0.f;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int nb = n0 + wc * 64 + ni * 16 + fg * 4;
        if (nb >= N) continue;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
            split_store_tile<T, EPI>(acc[mi][ni], m0 + wr * 64 + mi * 16 + fj, nb, M, N, e, amax);
    }
    if (IsF16<T>::value && (EPI == SEPI_QKV || EPI == SEPI_GELU) && e.ovf) {
        if (__any(amax >= 65488.f) && lane == 0) atomicOr(e.ovf, EPI == SEPI_QKV ? 2 : 4);
    }
#undef SP_FETCH
#undef SP_STASH
}

// ---- the same product on the LDS-DMA pipeline of gemm_wide_delta_kernel (round 6, second form) -----------------------------------
// 256 tokens x 128 features per workgroup of 8 waves (4 x 2, wave tile 64 x 64), 32-wide k-steps; a stage = A_hi | A_lo (256 rows x
// 64 B each) | W_hi | W_lo (128 rows x 64 B each) = 48 KB arrives by LDS-DMA (buffer_load_dwordx4 ... lds through descriptors; no staging registers, no ds_write),
// ring of three stages, two in flight.  LDS image and swizzle are gemm_tiled_kernel's (four 16-byte pieces per row, gswz), produced
// by giving every DMA lane the matching source address.  Used when N % 128 == 0 (every GEMM of the ViT-S / B / L blocks).
constexpr int SD_M = 256, SD_N = 128, SD_STAGES = 3;
constexpr int SD_STAGE_BYTES = (2 * SD_M + 2 * SD_N) * 64;   // 49152
constexpr int SD_REQ = (2 * SD_M + 2 * SD_N) / 16 / 8;       // DMA requests per wave and stage: 6

constexpr int SD_OPITCH = SD_N * 2 + 8;     // staged 16-bit output rows (hi and lo plane): 256 B + 8
constexpr int SD_FPITCH = SD_N * 4 + 16;    // staged fp32 update rows: 512 B + 16 (the 16-byte writes of 8 tokens cover the 32 banks)

template <typename T, int EPI, bool STAGED = true>
__global__ __launch_bounds__(512, 1) void gemm_split_dma_kernel(const T* __restrict__ Ah, const T* __restrict__ Al,
                                                                const T* __restrict__ Wh, const T* __restrict__ Wl, long long M, int N,
                                                                int K, SplitEpi<T> e) {
    typedef typename Vec<T>::t8 T8;
    operand_mode<T>();
    __shared__ __attribute__((aligned(1024))) unsigned char stages[SD_STAGES * SD_STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncol = N / SD_N;
    const long long nrow = (M + SD_M - 1) / SD_M;
    const long long kb = blockIdx.x >> 3;
    const long long row_blk = (kb / ncol) * 8 + (blockIdx.x & 7);   // the column tiles of a row block back to back on one XCD
    if (row_blk >= nrow) return;
    const long long m0 = row_blk * SD_M;
    const int n0 = (int)(kb % ncol) * SD_N;
    const int wr = w >> 1, wc = w & 1;
    const int fj = lane & 15, fg = lane >> 4;
    // request q = 6 w + i: 0..15 A_hi rows 16 q.., 16..31 A_lo, 32..39 W_hi rows n0 + 16 (q - 32).., 40..47 W_lo
    // (descriptor LDS-DMA, vit.hip wd_issue: the request's LDS slot is q KB into the stage)
    dtk_u4 srd[SD_REQ];
    unsigned voff[SD_REQ];
#pragma unroll
    for (int i = 0; i < SD_REQ; ++i) {
        const int q = w * SD_REQ + i;
        const bool isA = q < 32;
        const int blk = isA ? (q & 15) : ((q - 32) & 7);
        const bool lo = isA ? q >= 16 : q >= 40;
        const int row = blk * 16 + (lane >> 2);
        const int piece = (lane & 3) ^ ((0x1230 >> (((row >> 2) & 3) * 4)) & 3);
        const int trow = isA ? (int)(min(m0 + row, M - 1) - m0) : row;
        srd[i] = dtk_make_srd(isA ? (lo ? Al : Ah) + m0 * K : (lo ? Wl : Wh) + (long long)n0 * K);
        voff[i] = (unsigned)(trow * K + piece * 8) * 2u;
    }
    const unsigned lds0 = (unsigned)(size_t)&stages[0] + (unsigned)w * (SD_REQ * 1024);
    const int nk = K / SP_K;
    auto issue = [&](int ks, int buf) {
        const int kk = min(ks, nk - 1);   // past the end: a harmless repeat keeps the request count per stage uniform
        wd_issue<SD_REQ>(srd, voff, (unsigned)kk * (SP_K * 2), __builtin_amdgcn_readfirstlane(lds0 + buf * SD_STAGE_BYTES));
    };
    f4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f4{0.f, 0.f, 0.f, 0.f};
    const int fsw = (0x1230 >> (((fj >> 2) & 3) * 4)) & 3;
    const unsigned a_off = ((wr * 64 + fj) * 4 + (fg ^ fsw)) * 16;
    const unsigned b_off = 2 * SD_M * 64 + ((wc * 64 + fj) * 4 + (fg ^ fsw)) * 16;
    issue(0, 0);
    issue(1, 1);
    ws_wait<SD_REQ>();   // stage 0 landed (stage 1 may still fly)
    __syncthreads();
    int buf = 0;
    for (int ks = 0; ks < nk; ++ks) {
        issue(ks + 2, buf == 0 ? 2 : buf - 1);   // (buf + 2) % 3: the stage consumed in the previous iteration
        const unsigned char* sb = stages + buf * SD_STAGE_BYTES;
        T8 ah[4], al[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            ah[mi] = *reinterpret_cast<const T8*>(sb + a_off + mi * 1024);
            al[mi] = *reinterpret_cast<const T8*>(sb + SD_M * 64 + a_off + mi * 1024);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const T8 bh = *reinterpret_cast<const T8*>(sb + b_off + ni * 1024);
            const T8 bl = *reinterpret_cast<const T8*>(sb + SD_N * 64 + b_off + ni * 1024);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {   // (W tile) x (token tile)^T: D transposed -- the small terms first
                acc[mi][ni] = mfma16(bl, ah[mi], acc[mi][ni]);
                acc[mi][ni] = mfma16(bh, al[mi], acc[mi][ni]);
                acc[mi][ni] = mfma16(bh, ah[mi], acc[mi][ni]);
            }
        }
        ws_wait<SD_REQ>();   // stage ks + 1 landed; the requests of ks + 2 stay in flight
        __syncthreads();
        buf = buf == 2 ? 0 : buf + 1;
    }
    ws_wait<0>();
    float amax = 0.f;
    typedef typename Vec<T>::t4 T4;
    // Round 6: the epilogues leave through LDS as whole rows (vit.hip, gemm_wide_kernel: one workgroup per CU, nothing overlaps the
    // epilogue, and its 8-byte pieces -- 16 tokens x 32 B per store instruction -- ran at 1.5 TB/s).  The stages are dead here.
    static_assert(2 * SD_M * SD_OPITCH <= SD_STAGES * SD_STAGE_BYTES && SD_M * SD_FPITCH <= SD_STAGES * SD_STAGE_BYTES, "staged tile must fit");
    if (STAGED && (EPI == SEPI_GELU || (EPI == SEPI_QKV && n0 < 2 * e.D))) {
        // hi and lo planes of the 256 x 128 tile, 16-bit, then 256-byte rows of each plane: 16 lanes per row and plane
        __syncthreads();
        const int which = EPI == SEPI_QKV ? n0 / e.D : 0;
        const float sc = (EPI == SEPI_QKV && which == 0) ? e.qscale : 1.f;
        unsigned char* const lo_plane = stages + SD_M * SD_OPITCH;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int cb = wc * 64 + ni * 16 + fg * 4;
            const float4 b4 = e.bias ? *reinterpret_cast<const float4*>(e.bias + n0 + cb) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const f4& a = acc[mi][ni];
                float v[4] = {a[0] * e.inv_wscale + b4.x, a[1] * e.inv_wscale + b4.y, a[2] * e.inv_wscale + b4.z, a[3] * e.inv_wscale + b4.w};
                T4 h, l;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float g = EPI == SEPI_GELU ? gelu_erfc(v[r]) : v[r] * sc;
                    if (IsF16<T>::value) amax = fmaxf(amax, fabsf(g));
                    T x0, x1;
                    split2<T>(g, x0, x1);
                    h[r] = x0; l[r] = x1;
                }
                const int off = (wr * 64 + mi * 16 + fj) * SD_OPITCH + cb * 2;
                *reinterpret_cast<T4*>(stages + off) = h;
                *reinterpret_cast<T4*>(lo_plane + off) = l;
            }
        }
        __syncthreads();
        const int plane = (tid >> 4) & 1, piece = tid & 15;
        T* dst;
        if (EPI == SEPI_GELU) dst = (plane ? e.out_lo : e.out_hi) + n0 + piece * 8;
        else dst = (which == 0 ? (plane ? e.q_lo : e.q_hi) : (plane ? e.k_lo : e.k_hi)) + (piece & 7) * 8;
        const int hh = EPI == SEPI_QKV ? ((n0 - which * e.D) >> 6) + (piece >> 3) : 0;
#pragma unroll 4
        for (int it = 0; it < SD_M / 16; ++it) {
            const int row = it * 16 + (tid >> 5);
            const unsigned char* sp = stages + plane * (SD_M * SD_OPITCH) + row * SD_OPITCH + piece * 16;
            const uint2 lo = *reinterpret_cast<const uint2*>(sp), hi = *reinterpret_cast<const uint2*>(sp + 8);
            const long long m = m0 + row;
            if (m < M) {
                if (EPI == SEPI_GELU) {
                    *reinterpret_cast<uint4*>(dst + m * N) = make_uint4(lo.x, lo.y, hi.x, hi.y);
                } else {
                    const unsigned f = (unsigned)m / (unsigned)e.S, pos = (unsigned)m - f * (unsigned)e.S;   // (M < 2^31 tokens)
                    *reinterpret_cast<uint4*>(dst + (((size_t)f * e.heads + hh) * e.Sp + pos) * 64) = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
            }
        }
    } else if (STAGED && EPI == SEPI_QKV) {
        // V^T tiles (vit.hip, gemm_wide_kernel): both planes staged TRANSPOSED -- sT[feature][token], 16-bit, 520-byte rows -- and written
        // as 8-byte pieces of four tokens along a feature row where positions come in fours, element by element otherwise
        static_assert(2 * SD_N * (SD_M * 2 + 8) <= SD_STAGES * SD_STAGE_BYTES, "transposed planes must fit");
        constexpr int TP = SD_M * 2 + 8;
        __syncthreads();
        unsigned char* const lo_plane = stages + SD_N * TP;
        const int head0 = (n0 - 2 * e.D) >> 6;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int cb = wc * 64 + ni * 16 + fg * 4;
            const float4 b4 = e.bias ? *reinterpret_cast<const float4*>(e.bias + n0 + cb) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const f4& a = acc[mi][ni];
                const float v[4] = {a[0] * e.inv_wscale + b4.x, a[1] * e.inv_wscale + b4.y, a[2] * e.inv_wscale + b4.z, a[3] * e.inv_wscale + b4.w};
                const int off = cb * TP + (wr * 64 + mi * 16 + fj) * 2;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (IsF16<T>::value) amax = fmaxf(amax, fabsf(v[r]));
                    T x0, x1;
                    split2<T>(v[r], x0, x1);
                    *reinterpret_cast<T*>(stages + off + r * TP) = x0;
                    *reinterpret_cast<T*>(lo_plane + off + r * TP) = x1;
                }
            }
        }
        __syncthreads();
        if (((e.S | e.Sp) & 3) == 0) {
            const int g = tid & 63;
            const long long m = m0 + 4 * g;
            if (m < M) {
                const unsigned f = (unsigned)m / (unsigned)e.S, pos = (unsigned)m - f * (unsigned)e.S;
                const size_t o = (((size_t)f * e.heads + head0) * 64) * e.Sp + pos;
#pragma unroll 4
                for (int it = 0; it < SD_N / 8; ++it) {
                    const int c = it * 8 + (tid >> 6);
                    *reinterpret_cast<uint2*>(e.vt_hi + o + (size_t)c * e.Sp) = *reinterpret_cast<const uint2*>(stages + c * TP + g * 8);
                    *reinterpret_cast<uint2*>(e.vt_lo + o + (size_t)c * e.Sp) = *reinterpret_cast<const uint2*>(lo_plane + c * TP + g * 8);
                }
            }
        } else {
            const int t = tid & 255;
            const long long m = m0 + t;
            if (m < M) {
                const unsigned f = (unsigned)m / (unsigned)e.S, pos = (unsigned)m - f * (unsigned)e.S;
                const size_t o = (((size_t)f * e.heads + head0) * 64) * e.Sp + pos;
#pragma unroll 4
                for (int it = 0; it < SD_N / 2; ++it) {
                    const int c = it * 2 + (tid >> 8);
                    e.vt_hi[o + (size_t)c * e.Sp] = *reinterpret_cast<const T*>(stages + c * TP + t * 2);
                    e.vt_lo[o + (size_t)c * e.Sp] = *reinterpret_cast<const T*>(lo_plane + c * TP + t * 2);
                }
            }
        }
    } else if (STAGED && EPI == SEPI_RESID) {
        // acc / scale + bias of the 256 x 128 tile in fp32, then x += gamma * that over whole 512-byte rows
        __syncthreads();
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int cb = wc * 64 + ni * 16 + fg * 4;
            const float4 b4 = e.bias ? *reinterpret_cast<const float4*>(e.bias + n0 + cb) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const f4& a = acc[mi][ni];
                *reinterpret_cast<float4*>(stages + (wr * 64 + mi * 16 + fj) * SD_FPITCH + cb * 4) =
                    make_float4(a[0] * e.inv_wscale + b4.x, a[1] * e.inv_wscale + b4.y, a[2] * e.inv_wscale + b4.z, a[3] * e.inv_wscale + b4.w);
            }
        }
        __syncthreads();
        float* const xp = e.x + n0 + (tid & 31) * 4;
        const float4 g4 = *reinterpret_cast<const float4*>(e.gamma + n0 + (tid & 31) * 4);   // (LayerScale where x is updated: the same fma as the direct form)
#pragma unroll 4
        for (int it = 0; it < SD_M / 16; ++it) {
            const int row = it * 16 + (tid >> 5);
            const float4 u = *reinterpret_cast<const float4*>(stages + row * SD_FPITCH + (tid & 31) * 16);
            if (m0 + row < M) {
                float4* q = reinterpret_cast<float4*>(xp + (m0 + row) * N);
                float4 xv = *q;
                xv.x += g4.x * u.x; xv.y += g4.y * u.y; xv.z += g4.z * u.z; xv.w += g4.w * u.w;
                *q = xv;
            }
        }
    } else {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int nb = n0 + wc * 64 + ni * 16 + fg * 4;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                split_store_tile<T, EPI>(acc[mi][ni], m0 + wr * 64 + mi * 16 + fj, nb, M, N, e, amax);
        }
    }
    if (IsF16<T>::value && (EPI == SEPI_QKV || EPI == SEPI_GELU) && e.ovf) {
        if (__any(amax >= 65488.f) && lane == 0) atomicOr(e.ovf, EPI == SEPI_QKV ? 2 : 4);
    }
}

inline unsigned gemm_split_dma_grid(int N, long long rows) {
    const long long ncol = N / SD_N, nrow = dtk_cdiv(rows, SD_M);
    return (unsigned)(dtk_cdiv(nrow, 8) * 8 * ncol);
}

inline unsigned gemm_split_grid(int N, long long rows) {
    const long long ncol = dtk_cdiv(N, SP_N), nrow = dtk_cdiv(rows, SP_M);
    return (unsigned)(dtk_cdiv(nrow, 8) * 8 * ncol);
}

// ---- flash attention on split operands --------------------------------------------------------------------------------------
// Workgroup = 4 waves x 32 queries of one (frame, head); key tiles of 64 through LDS (K and V^T, hi and lo: 4 x 9 KB, rows padded to
// 144 B so that the 16-byte fragment reads of 32 consecutive rows are conflict-free); the next tile travels global -> registers while
// the current one is multiplied.
//   S^T[key][query]:  A = K rows (lane i <-> key pi(i), pi swaps bits 2 and 3 of i), B = Q^T (lane j <-> query j: contiguous d).
//                     With that row order the 8 accumulator registers 8s .. 8s+7 of lane (j, h) are the keys 16 s + 8 h + 0..7 of
//                     the 32-key sub-tile -- exactly the B operand (P, k-group h) of the second product, no data movement.
//   O^T[d][query]:    A = V^T rows (lane i <-> d, contiguous keys), B = P.
// Every per-query quantity (running maximum m, sum l, the rescale factor) lives in lane j and its partner j + 32.
constexpr int AS_KEYS = 64, AS_PITCH = 144, AS_PLANE = 64 * AS_PITCH;   // bytes
constexpr float AS_PSHIFT = 10.f;   // P carried as 2^10 p: fp16 lo halves of small probabilities stay normal (cancels in O / l)

template <typename T>
__global__ __launch_bounds__(256, 2) void attention_split_kernel(const T* __restrict__ Qh, const T* __restrict__ Ql,
                                                                 const T* __restrict__ Kh, const T* __restrict__ Kl,
                                                                 const T* __restrict__ Vh, const T* __restrict__ Vl,
                                                                 T* __restrict__ Oh, T* __restrict__ Ol, int S, int Sp, int heads,
                                                                 int nqb) {
    typedef typename Vec<T>::t8 T8;
    typedef typename Vec<T>::t4 T4;
    operand_mode<T>();
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * AS_PLANE];   // K_hi | K_lo | Vt_hi | Vt_lo
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fh = blockIdx.x / nqb, qb = blockIdx.x - fh * nqb;
    const int j = lane & 31, h = lane >> 5;
    const int q = qb * 128 + w * 32 + j;              // this lane's query
    const int qrow = min(q, Sp - 1);
    // Q fragments (B operand): lane (j, h) holds Q[q][16 ks + 8 h .. + 7]
    T8 qh[4], ql[4];
    {
        const size_t o = ((size_t)fh * Sp + qrow) * 64 + h * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qh[ks] = *reinterpret_cast<const T8*>(Qh + o + ks * 16);
            ql[ks] = *reinterpret_cast<const T8*>(Ql + o + ks * 16);
        }
    }
    // staging: thread -> row tid / 4 (key of the K tile, d of the V^T tile), pieces 2 (tid & 3), 2 (tid & 3) + 1 of its 8
    const int srow = tid >> 2, spiece = (tid & 3) * 2;
    const T* kh_src = Kh + ((size_t)fh * Sp + srow) * 64 + spiece * 8;
    const T* kl_src = Kl + ((size_t)fh * Sp + srow) * 64 + spiece * 8;
    const T* vh_src = Vh + ((size_t)fh * 64 + srow) * Sp + spiece * 8;
    const T* vl_src = Vl + ((size_t)fh * 64 + srow) * Sp + spiece * 8;
    const int sdst = srow * AS_PITCH + spiece * 16;
    uint4 s0, s1, s2, s3, s4, s5, s6, s7;   // (named registers: see gemm_split_kernel)
#define AS_FETCH(kt_)                                                                                                   \
    do {                                                                                                                \
        const size_t ko_ = (size_t)(kt_) * AS_KEYS * 64, vo_ = (size_t)(kt_) * AS_KEYS;                                  \
        s0 = *reinterpret_cast<const uint4*>(kh_src + ko_); s1 = *reinterpret_cast<const uint4*>(kh_src + ko_ + 8);      \
        s2 = *reinterpret_cast<const uint4*>(kl_src + ko_); s3 = *reinterpret_cast<const uint4*>(kl_src + ko_ + 8);      \
        s4 = *reinterpret_cast<const uint4*>(vh_src + vo_); s5 = *reinterpret_cast<const uint4*>(vh_src + vo_ + 8);      \
        s6 = *reinterpret_cast<const uint4*>(vl_src + vo_); s7 = *reinterpret_cast<const uint4*>(vl_src + vo_ + 8);      \
    } while (0)
#define AS_PUT(pl_, a_, b_)                                                         \
    *reinterpret_cast<uint4*>(lds + (pl_) * AS_PLANE + sdst) = a_;                  \
    *reinterpret_cast<uint4*>(lds + (pl_) * AS_PLANE + sdst + 16) = b_
    // fragment offsets: K rows in the pi order; V^T rows plain
    const int pi = (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1);
    const int k_off = pi * AS_PITCH + h * 16;   // + u * 32 rows + ks * 32 bytes
    const int v_off = j * AS_PITCH + h * 16;    // + dt * 32 rows + (u * 32 + s * 16) keys * 2 bytes
    f16v oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    const int nkt = (S + AS_KEYS - 1) / AS_KEYS;
    AS_FETCH(0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();   // the previous tile's fragment reads are done
        AS_PUT(0, s0, s1); AS_PUT(1, s2, s3); AS_PUT(2, s4, s5); AS_PUT(3, s6, s7);
        __syncthreads();
        if (kt + 1 < nkt) AS_FETCH(kt + 1);
        // ---- S^T = K Q^T, two 32-key sub-tiles.  The two accumulators take turns (a chain of dependent MFMAs on ONE accumulator
        // leaves the matrix pipe idle between links) and the fragments of d-step ks + 1 are requested before the MFMAs of ks (the
        // first form read two fragments, waited for them, issued three MFMAs: the LDS latency sat in front of every group)
        f16v sacc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[u][r] = 0.f;
        {
            T8 kh_[2][2], kl_[2][2];   // [ping-pong][u]
            auto kload = [&](int ks, int pp) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int o = k_off + u * 32 * AS_PITCH + ks * 32;
                    kh_[pp][u] = *reinterpret_cast<const T8*>(lds + o);
                    kl_[pp][u] = *reinterpret_cast<const T8*>(lds + AS_PLANE + o);
                }
            };
            kload(0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int pp = ks & 1;
                if (ks + 1 < 4) kload(ks + 1, pp ^ 1);
                sacc[0] = mfma32(kl_[pp][0], qh[ks], sacc[0]);
                sacc[1] = mfma32(kl_[pp][1], qh[ks], sacc[1]);
                sacc[0] = mfma32(kh_[pp][0], ql[ks], sacc[0]);
                sacc[1] = mfma32(kh_[pp][1], ql[ks], sacc[1]);
                sacc[0] = mfma32(kh_[pp][0], qh[ks], sacc[0]);
                sacc[1] = mfma32(kh_[pp][1], qh[ks], sacc[1]);
            }
        }
        // keys past S (the zero padding rows of the last tile) take no part
        if ((kt + 1) * AS_KEYS > S) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * AS_KEYS + u * 32 + (r >> 3) * 16 + h * 8 + (r & 7);
                    if (key >= S) sacc[u][r] = -INFINITY;
                }
        }
        // ---- online softmax (exp2 domain: Q carries log2(e) / sqrt(d))
        float tmax = sacc[0][0];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[u][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mnew = fmaxf(mrun, tmax);
        // (raw v_exp_f32: its only difference from exp2f -- results below 2^-126 flush to zero -- is irrelevant for weights of a sum
        //  whose largest term is 2^10; exp2f's subnormal-range fix-up was three more VALU instructions per score)
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);   // first tile: exp2(-inf) = 0
        mrun = mnew;
        const float mref = mnew - AS_PSHIFT;
        float lsum = 0.f;
        T8 p_h[2][2], p_l[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(sacc[u][r] - mref);
                lsum += p;
                T x0, x1;
                split2<T>(p, x0, x1);
                p_h[u][r >> 3][r & 7] = x0;
                p_l[u][r >> 3][r & 7] = x1;
            }
        lrun = lrun * alpha + lsum;
        if (!__all(alpha == 1.f)) {   // (after the first tiles the running maximum rarely moves: skip the 32 multiplies then)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        }
        // ---- O^T += V^T P: the two d blocks take turns, the V^T fragments of the next 16-key group are requested ahead
        {
            T8 vh_[2][2], vl_[2][2];   // [ping-pong][dt]
            auto vload = [&](int g, int pp) {   // g = 2 u + s: the 16-key group
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int o = 2 * AS_PLANE + v_off + dt * 32 * AS_PITCH + g * 32;
                    vh_[pp][dt] = *reinterpret_cast<const T8*>(lds + o);
                    vl_[pp][dt] = *reinterpret_cast<const T8*>(lds + AS_PLANE + o);
                }
            };
            vload(0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int pp = g & 1, u = g >> 1, sg = g & 1;
                if (g + 1 < 4) vload(g + 1, pp ^ 1);
                oacc[0] = mfma32(vl_[pp][0], p_h[u][sg], oacc[0]);
                oacc[1] = mfma32(vl_[pp][1], p_h[u][sg], oacc[1]);
                oacc[0] = mfma32(vh_[pp][0], p_l[u][sg], oacc[0]);
                oacc[1] = mfma32(vh_[pp][1], p_l[u][sg], oacc[1]);
                oacc[0] = mfma32(vh_[pp][0], p_h[u][sg], oacc[0]);
                oacc[1] = mfma32(vh_[pp][1], p_h[u][sg], oacc[1]);
            }
        }
    }
    const float ltot = lrun + __shfl_xor(lrun, 32);
    const float inv = 1.f / ltot;
    if (q < S) {
        const int f = fh / heads, head = fh - f * heads;
        const size_t o = ((size_t)f * S + q) * ((size_t)heads * 64) + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                T4 vh4, vl4;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    T x0, x1;
                    split2<T>(oacc[dt][4 * g + c] * inv, x0, x1);
                    vh4[c] = x0; vl4[c] = x1;
                }
                const int d = dt * 32 + 8 * g + 4 * h;
                *reinterpret_cast<T4*>(Oh + o + d) = vh4;
                *reinterpret_cast<T4*>(Ol + o + d) = vl4;
            }
    }
#undef AS_FETCH
#undef AS_PUT
}
