// common.h -- shared helpers for libdtk (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dtk.h"

void dtk_set_error(const char* fmt, ...);

#define DTK_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            dtk_set_error(__VA_ARGS__);       \
            return DTK_E_INVALID;             \
        }                                     \
    } while (0)

#define DTK_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            dtk_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return DTK_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)

#define DTK_LAUNCHED() DTK_HIP(hipGetLastError())

// optional per-kernel timing (dtk_profile_* in dtk.h): hipEvents on the launch stream, off by default
void dtk_prof_begin(const char* name, hipStream_t st);
void dtk_prof_end(const char* name, hipStream_t st);
#define DTK_LAUNCH(NAME, kernel, grid, block, lds, st, ...)           \
    do {                                                              \
        dtk_prof_begin(NAME, st);                                     \
        hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__); \
        dtk_prof_end(NAME, st);                                       \
        DTK_LAUNCHED();                                               \
    } while (0)

// Development switches: a DTK_DEV build (make DEV=1) reads the DTK_DEBUG bit mask from the environment once; the
// production library has none of it (dtk_dev_flags() == 0 and every DTK_DBG test folds to false at compile time).
#ifdef DTK_DEV
#include <stdlib.h>
static inline int dtk_dev_flags() {
    static const int v = [] { const char* e = getenv("DTK_DEBUG"); return e ? atoi(e) : 0; }();
    return v;
}
#define DTK_DBG(flags, bit) ((flags) & (bit))
#else
static inline int dtk_dev_flags() { return 0; }
#define DTK_DBG(flags, bit) 0
#endif

static inline hipStream_t dtk_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int dtk_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
constexpr int WAVE = 64;

// Wave-wide all-reduce without an LDS round trip (the __shfl_xor butterfly compiles to six ds_bpermute_b32, ~100 cycles of
// latency each -- 90 of them made refine_head latency-bound): four DPP steps inside a row of 16 lanes (quad permutes, then
// the half-row and row mirrors pair up the partial sums), then v_permlane16_swap / v_permlane32_swap (gfx950) for the rows.
// Every lane ends with the result.  (The swap results are copied to scalars before the bit cast: hipcc / ROCm 7.2 reads
// element 0 for both when the cast is applied to the vector elements directly.)
template <int CTRL>
__device__ __forceinline__ int dtk_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <class Op>
__device__ __forceinline__ int wave_allreduce_bits(int v, Op op) {
    v = op(v, dtk_dpp<0xB1>(v));    // quad_perm [1,0,3,2]
    v = op(v, dtk_dpp<0x4E>(v));    // quad_perm [2,3,0,1]
    v = op(v, dtk_dpp<0x141>(v));   // row_half_mirror: lane i <-> 7 - i of each 8
    v = op(v, dtk_dpp<0x140>(v));   // row_mirror:      lane i <-> 15 - i of each 16
    {
        const auto sw = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        const unsigned a = sw[0], b = sw[1];
        v = op((int)a, (int)b);
    }
    {
        const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        const unsigned a = sw[0], b = sw[1];
        v = op((int)a, (int)b);
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    return __int_as_float(wave_allreduce_bits(__float_as_int(v), [](int a, int b) {
        return __float_as_int(__int_as_float(a) + __int_as_float(b));
    }));
}
__device__ __forceinline__ float wave_max(float v) {
    return __int_as_float(wave_allreduce_bits(__float_as_int(v), [](int a, int b) {
        return __float_as_int(fmaxf(__int_as_float(a), __int_as_float(b)));
    }));
}
__device__ __forceinline__ int wave_sum_i(int v) {
    return wave_allreduce_bits(v, [](int a, int b) { return a + b; });
}
// LDS-DMA through a buffer descriptor (round 4): global address = descriptor base + scalar byte offset (the tile: one SALU add
// per request) + 32-bit per-lane offset (a VGPR that is constant for the whole kernel); LDS destination = wave-uniform scalar +
// immediate (+ 16 bytes per lane, the hardware's lane-linear image).  Three issue slots per request; the global_load_lds form
// with a 64-bit per-lane pointer cost a lone wave ~13 (64-bit address arithmetic on the VALU per request, m0 saved and restored)
// -- 9 % of the attention kernel, and more of corr_peaks.  A kernel that uses it must not use m0 otherwise (no indirect register
// indexing, no GWS / sendmsg, no __builtin_amdgcn_global_load_lds in the same kernel): the compiler reserves m0 but is not told
// about the write.  (ADVICE r4 asked for "m0" in the clobber list: hipcc 7.2 answers "inline asm clobber list contains reserved
// registers: m0 ... may not be preserved across the asm statement" -- a reserved register in a clobber list is NOT honoured, so the
// declaration would promise nothing; the rule above is what holds, and the kernels that use this helper -- attention4, gemm_ws,
// corr_peaks V2D, refine_corr_dma -- contain no other m0 user: checked in the ISA, tests/test_abi.py::test_m0_users.)
typedef unsigned dtk_u4 __attribute__((ext_vector_type(4)));
template <int LDS_IMM>
__device__ __forceinline__ void dtk_buffer_lds16(dtk_u4 srd, unsigned soff, unsigned voff, unsigned lds_dst) {
    asm volatile(
        "s_add_u32 m0, %3, %4\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %0, %1, %2 offen lds"
        :
        : "v"(voff), "s"(srd), "s"(soff), "s"(lds_dst), "i"(LDS_IMM)
        : "memory", "scc");   // (s_add_u32 writes SCC: without the clobber the compiler keeps a compare result live across the asm --
                              //  round 4 found this as wrong tokens in ONE instantiation of the weight-stationary GEMM)
}
// raw buffer descriptor (stride 0) over `p`.  A raw buffer range-checks the per-lane offset (VGPR offset + immediate; the scalar
// offset is NOT checked) against num_records and returns zeros beyond it, so the size is the largest one -- every 32-bit
// per-lane offset is in range, callers clamp their own.  (Rounds 3-4 had 0x7fffffff here: refine_corr_dma_kernel, whose cell
// offset lives in the VGPR, would have read zeros for cells past 2 GB of the split planes -- C = 384 volumes of 173 .. 344
// frames at 854 x 476, which has_split_planes() admits; ADVICE r4.  tests/test_gpu_p3.py::test_split_planes_beyond_2gb.)
__device__ __forceinline__ dtk_u4 dtk_make_srd(const void* p) {
    const unsigned long long a = (unsigned long long)(size_t)p;
    dtk_u4 r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = __builtin_amdgcn_readfirstlane(r[k]);
    return r;
}

// Zero `bytes` (a multiple of 4, `p` 4-byte aligned) with a KERNEL.  For launches that may be captured into a graph (the training
// iteration, trainer.GraphedIteration): a hipMemsetAsync captured as a memset node was observed NOT to clear its range on replays
// on this stack (round 6: the contrastive backward's |G| maximum started from the previous replay's bits -- gradients of 1e28 --
// and ATen's multi-block top-k, which memsets its semaphores the same way, faulted on its second replay).
static __global__ void dtk_zero_words_kernel(unsigned* __restrict__ p, size_t words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t dtk_zero_async(void* p, size_t bytes, hipStream_t st) {
    const size_t words = (bytes + 3) / 4;
    if (words == 0) return hipSuccess;
    const size_t blocks = (words + 1023) / 1024;
    hipLaunchKernelGGL(dtk_zero_words_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st,
                       reinterpret_cast<unsigned*>(p), words);
    return hipGetLastError();
}

// effective source count: min(M, *dM) when a device-side count is supplied
__device__ __forceinline__ int dtk_active(int M, const int32_t* dM) {
    if (dM == nullptr) return M;
    int v = *dM;
    return v < M ? v : M;
}
#endif
