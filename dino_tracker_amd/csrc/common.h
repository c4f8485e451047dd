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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
// effective source count: min(M, *dM) when a device-side count is supplied
__device__ __forceinline__ int dtk_active(int M, const int32_t* dM) {
    if (dM == nullptr) return M;
    int v = *dM;
    return v < M ? v : M;
}
#endif
