// Shared helpers for libdccn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dccn.h"

namespace dccn {

extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) {
    g_last_hip_error = (int)e;
    return DCCN_ERR_LAUNCH;
}

#define DCCN_HIP(expr)                                         \
    do {                                                       \
        hipError_t e__ = (expr);                               \
        if (e__ != hipSuccess) return ::dccn::hip_fail(e__);   \
    } while (0)

#define DCCN_LAUNCH_CHECK()                                    \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return ::dccn::hip_fail(e__);   \
    } while (0)

#define DCCN_TRY(expr)                 \
    do {                               \
        int s__ = (expr);              \
        if (s__ != DCCN_OK) return s__; \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kCUs = 256;          // MI355X

// ---- wave / block reductions (wave64) -------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;   // valid in lane 0
}

}  // namespace dccn
