// Shared helpers for libdccn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <map>
#include <utility>
#include <type_traits>
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
// compute units of the current device (MI355X: 256; fewer in the partitioned CPX/DPX modes): asked once PER DEVICE
// (a process may drive several), 256 when no device is visible (workspace-size queries on a host without a GPU).
// Host-side planning only.
static inline int device_cus() {
    constexpr int kMaxDev = 64;
    static std::atomic<int> cache[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (device, kernel): kernels that need more dynamic LDS than
// the default get it set once per device they are launched on (thread safe; a lookup per launch afterwards).
static inline int set_max_dynamic_smem(const void* kern, size_t bytes) {
    if (bytes <= 48 * 1024) return DCCN_OK;
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> done;      // largest size granted so far
    int dev = 0;
    DCCN_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = done[{dev, kern}];
    if (have >= bytes) return DCCN_OK;
    DCCN_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
    return DCCN_OK;
}
#define kCUs (dccn::device_cus())

// ---- step timeline stamps (dccn_step_trace_*: in-situ start/end of every instrumented launch, bench.py `step.boundaries`) ---
// A launch whose parameter block carries a non-null `stamp` pointer leaves, per workgroup (blockIdx.x < kStampBlocks, y = z = 0),
// four 64-bit words: {s_memrealtime at entry, s_memtime at entry, s_memrealtime at exit, s_memtime at exit} written by thread 0.
// s_memrealtime ticks at a constant 100 MHz, s_memtime once per shader cycle: the ratio is the clock the CU really ran at.
// Null pointer (every normal launch): one scalar compare and a branch per mark.
constexpr int kStampBlocks = 4096, kStampWords = 4, kStampLaunches = 8;
__device__ __forceinline__ void stamp_mark(unsigned long long* st, const int which) {
    if (st != nullptr && threadIdx.x == 0 && blockIdx.x < (unsigned)kStampBlocks && blockIdx.y == 0 && blockIdx.z == 0) {
        unsigned long long* p = st + (size_t)blockIdx.x * kStampWords + 2 * which;
        p[0] = wall_clock64();
        p[1] = clock64();
    }
}
// host side: the trace buffer is a ring of steps x kStampLaunches launches x kStampBlocks blocks x kStampWords words; a step
// implementation opens a StepTraceScope and names the launch that follows (`launch(i)`); every parameter block initialised
// while that is in force (gp_zero() / the optimizer's argument block) picks the pointer up.
struct StepTraceState {
    std::atomic<unsigned long long*> buf{nullptr};
    std::atomic<int> ring{0};
    std::atomic<long long> step{0};
};
extern StepTraceState g_step_trace;
extern thread_local unsigned long long* tl_stamp;
struct StepTraceScope {
    unsigned long long* base;
    StepTraceScope() : base(nullptr) {
        unsigned long long* b = g_step_trace.buf.load(std::memory_order_relaxed);
        const int ring = g_step_trace.ring.load(std::memory_order_relaxed);
        if (b != nullptr && ring > 0) {
            const long long st = g_step_trace.step.fetch_add(1, std::memory_order_relaxed);
            base = b + (size_t)(st % ring) * kStampLaunches * kStampBlocks * kStampWords;
        }
    }
    ~StepTraceScope() { tl_stamp = nullptr; }
    void launch(int i) const { tl_stamp = (base && i >= 0 && i < kStampLaunches) ? base + (size_t)i * kStampBlocks * kStampWords : nullptr; }
    void none() const { tl_stamp = nullptr; }
};

// ---- chain groups: G shape-identical, independent training chains carried by ONE launch sequence ---------------------------
// (the reference driver starts such chains as OS processes: dev/py/run_local_ofdm.py:61-118, one per modulation / variant.)
// Every buffer of chain g lies at the SAME offset inside that chain's arena, so a launch planned for chain 0's pointers serves
// chain g after adding ONE byte offset (arena_g - arena_0) to every pointer argument: the chain index is a grid dimension
// (blockIdx.z; blockIdx.y where z is taken), the offsets travel by value in the kernel arguments.  A converted kernel takes a
// trailing `ChainOffs`; outside a grouped call the table is {0} and the grid dimension is 1: same blocks, same arithmetic.
constexpr int kMaxChains = 8;
struct ChainOffs {
    long long off[kMaxChains];          // bytes
};
struct ChainCtx {
    int G;                              // chains carried by the launches issued from this thread right now (1 = no group)
    ChainOffs co;
    int nbits[kMaxChains];              // per-chain modulation (the frozen receiver's tail is the only nbits-dependent stage)
    // the fused generator's per-chain scalars, when its workgroups ride on a launch of the group's step (eq_step.h)
    unsigned long long gen_seed[kMaxChains];
    unsigned gen_offset[kMaxChains];
    int gen_nbits[kMaxChains];
};
extern thread_local ChainCtx tl_chain;
struct ChainScope {                     // RAII: a (sub-)group for the launches of a scope
    ChainCtx saved;
    explicit ChainScope(const ChainCtx& c) : saved(tl_chain) { tl_chain = c; }
    ~ChainScope() { tl_chain = saved; }
};
// (pointer arithmetic on char*, never a round trip through an integer: the compiler must keep seeing a GLOBAL address -- after
// an inttoptr it falls back to flat_load / flat_store, which wait on the LDS counter too and cost the step 20 %)
template <typename T>
__device__ __forceinline__ T* chain_at(T* p, const long long off) {
    using U = typename std::remove_const<T>::type;
    return p == nullptr ? p : reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<U*>(p)) + off);
}
// launch with the chain index on grid dimension z / y; the kernel's LAST parameter is the ChainOffs table
#define DCCN_LAUNCH_CHAINS_Z(kern, grid, block, smem, s, ...)                                        \
    do {                                                                                               \
        dim3 g__ = (grid);                                                                             \
        g__.z = (unsigned)::dccn::tl_chain.G;                                                          \
        hipLaunchKernelGGL(kern, g__, block, smem, s, __VA_ARGS__, ::dccn::tl_chain.co);               \
    } while (0)
#define DCCN_LAUNCH_CHAINS_Y(kern, grid, block, smem, s, ...)                                        \
    do {                                                                                               \
        dim3 g__ = (grid);                                                                             \
        g__.y = (unsigned)::dccn::tl_chain.G;                                                          \
        hipLaunchKernelGGL(kern, g__, block, smem, s, __VA_ARGS__, ::dccn::tl_chain.co);               \
    } while (0)
// a launch site that has NOT been converted refuses to run inside a group (it would serve chain 0 only)
#define DCCN_NO_CHAINS()                                               \
    do {                                                               \
        if (::dccn::tl_chain.G != 1 || ::dccn::tl_chain.co.off[0] != 0) return DCCN_ERR_UNSUPPORTED; \
    } while (0)

// ---- output stores of a step's launches: plain (write-back) or agent-scope (`sc1`: written through the XCD's L2) -------------
// A launch boundary drains every L2 of its dirty lines before the next launch starts: tools/gapdirty.hip measures the boundary
// at 1.05 us + 0.12 us per MB the finished kernel left dirty (profiles/r06_gapdirty.txt).  A stream written with agent-scope
// stores is written through while the kernel still computes and leaves nothing to drain.  DCCN_WT_STORES = bit mask of the
// streams written that way (build option; the stream index is the template argument):
//   0 C-Conv forward output   1 dz of the fused dense + tail   2 dense dW slabs, dWeff partials, column sums of the backward
//   3 x_norm of the optimizer launch's R0 blocks   4 parameters / Adam slots of the optimizer launch
//   5 y, noise and label bits of the fused static-channel generator (datagen.h)
// Measured per stream on the C2 step (tools/wtscan.sh, in-situ timeline, profiles/r06_wt_stores.txt): 0: boundary in front of the
// dense launch 1.73 -> 1.36 us for +0.1 us of C-Conv forward; 2: boundary in front of the optimizer launch 2.37 -> 1.50 us, the
// backward launch unchanged (its blocks end at different times: the write-through overlaps the launch's own tail); 1: the
// boundary gains 0.36 us but the dense launch loses 1.3 (every block stores at the very end); 3 / 4: the boundary in front of
// the next C-Conv forward 2.5 -> 1.6-2.0 us but the bandwidth-bound optimizer launch loses 0.4-1.0; 5: the generate-and-train
// loop 0.0987 -> 0.0962 ms per batch (13 MB less to drain in front of the step's first launch).  Shipped: streams 0, 2 and 5
// (step 72.2 -> 71.4 us); the values stored are the same either way.
#ifndef DCCN_WT_STORES
#define DCCN_WT_STORES 37
#endif
template <int STREAM>
__device__ __forceinline__ void out_store(float* p, const float v) {
    if constexpr ((DCCN_WT_STORES >> STREAM) & 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <int STREAM>
__device__ __forceinline__ void out_store(int* p, const int v) {
    if constexpr ((DCCN_WT_STORES >> STREAM) & 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <int STREAM>
__device__ __forceinline__ void out_store2(float* p, const float x, const float y) {      // p 8-byte aligned
    if constexpr ((DCCN_WT_STORES >> STREAM) & 1) {
        const unsigned long long b = ((unsigned long long)__builtin_bit_cast(unsigned, y) << 32) | __builtin_bit_cast(unsigned, x);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *reinterpret_cast<float2*>(p) = make_float2(x, y);
    }
}
template <int STREAM>
__device__ __forceinline__ void out_store4(float* p, const float x, const float y, const float z, const float w) {   // 16-byte aligned
    if constexpr ((DCCN_WT_STORES >> STREAM) & 1) {
        typedef float wt_f32x4 __attribute__((ext_vector_type(4)));
        const wt_f32x4 v = {x, y, z, w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    } else {
        *reinterpret_cast<float4*>(p) = make_float4(x, y, z, w);
    }
}

// ---- wave reductions (wave64) on the DPP crossbar: no LDS round trips ------------------------
// quad_perm[1,0,3,2] -> quad_perm[2,3,0,1] -> row_half_mirror -> row_mirror sums each 16-lane row
// (fixed order => deterministic), then the four row sums are combined through v_readlane.
// The result is returned in every lane.
__device__ __forceinline__ int dpp_mov_i32(int v, const int ctrl_sel) {
    switch (ctrl_sel) {
        case 0: return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm:[1,0,3,2]
        case 1: return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm:[2,3,0,1]
        case 2: return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
        default: return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);  // row_mirror
    }
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int s = 0; s < 4; ++s) v += __builtin_bit_cast(float, dpp_mov_i32(__builtin_bit_cast(int, v), s));
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int s = 0; s < 4; ++s) v += dpp_mov_i32(v, s);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
           (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const long long b = __builtin_bit_cast(long long, v);
        const int lo = dpp_mov_i32((int)(b & 0xffffffffLL), s), hi = dpp_mov_i32((int)(b >> 32), s);
        v += __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
    }
    double r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long b = __builtin_bit_cast(long long, v);
        const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 16 * q);
        const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 16 * q);
        r[q] = __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
    }
    return (r[0] + r[1]) + (r[2] + r[3]);
}
__device__ __forceinline__ long long wave_sum(long long v) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int lo = dpp_mov_i32((int)(v & 0xffffffffLL), s), hi = dpp_mov_i32((int)(v >> 32), s);
        v += ((long long)hi << 32) | (unsigned)lo;
    }
    long long r = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffLL), 16 * q);
        const int hi = __builtin_amdgcn_readlane((int)(v >> 32), 16 * q);
        r += ((long long)hi << 32) | (unsigned)lo;
    }
    return r;
}

}  // namespace dccn
