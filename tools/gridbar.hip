// Experiment: cost of a device-wide barrier inside one persistent launch on MI355X, against the launch boundary (~2.3 us + dispatch)
// that separates the stages of the 73-frame equaliser step.      hipcc --offload-arch=gfx950 -O3 tools/gridbar.hip -o abl/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        // one release fence, relaxed traffic on the counter, one acquire fence (an acquire LOAD in the poll loop invalidates the
        // L2 on every iteration: measured 11.3 us per barrier at 256 blocks)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// the same without contended atomics: every block raises its own flag, block 0 gathers them (one lane per flag) and raises `go`
__device__ __forceinline__ void flag_barrier(unsigned* flags, unsigned* go, unsigned phase) {
    __syncthreads();
    if (blockIdx.x == 0) {
        for (unsigned b = 1 + threadIdx.x; b < gridDim.x; b += 256)
            while (__hip_atomic_load(flags + 16 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            __hip_atomic_store(go, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(flags + 16 * blockIdx.x, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// no cache maintenance at all: the exchanged data is written with agent-scope (sc1, write-through) stores and read with agent-scope
// loads, so the barrier itself needs no buffer_wbl2 / buffer_inv -- only "my stores have left" (workgroup-scope release =
// s_waitcnt) in front of a relaxed counter
__device__ __forceinline__ void wt_barrier(unsigned* ctr, unsigned target) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// every phase: each thread writes phase-stamped data, barrier, then checks a slot written by a thread of ANOTHER block (another XCD)
template <int WORK, int FLAGS = 0>
__global__ __launch_bounds__(256) void persistent(unsigned* ctr, float* buf, int phases, int* errors) {
    const int n = gridDim.x * 256, me = blockIdx.x * 256 + threadIdx.x;
    int bad = 0;
    for (int p = 0; p < phases; ++p) {
        float v = (float)(p + 1);
        if (WORK) {
#pragma unroll 1
            for (int j = 0; j < WORK; ++j) v = v * 1.0000001f + 1e-9f;
        }
        if (FLAGS == 2) __hip_atomic_store(buf + me, (float)(p + 1) + 0.f * v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else buf[me] = (float)(p + 1) + 0.f * v;
        if (FLAGS == 2) wt_barrier(ctr, (unsigned)(p + 1) * gridDim.x);
        else if (FLAGS) flag_barrier(ctr + 64, ctr, 2 * p + 1);
        else grid_barrier(ctr, (unsigned)(p + 1) * gridDim.x);
        const int other = (me + 256 * (1 + (p % 7)) + 17) % n;
        const float got = FLAGS == 2 ? __hip_atomic_load(buf + other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                     : __builtin_nontemporal_load(buf + other);
        if (got != (float)(p + 1)) ++bad;
        if (FLAGS == 2) wt_barrier(ctr + 32, (unsigned)(p + 1) * gridDim.x);
        else if (FLAGS) flag_barrier(ctr + 64, ctr, 2 * p + 2);
        else grid_barrier(ctr + 32, (unsigned)(p + 1) * gridDim.x);  // (so that nobody overwrites before the readers are done)
    }
    if (bad) atomicAdd(errors, bad);
}

__global__ __launch_bounds__(256) void tiny(float* buf, int p) { buf[blockIdx.x * 256 + threadIdx.x] = (float)p; }

int main() {
    unsigned* ctr; float* buf; int* err;
    hipMalloc(&ctr, 65536); hipMalloc(&buf, 1024 * 256 * 4); hipMalloc(&err, 4);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int phases = 200;
    for (int blocks : {32, 64, 128, 256, 512}) {
        for (int rep = 0; rep < 2; ++rep) {
          for (int flags = 0; flags < 3; ++flags) {
            hipMemsetAsync(ctr, 0, 65536, s); hipMemsetAsync(err, 0, 4, s);
            hipEventRecord(a, s);
            if (flags == 2) hipLaunchKernelGGL((persistent<0, 2>), dim3(blocks), dim3(256), 0, s, ctr, buf, phases, err);
            else if (flags) hipLaunchKernelGGL((persistent<0, 1>), dim3(blocks), dim3(256), 0, s, ctr, buf, phases, err);
            else hipLaunchKernelGGL((persistent<0, 0>), dim3(blocks), dim3(256), 0, s, ctr, buf, phases, err);
            hipEventRecord(b, s);
            hipStreamSynchronize(s);
            float ms; hipEventElapsedTime(&ms, a, b);
            int e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
            if (rep) printf("persistent %s blocks %4d: %.3f us per barrier (two barriers + exchange per phase: %.3f us), errors %d\n",
                            flags == 2 ? "wt     " : flags ? "flags  " : "counter", blocks, ms * 1e3 / (2 * phases), ms * 1e3 / phases, e);
          }
        }
    }
    for (int blocks : {64, 256}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a, s);
            for (int p = 0; p < 2 * phases; ++p) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s, buf, p);
            hipEventRecord(b, s);
            hipStreamSynchronize(s);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("launches    blocks %4d: %.3f us per launch of an empty-ish kernel\n", blocks, ms * 1e3 / (2 * phases));
        }
    }
    return 0;
}
