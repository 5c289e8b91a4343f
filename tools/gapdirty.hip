// Experiment: what does the boundary between two dependent launches on one stream cost on MI355X, as a function of the bytes the
// first kernel leaves dirty in the L2s, and do write-through stores move that cost into the kernel?
//      hipcc --offload-arch=gfx950 -O3 tools/gapdirty.hip -o abl/gapdirty && abl/gapdirty
// Per (store mode, MB written): median over iterations of
//      gap    = first workgroup of the reader in  -  last workgroup of the writer out   (s_memrealtime, 100 MHz)
//      writer = first workgroup in -> last workgroup out
// The writer spins ~6 us before / between its stores (real kernels compute while they store; a 0.5-us kernel measures the command
// processor's packet rate, 2.5 us per launch, not the boundary).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
struct BigArgs { float v[120]; };

template <int MODE>
__device__ __forceinline__ void store4(f4* p, f4 v) {
    if (MODE == 0) *p = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, p);
    else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off nt sc1" ::"v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void writer(f4* buf, size_t n4, unsigned long long* st, float seed, int spin) {
    unsigned long long t0 = wall_clock64();
    const size_t stride = (size_t)gridDim.x * 256;
    float w = seed;
    int chunk = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        for (int j = 0; j < spin; ++j) w = w * 1.0000001f + 1e-9f;      // compute between the stores
        f4 v = {w, seed + 1.f, seed + 2.f, (float)i};
        store4<MODE>(buf + i, v);
        ++chunk;
    }
    if (n4 == 0 || chunk == 0) for (int j = 0; j < spin * 8; ++j) w = w * 1.0000001f + 1e-9f;
    if (w == 123.456f) buf[0].x = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        st[2 * blockIdx.x] = t0;
        st[2 * blockIdx.x + 1] = wall_clock64();
    }
}

template <int BIG>
__global__ __launch_bounds__(256) void reader(const f4* buf, size_t n4, unsigned long long* st, float* sink, int touch, BigArgs big) {
    unsigned long long t0 = wall_clock64();
    extern __shared__ float lds[];
    float acc = 0.f;
    if (BIG) { lds[threadIdx.x] = big.v[threadIdx.x % 120]; __syncthreads(); acc = lds[(threadIdx.x + 1) % 256]; }
    if (touch) {
        const size_t stride = (size_t)gridDim.x * 256;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) acc += buf[i].x;
    }
    if (acc == 12345.678f) sink[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        st[2 * blockIdx.x] = t0;
        st[2 * blockIdx.x + 1] = wall_clock64();
    }
}

static double med(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    const int blocks = 512, iters = 60;
    const size_t maxb = 64u << 20;
    f4* buf; float* sink; unsigned long long *sw, *sr;
    (void)hipMalloc(&buf, maxb); (void)hipMalloc(&sink, 64);
    (void)hipMalloc(&sw, (size_t)iters * blocks * 16); (void)hipMalloc(&sr, (size_t)iters * blocks * 16);
    hipStream_t s; (void)hipStreamCreate(&s);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(reader<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    std::vector<unsigned long long> hw((size_t)iters * blocks * 2), hr((size_t)iters * blocks * 2);
    BigArgs big; for (int i = 0; i < 120; ++i) big.v[i] = (float)i;
    const double mbs[] = {0.0, 1, 4, 8, 16};
    const char* names[] = {"plain", "nontemporal", "sc1", "sc0 sc1", "nt sc1"};
    for (int bigr = 0; bigr < 2; ++bigr)
    for (int mode = 0; mode < 5; ++mode) {
        for (double mb : mbs) {
            const size_t n4 = (size_t)(mb * 1048576.0 / 16.0);
            // ~6 us of arithmetic per thread in all: spread over the thread's stores
            const int per_thread = n4 ? (int)((n4 + (size_t)blocks * 256 - 1) / ((size_t)blocks * 256)) : 1;
            const int spin = n4 ? 1600 / per_thread : 200;
            for (int rep = 0; rep < 2; ++rep) {
                for (int it = 0; it < iters; ++it) {
                    unsigned long long* a = sw + (size_t)it * blocks * 2; unsigned long long* b = sr + (size_t)it * blocks * 2;
                    switch (mode) {
                        case 0: hipLaunchKernelGGL(writer<0>, dim3(blocks), dim3(256), 0, s, buf, n4, a, (float)it, spin); break;
                        case 1: hipLaunchKernelGGL(writer<1>, dim3(blocks), dim3(256), 0, s, buf, n4, a, (float)it, spin); break;
                        case 2: hipLaunchKernelGGL(writer<2>, dim3(blocks), dim3(256), 0, s, buf, n4, a, (float)it, spin); break;
                        case 3: hipLaunchKernelGGL(writer<3>, dim3(blocks), dim3(256), 0, s, buf, n4, a, (float)it, spin); break;
                        default: hipLaunchKernelGGL(writer<4>, dim3(blocks), dim3(256), 0, s, buf, n4, a, (float)it, spin); break;
                    }
                    if (bigr) hipLaunchKernelGGL(reader<1>, dim3(blocks), dim3(256), 65536, s, buf, n4, b, sink, 1, big);
                    else hipLaunchKernelGGL(reader<0>, dim3(blocks), dim3(256), 0, s, buf, n4, b, sink, 1, big);
                }
                (void)hipStreamSynchronize(s);
            }
            (void)hipMemcpy(hw.data(), sw, hw.size() * 8, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hr.data(), sr, hr.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> gap, wdur, rdur, gap2;
            unsigned long long prev_r_out = 0;
            for (int it = 0; it < iters; ++it) {
                unsigned long long win = ~0ull, wout = 0, rin = ~0ull, rout = 0;
                for (int b = 0; b < blocks; ++b) {
                    win = std::min(win, hw[((size_t)it * blocks + b) * 2]); wout = std::max(wout, hw[((size_t)it * blocks + b) * 2 + 1]);
                    rin = std::min(rin, hr[((size_t)it * blocks + b) * 2]); rout = std::max(rout, hr[((size_t)it * blocks + b) * 2 + 1]);
                }
                if (it >= 10) {
                    gap.push_back((double)(rin - wout) * 0.01); wdur.push_back((double)(wout - win) * 0.01); rdur.push_back((double)(rout - rin) * 0.01);
                    gap2.push_back((double)(win - prev_r_out) * 0.01);
                }
                prev_r_out = rout;
            }
            printf("reader %s | stores %-12s %6.2f MB : writer %6.2f us  gap->reader %5.2f us  reader %6.2f us  gap->next writer %5.2f us  (sum %6.2f)\n",
                   bigr ? "64KB LDS + 480B args" : "small               ", names[mode], mb, med(wdur), med(gap), med(rdur), med(gap2),
                   med(wdur) + med(gap) + med(rdur) + med(gap2));
        }
    }
    return 0;
}
