// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32): exact f32 products and a
// k-ascending fmaf accumulation chain per output element, so a launch without split-K is
// bitwise deterministic and bitwise reproducible by `for k: acc = fmaf(a,b,acc)`.
//
// C[M,N] = A[M,K] . B[K,N]  with both operands staged through LDS as [k][i] tiles
// (k-major, i contiguous), so one MFMA operand fetch is a conflict-free ds_read_b32 of 32
// consecutive floats per half-wave.  How a tile gets from HBM into that layout is the
// operand "kind":
//   OP_ICONTIG   element (k,i) at p[k*ld+i]  -> float4 along i, ds_write_b128
//   OP_KCONTIG   element (k,i) at p[i*ld+k]  -> float4 along k, 4 transposed ds_write_b32
//                (LDS row stride BI+1 makes those writes conflict-free)
//   OP_CCONV_W   virtual expanded complex-conv weights Weff[2kin,2F] built on the fly from
//                w[kin,2F] = [Wa|Wb] (dev/py/complex.py:185-188):
//                    Weff[2n  ,2f] =  Wa[n,f]   Weff[2n  ,2f+1] =  Wb[n,f]
//                    Weff[2n+1,2f] = -Wb[n,f]   Weff[2n+1,2f+1] = -Wa[n,f]
//                so that x[rows,(n,iq)] . Weff = out[rows,(f,re/im)] : the four real
//                sub-convolutions of the C-Conv are ONE GEMM with interleaved IQ in/out.
//   OP_CCONV_WT  Weff transposed (for dX = dOut . Weff^T)
//
// Block = 256 threads = 4 waves in a 2x2 grid; wave tile (BM/2)x(BN/2) made of 32x32 MFMA
// tiles; BK = 32; register-staged double buffering with one barrier per k-tile.
#pragma once
#include <type_traits>
#include "common.h"

namespace dccn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum OperandKind : int { OP_ICONTIG = 0, OP_KCONTIG = 1, OP_CCONV_W = 2, OP_CCONV_WT = 3 };
enum GemmTag : int { TAG_DENSE_FWD = 0, TAG_DENSE_BWD_X = 1, TAG_DENSE_BWD_W = 2, TAG_CCONV_FWD = 3, TAG_CCONV_BWD_X = 4,
                     TAG_CCONV_BWD_W = 5 };

struct GemmParams {
    const float* A;
    const float* B;
    float* C;
    const float* bias;   // [N], epilogue add (only when splits == 1), nullable
    float* colsum;       // [splits][N] partial column sums of the B rows, nullable
    int M, N, K;
    int lda, ldb, ldc;
    int klen;            // K range per split (multiple of 32)
    long long slab;      // elements between the split slabs of C
    int cF;              // F of the cconv operand kinds
    int vecA, vecB;      // vector (16 B) global loads legal for the operand
    int cbias;           // 1: bias is the C-Conv [ba|bb] pair -> col 2f: ba-bb, col 2f+1: bb-ba
};

constexpr int kBKmin = 32;      // K ranges (klen) are multiples of this

__device__ __forceinline__ float cconv_weff(const float* __restrict__ w, int F, int row, int col) {
    const int n = row >> 1, q = row & 1, f = col >> 1, c = col & 1;
    const float v = w[(size_t)n * 2 * F + ((q ^ c) ? F : 0) + f];
    return q ? -v : v;
}

template <int KIND, int BI, int BK, int THREADS>
struct Tile {
    static constexpr int NV = BK * BI / 4 / THREADS;   // 4-element pieces per thread per k-tile
    static constexpr bool KV = (KIND == OP_KCONTIG || KIND == OP_CCONV_WT);
    static constexpr int LD = KV ? BI + 1 : BI;
    float4 r[NV];
    unsigned okmask;     // bit v: piece v lies inside the k range (applied when it is written to LDS,
                         // so the global load itself has no consumer until then and stays in flight)

    // Piece v of the (k0..kend) x (i0..I) window.  Out-of-range k must read as 0 (it enters the sums);
    // out-of-range i only feeds outputs that are never stored.  Every access is issued unconditionally
    // from a clamped in-range address (no divergent control flow).  `vec` (block-uniform, decided on
    // the host, a template parameter so the loop stays straight-line) selects one 16-byte load per piece: it needs ld % 4 == 0, a 16-byte aligned base and
    // (ICONTIG) I % 4 == 0 / (KCONTIG) K % 4 == 0.
    template <bool VEC>
    __device__ __forceinline__ void load_piece(int v, const float* __restrict__ p, int ld, int k0, int kend, int K,
                                               int i0, int I, int cF, int tid) {
        const int idx = tid + v * THREADS;
        if constexpr (!KV) {
            const int k = k0 + idx / (BI / 4);
            const int i = i0 + (idx % (BI / 4)) * 4;
            const int kc = min(k, K - 1);
            bool vdone = false;
            if constexpr (KIND == OP_ICONTIG && VEC) {
                r[v] = *reinterpret_cast<const float4*>(p + (size_t)kc * ld + min(i, I - 4));
                vdone = true;
            }
            if (!vdone) {
                float e[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ic = min(i + q, I - 1);
                    float t;
                    if constexpr (KIND == OP_ICONTIG) t = p[(size_t)kc * ld + ic];
                    else t = cconv_weff(p, cF, kc, ic);
                    e[q] = (i + q < I) ? t : 0.f;
                }
                r[v] = make_float4(e[0], e[1], e[2], e[3]);
            }
            okmask = (okmask & ~(1u << v)) | ((k < kend ? 1u : 0u) << v);
        } else {
            const int i = i0 + idx / (BK / 4);
            const int k = k0 + (idx % (BK / 4)) * 4;
            const int ic = min(i, I - 1);
            bool vdone = false;
            if constexpr (KIND == OP_KCONTIG && VEC) {
                r[v] = *reinterpret_cast<const float4*>(p + (size_t)ic * ld + min(k, K - 4));
                okmask = (okmask & ~(1u << v)) | ((k < kend ? 1u : 0u) << v);
                vdone = true;
            }
            if (!vdone) {
                float e[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kc = min(k + q, K - 1);
                    float t;
                    if constexpr (KIND == OP_KCONTIG) t = p[(size_t)ic * ld + kc];
                    else t = cconv_weff(p, cF, ic, kc);
                    e[q] = (k + q < kend) ? t : 0.f;
                }
                r[v] = make_float4(e[0], e[1], e[2], e[3]);
                okmask |= 1u << v;
            }
        }
    }

    __device__ __forceinline__ void store_piece(int v, float* __restrict__ lds, int tid) const {
        const int idx = tid + v * THREADS;
        const bool ok = (okmask >> v) & 1u;
        const float4 val = make_float4(ok ? r[v].x : 0.f, ok ? r[v].y : 0.f, ok ? r[v].z : 0.f, ok ? r[v].w : 0.f);
        if constexpr (!KV) {
            const int k = idx / (BI / 4);
            const int i = (idx % (BI / 4)) * 4;
            *reinterpret_cast<float4*>(lds + k * LD + i) = val;
        } else {
            const int i = idx / (BK / 4);
            const int k = (idx % (BK / 4)) * 4;
            lds[(k + 0) * LD + i] = val.x;
            lds[(k + 1) * LD + i] = val.y;
            lds[(k + 2) * LD + i] = val.z;
            lds[(k + 3) * LD + i] = val.w;
        }
    }
};

template <int KA, int KB, int BM, int BN, int BK>
constexpr size_t gemm_smem_bytes() {
    return (size_t)(2 * BK * Tile<KA, BM, BK, 256>::LD + 2 * BK * Tile<KB, BN, BK, 256>::LD) * sizeof(float);
}

// TAG only makes the symbol unique per call site so profiles attribute time to the right operator
// THREADS = 256: four waves, one per SIMD.  THREADS = 512: a second group of four waves shares the
// same LDS tiles and takes the other half of every k-tile (in-block split-K, summed through LDS at the
// end): each SIMD then holds two independent MFMA chains, so one wave's barrier / LDS / global-load
// waits are covered by the other's matrix work.
template <int KA, int KB, int BM, int BN, int BK, int COLSUM, int TAG, bool VEC, int THREADS>
__global__ __launch_bounds__(THREADS) void gemm_f32_mfma_kernel(const GemmParams p) {
    constexpr int kBK = BK;
    constexpr int KG = THREADS / 256;
    using TA = Tile<KA, BM, BK, THREADS>;
    using TB = Tile<KB, BN, BK, THREADS>;
    constexpr int LDA = TA::LD, LDB = TB::LD;
    constexpr int TM = BM / 64, TN = BN / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                       // [2][kBK][LDA]
    float* sB = smem + 2 * kBK * LDA;       // [2][kBK][LDB]   (2*32*LDA is a multiple of 4)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int kg = wid >> 2;                               // k-group of this wave (0 when THREADS == 256)
    const int wm0 = ((wid & 3) >> 1) * (BM / 2), wn0 = (wid & 1) * (BN / 2);
    // XCD-aware tile order: the dispatcher places block b on XCD b % 8 (speed only, never
    // correctness); give each XCD a contiguous run of row-major tiles so its private L2 holds one
    // slice of A rows plus the B panel instead of everything.
    const int ntn = (p.N + BN - 1) / BN;
    int tile;
    {
        const int T = gridDim.x, L = blockIdx.x;
        const int xcd = L & 7, j = L >> 3, q = T >> 3, r = T & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int kbeg = blockIdx.z * p.klen;
    const int kend = min(p.K, kbeg + p.klen);
    const int ntiles = (kend - kbeg + kBK - 1) / kBK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float cs = 0.f;   // column sum of the B rows (COLSUM)
    const bool do_cs = COLSUM && p.colsum != nullptr && m0 == 0 && tid < BN;

    TA ta;
    TB tb;
    ta.okmask = 0u;
    tb.okmask = 0u;
    constexpr int PA = TA::NV, PB = TB::NV, NK = kBK / 2 / KG;      // MFMA steps per wave per k-tile
    static_assert(PA >= 1 && PB >= 1 && NK >= 2 * (PA + PB), "k-tile too shallow for the load/store slots");
    const int kofs = kg * NK;                                       // first k-step of this wave's group
    if (ntiles > 0) {
#pragma unroll
        for (int v = 0; v < PA; ++v) ta.template load_piece<VEC>(v, p.A, p.lda, kbeg, kend, p.K, m0, p.M, p.cF, tid);
#pragma unroll
        for (int v = 0; v < PB; ++v) tb.template load_piece<VEC>(v, p.B, p.ldb, kbeg, kend, p.K, n0, p.N, p.cF, tid);
#pragma unroll
        for (int v = 0; v < PA; ++v) ta.store_piece(v, sA, tid);
#pragma unroll
        for (int v = 0; v < PB; ++v) tb.store_piece(v, sB, tid);
    }
    __syncthreads();

    // One k-tile = NK dependent MFMA steps (64 cycles each per accumulator).  Everything else the wave
    // has to do for the pipeline is issued INSIDE that chain, in the shadow of the matrix pipe:
    //   - operand fragments are read from LDS two steps ahead (3-slot register ring),
    //   - the global loads of the next k-tile go out during the first PA+PB steps,
    //   - their LDS writes (other buffer) happen during the last PA+PB steps, ~NK*64 cycles later.
    // Only the barrier and the first two fragment reads of a tile are exposed.
    // (MORE = false for the last k-tile: no prefetch work, so the steady-state body has no branches and the
    // compiler keeps counted vmcnt/lgkmcnt waits instead of draining at control-flow joins.)
    int t = 0;
    auto ktile = [&](auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        const int cur = t & 1;
        const int k0n = kbeg + (t + 1) * kBK;
        const float* As = sA + cur * kBK * LDA + wm0 + l31;
        const float* Bs = sB + cur * kBK * LDB + wn0 + l31;
        float* An = sA + (cur ^ 1) * kBK * LDA;
        float* Bn = sB + (cur ^ 1) * kBK * LDB;
        float fa[3][TM], fb[3][TN];
#pragma unroll
        for (int pre = 0; pre < 2; ++pre) {
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[pre][a] = As[(2 * (kofs + pre) + h) * LDA + a * 32];
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[pre][b] = Bs[(2 * (kofs + pre) + h) * LDB + b * 32];
        }
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            if (kk + 2 < NK) {
#pragma unroll
                for (int a = 0; a < TM; ++a) fa[(kk + 2) % 3][a] = As[(2 * (kofs + kk + 2) + h) * LDA + a * 32];
#pragma unroll
                for (int b = 0; b < TN; ++b) fb[(kk + 2) % 3][b] = Bs[(2 * (kofs + kk + 2) + h) * LDB + b * 32];
            }
            if constexpr (MORE) {
                if (kk < PA) ta.template load_piece<VEC>(kk, p.A, p.lda, k0n, kend, p.K, m0, p.M, p.cF, tid);
                else if (kk < PA + PB) tb.template load_piece<VEC>(kk - PA, p.B, p.ldb, k0n, kend, p.K, n0, p.N, p.cF, tid);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk % 3][a], fb[kk % 3][b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MORE) {
                constexpr int S0 = NK - PA - PB;
                if (kk >= S0 && kk < S0 + PA) ta.store_piece(kk - S0, An, tid);
                else if (kk >= S0 + PA) tb.store_piece(kk - S0 - PA, Bn, tid);
            }
        }
        if (COLSUM && do_cs) {
            const float* Bc = sB + cur * kBK * LDB + tid;
#pragma unroll 8
            for (int k = 0; k < kBK; ++k) cs += Bc[k * LDB];
        }
        __syncthreads();
    };
    for (; t + 1 < ntiles; ++t) ktile(std::true_type{});
    if (ntiles > 0) ktile(std::false_type{});

    if constexpr (KG == 2) {
        // sum the two k-groups: group 1 parks its accumulators in LDS (the tile buffers are free after the
        // loop's last barrier), group 0 adds them in a fixed order and owns the epilogue
        float* xch = smem + ((wid & 3) * TM * TN * 16) * 64 + lane;
        if (kg == 1) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xch[((a * TN + b) * 16 + r) * 64] = acc[a][b][r];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] += xch[((a * TN + b) * 16 + r) * 64];
    }
    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cz = p.C + (size_t)blockIdx.z * p.slab;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = n0 + wn0 + b * 32 + l31;
            float bj = 0.f;
            if (p.bias != nullptr && col < p.N) {
                if (p.cbias) {
                    const float d = p.bias[col >> 1] - p.bias[p.cF + (col >> 1)];
                    bj = (col & 1) ? -d : d;
                } else {
                    bj = p.bias[col];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < p.M && col < p.N) Cz[(size_t)row * p.ldc + col] = acc[a][b][r] + bj;
            }
        }
    }
    if (COLSUM && do_cs) {
        const int col = n0 + tid;
        if (col < p.N) p.colsum[(size_t)blockIdx.z * p.N + col] = cs;
    }
}

template <int KA, int KB, int BM, int BN, int BK, int COLSUM, int TAG, bool VEC>
static int launch_gemm_cfg2(const GemmParams& p, int splits, hipStream_t s) {
    // two k-groups (8 waves) whenever a k-tile leaves each group enough MFMA steps for its load/store slots
    constexpr int THREADS = (BM == 64 && BN == 64 && BK == 64) ? 512 : 256;
    auto kern = gemm_f32_mfma_kernel<KA, KB, BM, BN, BK, COLSUM, TAG, VEC, THREADS>;
    constexpr size_t smem = gemm_smem_bytes<KA, KB, BM, BN, BK>();
    static bool attr_done = false;
    if (!attr_done) {
        if (smem > 48 * 1024)
            DCCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_done = true;
    }
    dim3 grid(ceil_div(p.N, BN) * ceil_div(p.M, BM), 1, splits);
    hipLaunchKernelGGL(kern, grid, dim3(THREADS), smem, s, p);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

template <int KA, int KB, int BM, int BN, int BK, int COLSUM, int TAG>
static int launch_gemm_cfg(const GemmParams& p, int splits, hipStream_t s) {
    constexpr bool needA = (KA == OP_ICONTIG || KA == OP_KCONTIG), needB = (KB == OP_ICONTIG || KB == OP_KCONTIG);
    const bool vec = (!needA || p.vecA) && (!needB || p.vecB);
    if (vec) return launch_gemm_cfg2<KA, KB, BM, BN, BK, COLSUM, TAG, true>(p, splits, s);
    return launch_gemm_cfg2<KA, KB, BM, BN, BK, COLSUM, TAG, false>(p, splits, s);
}

// tile choice: 128x128 only when it still yields >= 2 blocks per CU, else 64x64;
// k-tile depth 64 when the K range is long enough to amortise it (fewer barriers, longer MFMA
// runs to cover the global-load latency of the next tile), else 32
template <int KA, int KB, int COLSUM, int TAG>
static int launch_gemm(const GemmParams& p, int splits, hipStream_t s) {
    const long long big = (long long)ceil_div(p.M, 128) * ceil_div(p.N, 128) * splits;
    const int krange = p.klen < p.K ? p.klen : p.K;
    if (big >= 2 * kCUs) return launch_gemm_cfg<KA, KB, 128, 128, 32, COLSUM, TAG>(p, splits, s);
    if (krange >= 256 && krange % 64 == 0) return launch_gemm_cfg<KA, KB, 64, 64, 64, COLSUM, TAG>(p, splits, s);
    return launch_gemm_cfg<KA, KB, 64, 64, 32, COLSUM, TAG>(p, splits, s);
}

// split-K plan for the weight-gradient GEMMs (K = batch rows is the long axis)
struct SplitPlan {
    int splits, klen;
};
static inline SplitPlan plan_splitk(int M, int N, int K) {
    const long long tiles = (long long)ceil_div(M, 64) * ceil_div(N, 64);
    long long want = (2LL * kCUs + tiles - 1) / tiles;          // ~2 blocks per CU
    const long long max_splits = (K + 127) / 128;               // >= 4 k-tiles per split
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    int klen = (int)((K + want - 1) / want);
    klen = (klen + 63) / 64 * 64;
    SplitPlan sp;
    sp.klen = klen;
    sp.splits = (K + klen - 1) / klen;
    if (sp.splits < 1) sp.splits = 1;
    return sp;
}

// ---- split-K reductions (fixed summation order => deterministic) -------------------------
// Block = 64 element-lanes x 4 z-groups: a thread sums the slabs z = g, g+4, g+8, ... with all of its
// loads independent (no latency chain), the 4 group sums are combined through LDS in a fixed order.
constexpr int kRedLanes = 64, kRedGroups = 4;

// out[i] = sum_z partial[z*slab + i],  i in float4 units when vec4
template <bool VEC4>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int splits,
                                                            long long slab, float* __restrict__ out, long long n) {
    __shared__ float4 red[kRedGroups][kRedLanes];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long long i = ((long long)blockIdx.x * kRedLanes + lane) * (VEC4 ? 4 : 1);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
#pragma unroll 8
        for (int z = grp; z < splits; z += kRedGroups) {
            const float* q = partial + (size_t)z * slab + i;
            if constexpr (VEC4) {
                const float4 v = *reinterpret_cast<const float4*>(q);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            } else {
                s.x += q[0];
            }
        }
    }
    red[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && i < n) {
        float4 t = red[0][lane];
#pragma unroll
        for (int g = 1; g < kRedGroups; ++g) {
            t.x += red[g][lane].x; t.y += red[g][lane].y; t.z += red[g][lane].z; t.w += red[g][lane].w;
        }
        if constexpr (VEC4) *reinterpret_cast<float4*>(out + i) = t;
        else out[i] = t.x;
    }
}

static int launch_splitk_reduce(const float* partial, int splits, long long slab, float* out, long long n,
                                hipStream_t s) {
    const bool vec4 = (n % 4 == 0) && (slab % 4 == 0) && aligned16(partial) && aligned16(out);
    const long long units = vec4 ? n / 4 : n;
    const unsigned blocks = (unsigned)ceil_div_ll(units, kRedLanes);
    if (vec4) hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, s, partial, splits, slab, out, n);
    else hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, s, partial, splits, slab, out, n);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// C-Conv weight gradient fold: partial slabs hold dWeff[2kin,2F]; colsum [splits][2F]
//   dWa[n,f] = dWeff[2n,2f]   - dWeff[2n+1,2f+1]
//   dWb[n,f] = dWeff[2n,2f+1] - dWeff[2n+1,2f]
//   dba[f] = sum_r(dRe - dIm) = cs[2f] - cs[2f+1],  dbb = -dba     
// element index e in [0, kin*F) -> (n,f); e in [kin*F, kin*F+F) -> bias f
__global__ __launch_bounds__(256) void cconv_fold_kernel(const float* __restrict__ partial, int splits,
                                                         long long slab, const float* __restrict__ colsum,
                                                         float* __restrict__ dw, float* __restrict__ dbias,
                                                         int kin, int F) {
    __shared__ float2 red[kRedGroups][kRedLanes];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = blockIdx.x * kRedLanes + lane;
    const int total = kin * F, N2 = 2 * F;
    const bool is_w = e < total, is_b = (!is_w) && (e < total + F) && (dbias != nullptr);
    const int n = is_w ? e / F : 0, f = is_w ? e % F : (e - total);
    float a = 0.f, b = 0.f;
    if (is_w) {
#pragma unroll 8
        for (int z = grp; z < splits; z += kRedGroups) {
            const float* P = partial + (size_t)z * slab;
            const float2 top = *reinterpret_cast<const float2*>(P + (size_t)(2 * n) * N2 + 2 * f);
            const float2 bot = *reinterpret_cast<const float2*>(P + (size_t)(2 * n + 1) * N2 + 2 * f);
            a += top.x - bot.y;
            b += top.y - bot.x;
        }
    } else if (is_b) {
#pragma unroll 8
        for (int z = grp; z < splits; z += kRedGroups) {
            const float2 c = *reinterpret_cast<const float2*>(colsum + (size_t)z * N2 + 2 * f);
            a += c.x - c.y;
        }
    }
    red[grp][lane] = make_float2(a, b);
    __syncthreads();
    if (grp == 0) {
        float2 t = red[0][lane];
#pragma unroll
        for (int g = 1; g < kRedGroups; ++g) { t.x += red[g][lane].x; t.y += red[g][lane].y; }
        if (is_w) {
            dw[(size_t)n * N2 + f] = t.x;
            dw[(size_t)n * N2 + F + f] = t.y;
        } else if (is_b) {
            dbias[f] = t.x;
            dbias[F + f] = -t.x;
        }
    }
}

}  // namespace dccn
