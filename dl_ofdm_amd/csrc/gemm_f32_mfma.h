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

constexpr int kBK = 32;

__device__ __forceinline__ float cconv_weff(const float* __restrict__ w, int F, int row, int col) {
    const int n = row >> 1, q = row & 1, f = col >> 1, c = col & 1;
    const float v = w[(size_t)n * 2 * F + ((q ^ c) ? F : 0) + f];
    return q ? -v : v;
}

template <int KIND, int BI>
struct Tile {
    static constexpr int NV = kBK * BI / 4 / 256;
    static constexpr bool KV = (KIND == OP_KCONTIG || KIND == OP_CCONV_WT);
    static constexpr int LD = KV ? BI + 1 : BI;
    float4 r[NV];

    // (k0..kend) x (i0..I) window of the operand; out-of-range elements read as 0
    __device__ __forceinline__ void load(const float* __restrict__ p, int ld, int k0, int kend,
                                         int i0, int I, int cF, int vec, int tid) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * 256;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (!KV) {
                const int k = k0 + idx / (BI / 4);
                const int i = i0 + (idx % (BI / 4)) * 4;
                if (k < kend) {
                    if constexpr (KIND == OP_ICONTIG) {
                        const float* q = p + (size_t)k * ld + i;
                        if (vec && i + 3 < I) {
                            val = *reinterpret_cast<const float4*>(q);
                        } else {
                            if (i < I) val.x = q[0];
                            if (i + 1 < I) val.y = q[1];
                            if (i + 2 < I) val.z = q[2];
                            if (i + 3 < I) val.w = q[3];
                        }
                    } else {
                        if (i < I) val.x = cconv_weff(p, cF, k, i);
                        if (i + 1 < I) val.y = cconv_weff(p, cF, k, i + 1);
                        if (i + 2 < I) val.z = cconv_weff(p, cF, k, i + 2);
                        if (i + 3 < I) val.w = cconv_weff(p, cF, k, i + 3);
                    }
                }
            } else {
                const int i = i0 + idx / (kBK / 4);
                const int k = k0 + (idx % (kBK / 4)) * 4;
                if (i < I) {
                    if constexpr (KIND == OP_KCONTIG) {
                        const float* q = p + (size_t)i * ld + k;
                        if (vec && k + 3 < kend) {
                            val = *reinterpret_cast<const float4*>(q);
                        } else {
                            if (k < kend) val.x = q[0];
                            if (k + 1 < kend) val.y = q[1];
                            if (k + 2 < kend) val.z = q[2];
                            if (k + 3 < kend) val.w = q[3];
                        }
                    } else {
                        if (k < kend) val.x = cconv_weff(p, cF, i, k);
                        if (k + 1 < kend) val.y = cconv_weff(p, cF, i, k + 1);
                        if (k + 2 < kend) val.z = cconv_weff(p, cF, i, k + 2);
                        if (k + 3 < kend) val.w = cconv_weff(p, cF, i, k + 3);
                    }
                }
            }
            r[v] = val;
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ lds, int tid) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * 256;
            if constexpr (!KV) {
                const int k = idx / (BI / 4);
                const int i = (idx % (BI / 4)) * 4;
                *reinterpret_cast<float4*>(lds + k * LD + i) = r[v];
            } else {
                const int i = idx / (kBK / 4);
                const int k = (idx % (kBK / 4)) * 4;
                lds[(k + 0) * LD + i] = r[v].x;
                lds[(k + 1) * LD + i] = r[v].y;
                lds[(k + 2) * LD + i] = r[v].z;
                lds[(k + 3) * LD + i] = r[v].w;
            }
        }
    }
};

template <int KA, int KB, int BM, int BN>
constexpr size_t gemm_smem_bytes() {
    return (size_t)(2 * kBK * Tile<KA, BM>::LD + 2 * kBK * Tile<KB, BN>::LD) * sizeof(float);
}

// TAG only makes the symbol unique per call site so profiles attribute time to the right operator
template <int KA, int KB, int BM, int BN, int COLSUM, int TAG>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(const GemmParams p) {
    using TA = Tile<KA, BM>;
    using TB = Tile<KB, BN>;
    constexpr int LDA = TA::LD, LDB = TB::LD;
    constexpr int TM = BM / 64, TN = BN / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                       // [2][kBK][LDA]
    float* sB = smem + 2 * kBK * LDA;       // [2][kBK][LDB]   (2*32*LDA is a multiple of 4)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm0 = (wid >> 1) * (BM / 2), wn0 = (wid & 1) * (BN / 2);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
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
    const bool do_cs = COLSUM && p.colsum != nullptr && blockIdx.y == 0 && tid < BN;

    TA ta;
    TB tb;
    if (ntiles > 0) {
        ta.load(p.A, p.lda, kbeg, kend, m0, p.M, p.cF, p.vecA, tid);
        tb.load(p.B, p.ldb, kbeg, kend, n0, p.N, p.cF, p.vecB, tid);
        ta.store(sA, tid);
        tb.store(sB, tid);
    }
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1 < ntiles);
        if (more) {
            const int k0 = kbeg + (t + 1) * kBK;
            ta.load(p.A, p.lda, k0, kend, m0, p.M, p.cF, p.vecA, tid);
            tb.load(p.B, p.ldb, k0, kend, n0, p.N, p.cF, p.vecB, tid);
        }
        const float* As = sA + cur * kBK * LDA;
        const float* Bs = sB + cur * kBK * LDB;
#pragma unroll
        for (int kk = 0; kk < kBK / 2; ++kk) {
            float av[TM], bv[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) av[a] = As[(2 * kk + h) * LDA + wm0 + a * 32 + l31];
#pragma unroll
            for (int b = 0; b < TN; ++b) bv[b] = Bs[(2 * kk + h) * LDB + wn0 + b * 32 + l31];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (COLSUM && do_cs) {
#pragma unroll 8
            for (int k = 0; k < kBK; ++k) cs += Bs[k * LDB + tid];
        }
        if (more) {
            ta.store(sA + (cur ^ 1) * kBK * LDA, tid);
            tb.store(sB + (cur ^ 1) * kBK * LDB, tid);
        }
        __syncthreads();
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

template <int KA, int KB, int BM, int BN, int COLSUM, int TAG>
static int launch_gemm_cfg(const GemmParams& p, int splits, hipStream_t s) {
    auto kern = gemm_f32_mfma_kernel<KA, KB, BM, BN, COLSUM, TAG>;
    constexpr size_t smem = gemm_smem_bytes<KA, KB, BM, BN>();
    static bool attr_done = false;
    if (!attr_done) {
        if (smem > 48 * 1024)
            DCCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_done = true;
    }
    dim3 grid(ceil_div(p.N, BN), ceil_div(p.M, BM), splits);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// tile choice: 128x128 only when it still yields >= 2 blocks per CU, else 64x64
template <int KA, int KB, int COLSUM, int TAG>
static int launch_gemm(const GemmParams& p, int splits, hipStream_t s) {
    const long long big = (long long)ceil_div(p.M, 128) * ceil_div(p.N, 128) * splits;
    if (big >= 2 * kCUs) return launch_gemm_cfg<KA, KB, 128, 128, COLSUM, TAG>(p, splits, s);
    return launch_gemm_cfg<KA, KB, 64, 64, COLSUM, TAG>(p, splits, s);
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
    klen = (klen + kBK - 1) / kBK * kBK;
    SplitPlan sp;
    sp.klen = klen;
    sp.splits = (K + klen - 1) / klen;
    if (sp.splits < 1) sp.splits = 1;
    return sp;
}

// ---- split-K reductions (fixed summation order => deterministic) ---------------------
// out[i] = sum_z partial[z*slab + i]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial,
                                                            int splits, long long slab,
                                                            float* __restrict__ out, long long n) {
    const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 3 < n && (slab & 3) == 0) {
        float4 s = *reinterpret_cast<const float4*>(partial + i4);
        for (int z = 1; z < splits; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)z * slab + i4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if ((reinterpret_cast<uintptr_t>(out + i4) & 15u) == 0) {
            *reinterpret_cast<float4*>(out + i4) = s;
        } else {
            out[i4] = s.x; out[i4 + 1] = s.y; out[i4 + 2] = s.z; out[i4 + 3] = s.w;
        }
    } else {
        for (long long i = i4; i < n && i < i4 + 4; ++i) {
            float s = partial[i];
            for (int z = 1; z < splits; ++z) s += partial[(size_t)z * slab + i];
            out[i] = s;
        }
    }
}

// C-Conv weight gradient fold: partial slabs hold dWeff[2kin,2F]; colsum [splits][2F]
//   dWa[n,f] = dWeff[2n,2f]   - dWeff[2n+1,2f+1]
//   dWb[n,f] = dWeff[2n,2f+1] - dWeff[2n+1,2f]
//   dba[f] = sum_r(dRe - dIm) = cs[2f] - cs[2f+1],  dbb = -dba     
__global__ __launch_bounds__(256) void cconv_fold_kernel(const float* __restrict__ partial, int splits,
                                                         long long slab, const float* __restrict__ colsum,
                                                         float* __restrict__ dw, float* __restrict__ dbias,
                                                         int kin, int F) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // over kin*F (+F bias threads)
    const int total = kin * F;
    if (idx < total) {
        const int n = idx / F, f = idx % F;
        const int N2 = 2 * F;
        float a = 0.f, b = 0.f;
        for (int z = 0; z < splits; ++z) {
            const float* P = partial + (size_t)z * slab;
            const float2 top = *reinterpret_cast<const float2*>(P + (size_t)(2 * n) * N2 + 2 * f);
            const float2 bot = *reinterpret_cast<const float2*>(P + (size_t)(2 * n + 1) * N2 + 2 * f);
            a += top.x - bot.y;
            b += top.y - bot.x;
        }
        dw[(size_t)n * N2 + f] = a;
        dw[(size_t)n * N2 + F + f] = b;
    } else if (idx < total + F && dbias != nullptr) {
        const int f = idx - total;
        float s = 0.f;
        for (int z = 0; z < splits; ++z)
            s += colsum[(size_t)z * 2 * F + 2 * f] - colsum[(size_t)z * 2 * F + 2 * f + 1];
        dbias[f] = s;
        dbias[F + f] = -s;
    }
}

}  // namespace dccn
