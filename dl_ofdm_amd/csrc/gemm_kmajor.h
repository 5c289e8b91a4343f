// Weight-gradient GEMM with both operands "k-major": C[M,N] = A^T . B with A = [K][M] and B = [K][N] row-major -- the
// natural layout of dW = X^T . dY (k = batch row): no operand is transposed anywhere.
//
// A k-tile (64 rows of A and of B, 64 columns each) goes to LDS as it lies in memory (float4 copies, unpadded 256-byte
// rows) -- except that ODD rows are stored with their two 32-column halves swapped (word offset ^ 32: an address-only
// swizzle, the float4 stays a float4).  v_mfma_f32_16x16x4_f32 wants lane (c = l%16, kq = l/16) to supply A[i=c][k=kq]
// and B[k=kq][j=c]; the lane reads TWO adjacent columns of row k0+kq with one ds_read_b64 and uses them as the operands
// of two 16-wide sub-tiles: wave (hA, hB) owns the 32x32 quarter of the 64x64 tile, sub-tile s of A holds its rows
// m = 32 hA + 2c + s, sub-tile t of B its columns n = 32 hB + 2c + t (4 accumulators, 4 MFMAs per k-step for two
// ds_read_b64; nothing is exchanged between the waves).  ds_read_b64 is serviced in the lane groups {0-31}, {32-63}
// = (kq 0,1) and (kq 2,3): the even row's lanes cover the 32 banks of their half, the odd row's lanes -- swizzled --
// the other 32: every read is conflict free (round 2 read A with ds_read_b32 at column 4c+w: the four kq groups of a
// wave met on the same bank, SQ_LDS_BANK_CONFLICT = 25 % of the LDS cycles).  The ds_write_b128 of the staging stay
// conflict free (an 8-lane group = 32 consecutive words of one row).
// Accumulator (s,t), register r of lane l in wave (hA,hB) is C[m0 + 32hA + 8(l/16) + 2r + s][n0 + 32hB + 2(l%16) + t]:
// the two t of a lane are one float2, the 16 lanes of a row one 128-byte segment.
//
// Vector-legal operands only (M, N multiples of 4, 16-byte aligned, < 2 GiB); callers fall back to the
// gemm_f32_mfma.h path otherwise.
#pragma once
#include "gemm_f32_mfma.h"

namespace dccn {

typedef float kf32x4 __attribute__((ext_vector_type(4)));

template <int BK = 64>
constexpr size_t kmajor_smem_bytes() { return (size_t)(2 * 2 * BK * 64) * sizeof(float); }

// floor(n / d) for n < 2^31 by the multiplier / shift pair of patch_div_magic (gemm_f32_mfma.h): mul_hi + mul_lo + shift
__device__ __forceinline__ int patch_div(int n, unsigned mul, int shift) {
    return (int)(((unsigned long long)(unsigned)n * mul) >> shift);
}

// COLSUM: also the column sums of the B rows of this k range (bias gradient), written by the m0 == 0 tiles
// APATCH: operand A is the im2col patch matrix of p.pg, gathered from x = p.A (dev/py/complex.py:51-92, 140-196; the
//         weight gradient of a general-k C-Conv with no patch tensor): row k = output position (b, lo, wo), column
//         m = (ti, tj, c, iq).  Every piece of a thread lies in the same float4 column (tid % 16), so the tap
//         decomposition is a thread constant; a piece decomposes its ROW per k-tile (two multiply-shift divisions),
//         tests the padding and forms one address -- A always takes the masked form, B (dout) keeps its fast path.
template <int COLSUM, int BK = 64, bool APATCH = false>
__device__ __forceinline__ void kmajor_block(const GemmParams& p, const int L, const int T, const int z,
                                             const long long goA = 0, const long long goB = 0, const long long goC = 0,
                                             const long long goS = 0) {
    // go*: element offsets of this block's group from the operands in `p` (grouped launches; a shifted COPY of p would
    // turn the dynamically indexed koff[] into scratch memory)
    const float* __restrict__ pA = p.A + goA;
    const float* __restrict__ pB = p.B + goB;
    constexpr int BT = 64, NV = BK / 16;                // 64x64 output tile, BK-deep k-tiles, NV float4 per thread/operand
    constexpr int NKS = BK / 4;                         // k-steps per k-tile
    static_assert(BK == 32 || BK == 64, "k-tile depth");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                       // [2][BK][64]
    float* sB = smem + 2 * BK * BT;         // [2][BK][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, kq = lane >> 4;
    const int hA = w >> 1, hB = w & 1;
    const int ntn = (p.N + BT - 1) / BT;
    const int tile = xcd_tile(L, T);
    const int m0 = (tile / ntn) * BT, n0 = (tile % ntn) * BT;
    const int kbeg = p.nranges > 0 ? p.koff[z] : z * p.klen;
    const int kend = p.nranges > 0 ? p.koff[z + 1] : min(p.K, kbeg + p.klen);
    const int ntiles = (kend - kbeg + BK - 1) / BK;
    const int nfull = (kend - kbeg) / BK;

    kf32x4 acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[s][t][r] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_cs = COLSUM && p.colsum != nullptr && m0 == 0 && w == 0;       // wave-uniform

    // staging: piece v of a thread = row (tid + 256 v) / 16, float4 column (tid + 256 v) % 16 of the k-tile; the row
    // parity is bit 4 of tid for every piece, so the swizzle is one thread constant
    unsigned offA[NV], offB[NV];
    float4 ra[NV], rb[NV];
    unsigned okm = 0u;                                  // masked path: bit v = row of piece v lies inside the k range
    unsigned okb = 0u;                                  // APATCH: okm also carries A's padding test, B rows keep their own
    const unsigned sw = ((unsigned)(tid >> 4) & 1u) << 5;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int idx = tid + 256 * v, row = idx >> 4, c4 = idx & 15;
        offA[v] = (unsigned)((row * p.lda + min(m0 + 4 * c4, p.M - 4)) * 4);
        offB[v] = (unsigned)((row * p.ldb + min(n0 + 4 * c4, p.N - 4)) * 4);
    }
    int p_tl = 0, p_tw = 0, p_cc = 0;                   // APATCH: l / w offset of this thread's tap, float offset in the cell
    if constexpr (APATCH) {
        const int mc = min(m0 + 4 * (tid & 15), p.M - 4);
        const int seg = p.pg.ntw * p.pg.c2;
        const int ti = mc / seg, r2 = mc - ti * seg, tj = r2 / p.pg.c2;
        p_cc = r2 - tj * p.pg.c2;
        p_tl = p.pg.l0 + ti;
        p_tw = p.pg.w0 + tj;
    }
    auto load = [&](auto masked_tag, int q, int k0) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const int v = q % NV;
        if (APATCH && q < NV) {
            const int k = k0 + (tid >> 4) + 16 * v, kc = min(k, p.K - 1);
            const int b = patch_div(kc, p.pg.per_mul, p.pg.per_shift), rem = kc - b * (p.pg.Lo * p.pg.Wo);
            const int lo = patch_div(rem, p.pg.wo_mul, p.pg.wo_shift), wo = rem - lo * p.pg.Wo;
            const int l = lo * p.pg.sL + p_tl, wd = wo * p.pg.sW + p_tw;
            const bool ok = k < kend && (unsigned)l < (unsigned)p.pg.L && (unsigned)wd < (unsigned)p.pg.Wd;
            const unsigned off = ok ? (unsigned)(((b * p.pg.L + l) * p.pg.Wd + wd) * p.pg.c2 + p_cc) : 0u;
            ra[v] = *reinterpret_cast<const float4*>(pA + off);
            okm = (okm & ~(1u << v)) | ((ok ? 1u : 0u) << v);
        } else if constexpr (!MASKED) {
            const char* base = reinterpret_cast<const char*>(q < NV ? pA + (size_t)k0 * p.lda : pB + (size_t)k0 * p.ldb);
            if (q < NV) ra[v] = *reinterpret_cast<const float4*>(base + offA[v]);
            else rb[v] = *reinterpret_cast<const float4*>(base + offB[v]);
        } else {
            const int idx = tid + 256 * v, row = idx >> 4, c4 = idx & 15;
            const int k = k0 + row, kc = min(k, p.K - 1);
            if (q < NV) {
                ra[v] = *reinterpret_cast<const float4*>(pA + (size_t)kc * p.lda + min(m0 + 4 * c4, p.M - 4));
                okm = (okm & ~(1u << v)) | ((k < kend ? 1u : 0u) << v);
            } else {
                rb[v] = *reinterpret_cast<const float4*>(pB + (size_t)kc * p.ldb + min(n0 + 4 * c4, p.N - 4));
                if constexpr (APATCH) okb = (okb & ~(1u << v)) | ((k < kend ? 1u : 0u) << v);
            }
        }
    };
    auto store = [&](auto masked_tag, int q, float* An, float* Bn) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const int v = q % NV;
        float4 val = q < NV ? ra[v] : rb[v];
        if (MASKED || (APATCH && q < NV)) {
            const bool ok = (((APATCH && q >= NV) ? okb : okm) >> v) & 1u;
            val = make_float4(ok ? val.x : 0.f, ok ? val.y : 0.f, ok ? val.z : 0.f, ok ? val.w : 0.f);
        }
        *reinterpret_cast<float4*>((q < NV ? An : Bn) + ((unsigned)(4 * (tid + 256 * v)) ^ sw)) = val;
    };
    using TFalse = std::false_type;
    using TTrue = std::true_type;
    if (ntiles > 0) {
        if (nfull > 0) {
#pragma unroll
            for (int q = 0; q < 2 * NV; ++q) load(TFalse{}, q, kbeg);
#pragma unroll
            for (int q = 0; q < 2 * NV; ++q) store(TFalse{}, q, sA, sB);
        } else {
#pragma unroll
            for (int q = 0; q < 2 * NV; ++q) load(TTrue{}, q, kbeg);
#pragma unroll
            for (int q = 0; q < 2 * NV; ++q) store(TTrue{}, q, sA, sB);
        }
    }
    __syncthreads();

    // one k-tile = BK/4 k-steps (rows 4ks + kq) x 4 MFMAs per wave; the next k-tile's 8 global loads go out during the
    // first 8 steps, their LDS writes (other buffer) during the last 8
    const int rdA = kq * BT + ((32 * hA + 2 * c) ^ ((kq & 1) << 5));
    const int rdB = kq * BT + ((32 * hB + 2 * c) ^ ((kq & 1) << 5));
    int t = 0;
    auto ktile = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const int cur = t & 1;
        const int k0n = kbeg + (t + 1) * BK;
        const float* As = sA + cur * BK * BT + rdA;
        const float* Bs = sB + cur * BK * BT + rdB;
        float* An = sA + (cur ^ 1) * BK * BT;
        float* Bn = sB + (cur ^ 1) * BK * BT;
        float2 fa[2], fb[2];
        fa[0] = *reinterpret_cast<const float2*>(As);
        fb[0] = *reinterpret_cast<const float2*>(Bs);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks + 1 < NKS) {
                fa[(ks + 1) & 1] = *reinterpret_cast<const float2*>(As + 4 * (ks + 1) * BT);
                fb[(ks + 1) & 1] = *reinterpret_cast<const float2*>(Bs + 4 * (ks + 1) * BT);
            }
            if constexpr (MODE != PF_NONE) {
                if (ks < 2 * NV) {
                    if constexpr (MODE == PF_FAST) load(TFalse{}, ks, k0n);
                    else load(TTrue{}, ks, k0n);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const float2 a = fa[ks & 1], b = fb[ks & 1];
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.y, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.x, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE != PF_NONE) {
                if (ks >= NKS - 2 * NV) {
                    if constexpr (MODE == PF_FAST) store(TFalse{}, ks - (NKS - 2 * NV), An, Bn);
                    else store(TTrue{}, ks - (NKS - 2 * NV), An, Bn);
                }
            }
        }
        // bias gradient: wave 0 of the m0 == 0 tiles re-reads the B rows it just multiplied (a real branch around LDS
        // loads: as selects inside the loop above the compiler executed these adds in every wave of every block).
        // The swizzled address hands lane (c, kq) the logical columns 4c..4c+3 of row 4ks + kq.
        if (COLSUM && do_cs) {
            const float* Bc = sB + cur * BK * BT + kq * BT + ((4 * c) ^ ((kq & 1) << 5));
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const float4 b = *reinterpret_cast<const float4*>(Bc + 4 * ks * BT);
                bsum.x += b.x; bsum.y += b.y; bsum.z += b.z; bsum.w += b.w;
            }
        }
        __syncthreads();
        ++t;
    };
    while (t + 1 < nfull) ktile(std::integral_constant<int, PF_FAST>{});
    while (t + 1 < ntiles) ktile(std::integral_constant<int, PF_MASKED>{});
    if (ntiles > 0) ktile(std::integral_constant<int, PF_NONE>{});

    float* Cz = p.C + goC + (size_t)z * p.slab;
    const int col = n0 + 32 * hB + 2 * c;
    const int rowb = m0 + 32 * hA + 8 * kq;
    if (m0 + BT <= p.M && n0 + BT <= p.N) {                // interior tile (block-uniform): stores without exec masks
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                out_store2<2>(Cz + (size_t)(rowb + 2 * r + s) * p.ldc + col, acc[s][0][r], acc[s][1][r]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int row = rowb + 2 * r + s;
                if (row < p.M && col < p.N)
                    out_store2<2>(Cz + (size_t)row * p.ldc + col, acc[s][0][r], acc[s][1][r]);
            }
    }
    if (COLSUM && do_cs) {
        // wave 0 has seen every element of the B rows once: lanes c, c+16, c+32, c+48 hold the same four columns
#pragma unroll
        for (int m = 16; m < 64; m <<= 1) {
            bsum.x += __shfl_xor(bsum.x, m, 64);
            bsum.y += __shfl_xor(bsum.y, m, 64);
            bsum.z += __shfl_xor(bsum.z, m, 64);
            bsum.w += __shfl_xor(bsum.w, m, 64);
        }
        const int cc = n0 + 4 * c;
        if (lane < 16 && cc < p.N) out_store4<2>(p.colsum + goS + (size_t)z * p.N + cc, bsum.x, bsum.y, bsum.z, bsum.w);
    }
}

static inline bool kmajor_ok(const GemmParams& p) {
    return p.vecA && p.vecB && (p.M % 4 == 0) && (p.N % 4 == 0) && (p.lda % 4 == 0) && (p.ldb % 4 == 0) && (p.ldc % 4 == 0) &&
           p.M >= 4 && p.N >= 4 && ((reinterpret_cast<uintptr_t>(p.C) | (uintptr_t)(p.slab * 4)) & 15) == 0;
}

// stand-alone launch: grid (tiles, 1, splits)
template <int COLSUM, int TAG, bool APATCH = false>
__global__ __launch_bounds__(kGemmThreads) void gemm_kmajor_kernel(const GemmParams p0, const ChainOffs co) {
    // chain groups (common.h): y = the chain (z carries the k ranges); its arena offset enters as the block's operand
    // offsets -- a shifted copy of the parameter block would put the dynamically indexed koff[] into scratch memory
    const long long c4 = co.off[blockIdx.y] / 4;
    kmajor_block<COLSUM, 64, APATCH>(p0, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.z, c4, c4, c4, c4);
}

template <int COLSUM, int TAG, bool APATCH = false>
static int launch_kmajor(const GemmParams& p, int splits, hipStream_t s) {
    auto kern = gemm_kmajor_kernel<COLSUM, TAG, APATCH>;
    DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), kmajor_smem_bytes<64>()));
    dim3 grid(ceil_div(p.N, 64) * ceil_div(p.M, 64), 1, splits);
    DCCN_LAUNCH_CHAINS_Y(kern, grid, dim3(kGemmThreads), kmajor_smem_bytes<64>(), s, p);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// dense backward in one grid: blocks [0, nx) = the dX tiles (gemm_f32_mfma.h, k-contiguous operands), the rest = the
// dW (tile, split) items in the k-major form.  CMAP: column map of the dX stores (gemm_store)
template <int BK, int CMAP = CMAP_NONE>
__global__ __launch_bounds__(kGemmThreads) void dense_bwd_grouped_km_kernel(const GemmParams px0, const GemmParams pw0,
                                                                            const int nx, const int tw, const ChainOffs co) {
    const long long coff = co.off[blockIdx.z];              // chain groups (common.h)
    const int b = (int)blockIdx.x;
#ifdef GROUPED_ABL          // timing experiments only: 1 = the dX blocks return at once, 2 = the dW blocks do
    if ((GROUPED_ABL == 1) == (b < nx)) return;
#endif
    if (b < nx) {
        const GemmParams px = px0.at_chain(coff);
        gemm_block<OP_KCONTIG, OP_KCONTIG, 64, 64, BK, 0, true, 2, CMAP>(px, b, nx, 0);
    } else {
        const int c = b - nx;
        const long long c4 = coff / 4;                      // (kmajor_block: offsets, not a shifted copy -- koff[] is indexed)
        kmajor_block<1, BK>(pw0, c % tw, tw, c / tw, c4, c4, c4, c4);
    }
}

template <int BK, int CMAP = CMAP_NONE>
static int launch_dense_bwd_grouped_km(const GemmParams& px, const GemmParams& pw, int splits_w, hipStream_t s) {
    constexpr size_t sx = gemm_smem_bytes<OP_KCONTIG, OP_KCONTIG, 64, 64, BK>();
    constexpr size_t smem = sx > kmajor_smem_bytes<BK>() ? sx : kmajor_smem_bytes<BK>();
    auto kern = dense_bwd_grouped_km_kernel<BK, CMAP>;
    DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), smem));
    const int nx = ceil_div(px.N, 64) * ceil_div(px.M, 64);
    const int tw = ceil_div(pw.N, 64) * ceil_div(pw.M, 64);
    DCCN_LAUNCH_CHAINS_Z(kern, dim3(nx + tw * splits_w), dim3(kGemmThreads), smem, s, px, pw, nx, tw);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// C-Conv backward of `groups` same-shaped layers in one grid (the equaliser's corr / eq pair, or one layer):
// blocks [0, groups*nx) = the dX tiles  dx_g = dout_g . Weff_g^T  (group-major), the rest = the (group, tile, split)
// items of  dWeff_g = x_g^T . dout_g  in the k-major form, slabs left for the optimizer launch to fold.
// Four launches of the pair (two dX, two dW) + two folds become one grid.
template <int BK>
__global__ __launch_bounds__(kGemmThreads) void cconv_bwd_grouped_km_kernel(const GemmParams px0, const GroupStride gx,
                                                                            const GemmParams pw0, const GroupStride gw,
                                                                            const int nx, const int tw, const int splits,
                                                                            const int groups, const ChainOffs co) {
    const long long coff = co.off[blockIdx.z];              // chain groups (common.h)
    const int b = (int)blockIdx.x;
    if (b < nx * groups) {
        const GemmParams q = group_params(px0.at_chain(coff), gx, b / nx);
        gemm_block<OP_KCONTIG, OP_CCONV_WT, 64, 64, BK, 0, true>(q, b % nx, nx, 0);
    } else {
        const int c = b - nx * groups, per = tw * splits;
        const int g = c / per, i = c % per;
        const long long c4 = coff / 4;
        kmajor_block<1, BK>(pw0, i % tw, tw, i / tw, g * gw.a + c4, g * gw.b + c4, g * gw.c + c4, g * gw.colsum + c4);
    }
}

template <int BK>
static int launch_cconv_bwd_grouped_km(const GemmParams& px, const GroupStride& gx, const GemmParams& pw,
                                       const GroupStride& gw, int splits_w, int groups, hipStream_t s) {
    constexpr size_t sx = gemm_smem_bytes<OP_KCONTIG, OP_CCONV_WT, 64, 64, BK>();
    constexpr size_t smem = sx > kmajor_smem_bytes<BK>() ? sx : kmajor_smem_bytes<BK>();
    auto kern = cconv_bwd_grouped_km_kernel<BK>;
    DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), smem));
    const int nx = ceil_div(px.N, 64) * ceil_div(px.M, 64);
    const int tw = ceil_div(pw.N, 64) * ceil_div(pw.M, 64);
    DCCN_LAUNCH_CHAINS_Z(kern, dim3((nx + tw * splits_w) * groups), dim3(kGemmThreads), smem, s, px, gx, pw, gw, nx, tw,
                         splits_w, groups);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

}  // namespace dccn
