// Few-row GEMMs (the equaliser's 73-frame batch: [73, 896] . [896, 896] and friends, dev/py/model.py:393-428 and their
// backward): C[M, N] = A[M, K] . B with M <= 96 (or a short k range on a few hundred tiles), K a multiple of 16.
//
// Why a kernel of its own.  On 16x64 tiles of the general gemm16 kernel such a product is 70 blocks that each walk 14
// k-tiles with loads two tiles ahead: 14 dependent memory latencies on 70 of 256 CUs, 13-14 us for 0.117 GFLOP
// (profiles/r03_eq73_kernel_stats.txt) -- a latency chain, not work.  Here:
//   * one block per 16x16 output tile (M = 73, N = 896: 280 blocks, every CU has one);
//   * the block's four waves split the k range (wave w owns the 16-deep k groups w, w+4, ...) and meet once, at the end,
//     through a 4 KB LDS exchange summed in wave order (fixed order => deterministic);
//   * BOTH operands go from global memory straight into the registers of the MFMA that consumes them -- no LDS staging,
//     no barrier in the loop: lane (c = lane % 16, kq = lane / 16) of v_mfma_f32_16x16x4_f32 supplies A[row c][k] and
//     B[k][col c] for the k of its quarter kq, so a float4 of four consecutive k of row c is the A operand of four
//     consecutive MFMAs (the k order inside a group of 16 is permuted, identically on both operands);
//   * EVERY load of the block is issued before the first MFMA: one memory latency per launch instead of fourteen.
// B operand kinds: OP_ICONTIG (forward, W[k][n]: four dword loads per k group, 64-byte segments; the tiles of one 16-column
// strip and of its neighbour in the same 128-byte lines run on the same XCD, so each line leaves HBM/MALL once) and
// OP_KCONTIG (dX = dY . W^T, W[n][k]: float4 loads).  Store stages as in gemm16.h EPI_STORE: 1 plain (+ bias), 2 tanh,
// 3 times (1 - aux^2), 4 plus aux, 5 equalise (model.py:431-438).
#pragma once
#include "gemm16.h"

namespace dccn {

// tile L of T = ntm * ntn tiles, strip-major (the row tiles of one 16-column strip are consecutive); XCD-aware order as in
// gemm16.h: block b runs on XCD b % 8, every XCD gets a contiguous run of that order (whole strips, neighbours together)
template <int BKIND, int NG, int NB>
__device__ __forceinline__ void fewrow_tile(const GemmParams& p, const int L, const int T) {
    static_assert(BKIND == OP_ICONTIG || BKIND == OP_KCONTIG, "B operand kind");
    __shared__ float xch[4][4][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, kq = lane >> 4;
    const int ntm = (p.M + 15) >> 4;
    int tile;
    {
        const int xcd = L & 7, j = L >> 3, q = T >> 3, r = T & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int n0 = (tile / ntm) * 16, m0 = (tile % ntm) * 16;
    const int row = min(m0 + c, p.M - 1);                 // rows past M: clamped duplicates, masked at the store
    // k groups of 16: wave w owns groups w, w + 4, ...; K / 16 need not be a multiple of 4 * NG -- a slot past the last
    // group re-reads the last group (valid addresses) and contributes zeros (wave-uniform select, no divergence)
    const int G = p.K >> 4;
    // ---- every operand of this wave, requested up front ----
    float4 a[NG];
    const float* Ap = p.A + (size_t)row * p.lda + 4 * kq;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int g = w + 4 * i;
        const float4 t = *reinterpret_cast<const float4*>(Ap + 16 * min(g, G - 1));
        a[i] = g < G ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 b[NG];
    if constexpr (BKIND == OP_KCONTIG) {
        const float* Bp = p.B + (size_t)(n0 + c) * p.ldb + 4 * kq;
#pragma unroll
        for (int i = 0; i < NG; ++i) b[i] = *reinterpret_cast<const float4*>(Bp + 16 * min(w + 4 * i, G - 1));
    } else {
        const float* Bp = p.B + (size_t)(4 * kq) * p.ldb + n0 + c;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const float* q = Bp + (size_t)(16 * min(w + 4 * i, G - 1)) * p.ldb;
            b[i].x = q[0];
            b[i].y = q[(size_t)p.ldb];
            b[i].z = q[(size_t)2 * p.ldb];
            b[i].w = q[(size_t)3 * p.ldb];
        }
    }
    // what the store stage needs from memory is requested now as well (wave w finishes register w of the tile)
    const int orow = m0 + 4 * kq + w, ocol = n0 + c;
    const bool live = orow < p.M;
    const size_t ci = (size_t)min(orow, p.M - 1) * p.ldc + ocol;
    float bj = 0.f, auxv = 0.f;
    float2 yv = make_float2(0.f, 0.f);
    if (p.bias != nullptr) bj = p.bias[ocol];
    if constexpr (NB == 3 || NB == 4) auxv = p.aux[ci];
    if constexpr (NB == 5) yv = *reinterpret_cast<const float2*>(p.aux + (ci - (size_t)(ocol & 1)));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[i].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[i].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[i].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[i].w, acc, 0, 0, 0);
    }
    // ---- the four k quarters meet: C/D layout col = lane % 16, row = 4 (lane / 16) + r ----
#pragma unroll
    for (int r = 0; r < 4; ++r) xch[w][r][lane] = acc[r];
    __syncthreads();
    const float v = ((xch[0][w][lane] + xch[1][w][lane]) + xch[2][w][lane]) + xch[3][w][lane];
    float o = v + bj;
    if constexpr (NB == 5) {
        // channel estimate + equalise + autocorrelation, the expressions of gemm16.h EPI_STORE<5> / equalizer.h equalize_one
        const float partner = __shfl_xor(o, 1, 64);
        if (live) {
            const size_t c0 = ci - (size_t)(ocol & 1);
            p.C[ci] = o;
            const float hr = (ocol & 1) ? partner : o, hi = (ocol & 1) ? o : partner;
            const float aa = sqrtf(hr * hr + hi * hi);
            const float cr = hr / aa, cim = (-hi) / aa;
            const float er = yv.x * cr - yv.y * cim, ei = yv.x * cim + yv.y * cr;
            if ((ocol & 1) == 0) *reinterpret_cast<float2*>(p.out2 + c0) = make_float2(er, ei);
            else *reinterpret_cast<float2*>(p.out3 + c0) = make_float2(er * er - ei * (-ei), er * (-ei) + ei * er);
        }
    } else {
        if constexpr (NB == 2) o = tanhf(o);
        else if constexpr (NB == 3) o = o * (1.0f - auxv * auxv);
        else if constexpr (NB == 4) o = auxv + o;
        if (live) p.C[ci] = o;
    }
}

// The same tile with the demodulation tail (R3-R6 forward, and backward to dz when BWD) behind it, for NB <= 2
// (dev/py/model.py:1278-1291 on the frozen receiver of the equaliser step): after the k quarters have met, wave w holds
// row 4 kq + w of the tile, lanes 2d / 2d+1 its columns 2d (I) and 2d+1 (Q) of data cell d -- the pair swaps halves on the
// DPP crossbar and the EVEN lane runs the cell (tail.h tail_cells, one cell per lane), the odd lane contributes nothing.
// Per-block metric / tail-gradient slabs as the fused gemm16 launch leaves them (slab = the block's index).
template <int NG, int NB, bool BWD>
__device__ __forceinline__ void fewrow_tail_tile(const GemmParams& p, const TailEpiParams& tp, const int L, const int T) {
    static_assert(NB == 1 || NB == 2, "register tail: BPSK / QPSK");
    __shared__ float xch[4][4][64];
    __shared__ __attribute__((aligned(8))) float red[tail_reduce_lds_floats<NB, BWD>(256)];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, kq = lane >> 4;
    const int ntm = (p.M + 15) >> 4;
    int tile;
    {
        const int xcd = L & 7, j = L >> 3, q = T >> 3, r = T & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int n0 = (tile / ntm) * 16, m0 = (tile % ntm) * 16;
    const int row = min(m0 + c, p.M - 1);
    const int G = p.K >> 4;
    float4 a[NG], b[NG];
    const float* Ap = p.A + (size_t)row * p.lda + 4 * kq;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int g = w + 4 * i;
        const float4 t = *reinterpret_cast<const float4*>(Ap + 16 * min(g, G - 1));
        a[i] = g < G ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float* Bp = p.B + (size_t)(4 * kq) * p.ldb + n0 + c;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const float* q = Bp + (size_t)(16 * min(w + 4 * i, G - 1)) * p.ldb;
        b[i].x = q[0];
        b[i].y = q[(size_t)p.ldb];
        b[i].z = q[(size_t)2 * p.ldb];
        b[i].w = q[(size_t)3 * p.ldb];
    }
    // this lane's cell: row orow, data cell (n0 + c) / 2 (even lanes run it); its labels and bias are requested now
    const int orow = m0 + 4 * kq + w, ocol = n0 + c;
    const int Dn = p.N >> 1;
    const bool live = orow < p.M && (c & 1) == 0;
    const long long cell = (long long)min(orow, p.M - 1) * Dn + (ocol >> 1);
    int lab[1][NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) lab[0][j] = tp.bits[cell * NB + j];
    const float bj = p.bias != nullptr ? p.bias[ocol] : 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[i].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[i].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[i].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[i].w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) xch[w][r][lane] = acc[r];
    __syncthreads();
    const float o = (((xch[0][w][lane] + xch[1][w][lane]) + xch[2][w][lane]) + xch[3][w][lane]) + bj;
    if (p.C != nullptr && orow < p.M) p.C[(size_t)orow * p.ldc + ocol] = o;          // z, when the caller wants it
    const float partner = __builtin_bit_cast(float, dpp_mov_i32(__builtin_bit_cast(int, o), 0));      // quad_perm [1,0,3,2]
    const float z0[1] = {o}, z1[1] = {partner};
    const bool vd[1] = {live};
    float* const pc[1] = {(tp.prob != nullptr && live) ? tp.prob + cell * NB * 2 : nullptr};
    float2 dv[1];
    TailLaneAcc<NB, BWD> A;
    A.clear();
    tail_cells<NB, BWD, 1>(z0, z1, lab, vd, tp.tailp, tp.inv_count, pc, A, dv);
    if constexpr (BWD) {
        if (live) *reinterpret_cast<float2*>(tp.dz + cell * 2) = dv[0];
    }
    tail_block_reduce<NB, BWD, 256>(A, red, tp.blk_metrics, tp.blk_grads, L);
}

// (chain groups, common.h: blockIdx.z = chain, every pointer moves by that chain's arena offset)
template <int NG, int NB, bool BWD>
__global__ __launch_bounds__(256) void fewrow_tail_kernel(const GemmParams p0, const TailEpiParams tp0, const ChainOffs co) {
    const long long coff = co.off[blockIdx.z];
    const GemmParams p = p0.at_chain(coff);
    const TailEpiParams tp = tp0.at_chain(coff);
    stamp_mark(p.stamp, 0);
    fewrow_tail_tile<NG, NB, BWD>(p, tp, (int)blockIdx.x, (int)gridDim.x);
    stamp_mark(p.stamp, 1);
}

template <int BKIND, int NG, int NB, int TAG>
__global__ __launch_bounds__(256) void fewrow_kernel(const GemmParams p0, const ChainOffs co) {
    const GemmParams p = p0.at_chain(co.off[blockIdx.z]);
    stamp_mark(p.stamp, 0);
    fewrow_tile<BKIND, NG, NB>(p, (int)blockIdx.x, (int)gridDim.x);
    stamp_mark(p.stamp, 1);
}

// dW[Ko, N] = X^T . dY with the FEW rows as the contraction (k <= 96): one block per 64x64 output tile, every operand
// requested up front.  X[k][.] and dY[k][.] are k-major as they lie in memory: lane (c, kq) loads the float4 X[k][m0 + 4c ..]
// and dY[k][n0 + 4c ..] of row k = 4 t + kq and uses their four components as the operands of 4 x 4 MFMA sub-tiles whose rows /
// columns interleave (sub-tile (ai, bj) holds rows m0 + 4 i + ai, columns n0 + 4 j + bj): no transposes, no LDS staging.
// Wave w owns the k steps t = w, w + 4, ...; the waves meet once in LDS (fixed order), wave w finishes the sub-tiles ai = w
// and stores rows of four consecutive columns as float4.  Column sums of dY (bias gradient) by the m0 = 0 tiles.
constexpr int kFewrowDwSteps = 6;             // k steps per wave: 4 waves x 6 steps x 4 rows = 96 rows
constexpr size_t kFewrowDwSmem = (size_t)(4 * 16 * 4 * 64 + 4 * 64 * 4) * sizeof(float);
__device__ __forceinline__ void fewrow_dw_tile(const GemmParams& p, const int tile, float* __restrict__ smem) {
    float* xch = smem;                         // [wave][ai][bj][r][lane]
    float* csx = smem + 4 * 16 * 4 * 64;       // [wave][bj][lane]   (column sums, m0 = 0 tiles)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, kq = lane >> 4;
    const int ntn = p.N >> 6;
    const int m0 = (tile / ntn) * 64, n0 = (tile % ntn) * 64;
    const int rows = p.K;                      // the few rows
    float4 a[kFewrowDwSteps], b[kFewrowDwSteps];
#pragma unroll
    for (int i = 0; i < kFewrowDwSteps; ++i) {
        const int k = 4 * (w + 4 * i) + kq;
        const int kc = min(k, rows - 1);
        const float4 ta = *reinterpret_cast<const float4*>(p.A + (size_t)kc * p.lda + m0 + 4 * c);
        const float4 tb = *reinterpret_cast<const float4*>(p.B + (size_t)kc * p.ldb + n0 + 4 * c);
        const bool ok = k < rows;
        a[i] = ok ? ta : make_float4(0.f, 0.f, 0.f, 0.f);
        b[i] = ok ? tb : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int ai = 0; ai < 4; ++ai)
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) acc[ai][bj] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_cs = p.colsum != nullptr && m0 == 0;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kFewrowDwSteps; ++i) {
        const float av[4] = {a[i].x, a[i].y, a[i].z, a[i].w}, bv[4] = {b[i].x, b[i].y, b[i].z, b[i].w};
#pragma unroll
        for (int ai = 0; ai < 4; ++ai)
#pragma unroll
            for (int bj = 0; bj < 4; ++bj)
                acc[ai][bj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ai], bv[bj], acc[ai][bj], 0, 0, 0);
        if (do_cs) {
#pragma unroll
            for (int bj = 0; bj < 4; ++bj) cs[bj] += bv[bj];
        }
    }
    // every wave leaves the sub-tiles it does not finish itself
#pragma unroll
    for (int ai = 0; ai < 4; ++ai) {
        if (ai == w) continue;
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
            for (int r = 0; r < 4; ++r) xch[(((w * 4 + ai) * 4 + bj) * 4 + r) * 64 + lane] = acc[ai][bj][r];
    }
    if (do_cs) {
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) {
            float v = cs[bj];
            v += __shfl_xor(v, 16, 64);        // the four k quarters of the wave
            v += __shfl_xor(v, 32, 64);
            csx[(w * 4 + bj) * 64 + lane] = v;
        }
    }
    __syncthreads();
    // wave w: sub-tiles (ai = w, bj = 0..3), partial sums in wave order 0..3 (its own share taken from registers)
    f32x4 own[4];
#pragma unroll
    for (int bj = 0; bj < 4; ++bj) {
        own[bj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ai = 0; ai < 4; ++ai)
            if (ai == w) own[bj] = acc[ai][bj];                    // (wave-uniform select)
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float o[4];
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) {
            float v = 0.f;
#pragma unroll
            for (int ws = 0; ws < 4; ++ws)
                v += (ws == w) ? own[bj][r] : xch[(((ws * 4 + w) * 4 + bj) * 4 + r) * 64 + lane];
            o[bj] = v;
        }
        const int row = m0 + 4 * (4 * kq + r) + w;
        *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + n0 + 4 * c) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (do_cs && w == 0 && kq == 0) {
        float o[4];
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
            o[bj] = ((csx[(0 * 4 + bj) * 64 + lane] + csx[(1 * 4 + bj) * 64 + lane]) + csx[(2 * 4 + bj) * 64 + lane]) +
                    csx[(3 * 4 + bj) * 64 + lane];
        *reinterpret_cast<float4*>(p.colsum + n0 + 4 * c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
static inline bool fewrow_dw_ok(const GemmParams& pw) {
    return pw.K >= 1 && pw.K <= 4 * 4 * kFewrowDwSteps && (pw.M % 64) == 0 && (pw.N % 64) == 0 && pw.vecA && pw.vecB &&
           (pw.lda % 4) == 0 && (pw.ldb % 4) == 0 && (pw.ldc % 4) == 0 && aligned16(pw.C) &&
           (pw.colsum == nullptr || aligned16(pw.colsum));
}

// dense backward of a few-row batch in ONE grid: dX tiles as above (ACTX: element-wise stage of the caller's graph on the
// way out) and the unsplit dW = X^T . dY tiles of gemm16.h behind them (k = the few rows: one or two k-tiles)
template <int NG, int ACTX, int WWGM, int WWGN, int WTM, int WTN, int BK, bool DWFEW>
__global__ __launch_bounds__(256) void dense_bwd_fewrow_kernel(const GemmParams px0, const GemmParams pw0, const int nx, const int tw,
                                                               const ChainOffs co) {
    const long long coff = co.off[blockIdx.z];
    const GemmParams px = px0.at_chain(coff), pw = pw0.at_chain(coff);
    const int b = (int)blockIdx.x;
    stamp_mark(px.stamp, 0);
    if (b < nx) {
        fewrow_tile<OP_KCONTIG, NG, ACTX>(px, b, nx);
    } else {
        const int c = b - nx;
        if constexpr (DWFEW) {
            extern __shared__ __attribute__((aligned(16))) float smem[];
            fewrow_dw_tile(pw, c, smem);
        } else {
            TailEpiParams none{};
            gemm16_block<OP_ICONTIG, OP_ICONTIG, WWGM, WWGN, WTM, WTN, BK, 1, 1, EPI_STORE, 1, false>(pw, none, c % tw, tw, c / tw, 0);
        }
    }
    stamp_mark(px.stamp, 1);
}

// shapes the kernel takes: vector-legal operands, 16-column strips, K a multiple of 16 up to 1152; few rows (<= 96), or a
// short k range (<= 256) on at most 640 tiles (the equaliser's [511, 160] . [160, 128] and [511, 256] . [256, 160] layers).
// Returns the instantiated depth (k-group slots per wave) that covers K, 0 = not taken.
static inline int fewrow_ng_c(const GemmParams& p, bool need_c);
static inline int fewrow_ng(const GemmParams& p) { return fewrow_ng_c(p, true); }
// need_c = false: the fused dense + tail form, whose output z is optional
static inline int fewrow_ng_c(const GemmParams& p, bool need_c) {
    if (p.M < 1 || (p.N % 16) != 0 || (p.K % 16) != 0 || !p.vecA || !p.vecB || (need_c && p.C == nullptr)) return 0;
    if ((p.lda % 4) != 0 || (p.ldb % 4) != 0 || p.K < 128 || p.K > 1152) return 0;
    const long long tiles = (long long)ceil_div(p.M, 16) * (p.N / 16);
    if (!(p.M <= 96 || (p.K <= 256 && tiles <= 640))) return 0;
    const int need = ceil_div(p.K / 16, 4);
    if (need <= 3) return 3;
    if (need <= 4) return 4;
    if (need <= 10) return 10;
    if (need <= 14) return 14;
    return 18;
}

template <int BKIND, int NB, int TAG>
static int launch_fewrow(const GemmParams& p, hipStream_t s) {
    const int T = ceil_div(p.M, 16) * (p.N / 16);
    switch (fewrow_ng(p)) {
        case 18: DCCN_LAUNCH_CHAINS_Z((fewrow_kernel<BKIND, 18, NB, TAG>), dim3(T), dim3(256), 0, s, p); break;
        case 14: DCCN_LAUNCH_CHAINS_Z((fewrow_kernel<BKIND, 14, NB, TAG>), dim3(T), dim3(256), 0, s, p); break;
        case 10: DCCN_LAUNCH_CHAINS_Z((fewrow_kernel<BKIND, 10, NB, TAG>), dim3(T), dim3(256), 0, s, p); break;
        case 4: DCCN_LAUNCH_CHAINS_Z((fewrow_kernel<BKIND, 4, NB, TAG>), dim3(T), dim3(256), 0, s, p); break;
        case 3: DCCN_LAUNCH_CHAINS_Z((fewrow_kernel<BKIND, 3, NB, TAG>), dim3(T), dim3(256), 0, s, p); break;
        default: return DCCN_ERR_INVALID_ARG;
    }
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// dense forward + tail of a few-row batch in one launch (NB <= 2); blocks = tiles <= kTailBlocksMax slabs
template <int NB, bool BWD>
static int launch_fewrow_tail(const GemmParams& p, const TailEpiParams& tp, hipStream_t s) {
    const int T = ceil_div(p.M, 16) * (p.N / 16);
    switch (fewrow_ng_c(p, false)) {
        case 18: DCCN_LAUNCH_CHAINS_Z((fewrow_tail_kernel<18, NB, BWD>), dim3(T), dim3(256), 0, s, p, tp); break;
        case 14: DCCN_LAUNCH_CHAINS_Z((fewrow_tail_kernel<14, NB, BWD>), dim3(T), dim3(256), 0, s, p, tp); break;
        case 10: DCCN_LAUNCH_CHAINS_Z((fewrow_tail_kernel<10, NB, BWD>), dim3(T), dim3(256), 0, s, p, tp); break;
        default: return DCCN_ERR_INVALID_ARG;
    }
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

template <int NG, int ACTX>
static int launch_dense_bwd_fewrow_ng(const GemmParams& px, const GemmParams& pw, hipStream_t s) {
    using CW = Cfg16<OP_ICONTIG, OP_ICONTIG, 2, 2, 2, 2, 64, 1>;
    const int nx = ceil_div(px.M, 16) * (px.N / 16);
    const int tw = ceil_div(pw.N, CW::BN) * ceil_div(pw.M, CW::BM);
    if (fewrow_dw_ok(pw)) {                      // the weight gradient on one-latency tiles as well
        auto kern = dense_bwd_fewrow_kernel<NG, ACTX, 2, 2, 2, 2, 64, true>;
        DCCN_TRY(set_smem_attr(kern, kFewrowDwSmem));
        DCCN_LAUNCH_CHAINS_Z(kern, dim3(nx + tw), dim3(256), kFewrowDwSmem, s, px, pw, nx, tw);
    } else {
        const size_t smem = CW::smem_bytes(0);
        auto kern = dense_bwd_fewrow_kernel<NG, ACTX, 2, 2, 2, 2, 64, false>;
        DCCN_TRY(set_smem_attr(kern, smem));
        DCCN_LAUNCH_CHAINS_Z(kern, dim3(nx + tw), dim3(256), smem, s, px, pw, nx, tw);
    }
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
template <int ACTX>
static int launch_dense_bwd_fewrow(const GemmParams& px, const GemmParams& pw, hipStream_t s) {
    switch (fewrow_ng(px)) {
        case 18: return launch_dense_bwd_fewrow_ng<18, ACTX>(px, pw, s);
        case 14: return launch_dense_bwd_fewrow_ng<14, ACTX>(px, pw, s);
        case 10: return launch_dense_bwd_fewrow_ng<10, ACTX>(px, pw, s);
        case 4: return launch_dense_bwd_fewrow_ng<4, ACTX>(px, pw, s);
        case 3: return launch_dense_bwd_fewrow_ng<3, ACTX>(px, pw, s);
        default: return DCCN_ERR_INVALID_ARG;
    }
}

}  // namespace dccn
