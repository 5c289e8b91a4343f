// fp32 GEMM on v_mfma_f32_16x16x4_f32 with small, shape-fitted tiles and several resident blocks per CU.
//
// Why a second GEMM family next to gemm_f32_mfma.h (32x32x2 tiles, one 64x64 / 128x128 tile per block, one wave per
// SIMD): at the sizes of the N=64 receiver step (M = 1170 frames or 8190 symbol rows, N = 128..896) the 32-granular
// tiles leave a quarter of the chip without a tile (dense forward: 190 tiles of 64x64 on 256 CUs) and one wave per SIMD
// has nothing to issue while it waits at the per-k-tile barrier.  Here
//   * the MFMA is 16x16x4 (4 accumulator registers per tile): a wave owns TM x TN such tiles, a block WGM x WGN waves,
//     so block tiles come in steps of 16 (48x64 gives the dense forward 250 blocks for 256 CUs);
//   * KS > 1 puts KS wave sets (KS*256 threads) on one tile, each taking a 1/KS share of every k-tile's 16-deep groups
//     (intra-block split-K): two waves per SIMD even when there is exactly one block per CU; the partial tiles meet in
//     LDS once, after the k-loop, and every wave then finishes its share of the output;
//   * accumulators + fragments stay under 64 VGPRs, LDS per block is 30-65 KB: 2-4 blocks per CU for the grouped
//     launches, whose mixed-length blocks then fill each other's barrier and prologue gaps.
// Operand staging is the scheme of gemm_f32_mfma.h ([i][k] LDS tiles, k contiguous, one ds_read_b128 = the operands of
// four consecutive MFMAs); the row stride is BK+8 floats, which is the conflict-free one for this fragment shape:
// a ds_read_b128 is served in 16-lane groups {0-3,12-15,20-27},{4-11,16-19,28-31} (+32), i.e. 8 rows at k-slot c plus the
// other 8 rows at k-slot c+1; with a stride of s = (BK+8)/4 = 2 mod 4 sixteen-byte slots those hit 16 distinct slots.
// Inside a 16-deep group the MFMA with float4 component j multiplies k = {j, 4+j, 8+j, 12+j} (same permutation on A
// and B, so the sum is unchanged).
//
// EPI_TAIL: the dense forward's epilogue runs the demodulation tail (R3-R6 forward and backward, tail.h tail_cell) on
// the output tile while it is in registers -- adjacent lanes hold the I and Q column of one data cell -- and writes
// dz / prob / the per-block metric and gradient slabs instead of (or besides) z.
#pragma once
#include "gemm_f32_mfma.h"
#include "tail.h"

// GEMM16_ABL: timing ablations for experiments (results are wrong by construction when set):
// bit0 no global loads in the k-loop, bit1 no LDS stores, bit2 no per-k-tile barrier, bit3 no fragment re-reads,
// bit4 no prob store, bit5 no dz store, bit6 no tail evaluation at all (block reduction of zeros only)
#ifndef GEMM16_ABL
#define GEMM16_ABL 0
#endif

namespace dccn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Epilogue16 : int { EPI_STORE = 0, EPI_TAIL = 1 };

struct TailEpiParams {
    const int32_t* bits;            // [M, N/2, NB]
    const float* tailp;
    float* prob;                    // [M, N/2, NB, 2], nullable
    float* dz;                      // [M, N] (training)
    TailBlockMetrics* blk_metrics;  // one slab per block
    float* blk_grads;
    float inv_count;                // 1 / (cells * NB)
    __device__ __forceinline__ TailEpiParams at_chain(const long long off) const {     // chain groups (common.h)
        TailEpiParams q = *this;
        q.bits = chain_at(bits, off); q.tailp = chain_at(tailp, off); q.prob = chain_at(prob, off); q.dz = chain_at(dz, off);
        q.blk_metrics = chain_at(blk_metrics, off); q.blk_grads = chain_at(blk_grads, off);
        return q;
    }
};

// One operand tile: BI rows (the M or N index) x BK columns (k), staged by NT threads.
// A "unit" is one thread's work item: a float4 of 4 consecutive k (KCONTIG / CCONV_W) or a 4(k) x 4(i) block that is
// transposed in registers (ICONTIG).  When the unit count is not a multiple of NT the surplus threads wrap around and
// repeat the first units (same loads, same values to the same LDS addresses): no thread-dependent control flow inside
// the k-loop, so the compiler keeps counted vmcnt/lgkmcnt waits instead of draining at every join.
template <int KIND, int BI, int BK, int NT>
struct Tile16 {
    static_assert(BK % 32 == 0 && BI % 16 == 0, "tile shape");
    static_assert(KIND == OP_ICONTIG || KIND == OP_KCONTIG || KIND == OP_CCONV_W, "operand kind");
    static constexpr int K4 = BK / 4;
    static constexpr int LD = BK + 8;
    static constexpr bool IC = (KIND == OP_ICONTIG);
    static constexpr bool CC = (KIND == OP_CCONV_W);
    static constexpr int UNITS = IC ? (BI / 4) * K4 : BI * K4;
    static constexpr int NU = (UNITS + NT - 1) / NT;
    static constexpr int NV = IC ? 4 * NU : NU;          // float4 registers = global loads = LDS stores per thread
    static constexpr int NOFF = CC ? 2 * NV : NV;
    float4 r[2][NV];            // staging registers; set 1 only exists in the prefetch-distance-2 loop
    unsigned voff[NOFF];        // fast path: byte offsets from the uniform tile base
    unsigned soff[NU];          // LDS float offset of the unit
    unsigned okmask;

    static __device__ __forceinline__ int unit_index(int u, int tid) { return (tid + u * NT) % UNITS; }
    // ICONTIG: the 8 lanes of a ds_write_b128 group hold one i4 and 8 consecutive k4 -> each transposed row they
    // write is 128 contiguous bytes (conflict free for any row stride); the 8 groups of a wave hold 8 adjacent i4,
    // so every global row segment a wave touches is one 128-byte line.
    static __device__ __forceinline__ void ic_coords(int bidx, int& i4, int& k4) {
        const int sub = bidx & 7, rest = bidx >> 3;
        i4 = rest % (BI / 4);
        k4 = (rest / (BI / 4)) * 8 + sub;
    }

    // per-thread constants of both load paths
    __device__ __forceinline__ void init(int ld, int i0, int I, int cF, int tid) {
        okmask = 0u;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = unit_index(u, tid);
            if constexpr (IC) {
                int i4, k4;
                ic_coords(idx, i4, k4);
                soff[u] = (unsigned)((4 * i4) * LD + 4 * k4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    voff[4 * u + e] = (unsigned)(((4 * k4 + e) * ld + min(i0 + 4 * i4, I - 4)) * 4);
            } else {
                soff[u] = (unsigned)((idx / K4) * LD + (idx % K4) * 4);
                const int ic = min(i0 + idx / K4, I - 1);
                const int k = (idx % K4) * 4;
                if constexpr (KIND == OP_KCONTIG) {
                    voff[u] = (unsigned)((ic * ld + k) * 4);
                } else {                                          // element (k,i) = Weff[k][i], column i = ic fixed
                    const int f = ic >> 1, c = ic & 1, n0 = k >> 1;
                    voff[2 * u] = (unsigned)((n0 * 2 * cF + f + (c ? cF : 0)) * 4);       // q = 0 rows, +
                    voff[2 * u + 1] = (unsigned)((n0 * 2 * cF + f + (c ? 0 : cF)) * 4);   // q = 1 rows, negated
                }
            }
        }
    }
    static __device__ __forceinline__ const char* tile_base(const float* p, int ld, int k0, int cF) {
        if constexpr (KIND == OP_ICONTIG) return reinterpret_cast<const char*>(p + (size_t)k0 * ld);
        else if constexpr (KIND == OP_KCONTIG) return reinterpret_cast<const char*>(p + k0);
        else return reinterpret_cast<const char*>(p + (size_t)(k0 >> 1) * 2 * cF);
    }
    // ---- fast path: k-tiles completely inside the k range; address = uniform tile base + per-thread constant ----
    __device__ __forceinline__ void load_fast(int v, const char* __restrict__ base, int cF, int s = 0) {
        if constexpr (!CC) {
            r[s][v] = *reinterpret_cast<const float4*>(base + voff[v]);
        } else {
            const unsigned row = (unsigned)(2 * cF * 4);
            const float a0 = *reinterpret_cast<const float*>(base + voff[2 * v]);
            const float b0 = *reinterpret_cast<const float*>(base + voff[2 * v + 1]);
            const float a1 = *reinterpret_cast<const float*>(base + voff[2 * v] + row);
            const float b1 = *reinterpret_cast<const float*>(base + voff[2 * v + 1] + row);
            r[s][v] = make_float4(a0, -b0, a1, -b1);
        }
    }

    // ---- masked path (ragged last k-tile): clamped addresses, pieces past kend become zero at LDS-write time ----
    // Vector-legal operands only: K % 4 == 0 (KCONTIG / CCONV_W), I % 4 == 0 (ICONTIG), 16-byte aligned rows.
    __device__ __forceinline__ void load_masked(int v, const float* __restrict__ p, int ld, int k0, int kend, int K,
                                                int i0, int I, int cF, int tid) {
        const int u = IC ? v / 4 : v;
        const int idx = unit_index(u, tid);
        if constexpr (IC) {
            int i4, k4;
            ic_coords(idx, i4, k4);
            const int k = k0 + 4 * k4 + (v & 3);
            r[0][v] = *reinterpret_cast<const float4*>(p + (size_t)min(k, K - 1) * ld + min(i0 + 4 * i4, I - 4));
            okmask = (okmask & ~(1u << v)) | ((k < kend ? 1u : 0u) << v);
        } else {
            const int ic = min(i0 + idx / K4, I - 1);
            const int k = k0 + (idx % K4) * 4;
            const int kc = min(k, K - 4);
            if constexpr (KIND == OP_KCONTIG) {
                r[0][v] = *reinterpret_cast<const float4*>(p + (size_t)ic * ld + kc);
            } else {
                r[0][v] = make_float4(cconv_weff(p, cF, kc, ic), cconv_weff(p, cF, kc + 1, ic), cconv_weff(p, cF, kc + 2, ic),
                                   cconv_weff(p, cF, kc + 3, ic));
            }
            okmask = (okmask & ~(1u << v)) | ((k < kend ? 1u : 0u) << v);
        }
    }

    // v must be a compile-time constant after unrolling
    template <bool MASKED>
    __device__ __forceinline__ void store_piece(int v, float* __restrict__ lds, int s = 0) const {
        if constexpr (IC) {
            const int b = v & ~3, q = v & 3;               // row i = 4*i4 + q of the transposed block
            float4 val = make_float4(f4c(r[s][b + 0], q), f4c(r[s][b + 1], q), f4c(r[s][b + 2], q), f4c(r[s][b + 3], q));
            if constexpr (MASKED) {
                val.x = ((okmask >> (b + 0)) & 1u) ? val.x : 0.f;
                val.y = ((okmask >> (b + 1)) & 1u) ? val.y : 0.f;
                val.z = ((okmask >> (b + 2)) & 1u) ? val.z : 0.f;
                val.w = ((okmask >> (b + 3)) & 1u) ? val.w : 0.f;
            }
            *reinterpret_cast<float4*>(lds + soff[v / 4] + q * LD) = val;
        } else {
            float4 val = r[s][v];
            if constexpr (MASKED) {
                const bool ok = (okmask >> v) & 1u;
                val = make_float4(ok ? val.x : 0.f, ok ? val.y : 0.f, ok ? val.z : 0.f, ok ? val.w : 0.f);
            }
            *reinterpret_cast<float4*>(lds + soff[v]) = val;
        }
    }
};

template <int KA, int KB, int WGM, int WGN, int TM, int TN, int BK, int KS>
struct Cfg16 {
    static constexpr int NT = 256 * KS;
    static constexpr int BM = WGM * TM * 16, BN = WGN * TN * 16;
    using TA = Tile16<KA, BM, BK, NT>;
    using TB = Tile16<KB, BN, BK, NT>;
    static constexpr size_t tile_bytes = (size_t)(2 * BM * TA::LD + 2 * BN * TB::LD) * sizeof(float);
    // after the k-loop the tile buffers are reused: partial-tile exchange (KS > 1) and the tail's block reduction
    static constexpr size_t xchg_bytes = KS > 1 ? (size_t)KS * 4 * TM * TN * 4 * 64 * sizeof(float) : 0;
    static constexpr size_t smem_bytes(int tail_floats) {
        const size_t e = xchg_bytes + (size_t)tail_floats * sizeof(float) + 16;
        return tile_bytes > e ? tile_bytes : e;
    }
};

// One BM x BN output tile (block `L` of `T` tiles, grid split `z`) of C = A.B
// PD = 2: global loads run TWO k-tiles ahead of the MFMAs (second staging register set, loop unrolled by two): with a
// single block per CU nothing else covers the ~0.6 us load latency, which is longer than half a 48x64x64 tile's MFMA
// time; needs the whole k range on the fast loaders (else the block falls back to the PD = 1 schedule).
template <int KA, int KB, int WGM, int WGN, int TM, int TN, int BK, int KS, int COLSUM, int EPI, int NB, bool BWD,
          int PD = 1>
__device__ __forceinline__ void gemm16_block(const GemmParams& p, const TailEpiParams& tp, const int L, const int T,
                                             const int z, const int slab) {
    static_assert(WGM * WGN == 4, "four waves per wave set");
    using CF = Cfg16<KA, KB, WGM, WGN, TM, TN, BK, KS>;
    using TA = typename CF::TA;
    using TB = typename CF::TB;
    constexpr int NT = CF::NT, BM = CF::BM, BN = CF::BN;
    constexpr int LDA = TA::LD, LDB = TB::LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                       // [2][BM][LDA]
    float* sB = smem + 2 * BM * LDA;        // [2][BN][LDB]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = (tid >> 6) & 3, set = tid >> 8;
    const int l15 = lane & 15, kg = lane >> 4;
    const int wm0 = (wid / WGN) * (TM * 16), wn0 = (wid % WGN) * (TN * 16);
    // XCD-aware tile order (block b runs on XCD b % 8; speed only): each XCD gets a contiguous run of row-major tiles
    const int ntn = (p.N + BN - 1) / BN;
    int tile;
    {
        const int xcd = L & 7, j = L >> 3, q = T >> 3, r = T & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int kbeg = z * p.klen;
    const int kend = min(p.K, kbeg + p.klen);
    const int ntiles = (kend - kbeg + BK - 1) / BK;
    const int nfull = (kend - kbeg) / BK;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;

    float cs = 0.f;   // column sum of the B rows (COLSUM)
    const bool do_cs = COLSUM && p.colsum != nullptr && m0 == 0 && tid < BN;

    // EPI_TAIL: the label bits of this lane's cells are requested before the k-loop (no exposed latency afterwards).
    // Cell (tile a,b ; c): row 4*kg + (odd ? 2+c : c) of the tile, data cell col/2; KS == 2: set s owns c == s.
    constexpr int NCELL = (EPI == EPI_TAIL) ? TM * TN * 2 / KS : 1;
    // NB >= 3: the tail does not run on the MFMA register layout (90 / 200 gradient accumulators per cell-lane leave no
    // room for it): the finished tile goes through LDS and the block walks its BM x BN/2 cells like the stand-alone
    // tail kernels do -- a lane per cell (8-QAM, 16-QAM evaluation) or a quad of lanes per cell (16-QAM training,
    // tail.h tail_quad4_cell).  Each thread requests the label bits of CPT cells (cell c = tid + NT i of the tile, row
    // major) before the k-loop and keeps them packed, one int per cell.
    constexpr bool TAIL_LDS = (EPI == EPI_TAIL) && NB >= 3;
    constexpr bool QUAD4 = TAIL_LDS && NB == 4 && BWD;
    static_assert(!TAIL_LDS || KS == 1, "LDS-staged tail: one wave set");
    constexpr int TCELLS = BM * (BN / 2);
    constexpr int CPT = TAIL_LDS ? (TCELLS + NT - 1) / NT : 1;
    int plab[CPT];
    if constexpr (TAIL_LDS) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = min(tid + NT * i, TCELLS - 1);
            const int row = min(m0 + c / (BN / 2), p.M - 1), dc = min((n0 >> 1) + c % (BN / 2), (p.N >> 1) - 1);
            const int32_t* lb = tp.bits + ((long long)row * (p.N >> 1) + dc) * NB;
            int v = 0;
#pragma unroll
            for (int j = 0; j < NB; ++j) v |= (lb[j] != 0 ? 1 : 0) << j;
            plab[i] = v;
        }
    }
    int labs[NCELL][NB];                                 // lane-cell labels of the register-layout tail (NB <= 2)
    if constexpr (EPI == EPI_TAIL && !TAIL_LDS) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int cc = 0; cc < 2 / KS; ++cc) {
                    const int c = KS > 1 ? set : cc;
                    const int row = min(m0 + wm0 + a * 16 + 4 * kg + ((lane & 1) ? 2 + c : c), p.M - 1);
                    const int col = min(n0 + wn0 + b * 16 + l15, p.N - 1);
                    const long long cell = (long long)row * (p.N >> 1) + (col >> 1);
#pragma unroll
                    for (int j = 0; j < NB; ++j) labs[(a * TN + b) * (2 / KS) + cc][j] = tp.bits[cell * NB + j];
                }
    }

    // EPI_TAIL: this lane's bias values are requested before the k-loop as well
    float bjv[(EPI == EPI_TAIL) ? TN : 1];
    if constexpr (EPI == EPI_TAIL) {
#pragma unroll
        for (int b = 0; b < TN; ++b) bjv[b] = p.bias != nullptr ? p.bias[min(n0 + wn0 + b * 16 + l15, p.N - 1)] : 0.f;
    }

    TA ta;
    TB tb;
    ta.init(p.lda, m0, p.M, p.cF, tid);
    tb.init(p.ldb, n0, p.N, p.cF, tid);
    constexpr int PA = TA::NV, PB = TB::NV;
    constexpr int NG = BK / 16;                          // groups of 16 k per k-tile
    static_assert(NG % KS == 0, "k-tile groups must split evenly over the wave sets");
    constexpr int NGS = NG / KS;                         // groups per wave set
    constexpr int NSTEP = NGS * 4;                       // MFMA steps (of TM*TN MFMAs) per k-tile and wave
    constexpr int HALF = NSTEP / 2;
    constexpr int LPS = (PA + PB + HALF - 1) / HALF;     // global loads (first half) / LDS stores (second half) per step

    auto load_ab = [&](auto mode_tag, int q, int k0) {
        constexpr int MODE = decltype(mode_tag)::value;
        if (q < PA) {
            if constexpr (MODE == PF_FAST) ta.load_fast(q, TA::tile_base(p.A, p.lda, k0, p.cF), p.cF);
            else ta.load_masked(q, p.A, p.lda, k0, kend, p.K, m0, p.M, p.cF, tid);
        } else {
            if constexpr (MODE == PF_FAST) tb.load_fast(q - PA, TB::tile_base(p.B, p.ldb, k0, p.cF), p.cF);
            else tb.load_masked(q - PA, p.B, p.ldb, k0, kend, p.K, n0, p.N, p.cF, tid);
        }
    };
    auto store_ab = [&](auto mode_tag, int q, float* An, float* Bn) {
        constexpr int MODE = decltype(mode_tag)::value;
        if (q < PA) ta.template store_piece<MODE == PF_MASKED>(q, An);
        else tb.template store_piece<MODE == PF_MASKED>(q - PA, Bn);
    };
    auto stage_first = [&](auto mode_tag) {
#pragma unroll
        for (int q = 0; q < PA + PB; ++q) load_ab(mode_tag, q, kbeg);
#pragma unroll
        for (int q = 0; q < PA + PB; ++q) store_ab(mode_tag, q, sA, sB);
    };
    if (ntiles > 0) {
        if (nfull > 0) stage_first(std::integral_constant<int, PF_FAST>{});
        else stage_first(std::integral_constant<int, PF_MASKED>{});
    }
    __syncthreads();

    // k-tile body: NSTEP steps of TM*TN MFMAs; the operand fragments of the next 16-deep group are read while the
    // current group's MFMAs run; the global loads of k-tile t+DIST go out in the first half of the steps (register set
    // LSET) and the registers holding k-tile t+1 (set SSET) are written to the other LDS buffer in the second half.
    int t = 0;
    auto ktile2 = [&](auto load_tag, auto store_tag, auto lset_tag, auto sset_tag, auto dist_tag) {
        constexpr int LMODE = decltype(load_tag)::value, SMODE = decltype(store_tag)::value;
        constexpr int LSET = decltype(lset_tag)::value, SSET = decltype(sset_tag)::value, DIST = decltype(dist_tag)::value;
        const int cur = t & 1;
        const int k0n = kbeg + (t + DIST) * BK;
        const float* As = sA + cur * BM * LDA + (wm0 + l15) * LDA + 4 * kg + 16 * NGS * set;
        const float* Bs = sB + cur * BN * LDB + (wn0 + l15) * LDB + 4 * kg + 16 * NGS * set;
        float* An = sA + (cur ^ 1) * BM * LDA;
        float* Bn = sB + (cur ^ 1) * BN * LDB;
        float4 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) fa[0][a] = *reinterpret_cast<const float4*>(As + a * 16 * LDA);
#pragma unroll
        for (int b = 0; b < TN; ++b) fb[0][b] = *reinterpret_cast<const float4*>(Bs + b * 16 * LDB);
#pragma unroll
        for (int g = 0; g < NGS; ++g) {
            if (g + 1 < NGS && !(GEMM16_ABL & 8)) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
                    fa[(g + 1) & 1][a] = *reinterpret_cast<const float4*>(As + a * 16 * LDA + 16 * (g + 1));
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    fb[(g + 1) & 1][b] = *reinterpret_cast<const float4*>(Bs + b * 16 * LDB + 16 * (g + 1));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int step = g * 4 + j;
                if constexpr (LMODE != PF_NONE && !(GEMM16_ABL & 1)) {
                    if (step < HALF) {
#pragma unroll
                        for (int q = step * LPS; q < (step + 1) * LPS && q < PA + PB; ++q) {
                            if constexpr (LMODE == PF_FAST) {
                                if (q < PA) ta.load_fast(q, TA::tile_base(p.A, p.lda, k0n, p.cF), p.cF, LSET);
                                else tb.load_fast(q - PA, TB::tile_base(p.B, p.ldb, k0n, p.cF), p.cF, LSET);
                            } else {
                                load_ab(load_tag, q, k0n);
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(fa[g & 1][a], j), f4c(fb[g & 1][b], j),
                                                                         acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (SMODE != PF_NONE && !(GEMM16_ABL & 2)) {
                    if (step >= HALF) {
#pragma unroll
                        for (int q = (step - HALF) * LPS; q < (step - HALF + 1) * LPS && q < PA + PB; ++q) {
                            if (q < PA) ta.template store_piece<SMODE == PF_MASKED>(q, An, SSET);
                            else tb.template store_piece<SMODE == PF_MASKED>(q - PA, Bn, SSET);
                        }
                    }
                }
            }
        }
        if (COLSUM && do_cs) {
            const float* Bc = sB + cur * BN * LDB + tid * LDB;
#pragma unroll
            for (int k = 0; k < BK; k += 4) {
                const float4 v = *reinterpret_cast<const float4*>(Bc + k);
                cs += (v.x + v.y) + (v.z + v.w);
            }
        }
        if constexpr (!(GEMM16_ABL & 4)) __syncthreads();
        ++t;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using TF = std::integral_constant<int, PF_FAST>;
    using TM_ = std::integral_constant<int, PF_MASKED>;
    using TN_ = std::integral_constant<int, PF_NONE>;
    bool deep = false;
    if constexpr (PD == 2) deep = nfull == ntiles && ntiles >= 2;
    if (deep) {
        // k-tile n travels through register set n & 1; tile 1 is already in flight when the first MFMA issues
#pragma unroll
        for (int q = 0; q < PA + PB; ++q) {
            if (q < PA) ta.load_fast(q, TA::tile_base(p.A, p.lda, kbeg + BK, p.cF), p.cF, 1);
            else tb.load_fast(q - PA, TB::tile_base(p.B, p.ldb, kbeg + BK, p.cF), p.cF, 1);
        }
        while (t + 3 < ntiles) {                              // t even here
            ktile2(TF{}, TF{}, I0{}, I1{}, I2{});
            ktile2(TF{}, TF{}, I1{}, I0{}, I2{});
        }
        const int rem = ntiles - t;                           // 1..3 tiles left, t even
        if (rem == 3) {
            ktile2(TF{}, TF{}, I0{}, I1{}, I2{});
            ktile2(TN_{}, TF{}, I1{}, I0{}, I2{});
            ktile2(TN_{}, TN_{}, I0{}, I0{}, I2{});
        } else if (rem == 2) {
            ktile2(TN_{}, TF{}, I0{}, I1{}, I2{});
            ktile2(TN_{}, TN_{}, I0{}, I0{}, I2{});
        } else {
            ktile2(TN_{}, TN_{}, I0{}, I0{}, I2{});
        }
    } else {
        while (t + 1 < nfull) ktile2(TF{}, TF{}, I0{}, I0{}, I1{});
        while (t + 1 < ntiles) ktile2(TM_{}, TM_{}, I0{}, I0{}, I1{});
        if (ntiles > 0) ktile2(TN_{}, TN_{}, I0{}, I0{}, I1{});
    }

    if (COLSUM && do_cs) {
        const int col = n0 + tid;
        if (col < p.N) p.colsum[(size_t)z * p.N + col] = cs;
    }

    // ---- the partial tiles of the KS wave sets meet in LDS (the k-loop ended with a barrier: buffers are free) ----
    // layout [set][wave][tile][reg][lane]: lane-contiguous, conflict free
    float* xch = smem;
    if constexpr (KS > 1) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    xch[(((set * 4 + wid) * (TM * TN) + a * TN + b) * 4 + r) * 64 + lane] = acc[a][b][r];
        __syncthreads();
    }
    // value of output element (tile a,b ; register r) as held by lane `ln` of this wave position: the fixed-order sum of
    // the sets' partials (KS > 1) or this wave's own accumulator (KS == 1, ln must be the calling lane)
    auto tile_val = [&](int a, int b, int r, int ln) -> float {
        if constexpr (KS > 1) {
            float v = xch[(((0 * 4 + wid) * (TM * TN) + a * TN + b) * 4 + r) * 64 + ln];
#pragma unroll
            for (int s2 = 1; s2 < KS; ++s2) v += xch[(((s2 * 4 + wid) * (TM * TN) + a * TN + b) * 4 + r) * 64 + ln];
            return v;
        } else {
            return acc[a][b][r];
        }
    };
    static_assert(KS == 1 || KS == 2, "wave sets");

    // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r
    if constexpr (EPI == EPI_STORE) {
        float* Cz = p.C + (size_t)z * p.slab;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int col = n0 + wn0 + b * 16 + l15;
                float bj = 0.f;
                if (p.bias != nullptr && col < p.N) {
                    if (p.cbias) {
                        const float d = p.bias[col >> 1] - p.bias[p.cF + (col >> 1)];
                        bj = (col & 1) ? -d : d;
                    } else {
                        bj = p.bias[col];
                    }
                }
                // KS == 2: set s finishes registers 2s, 2s+1 of every tile
#pragma unroll
                for (int rr = 0; rr < 4 / KS; ++rr) {
                    const int r = KS > 1 ? 2 * set + rr : rr;
                    const int row = m0 + wm0 + a * 16 + 4 * kg + r;
                    float v;
                    if constexpr (KS > 1) v = set == 0 ? tile_val(a, b, rr, lane) : tile_val(a, b, 2 + rr, lane);
                    else v = acc[a][b][rr];
                    if constexpr (NB == 5) {
                        // the channel estimate h leaves the GEMM together with what model.py:431-438 computes from it:
                        // eq = y conj(h)/|h| and corr = eq conj(eq).  Adjacent lanes hold re and im of one cell: they
                        // swap halves, the even lane stores the cell of eq, the odd lane that of corr (aux = y, out2 = eq,
                        // out3 = corr, all [M, N] like C).  Same expressions as equalizer.h equalize_one: same bits.
                        static_assert(KS == 1, "equalise stage: one wave set");
                        const float o = v + bj;
                        const float partner = __shfl_xor(o, 1, 64);
                        if (row < p.M && col < p.N) {
                            const size_t ci = (size_t)row * p.ldc + col, c0 = ci - (size_t)(col & 1);
                            Cz[ci] = o;
                            const float hr = (col & 1) ? partner : o, hi = (col & 1) ? o : partner;
                            const float2 yv = *reinterpret_cast<const float2*>(p.aux + c0);
                            const float aa = sqrtf(hr * hr + hi * hi);
                            const float cr = hr / aa, cim = (-hi) / aa;
                            const float er = yv.x * cr - yv.y * cim, ei = yv.x * cim + yv.y * cr;
                            if ((col & 1) == 0) *reinterpret_cast<float2*>(p.out2 + c0) = make_float2(er, ei);
                            else *reinterpret_cast<float2*>(p.out3 + c0) = make_float2(er * er - ei * (-ei), er * (-ei) + ei * er);
                        }
                    } else
                    if (row < p.M && col < p.N) {
                        // EPI_STORE: NB selects an element-wise stage on the way out (compile time: the plain store pays
                        // nothing): the equaliser's tanh, its gradient, and "add to what is there" (model.py:424, 393-462)
                        const size_t ci = (size_t)row * p.ldc + col;
                        float o = v + bj;
                        if constexpr (NB == 2) o = tanhf(o);
                        else if constexpr (NB == 3) { const float y = p.aux[ci]; o = o * (1.0f - y * y); }
                        else if constexpr (NB == 4) o = p.aux[ci] + o;
                        Cz[ci] = o;
                    }
                }
            }
        }
    } else {
        // ---- fused demodulation tail ------------------------------------------------------------------------
        // Lanes 2d, 2d+1 hold columns 2d (I) and 2d+1 (Q) of data cell d for 4 rows each.  KS == 1: the pair swaps
        // values on the DPP crossbar, the even lane takes the cells of registers 0,1, the odd lane those of 2,3.
        // KS == 2: every lane reads both columns of its cell from the exchange buffer; set s takes cell s of each
        // lane's two.  Either way a lane ends up with TM*TN*2/KS cells.
        const int odd = lane & 1;
        const int Dn = p.N >> 1;                           // cells per row
        if constexpr (TAIL_LDS) {
            // ---- tile -> LDS, then the tail over the tile's cells (see the label prefetch above) ----
            constexpr int PW = tail_param_count(NB);
            constexpr int ZS = BN + 2;                          // row stride of the staged tile (even: float2 reads)
            float* zt = smem;                                   // [BM][ZS]
            int* lt = reinterpret_cast<int*>(smem + BM * ZS);   // [TCELLS] packed label bits
            float* swl = smem + BM * ZS + TCELLS;               // [PW] tail weights (90 / 200 do not fit the SGPR file)
            float* red = swl + ((PW + 3) & ~3);
#pragma unroll
            for (int a = 0; a < TM; ++a) {
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int col = n0 + wn0 + b * 16 + l15;
                    const float bj = bjv[b];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[a][b][r] + bj;
                        const int rl = wm0 + a * 16 + 4 * kg + r;
                        zt[rl * ZS + wn0 + b * 16 + l15] = v;
                        if (p.C != nullptr && m0 + rl < p.M && col < p.N) p.C[(size_t)(m0 + rl) * p.ldc + col] = v;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < CPT; ++i)
                if (tid + NT * i < TCELLS) lt[tid + NT * i] = plab[i];
            for (int i = tid; i < PW; i += NT) swl[i] = tp.tailp[i];
            __syncthreads();
            if constexpr (QUAD4) {
                const int q = lane & 3;
                TailQuad4W W;
                W.load(swl, q);                                  // this lane's 92 weights: registers for the whole loop
                TailQuad4Acc A;
                A.clear();
                for (int c = tid >> 2; c < TCELLS; c += NT / 4) {
                    const int rl = c / (BN / 2), dl = c % (BN / 2);
                    const int row = m0 + rl, dcol = (n0 >> 1) + dl;
                    const bool ok = row < p.M && dcol < Dn;
                    const float2 zv = *reinterpret_cast<const float2*>(zt + rl * ZS + 2 * dl);
                    const long long cell = (long long)min(row, p.M - 1) * Dn + min(dcol, Dn - 1);
                    float* pq = (tp.prob != nullptr && ok) ? tp.prob + (cell * NB + q) * 2 : nullptr;
                    const float2 d = tail_quad4_cell(zv.x, zv.y, (lt[c] >> q) & 1, ok, W, tp.inv_count, pq, q, A);
                    if (q == 0 && ok) *reinterpret_cast<float2*>(tp.dz + cell * 2) = d;
                }
                tail_quad4_block_reduce<NT>(A, red, tp.blk_metrics, tp.blk_grads, slab);
            } else {
                TailLaneAcc<NB, BWD> A;
                A.clear();
                for (int c = tid; c < TCELLS; c += NT) {
                    asm volatile("" ::: "memory");
                    const int rl = c / (BN / 2), dl = c % (BN / 2);
                    const int row = m0 + rl, dcol = (n0 >> 1) + dl;
                    const float2 zv = *reinterpret_cast<const float2*>(zt + rl * ZS + 2 * dl);
                    const int pk = lt[c];
                    const float a0[1] = {zv.x}, a1[1] = {zv.y};
                    int lb[1][NB];
#pragma unroll
                    for (int j = 0; j < NB; ++j) lb[0][j] = (pk >> j) & 1;
                    const bool vd[1] = {row < p.M && dcol < Dn};
                    const long long cell = (long long)min(row, p.M - 1) * Dn + min(dcol, Dn - 1);
                    float* const pc[1] = {(tp.prob != nullptr && vd[0]) ? tp.prob + cell * NB * 2 : nullptr};
                    float2 dv[1];
                    tail_cells<NB, BWD, 1>(a0, a1, lb, vd, swl, tp.inv_count, pc, A, dv);
                    if constexpr (BWD) {
                        if (vd[0]) *reinterpret_cast<float2*>(tp.dz + cell * 2) = dv[0];
                    }
                }
                tail_block_reduce<NB, BWD, NT>(A, red, tp.blk_metrics, tp.blk_grads, slab);
            }
            return;
        }
        const float* __restrict__ sw = tp.tailp;
        float* red = smem + (KS > 1 ? (int)(CF::xchg_bytes / sizeof(float)) : 0);
        TailLaneAcc<NB, BWD> A;
        A.clear();
        float cz0[NCELL], cz1[NCELL];
        bool cvalid[NCELL];
        float* cprob[NCELL];
        float* cdz[NCELL];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int col = n0 + wn0 + b * 16 + l15;   // this lane's own column
                const int colc = min(col, p.N - 1);
                const float bj = bjv[b];
                float own[4], oth[4];
                if constexpr (KS == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        own[r] = acc[a][b][r] + bj;
                        oth[r] = __builtin_bit_cast(float, dpp_mov_i32(__builtin_bit_cast(int, own[r]), 0));
                    }
                    if (p.C != nullptr) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = m0 + wm0 + a * 16 + 4 * kg + r;
                            if (row < p.M && col < p.N) p.C[(size_t)row * p.ldc + col] = own[r];
                        }
                    }
                } else {
                    const float bo = __builtin_bit_cast(float, dpp_mov_i32(__builtin_bit_cast(int, bj), 0));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        own[r] = tile_val(a, b, r, lane) + bj;
                        oth[r] = tile_val(a, b, r, lane ^ 1) + bo;
                    }
                    if (p.C != nullptr && set == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = m0 + wm0 + a * 16 + 4 * kg + r;
                            if (row < p.M && col < p.N) p.C[(size_t)row * p.ldc + col] = own[r];
                        }
                    }
                }
#pragma unroll
                for (int cc = 0; cc < 2 / KS; ++cc) {
                    const int ci = (a * TN + b) * (2 / KS) + cc;
                    float mine, othr;
                    int rsel;
                    if constexpr (KS > 1) {              // c = set
                        mine = set ? (odd ? own[3] : own[1]) : (odd ? own[2] : own[0]);
                        othr = set ? (odd ? oth[3] : oth[1]) : (odd ? oth[2] : oth[0]);
                        rsel = (odd ? 2 : 0) + set;
                    } else {
                        mine = odd ? own[2 + cc] : own[cc];
                        othr = odd ? oth[2 + cc] : oth[cc];
                        rsel = (odd ? 2 : 0) + cc;
                    }
                    cz0[ci] = odd ? othr : mine;
                    cz1[ci] = odd ? mine : othr;
                    const int row = m0 + wm0 + a * 16 + 4 * kg + rsel;
                    const bool ok = row < p.M && col < p.N;
                    const long long cell = (long long)min(row, p.M - 1) * Dn + (colc >> 1);
                    cvalid[ci] = ok;
                    cprob[ci] = (tp.prob != nullptr && ok) ? tp.prob + cell * NB * 2 : nullptr;
                    cdz[ci] = BWD ? tp.dz + cell * 2 : nullptr;
                }
            }
        }
        // all cells of the lane in one interleaved instruction stream (tail.h tail_cells)
        float2 cdzv[NCELL];
        if constexpr (GEMM16_ABL & 16) {
#pragma unroll
            for (int ci = 0; ci < NCELL; ++ci) cprob[ci] = nullptr;
        }
        // (at most 8 cells share one interleaved instruction stream: larger tiles run the tail in batches -- same
        // per-cell arithmetic and the same accumulation order as one batch)
        if constexpr (!(GEMM16_ABL & 64)) {
            constexpr int TW = NCELL <= 8 ? NCELL : (NCELL % 8 == 0 ? 8 : (NCELL % 6 == 0 ? 6 : (NCELL % 5 == 0 ? 5 : 4)));
            static_assert(NCELL % TW == 0, "tail batches");
            if constexpr (TW == NCELL) {
                tail_cells<NB, BWD, NCELL>(cz0, cz1, labs, cvalid, sw, tp.inv_count, cprob, A, cdzv);
            } else {
#pragma unroll
                for (int g0 = 0; g0 < NCELL; g0 += TW) {
                    float a0[TW], a1[TW];
                    int lb[TW][NB];
                    bool vd[TW];
                    float* pc[TW];
                    float2 dv[TW];
#pragma unroll
                    for (int u = 0; u < TW; ++u) {
                        a0[u] = cz0[g0 + u]; a1[u] = cz1[g0 + u]; vd[u] = cvalid[g0 + u]; pc[u] = cprob[g0 + u];
#pragma unroll
                        for (int j = 0; j < NB; ++j) lb[u][j] = labs[g0 + u][j];
                    }
                    tail_cells<NB, BWD, TW>(a0, a1, lb, vd, sw, tp.inv_count, pc, A, dv);
#pragma unroll
                    for (int u = 0; u < TW; ++u) cdzv[g0 + u] = dv[u];
                }
            }
        }
        if constexpr (BWD && !(GEMM16_ABL & 32)) {
#pragma unroll
            for (int ci = 0; ci < NCELL; ++ci)
                if (cvalid[ci]) out_store2<1>(cdz[ci], cdzv[ci].x, cdzv[ci].y);
        }
        // (KS == 1: the k-loop ended with a barrier, nobody reads the tile buffers any more; KS > 1: red lies behind the
        // exchange region, which is only read above)
        tail_block_reduce<NB, BWD, NT>(A, red, tp.blk_metrics, tp.blk_grads, slab);
    }
}

// TAG only makes the symbol unique per call site so profiles attribute time to the right operator
template <int KA, int KB, int WGM, int WGN, int TM, int TN, int BK, int KS, int COLSUM, int EPI, int NB, bool BWD, int TAG,
          int PD = 1>
__global__ __launch_bounds__(256 * KS) void gemm16_kernel(const GemmParams p, const TailEpiParams tp) {
    stamp_mark(p.stamp, 0);
    gemm16_block<KA, KB, WGM, WGN, TM, TN, BK, KS, COLSUM, EPI, NB, BWD, PD>(p, tp, (int)blockIdx.x, (int)gridDim.x,
                                                                        (int)blockIdx.z, (int)blockIdx.x);
    stamp_mark(p.stamp, 1);
}

// dense backward: dX = dY.W^T tiles and the split-K slabs of dW = X^T.dY in ONE grid (independent GEMMs sharing dY)
template <int WGM, int WGN, int TM, int TN, int BK, int WWGM, int WWGN, int WTM, int WTN, int ACTX = 1>
__global__ __launch_bounds__(256) void dense_bwd16_kernel(const GemmParams px, const GemmParams pw, const int nx,
                                                          const int tw) {
    const int b = (int)blockIdx.x;
    TailEpiParams none{};
    if (b < nx) {
        gemm16_block<OP_KCONTIG, OP_KCONTIG, WGM, WGN, TM, TN, BK, 1, 0, EPI_STORE, ACTX, false>(px, none, b, nx, 0, 0);
    } else {
        const int c = b - nx;
        gemm16_block<OP_ICONTIG, OP_ICONTIG, WWGM, WWGN, WTM, WTN, BK, 1, 1, EPI_STORE, 1, false>(pw, none, c % tw, tw,
                                                                                                 c / tw, 0);
    }
}

// split-K weight-gradient GEMM (both operands i-contiguous, bias column sums) with the tail's slab reduction riding on
// extra blocks of the same launch
template <int WGM, int WGN, int TM, int TN, int BK>
__global__ __launch_bounds__(256) void gemm16_bwd_w_finalize_kernel(const GemmParams p, int tiles, int gemm_blocks,
                                                                    TailFinalizeArgs a) {
    const int b = (int)blockIdx.x;
    if (b < gemm_blocks) {
        TailEpiParams none{};
        gemm16_block<OP_ICONTIG, OP_ICONTIG, WGM, WGN, TM, TN, BK, 1, 1, EPI_STORE, 1, false>(p, none, b % tiles, tiles,
                                                                                             b / tiles, 0);
    } else {
        demod_tail_finalize_body(a, b - gemm_blocks);
    }
}

template <typename K>
static int set_smem_attr(K kern, size_t smem) {         // once per (device, kernel, size): common.h
    return set_max_dynamic_smem(reinterpret_cast<const void*>(kern), smem);
}

// plain launch of one configuration; smem_pad lets a caller force fewer resident blocks per CU (experiments)
template <int KA, int KB, int WGM, int WGN, int TM, int TN, int BK, int KS, int COLSUM, int TAG, int PD = 1, int ACT = 1>
static int launch_gemm16(const GemmParams& p, int splits, hipStream_t s, size_t smem_min = 0) {
    using CF = Cfg16<KA, KB, WGM, WGN, TM, TN, BK, KS>;
    auto kern = gemm16_kernel<KA, KB, WGM, WGN, TM, TN, BK, KS, COLSUM, EPI_STORE, ACT, false, TAG, PD>;
    size_t smem = CF::smem_bytes(0);
    if (smem < smem_min) smem = smem_min;
    DCCN_TRY(set_smem_attr(kern, smem));
    TailEpiParams none{};
    dim3 grid(ceil_div(p.N, CF::BN) * ceil_div(p.M, CF::BM), 1, splits);
    DCCN_NO_CHAINS();
    hipLaunchKernelGGL(kern, grid, dim3(CF::NT), smem, s, p, none);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// dense forward + fused tail
template <int WGM, int WGN, int TM, int TN, int BK, int KS, int NB, bool BWD, int PD = 1>
static int launch_dense_tail16(const GemmParams& p, const TailEpiParams& tp, hipStream_t s, size_t smem_min = 0) {
    using CF = Cfg16<OP_KCONTIG, OP_ICONTIG, WGM, WGN, TM, TN, BK, KS>;
    auto kern = gemm16_kernel<OP_KCONTIG, OP_ICONTIG, WGM, WGN, TM, TN, BK, KS, 0, EPI_TAIL, NB, BWD, TAG_DENSE_FWD, PD>;
    // after the k-loop: NB >= 3: [staged tile][packed labels][tail weights] + the block reduction's scratch
    constexpr int wfl = NB >= 3 ? CF::BM * (CF::BN + 2) + CF::BM * (CF::BN / 2) + ((tail_param_count(NB) + 3) & ~3) : 0;
    constexpr int rfl = (NB == 4 && BWD) ? tail_quad4_lds_floats(CF::NT) : tail_reduce_lds_floats<NB, BWD>(CF::NT);
    size_t smem = CF::smem_bytes(wfl + rfl);
    if (smem < smem_min) smem = smem_min;
    DCCN_TRY(set_smem_attr(kern, smem));
    dim3 grid(ceil_div(p.N, CF::BN) * ceil_div(p.M, CF::BM), 1, 1);
    DCCN_NO_CHAINS();
    hipLaunchKernelGGL(kern, grid, dim3(CF::NT), smem, s, p, tp);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// Large layers whose row count leaves a short last row tile (N = 1024: 585 frames = 7 x 80 + 25): the last tile row runs
// TM2 x 16 rows instead of a mostly empty TM1 x 16 one -- both tile shapes in ONE grid, the long blocks first (they are the
// critical path; the short ones fill the second round).  p1 / t1: rows [0, M1); p2 / t2: the rest, pointers already offset.
template <int TM1, int TM2, int BK, int NB, bool BWD, int PD>
__global__ __launch_bounds__(256) void dense_tail_ragged_kernel(const GemmParams p1, const TailEpiParams t1, const GemmParams p2,
                                                                const TailEpiParams t2, const int T1, const int T2) {
    const int b = (int)blockIdx.x;
    if (b < T1) gemm16_block<OP_KCONTIG, OP_ICONTIG, 1, 4, TM1, 1, BK, 1, 0, EPI_TAIL, NB, BWD, PD>(p1, t1, b, T1, 0, b);
    else gemm16_block<OP_KCONTIG, OP_ICONTIG, 1, 4, TM2, 1, BK, 1, 0, EPI_TAIL, NB, BWD, PD>(p2, t2, b - T1, T2, 0, b);
}
template <int TM, int BK, int NB, bool BWD>
constexpr size_t dense_tail16_smem() {
    using CF = Cfg16<OP_KCONTIG, OP_ICONTIG, 1, 4, TM, 1, BK, 1>;
    constexpr int wfl = NB >= 3 ? CF::BM * (CF::BN + 2) + CF::BM * (CF::BN / 2) + ((tail_param_count(NB) + 3) & ~3) : 0;
    constexpr int rfl = (NB == 4 && BWD) ? tail_quad4_lds_floats(CF::NT) : tail_reduce_lds_floats<NB, BWD>(CF::NT);
    return CF::smem_bytes(wfl + rfl);
}
template <int TM1, int TM2, int BK, int NB, bool BWD, int PD>
static int launch_dense_tail16_ragged(const GemmParams& p1, const TailEpiParams& t1, const GemmParams& p2, const TailEpiParams& t2,
                                      hipStream_t s, size_t smem_min = 0) {
    auto kern = dense_tail_ragged_kernel<TM1, TM2, BK, NB, BWD, PD>;
    size_t smem = dense_tail16_smem<TM1, BK, NB, BWD>();
    if (dense_tail16_smem<TM2, BK, NB, BWD>() > smem) smem = dense_tail16_smem<TM2, BK, NB, BWD>();
    if (smem < smem_min) smem = smem_min;
    DCCN_TRY(set_smem_attr(kern, smem));
    const int T1 = ceil_div(p1.N, 64) * ceil_div(p1.M, 16 * TM1), T2 = ceil_div(p2.N, 64) * ceil_div(p2.M, 16 * TM2);
    DCCN_NO_CHAINS();
    hipLaunchKernelGGL(kern, dim3(T1 + T2), dim3(256), smem, s, p1, t1, p2, t2, T1, T2);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

template <int WGM, int WGN, int TM, int TN, int BK, int WWGM, int WWGN, int WTM, int WTN, int ACTX = 1>
static int launch_dense_bwd16(const GemmParams& px, const GemmParams& pw, int splits_w, hipStream_t s,
                              size_t smem_min = 0) {
    using CX = Cfg16<OP_KCONTIG, OP_KCONTIG, WGM, WGN, TM, TN, BK, 1>;
    using CW = Cfg16<OP_ICONTIG, OP_ICONTIG, WWGM, WWGN, WTM, WTN, BK, 1>;
    auto kern = dense_bwd16_kernel<WGM, WGN, TM, TN, BK, WWGM, WWGN, WTM, WTN, ACTX>;
    size_t smem = CX::smem_bytes(0) > CW::smem_bytes(0) ? CX::smem_bytes(0) : CW::smem_bytes(0);
    if (smem < smem_min) smem = smem_min;
    DCCN_TRY(set_smem_attr(kern, smem));
    const int nx = ceil_div(px.N, CX::BN) * ceil_div(px.M, CX::BM);
    const int tw = ceil_div(pw.N, CW::BN) * ceil_div(pw.M, CW::BM);
    DCCN_NO_CHAINS();
    hipLaunchKernelGGL(kern, dim3(nx + tw * splits_w), dim3(256), smem, s, px, pw, nx, tw);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

template <int WGM, int WGN, int TM, int TN, int BK>
static int launch_bwd_w16_finalize(const GemmParams& p, int splits, const TailFinalizeArgs& fin, hipStream_t s) {
    using CF = Cfg16<OP_ICONTIG, OP_ICONTIG, WGM, WGN, TM, TN, BK, 1>;
    auto kern = gemm16_bwd_w_finalize_kernel<WGM, WGN, TM, TN, BK>;
    const size_t smem = CF::smem_bytes(0);
    DCCN_TRY(set_smem_attr(kern, smem));
    const int tiles = ceil_div(p.N, CF::BN) * ceil_div(p.M, CF::BM), gemm_blocks = tiles * splits;
    DCCN_NO_CHAINS();
    hipLaunchKernelGGL(kern, dim3(gemm_blocks + tail_finalize_blocks(fin.P)), dim3(256), smem, s, p, tiles, gemm_blocks,
                       fin);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// split-K plan with an explicit split count (k ranges are multiples of BK; the last one may be ragged)
static inline SplitPlan plan_splitk_n(int K, int want, int BK) {
    if (want < 1) want = 1;
    int klen = (K + want - 1) / want;
    klen = (klen + BK - 1) / BK * BK;
    SplitPlan sp;
    sp.klen = klen;
    sp.splits = (K + klen - 1) / klen;
    return sp;
}

}  // namespace dccn
