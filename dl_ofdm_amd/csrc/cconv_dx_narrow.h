// Input gradient of a general-k complex convolution with FEW channels (2C <= 32 output columns), any stride, one or two tap axes
// (dev/py/complex.py:51-92 layers_conv1d_complex, :140-196 layers_conv2d_complex with k > 1 / strides > 1; backward).
//
//   dx[b, l, w, (c,iq)] = sum over taps (ti', tj') and n of dout[b, (l + l0 + ti')/sL, (w + w0 + tj')/sW, n] . Bt[(c,iq)][(ti', tj', n)]
//
// (flipped taps t' = nt-1-t, origin l0 = -(tl0 - pl0) - (ntl - 1); Bt = the transposed, tap-flipped Weff built by
// cconv_flip_wt_kernel).  As a GEMM this is [positions] x [2C] over k = taps x 2F.  The 64x64 tiles of the implicit GEMM
// (gemm_f32_mfma.h OP_KPATCH) leave 60 of 64 columns empty at C = 2 -- which is why rounds 3-5 sent these shapes through a
// [rows, kin, 2] tensor and col2im.  Here:
//   * a wave owns 16 positions x 16 (or 32) columns on v_mfma_f32_16x16x4_f32; lane (c = l % 16, kq = l / 16) supplies
//     A[row c][k] and B[k][col c] for four consecutive k of its quarter of a 16-deep group straight from global memory (one float4
//     each: the operands of four consecutive MFMAs; the k permutation inside a group is the same on both sides) -- no LDS, no
//     barrier, 8 waves per SIMD hide the latency; dout rows are re-read per tap out of L1 / L2 (neighbouring positions share them);
//   * STRIDES by phase decomposition: positions with the same (l mod sL, w mod sW) meet data at the same taps (every sL-th /
//     sW-th one), so a wave's 16 rows come from one phase and its k range holds only that phase's live taps -- no multiplications
//     by the zeros a strided transposed convolution is made of, no col2im scatter;
//   * rows of a tile are decomposed once per lane; the tap walk per k group is incremental (no divisions in the loop).
// Accumulation order: taps in flipped order, n ascending, fp32 MFMA -- its own (deterministic) order; parity 1e-5.
#pragma once
#include "gemm_kmajor.h"

namespace dccn {

constexpr int kDxMaxPhases = 16;

struct DxNarrowArgs {
    const float* dout;      // [B, Lo, Wo, F2]
    const float* bt;        // [C2][ntl * ntw * F2]  (cconv_flip_wt_kernel)
    float* dx;              // [B, L, Wd, C2]
    int B, L, Wd, C2, Lo, Wo, ntl, ntw, F2;
    int l0, w0;             // fine origin of flipped tap 0 (see above)
    int sL, sW;             // forward strides
    int nphase;             // sL * sW
    int tile0[kDxMaxPhases + 1];   // first block of every phase
    int bstride;            // BLDS: row stride of the LDS copy of Bt (floats)
};

// NT16: column tiles of 16; DEEP: float4s per lane and k group (2: groups of 32 k, a lane's two float4s are 32 contiguous bytes
// and four lanes cover one 128-byte line of a dout row -- needs 2F % 8 == 0; 1: groups of 16 k, any even F); BLDS: Bt staged in
// LDS once per block (row stride = 16 mod 32 floats: conflict-free ds_read_b128) instead of read through the L1 per group.
// Why both matter: a wave's load instruction touches 16 different lines (one per row); the CU's L1 looks up about one line per
// clock and serves four SIMDs, so with 16-deep groups and both operands through the L1 (32 line lookups per 128 MFMA cycles
// and SIMD) the L1, not the matrix pipe, set the pace (first version: 0.267 ms at the C = 2 bench shape).
// NMAJ (2F a multiple of the group depth): the k walk runs over the TAPS fastest and the 2F columns slowest -- B sits in LDS and
// is indifferent to the order, and a 128-byte line of a dout row is then used by all its taps (positions q .. q + ntaps - 1)
// within ntaps consecutive groups, out of the L1 / L2, instead of once per pass over the whole row: tap-major, 768 waves per XCD
// x 20 live rows x 512 B = 7.7 MB of live lines against a 4-MB L2 -- measured 32 % L2 misses, 540 MB fetched for a 335-MB
// tensor, waves waiting for memory 62 % of their cycles (profiles/r06_conv_dx_narrow_pmc.txt).
// NMAJ also loads dout COALESCED: the MFMA wants lane (c, kq) to hold row c, but 16 consecutive lanes = 16 different rows makes
// every quarter-wave of a dwordx4 load 16 separate line accesses in the L1 (64 per instruction: 105 M per launch at the bench
// shape = 172 us of one-access-per-clock L1 time per CU -- the actual bound of every earlier version, whatever else changed).
// Here lane l loads row l / 4, piece l % 4 (a quarter-wave = 4 rows x one line), and the MFMA layout is restored by four
// ds_bpermute_b32 per float4 (lane (c, kq) takes what lane 4 c + kq loaded): a quarter of the L1 accesses for 1 KB of LDS
// crossbar traffic per 4 MFMAs.
// Where it stands (C = 2, k = 5, 2F = 128, 655 200 positions): 0.267 -> 0.193 ms; 356 MB fetched (335 algorithmic), L1 accesses
// 105 M -> 27 M, matrix pipe busy 44 % on tiles that are 3/4 padding.  The 335-MB tensor is read at 1.85 TB/s where this
// box's read-mostly launches top out near 2.3-2.7 TB/s (the k-major weight gradient and the forward of the same layer): the
// remaining step is ONE pass over dout for both gradients, not a faster dX.
template <int NT16, int DEEP, bool BLDS, bool NMAJ>
__global__ __launch_bounds__(256) void cconv_dx_narrow_kernel(const DxNarrowArgs a) {
    extern __shared__ __attribute__((aligned(16))) float dxn_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, kq = lane >> 4;
    constexpr int GK = 16 * DEEP;                          // k per group
    const int Kfull = a.ntl * a.ntw * a.F2;
    if constexpr (BLDS) {                                  // Bt -> LDS, once per (persistent) block: the phases index it absolutely
        const int k4 = Kfull >> 2;
        for (int i = threadIdx.x; i < a.C2 * k4; i += 256) {
            const int row = i / k4, col = i - row * k4;
            *reinterpret_cast<kf32x4*>(dxn_smem + (size_t)row * a.bstride + 4 * col) =
                *reinterpret_cast<const kf32x4*>(a.bt + (size_t)row * Kfull + 4 * col);
        }
        __syncthreads();
    }
    // persistent blocks: a tile of 64 positions lives ~2 us, the staging of Bt and the launch ramp are paid once per block
    for (int gt = (int)blockIdx.x; gt < a.tile0[kDxMaxPhases]; gt += (int)gridDim.x) {
    int ph = 0;
    while (ph + 1 < a.nphase && gt >= a.tile0[ph + 1]) ++ph;
    const int tile = gt - a.tile0[ph];
    const int pl = ph / a.sW, pw = ph - pl * a.sW;
    const int Lp = pl < a.L ? (a.L - pl + a.sL - 1) / a.sL : 0, Wp = pw < a.Wd ? (a.Wd - pw + a.sW - 1) / a.sW : 0;
    const int per = Lp * Wp, Np = a.B * per;
    // live taps of this phase: ti' = ti0 + jl sL (fine position (pl + l0 + ti') a multiple of sL), likewise tj'
    const int ti0 = ((-(pl + a.l0)) % a.sL + a.sL) % a.sL, tj0 = ((-(pw + a.w0)) % a.sW + a.sW) % a.sW;
    const int ntl_p = ti0 < a.ntl ? (a.ntl - 1 - ti0) / a.sL + 1 : 0, ntw_p = tj0 < a.ntw ? (a.ntw - 1 - tj0) / a.sW + 1 : 0;
    const int Kp = ntl_p * ntw_p * a.F2;

    // this lane's A row: position r of the phase (NMAJ: the row it LOADS, l / 4, and the piece l % 4 of the group it loads)
    const int lr = NMAJ ? (lane >> 2) : c, lk = NMAJ ? (lane & 3) : kq;
    const int r = min(tile * 64 + 16 * wave + lr, Np - 1);
    const int b = r / per, rem = r - b * per, lq = rem / Wp, wq = rem - lq * Wp;
    const int l = lq * a.sL + pl, w = wq * a.sW + pw;
    const int lo0 = (l + a.l0 + ti0) / a.sL, wo0 = (w + a.w0 + tj0) / a.sW;      // exact divisions (may be negative)
    const int pos = (b * a.L + l) * a.Wd + w;                                     // linear position of the row (for the stores)
    const float* drow = a.dout + (size_t)b * a.Lo * a.Wo * a.F2;
    const float* brow[NT16];
    bool bok[NT16];
#pragma unroll
    for (int t = 0; t < NT16; ++t) {
        bok[t] = 16 * t + c < a.C2;
        const int rowb = min(16 * t + c, a.C2 - 1);
        brow[t] = BLDS ? dxn_smem + (size_t)rowb * a.bstride : a.bt + (size_t)rowb * Kfull;
    }

    kf32x4 acc[NT16];
#pragma unroll
    for (int t = 0; t < NT16; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = 0.f;

    // tap walk of this lane: group g covers k = (jl, jw, n .. n + 4 DEEP) of the phase's k range
    // (NMAJ: n is the group's base, the same in every lane; the lane's own offset is 4 DEEP lk for the load, 4 DEEP kq for B)
    int n = NMAJ ? 0 : 4 * DEEP * kq, jl = 0, jw = 0;
    auto advance = [&]() {
        if constexpr (NMAJ) {                              // taps fastest
            if (++jw == ntw_p) {
                jw = 0;
                if (++jl == ntl_p) { jl = 0; n += GK; }
            }
        } else {                                           // 2F columns fastest (any even F: a group may span taps)
            n += GK;
            while (n >= a.F2) {
                n -= a.F2;
                if (++jw == ntw_p) { jw = 0; ++jl; }
            }
        }
    };
    if constexpr (!NMAJ) { n -= GK; advance(); }          // (normalises the start when 2F < the lane's first offset)
    // U groups per chunk (64 k), two register sets: the global loads of chunk i+1 are in flight while the 16 MFMAs of chunk i
    // run; the other waves of the SIMD cover the rest of the latency.  Masks ride along as bit sets and are applied when the
    // chunk is consumed, so that nothing between a load and its use waits for it.
    constexpr int U = 4 / DEEP;
    const int groups = (Kp + GK - 1) / GK;
    constexpr int NS = 2;                                  // register sets (three -- loads two chunks ahead, 100 VGPRs -- measured the same)
    kf32x4 av[NS][U][DEEP];
    int kbv[NS][U];
    unsigned aok[NS], blive[NS];
    auto load_chunk = [&](const int s) {
        aok[s] = 0u; blive[s] = 0u;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = NMAJ ? n < a.F2 : jl < ntl_p;     // (past the k range: zeros)
            const int lo = lo0 + jl, wo = wo0 + jw;
            const bool ok = live && (unsigned)lo < (unsigned)a.Lo && (unsigned)wo < (unsigned)a.Wo;
            const size_t ao = ok ? ((size_t)lo * a.Wo + wo) * a.F2 + n + (NMAJ ? 4 * DEEP * lk : 0) : 0;
#pragma unroll
            for (int d = 0; d < DEEP; ++d) av[s][u][d] = *reinterpret_cast<const kf32x4*>(drow + ao + 4 * d);
            kbv[s][u] = live ? ((ti0 + jl * a.sL) * a.ntw + tj0 + jw * a.sW) * a.F2 + n + (NMAJ ? 4 * DEEP * kq : 0) : 0;
            aok[s] |= (ok ? 1u : 0u) << u;
            blive[s] |= (live ? 1u : 0u) << u;
            advance();
        }
    };
    auto mfma_chunk = [&](const int s) {
        kf32x4 bv[U][DEEP][NT16];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int d = 0; d < DEEP; ++d)
#pragma unroll
                for (int t = 0; t < NT16; ++t) bv[u][d][t] = *reinterpret_cast<const kf32x4*>(brow[t] + kbv[s][u] + 4 * d);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = (aok[s] >> u) & 1u, live = (blive[s] >> u) & 1u;
#pragma unroll
            for (int d = 0; d < DEEP; ++d)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = ok ? av[s][u][d][j] : 0.f;
                    if constexpr (NMAJ)       // row c, quarter kq of the group was loaded by lane 4 c + kq
                        x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (4 * c + kq), __builtin_bit_cast(int, x)));
#pragma unroll
                    for (int t = 0; t < NT16; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, (live && bok[t]) ? bv[u][d][t][j] : 0.f, acc[t], 0, 0, 0);
                }
        }
    };
    if (groups > 0) load_chunk(0);
    for (int g0 = 0; g0 < groups; g0 += 2 * U) {
        if (g0 + U < groups) load_chunk(1);
        mfma_chunk(0);
        if (g0 + U < groups) {
            if (g0 + 2 * U < groups) load_chunk(0);
            mfma_chunk(1);
        }
    }
    // C/D layout: col = lane & 15, row = 4 kq + q: the linear position of that row sits in lane (4 kq + q) of this wave
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 4 * kq + q;
        const int prow = __shfl(pos, NMAJ ? 4 * row : row, 64);
        const bool rok = tile * 64 + 16 * wave + row < Np;
#pragma unroll
        for (int t = 0; t < NT16; ++t)
            if (rok && 16 * t + c < a.C2) a.dx[(size_t)prow * a.C2 + 16 * t + c] = acc[t][q];
    }
    }   // tiles of this block
}

static inline bool cconv_dx_narrow_ok(int C, int F, int sL, int sW) {
    return 2 * C <= 32 && (F % 2) == 0 && sL >= 1 && sW >= 1 && sL * sW <= kDxMaxPhases;
}

static int launch_cconv_dx_narrow(DxNarrowArgs a, hipStream_t s) {
    a.nphase = a.sL * a.sW;
    int blocks = 0;
    for (int ph = 0; ph < a.nphase; ++ph) {
        const int pl = ph / a.sW, pw = ph % a.sW;
        const int Lp = pl < a.L ? (a.L - pl + a.sL - 1) / a.sL : 0, Wp = pw < a.Wd ? (a.Wd - pw + a.sW - 1) / a.sW : 0;
        a.tile0[ph] = blocks;
        blocks += ceil_div(a.B * Lp * Wp, 64);
    }
    for (int ph = a.nphase; ph <= kDxMaxPhases; ++ph) a.tile0[ph] = blocks;
    if (blocks <= 0) return DCCN_ERR_INVALID_ARG;
    DCCN_NO_CHAINS();
    // Bt in LDS when it fits beside nothing else (<= 64 KB: two blocks per CU); rows 16 mod 32 floats apart
    const int Kfull = a.ntl * a.ntw * a.F2;
    a.bstride = (Kfull + 31) / 32 * 32 + 16;
    const size_t smem = (size_t)a.C2 * a.bstride * sizeof(float);
    const bool blds = smem <= 64 * 1024, deep = (a.F2 % 8) == 0, two = a.C2 > 16;
    const bool nmaj = blds && (a.F2 % (deep ? 32 : 16)) == 0;
    const int grid = blocks < 6 * kCUs ? blocks : 6 * kCUs;           // (6 blocks = 24 waves per CU: the occupancy the registers allow)
#define DCCN_DXN_LAUNCH(NT, DP, BL)                                                                              \
    do {                                                                                                         \
        auto kern = (BL && nmaj) ? cconv_dx_narrow_kernel<NT, DP, BL, BL> : cconv_dx_narrow_kernel<NT, DP, BL, false>; \
        if (BL) DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), smem));                       \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), BL ? smem : 0, s, a);                                    \
    } while (0)
    if (two) {
        if (deep) { if (blds) DCCN_DXN_LAUNCH(2, 2, true); else DCCN_DXN_LAUNCH(2, 2, false); }
        else { if (blds) DCCN_DXN_LAUNCH(2, 1, true); else DCCN_DXN_LAUNCH(2, 1, false); }
    } else {
        if (deep) { if (blds) DCCN_DXN_LAUNCH(1, 2, true); else DCCN_DXN_LAUNCH(1, 2, false); }
        else { if (blds) DCCN_DXN_LAUNCH(1, 1, true); else DCCN_DXN_LAUNCH(1, 1, false); }
    }
#undef DCCN_DXN_LAUNCH
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

}  // namespace dccn
