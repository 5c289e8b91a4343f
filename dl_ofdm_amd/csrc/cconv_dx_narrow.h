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
};

template <int NT16>
__global__ __launch_bounds__(256) void cconv_dx_narrow_kernel(const DxNarrowArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, kq = lane >> 4;
    int ph = 0;
    while (ph + 1 < a.nphase && (int)blockIdx.x >= a.tile0[ph + 1]) ++ph;
    const int tile = (int)blockIdx.x - a.tile0[ph];
    const int pl = ph / a.sW, pw = ph - pl * a.sW;
    const int Lp = pl < a.L ? (a.L - pl + a.sL - 1) / a.sL : 0, Wp = pw < a.Wd ? (a.Wd - pw + a.sW - 1) / a.sW : 0;
    const int per = Lp * Wp, Np = a.B * per;
    // live taps of this phase: ti' = ti0 + jl sL (fine position (pl + l0 + ti') a multiple of sL), likewise tj'
    const int ti0 = ((-(pl + a.l0)) % a.sL + a.sL) % a.sL, tj0 = ((-(pw + a.w0)) % a.sW + a.sW) % a.sW;
    const int ntl_p = ti0 < a.ntl ? (a.ntl - 1 - ti0) / a.sL + 1 : 0, ntw_p = tj0 < a.ntw ? (a.ntw - 1 - tj0) / a.sW + 1 : 0;
    const int Kp = ntl_p * ntw_p * a.F2;

    // this lane's A row: position r of the phase
    const int r = min(tile * 64 + 16 * wave + c, Np - 1);
    const int b = r / per, rem = r - b * per, lq = rem / Wp, wq = rem - lq * Wp;
    const int l = lq * a.sL + pl, w = wq * a.sW + pw;
    const int lo0 = (l + a.l0 + ti0) / a.sL, wo0 = (w + a.w0 + tj0) / a.sW;      // exact divisions (may be negative)
    const int pos = (b * a.L + l) * a.Wd + w;                                     // linear position of the row (for the stores)
    const float* drow = a.dout + (size_t)b * a.Lo * a.Wo * a.F2;
    const int Kfull = a.ntl * a.ntw * a.F2;
    const float* brow[NT16];
    bool bok[NT16];
#pragma unroll
    for (int t = 0; t < NT16; ++t) {
        bok[t] = 16 * t + c < a.C2;
        brow[t] = a.bt + (size_t)min(16 * t + c, a.C2 - 1) * Kfull;
    }

    kf32x4 acc[NT16];
#pragma unroll
    for (int t = 0; t < NT16; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = 0.f;

    // tap walk of this lane: k = 16 g + 4 kq inside the phase's k range = (jl, jw, n)
    int n = 4 * kq, jl = 0, jw = 0;
    auto norm = [&]() {
        while (n >= a.F2) {
            n -= a.F2;
            if (++jw == ntw_p) { jw = 0; ++jl; }
        }
    };
    norm();
    constexpr int U = 4;                                   // groups in flight
    const int groups = (Kp + 15) / 16;
    for (int g0 = 0; g0 < groups; g0 += U) {
        kf32x4 av[U], bv[U][NT16];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = jl < ntl_p;                  // (past the k range: zeros)
            const int lo = lo0 + jl, wo = wo0 + jw;
            const bool ok = live && (unsigned)lo < (unsigned)a.Lo && (unsigned)wo < (unsigned)a.Wo;
            const size_t ao = ok ? ((size_t)lo * a.Wo + wo) * a.F2 + n : 0;
            av[u] = *reinterpret_cast<const kf32x4*>(drow + ao);
            if (!ok) av[u] = kf32x4{0.f, 0.f, 0.f, 0.f};
            const int kb = live ? ((ti0 + jl * a.sL) * a.ntw + tj0 + jw * a.sW) * a.F2 + n : 0;
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
                bv[u][t] = *reinterpret_cast<const kf32x4*>(brow[t] + kb);
                if (!(live && bok[t])) bv[u][t] = kf32x4{0.f, 0.f, 0.f, 0.f};
            }
            n += 16;
            norm();
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NT16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][j], bv[u][t][j], acc[t], 0, 0, 0);
    }
    // C/D layout: col = lane & 15, row = 4 kq + q: the linear position of that row sits in lane (4 kq + q) of this wave
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 4 * kq + q;
        const int prow = __shfl(pos, row, 64);
        const bool rok = tile * 64 + 16 * wave + row < Np;
#pragma unroll
        for (int t = 0; t < NT16; ++t)
            if (rok && 16 * t + c < a.C2) a.dx[(size_t)prow * a.C2 + 16 * t + c] = acc[t][q];
    }
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
    if (a.C2 <= 16) hipLaunchKernelGGL(cconv_dx_narrow_kernel<1>, dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(cconv_dx_narrow_kernel<2>, dim3(blocks), dim3(256), 0, s, a);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

}  // namespace dccn
