// Backward of a few-channel 1-D complex convolution (dev/py/complex.py:51-92 layers_conv1d_complex: x [B, L, C, 2], k taps,
// stride s) in ONE pass over dout: input gradient, weight gradient and bias gradient together.
//
// Why: at C = 2, k = 5, 2F = 128 (one 560-sample frame per row, tools/convbench.py) dout is 335 MB and x 10 MB; the separate
// launches -- cconv_dx_narrow_kernel 0.193 ms, the k-major weight gradient 0.150 ms -- each stream dout at 1.8-2.3 TB/s and
// spend most of their MFMA time on padding (16- and 64-row tiles for 4 columns / 20 rows of payload).  Here a block walks
// chunks of <= 64 output rows q of one batch item; per chunk
//   D  [64][2F]   the dout rows, staged ONCE in LDS -- it is the k-contiguous A operand of (1) and the k-major B operand of (2);
//   (1) Y[q][(t,c,iq)] = D[q][:] . Weff[(t,c,iq)][:]          (rows x 32 columns: 20 used; Weff^T sits in LDS for the whole launch)
//       dx[p] = sum over the taps t with p = q s + o + t of Y[q][(t, c, iq)]     -- gathered out of LDS in tap order: the
//       col2im of rounds 3-5 without the [rows, kin, 2] tensor; a chunk owns the positions whose rows it holds completely
//       (59 of 64 rows' worth at k = 5, s = 1: the halo rows are read twice, 8 % of dout);
//   (2) dWeff[(t,c,iq)][n] += Xp[q][(t,c,iq)] . D[q][n]       (32 rows: 20 used + one row of ones = the bias gradient's column
//       sums; Xp = the patch rows gathered from x, zero for the halo rows a neighbouring chunk owns), accumulated in registers
//       over all chunks of the block, one [32][2F] slab per block at the end, folded by cconv_fold_kernel like every C-Conv's.
// 128 v_mfma_f32_16x16x4_f32 per wave and chunk; the next chunk's dout rows are in flight (registers) while a chunk computes.
// Summation orders: taps ascending for dx; rows ascending within a block, blocks in fold order for dW -- deterministic.
#pragma once
#include "gemm_kmajor.h"

namespace dccn {

struct Conv1dBwdArgs {
    const float* x;         // [B, L, C2]
    const float* dout;      // [B, Lo, F2]
    const float* w;         // [kin][2F] = [Wa|Wb], kin = nt * C
    float* dx;              // [B, L, C2]
    float* slabs;           // [blocks][32][F2]   dWeff partials (rows >= NC unused)
    float* colsum;          // [blocks][F2]       bias-gradient partials
    int B, L, C2, Lo, nt, F2, NC;      // NC = nt * C2 <= 30
    int o, s;               // x position of output row q at tap t: q s + o + t   (o = first live tap - padding before)
    int PL, nch;            // positions a chunk owns, chunks per batch item
};

__device__ __forceinline__ int floordiv(const int a, const int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
__device__ __forceinline__ int ceildiv_s(const int a, const int b) { return -floordiv(-a, b); }

constexpr int kC1dLdD(int F2) { return F2 + 4; }
constexpr int kC1dLdX = 36, kC1dLdY = 33;
template <int F2>
constexpr size_t conv1d_bwd_smem_bytes() {
    return (size_t)(64 * kC1dLdD(F2) + 32 * kC1dLdD(F2) + 64 * kC1dLdX + 64 * kC1dLdY) * sizeof(float);
}

template <int F2>
__global__ __launch_bounds__(256) void cconv1d_bwd_fused_kernel(const Conv1dBwdArgs a) {
    constexpr int LDD = kC1dLdD(F2), G = F2 / 16, NTW = F2 / 64;      // NTW: 16-column tiles of dW per wave = (F2 / 16) / 4
    static_assert(F2 % 64 == 0 && F2 <= 128, "2F: 64 or 128");
    constexpr int NLD = 64 * (F2 / 4) / 256;                           // float4 pieces of a dout tile per thread
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sD = sm;                          // [64][LDD]  dout rows of the chunk
    float* sW = sD + 64 * LDD;               // [32][LDD]  Weff rows (t, c, iq), k = n contiguous
    float* sX = sW + 32 * LDD;               // [64][36]   patch rows of x (+ the ones column NC)
    float* sY = sX + 64 * kC1dLdX;           // [64][33]   Y tile
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, kq = lane >> 4;
    const int F = F2 / 2;

    // Weff[(t,c,iq)][2f + e] from [Wa|Wb] (gemm_f32_mfma.h cconv_weff): once per block
    for (int i = tid; i < 32 * F2; i += 256) {
        const int row = i / F2, col = i - row * F2;
        sW[row * LDD + col] = row < a.NC ? cconv_weff(a.w, F, row, col) : 0.f;
    }

    kf32x4 accw[2][NTW];                     // dWeff tiles: rows 16 it + ..., columns 16 (NTW wv + nt) + ...
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) accw[it][nt][r] = 0.f;

    const int total = a.B * a.nch;
    // chunk geometry (uniform)
    auto geom = [&](const int ch, int& b, int& p0, int& p1, int& qs, int& qe, int& qown) {
        b = ch / a.nch;
        const int j = ch - b * a.nch;
        p0 = j * a.PL;
        p1 = min(p0 + a.PL, a.L);
        qs = max(0, ceildiv_s(p0 - a.o - (a.nt - 1), a.s));
        const int qn = (j + 1 < a.nch) ? min(a.Lo, max(0, ceildiv_s(p1 - a.o - (a.nt - 1), a.s))) : a.Lo;      // next chunk's first row
        qown = qn;                                                                                              // this chunk owns [qs, qn)
        qe = min(a.Lo - 1, max(floordiv(p1 - 1 - a.o, a.s), qn - 1));
        qe = min(qe, qs + 63);
    };
    kf32x4 rd[NLD];
    float xr[8];                             // this thread's 8 entries of the chunk's patch tile: column tid % 32, rows tid / 32 + 8 v
    const int xcol = tid & 31, xt = xcol / a.C2, xciq = xcol - xt * a.C2;
    auto request = [&](const int ch) {       // the chunk's dout rows and patch entries -> registers (zeros behind the last row)
        int b, p0, p1, qs, qe, qown;
        geom(ch, b, p0, p1, qs, qe, qown);
#pragma unroll
        for (int v = 0; v < 8; ++v) {        // patch rows of x for the OWNED rows (others zero), column NC = 1 for owned rows (bias gradient)
            const int q = qs + (tid >> 5) + 8 * v;
            const int l = q * a.s + a.o + xt;
            const bool own = q < qown && q <= qe;
            const bool in = own && xcol < a.NC && l >= 0 && l < a.L;
            const float xv = a.x[((size_t)b * a.L + (in ? l : 0)) * a.C2 + (in ? xciq : 0)];
            xr[v] = in ? xv : ((own && xcol == a.NC) ? 1.f : 0.f);
        }
#pragma unroll
        for (int v = 0; v < NLD; ++v) {
            const int idx = tid + 256 * v, row = idx / (F2 / 4), c4 = idx - row * (F2 / 4);
            const int q = min(qs + row, a.Lo - 1);
            rd[v] = *reinterpret_cast<const kf32x4*>(a.dout + ((size_t)b * a.Lo + q) * F2 + 4 * c4);
            if (qs + row > qe) rd[v] = kf32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    int ch = (int)blockIdx.x;
    if (ch < total) request(ch);
    for (; ch < total; ch += (int)gridDim.x) {
        int b, p0, p1, qs, qe, qown;
        geom(ch, b, p0, p1, qs, qe, qown);
        __syncthreads();                     // the previous chunk's gather has finished with sY / sD / sX
#pragma unroll
        for (int v = 0; v < NLD; ++v) {
            const int idx = tid + 256 * v, row = idx / (F2 / 4), c4 = idx - row * (F2 / 4);
            *reinterpret_cast<kf32x4*>(sD + row * LDD + 4 * c4) = rd[v];
        }
#pragma unroll
        for (int v = 0; v < 8; ++v) sX[((tid >> 5) + 8 * v) * kC1dLdX + xcol] = xr[v];
        if (ch + (int)gridDim.x < total) request(ch + (int)gridDim.x);          // next chunk's rows: in flight during the MFMAs
        __syncthreads();

        // (1) Y = D . Weff^T : wave wv rows 16 wv .., 2 column tiles
        kf32x4 accy[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) { accy[0][r] = 0.f; accy[1][r] = 0.f; }
        const float* Dr = sD + (16 * wv + c) * LDD + 4 * kq;
        const float* W0 = sW + c * LDD + 4 * kq;
        const float* W1 = sW + (16 + c) * LDD + 4 * kq;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const kf32x4 av = *reinterpret_cast<const kf32x4*>(Dr + 16 * g);
            const kf32x4 b0 = *reinterpret_cast<const kf32x4*>(W0 + 16 * g);
            const kf32x4 b1 = *reinterpret_cast<const kf32x4*>(W1 + 16 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                accy[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], b0[j], accy[0], 0, 0, 0);
                accy[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], b1[j], accy[1], 0, 0, 0);
            }
        }
        // (2) dWeff += Xp^T . D : k = the chunk's 64 rows; lane (c, kq) supplies A[i = 16 it + c][k = 4 ks + kq], B[k][n = 16 nt + c]
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int k = 4 * ks + kq;
            const float a0 = sX[k * kC1dLdX + c], a1 = sX[k * kC1dLdX + 16 + c];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const float bv = sD[k * LDD + 16 * (NTW * wv + nt) + c];
                accw[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, accw[0][nt], 0, 0, 0);
                accw[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, accw[1][nt], 0, 0, 0);
            }
        }
        // Y tile -> LDS (C/D layout: row 4 kq + r, column c)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) sY[(16 * wv + 4 * kq + r) * kC1dLdY + 16 * ct + c] = accy[ct][r];
        __syncthreads();
        // dx of the owned positions: taps in ascending order
        const int npos = p1 - p0;
        for (int e = tid; e < npos * a.C2; e += 256) {
            const int pl = e / a.C2, ciq = e - pl * a.C2, p = p0 + pl;
            float acc = 0.f;
            for (int t = 0; t < a.nt; ++t) {
                const int num = p - a.o - t;
                if (num < 0) continue;
                const int q = num / a.s;
                if (q * a.s != num || q < qs || q > qe) continue;
                acc += sY[(q - qs) * kC1dLdY + t * a.C2 + ciq];
            }
            a.dx[((size_t)b * a.L + p) * a.C2 + ciq] = acc;
        }
    }
    // the block's dWeff partial (+ the bias gradient's column sums from row NC)
    float* slab = a.slabs + (size_t)blockIdx.x * 32 * F2;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * it + 4 * kq + r, n = 16 * (NTW * wv + nt) + c;
                if (i < a.NC) slab[(size_t)i * F2 + n] = accw[it][nt][r];
                else if (i == a.NC) a.colsum[(size_t)blockIdx.x * F2 + n] = accw[it][nt][r];
            }
}

}  // namespace dccn
