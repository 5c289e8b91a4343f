// The equaliser's pilot bottleneck (dev/py/model.py:394-412): dense SK2 -> P (pilot extraction), dense P -> SK2, no
// activation in between, P = 2 * pilot_size = 32 at the reference's N = 64 frame (16 or 32 supported: NP below).  As
// separate GEMMs these are the worst-shaped launches of the step: the first has P output columns (25 tiles at 1170
// frames, 2 at 73, each with a 896-deep k-loop), the second a P-deep
// k-loop; their backward is two more.  Here each direction is ONE launch, a block per 16 frames:
//
//   forward   d1 = y.W1 + b1  (k = SK2 split over the four waves, fixed-order sum through LDS), then every wave takes
//             column tiles of  d2 = d1.W2 + b2  (P/4 MFMAs each, d1 from LDS)
//   backward  dd1 = dd2.W2^T the same way, then per column tile: the pilot branch's input gradient added to what is there
//             (dy_out = dy_in + dd1.W1^T), and this block's partial of dW2 = d1^T.dd2, db2, dW1 = y^T.dd1, db1 -- one
//             slab per block, summed in a fixed order by the optimizer launch (eq_opt.h EQJ_SUM).
//
// v_mfma_f32_16x16x4_f32: lane (c = l % 16, kq = l / 16) supplies A[i=c][k=kq] and B[k=kq][j=c] and receives
// D[i = 4 kq + r][j = c], r = 0..3.  A lane that loads a float4 along k at column 16 g + 4 kq feeds four consecutive
// MFMAs (step e uses the k set {16 g + 4 kq + e}); the other operand follows the same k sets.
#pragma once
#include "common.h"
#include "datagen.h"

namespace dccn {

typedef float bn_f32x4 __attribute__((ext_vector_type(4)));
// NP = P / 16 MFMA tiles across the bottleneck (P = 2 * pilot_size: 16 or 32; the reference's N = 64 frame has 16 pilots)

// acc[jt] (16 x 16 each, jt < NP) += A[rows m0.., k range of this wave] . B over k = SK2 / 4 per wave; A rows are frames
// (row-major, ld = SK2, rows past `rows` read as zero), B element (k, j) = BT ? Bm[j * SK2 + k] : Bm[k * P + j]
template <bool BT, int NP>
__device__ __forceinline__ void bn_long_k(const float* __restrict__ A, const float* __restrict__ Bm, const int m0,
                                          const int rows, const int SK2, const int wave, const int c, const int kq,
                                          bn_f32x4 (&acc)[NP]) {
    constexpr int P = 16 * NP;
#pragma unroll
    for (int jt = 0; jt < NP; ++jt) acc[jt] = bn_f32x4{0.f, 0.f, 0.f, 0.f};
    const int Kw = SK2 / 4, kbeg = wave * Kw;
    const int row = m0 + c;
    const bool rok = row < rows;
    const float* ar = A + (size_t)min(row, rows - 1) * SK2 + kbeg + 4 * kq;
    constexpr int G = 4;                                  // groups of 16 k in flight
    for (int g0 = 0; g0 < Kw / 16; g0 += G) {
        float4 a[G];
        float4 b[G][NP];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int g = min(g0 + u, Kw / 16 - 1);
            a[u] = *reinterpret_cast<const float4*>(ar + 16 * g);
            const int k = kbeg + 16 * g + 4 * kq;
#pragma unroll
            for (int jt = 0; jt < NP; ++jt) {
                const int j = 16 * jt + c;
                if constexpr (BT) {
                    b[u][jt] = *reinterpret_cast<const float4*>(Bm + (size_t)j * SK2 + k);
                } else {
                    b[u][jt] = make_float4(Bm[(size_t)(k + 0) * P + j], Bm[(size_t)(k + 1) * P + j],
                                           Bm[(size_t)(k + 2) * P + j], Bm[(size_t)(k + 3) * P + j]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if (g0 + u < Kw / 16) {
                const float4 av = rok ? a[u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int jt = 0; jt < NP; ++jt) {
                    acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b[u][jt].x, acc[jt], 0, 0, 0);
                    acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b[u][jt].y, acc[jt], 0, 0, 0);
                    acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b[u][jt].z, acc[jt], 0, 0, 0);
                    acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b[u][jt].w, acc[jt], 0, 0, 0);
                }
            }
        }
    }
}

// the four waves' partial 16 x P tiles -> one tile in LDS (fixed order 0+1+2+3), + bias (nullable)
template <int NP>
__device__ __forceinline__ void bn_join(const bn_f32x4 (&acc)[NP], float (*part)[16][16 * NP + 1], float (*tile)[16 * NP + 1],
                                        const float* __restrict__ bias, const int wave, const int c, const int kq) {
    constexpr int P = 16 * NP;
#pragma unroll
    for (int jt = 0; jt < NP; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][4 * kq + r][16 * jt + c] = acc[jt][r];
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * P; e += 256) {
        const int i = e / P, j = e % P;
        float v = ((part[0][i][j] + part[1][i][j]) + part[2][i][j]) + part[3][i][j];
        if (bias) v += bias[j];
        tile[i][j] = v;
    }
    __syncthreads();
}

// grid = (column chunks, ceil(B / 16)) blocks of 256 threads: a block owns `q` column tiles (16 columns each) of its 16
// frames and recomputes the frames' P bottleneck values for itself -- redundant long-k work (P columns only) that buys
// enough blocks to fill the chip at 73 frames (5 row tiles)
template <int NP>
__global__ __launch_bounds__(256) void eq_bottleneck_fwd_kernel(const float* y_, const float* W1_, const float* b1_, const float* W2_,
                                                                const float* b2_, float* d1_, float* d2_, const int B, const int SK2,
                                                                const int q, const ChainOffs co) {
    const long long coff = co.off[blockIdx.z];                     // chain groups (common.h)
    const float* __restrict__ y = chain_at(y_, coff);
    const float* __restrict__ W1 = chain_at(W1_, coff);
    const float* __restrict__ b1 = chain_at(b1_, coff);
    const float* __restrict__ W2 = chain_at(W2_, coff);
    const float* __restrict__ b2 = chain_at(b2_, coff);
    float* __restrict__ d1 = chain_at(d1_, coff);
    float* __restrict__ d2 = chain_at(d2_, coff);
    constexpr int P = 16 * NP;
    __shared__ float part[4][16][P + 1];
    __shared__ float t1[16][P + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, kq = lane >> 4;
    const int m0 = (int)blockIdx.y * 16;
    const int ct0 = (int)blockIdx.x * q, ct1 = min(ct0 + q, SK2 / 16);
    bn_f32x4 acc[NP];
    bn_long_k<false, NP>(y, W1, m0, B, SK2, wave, c, kq, acc);
    bn_join<NP>(acc, part, t1, b1, wave, c, kq);
    if (blockIdx.x == 0) {
        for (int e = threadIdx.x; e < 16 * P; e += 256) {
            const int i = e / P, j = e % P;
            if (m0 + i < B) d1[(size_t)(m0 + i) * P + j] = t1[i][j];
        }
    }
    float a[4 * NP];
#pragma unroll
    for (int s = 0; s < 4 * NP; ++s) a[s] = t1[c][4 * s + kq];
    for (int ct = ct0 + wave; ct < ct1; ct += 4) {
        const int n = ct * 16 + c;
        float bv[4 * NP];
#pragma unroll
        for (int s = 0; s < 4 * NP; ++s) bv[s] = W2[(size_t)(4 * s + kq) * SK2 + n];
        bn_f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4 * NP; ++s) o = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bv[s], o, 0, 0, 0);
        const float bj = b2 ? b2[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 4 * kq + r;
            if (row < B) d2[(size_t)row * SK2 + n] = o[r] + bj;
        }
    }
}

// slabs of block t: pW2 + t*P*SK2 ([P][SK2]), pb2 + t*SK2, pW1 + t*SK2*P ([SK2][P]), pb1 + t*P
// gen_rows > 0 (round 6): the FIRST grid rows are workgroups of the fused static-channel generator (datagen.h
// gen_static_frames_body) producing the NEXT batch of the training loop -- 15 us of integer / transcendental VALU work on 37
// workgroups that depends on nothing of this step and used to be a launch of its own in front of it; here it runs beside the
// bottleneck's MFMA tiles and the riders' HBM streams (dispatched first: it is the longest work item of the launch).
template <int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void eq_bottleneck_bwd_kernel(const float* dd2_, const float* d1_, const float* y_, const float* W1_, const float* W2_, const float* dy_in_,
                              float* dy_out_, float* pW2_, float* pb2_, float* pW1_, float* pb1_, const int B, const int SK2, const int q,
                              const int row_tiles, const EqRideArgs ride0, const dccn_adam_hparams hp, const int gen_rows,
                              const int gen_blocks, const GenStaticArgs gen_args, const GenChainScalars gen_sc, const ChainOffs co) {
    const long long coff = co.off[blockIdx.z];                     // chain groups (common.h)
    if ((int)blockIdx.y < gen_rows) {
        const int gb = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
        if (gb < gen_blocks) {
            DCCN_GEN_ARGS_OF_CHAIN(a, P0, P1, gen_args, gen_sc, coff, blockIdx.z, gb)
            gen_static_frames_body<7, 64, 16>(a, P0, P1, gb);
        }
        return;
    }
    const int by = (int)blockIdx.y - gen_rows;
    const float* __restrict__ dd2 = chain_at(dd2_, coff);
    const float* __restrict__ d1 = chain_at(d1_, coff);
    const float* __restrict__ y = chain_at(y_, coff);
    const float* __restrict__ W1 = chain_at(W1_, coff);
    const float* __restrict__ W2 = chain_at(W2_, coff);
    const float* __restrict__ dy_in = chain_at(dy_in_, coff);
    float* __restrict__ dy_out = chain_at(dy_out_, coff);
    float* __restrict__ pW2 = chain_at(pW2_, coff);
    float* __restrict__ pb2 = chain_at(pb2_, coff);
    float* __restrict__ pW1 = chain_at(pW1_, coff);
    float* __restrict__ pb1 = chain_at(pb1_, coff);
    constexpr int P = 16 * NP;
    // grid rows behind the batch's row tiles: optimizer riders (eq_opt.h EqRideArgs), dispatched after every block of the
    // launch's own work
    if (by >= row_tiles) {
        eq_ride_body(ride0, hp, (by - row_tiles) * (int)gridDim.x + (int)blockIdx.x, coff);
        return;
    }
    __shared__ float part[4][16][P + 1];
    __shared__ float g1[16][P + 1];                       // dd1 tile (rows past the batch are zero)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, kq = lane >> 4;
    const int m0 = by * 16, t = by;
    const int ct0 = (int)blockIdx.x * q, ct1 = min(ct0 + q, SK2 / 16);
    bn_f32x4 acc[NP];
    bn_long_k<true, NP>(dd2, W2, m0, B, SK2, wave, c, kq, acc);
    bn_join<NP>(acc, part, g1, nullptr, wave, c, kq);
    if (blockIdx.x == 0 && threadIdx.x < P) {             // db1 partial = column sums of the dd1 tile
        float sacc = 0.f;
        for (int i = 0; i < 16; ++i) sacc += g1[i][threadIdx.x];
        pb1[(size_t)t * P + threadIdx.x] = sacc;
    }
    float ga[4 * NP];                                     // A[i = frame c][k = p]            (dy branch, k over P)
    float gb[NP][4], da[NP][4];                           // B[k = frame][j = p] (dW1) / A[i = p][k = frame] (dW2), per p tile
    bool rok[4];
#pragma unroll
    for (int s = 0; s < 4 * NP; ++s) ga[s] = g1[c][4 * s + kq];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int row = m0 + 4 * s + kq;
        rok[s] = row < B;
#pragma unroll
        for (int jt = 0; jt < NP; ++jt) {
            gb[jt][s] = g1[4 * s + kq][16 * jt + c];
            da[jt][s] = rok[s] ? d1[(size_t)row * P + 16 * jt + c] : 0.f;
        }
    }
    float* W2p = pW2 + (size_t)t * P * SK2;
    float* W1p = pW1 + (size_t)t * SK2 * P;
    for (int ct = ct0 + wave; ct < ct1; ct += 4) {
        const int n = ct * 16 + c;
        float dv[4], yv[4], w1v[4 * NP];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const size_t ri = (size_t)min(m0 + 4 * s + kq, B - 1) * SK2 + n;
            dv[s] = rok[s] ? dd2[ri] : 0.f;                // B[k = frame][j = n]        (dW2, db2)
            yv[s] = rok[s] ? y[ri] : 0.f;                  // A[i = n][k = frame]        (dW1)
        }
#pragma unroll
        for (int s = 0; s < 4 * NP; ++s) w1v[s] = W1[(size_t)n * P + 4 * s + kq];      // B[k = p][j = n] = W1[n][p]
        bn_f32x4 oy = {0.f, 0.f, 0.f, 0.f}, ow2[NP], ow1[NP];
#pragma unroll
        for (int s = 0; s < 4 * NP; ++s) oy = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], w1v[s], oy, 0, 0, 0);
#pragma unroll
        for (int jt = 0; jt < NP; ++jt) {
            ow2[jt] = bn_f32x4{0.f, 0.f, 0.f, 0.f};
            ow1[jt] = bn_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                ow2[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(da[jt][s], dv[s], ow2[jt], 0, 0, 0);
                ow1[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[s], gb[jt][s], ow1[jt], 0, 0, 0);
            }
        }
        // bias gradient of dense_2: column sum of the dd2 tile -- the lane's four frames, then the four kq groups
        float bs = (dv[0] + dv[1]) + (dv[2] + dv[3]);
        bs += __shfl_xor(bs, 16, 64);
        bs += __shfl_xor(bs, 32, 64);
        if (kq == 0) pb2[(size_t)t * SK2 + n] = bs;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 4 * kq + r;               // oy: D[i = frame][j = n]
            if (row < B) {
                const size_t i = (size_t)row * SK2 + n;
                dy_out[i] = dy_in[i] + oy[r];
            }
#pragma unroll
            for (int jt = 0; jt < NP; ++jt) {
                W2p[(size_t)(16 * jt + 4 * kq + r) * SK2 + n] = ow2[jt][r];                 // D[i = p][j = n]
                W1p[(size_t)(ct * 16 + 4 * kq + r) * P + 16 * jt + c] = ow1[jt][r];         // D[i = n][j = p]
            }
        }
    }
}

static inline bool eq_bottleneck_ok(int B, int SK2, int P, const float* y, const float* W1, const float* W2) {
    return (P == 16 || P == 32) && B > 0 && SK2 >= 64 && (SK2 % 64) == 0 && aligned16(y) && aligned16(W1) && aligned16(W2);
}
// column tiles per block: about two blocks per CU in all, at least one tile per wave
static inline int eq_bottleneck_q(int B, int SK2) {
    const int tiles = SK2 / 16, rt = ceil_div(B, 16);
    int chunks = ceil_div(2 * kCUs, rt);
    if (chunks > ceil_div(tiles, 4)) chunks = ceil_div(tiles, 4);
    if (chunks < 1) chunks = 1;
    return ceil_div(ceil_div(tiles, chunks), 4) * 4;
}
static inline size_t eq_bottleneck_part_floats(int B, int SK2, int P) {
    return (size_t)ceil_div(B, 16) * ((size_t)2 * P * SK2 + SK2 + P);
}

}  // namespace dccn
