// The backward half of the basic receiver's training step as ONE launch (small layers: N = 64):
//   * dense dX tiles (gemm_f32_mfma.h, 64x64x64) whose epilogue contracts the finished dfft tile with the matching
//     rows of the normalised input -- the C-Conv weight gradient dWeff = x^T . dfft (dev/py/complex.py:183-192 backward;
//     the C-Conv input is data, so dfft has no other consumer) -- and leaves one [2kin, 64] partial per tile;
//   * the dense dW (tile, k range) items in the k-major form (gemm_kmajor.h);
//   * the tail's slab reduction + optimizer bookkeeping (a few leading blocks);
//   * optionally R0 of the NEXT batch into the other x_norm buffer (leading blocks).
// Round 2 ran the C-Conv weight gradient as a launch of its own (split-K 64 over 8190 rows, 8.8 us at 24 % MFMA, most
// of it launch ramp + drain) behind a 4.2 MB dfft round trip through HBM.  Here the same 0.335 GFLOP ride on tiles that
// already hold dfft in registers: one launch, one boundary and the dfft write + read less per step.
//
// Epilogue of a dX tile (64 frames x 64 columns of dfft, columns = one half of symbol s):
//   dfft tile (accumulators)      -> LDS D[k = frame][n]           (rows past the batch zeroed)
//   x_norm[frame, s, 0..2kin)     -> LDS X[k = frame][m]           (float4 copies)
//   partial[m][n] = sum_k X[k][m] * D[k][n]   on v_mfma_f32_16x16x4_f32 in the k-major form of gemm_kmajor.h:
//   odd rows stored with their 32-column halves swapped, lane (c, kq) reads two adjacent columns of row 4ks + kq with
//   one ds_read_b64 (conflict free), wave (hA, hB) owns rows 64ch + 32hA + 2c + s and columns 32hB + 2c + t.
//   2kin = 32 NU rows: NU/2 chunks of 64 rows (4 MFMAs per wave and k-step each) and, for odd NU, one unit of 32 rows of
//   which a wave takes the row class s = hA (2 MFMAs): 10 MFMAs per wave and k-step at 2kin = 160, nothing idle.
//   Column sums of the tile (C-Conv bias gradient) come from the accumulators before they are written.
// The partials are summed (fixed order => deterministic) onto [Wa|Wb] by the optimizer launch (norm_adam.h).
//   Round 5: a tile holds all four dWeff entries of each (n, f) pair it touches -- rows 2n, 2n+1 x columns 2f, 2f+1 sit in
//   ONE lane's accumulators -- so the fold of Appendix A.2, dWa = dWeff[2n,2f] - dWeff[2n+1,2f+1], dWb = dWeff[2n,2f+1] -
//   dWeff[2n+1,2f], is taken per tile BEFORE the store: the partial is [kin][32][{a, b}] = half the bytes (266 x 20 KB
//   instead of 266 x 40 KB written here and read back by the optimizer launch).  The optimizer summed fl(top - bot) per
//   term before; it now sums the stored fl(top - bot): the same additions of the same values => bit-identical gradients.
#pragma once
#include "gemm_kmajor.h"
#include "norm_adam.h"
#include "tail.h"

namespace dccn {

struct DweffArgs {
    const float* xn;      // x_norm [batch][S][2kin]
    float* partial;       // [dX tiles][kin][32][2]: folded per tile (a = dWa term, b = dWb term)
    float* colsum;        // [dX tiles][64]
    int batch, ldx;       // ldx = S * 2kin
    int two_kin, two_F;
    int prio;             // s_setprio level of the dX blocks (they are the long items of the grid)
};

template <int NU>
constexpr size_t dweff_smem_bytes() { return (size_t)(64 * 64 * ((NU + 1) / 2) + 64 * 64 + 128) * sizeof(float); }

// the x rows of a tile's frames, requested BEFORE the tile's k-loop (they depend on nothing the loop computes): 2 NU
// float4 per thread wait in registers, their latency hidden behind the ten k-tiles of the dX contraction
template <int NU>
__device__ __forceinline__ void dweff_piece(const int tid, const int v, int& row, int& c4) {
    // piece v of a thread: 8 consecutive lanes = 128 contiguous bytes of one row
    const int idx = tid + 256 * v, c8 = idx & 7, g = idx >> 3;
    row = g / NU;
    c4 = 8 * (g - row * NU) + c8;
}
template <int NU>
__device__ __forceinline__ void dweff_request(const DweffArgs& d, const int m0, const int n0, kf32x4 (&rx)[2 * NU]) {
    const int tid = threadIdx.x;
    const int sym = n0 / d.two_F;
#pragma unroll
    for (int v = 0; v < 2 * NU; ++v) {
        int row, c4;
        dweff_piece<NU>(tid, v, row, c4);
        const int f = min(m0 + row, d.batch - 1);
        rx[v] = *reinterpret_cast<const kf32x4*>(d.xn + (size_t)f * d.ldx + sym * d.two_kin + 4 * c4);
    }
}

template <int NU>
__device__ __forceinline__ void dweff_epilogue(const DweffArgs& d, const int ntn, const f32x16& acc, const int m0,
                                               const int n0, const kf32x4 (&rx)[2 * NU]) {
    constexpr int RA = 64 * ((NU + 1) / 2);         // LDS row stride of the x tile (floats); odd NU: 32 columns of slack
    constexpr int NP = 2 * NU;                      // float4 pieces per thread: 64 rows x 8 NU float4 / 256 threads
    constexpr int NCH = NU / 2;
    constexpr bool ODD = (NU & 1) != 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sX = smem;                   // [64][RA]
    float* sD = smem + 64 * RA;         // [64][64]
    float* sC = sD + 64 * 64;           // [2][64]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int q = (m0 >> 6) * ntn + (n0 >> 6);

    // 1. dfft tile -> LDS; rows past the batch hold clamped duplicates in the accumulators: zero them
    {
        const int l31 = lane & 31, h = lane >> 5;
        const int wm0 = (wid >> 1) * 32, col = (wid & 1) * 32 + l31;
        float cs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm0 + (r & 3) + 8 * (r >> 2) + 4 * h;          // row parity = r & 1
            const float v = (m0 + row < d.batch) ? acc[r] : 0.f;
            sD[row * 64 + (col ^ ((r & 1) << 5))] = v;
            cs += v;
        }
        cs += __shfl_xor(cs, 32, 64);
        if (h == 0) sC[(wid >> 1) * 64 + col] = cs;
    }
    // 2. x tile -> LDS (odd rows: 32-column halves swapped)
#pragma unroll
    for (int v = 0; v < NP; ++v) {
        int row, c4;
        dweff_piece<NU>(tid, v, row, c4);
        *reinterpret_cast<kf32x4*>(sX + row * RA + ((unsigned)(4 * c4) ^ ((unsigned)(row & 1) << 5))) = rx[v];
    }
    __syncthreads();
    if (tid < 64) out_store<2>(d.colsum + (size_t)q * 64 + tid, sC[tid] + sC[64 + tid]);

    // 3. partial[m][n] = sum over the tile's 64 frames: 16 k-steps, fragments of step ks+1 read under the MFMAs of ks
    const int c = lane & 15, kq = lane >> 4, hA = wid >> 1, hB = wid & 1;
    const unsigned sw = (unsigned)(kq & 1) << 5;
    kf32x4 aw[NCH > 0 ? NCH : 1][2][2];
    kf32x4 ao[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) { aw[ch][0][0][r] = 0.f; aw[ch][0][1][r] = 0.f; aw[ch][1][0][r] = 0.f; aw[ch][1][1][r] = 0.f; }
        ao[0][r] = 0.f; ao[1][r] = 0.f;
    }
    const float* xr = sX + kq * RA;
    const float* dr = sD + kq * 64 + ((unsigned)(32 * hB + 2 * c) ^ sw);
    float2 fb[2], fa[2][NCH > 0 ? NCH : 1];
    float fo[2];
    auto frag = [&](const int ks, const int buf) {
        fb[buf] = *reinterpret_cast<const float2*>(dr + 4 * ks * 64);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
            fa[buf][ch] = *reinterpret_cast<const float2*>(xr + 4 * ks * RA + ((unsigned)(64 * ch + 32 * hA + 2 * c) ^ sw));
        // odd unit (32 rows): wave hA takes the row pairs (2n, 2n+1) with n = hA (mod 2): MFMA row i = c is dWeff row
        // 32 (NU-1) + 4 (c/2) + 2 hA + c%2, so that both rows of a pair end up in one lane (r = 0,1 and r = 2,3)
        if constexpr (ODD) fo[buf] = xr[4 * ks * RA + ((unsigned)(32 * (NU - 1) + 4 * (c >> 1) + 2 * hA + (c & 1)) ^ sw)];
    };
    frag(0, 0);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        if (ks + 1 < 16) frag(ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const float2 b = fb[ks & 1];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const float2 a = fa[ks & 1][ch];
            aw[ch][0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, aw[ch][0][0], 0, 0, 0);
            aw[ch][0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.y, aw[ch][0][1], 0, 0, 0);
            aw[ch][1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.x, aw[ch][1][0], 0, 0, 0);
            aw[ch][1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, aw[ch][1][1], 0, 0, 0);
        }
        if constexpr (ODD) {
            const float av = fo[ks & 1];
            ao[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.x, ao[0], 0, 0, 0);
            ao[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.y, ao[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // 4. the tile's FOLDED partial [kin][32][{a, b}]: lane (c, kq) of wave (hA, hB) holds, per chunk and r, the 2x2 block
    //    rows 2n, 2n+1 (s = 0, 1) x columns 2f, 2f+1 (t = 0, 1) with n = 32 ch + 16 hA + 4 kq + r, f = 16 hB + c:
    //    a = [2n][2f] - [2n+1][2f+1], b = [2n][2f+1] - [2n+1][2f] (complex.py:185-188 backward); 16 lanes = 128 contiguous bytes
    float* P = d.partial + (size_t)q * (16 * NU * 64) + 2 * (16 * hB + c);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            out_store2<2>(P + (size_t)(32 * ch + 16 * hA + 4 * kq + r) * 64, aw[ch][0][0][r] - aw[ch][1][1][r],
                          aw[ch][0][1][r] - aw[ch][1][0][r]);
    if constexpr (ODD) {
        // odd unit: accumulator row i = 4 kq + r is dWeff row 32 (NU-1) + 8 kq + 4 (r/2) + 2 hA + r%2: n = 16 (NU-1) + 4 kq + 2 (r/2) + hA
#pragma unroll
        for (int rp = 0; rp < 2; ++rp)
            out_store2<2>(P + (size_t)(16 * (NU - 1) + 4 * kq + 2 * rp + hA) * 64, ao[0][2 * rp] - ao[1][2 * rp + 1],
                          ao[1][2 * rp] - ao[0][2 * rp + 1]);
    }
}

// Block timeline instrumentation (`make trace`: -DDCCN_TRACE, a separate build used by tools/blocktrace.py only):
// thread 0 of every block stamps s_memrealtime (100 MHz) at entry / after the k-loop / at exit and its hardware id.
#ifdef DCCN_TRACE
__device__ unsigned long long* g_trace = nullptr;      // [blocks][4]
#define DCCN_TRACE_MARK(slot, extra)                                                                  \
    do {                                                                                              \
        if (threadIdx.x == 0 && g_trace) g_trace[(size_t)blockIdx.x * 4 + (slot)] = (slot) == 3 ? (unsigned long long)(extra) : wall_clock64(); \
    } while (0)
__device__ __forceinline__ unsigned long long trace_hwid(int role) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
    return ((unsigned long long)role << 48) | ((unsigned long long)(xcc & 0xf) << 32) | hw;
}
#else
#define DCCN_TRACE_MARK(slot, extra) do { } while (0)
#endif

struct NormRideArgs {      // R0 of the next batch on `blocks` extra workgroups (0: none)
    const float* x; float* y; double* power;
    int batch, cols, blocks;
    float eps, peak;
    int trail;             // 0: they lead the grid; 1: they close it -- dispatched into the slots the last dW items free, where
                           // the launch otherwise ends with a mostly idle chip (profiles/r03_blocktrace.txt: 4-5 us of tail)
};

// grid: [R0 blocks][tail-finalize blocks][dX tiles (+ dWeff partial)][dW (tile, k range) items]
// (two blocks per CU by LDS => two waves per SIMD: the allocator may use up to 256 registers, no spills for the staged x rows)
template <int BK, int NU>
__global__ __launch_bounds__(kGemmThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void rx_bwd_fused_kernel(const GemmParams px, const GemmParams pw, const DweffArgs de,
                                                                    const int nx, const int tw, const NormRideArgs nr,
                                                                    const int fin_blocks, const TailFinalizeArgs fin,
                                                                    const dccn_adam_hparams hp) {
    int b = (int)blockIdx.x;
    DCCN_TRACE_MARK(0, 0);
    stamp_mark(px.stamp, 0);
    const int nlead = nr.trail ? 0 : nr.blocks;
    const int ntrail0 = (int)gridDim.x - (nr.trail ? nr.blocks : 0);           // first trailing R0 block
    if (b < nlead || b >= ntrail0) {
        norm_fused_body<kNormFusedCG, kNormFusedRPT>(nr.x, nr.y, nr.batch, nr.cols, nr.eps, nr.peak, nr.power, nullptr,
                                                     nullptr, nullptr, hp, b < nlead ? b : b - ntrail0, nr.blocks);
        DCCN_TRACE_MARK(3, trace_hwid(0));
        DCCN_TRACE_MARK(2, 0);
        stamp_mark(px.stamp, 1);
        return;
    }
    b -= nlead;
    if (b < fin_blocks) {
        demod_tail_finalize_body(fin, b);
        DCCN_TRACE_MARK(3, trace_hwid(1));
        DCCN_TRACE_MARK(2, 0);
        stamp_mark(px.stamp, 1);
        return;
    }
    b -= fin_blocks;
    if (b < nx) {
        f32x16 acc[1][1];
        int m0, n0;
        float cs;
        const int ntn = (px.N + 63) / 64;
        if (de.prio == 1) __builtin_amdgcn_s_setprio(1);
        else if (de.prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (de.prio == 3) __builtin_amdgcn_s_setprio(3);
        kf32x4 st[2 * NU];          // (a native vector type: HIP's float4 struct copies become memcpys that keep the array in scratch)
        {
            const int tile = xcd_tile(b, nx);                  // the tile gemm_mainloop is about to take
            dweff_request<NU>(de, (tile / ntn) * 64, (tile % ntn) * 64, st);
        }
        __builtin_amdgcn_sched_barrier(0);
        gemm_mainloop<OP_KCONTIG, OP_KCONTIG, 64, 64, BK, 0, true>(px, b, nx, 0, acc, m0, n0, cs);
        DCCN_TRACE_MARK(1, 0);
        if (px.C != nullptr) gemm_store<64, 64, 0>(px, 0, acc, m0, n0, cs);
        dweff_epilogue<NU>(de, ntn, acc[0][0], m0, n0, st);
        DCCN_TRACE_MARK(3, trace_hwid(2));
    } else {
        const int c = b - nx;
        kmajor_block<1, BK>(pw, c % tw, tw, c / tw);
        DCCN_TRACE_MARK(3, trace_hwid(3));
    }
    DCCN_TRACE_MARK(2, 0);
    stamp_mark(px.stamp, 1);
}

template <int NU>
static int launch_rx_bwd_fused(const GemmParams& px, const GemmParams& pw, const DweffArgs& de, int splits_w,
                               const NormRideArgs& nr, const TailFinalizeArgs& fin, dccn_adam_hparams hp, hipStream_t s) {
    constexpr int BK = 64;
    constexpr size_t sx = gemm_smem_bytes<OP_KCONTIG, OP_KCONTIG, 64, 64, BK>();
    constexpr size_t s1 = sx > kmajor_smem_bytes<BK>() ? sx : kmajor_smem_bytes<BK>();
    constexpr size_t smem = s1 > dweff_smem_bytes<NU>() ? s1 : dweff_smem_bytes<NU>();
    auto kern = rx_bwd_fused_kernel<BK, NU>;
    DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), smem));
    const int nx = ceil_div(px.N, 64) * ceil_div(px.M, 64);
    const int tw = ceil_div(pw.N, 64) * ceil_div(pw.M, 64);
    const int fin_blocks = fin.metrics != nullptr ? tail_finalize_blocks(fin.P) : 0;     // stand-alone operator: none
    hipLaunchKernelGGL(kern, dim3(nr.blocks + fin_blocks + nx + tw * splits_w), dim3(kGemmThreads), smem, s, px, pw, de, nx,
                       tw, nr, fin_blocks, fin, hp);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

}  // namespace dccn
