// R3-R6 fused: the per-data-cell demodulation MLP (dev/py/model.py:1278-1291), the double
// softmax cross-entropy, hard decision and 2x2 confusion counts
// (dev/py/ofdmreceiver_np.py:154-169), and -- in the training variant -- the whole backward
// of that tail in the same pass (the loss scale 1/count is known before the launch, so no
// second sweep over the activations is needed).
//
// One block per CU; a thread walks its cells with a grid stride, prefetching the next cell's
// inputs before it evaluates the current one (one wave per SIMD has no other latency hiding).
// The ~40-200 tail weights are read through uniform (scalar-cache) loads.  Per-thread gradient /
// metric accumulators are reduced wave (DPP) -> block (LDS) -> per-block slab; a small second
// kernel sums the slabs with one wave per output in a fixed order, so everything is deterministic.
#pragma once
#include "common.h"

namespace dccn {

constexpr float kLeaky = 0.2f;
constexpr int kTailBlocksMax = 512;   // slab capacity of the workspace / finalize stage
constexpr int kTailBlocks = 256;      // one block per CU (two per CU measured no better for nbits<=2 and worse for nbits>=3: 256 VGPRs)
constexpr int kTailThreads = 256;

__host__ __device__ constexpr int tail_param_count(int nb) {
    return 2 * (1 << nb) + (1 << nb) + ((1 << nb) + 2) * 2 * nb + 2 * nb;
}

__device__ __forceinline__ float leaky_relu(float x) { return fmaxf(kLeaky * x, x); }

// The two softmaxes of a bit pair need exp of a non-positive argument, log of a value in (1, 2] and reciprocals of
// values in (1, 2]: the hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, 1 ulp each) do that in 1-2
// instructions where the IEEE-exact library forms (range reduction, denormal scaling, division fix-up) take 8-12.
// Error budget: exp2(x*log2e) carries the rounding of the product, |x| * 6e-8 relative (<= 2e-6 for |x| <= 30, on a
// result that is then <= 1e-13); log and reciprocal stay at 1-2 ulp -- all far inside the 1e-5 parity bound, and the hard
// decision p1 > p0 is unaffected (both probabilities are the same positive reciprocal times e0 / e1).
__device__ __forceinline__ float exp_nonpos(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float log_1_2(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }

struct TailBlockMetrics {      // one per block in the workspace
    double ce_sum;
    long long conf[4];
};

template <int NB>
struct TailCellIn {
    float2 z;
    int lab[NB];
};

template <int NB>
__device__ __forceinline__ TailCellIn<NB> tail_load_cell(const float* __restrict__ z, const int32_t* __restrict__ bits,
                                                         long long cell, long long cells) {
    TailCellIn<NB> c;
    const long long cc = cell < cells ? cell : cells - 1;        // clamped: the load is always legal
    c.z = *reinterpret_cast<const float2*>(z + 2 * cc);
#pragma unroll
    for (int j = 0; j < NB; ++j) c.lab[j] = bits[cc * NB + j];
    return c;
}

// Per-lane running sums of the tail: weight gradients (training), CE sum, confusion tallies.
template <int NB, bool BWD>
struct TailLaneAcc {
    static constexpr int P = tail_param_count(NB);
    float g[BWD ? P : 1];
    double ce;
    int c00, c01, c10, c11;
    __device__ __forceinline__ void clear() {
        if constexpr (BWD) {
#pragma unroll
            for (int i = 0; i < P; ++i) g[i] = 0.f;
        } else {
            g[0] = 0.f;
        }
        ce = 0.0;
        c00 = c01 = c10 = c11 = 0;
    }
};

// W data cells through R3-R6 (and back) in one instruction stream: z = (I,Q) of the dense output, lab = the NB label
// bits, sw = tail weights (uniform address: scalar loads for NB <= 2, an LDS copy for NB >= 3).  Every statement is
// written for all W cells before the next one, so the W independent dependency chains sit next to each other and the
// long-latency steps (exp, log, reciprocal refinement) of one cell are covered by the others even with a single wave
// per SIMD; per cell the operations and their order are those of the W = 1 form, so any W gives the same bits.
// valid[u] == false: the cell contributes nothing to the sums (its dz is 0); prob_cell[u] (nullable) -> NB float2.
// Shared by demod_tail_kernel and the fused dense-forward epilogue (gemm16.h).
template <int NB, bool BWD, int W>
__device__ __forceinline__ void tail_cells(const float (&z0)[W], const float (&z1)[W], const int (&lab)[W][NB],
                                           const bool (&valid)[W], const float* __restrict__ sw, const float inv_count,
                                           float* const (&prob_cell)[W], TailLaneAcc<NB, BWD>& A, float2 (&dzv)[W]) {
    constexpr int M = 1 << NB;
    constexpr int O = 2 * NB;
    constexpr int oW1 = 0, oB1 = 2 * M, oW2 = 3 * M, oB2 = 3 * M + (M + 2) * O;
    // the two small matrix products of the forward are fused multiply-adds with the bias as the chain's start value
    // (36 VALU instructions fewer per QPSK cell); demod_tail_quad4_kernel evaluates the same expressions, so training
    // and evaluation produce identical probabilities for every modulation
    constexpr bool FMA_FWD = true;
    float c[M + 2][W], pre1[M][W];
#pragma unroll
    for (int j = 0; j < M; ++j) {
#pragma unroll
        for (int u = 0; u < W; ++u) {
            if constexpr (FMA_FWD)
                pre1[j][u] = __builtin_fmaf(z1[u], sw[oW1 + M + j], __builtin_fmaf(z0[u], sw[oW1 + j], sw[oB1 + j]));
            else
                pre1[j][u] = (z0[u] * sw[oW1 + j] + z1[u] * sw[oW1 + M + j]) + sw[oB1 + j];
            c[j][u] = leaky_relu(pre1[j][u]);
        }
    }
#pragma unroll
    for (int u = 0; u < W; ++u) {
        c[M][u] = z0[u];
        c[M + 1][u] = z1[u];
    }
    float pre2[O][W];
#pragma unroll
    for (int o = 0; o < O; ++o) {
        float s[W];
#pragma unroll
        for (int u = 0; u < W; ++u) s[u] = FMA_FWD ? sw[oB2 + o] : 0.f;
#pragma unroll
        for (int i = 0; i < M + 2; ++i)
#pragma unroll
            for (int u = 0; u < W; ++u) {
                if constexpr (FMA_FWD) s[u] = __builtin_fmaf(c[i][u], sw[oW2 + i * O + o], s[u]);
                else s[u] += c[i][u] * sw[oW2 + i * O + o];
            }
#pragma unroll
        for (int u = 0; u < W; ++u) pre2[o][u] = FMA_FWD ? s[u] : s[u] + sw[oB2 + o];
    }
    float dpre2[O][W];
    float inv_eff[W];
#pragma unroll
    for (int u = 0; u < W; ++u) inv_eff[u] = valid[u] ? inv_count : 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        float p0[W], p1[W], f0[W], f1[W], fs[W];
        bool big[W];
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const float u0 = leaky_relu(pre2[2 * j][u]), u1 = leaky_relu(pre2[2 * j + 1][u]);
            // softmax over the pair: exp(u - max) is exactly 1 for the larger logit, so one expf suffices
            // (bit-identical to evaluating both); likewise for the second softmax on the probabilities
            // (the smaller logit minus the larger one is -|u0 - u1| either way: abs/neg are free source modifiers)
            big[u] = u1 > u0;
            const float eo = exp_nonpos(-fabsf(u0 - u1));
            const float e0 = big[u] ? eo : 1.0f, e1 = big[u] ? 1.0f : eo;
            const float res = rcp_fast(e0 + e1);
            p0[u] = e0 * res;
            p1[u] = e1 * res;
        }
#pragma unroll
        for (int u = 0; u < W; ++u) {
            if (prob_cell[u] != nullptr) *reinterpret_cast<float2*>(prob_cell[u] + 2 * j) = make_float2(p0[u], p1[u]);
        }
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const int label = lab[u][j];
            // second softmax on the probabilities (softmax_cross_entropy_with_logits_v2)
            const bool p1_big = p1[u] > p0[u];
            const float mx2 = p1_big ? p1[u] : p0[u];
            const float fo = exp_nonpos(-fabsf(p0[u] - p1[u]));
            f0[u] = p1_big ? fo : 1.0f;
            f1[u] = p1_big ? 1.0f : fo;
            fs[u] = f0[u] + f1[u];
            const float lse = log_1_2(fs[u]) + mx2;
            const float ce = lse - (label ? p1[u] : p0[u]);
            A.ce += (double)(valid[u] ? ce : 0.f);
            const int pred = (p1[u] > p0[u]) ? 1 : 0;    // argmax, first index on ties
            const int l1 = label != 0 ? 1 : 0;           // branch-free tallies (divergent ifs cost exec-mask regions)
            const int vb = valid[u] ? 1 : 0;
            A.c00 += (1 - l1) & (1 - pred) & vb;
            A.c01 += (1 - l1) & pred & vb;
            A.c10 += l1 & (1 - pred) & vb;
            A.c11 += l1 & pred & vb;
        }
        if constexpr (BWD) {
#pragma unroll
            for (int u = 0; u < W; ++u) {
                const int label = lab[u][j];
                const float rfs = rcp_fast(fs[u]);
                const float q0 = f0[u] * rfs, q1 = f1[u] * rfs;
                const float g0 = (q0 - (label ? 0.f : 1.f)) * inv_eff[u];
                const float g1 = (q1 - (label ? 1.f : 0.f)) * inv_eff[u];
                const float dot = g0 * p0[u] + g1 * p1[u];
                const float du0 = p0[u] * (g0 - dot), du1 = p1[u] * (g1 - dot);
                dpre2[2 * j][u] = du0 * (pre2[2 * j][u] > 0.f ? 1.f : kLeaky);
                dpre2[2 * j + 1][u] = du1 * (pre2[2 * j + 1][u] > 0.f ? 1.f : kLeaky);
            }
        }
    }
    if constexpr (BWD) {
        float dc[M + 2][W];
#pragma unroll
        for (int i = 0; i < M + 2; ++i) {
            float s[W];
#pragma unroll
            for (int u = 0; u < W; ++u) s[u] = 0.f;
#pragma unroll
            for (int o = 0; o < O; ++o) {
                // backward-only sums use fused multiply-adds (one rounding, half the VALU instructions); the
                // forward stays unfused so that training and evaluation produce identical probabilities
#pragma unroll
                for (int u = 0; u < W; ++u) {
                    A.g[oW2 + i * O + o] = __builtin_fmaf(c[i][u], dpre2[o][u], A.g[oW2 + i * O + o]);
                    s[u] = __builtin_fmaf(dpre2[o][u], sw[oW2 + i * O + o], s[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < W; ++u) dc[i][u] = s[u];
        }
#pragma unroll
        for (int o = 0; o < O; ++o)
#pragma unroll
            for (int u = 0; u < W; ++u) A.g[oB2 + o] += dpre2[o][u];
        float d0[W], d1[W];
#pragma unroll
        for (int u = 0; u < W; ++u) {
            d0[u] = dc[M][u];
            d1[u] = dc[M + 1][u];
        }
#pragma unroll
        for (int j = 0; j < M; ++j) {
#pragma unroll
            for (int u = 0; u < W; ++u) {
                const float dp = dc[j][u] * (pre1[j][u] > 0.f ? 1.f : kLeaky);
                A.g[oW1 + j] = __builtin_fmaf(z0[u], dp, A.g[oW1 + j]);
                A.g[oW1 + M + j] = __builtin_fmaf(z1[u], dp, A.g[oW1 + M + j]);
                A.g[oB1 + j] += dp;
                d0[u] = __builtin_fmaf(dp, sw[oW1 + j], d0[u]);
                d1[u] = __builtin_fmaf(dp, sw[oW1 + M + j], d1[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < W; ++u) dzv[u] = make_float2(d0[u], d1[u]);
    } else {
#pragma unroll
        for (int u = 0; u < W; ++u) dzv[u] = make_float2(0.f, 0.f);
    }
}

// one cell (W = 1)
template <int NB, bool BWD>
__device__ __forceinline__ float2 tail_cell(const float z0, const float z1, const int (&lab)[NB],
                                            const float* __restrict__ sw, const float inv_count,
                                            float* __restrict__ prob_cell, TailLaneAcc<NB, BWD>& A) {
    const float a0[1] = {z0}, a1[1] = {z1};
    int l[1][NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) l[0][j] = lab[j];
    const bool v[1] = {true};
    float* const pc[1] = {prob_cell};
    float2 d[1];
    tail_cells<NB, BWD, 1>(a0, a1, l, v, sw, inv_count, pc, A, d);
    return d[0];
}

// LDS the block reduction of the lane accumulators needs (floats), for NT threads
template <int NB, bool BWD>
constexpr int tail_reduce_lds_floats(int nt) {
    // smat [nt/4][P|1] floats + per-wave ce (double) + per-wave conf (4 ints)
    return (BWD ? (nt / 4) * (tail_param_count(NB) | 1) : 0) + (nt / 64) * 2 + (nt / 64) * 4 + 2;
}

// Block reduction: DPP within the wave, LDS across the waves, fixed order (deterministic); writes the block's slab.
// lds must hold tail_reduce_lds_floats<NB,BWD>(NT) floats, 8-byte aligned.  Starts and ends with barriers of its own.
template <int NB, bool BWD, int NT>
__device__ __forceinline__ void tail_block_reduce(TailLaneAcc<NB, BWD>& A, float* __restrict__ lds,
                                                  TailBlockMetrics* __restrict__ blk_metrics,
                                                  float* __restrict__ blk_grads, const int slab) {
    constexpr int P = tail_param_count(NB);
    constexpr int PS = P | 1;                                  // odd row stride: column reads spread over the banks
    constexpr int NW = NT / 64, NQ = NT / 4;
    double* sce = reinterpret_cast<double*>(lds);              // [NW]
    int* sconf = reinterpret_cast<int*>(lds + 2 * NW);         // [NW][4]
    float* smat = lds + 2 * NW + 4 * NW + 2;                   // [NQ][PS] partial gradient sums
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const double ce = wave_sum(A.ce);
    const int c00 = wave_sum(A.c00), c01 = wave_sum(A.c01), c10 = wave_sum(A.c10), c11 = wave_sum(A.c11);
    if (lane == 0) {
        sce[wid] = ce;
        sconf[wid * 4 + 0] = c00; sconf[wid * 4 + 1] = c01; sconf[wid * 4 + 2] = c10; sconf[wid * 4 + 3] = c11;
    }
    if constexpr (BWD) {
        // gradients: sum over the 4 lanes of a quad on the DPP crossbar (2 adds per value instead of a full wave
        // reduction per value), park the quad sums per parameter in LDS ...
        // (the stores stand in ONE exec-mask region behind the sums: written inside the loop above, each of the P conditional
        // stores was its own s_and_saveexec / s_or pair -- 92 of them for 8-QAM)
        const int quad = threadIdx.x >> 2;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            float v = A.g[i];
            v += __builtin_bit_cast(float, dpp_mov_i32(__builtin_bit_cast(int, v), 0));
            v += __builtin_bit_cast(float, dpp_mov_i32(__builtin_bit_cast(int, v), 1));
            A.g[i] = v;
        }
        if ((lane & 3) == 0) {
#pragma unroll
            for (int i = 0; i < P; ++i) smat[quad * PS + i] = A.g[i];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        TailBlockMetrics bm;
        if constexpr (NW == 4) {
            bm.ce_sum = (sce[0] + sce[1]) + (sce[2] + sce[3]);
        } else {
            double t = 0.0;
            for (int w = 0; w < NW; w += 4) t += (sce[w] + sce[w + 1]) + (sce[w + 2] + sce[w + 3]);
            bm.ce_sum = t;
        }
        for (int k = 0; k < 4; ++k) {
            long long t = 0;
            for (int w = 0; w < NW; ++w) t += sconf[w * 4 + k];
            bm.conf[k] = t;
        }
        blk_metrics[slab] = bm;
    }
    if constexpr (BWD) {
        // ... then 4 threads per parameter column add NQ/4 quad sums each (fixed order) and combine on the crossbar
        const int slot = threadIdx.x >> 2, part = threadIdx.x & 3;
        for (int col0 = 0; col0 < P; col0 += NQ) {
            const int col = col0 + slot;
            const int cc = col < P ? col : P - 1;
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < NQ / 4; ++r) v += smat[(part * (NQ / 4) + r) * PS + cc];
            v += __builtin_bit_cast(float, dpp_mov_i32(__builtin_bit_cast(int, v), 0));
            v += __builtin_bit_cast(float, dpp_mov_i32(__builtin_bit_cast(int, v), 1));
            if (part == 0 && col < P) blk_grads[(size_t)slab * P + col] = v;
        }
    }
}

template <int NB, bool BWD>
__global__ __launch_bounds__(kTailThreads) void demod_tail_kernel(
    const float* z_, const int32_t* bits_, const float* tailp_, float* prob_, float* dz_, long long cells,
    TailBlockMetrics* blk_metrics_, float* blk_grads_, const ChainOffs co) {
    const long long coff = co.off[blockIdx.z];                     // chain groups (common.h)
    const float* __restrict__ z = chain_at(z_, coff);
    const int32_t* __restrict__ bits = chain_at(bits_, coff);
    const float* __restrict__ tailp = chain_at(tailp_, coff);
    float* __restrict__ prob = chain_at(prob_, coff);
    float* __restrict__ dz = chain_at(dz_, coff);
    TailBlockMetrics* __restrict__ blk_metrics = chain_at(blk_metrics_, coff);
    float* __restrict__ blk_grads = chain_at(blk_grads_, coff);
    constexpr int P = tail_param_count(NB);
    __shared__ __attribute__((aligned(8))) float sred[tail_reduce_lds_floats<NB, BWD>(kTailThreads)];

    // nbits <= 2: the <= 40 weights are read through uniform addresses -> scalar (SGPR) loads.  nbits >= 3: 90 / 200
    // weights do not fit the SGPR file (the compiler spilled hundreds of them into VGPR lanes); they are staged in
    // LDS once and re-read each cell as broadcast loads (the per-iteration compiler barrier keeps them from being
    // hoisted into 200 live VGPRs).
    constexpr bool LDSW = NB >= 3;
    __shared__ float swl[LDSW ? P : 1];
    if constexpr (LDSW) {
        for (int i = threadIdx.x; i < P; i += kTailThreads) swl[i] = tailp[i];
        __syncthreads();
    }
    const float* __restrict__ sw = LDSW ? swl : tailp;

    TailLaneAcc<NB, BWD> A;
    A.clear();
    const float inv_count = 1.0f / (float)(cells * NB);

    // W cells per iteration (cell, cell + stride, ...) in one interleaved instruction stream, the next W prefetched
    constexpr int W = NB <= 2 ? 2 : 1;
    const long long stride = (long long)gridDim.x * kTailThreads;
    long long cell = (long long)blockIdx.x * kTailThreads + threadIdx.x;
    TailCellIn<NB> cur[W];
#pragma unroll
    for (int u = 0; u < W; ++u) cur[u] = tail_load_cell<NB>(z, bits, cell + u * stride, cells);
    while (cell < cells) {
        if constexpr (LDSW) asm volatile("" ::: "memory");
        TailCellIn<NB> nxt[W];
#pragma unroll
        for (int u = 0; u < W; ++u) nxt[u] = tail_load_cell<NB>(z, bits, cell + (W + u) * stride, cells);     // prefetch
        float z0[W], z1[W];
        int lab[W][NB];
        bool valid[W];
        float* pc[W];
        float2 d[W];
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const long long cu = cell + u * stride;
            z0[u] = cur[u].z.x;
            z1[u] = cur[u].z.y;
#pragma unroll
            for (int j = 0; j < NB; ++j) lab[u][j] = cur[u].lab[j];
            valid[u] = cu < cells;
            pc[u] = (prob != nullptr && valid[u]) ? prob + cu * NB * 2 : nullptr;
        }
        tail_cells<NB, BWD, W>(z0, z1, lab, valid, sw, inv_count, pc, A, d);
        if constexpr (BWD) {
#pragma unroll
            for (int u = 0; u < W; ++u)
                if (valid[u]) *reinterpret_cast<float2*>(dz + 2 * (cell + u * stride)) = d[u];
        }
#pragma unroll
        for (int u = 0; u < W; ++u) cur[u] = nxt[u];
        cell += W * stride;
    }
    tail_block_reduce<NB, BWD, kTailThreads>(A, sred, blk_metrics, blk_grads, (int)blockIdx.x);
}

// ---- nbits = 4 training: four lanes per cell ------------------------------------------------------------
// One thread per cell needs 200 gradient accumulators (256 VGPRs + AGPR spills, one wave per SIMD).  Here lane q of
// a quad owns bit q of the cell: the two logits (2q, 2q+1), their softmax / cross-entropy / decision, the matching
// two columns of the dense_1 gradient (36 + 2 accumulators) and hidden units 4q..4q+3 of the 1x1-conv gradient (12):
// 50 accumulators per lane.  What a lane does not own it gets from its quad on the DPP crossbar (the backward into
// the 18 concatenated features and the two dz sums).  The forward uses the same operations in the same order as
// demod_tail_kernel, so training and evaluation probabilities stay bit-identical.
// Shared by demod_tail_quad4_kernel (z from memory) and the fused dense-forward epilogue (gemm16.h: z from the
// accumulators of the quad's lanes).
struct TailQuad4Acc {
    static constexpr int M = 16, O = 8;
    float g2[M + 2][2], gb2[2], gw1a[4], gw1b[4], gb1[4];
    double ce;
    int c00, c01, c10, c11;
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < M + 2; ++i) g2[i][0] = g2[i][1] = 0.f;
        gb2[0] = gb2[1] = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) gw1a[u] = gw1b[u] = gb1[u] = 0.f;
        ce = 0.0;
        c00 = c01 = c10 = c11 = 0;
    }
};
// the tail weights a lane of the quad form needs, in registers for the whole cell loop (50 values: this lane's two columns
// of dense_1 and their bias, the 1x1-conv rows and bias of its four hidden units) -- re-reading them from LDS for every
// cell (~80 broadcast reads, each a full LDS latency for the single wave of a SIMD) was most of a cell's time
struct TailQuad4W {
    static constexpr int M = 16, O = 8;
    float w2[M + 2][2], b2[2];
    float own_a[4], own_b[4], own_b1[4];
    __device__ __forceinline__ void load(const float* __restrict__ sw, const int q) {
        constexpr int oW1 = 0, oB1 = 2 * M, oW2 = 3 * M, oB2 = 3 * M + (M + 2) * O;
#pragma unroll
        for (int i = 0; i < M + 2; ++i) { w2[i][0] = sw[oW2 + i * O + 2 * q]; w2[i][1] = sw[oW2 + i * O + 2 * q + 1]; }
        b2[0] = sw[oB2 + 2 * q];
        b2[1] = sw[oB2 + 2 * q + 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            own_a[u] = sw[oW1 + 4 * q + u];
            own_b[u] = sw[oW1 + M + 4 * q + u];
            own_b1[u] = sw[oB1 + 4 * q + u];
        }
    }
};
// lane k of every quad to all four lanes
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), K * 0x55, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, dpp_mov_i32(__builtin_bit_cast(int, v), 0));
    v += __builtin_bit_cast(float, dpp_mov_i32(__builtin_bit_cast(int, v), 1));
    return v;
}
constexpr int tail_quad4_lds_floats(int nt) { return (nt / 4) * (tail_param_count(4) | 1) + (nt / 64) * 2 + (nt / 64) * 4 + 2; }

// one cell through R3-R6 and back on a quad of lanes: (z0, z1) and `valid` are quad-uniform, `label` is bit q of the
// cell, swl the tail weights in LDS, prob_q (nullable) this lane's float2 of `output`; returns dz of the cell (quad-uniform)
__device__ __forceinline__ float2 tail_quad4_cell(const float z0, const float z1, const int label, const bool valid,
                                                  const TailQuad4W& W, const float inv_count,
                                                  float* __restrict__ prob_q, const int q, TailQuad4Acc& A) {
    constexpr int M = 16;
    // a[4*q + u] without dynamic register indexing: three lane-mask selects.  (Written as ?: the compiler turned the
    // eight picks into nested exec-mask regions -- 130 scalar instructions per cell around ~330 vector ones -- although
    // every quad takes all four paths.)
    const unsigned long long qm1 = __builtin_amdgcn_ballot_w64(q == 1), qm2 = __builtin_amdgcn_ballot_w64(q == 2),
                             qm3 = __builtin_amdgcn_ballot_w64(q == 3);
    auto sel = [](float a, float b, unsigned long long m) {
        float r;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
        return r;
    };
    auto pick4 = [&](const float* a, int u) { return sel(sel(sel(a[u], a[4 + u], qm1), a[8 + u], qm2), a[12 + u], qm3); };
    // the 1x1 conv: every lane evaluates its own four hidden units (same expression as tail_cells) and the quad trades them
    // on the DPP crossbar -- 16 + 16 instructions where four redundant evaluations of all 16 units took 64
    float c[M + 2], pre_own[4], c_own[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        pre_own[u] = __builtin_fmaf(z1, W.own_b[u], __builtin_fmaf(z0, W.own_a[u], W.own_b1[u]));
        c_own[u] = leaky_relu(pre_own[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        c[u] = quad_bcast<0>(c_own[u]);
        c[4 + u] = quad_bcast<1>(c_own[u]);
        c[8 + u] = quad_bcast<2>(c_own[u]);
        c[12 + u] = quad_bcast<3>(c_own[u]);
    }
    c[M] = z0;
    c[M + 1] = z1;
    float s0 = W.b2[0], s1 = W.b2[1];                // this lane's two columns of dense_1
#pragma unroll
    for (int i = 0; i < M + 2; ++i) {
        s0 = __builtin_fmaf(c[i], W.w2[i][0], s0);
        s1 = __builtin_fmaf(c[i], W.w2[i][1], s1);
    }
    const float p20 = s0, p21 = s1;
    const float u0 = leaky_relu(p20), u1 = leaky_relu(p21);
    const bool u1_big = u1 > u0;
    const float eo = exp_nonpos(u1_big ? (u0 - u1) : (u1 - u0));
    const float e0 = u1_big ? eo : 1.0f, e1 = u1_big ? 1.0f : eo;
    const float res = rcp_fast(e0 + e1);
    const float p0 = e0 * res, p1 = e1 * res;
    if (prob_q != nullptr) *reinterpret_cast<float2*>(prob_q) = make_float2(p0, p1);
    const bool p1_big = p1 > p0;
    const float mx2 = p1_big ? p1 : p0;
    const float fo = exp_nonpos(p1_big ? (p0 - p1) : (p1 - p0));
    const float f0 = p1_big ? fo : 1.0f, f1 = p1_big ? 1.0f : fo;
    const float fs = f0 + f1;
    const float lse = log_1_2(fs) + mx2;
    const float ce = lse - (label ? p1 : p0);
    A.ce += (double)(valid ? ce : 0.f);
    const int pred = (p1 > p0) ? 1 : 0;
    const int l1 = label != 0 ? 1 : 0;
    const int vb = valid ? 1 : 0;
    A.c00 += (1 - l1) & (1 - pred) & vb;
    A.c01 += (1 - l1) & pred & vb;
    A.c10 += l1 & (1 - pred) & vb;
    A.c11 += l1 & pred & vb;
    // backward of this lane's bit
    const float inv_eff = valid ? inv_count : 0.f;
    const float rfs = rcp_fast(fs);
    const float q0 = f0 * rfs, q1 = f1 * rfs;
    const float ga = (q0 - (label ? 0.f : 1.f)) * inv_eff;
    const float gb = (q1 - (label ? 1.f : 0.f)) * inv_eff;
    const float dot = ga * p0 + gb * p1;
    const float du0 = p0 * (ga - dot), du1 = p1 * (gb - dot);
    const float d20 = du0 * (p20 > 0.f ? 1.f : kLeaky), d21 = du1 * (p21 > 0.f ? 1.f : kLeaky);
    A.gb2[0] += d20;
    A.gb2[1] += d21;
    float dc[M + 2];
#pragma unroll
    for (int i = 0; i < M + 2; ++i) {
        A.g2[i][0] = __builtin_fmaf(c[i], d20, A.g2[i][0]);
        A.g2[i][1] = __builtin_fmaf(c[i], d21, A.g2[i][1]);
        dc[i] = quad_sum(__builtin_fmaf(d21, W.w2[i][1], d20 * W.w2[i][0]));      // all four bits' contributions
    }
    float d0p = 0.f, d1p = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {                    // hidden units 4q .. 4q+3
        const float dp = pick4(dc, u) * (pre_own[u] > 0.f ? 1.f : kLeaky);
        A.gw1a[u] = __builtin_fmaf(z0, dp, A.gw1a[u]);
        A.gw1b[u] = __builtin_fmaf(z1, dp, A.gw1b[u]);
        A.gb1[u] += dp;
        d0p = __builtin_fmaf(dp, W.own_a[u], d0p);
        d1p = __builtin_fmaf(dp, W.own_b[u], d1p);
    }
    return make_float2(dc[M] + quad_sum(d0p), dc[M + 1] + quad_sum(d1p));
}

// block reduction of the quad-lane accumulators: every quad row of the LDS matrix gets each parameter column from the
// lane that owns it; lds holds tail_quad4_lds_floats(NT) floats (8-byte aligned).  Starts with no barrier of its own:
// the caller must have finished with `lds`; ends without one.
template <int NT>
__device__ __forceinline__ void tail_quad4_block_reduce(TailQuad4Acc& A, float* __restrict__ lds,
                                                        TailBlockMetrics* __restrict__ blk_metrics,
                                                        float* __restrict__ blk_grads, const int slab) {
    constexpr int M = 16, O = 8, P = tail_param_count(4);
    constexpr int oW1 = 0, oB1 = 2 * M, oW2 = 3 * M, oB2 = 3 * M + (M + 2) * O;
    constexpr int PS = P | 1, NW = NT / 64, NQ = NT / 4;
    double* sce = reinterpret_cast<double*>(lds);              // [NW]
    int* sconf = reinterpret_cast<int*>(lds + 2 * NW);         // [NW][4]
    float* smat = lds + 2 * NW + 4 * NW + 2;                   // [NQ][PS]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, q = threadIdx.x & 3;
    const double ce = wave_sum(A.ce);
    const int c00 = wave_sum(A.c00), c01 = wave_sum(A.c01), c10 = wave_sum(A.c10), c11 = wave_sum(A.c11);
    if (lane == 0) {
        sce[wid] = ce;
        sconf[wid * 4 + 0] = c00; sconf[wid * 4 + 1] = c01; sconf[wid * 4 + 2] = c10; sconf[wid * 4 + 3] = c11;
    }
    float* row = smat + (threadIdx.x >> 2) * PS;
#pragma unroll
    for (int i = 0; i < M + 2; ++i) {
        row[oW2 + i * O + 2 * q] = A.g2[i][0];
        row[oW2 + i * O + 2 * q + 1] = A.g2[i][1];
    }
    row[oB2 + 2 * q] = A.gb2[0];
    row[oB2 + 2 * q + 1] = A.gb2[1];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        row[oW1 + 4 * q + u] = A.gw1a[u];
        row[oW1 + M + 4 * q + u] = A.gw1b[u];
        row[oB1 + 4 * q + u] = A.gb1[u];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        TailBlockMetrics bm;
        double t = 0.0;
        for (int w = 0; w < NW; w += 4) t += (sce[w] + sce[w + 1]) + (sce[w + 2] + sce[w + 3]);
        bm.ce_sum = t;
        for (int k = 0; k < 4; ++k) {
            long long c = 0;
            for (int w = 0; w < NW; ++w) c += sconf[w * 4 + k];
            bm.conf[k] = c;
        }
        blk_metrics[slab] = bm;
    }
    const int slot = threadIdx.x >> 2, part = threadIdx.x & 3;
    for (int col0 = 0; col0 < P; col0 += NQ) {
        const int col = col0 + slot;
        const int cc = col < P ? col : P - 1;
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < NQ / 4; ++r) v += smat[(part * (NQ / 4) + r) * PS + cc];
        v = quad_sum(v);
        if (part == 0 && col < P) blk_grads[(size_t)slab * P + col] = v;
    }
}

template <bool WRITE_PROB>
__global__ __launch_bounds__(kTailThreads) void demod_tail_quad4_kernel(
    const float* z_, const int32_t* bits_, const float* tailp_, float* prob_, float* dz_, long long cells,
    TailBlockMetrics* blk_metrics_, float* blk_grads_, unsigned long long* stamp, const ChainOffs co) {
    const long long coff = co.off[blockIdx.z];                     // chain groups (common.h)
    const float* __restrict__ z = chain_at(z_, coff);
    const int32_t* __restrict__ bits = chain_at(bits_, coff);
    const float* __restrict__ tailp = chain_at(tailp_, coff);
    float* __restrict__ prob = chain_at(prob_, coff);
    float* __restrict__ dz = chain_at(dz_, coff);
    TailBlockMetrics* __restrict__ blk_metrics = chain_at(blk_metrics_, coff);
    float* __restrict__ blk_grads = chain_at(blk_grads_, coff);
    constexpr int NB = 4;
    stamp_mark(stamp, 0);
    __shared__ __attribute__((aligned(8))) float sred[tail_quad4_lds_floats(kTailThreads)];
    const int q = threadIdx.x & 3;
    TailQuad4W W;
    W.load(tailp, q);
    TailQuad4Acc A;
    A.clear();
    const float inv_count = 1.0f / (float)(cells * NB);
    const long long stride = (long long)gridDim.x * (kTailThreads / 4);
    // the next cell's operands are requested before this cell's ~250 instructions run: with two waves per SIMD nothing
    // else would cover the load latency of every pass
    long long cell = (long long)blockIdx.x * (kTailThreads / 4) + (threadIdx.x >> 2);
    float2 zv = make_float2(0.f, 0.f);
    int label = 0;
    if (cell < cells) {
        zv = *reinterpret_cast<const float2*>(z + 2 * cell);
        label = bits[cell * NB + q];
    }
    while (cell < cells) {
        const long long nxt = cell + stride;
        float2 zn = zv;
        int ln = label;
        if (nxt < cells) {
            zn = *reinterpret_cast<const float2*>(z + 2 * nxt);
            ln = bits[nxt * NB + q];
        }
        const float2 d = tail_quad4_cell(zv.x, zv.y, label, true, W, inv_count,
                                         WRITE_PROB ? prob + (cell * NB + q) * 2 : nullptr, q, A);
        if (q == 0) *reinterpret_cast<float2*>(dz + 2 * cell) = d;
        zv = zn;
        label = ln;
        cell = nxt;
    }
    tail_quad4_block_reduce<kTailThreads>(A, sred, blk_metrics, blk_grads, (int)blockIdx.x);
    stamp_mark(stamp, 1);
}

// Slab reduction: wave g < P sums gradient column g over the per-block slabs (lane l owns slabs
// l, l+64, ...), wave g == P builds the metrics record, wave g == P+1 (fused receiver step
// only) finishes the mean clipped power of R8 from the normalise kernel's per-block partial sums.
// grid = ceil((P+2)/4) blocks.
struct TailFinalizeArgs {
    const TailBlockMetrics* blk_metrics;
    const float* blk_grads;
    int nblocks, P;
    long long count;
    dccn_metrics* metrics;
    float* dtailp;
    const double* power_partial;
    int n_power;
    double power_denom;
    float* power_out;
    // fused training step: the optimizer's per-step bookkeeping (alpha of THIS step from the current global_step /
    // beta powers, then advance them) rides on one wave of this stage: it runs after the previous step's Adam kernel
    // and before this step's.  nullptr elsewhere.
    dccn_adam_state* adam;
    dccn_adam_hparams hp;
    unsigned* zero_word;        // nullable: hand-off words of a later launch of the same step, reset here: the arrival
    unsigned* zero_flags;       // counter and n_zero_flags flag words at a stride of zero_stride (fresh or recycled
    int n_zero_flags, zero_stride;   // workspace memory may hold anything, including a stale epoch)
    // nullable: the training loop's epoch accumulators acc5 = {ce_mean, berlin, tx_power, noise_power, chan_rms} (equalizer.h
    // EqMonitorArgs): the wave that writes a scalar adds it on (round 6: the monitor rides on the optimizer launch)
    float* mon_acc;
    const float* mon_noise;     // nullable: this batch's noise power, added by the metrics wave
    __device__ __forceinline__ TailFinalizeArgs at_chain(const long long coff) const {   // chain groups (common.h)
        TailFinalizeArgs q = *this;
        q.mon_acc = chain_at(mon_acc, coff); q.mon_noise = chain_at(mon_noise, coff);
        q.blk_metrics = chain_at(blk_metrics, coff); q.blk_grads = chain_at(blk_grads, coff); q.metrics = chain_at(metrics, coff);
        q.dtailp = chain_at(dtailp, coff); q.power_partial = chain_at(power_partial, coff); q.power_out = chain_at(power_out, coff);
        q.adam = chain_at(adam, coff); q.zero_word = chain_at(zero_word, coff); q.zero_flags = chain_at(zero_flags, coff);
        return q;
    }
};
static inline int tail_finalize_blocks(int P) { return (P + 3 + 3) / 4; }

__device__ __forceinline__ void demod_tail_finalize_body(const TailFinalizeArgs& a, const int block) {
    if (a.metrics == nullptr) return;                    // stage already done by an earlier launch of the step
    const TailBlockMetrics* __restrict__ blk_metrics = a.blk_metrics;
    const float* __restrict__ blk_grads = a.blk_grads;
    const int nblocks = a.nblocks, P = a.P, n_power = a.n_power;
    const long long count = a.count;
    dccn_metrics* __restrict__ metrics = a.metrics;
    float* __restrict__ dtailp = a.dtailp;
    const double* __restrict__ power_partial = a.power_partial;
    const double power_denom = a.power_denom;
    float* __restrict__ power_out = a.power_out;
    const int lane = threadIdx.x & 63;
    const int g = block * 4 + (threadIdx.x >> 6);
    if (g < P) {
        // lane l owns slabs l, l+64, ...: batches of 8 independent loads, summed in slab order
        float acc = 0.f;
        for (int b0 = 0; b0 < nblocks; b0 += 8 * 64) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int b = b0 + lane + 64 * q;
                v[q] = (b < nblocks) ? blk_grads[(size_t)b * P + g] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
        }
        const float s = wave_sum(acc);
        if (lane == 0) dtailp[g] = s;
    } else if (g == P) {
        double ce = 0.0;
        long long cf[4] = {0, 0, 0, 0};
        for (int b = lane; b < nblocks; b += 64) {
            ce += blk_metrics[b].ce_sum;
#pragma unroll
            for (int k = 0; k < 4; ++k) cf[k] += blk_metrics[b].conf[k];
        }
        ce = wave_sum(ce);
#pragma unroll
        for (int k = 0; k < 4; ++k) cf[k] = wave_sum(cf[k]);
        if (lane == 0) {
            metrics->ce_sum = ce;
            for (int k = 0; k < 4; ++k) metrics->conf[k] = cf[k];
            metrics->count = count;
            metrics->ce_mean = (float)(ce / (double)count);
            const double ber = (double)(cf[1] + cf[2]) / (double)(cf[0] + cf[1] + cf[2] + cf[3]);
            metrics->berlin = (float)ber;
            metrics->log_ber = (float)log(ber);
            metrics->reserved = 0.f;
            if (a.mon_acc != nullptr) {
                a.mon_acc[0] += metrics->ce_mean;
                a.mon_acc[1] += metrics->berlin;
                if (a.mon_noise != nullptr) a.mon_acc[3] += a.mon_noise[0];
            }
        }
    } else if (g == P + 1 && power_out != nullptr) {
        double s = 0.0;
        for (int i = lane; i < n_power; i += 64) s += power_partial[i];
        s = wave_sum(s);
        if (lane == 0) {
            power_out[0] = (float)(s / power_denom);
            if (a.mon_acc != nullptr) a.mon_acc[2] += power_out[0];
        }
    } else if (g == P + 2 && a.adam != nullptr && lane == 0) {
        dccn_adam_state* st = a.adam;
        const float lr = a.hp.lr0 * powf(a.hp.decay_rate, floorf(st->global_step / a.hp.decay_steps));
        st->alpha = lr * sqrtf(1.0f - st->beta2_power) / (1.0f - st->beta1_power);
        st->beta1_power = st->beta1_power * a.hp.beta1;
        st->beta2_power = st->beta2_power * a.hp.beta2;
        st->global_step = st->global_step + 1.0f;
    }
    if (g == P + 2 && a.zero_word != nullptr) {
        if (lane == 0) *a.zero_word = 0u;
        for (int i = lane; i < a.n_zero_flags; i += 64) a.zero_flags[(size_t)i * a.zero_stride] = 0u;
    }
}

static __global__ __launch_bounds__(256) void demod_tail_finalize_kernel(TailFinalizeArgs a) {
    demod_tail_finalize_body(a, blockIdx.x);
}

}  // namespace dccn
