// libdccn.so -- C ABI over the gfx950 kernels (see include/dccn.h for the contract and the
// reference call site each entry point replaces).
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "common.h"
#include "gemm_f32_mfma.h"
#include "cconv_fwd.h"
#include "norm_adam.h"
#include "tail.h"
#include "gemm16.h"
#include "gemm_kmajor.h"
#include "rx_bwd.h"
#include "equalizer.h"
#include "fewrow.h"
#include "eq_opt.h"
#include "eq_bottleneck.h"
#include "datagen.h"
#include "im2col.h"
#include "cconv_dx_narrow.h"
#include "cconv1d_bwd.h"
#include "classical.h"

namespace dccn {
thread_local int g_last_hip_error = 0;
thread_local ChainCtx tl_chain = {1, {{0, 0, 0, 0, 0, 0, 0, 0}}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
StepTraceState g_step_trace;
thread_local unsigned long long* tl_stamp = nullptr;

// bump allocator over a caller-provided workspace
struct Carver {
    char* base;
    size_t off, cap;
    Carver(void* p, size_t n) : base(static_cast<char*>(p)), off(0), cap(n) {}
    template <typename T>
    T* take(size_t count) {
        off = align_up(off, 256);
        T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;     // base == nullptr: size-only dry run
        off += count * sizeof(T);
        return r;
    }
    bool ok() const { return off <= cap && (base != nullptr || off == 0); }
};
static size_t carve_size(size_t off, size_t bytes) { return align_up(off, 256) + bytes; }

// Kernel-configuration knobs (dccn_set_tuning): which tile configuration the GEMM-shaped operators launch.
// 0 = the 32x32x2 family of gemm_f32_mfma.h, > 0 = a gemm16.h configuration (see the *_impl functions).
enum TuneKey : int {
    // (keys 15, 16, 22, 23, 26 -- the optimizer launch that also ran the next C-Conv forward, Adam in the dW epilogue, non-temporal
    // gradient loads, 160x64 / 128x64 dense + tail tiles, prefetch_fwd -- were built, measured without gain in rounds 3-5 and
    // removed in round 6 together with their kernels; dccn_set_tuning refuses them)
    TUNE_DENSE_FWD = 0,         // > 0: the dense forward runs with the demodulation tail in its epilogue (48x64 / 80x64 tiles)
    TUNE_DENSE_BWD = 1,         // grouped dX + dW
    TUNE_CCONV_FWD = 2,
    TUNE_CCONV_BWD_W = 3,
    TUNE_DENSE_BWD_SPLITS = 4,  // 0 = automatic
    TUNE_CCONV_BWD_SPLITS = 5,  // 0 = automatic
    TUNE_SMEM_MIN_KB = 6,       // minimum dynamic LDS per block of the gemm16 launches (caps resident blocks per CU)
    TUNE_WHOLE_K = 7,           // 1: short-k GEMMs (C-Conv forward / weight gradient at N=64) run their k range as one tile
    TUNE_SKINNY = 8,            // > 0: GEMMs with <= 96 output rows (the equaliser's 73-frame batch) use small gemm16 tiles
    TUNE_DENSE_BWD_BIG = 9,     // 1: large dense layers run dX and dW (128x128x32 tiles) as one grouped launch
    TUNE_DENSE_FWD_PLAIN = 10,  // 1: the un-fused dense forward (nbits >= 3, layer API) of small layers runs 48x64 gemm16 tiles
    TUNE_FUSED_BWD = 11,        // 1: small layers: the C-Conv weight gradient rides in the epilogue of the dense dX tiles (rx_bwd.h)
    TUNE_BWD_PRIO = 12,         // s_setprio level (0-3) of the dX blocks of the fused backward launch
    TUNE_TAIL_FUSE_HI = 13,     // 8-QAM / 16-QAM tail inside the dense forward launch: bit 0 lane-per-cell forms, bit 1 quad-lane training
    TUNE_DW_GRADED = 14,        // > 0: graded k ranges for the dense dW items of the fused backward launch (preset number)
    TUNE_SKINNY_GROUPED = 17,   // 1: few-row dense backward (<= 96 rows): dX (16x64 tiles) and the unsplit dW in one grid
    TUNE_NORM_ON_BWD = 18,      // 1 / 2: double-buffered pipelining: R0 of the next batch rides on the backward launch (leading / closing workgroups)
    TUNE_EQ_EPILOGUES = 19,     // equaliser step: tanh / tanh-gradient / gradient add in GEMM stores: 1 = the few-row GEMMs,
                                //    2 = also the 48x64 / 64x64 tiles of larger batches
    TUNE_EQ_REPLAN = 20,        // 1: equaliser step: one job-table optimizer launch, corr/eq C-Conv pair as grouped launches,
                                //    concat / split in GEMM stores, merged element-wise launches (eq_step.h)
    TUNE_FEWROW = 21,           // 1: few-row GEMMs (<= 96 rows, K = 640 / 896) on the one-latency 16x16 tiles of fewrow.h
    TUNE_ADAM_OVERLAP = 25,     // 1: large layers: the dense kernel's Adam update runs on a second stream next to the C-Conv weight-gradient launch
                                //    2: ... with non-temporal loads and stores (it must not displace the GEMM's operand panels)
    TUNE_EQ_RIDERS = 24,        // 1: equaliser step: the Adam updates of dense_3 / dense_4 ride behind the pilot bottleneck's backward launch
    TUNE_DENSE_RAGGED = 27,     // 1: large layers' fused dense + tail: a short last row tile (<= 32 rows) runs 32x64 blocks in the same grid
    TUNE_COUNT = 28
};
// The knobs are process-global DEFAULTS (relaxed atomics: one thread may turn them while another plans a launch); a call never
// sees them change under it: every step / operator entry opens a TuneScope, which copies the table once -- from the plan's own
// table when the caller captured one at plan creation (dccn_rx_buffers.tuning / dccn_eq_buffers.tuning, dccn_tuning_snapshot),
// from the globals otherwise -- and everything the call plans reads that copy.
thread_local int tl_whole_k = -1;
thread_local int tl_tune_depth = 0;
thread_local int tl_tune_vals[TUNE_COUNT];
struct TuneTable {
    std::atomic<int> v[TUNE_COUNT];
    int operator[](int k) const { return tl_tune_depth > 0 ? tl_tune_vals[k] : v[k].load(std::memory_order_relaxed); }
    int global(int k) const { return v[k].load(std::memory_order_relaxed); }
    void set(int k, int x) { v[k].store(x, std::memory_order_relaxed); }
};
// defaults = the fastest measured (A/B runs of tools/ab.py inside one process on one box, medians of 4-6 rounds of 300 steps;
// boxes of the pool differ by up to 9 % in absolute time, so only same-process comparisons decide):
//    2 = 7  round 5: the staged whole-k C-Conv forward of cconv_fwd.h (stages of 32 k consumed as they land): 7.15 -> 5.04 us in
//           situ, C2 step 76.36 -> 74.67 us, bit-identical (gpurun_out/r05a; 8 / 9 = other LDS-store slots: 74.59 / 74.76);
//   12 = 3  dX tiles at wave priority 3: -0.4 us per C2 step;
//   13 = 1  8-QAM training 86.7 -> 84.4 us with the tail in the dense launch; the quad-lane form of 16-QAM training is
//           built and parity-tested but slower than its own launch (104.8 vs 98.8 us: one wave per SIMD cannot hide the
//           transcendental / DPP latencies of 24 cells per quad), so bit 1 stays off;
//   14 = 14 graded dense-dW ranges {9,5,2,2,1}/19 of the batch (k-tiles of 64 frames; round 4: {8,6,3,2}, 80.2 -> 77.4 us per C2
//           step over three uniform ranges).  Round 5 re-scanned 24 presets on the lighter launch (folded dWeff partials):
//           steeper grading packs the grid's tail better although a fifth slab is written and summed -- backward launch
//           33.1 -> 31.0 us in situ, optimizer 6.9 -> 7.2, step 74.4 -> 73.5 us ({8,5,3,2,1} 73.6, {9,5,3,2} 73.8, {10,5,3,1}
//           73.7, six ranges 74.3-74.8, three ranges 76.3-77.4, two 80.8: gpurun_out/r05a, r05c-r05e);
//   17 = 1  few-row dense backward as one grid: equaliser step at 73 frames 0.401 -> 0.375 ms (five launches fewer);
//   19 = 2  element-wise stages in GEMM stores: 73 frames 0.2688 -> 0.2663 ms (few-row tiles); 1170 frames 0.5276 -> 0.5247 ms
//           (the stage costs the GEMM 8-10 us where the stand-alone launch cost 5: a small net gain);
//   20 = 1  equaliser re-plan: 73 frames 0.319 -> 0.263 ms, 1170 frames 0.570 -> 0.509 ms (tools/eqbench.py --ab 20=0,1,2);
//   18 = 0  R0 of the next batch on the backward launch (second x_norm buffer): 78.8 vs 78.6 us -- the optimizer launch it
//           came from is bounded by the 133-term C-Conv fold, not by R0; built, bitwise-tested, off.
//   21 = 1  few-row GEMMs on the one-latency tiles of fewrow.h: equaliser step at 73 frames 0.243 -> 0.180 ms (tools/eqbench.py --ab 21=0,1);
//   24 = 1  equaliser step: Adam updates of dense_3 / dense_4 and the smoothing kernel's fold as riders of the bottleneck backward
//           launch: 73 frames 0.1749 -> 0.1707 ms (tools/eqbench.py --ab 24=0,1);
//   25 = 2  large layers: the dense kernel's Adam update (3.2 GB at N = 1024) on the library's low-priority second stream next to the
//           C-Conv weight-gradient launch: C4 step 4998 -> 4804 us, with non-temporal loads / stores 4775 us (tools/ab.py --config c4).
static TuneTable g_tune = {{{9}, {7}, {7}, {7}, {0}, {0}, {0}, {1}, {1}, {1}, {1}, {1}, {3}, {1}, {14}, {0}, {0}, {1}, {0}, {2}, {1}, {1}, {0}, {0}, {1}, {2}, {0}, {1}}};

struct TuneScope {
    explicit TuneScope(const int* plan = nullptr) {
        if (tl_tune_depth++ == 0) {
            for (int k = 0; k < TUNE_COUNT; ++k) tl_tune_vals[k] = plan ? plan[k] : g_tune.global(k);
            tl_whole_k = tl_tune_vals[TUNE_WHOLE_K];
        }
    }
    ~TuneScope() {
        if (--tl_tune_depth == 0) tl_whole_k = -1;
    }
    TuneScope(const TuneScope&) = delete;
    TuneScope& operator=(const TuneScope&) = delete;
};

// few output rows, long k: 64x64 tiles leave most CUs without a block (73x896 = 28 tiles); 16- or 32-row tiles give 2-5x
// the blocks, and loads two k-tiles ahead cover the latency that the short MFMA phases cannot
template <int KA, int KB, int TAG, int ACT = 1>
static int skinny_launch(int variant, const GemmParams& p, hipStream_t s) {
    if constexpr (ACT != 1) return launch_gemm16<KA, KB, 1, 4, 1, 1, 64, 1, 0, TAG, 2, ACT>(p, 1, s);   // 16x64 + element-wise stage
    switch (variant) {
        case 1: return launch_gemm16<KA, KB, 1, 4, 1, 1, 64, 1, 0, TAG, 2>(p, 1, s);       // 16x64
        case 2: return launch_gemm16<KA, KB, 1, 4, 2, 1, 64, 1, 0, TAG, 2>(p, 1, s);       // 32x64
        case 3: return launch_gemm16<KA, KB, 2, 2, 1, 1, 64, 1, 0, TAG, 2>(p, 1, s);       // 32x32
        case 4: return launch_gemm16<KA, KB, 1, 4, 1, 1, 64, 2, 0, TAG, 2>(p, 1, s);       // 16x64, 8 waves
        default: return DCCN_ERR_INVALID_ARG;
    }
}
static bool skinny_ok(const GemmParams& p) {
    return g_tune[TUNE_SKINNY] > 0 && p.M <= 96 && p.N >= 256 && p.K >= 256 && p.vecA && p.vecB && (p.K % 4 == 0) &&
           (p.N % 4 == 0);
}
constexpr int kVariantKmajor = 7;   // TUNE_DENSE_BWD / TUNE_CCONV_BWD_W value: weight gradient in the k-major form of gemm_kmajor.h
static size_t tune_smem_min() { return (size_t)g_tune[TUNE_SMEM_MIN_KB] * 1024; }

// ---------------------------------------------------------------------------------------
// R0
// ---------------------------------------------------------------------------------------
static int norm_grid_x(int cols) { return ceil_div(ceil_div(cols, 4), 64); }
static int norm_grid_y(int batch) { return ceil_div(batch, kNormRowsPerBlock); }

static int norm_fused_blocks(int cols) { return ceil_div(ceil_div(cols, 4 * kNormFusedCG), 8) * 8; }
static size_t norm_power_slots(int batch, int cols) {
    const size_t a = (size_t)norm_grid_x(cols) * norm_grid_y(batch), b = (size_t)norm_fused_blocks(cols);
    return a > b ? a : b;
}
static size_t norm_ws_bytes(int batch, int cols) {
    size_t o = 0;
    o = carve_size(o, (size_t)kNormRowChunks * cols * 2 * sizeof(double));
    o = carve_size(o, 2 * norm_power_slots(batch, cols) * sizeof(double));      // two slots: dccn_rx_buffers.norm_slot
    return align_up(o, 256);
}

static bool norm_fused_ok(const float* x, const float* y, int batch, int cols) {
    return (cols % 4 == 0) && batch <= 128 * kNormFusedRPT && aligned16(x) && aligned16(y);
}

struct PowerPartials {      // where normalise left the R8 partial sums (finished by a later kernel)
    const double* partial;
    int n;
    double denom;
};

// want_power: also emit the per-block partial sums of the clipped power (R8); adam != nullptr: the
// optimizer bookkeeping of the fused training step rides on the first kernel
// where norm_impl leaves the R8 partial sums for a [batch, cols] input in workspace `ws` (no launch)
static void norm_power_partials(int batch, int cols, void* ws, size_t ws_bytes, const float* x, const float* y,
                                PowerPartials* pp, int slot = 0) {
    Carver c(ws, ws_bytes);
    c.take<double>((size_t)kNormRowChunks * cols * 2);
    pp->partial = c.take<double>(2 * norm_power_slots(batch, cols)) + (slot ? norm_power_slots(batch, cols) : 0);
    pp->n = norm_fused_ok(x, y, batch, cols) ? norm_fused_blocks(cols) : norm_grid_x(cols) * norm_grid_y(batch);
    pp->denom = (double)batch * (double)(cols / 2);
}

static int norm_impl(const float* x, float* y, float* mean, float* var, bool want_power, PowerPartials* pp, int batch,
                     int cols, float eps, float peak, dccn_adam_state* adam, dccn_adam_hparams hp, void* ws,
                     size_t ws_bytes, hipStream_t s, int slot = 0) {
    if (!x || !y || batch <= 0 || cols <= 0 || (want_power && (cols & 1))) return DCCN_ERR_INVALID_ARG;
    if (ws_bytes < norm_ws_bytes(batch, cols) || !ws) return DCCN_ERR_WORKSPACE;
    Carver c(ws, ws_bytes);
    double* partial = c.take<double>((size_t)kNormRowChunks * cols * 2);
    const int gx = norm_grid_x(cols), gy = norm_grid_y(batch);
    double* pw = c.take<double>(2 * norm_power_slots(batch, cols)) + (slot ? norm_power_slots(batch, cols) : 0);
    if (norm_fused_ok(x, y, batch, cols)) {
        // the whole batch of a column strip fits in one block's registers: single pass, single launch
        const int blocks = norm_fused_blocks(cols);
        DCCN_LAUNCH_CHAINS_Z((norm_fused_kernel<kNormFusedCG, kNormFusedRPT>), dim3(blocks), dim3(128 * kNormFusedCG), 0, s,
                             x, y, batch, cols, eps, peak, want_power ? pw : nullptr, mean, var, adam, hp);
        DCCN_LAUNCH_CHECK();
        if (pp) {
            pp->partial = pw;
            pp->n = blocks;
            pp->denom = (double)batch * (double)(cols / 2);
        }
        return DCCN_OK;
    }
    DCCN_NO_CHAINS();
    hipLaunchKernelGGL(moments_kernel, dim3(gx, kNormRowChunks), dim3(64, 4), 0, s, x, batch, cols, partial, adam, hp);
    DCCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(normalise_kernel, dim3(gx, gy), dim3(64, 4), 0, s, x, y, partial, batch, cols, eps, peak,
                       want_power ? pw : nullptr, mean, var);
    DCCN_LAUNCH_CHECK();
    if (pp) {
        pp->partial = pw;
        pp->n = gx * gy;
        pp->denom = (double)batch * (double)(cols / 2);
    }
    return DCCN_OK;
}

// ---------------------------------------------------------------------------------------
// GEMM-shaped ops
// ---------------------------------------------------------------------------------------
static GemmParams gp_zero() {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.stamp = tl_stamp;         // non-null only while a StepTraceScope names the launch being built (common.h)
    return p;
}
static int round_k(int K) { return ceil_div(K, 64) * 64; }
// the fast GEMM loaders use 32-bit byte offsets from a uniform base: operand must be < 2 GiB
static bool small_enough(long long rows, long long ld) { return rows * ld * 4 < (1LL << 31); }

// ldx > 0: x is a column window of a wider row-major matrix (row stride ldx)
// act = 2: y = tanh(x.w + bias) when the launch plan has the stage (few-row 16x64 tiles); *act_done tells
// act = 5 (+ aux = the received cells, out2 / out3): the output is a channel estimate; eq and corr of model.py:431-438
// leave the same launch
static int dense_fwd_impl(const float* x, const float* w, const float* bias, float* y, int M, int K, int N,
                          hipStream_t s, int ldx = 0, int act = 1, bool* act_done = nullptr, const float* aux = nullptr,
                          float* out2 = nullptr, float* out3 = nullptr) {
    if (act_done) *act_done = false;
    if (!x || !w || !y || M <= 0 || K <= 0 || N <= 0) return DCCN_ERR_INVALID_ARG;
    const int lda = ldx > 0 ? ldx : K;
    GemmParams p = gp_zero();
    p.A = x; p.B = w; p.C = y; p.bias = bias;
    p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldb = N; p.ldc = N;
    p.klen = round_k(K);
    p.vecA = (K % 4 == 0) && (lda % 4 == 0) && aligned16(x) && small_enough(M, lda);     // KCONTIG: k extent K
    p.vecB = (N % 4 == 0) && aligned16(w) && small_enough(K, N);     // ICONTIG: ld = N, i extent N
    const bool eq_stage = act == 5 && act_done && aux && out2 && out3 && g_tune[TUNE_EQ_EPILOGUES] && (N % 2 == 0) &&
                          aligned16(aux) && aligned16(out2) && aligned16(out3);
    if (eq_stage) { p.aux = aux; p.out2 = out2; p.out3 = out3; }
    if (g_tune[TUNE_FEWROW] && g_tune[TUNE_SKINNY] > 0 && fewrow_ng(p) && aligned16(y)) {
        // few rows, K = 640 / 896: every operand of a 16x16 tile requested at once, no LDS staging (fewrow.h)
        if (eq_stage) {
            *act_done = true;
            return launch_fewrow<OP_ICONTIG, 5, TAG_DENSE_FWD>(p, s);
        }
        if (act == 2 && act_done && g_tune[TUNE_EQ_EPILOGUES]) {
            *act_done = true;
            return launch_fewrow<OP_ICONTIG, 2, TAG_DENSE_FWD>(p, s);
        }
        return launch_fewrow<OP_ICONTIG, 1, TAG_DENSE_FWD>(p, s);
    }
    if (skinny_ok(p)) {
        if (eq_stage) {
            *act_done = true;
            return skinny_launch<OP_KCONTIG, OP_ICONTIG, TAG_DENSE_FWD, 5>(1, p, s);
        }
        if (act == 2 && act_done && g_tune[TUNE_EQ_EPILOGUES]) {
            *act_done = true;
            return skinny_launch<OP_KCONTIG, OP_ICONTIG, TAG_DENSE_FWD, 2>(1, p, s);
        }
        return skinny_launch<OP_KCONTIG, OP_ICONTIG, TAG_DENSE_FWD>(g_tune[TUNE_SKINNY], p, s);
    }
    // small layers (64x64 tiles would leave CUs without a block: 1170x640 = 190 tiles): the 48x64 tiles of the fused
    // kernel, loads two k-tiles ahead
    if (g_tune[TUNE_DENSE_FWD_PLAIN] && p.vecA && p.vecB && (K % 4 == 0) && (N % 4 == 0) && K >= 128 &&
        (long long)ceil_div(M, 128) * ceil_div(N, 128) < 2 * kCUs) {
        if (eq_stage && g_tune[TUNE_EQ_EPILOGUES] >= 2) {
            *act_done = true;
            return launch_gemm16<OP_KCONTIG, OP_ICONTIG, 1, 4, 3, 1, 64, 1, 0, TAG_DENSE_FWD, 2, 5>(p, 1, s);
        }
        if (act == 2 && act_done && g_tune[TUNE_EQ_EPILOGUES] >= 2) {
            *act_done = true;                       // tanh in the store of the same tiles
            return launch_gemm16<OP_KCONTIG, OP_ICONTIG, 1, 4, 3, 1, 64, 1, 0, TAG_DENSE_FWD, 2, 2>(p, 1, s);
        }
        return launch_gemm16<OP_KCONTIG, OP_ICONTIG, 1, 4, 3, 1, 64, 1, 0, TAG_DENSE_FWD, 2>(p, 1, s);
    }
    return launch_gemm<OP_KCONTIG, OP_ICONTIG, 0, TAG_DENSE_FWD>(p, 1, s);
}

static GemmParams dense_bwd_x_params(const float* dy, const float* w, float* dx, int M, int K, int N) {
    GemmParams p = gp_zero();                 // dx[M,K] = dy[M,N] . w[K,N]^T
    p.A = dy; p.B = w; p.C = dx;
    p.M = M; p.N = K; p.K = N;
    p.lda = N; p.ldb = N; p.ldc = K;
    p.klen = round_k(N);
    p.vecA = (N % 4 == 0) && aligned16(dy) && small_enough(M, N);
    p.vecB = (N % 4 == 0) && aligned16(w) && small_enough(K, N);
    return p;
}

static int dense_bwd_x_impl(const float* dy, const float* w, float* dx, int M, int K, int N, hipStream_t s) {
    if (!dy || !w || !dx || M <= 0 || K <= 0 || N <= 0) return DCCN_ERR_INVALID_ARG;
    const GemmParams p = dense_bwd_x_params(dy, w, dx, M, K, N);
    if (g_tune[TUNE_FEWROW] && g_tune[TUNE_SKINNY] > 0 && fewrow_ng(p)) return launch_fewrow<OP_KCONTIG, 1, TAG_DENSE_BWD_X>(p, s);
    if (skinny_ok(p)) return skinny_launch<OP_KCONTIG, OP_KCONTIG, TAG_DENSE_BWD_X>(g_tune[TUNE_SKINNY], p, s);
    return launch_gemm<OP_KCONTIG, OP_KCONTIG, 0, TAG_DENSE_BWD_X>(p, 1, s);
}

// slab capacity for the weight-gradient split-K plans: small outputs may be cut into up to 8 k ranges (gemm16 paths)
static int max_splits16(int Mo, int No) {
    const long long tiles = (long long)ceil_div(Mo, 64) * ceil_div(No, 64);
    long long m = (1024 + tiles - 1) / tiles;
    return (int)(m < 1 ? 1 : (m > 8 ? 8 : m));
}
static size_t splitk_ws_bytes(int Mo, int No, int Kr) {
    const SplitPlan sp = plan_splitk(Mo, No, Kr);
    const int ms = max_splits16(Mo, No);
    const int n = sp.splits > ms ? sp.splits : ms;
    size_t o = 0;
    o = carve_size(o, (size_t)n * Mo * No * sizeof(float));
    o = carve_size(o, (size_t)n * No * sizeof(float));
    return align_up(o, 256);
}

// split plan of the dense weight gradient dw[K,N] = x[M,K]^T . dy[M,N] (one rule for the stand-alone operator and the
// grouped launch: the composed and the fused step must sum in the same order)
// range_rows: preferred k-range length of the k-major form.  256 (4 k-tiles) next to plain dX tiles: the blocks are cheap
// to start and pack the grid's tail better than the 320-row ranges of the generic plan (C2: 5 ranges instead of 4, -1 us
// per step even with one more slab to sum).  448 (7 k-tiles) in the fused backward launch of rx_bwd.h, whose dX tiles carry
// the C-Conv contraction and run 26 us: fewer, longer dW items amortise their load/store phases and leave two slabs less
// for the optimizer launch (C2: 3 ranges, -2.0 us per step; 2, 4 and 6 ranges measured +1.2 ... +1.5 us over 3).
static SplitPlan dense_dw_plan(int M, int K, int N, int range_rows = 256) {
    SplitPlan sp = plan_splitk(K, N, M);
    const int cap = max_splits16(K, N);
    if (g_tune[TUNE_DENSE_BWD_SPLITS] > 0 && sp.splits > 1) {
        sp = plan_splitk_n(M, g_tune[TUNE_DENSE_BWD_SPLITS] < cap ? g_tune[TUNE_DENSE_BWD_SPLITS] : cap, 64);
    } else if (g_tune[TUNE_DENSE_BWD] == kVariantKmajor && sp.splits > 1 && (range_rows > 256 || sp.klen > range_rows)) {
        const int want = ceil_div(M, range_rows);
        const SplitPlan alt = plan_splitk_n(M, want < cap ? want : cap, 64);
        if (range_rows > 256 || alt.splits > sp.splits) sp = alt;
    }
    return sp;
}
constexpr int kFusedBwdRangeRows = 448;
// graded k ranges (in 64-row k-tiles, as shares of the total): the items are dispatched range by range, so the last ones
// handed out are the short ones
static int graded_ranges(int preset, int M, int off[9]) {
    static const int shares[][8] = {{0}, {8, 6, 3, 2, 0}, {9, 6, 4, 0}, {7, 5, 4, 3, 0}, {10, 9, 0}, {8, 7, 4, 0}, {6, 5, 4, 3, 1, 0},
                                    {9, 7, 3, 0}, {9, 6, 3, 1, 0}, {8, 5, 3, 2, 1, 0}, {10, 5, 3, 1, 0}, {7, 6, 4, 2, 0},
                                    {8, 6, 4, 1, 0}, {9, 5, 3, 2, 0}, {9, 5, 2, 2, 1, 0}, {8, 4, 3, 2, 2, 0}, {10, 4, 2, 2, 1, 0},
                                    {7, 5, 3, 2, 2, 0}, {9, 4, 3, 2, 1, 0}, {8, 5, 3, 1, 2, 0}, {7, 5, 4, 2, 1, 0},
                                    {8, 4, 3, 2, 1, 1, 0}, {9, 4, 2, 2, 1, 1, 0}, {9, 5, 2, 1, 1, 1, 0}, {10, 4, 2, 1, 1, 1, 0}};
    if (preset < 1 || preset > 24) return 0;
    const int nt = ceil_div(M, 64);
    int tot = 0, n = 0;
    while (shares[preset][n]) tot += shares[preset][n++];
    int used = 0, acc = 0, cnt = 0;
    off[0] = 0;
    for (int i = 0; i < n; ++i) {
        acc += shares[preset][i];
        int upto = (int)((long long)acc * nt / tot);
        if (i == n - 1) upto = nt;
        if (upto <= used) continue;
        used = upto;
        off[++cnt] = upto * 64 < M ? upto * 64 : M;
    }
    return cnt;
}

// defer != nullptr: leave the split-K slabs un-reduced (the fused Adam kernel sums them) and report them
struct DeferredSlabs {
    const float* dw_slabs;
    const float* db_slabs;
    int splits;
};
static int dense_bwd_w_impl(const float* x, const float* dy, float* dw, float* dbias, int M, int K, int N, void* ws,
                            size_t ws_bytes, hipStream_t s, DeferredSlabs* defer = nullptr, int ldx = 0) {
    if (!x || !dy || !dw || M <= 0 || K <= 0 || N <= 0) return DCCN_ERR_INVALID_ARG;
    if (!ws || ws_bytes < splitk_ws_bytes(K, N, M)) return DCCN_ERR_WORKSPACE;
    const SplitPlan sp = dense_dw_plan(M, K, N);
    Carver c(ws, ws_bytes);
    float* slabs = c.take<float>((size_t)sp.splits * K * N);
    float* cs = c.take<float>((size_t)sp.splits * N);
    GemmParams p = gp_zero();                 // dw[K,N] = x[M,K]^T . dy[M,N]
    p.A = x; p.B = dy;
    p.M = K; p.N = N; p.K = M;
    p.lda = ldx > 0 ? ldx : K; p.ldb = N; p.ldc = N;
    p.klen = sp.klen;
    p.slab = (long long)K * N;
    p.vecA = (K % 4 == 0) && (p.lda % 4 == 0) && aligned16(x) && small_enough(M, p.lda);
    p.vecB = (N % 4 == 0) && aligned16(dy) && small_enough(M, N);
    if (defer) { defer->dw_slabs = nullptr; defer->db_slabs = nullptr; defer->splits = 1; }
    if (sp.splits == 1) {
        p.C = dw;
        p.colsum = dbias;
        return launch_gemm<OP_ICONTIG, OP_ICONTIG, 1, TAG_DENSE_BWD_W>(p, 1, s);
    }
    p.C = slabs;
    p.colsum = dbias ? cs : nullptr;
    const long long big = (long long)ceil_div(p.M, 128) * ceil_div(p.N, 128) * sp.splits;
    if (g_tune[TUNE_DENSE_BWD] == kVariantKmajor && kmajor_ok(p) && big < 2 * kCUs)
        DCCN_TRY((launch_kmajor<1, TAG_DENSE_BWD_W>(p, sp.splits, s)));
    else
        DCCN_TRY((launch_gemm<OP_ICONTIG, OP_ICONTIG, 1, TAG_DENSE_BWD_W>(p, sp.splits, s)));
    if (defer) {
        defer->dw_slabs = slabs;
        defer->db_slabs = dbias ? cs : nullptr;
        defer->splits = sp.splits;
        return DCCN_OK;
    }
    const long long n = (long long)K * N;
    if (dbias) DCCN_TRY(launch_splitk_reduce2(slabs, sp.splits, n, dw, n, cs, (long long)N, dbias, (long long)N, s));
    else DCCN_TRY(launch_splitk_reduce(slabs, sp.splits, n, dw, n, s));
    return DCCN_OK;
}

// dense backward as ONE grouped launch: dx = dy.w^T together with the split-K slabs of dw = x^T.dy
// (left un-reduced for the fused Adam kernel).  Falls back to two launches when the grouped
// configuration does not apply (128x128 tiles, unaligned operands, single split).
static int dense_bwd16_launch(int variant, const GemmParams& px, const GemmParams& pw, int splits, hipStream_t s) {
    const size_t sm = tune_smem_min();
    switch (variant) {
        case 1: return launch_dense_bwd16<2, 2, 2, 2, 32, 2, 2, 2, 2>(px, pw, splits, s, sm);     // dX 64x64, dW 64x64
        case 2: return launch_dense_bwd16<1, 4, 3, 1, 32, 2, 2, 2, 2>(px, pw, splits, s, sm);     // dX 48x64
        case 3: return launch_dense_bwd16<2, 2, 2, 2, 64, 2, 2, 2, 2>(px, pw, splits, s, sm);     // 64-deep k-tiles
        case 4: return launch_dense_bwd16<2, 2, 3, 2, 32, 2, 2, 2, 2>(px, pw, splits, s, sm);     // dX 96x64
        case 5: return launch_dense_bwd16<2, 2, 2, 4, 32, 2, 2, 2, 4>(px, pw, splits, s, sm);     // 64x128 both
        case 6: return launch_dense_bwd16<2, 2, 2, 2, 32, 2, 2, 2, 4>(px, pw, splits, s, sm);     // dX 64x64, dW 64x128
        default: return DCCN_ERR_INVALID_ARG;
    }
}
static int dense_bwd16_bk(int variant) { return variant == 3 ? 64 : 32; }
static void dense_bwd16_tiles(int variant, int& xm, int& xn, int& wm, int& wn) {
    xm = 64; xn = 64; wm = 64; wn = 64;
    if (variant == 2) xm = 48;
    if (variant == 4) xm = 96;
    if (variant == 5) { xn = 128; wn = 128; }
    if (variant == 6) wn = 128;
}

// dense backward as ONE grouped launch: dx = dy.w^T together with the split-K slabs of dw = x^T.dy
// (left un-reduced for the fused Adam kernel).  Falls back to two launches when the grouped
// configuration does not apply (128x128 tiles, unaligned operands, single split).
static int dense_bwd_grouped_impl(const float* x, const float* dy, const float* w, float* dx, float* dw, float* dbias,
                                  int M, int K, int N, void* ws, size_t ws_bytes, hipStream_t s, DeferredSlabs* defer,
                                  int actx = 1, const float* aux = nullptr, bool* act_done = nullptr,
                                  float* split_dst = nullptr, long long split_pairs_gc = 0, bool* split_done = nullptr) {
    if (act_done) *act_done = false;
    if (split_done) *split_done = false;
    if (!x || !dy || !w || !dx || !dw || !defer || M <= 0 || K <= 0 || N <= 0) return DCCN_ERR_INVALID_ARG;
    if (!ws || ws_bytes < splitk_ws_bytes(K, N, M)) return DCCN_ERR_WORKSPACE;
    GemmParams px = dense_bwd_x_params(dy, w, dx, M, K, N);
    GemmParams pw = gp_zero();                // dw[K,N] = x[M,K]^T . dy[M,N]
    pw.A = x; pw.B = dy;
    pw.M = K; pw.N = N; pw.K = M;
    pw.lda = K; pw.ldb = N; pw.ldc = N;
    pw.slab = (long long)K * N;
    pw.vecA = (K % 4 == 0) && aligned16(x) && small_enough(M, K);
    pw.vecB = (N % 4 == 0) && aligned16(dy) && small_enough(M, N);
    const bool vec = px.vecA && px.vecB && pw.vecA && pw.vecB;
    // few rows (the equaliser's 73-frame batch): dX on 16x64 tiles and the unsplit dW (k = the few rows: one or two
    // k-tiles) in ONE grid -- round 2 ran them as two launches of 6-10 us each, almost all of it launch ramp and drain
    if (g_tune[TUNE_SKINNY] > 0 && g_tune[TUNE_SKINNY_GROUPED] && vec && M <= 96 && (K % 4 == 0) && (N % 4 == 0)) {
        pw.klen = round_k(M);
        pw.C = dw; pw.colsum = dbias; pw.slab = 0;
        // the dX tiles may carry an element-wise stage of the caller's graph: 3 = times (1 - aux^2) (tanh gradient),
        // 4 = plus aux (gradient accumulation): two 5 us launches of the equaliser step less
        const bool few = g_tune[TUNE_FEWROW] && fewrow_ng(px) != 0;        // dX on the one-latency 16x16 tiles of fewrow.h
        if (actx != 1 && act_done && aux && g_tune[TUNE_EQ_EPILOGUES]) {
            px.aux = aux;
            *act_done = true;
            if (actx == 3) DCCN_TRY(few ? launch_dense_bwd_fewrow<3>(px, pw, s)
                                        : (launch_dense_bwd16<1, 4, 1, 1, 64, 2, 2, 2, 2, 3>(px, pw, 1, s, tune_smem_min())));
            else if (actx == 4) DCCN_TRY(few ? launch_dense_bwd_fewrow<4>(px, pw, s)
                                             : (launch_dense_bwd16<1, 4, 1, 1, 64, 2, 2, 2, 2, 4>(px, pw, 1, s, tune_smem_min())));
            else return DCCN_ERR_INVALID_ARG;
        } else
        DCCN_TRY(few ? launch_dense_bwd_fewrow<1>(px, pw, s)
                     : (launch_dense_bwd16<1, 4, 1, 1, 64, 2, 2, 2, 2>(px, pw, 1, s, tune_smem_min())));
        defer->dw_slabs = nullptr; defer->db_slabs = nullptr; defer->splits = 1;
        return DCCN_OK;
    }
    const int variant = g_tune[TUNE_DENSE_BWD] == kVariantKmajor ? 0 : g_tune[TUNE_DENSE_BWD];
    const long long big = (long long)ceil_div(M, 128) * ceil_div(K, 128);
    if (variant > 0 && vec && big < 2 * kCUs) {
        int xm, xn, wm, wn;
        dense_bwd16_tiles(variant, xm, xn, wm, wn);
        const int nx = ceil_div(px.M, xm) * ceil_div(px.N, xn), tw = ceil_div(pw.M, wm) * ceil_div(pw.N, wn);
        int want = g_tune[TUNE_DENSE_BWD_SPLITS];
        if (want <= 0) want = (3 * kCUs - nx + tw / 2) / tw;            // about three resident blocks per CU in all
        const int cap = max_splits16(K, N);
        if (want > cap) want = cap;
        const SplitPlan sp = plan_splitk_n(M, want, dense_bwd16_bk(variant));
        Carver c(ws, ws_bytes);
        float* slabs = c.take<float>((size_t)sp.splits * K * N);
        float* cs = c.take<float>((size_t)sp.splits * N);
        pw.klen = sp.klen;
        if (sp.splits == 1) {
            pw.C = dw; pw.colsum = dbias;
        } else {
            pw.C = slabs; pw.colsum = dbias ? cs : nullptr;
        }
        DCCN_TRY(dense_bwd16_launch(variant, px, pw, sp.splits, s));
        defer->dw_slabs = sp.splits > 1 ? slabs : nullptr;
        defer->db_slabs = (sp.splits > 1 && dbias) ? cs : nullptr;
        defer->splits = sp.splits;
        return DCCN_OK;
    }
    const SplitPlan sp = dense_dw_plan(M, K, N);
    Carver c(ws, ws_bytes);
    float* slabs = c.take<float>((size_t)sp.splits * K * N);
    float* cs = c.take<float>((size_t)sp.splits * N);
    pw.C = slabs; pw.colsum = dbias ? cs : nullptr;
    pw.klen = sp.klen;
    if (vec && g_tune[TUNE_DENSE_BWD_BIG] && grouped_big_ok(px, pw, sp.splits)) {
        pw.ldc = N;
        if (sp.splits == 1) { pw.C = dw; pw.colsum = dbias; pw.slab = 0; }
        DCCN_TRY((launch_dense_bwd_grouped<true, 128, 128, 32>(px, pw, sp.splits, s)));
        defer->dw_slabs = sp.splits > 1 ? slabs : nullptr;
        defer->db_slabs = (sp.splits > 1 && dbias) ? cs : nullptr;
        defer->splits = sp.splits;
        return DCCN_OK;
    }
    if (sp.splits < 2 || !vec || !grouped_ok(px, pw, sp.splits)) {
        DCCN_TRY(dense_bwd_w_impl(x, dy, dw, dbias, M, K, N, ws, ws_bytes, s, defer));
        return dense_bwd_x_impl(dy, w, dx, M, K, N, s);
    }
    pw.ldc = N;
    if (g_tune[TUNE_DENSE_BWD] == kVariantKmajor && kmajor_ok(pw)) {
        if (split_done && split_dst && split_pairs_gc != 0 && (K % 4 == 0)) {
            // dx is the gradient of a concat of two IQ-pair streams: the stores write the two streams' own buffers
            // instead of dx (split_dst = stream 0, [M, K/2]; stream 1 split_pairs_gc elements further)
            px.C = split_dst; px.ldc = K / 2; px.gC = split_pairs_gc;
            *split_done = true;
            DCCN_TRY((launch_dense_bwd_grouped_km<64, CMAP_SPLIT_PAIRS>(px, pw, sp.splits, s)));
        } else if (actx != 1 && act_done && aux && g_tune[TUNE_EQ_EPILOGUES] >= 2) {
            // element-wise stage of the caller's graph on the dX stores (as in the few-row grid above)
            px.aux = aux;
            *act_done = true;
            if (actx == 3) DCCN_TRY((launch_dense_bwd_grouped_km<64, CMAP_TANHGRAD>(px, pw, sp.splits, s)));
            else if (actx == 4) DCCN_TRY((launch_dense_bwd_grouped_km<64, CMAP_ADD>(px, pw, sp.splits, s)));
            else return DCCN_ERR_INVALID_ARG;
        } else {
            DCCN_TRY(launch_dense_bwd_grouped_km<64>(px, pw, sp.splits, s));
        }
    } else DCCN_TRY(launch_dense_bwd_grouped<true>(px, pw, sp.splits, s));
    defer->dw_slabs = slabs;
    defer->db_slabs = dbias ? cs : nullptr;
    defer->splits = sp.splits;
    return DCCN_OK;
}

static int cconv_fwd_impl(const float* x, const float* w, const float* bias, float* out, int rows, int kin, int F,
                          hipStream_t s, int ldx = 0) {
    if (!x || !w || !out || rows <= 0 || kin <= 0 || F <= 0) return DCCN_ERR_INVALID_ARG;
    GemmParams p = gp_zero();                 // out[rows,2F] = x[rows,2kin] . Weff[2kin,2F]
    p.A = x; p.B = w; p.C = out; p.bias = bias; p.cbias = 1;
    p.M = rows; p.N = 2 * F; p.K = 2 * kin;
    p.lda = ldx > 0 ? ldx : 2 * kin; p.ldb = 2 * F; p.ldc = 2 * F;
    p.klen = round_k(2 * kin);
    p.cF = F;
    p.vecA = (kin % 2 == 0) && (p.lda % 4 == 0) && aligned16(x) && small_enough(rows, (long long)p.lda);
    p.vecB = (F % 2 == 0) && aligned16(w) && small_enough(kin, 2LL * F);      // float2 loads of [Wa|Wb] rows
    const int variant = g_tune[TUNE_CCONV_FWD];
    const long long big = (long long)ceil_div(p.M, 128) * ceil_div(p.N, 128);
    // 7-11 (default 7): the staged whole-k tile of cconv_fwd.h (K = 160 / 128, i.e. N = 64 with / without the cyclic
    // prefix): bit-identical to the whole-k tile of gemm_f32_mfma.h it replaces; 8 / 9 = other LDS-store slots
    // 10 / 11: 32 x 128 tiles (every x row read by one block)
    if (variant >= 7 && variant <= 11 && big < 2 * kCUs && cconv_fwd_staged_ok(p)) {
        if (variant == 7) return launch_cconv_fwd_staged<8>(p, s);
        if (variant == 8) return launch_cconv_fwd_staged<4>(p, s);
        if (variant == 9) return launch_cconv_fwd_staged<10>(p, s);
        if (variant == 10) return launch_cconv_fwd_staged<8, 32, 128>(p, s);
        return launch_cconv_fwd_staged<4, 32, 128>(p, s);
    }
    if (variant >= 7) return launch_gemm<OP_KCONTIG, OP_CCONV_W, 0, TAG_CCONV_FWD>(p, 1, s);
    if (variant > 0 && p.vecA && p.vecB && big < 2 * kCUs && kin % 2 == 0) {
        const size_t sm = tune_smem_min();
        switch (variant) {
            case 1: return launch_gemm16<OP_KCONTIG, OP_CCONV_W, 2, 2, 1, 4, 32, 1, 0, TAG_CCONV_FWD>(p, 1, s, sm);   // 32x128
            case 2: return launch_gemm16<OP_KCONTIG, OP_CCONV_W, 1, 4, 2, 2, 32, 1, 0, TAG_CCONV_FWD>(p, 1, s, sm);   // 32x128, wave 32x32
            case 3: return launch_gemm16<OP_KCONTIG, OP_CCONV_W, 1, 4, 1, 2, 32, 1, 0, TAG_CCONV_FWD>(p, 1, s, sm);   // 16x128
            case 4: return launch_gemm16<OP_KCONTIG, OP_CCONV_W, 2, 2, 2, 2, 32, 1, 0, TAG_CCONV_FWD>(p, 1, s, sm);   // 64x64
            case 5: return launch_gemm16<OP_KCONTIG, OP_CCONV_W, 1, 4, 1, 2, 32, 2, 0, TAG_CCONV_FWD>(p, 1, s, sm);   // 16x128, 8 waves
            case 6: return launch_gemm16<OP_KCONTIG, OP_CCONV_W, 2, 2, 1, 2, 32, 1, 0, TAG_CCONV_FWD>(p, 1, s, sm);   // 32x64
            default: return DCCN_ERR_INVALID_ARG;
        }
    }
    return launch_gemm<OP_KCONTIG, OP_CCONV_W, 0, TAG_CCONV_FWD>(p, 1, s);
}

static int cconv_bwd_x_impl(const float* dout, const float* w, float* dx, int rows, int kin, int F, hipStream_t s,
                            int ldc = 0) {
    if (!dout || !w || !dx || rows <= 0 || kin <= 0 || F <= 0) return DCCN_ERR_INVALID_ARG;
    GemmParams p = gp_zero();                 // dx[rows,2kin] = dout[rows,2F] . Weff^T
    p.A = dout; p.B = w; p.C = dx;
    p.M = rows; p.N = 2 * kin; p.K = 2 * F;
    p.lda = 2 * F; p.ldb = 2 * F; p.ldc = ldc > 0 ? ldc : 2 * kin;
    p.klen = round_k(2 * F);
    p.cF = F;
    p.vecA = (F % 2 == 0) && aligned16(dout) && small_enough(rows, 2LL * F);
    p.vecB = (F % 2 == 0) && aligned16(w) && small_enough(kin, 2LL * F);
    return launch_gemm<OP_KCONTIG, OP_CCONV_WT, 0, TAG_CCONV_BWD_X>(p, 1, s);
}

// C-Conv fold and the tail's slab reduction in one launch: both are tiny, and the reduction has no consumer
// before the optimizer, so it rides on the fold instead of sitting on the critical path after the tail kernel
__global__ __launch_bounds__(256) void cconv_fold_finalize_kernel(const float* __restrict__ partial, int splits,
                                                                  long long slab, const float* __restrict__ colsum,
                                                                  float* __restrict__ dw, float* __restrict__ dbias,
                                                                  int kin, int F, int fold_blocks, TailFinalizeArgs a) {
    if ((int)blockIdx.x < fold_blocks) {
        cconv_fold_body(partial, splits, slab, colsum, dw, dbias, kin, F, blockIdx.x);
    } else {
        demod_tail_finalize_body(a, (int)blockIdx.x - fold_blocks);
    }
}

// the split-K C-Conv weight-gradient GEMM with the tail's slab reduction riding on extra blocks of the same launch
template <bool VEC, int BK = 64, int NBUF = 2>
__global__ __launch_bounds__(kGemmThreads) void cconv_bwd_w_finalize_kernel(const GemmParams p, int tiles, int gemm_blocks,
                                                                            TailFinalizeArgs a) {
    const int b = (int)blockIdx.x;
    if (b < gemm_blocks) {
        gemm_block<OP_ICONTIG, OP_ICONTIG, 64, 64, BK, 1, VEC, NBUF>(p, b % tiles, tiles, b / tiles);
    } else {
        demod_tail_finalize_body(a, b - gemm_blocks);
    }
}

// the same with the GEMM blocks in the k-major form (gemm_kmajor.h)
template <int BK>
__global__ __launch_bounds__(kGemmThreads) void cconv_bwd_w_km_finalize_kernel(const GemmParams p, int tiles, int gemm_blocks,
                                                                               TailFinalizeArgs a) {
    const int b = (int)blockIdx.x;
    if (b < gemm_blocks) {
        kmajor_block<1, BK>(p, b % tiles, tiles, b / tiles);
    } else {
        demod_tail_finalize_body(a, b - gemm_blocks);
    }
}

constexpr int kCconvBwMaxSplits = 128;
static void cconv_bw16_tiles(int variant, int& tm, int& tn) {
    tm = 64; tn = 64;
    if (variant == 2) tm = 32;
    if (variant == 3) { tm = 80; tn = 128; }
    if (variant == 4) { tm = 32; tn = 128; }
}
// workspace of the C-Conv weight gradient: the legacy split plan or up to kCconvBwMaxSplits slabs of a small output
static size_t cconv_bw_ws_bytes(int rows, int kin, int F) {
    const size_t legacy = splitk_ws_bytes(2 * kin, 2 * F, rows);
    if (4LL * kin * F > 512 * 512) return legacy;
    size_t o = 0;
    o = carve_size(o, (size_t)kCconvBwMaxSplits * 4 * kin * F * sizeof(float));
    o = carve_size(o, (size_t)kCconvBwMaxSplits * 2 * F * sizeof(float));
    o = align_up(o, 256);
    return o > legacy ? o : legacy;
}

struct FoldDefer {          // the fold left to the optimizer kernel (fused training step)
    const float* slabs; const float* colsum;
    int splits; long long slab;
};

static int cconv_bwd_w_impl(const float* x, const float* dout, float* dw, float* dbias, int rows, int kin, int F,
                            void* ws, size_t ws_bytes, hipStream_t s, const TailFinalizeArgs* fin = nullptr,
                            FoldDefer* defer = nullptr) {
    if (!x || !dout || !dw || rows <= 0 || kin <= 0 || F <= 0) return DCCN_ERR_INVALID_ARG;
    if (!ws || ws_bytes < cconv_bw_ws_bytes(rows, kin, F)) return DCCN_ERR_WORKSPACE;
    const int variant = g_tune[TUNE_CCONV_BWD_W] == kVariantKmajor ? 0 : g_tune[TUNE_CCONV_BWD_W];
    const bool v16 = variant > 0 && defer && fin && (kin % 2 == 0) && (F % 2 == 0) && aligned16(x) && aligned16(dout) &&
                     small_enough(rows, 2LL * kin) && small_enough(rows, 2LL * F) && 4LL * kin * F <= 512 * 512;
    SplitPlan sp = plan_splitk(2 * kin, 2 * F, rows);
    if (!v16 && g_tune[TUNE_CCONV_BWD_SPLITS] > 0 && defer && fin && 4LL * kin * F <= 512 * 512) {
        const int want = g_tune[TUNE_CCONV_BWD_SPLITS];
        sp = plan_splitk_n(rows, want < kCconvBwMaxSplits ? want : kCconvBwMaxSplits, 64);
    }
    if (v16) {
        int want = g_tune[TUNE_CCONV_BWD_SPLITS];
        int tm, tn;
        cconv_bw16_tiles(variant, tm, tn);
        const int tiles = ceil_div(2 * kin, tm) * ceil_div(2 * F, tn);
        if (want <= 0) want = (2 * kCUs + tiles - 1) / tiles;
        if (want > kCconvBwMaxSplits) want = kCconvBwMaxSplits;
        sp = plan_splitk_n(rows, want, 32);
    }
    Carver c(ws, ws_bytes);
    float* slabs = c.take<float>((size_t)sp.splits * 4 * kin * F);
    float* cs = c.take<float>((size_t)sp.splits * 2 * F);
    GemmParams p = gp_zero();                 // dWeff[2kin,2F] = x[rows,2kin]^T . dout[rows,2F]
    p.A = x; p.B = dout; p.C = slabs; p.colsum = cs;
    p.M = 2 * kin; p.N = 2 * F; p.K = rows;
    p.lda = 2 * kin; p.ldb = 2 * F; p.ldc = 2 * F;
    p.klen = sp.klen;
    p.slab = (long long)4 * kin * F;
    p.vecA = (kin % 2 == 0) && aligned16(x) && small_enough(rows, 2LL * kin);
    p.vecB = (F % 2 == 0) && aligned16(dout) && small_enough(rows, 2LL * F);
    if (v16) {
        switch (variant) {
            case 1: DCCN_TRY((launch_bwd_w16_finalize<2, 2, 2, 2, 32>(p, sp.splits, *fin, s))); break;     // 64x64
            case 2: DCCN_TRY((launch_bwd_w16_finalize<2, 2, 1, 2, 32>(p, sp.splits, *fin, s))); break;     // 32x64
            case 3: DCCN_TRY((launch_bwd_w16_finalize<1, 4, 5, 2, 32>(p, sp.splits, *fin, s))); break;     // 80x128
            case 4: DCCN_TRY((launch_bwd_w16_finalize<2, 2, 1, 4, 32>(p, sp.splits, *fin, s))); break;     // 32x128
            default: return DCCN_ERR_INVALID_ARG;
        }
        defer->slabs = slabs; defer->colsum = cs; defer->splits = sp.splits; defer->slab = p.slab;
        return DCCN_OK;
    }
    if (defer && fin && p.vecA && p.vecB && (F % 2 == 0)) {
        // fused step: GEMM + tail finalize in one launch; the fold happens inside the optimizer kernel
        const int tiles = ceil_div(p.N, 64) * ceil_div(p.M, 64), gemm_blocks = tiles * sp.splits;
        const dim3 grid(gemm_blocks + tail_finalize_blocks(fin->P));
        // (large outputs, e.g. N = 1024: the 32x32x2 form measured 0.7 % faster per step -- the k-major form pays off where
        // the k-loops are short)
        if (g_tune[TUNE_CCONV_BWD_W] == kVariantKmajor && kmajor_ok(p) && tiles <= 2 * kCUs) {
            auto kern = cconv_bwd_w_km_finalize_kernel<64>;
            constexpr size_t smem = kmajor_smem_bytes<64>();
            DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), smem));
            hipLaunchKernelGGL(kern, grid, dim3(kGemmThreads), smem, s, p, tiles, gemm_blocks, *fin);
        } else if (sp.klen == 128 && g_whole_k) {
            // 128 rows per split: the whole k range of a block as one tile (all loads in flight at once, no k-tile barrier)
            auto kern = cconv_bwd_w_finalize_kernel<true, 128, 1>;
            constexpr size_t smem = gemm_smem_bytes<OP_ICONTIG, OP_ICONTIG, 64, 64, 128, 1>();
            DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), smem));
            hipLaunchKernelGGL(kern, grid, dim3(kGemmThreads), smem, s, p, tiles, gemm_blocks, *fin);
        } else {
            auto kern = cconv_bwd_w_finalize_kernel<true>;
            constexpr size_t smem = gemm_smem_bytes<OP_ICONTIG, OP_ICONTIG, 64, 64, 64>();
            DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), smem));
            hipLaunchKernelGGL(kern, grid, dim3(kGemmThreads), smem, s, p, tiles, gemm_blocks, *fin);
        }
        DCCN_LAUNCH_CHECK();
        defer->slabs = slabs; defer->colsum = cs; defer->splits = sp.splits; defer->slab = p.slab;
        return DCCN_OK;
    }
    if (defer) defer->slabs = nullptr;
    if (g_tune[TUNE_CCONV_BWD_W] == kVariantKmajor && kmajor_ok(p) && ceil_div(p.N, 64) * ceil_div(p.M, 64) <= 2 * kCUs)
        DCCN_TRY((launch_kmajor<1, TAG_CCONV_BWD_W>(p, sp.splits, s)));
    else
        DCCN_TRY((launch_gemm<OP_ICONTIG, OP_ICONTIG, 1, TAG_CCONV_BWD_W>(p, sp.splits, s)));
    if (defer && !fin) {                     // the caller's optimizer launch folds the slabs (eq_opt.h)
        defer->slabs = slabs; defer->colsum = cs; defer->splits = sp.splits; defer->slab = p.slab;
        return DCCN_OK;
    }
    const int nthreads = kin * F + F;
    const int fold_blocks = ceil_div(nthreads, kRedLanes);
    if (fin)
        hipLaunchKernelGGL(cconv_fold_finalize_kernel, dim3(fold_blocks + tail_finalize_blocks(fin->P)), dim3(256), 0, s,
                           slabs, sp.splits, p.slab, cs, dw, dbias, kin, F, fold_blocks, *fin);
    else
        hipLaunchKernelGGL(cconv_fold_kernel, dim3(fold_blocks), dim3(256), 0, s, slabs, sp.splits, p.slab, cs, dw, dbias,
                           kin, F);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// ---------------------------------------------------------------------------------------
// same-shaped C-Conv layers as grouped launches (the equaliser's corr / eq pair, model.py:439-449)
// ---------------------------------------------------------------------------------------
// forward of `groups` (1,kin)->F C-Convs: operands of group g at element strides gx / gw / gb from group 0.
// join_pairs: the two outputs are the IQ-pair streams of ONE [rows, F, 4] tensor (tf.concat on the last axis,
// model.py:456): out = that tensor, group g writes floats 2g, 2g+1 of every cell.  Returns false when the shapes do
// not qualify for the grouped kernels (the caller then runs the layers one by one).
static bool cconv_pair_ok(const float* x, const float* w, const float* o, int rows, int kin, int F, long long gx, long long gw,
                          long long go) {
    return (kin % 2 == 0) && (F % 2 == 0) && (kin % 32 == 0) && (F % 32 == 0) && aligned16(x) && aligned16(w) && aligned16(o) &&
           (gx % 4 == 0) && (gw % 4 == 0) && (go % 2 == 0) && small_enough(rows, 4LL * (kin > F ? kin : F)) &&
           small_enough(kin, 2LL * F) && (long long)ceil_div(rows, 128) * ceil_div(2 * F, 128) < 2 * kCUs;
}
static int cconv_fwd_grouped_impl(const float* x, const float* w, const float* bias, float* out, int rows, int kin, int F,
                                  int groups, long long gx, long long gw, long long gb, bool join_pairs, hipStream_t s) {
    if (!x || !w || !out || rows <= 0 || groups < 1 || groups > 2 || (join_pairs && groups != 2)) return DCCN_ERR_INVALID_ARG;
    GemmParams p = gp_zero();                 // out_g[rows,2F] = x_g[rows,2kin] . Weff_g[2kin,2F]
    p.A = x; p.B = w; p.C = out; p.bias = bias; p.cbias = 1;
    p.M = rows; p.N = 2 * F; p.K = 2 * kin;
    p.lda = 2 * kin; p.ldb = 2 * F; p.ldc = join_pairs ? 4 * F : 2 * F;
    p.klen = round_k(2 * kin);
    p.cF = F;
    p.vecA = 1; p.vecB = 1;
    GroupStride gs;
    gs.a = gx; gs.b = gw; gs.bias = gb; gs.colsum = 0;
    gs.c = join_pairs ? 2 : (long long)rows * 2 * F;
    if (join_pairs) {
        if (p.K == 128 && g_whole_k)
            return launch_gemm_grouped<OP_KCONTIG, OP_CCONV_W, 64, 64, 128, TAG_CCONV_FWD, 1, CMAP_JOIN_PAIRS>(p, gs, groups, s);
        return launch_gemm_grouped<OP_KCONTIG, OP_CCONV_W, 64, 64, 64, TAG_CCONV_FWD, 2, CMAP_JOIN_PAIRS>(p, gs, groups, s);
    }
    return launch_gemm_grouped<OP_KCONTIG, OP_CCONV_W, 64, 64, 64, TAG_CCONV_FWD, 2, CMAP_NONE>(p, gs, groups, s);
}

// backward of `groups` (1,kin)->F C-Convs in ONE launch: dx_g = dout_g . Weff_g^T and the split-K slabs of
// dWeff_g = x_g^T . dout_g (+ column sums), left un-folded in ws for the optimizer launch (defer[g]).
// Strides in elements: gx (x and dx), gd (dout), gw (kernels).
static size_t cconv_bwd_grouped_ws_bytes(int rows, int kin, int F, int groups) {
    return (size_t)groups * cconv_bw_ws_bytes(rows, kin, F);
}
static int cconv_bwd_grouped_impl(const float* x, const float* dout, const float* w, float* dx, int rows, int kin, int F,
                                  int groups, long long gx, long long gd, long long gw, void* ws, size_t ws_bytes,
                                  FoldDefer* defer, hipStream_t s) {
    if (!x || !dout || !w || !dx || !defer || rows <= 0 || groups < 1 || groups > 2) return DCCN_ERR_INVALID_ARG;
    if (!ws || ws_bytes < cconv_bwd_grouped_ws_bytes(rows, kin, F, groups)) return DCCN_ERR_WORKSPACE;
    SplitPlan sp = plan_splitk(2 * kin, 2 * F, rows);
    // (one output tile and > 16 384 rows would plan more slabs than the workspace holds: clamp like cconv_bwd_w_impl does)
    if (sp.splits > kCconvBwMaxSplits) sp = plan_splitk_n(rows, kCconvBwMaxSplits, 64);
    if (sp.splits > kCconvBwMaxSplits) return DCCN_ERR_STATE;
    const size_t per = cconv_bw_ws_bytes(rows, kin, F) / sizeof(float);
    float* slabs = static_cast<float*>(ws);
    float* cs = slabs + (size_t)sp.splits * 4 * kin * F;
    GemmParams px = gp_zero();                // dx[rows,2kin] = dout[rows,2F] . Weff^T
    px.A = dout; px.B = w; px.C = dx;
    px.M = rows; px.N = 2 * kin; px.K = 2 * F;
    px.lda = 2 * F; px.ldb = 2 * F; px.ldc = 2 * kin;
    px.klen = round_k(2 * F);
    px.cF = F;
    px.vecA = 1; px.vecB = 1;
    GroupStride g1;
    g1.a = gd; g1.b = gw; g1.c = gx; g1.bias = 0; g1.colsum = 0;
    GemmParams pw = gp_zero();                // dWeff[2kin,2F] = x[rows,2kin]^T . dout[rows,2F]
    pw.A = x; pw.B = dout; pw.C = slabs; pw.colsum = cs;
    pw.M = 2 * kin; pw.N = 2 * F; pw.K = rows;
    pw.lda = 2 * kin; pw.ldb = 2 * F; pw.ldc = 2 * F;
    pw.klen = sp.klen;
    pw.slab = (long long)4 * kin * F;
    pw.vecA = 1; pw.vecB = 1;
    if (!kmajor_ok(pw)) return DCCN_ERR_STATE;
    GroupStride g2;
    g2.a = gx; g2.b = gd; g2.c = (long long)per; g2.bias = 0; g2.colsum = (long long)per;
    DCCN_TRY(launch_cconv_bwd_grouped_km<64>(px, g1, pw, g2, sp.splits, groups, s));
    for (int g = 0; g < groups; ++g) {
        defer[g].slabs = slabs + (size_t)g * per; defer[g].colsum = cs + (size_t)g * per;
        defer[g].splits = sp.splits; defer[g].slab = pw.slab;
    }
    return DCCN_OK;
}

// ---------------------------------------------------------------------------------------
// backward of the basic receiver's training step in one launch (rx_bwd.h)
// ---------------------------------------------------------------------------------------
static int rx_bwd_fused_tiles(int batch, int S, int F) { return ceil_div(batch, 64) * ceil_div(S * 2 * F, 64); }
static size_t rx_bwd_fused_ws_bytes(int batch, int S, int kin, int F) {
    const size_t tiles = (size_t)rx_bwd_fused_tiles(batch, S, F);
    size_t o = 0;
    o = carve_size(o, tiles * kin * 64 * sizeof(float));        // folded per tile: [kin][32][{a, b}] (rx_bwd.h)
    o = carve_size(o, tiles * 64 * sizeof(float));
    return align_up(o, 256);
}
// applicable: 64x64 dX tiles that lie inside one symbol's 2F columns, 2kin = 128 or 160 (N = 64 without / with the
// cyclic prefix), the grouped k-major plan for the dense gradients, vector-legal operands
static bool rx_bwd_fused_ok(int batch, int S, int kin, int F, int D, const float* x_norm, const float* fft_out,
                            const float* dz, const float* wd) {
    const int dK = S * 2 * F, dN = 2 * D;
    if (!g_tune[TUNE_FUSED_BWD] || g_tune[TUNE_DENSE_BWD] != kVariantKmajor) return false;
    if ((2 * F) % 64 != 0 || (2 * kin != 128 && 2 * kin != 160)) return false;
    if (dN % 4 != 0 || !aligned16(x_norm) || !aligned16(fft_out) || !aligned16(dz) || !aligned16(wd)) return false;
    if (!small_enough(batch, (long long)S * 2 * kin) || !small_enough(batch, dK) || !small_enough(dK, dN)) return false;
    const long long big = (long long)ceil_div(batch, 128) * ceil_div(dK, 128);
    if (big >= 2 * kCUs) return false;
    const SplitPlan sp = dense_dw_plan(batch, dK, dN, kFusedBwdRangeRows);
    return (long long)ceil_div(dK, 128) * ceil_div(dN, 128) * sp.splits < 2 * kCUs;
}

// dfft nullable: the dX tiles then only feed their epilogue
static int rx_bwd_fused_impl(const float* x_norm, const float* fft_out, const float* dz, const float* wd, float* dfft,
                             float* dbias_dense, int batch, int S, int kin, int F, int D, void* ws_dense, size_t ws_dense_bytes,
                             void* ws_conv, size_t ws_conv_bytes, const NormRideArgs& nr, const TailFinalizeArgs& fin,
                             dccn_adam_hparams hp, hipStream_t s, DeferredSlabs* ds, FoldDefer* fd, int* fold_tilew) {
    const int dK = S * 2 * F, dN = 2 * D;
    if (!ws_dense || ws_dense_bytes < splitk_ws_bytes(dK, dN, batch)) return DCCN_ERR_WORKSPACE;
    if (!ws_conv || ws_conv_bytes < rx_bwd_fused_ws_bytes(batch, S, kin, F)) return DCCN_ERR_WORKSPACE;
    GemmParams px = dense_bwd_x_params(dz, wd, dfft, batch, dK, dN);
    GemmParams pw = gp_zero();                // dw[K,N] = x[M,K]^T . dy[M,N]
    pw.A = fft_out; pw.B = dz;
    pw.M = dK; pw.N = dN; pw.K = batch;
    pw.lda = dK; pw.ldb = dN; pw.ldc = dN;
    pw.slab = (long long)dK * dN;
    pw.vecA = 1; pw.vecB = 1;
    const SplitPlan sp = dense_dw_plan(batch, dK, dN, kFusedBwdRangeRows);
    pw.klen = sp.klen;
    int nsplit = sp.splits;
    if (g_tune[TUNE_DW_GRADED] > 0 && sp.splits > 1 && batch >= 768) {        // (>= 12 k-tiles: else the last ranges get too short)
        const int n = graded_ranges(g_tune[TUNE_DW_GRADED], batch, pw.koff);
        if (n > 1 && n <= max_splits16(dK, dN)) { pw.nranges = n; nsplit = n; }
    }
    Carver c(ws_dense, ws_dense_bytes);                   // (sized for max_splits16 slabs: splitk_ws_bytes)
    float* slabs = c.take<float>((size_t)nsplit * dK * dN);
    float* cs = c.take<float>((size_t)nsplit * dN);
    if (!c.ok()) return DCCN_ERR_WORKSPACE;
    pw.C = slabs; pw.colsum = dbias_dense ? cs : nullptr;
    if (!kmajor_ok(pw)) return DCCN_ERR_INVALID_ARG;
    const int tiles = rx_bwd_fused_tiles(batch, S, F);
    Carver cc(ws_conv, ws_conv_bytes);
    DweffArgs de;
    de.xn = x_norm;
    de.partial = cc.take<float>((size_t)tiles * kin * 64);
    de.colsum = cc.take<float>((size_t)tiles * 64);
    de.batch = batch; de.ldx = S * 2 * kin; de.two_kin = 2 * kin; de.two_F = 2 * F;
    de.prio = g_tune[TUNE_BWD_PRIO];
    if (2 * kin == 160) DCCN_TRY(launch_rx_bwd_fused<5>(px, pw, de, nsplit, nr, fin, hp, s));
    else DCCN_TRY(launch_rx_bwd_fused<4>(px, pw, de, nsplit, nr, fin, hp, s));
    ds->dw_slabs = slabs; ds->db_slabs = dbias_dense ? cs : nullptr; ds->splits = nsplit;
    fd->slabs = de.partial; fd->colsum = de.colsum;
    fd->splits = ceil_div(batch, 64) * S;                       // terms per element: (row tile, symbol)
    fd->slab = (long long)((2 * F) / 64) * kin * 64;            // distance between consecutive terms (tiles folded: [kin][32][2])
    *fold_tilew = 64;
    return DCCN_OK;
}

// ---------------------------------------------------------------------------------------
// tail
// ---------------------------------------------------------------------------------------
static int tail_blocks(long long cells, bool quad4 = false) {
    // quad4 (nbits = 4 training): four lanes per cell, 194 VGPRs -> two blocks per CU
    long long b = ceil_div_ll(quad4 ? 4 * cells : cells, kTailThreads);
    const long long cap = quad4 ? kTailBlocksMax : kTailBlocks;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}
static size_t tail_ws_bytes(long long cells, int nbits) {
    size_t o = 0;
    o = carve_size(o, (size_t)kTailBlocksMax * sizeof(TailBlockMetrics));
    o = carve_size(o, (size_t)kTailBlocksMax * tail_param_count(nbits) * sizeof(float));
    (void)cells;
    return align_up(o, 256);
}


template <int NB>
static int tail_launch(bool bwd, const float* z, const int32_t* bits, const float* tailp, float* prob, float* dz,
                       long long cells, int nblk, TailBlockMetrics* bm, float* bg, hipStream_t s) {
    if (bwd)
        DCCN_LAUNCH_CHAINS_Z((demod_tail_kernel<NB, true>), dim3(nblk), dim3(kTailThreads), 0, s, z, bits, tailp, prob,
                             dz, cells, bm, bg);
    else
        DCCN_LAUNCH_CHAINS_Z((demod_tail_kernel<NB, false>), dim3(nblk), dim3(kTailThreads), 0, s, z, bits, tailp, prob,
                             dz, cells, bm, bg);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// pp/power_out: optional R8 finish riding on the slab-reduction kernel (fused receiver step)
// defer != nullptr: do not launch the slab reduction; hand its arguments to the caller (fused receiver step)
static int tail_impl(bool bwd, const float* z, const int32_t* bits, const float* tailp, float* prob,
                     dccn_metrics* metrics, float* dz, float* dtailp, long long cells, int nbits,
                     const PowerPartials* pp, float* power_out, void* ws, size_t ws_bytes, hipStream_t s,
                     TailFinalizeArgs* defer = nullptr) {
    if (!z || !bits || !tailp || !metrics || cells <= 0 || nbits < 1 || nbits > 4) return DCCN_ERR_INVALID_ARG;
    if (bwd && (!dz || !dtailp)) return DCCN_ERR_INVALID_ARG;
    if (!ws || ws_bytes < tail_ws_bytes(cells, nbits)) return DCCN_ERR_WORKSPACE;
    Carver c(ws, ws_bytes);
    TailBlockMetrics* bm = c.take<TailBlockMetrics>(kTailBlocksMax);
    float* bg = c.take<float>((size_t)kTailBlocksMax * tail_param_count(nbits));
    const int nblk = tail_blocks(cells, bwd && nbits == 4);
    int st = DCCN_ERR_INVALID_ARG;
    switch (nbits) {
        case 1: st = tail_launch<1>(bwd, z, bits, tailp, prob, dz, cells, nblk, bm, bg, s); break;
        case 2: st = tail_launch<2>(bwd, z, bits, tailp, prob, dz, cells, nblk, bm, bg, s); break;
        case 3: st = tail_launch<3>(bwd, z, bits, tailp, prob, dz, cells, nblk, bm, bg, s); break;
        case 4:
            if (bwd) {                                   // four lanes per cell (tail.h): 50 accumulators per lane, not 200
                if (prob)
                    DCCN_LAUNCH_CHAINS_Z(demod_tail_quad4_kernel<true>, dim3(nblk), dim3(kTailThreads), 0, s, z, bits, tailp,
                                         prob, dz, cells, bm, bg, tl_stamp);
                else
                    DCCN_LAUNCH_CHAINS_Z(demod_tail_quad4_kernel<false>, dim3(nblk), dim3(kTailThreads), 0, s, z, bits,
                                         tailp, prob, dz, cells, bm, bg, tl_stamp);
                DCCN_LAUNCH_CHECK();
                st = DCCN_OK;
            } else {
                st = tail_launch<4>(bwd, z, bits, tailp, prob, dz, cells, nblk, bm, bg, s);
            }
            break;
    }
    DCCN_TRY(st);
    const int P = bwd ? tail_param_count(nbits) : 0;
    const bool pw = pp != nullptr && power_out != nullptr;
    TailFinalizeArgs fa;
    fa.blk_metrics = bm; fa.blk_grads = bwd ? bg : nullptr; fa.nblocks = nblk; fa.P = P; fa.count = cells * nbits;
    fa.metrics = metrics; fa.dtailp = bwd ? dtailp : nullptr; fa.power_partial = pw ? pp->partial : nullptr;
    fa.n_power = pw ? pp->n : 0; fa.power_denom = pw ? pp->denom : 1.0; fa.power_out = pw ? power_out : nullptr;
    fa.adam = nullptr; memset(&fa.hp, 0, sizeof(fa.hp)); fa.zero_word = nullptr; fa.zero_flags = nullptr; fa.n_zero_flags = 0; fa.zero_stride = 0; fa.mon_acc = nullptr; fa.mon_noise = nullptr;
    if (defer) {
        *defer = fa;
        return DCCN_OK;
    }
    DCCN_NO_CHAINS();
    hipLaunchKernelGGL(demod_tail_finalize_kernel, dim3(tail_finalize_blocks(P)), dim3(256), 0, s, fa);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// ---------------------------------------------------------------------------------------
// dense forward with the tail fused into its epilogue (gemm16.h EPI_TAIL: register layout for nbits <= 2, tile staged
// through LDS for nbits >= 3)
// ---------------------------------------------------------------------------------------
static void dense_tail_tiles(int variant, int& bm, int& bn) {
    bn = 64;
    bm = variant == 13 ? 80 : 48;
}
static int dense_tail_max_blocks(int M, int N) {
    const int b = ceil_div(M, 32) * ceil_div(N, 64);
    // few rows: the one-latency 16x16 tiles of fewrow.h carry the tail as well (one slab per tile)
    const int few = (M <= 96 && (N % 16) == 0) ? ceil_div(M, 16) * (N / 16) : 0;
    return few > b && few <= kTailBlocksMax ? few : b;
}
static size_t dense_tail_ws_bytes(int M, int N, int nbits) {
    const size_t nb = (size_t)dense_tail_max_blocks(M, N);
    size_t o = 0;
    o = carve_size(o, nb * sizeof(TailBlockMetrics));
    o = carve_size(o, nb * tail_param_count(nbits) * sizeof(float));
    return align_up(o, 256);
}
static bool dense_tail_shape_ok(int M, int K, int N, int nbits) {
    // nbits >= 3 (tail weights in LDS; nbits = 4 training in the quad-lane form): knob 13
    return g_tune[TUNE_DENSE_FWD] > 0 && nbits >= 1 && nbits <= 4 && M > 0 && K > 0 && N > 0 && (K % 4 == 0) && (N % 4 == 0) &&
           small_enough(M, K) && small_enough(K, N) && (long long)ceil_div(M, 128) * ceil_div(N, 128) < 2 * kCUs;
}
static bool dense_tail_ok(const float* x, const float* w, int M, int K, int N, int nbits) {
    return dense_tail_shape_ok(M, K, N, nbits) && aligned16(x) && aligned16(w);
}
// which steps take the fused launch for 8-QAM / 16-QAM (knob 13: bit 0 = the lane-per-cell forms -- nbits 3, and nbits 4
// evaluation; bit 1 = the quad-lane form of 16-QAM training).  The operator dccn_dense_tail_* itself accepts every nbits.
// bit 2: BPSK / QPSK steps of LARGE layers (>= two rounds of 128x128 tiles) run the dense forward on the 128x128x32 tile
// family and the tail as its own launch.
static bool dense_tail_planned(int nbits, bool train, int M = 0, int N = 0) {
    const int k = g_tune[TUNE_TAIL_FUSE_HI];
    if (nbits <= 2) return !((k & 4) && (long long)ceil_div(M, 128) * ceil_div(N, 128) >= 2 * kCUs);
    return (nbits == 4 && train) ? (k & 2) != 0 : (k & 1) != 0;
}

// the two tile shapes the fused dense + tail launch runs: 48x64 with loads two k-tiles ahead (small layers), 80x64 (large layers)
template <int NB, bool BWD>
static int dense_tail_launch(int variant, const GemmParams& p, const TailEpiParams& tp, hipStream_t s) {
    const size_t sm = tune_smem_min();
    if (variant == 13) return launch_dense_tail16<1, 4, 5, 1, 64, 1, NB, BWD, 2>(p, tp, s, sm);    // 80x64 (large layers)
    return launch_dense_tail16<1, 4, 3, 1, 64, 1, NB, BWD, 2>(p, tp, s, sm);                        // 48x64
}

// nbits >= 3: the two tile shapes the training / sweep steps use (every other variant number runs 48x64)
template <int NB, bool BWD>
static int dense_tail_hi_launch(int variant, const GemmParams& p, const TailEpiParams& tp, hipStream_t s) {
    const size_t sm = tune_smem_min();
    if (variant == 13) return launch_dense_tail16<1, 4, 5, 1, 64, 1, NB, BWD, 2>(p, tp, s, sm);      // 80x64 (large layers)
    return launch_dense_tail16<1, 4, 3, 1, 64, 1, NB, BWD, 2>(p, tp, s, sm);                          // 48x64
}

// z nullable (not materialised then).  defer: as tail_impl.
static int dense_tail_impl(bool bwd, const float* x, const float* w, const float* bias, float* z, const int32_t* bits,
                           const float* tailp, float* prob, dccn_metrics* metrics, float* dz, float* dtailp, int M,
                           int K, int N, int nbits, const PowerPartials* pp, float* power_out, void* ws, size_t ws_bytes,
                           hipStream_t s, TailFinalizeArgs* defer = nullptr) {
    if (!x || !w || !bits || !tailp || !metrics || M <= 0 || K <= 0 || N <= 0 || (N & 1) || nbits < 1 || nbits > 4)
        return DCCN_ERR_INVALID_ARG;
    if (bwd && (!dz || !dtailp)) return DCCN_ERR_INVALID_ARG;
    if (!dense_tail_ok(x, w, M, K, N, nbits)) return DCCN_ERR_INVALID_ARG;
    if (!ws || ws_bytes < dense_tail_ws_bytes(M, N, nbits)) return DCCN_ERR_WORKSPACE;
    // large layers (several rounds of 48x64 tiles): 80x64 tiles re-use the B tile for five row blocks instead of three
    // (C4: 1.42 -> 1.34 ms); the lane's ten cells go through the tail in two batches of five
    const int variant = ((long long)ceil_div(M, 48) * ceil_div(N, 64) >= 4LL * kCUs) ? 13 : 9;
    int bm, bn;
    dense_tail_tiles(variant, bm, bn);
    const int nblk = ceil_div(M, bm) * ceil_div(N, bn);
    Carver c(ws, ws_bytes);
    TailBlockMetrics* bmx = c.take<TailBlockMetrics>((size_t)dense_tail_max_blocks(M, N));
    float* bg = c.take<float>((size_t)dense_tail_max_blocks(M, N) * tail_param_count(nbits));
    GemmParams p = gp_zero();
    p.A = x; p.B = w; p.C = z; p.bias = bias;
    p.M = M; p.N = N; p.K = K;
    p.lda = K; p.ldb = N; p.ldc = N;
    p.klen = round_k(K);
    p.vecA = 1; p.vecB = 1;
    const long long cells = (long long)M * (N / 2);
    TailEpiParams tp;
    tp.bits = bits; tp.tailp = tailp; tp.prob = prob; tp.dz = dz; tp.blk_metrics = bmx; tp.blk_grads = bg;
    tp.inv_count = 1.0f / (float)(cells * nbits);
    // few rows, BPSK / QPSK: every operand of a 16x16 tile requested at once, the tail on the tile's own registers (fewrow.h)
    const int few_tiles = ceil_div(M, 16) * (N / 16);
    const bool few = g_tune[TUNE_FEWROW] && g_tune[TUNE_SKINNY] > 0 && nbits <= 2 && M <= 96 && fewrow_ng_c(p, false) >= 10 &&
                     few_tiles <= kTailBlocksMax && few_tiles <= dense_tail_max_blocks(M, N);
    int st;
    // 80x64 tiles over a row count that leaves <= 32 rows for the last tile row (C4: 585 = 7 x 80 + 25): that row as 32x64 blocks
    const int rag = (variant == 13 && nbits <= 2 && g_tune[TUNE_DENSE_RAGGED] && M > 80) ? M % 80 : 0;
    if (rag > 0 && rag <= 32) {
        const int M1 = M - rag;
        GemmParams p1 = p, p2 = p;
        TailEpiParams t1 = tp, t2 = tp;
        p1.M = M1;
        p2.M = rag; p2.A = p.A + (size_t)M1 * p.lda; p2.C = p.C ? p.C + (size_t)M1 * p.ldc : nullptr;
        t2.bits = tp.bits + (size_t)M1 * (N / 2) * nbits;
        t2.prob = tp.prob ? tp.prob + (size_t)M1 * (N / 2) * nbits * 2 : nullptr;
        t2.dz = tp.dz ? tp.dz + (size_t)M1 * N : nullptr;
        const size_t sm = tune_smem_min();
        if (nbits == 1) st = bwd ? launch_dense_tail16_ragged<5, 2, 64, 1, true, 2>(p1, t1, p2, t2, s, sm) : launch_dense_tail16_ragged<5, 2, 64, 1, false, 2>(p1, t1, p2, t2, s, sm);
        else st = bwd ? launch_dense_tail16_ragged<5, 2, 64, 2, true, 2>(p1, t1, p2, t2, s, sm) : launch_dense_tail16_ragged<5, 2, 64, 2, false, 2>(p1, t1, p2, t2, s, sm);
        DCCN_TRY(st);
        const int nrag = (M1 / 80) * ceil_div(N, 64) + ceil_div(N, 64);
        const int P = bwd ? tail_param_count(nbits) : 0;
        const bool pw = pp != nullptr && power_out != nullptr;
        TailFinalizeArgs fa;
        fa.blk_metrics = bmx; fa.blk_grads = bwd ? bg : nullptr; fa.nblocks = nrag; fa.P = P; fa.count = cells * nbits;
        fa.metrics = metrics; fa.dtailp = bwd ? dtailp : nullptr; fa.power_partial = pw ? pp->partial : nullptr;
        fa.n_power = pw ? pp->n : 0; fa.power_denom = pw ? pp->denom : 1.0; fa.power_out = pw ? power_out : nullptr;
        fa.adam = nullptr; memset(&fa.hp, 0, sizeof(fa.hp)); fa.zero_word = nullptr; fa.zero_flags = nullptr; fa.n_zero_flags = 0; fa.zero_stride = 0; fa.mon_acc = nullptr; fa.mon_noise = nullptr;
        if (defer) {
            *defer = fa;
            return DCCN_OK;
        }
        hipLaunchKernelGGL(demod_tail_finalize_kernel, dim3(tail_finalize_blocks(P)), dim3(256), 0, s, fa);
        DCCN_LAUNCH_CHECK();
        return DCCN_OK;
    }
    if (few && nbits == 1) st = bwd ? launch_fewrow_tail<1, true>(p, tp, s) : launch_fewrow_tail<1, false>(p, tp, s);
    else if (few && nbits == 2) st = bwd ? launch_fewrow_tail<2, true>(p, tp, s) : launch_fewrow_tail<2, false>(p, tp, s);
    else if (nbits == 1) st = bwd ? dense_tail_launch<1, true>(variant, p, tp, s) : dense_tail_launch<1, false>(variant, p, tp, s);
    else if (nbits == 2) st = bwd ? dense_tail_launch<2, true>(variant, p, tp, s) : dense_tail_launch<2, false>(variant, p, tp, s);
    else if (nbits == 3) st = bwd ? dense_tail_hi_launch<3, true>(variant, p, tp, s) : dense_tail_hi_launch<3, false>(variant, p, tp, s);
    else st = bwd ? dense_tail_hi_launch<4, true>(variant, p, tp, s) : dense_tail_hi_launch<4, false>(variant, p, tp, s);
    DCCN_TRY(st);
    const int P = bwd ? tail_param_count(nbits) : 0;
    const bool pw = pp != nullptr && power_out != nullptr;
    TailFinalizeArgs fa;
    fa.blk_metrics = bmx; fa.blk_grads = bwd ? bg : nullptr; fa.nblocks = few ? few_tiles : nblk; fa.P = P; fa.count = cells * nbits;
    fa.metrics = metrics; fa.dtailp = bwd ? dtailp : nullptr; fa.power_partial = pw ? pp->partial : nullptr;
    fa.n_power = pw ? pp->n : 0; fa.power_denom = pw ? pp->denom : 1.0; fa.power_out = pw ? power_out : nullptr;
    fa.adam = nullptr; memset(&fa.hp, 0, sizeof(fa.hp)); fa.zero_word = nullptr; fa.zero_flags = nullptr; fa.n_zero_flags = 0; fa.zero_stride = 0; fa.mon_acc = nullptr; fa.mon_noise = nullptr;
    if (defer) {
        *defer = fa;
        return DCCN_OK;
    }
    hipLaunchKernelGGL(demod_tail_finalize_kernel, dim3(tail_finalize_blocks(P)), dim3(256), 0, s, fa);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// prep = false: the per-step bookkeeping (alpha, beta powers, global_step) already rode on an earlier kernel of the step
static int adam_impl(float* param, const float* grad, float* m, float* v, const float* reg_coef,
                     const float* reg_gate, dccn_adam_state* st, dccn_adam_hparams hp, long long n, hipStream_t s,
                     bool prep = true) {
    if (!param || !grad || !m || !v || !st || n <= 0) return DCCN_ERR_INVALID_ARG;
    if (prep) {
        hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, s, st, hp);
        DCCN_LAUNCH_CHECK();
    }
    long long blocks = ceil_div_ll(ceil_div_ll(n, 4), 256);
    if (blocks > 4 * kCUs) blocks = 4 * kCUs;
    hipLaunchKernelGGL(adam_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, s, param, grad, m, v, reg_coef,
                       reg_gate, st, hp, n);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// ---------------------------------------------------------------------------------------
// fused receiver step
// ---------------------------------------------------------------------------------------
static bool shape_ok(const dccn_rx_shape* sh) {
    return sh && sh->batch > 0 && sh->S > 0 && sh->kin > 0 && sh->F > 0 && sh->D > 0 && sh->nbits >= 1 &&
           sh->nbits <= 4;
}

struct RxLayout {
    long long o_conv_w, o_conv_b, o_dense_w, o_dense_b, o_tail, total;
    int rows, cols, dK, dN;
    long long cells;
    size_t ws_norm, ws_tail, ws_dense_bw, ws_conv_bw, ws_sync;
};
static RxLayout rx_layout(const dccn_rx_shape* sh) {
    RxLayout L;
    const long long F2 = 2LL * sh->F;
    L.o_conv_w = 0;
    L.o_conv_b = L.o_conv_w + (long long)sh->kin * F2;
    L.o_dense_w = L.o_conv_b + F2;
    L.o_dense_b = L.o_dense_w + (long long)sh->S * F2 * 2 * sh->D;
    L.o_tail = L.o_dense_b + 2LL * sh->D;
    L.total = L.o_tail + tail_param_count(sh->nbits);
    L.rows = sh->batch * sh->S;
    L.cols = sh->S * sh->kin * 2;
    L.dK = sh->S * sh->F * 2;
    L.dN = 2 * sh->D;
    L.cells = (long long)sh->batch * sh->D;
    L.ws_norm = norm_ws_bytes(sh->batch, L.cols);
    L.ws_tail = tail_ws_bytes(L.cells, sh->nbits);
    {
        const size_t f = dense_tail_ws_bytes(sh->batch, L.dN, sh->nbits);
        if (f > L.ws_tail) L.ws_tail = f;
    }
    L.ws_dense_bw = splitk_ws_bytes(L.dK, L.dN, sh->batch);
    // hand-off words of the update + prefetch launch: arrival counter + one flag per C-Conv forward tile, 256 B apart
    L.ws_sync = 256 + (size_t)ceil_div(L.rows, 64) * ceil_div(2 * sh->F, 64) * kFlagStride * sizeof(unsigned);
    L.ws_conv_bw = cconv_bw_ws_bytes(L.rows, sh->kin, sh->F);
    {
        const size_t f = rx_bwd_fused_ws_bytes(sh->batch, sh->S, sh->kin, sh->F);
        if (2LL * sh->kin <= 192 && f > L.ws_conv_bw) L.ws_conv_bw = f;
    }
    return L;
}
static size_t rx_ws_bytes(const dccn_rx_shape* sh, int train) {
    const RxLayout L = rx_layout(sh);
    size_t o = 0;
    o = carve_size(o, L.ws_norm);
    o = carve_size(o, L.ws_tail);
    if (train) {
        o = carve_size(o, L.ws_dense_bw);
        o = carve_size(o, L.ws_conv_bw);
        o = carve_size(o, L.ws_sync);
    }
    return align_up(o, 256);
}

// The library's own second stream (one per device) and a pair of events per host thread: large layers run the dense kernel's
// optimizer update -- 3.2 GB of pure HBM traffic at N = 1024 -- NEXT TO the MFMA-bound C-Conv weight-gradient launch instead
// of behind it.  (Two kernels of one stream never overlap; blocks of two streams share the CUs.)
struct OverlapStreams {
    hipStream_t side;
    hipEvent_t fork, join;
};
static bool overlap_streams(OverlapStreams* o) {
    static std::mutex mu;
    static hipStream_t sides[64];
    thread_local hipEvent_t ev[64][2];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!sides[dev]) {
            // lowest priority: its own hardware queue (streams of one priority share a small pool of queues round-robin, and a
            // stream that lands on the main stream's queue does not overlap with it at all), and the MFMA-bound launch it
            // runs next to is served first whenever a CU has room
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
            if (hipStreamCreateWithPriority(&sides[dev], hipStreamNonBlocking, least) != hipSuccess) { sides[dev] = nullptr; return false; }
        }
    }
    for (int k = 0; k < 2; ++k)
        if (!ev[dev][k] && hipEventCreateWithFlags(&ev[dev][k], hipEventDisableTiming) != hipSuccess) { ev[dev][k] = nullptr; return false; }
    o->side = sides[dev]; o->fork = ev[dev][0]; o->join = ev[dev][1];
    return true;
}

// ---- fused static-channel generator launch (datagen.h gen_static_frames_kernel; ABI entry points further down) ----
static bool gen_static_ok(const dccn_gen_static* g) {
    if (!g || !g->bits_out || !g->cell_map || !g->const_tab || !g->idft || !g->snr_db || !g->y || !g->noise || !g->power_partial)
        return false;
    if (g->frames <= 0 || g->frames > 65535 || g->S <= 0 || 2 * g->S > 16 || g->K <= 0 || g->CP < 0 || g->D <= 0 || g->nbits < 1 ||
        g->nbits > 4 || (g->n_profiles == 0 && (g->L <= 0 || g->L > 64)))
        return false;
    if (g->n_profiles < 0 || g->n_profiles > kGenMaxProfiles || (g->n_profiles > 0 && !g->profiles) || g->tap_stride < 0 ||
        (g->H_out && g->h_rep <= 0))
        return false;
    if (g->n_profiles == 0) {
        if (!g->identity && (!g->coeff || !g->alpha || g->n_taps <= 0 || g->n_taps > 16)) return false;
        if (g->tap_stride != 0 && g->tap_stride < g->n_taps) return false;
    }
    for (int i = 0; i < g->n_profiles; ++i) {
        const dccn_gen_profile& q = g->profiles[i];
        if (q.L <= 0 || q.L > 64) return false;
        if (!q.identity && (!q.coeff || !q.alpha || q.n_taps <= 0 || q.n_taps > 16 || g->tap_stride < q.n_taps)) return false;
    }
    // the instantiated shape: the reference's N = 64 frame with the long cyclic prefix, 7 symbols x (64 + 16) samples
    return g->S == 7 && g->K == 64 && g->CP == 16 && aligned16(g->y) && aligned16(g->noise);
}
// the generator launch's argument block from its descriptor (also used by launches that carry the generator's workgroups as
// riders: eq_step.h)
static int gen_static_args(const dccn_gen_static* g, GenStaticArgs* out) {
    if (!gen_static_ok(g)) return DCCN_ERR_INVALID_ARG;
    if (ceil_div(g->frames, kGenFramesPerBlock) > kChanPartials) return DCCN_ERR_INVALID_ARG;
    GenStaticArgs& a = *out;
    a.bits_out = g->bits_out; a.cell_map = g->cell_map; a.const_tab = reinterpret_cast<const float2*>(g->const_tab);
    a.pilot = make_float2(g->pilot_re, g->pilot_im); a.idft = g->idft;
    memset(a.prof, 0, sizeof(a.prof));
    if (g->n_profiles == 0) {
        a.prof[0].coeff = g->coeff; a.prof[0].alpha = g->alpha; a.prof[0].n_taps = g->n_taps; a.prof[0].L = g->L;
        a.prof[0].identity = g->identity;
        a.n_prof = 1;
        a.tap_stride = g->tap_stride > 0 ? g->tap_stride : g->n_taps;
    } else {
        for (int i = 0; i < g->n_profiles; ++i) {
            a.prof[i].coeff = g->profiles[i].coeff; a.prof[i].alpha = g->profiles[i].alpha;
            a.prof[i].n_taps = g->profiles[i].n_taps; a.prof[i].L = g->profiles[i].L; a.prof[i].identity = g->profiles[i].identity;
        }
        a.n_prof = g->n_profiles;
        a.tap_stride = g->tap_stride;
    }
    a.H = reinterpret_cast<float2*>(g->H_out); a.h_rep = g->h_rep;
    a.snr_db = g->snr_db; a.y = reinterpret_cast<float2*>(g->y); a.noise = reinterpret_cast<float2*>(g->noise);
    a.power_partial = g->power_partial; a.noise_partial = g->noise_partial; a.tx_out = g->tx_out;
    a.frames = g->frames; a.S = g->S; a.K = g->K; a.CP = g->CP; a.D = g->D; a.nbits = g->nbits;
    a.offset = g->offset; a.seed = g->seed;
#ifdef DCCN_ABLATION
    {
        static const int abl = getenv("DCCN_GEN_ABL") ? atoi(getenv("DCCN_GEN_ABL")) : 0;      // timing experiments only
        a.abl = abl;
    }
#else
    a.abl = 0;              // (the ablation switches of tools/genbench.py exist in `make ablation` builds only)
#endif
    return DCCN_OK;
}
static int gen_static_launch(const dccn_gen_static* g, hipStream_t s, const GenChainScalars* chains = nullptr) {
    GenStaticArgs a;
    DCCN_TRY(gen_static_args(g, &a));
    const size_t smem = gen_static_smem_bytes<7, 64, 16>();
    const int blocks = ceil_div(g->frames, kGenFramesPerBlock);
    GenChainScalars gc;
    if (chains) gc = *chains;
    else memset(&gc, 0, sizeof(gc));
    if (tl_chain.G > 1 && gc.n != tl_chain.G) return DCCN_ERR_UNSUPPORTED;      // (a group needs every chain's seed / offset)
    DCCN_LAUNCH_CHAINS_Z((gen_static_frames_kernel<7, 64, 16>), dim3(blocks), dim3(256), smem, s, a, gc);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// batches whose pipelined normalisation can read the fused generator's (y, noise, partials) as its input: the single-pass R0
// (norm_fused_kernel: <= 128 * kNormFusedRPT rows) and one power partial per generator block
static bool rx_gen_next_shape_ok(int batch, int cols) {
    return kNormFusedCG == 2 && (cols % 4) == 0 && batch > 0 && batch <= 128 * kNormFusedRPT &&
           ceil_div(batch, kGenFramesPerBlock) <= kChanPartials;
}

// side != nullptr: run the dense weight-gradient branch on `side` (fork/join by events),
// concurrently with dX -> C-Conv weight gradient on the main stream.
static int rx_step_impl(const dccn_rx_shape* sh, const dccn_rx_buffers* b, bool train, dccn_adam_hparams hp,
                        hipStream_t s, hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join) {
    if (!shape_ok(sh) || !b) return DCCN_ERR_INVALID_ARG;
    const TuneScope tune(b->tuning);
    if (!b->x || !b->bits || !b->params || !b->x_norm || !b->fft_out || !b->metrics) return DCCN_ERR_INVALID_ARG;
    if (train && (!b->grads || !b->adam_m || !b->adam_v || !b->adam || !b->dz)) return DCCN_ERR_INVALID_ARG;
    if (!b->workspace || b->workspace_bytes < rx_ws_bytes(sh, train ? 1 : 0)) return DCCN_ERR_WORKSPACE;
    const RxLayout L = rx_layout(sh);
    Carver c(b->workspace, b->workspace_bytes);
    void* ws_norm = c.take<char>(L.ws_norm);
    void* ws_tail = c.take<char>(L.ws_tail);
    void* ws_dbw = train ? c.take<char>(L.ws_dense_bw) : nullptr;
    void* ws_cbw = train ? c.take<char>(L.ws_conv_bw) : nullptr;
    if (train) (void)c.take<char>(L.ws_sync);       // (reserved: keeps the workspace layout of earlier builds)
    float* P = b->params;
    float* G = b->grads;
    // step timeline (dccn_step_trace_enable): launch slots 1 C-Conv forward, 2 dense forward (+ tail), 3 tail (own launch),
    // 4 backward (fused, or grouped dX+dW), 5 C-Conv weight gradient (own launch), 6 optimizer
    const StepTraceScope trace;

    // the fused generator of the NEXT batch (dccn_rx_buffers.gen_next): first launch of the step, consumed by its last one
    // (x_next_ready set as well: the caller has issued that launch itself on ANOTHER stream -- it then overlaps the first three
    // launches of this step -- and the optimizer launch waits for the event)
    if (train && b->gen_next != nullptr) {
        if (b->gen_next->frames != sh->batch || b->gen_next->S != sh->S || 2 * (b->gen_next->K + b->gen_next->CP) * sh->S != L.cols)
            return DCCN_ERR_INVALID_ARG;
        // (the double-buffered pipelining -- R0 on the backward launch, knob 18 -- has no virtual-input form: refuse rather than
        // normalise a stale x_next)
        if (b->x_norm_next != nullptr) return DCCN_ERR_INVALID_ARG;
        // ... and everything the optimizer launch will need to form that batch is checked HERE, before the generator, the
        // forward, the backward and the update have been issued (dccn_rx_gen_next_supported is the caller's query)
        if (!rx_gen_next_shape_ok(sh->batch, L.cols) || !gen_static_ok(b->gen_next) ||
            !norm_fused_ok(b->gen_next->y, b->x_norm, sh->batch, L.cols))
            return DCCN_ERR_INVALID_ARG;
        if (b->x_next_ready == nullptr) {
            trace.launch(7);
            DCCN_TRY(gen_static_launch(b->gen_next, s));
        }
    }
    // R0 (+R8 partial sums) -- unless the previous call already normalised this batch behind its Adam update
    PowerPartials pp;
    const bool pre = train && b->x_prenormalised != 0;
    const int nslot = b->norm_slot ? 1 : 0;
    if (pre) {
        if (L.cols & 1) return DCCN_ERR_INVALID_ARG;
        norm_power_partials(sh->batch, L.cols, ws_norm, L.ws_norm, b->x_next ? b->x_next : b->x, b->x_norm, &pp, nslot);
    } else {
        DCCN_TRY(norm_impl(b->x, b->x_norm, nullptr, nullptr, b->tx_power != nullptr, &pp, sh->batch, L.cols, 1e-9f, 8.0f,
                           nullptr, hp, ws_norm, L.ws_norm, s, nslot));
    }
    // R1
    trace.launch(1);
    if (b->x_prenormalised != 0 && b->x_prenormalised != 1) return DCCN_ERR_INVALID_ARG;
    DCCN_TRY(cconv_fwd_impl(b->x_norm, P + L.o_conv_w, P + L.o_conv_b, b->fft_out, L.rows, sh->kin, sh->F, s));
    TailFinalizeArgs fin;
    trace.launch(2);
    if (dense_tail_planned(sh->nbits, train, sh->batch, L.dN) &&
        dense_tail_ok(b->fft_out, P + L.o_dense_w, sh->batch, L.dK, L.dN, sh->nbits)) {
        // R2 with R3-R6 (+ tail backward) in its epilogue; z is materialised only when the caller gave a buffer
        DCCN_TRY(dense_tail_impl(train, b->fft_out, P + L.o_dense_w, P + L.o_dense_b, b->z, b->bits, P + L.o_tail, b->prob,
                                 b->metrics, b->dz, train ? G + L.o_tail : nullptr, sh->batch, L.dK, L.dN, sh->nbits, &pp,
                                 b->tx_power, ws_tail, L.ws_tail, s, train ? &fin : nullptr));
    } else {
        if (!b->z) return DCCN_ERR_INVALID_ARG;
        // R2
        DCCN_TRY(dense_fwd_impl(b->fft_out, P + L.o_dense_w, P + L.o_dense_b, b->z, sh->batch, L.dK, L.dN, s));
        // R3-R6 (+ tail backward)
        trace.launch(3);
        DCCN_TRY(tail_impl(train, b->z, b->bits, P + L.o_tail, b->prob, b->metrics, b->dz, train ? G + L.o_tail : nullptr,
                           L.cells, sh->nbits, &pp, b->tx_power, ws_tail, L.ws_tail, s, train ? &fin : nullptr));
    }
    if (!train) return DCCN_OK;
    trace.launch(4);
    fin.adam = b->adam;                 // the optimizer's per-step bookkeeping rides on the tail finalize stage
    fin.hp = hp;

    DeferredSlabs ds;
    FoldDefer fd;
    fd.slabs = nullptr;
    OverlapStreams ovs{};
    bool overlap = false;                 // the dense kernel's update runs on ovs.side (large layers)
    // every return between the fork and the join below (a failed launch, a DCCN_TRY) still joins: the side stream never
    // keeps running behind a call that has returned, and a capture of `s` is never left with an un-joined branch
    struct OverlapJoin {
        OverlapStreams* o; hipStream_t s; bool forked = false, joined = false;
        ~OverlapJoin() {
            if (forked && !joined) {
                (void)hipEventRecord(o->join, o->side);
                (void)hipStreamWaitEvent(s, o->join, 0);
            }
        }
    } ojoin{&ovs, s};
    int fold_tilew = 0;
    const bool can_defer = L.o_conv_w == 0 && (L.o_dense_w % 4) == 0;     // optimizer kernel takes over the fold
    // small layers: dX tiles + C-Conv weight-gradient partials in their epilogue + dW items + tail finalize: one launch
    const bool fuse_bw = !side && can_defer &&
                         rx_bwd_fused_ok(sh->batch, sh->S, sh->kin, sh->F, sh->D, b->x_norm, b->fft_out, b->dz, P + L.o_dense_w);
    if (!fuse_bw && !b->dfft) return DCCN_ERR_INVALID_ARG;
    // R0 of the next batch: on leading blocks of the fused backward launch when the caller gave the second x_norm buffer
    const bool ride_bw = fuse_bw && b->x_next != nullptr && b->x_norm_next != nullptr && kNormFusedCG == 2 &&
                         norm_fused_ok(b->x_next, b->x_norm_next, sh->batch, L.cols);
    if (b->x_norm_next != nullptr && b->x_next != nullptr && !ride_bw) return DCCN_ERR_INVALID_ARG;   // ask dccn_rx_norm_rides_backward first
    // a producer on another stream is filling x_next: the launch that reads it waits for the producer's event
    const bool wait_x = (b->x_next != nullptr || b->gen_next != nullptr) && b->x_next_ready != nullptr;
    if (fuse_bw) {
        NormRideArgs nr;
        memset(&nr, 0, sizeof(nr));
        if (ride_bw && wait_x) DCCN_HIP(hipStreamWaitEvent(s, (hipEvent_t)b->x_next_ready, 0));
        if (ride_bw) {
            PowerPartials np;
            norm_power_partials(sh->batch, L.cols, ws_norm, L.ws_norm, b->x_next, b->x_norm_next, &np, nslot ^ 1);
            nr.x = b->x_next; nr.y = b->x_norm_next; nr.power = b->tx_power ? const_cast<double*>(np.partial) : nullptr;
            nr.batch = sh->batch; nr.cols = L.cols; nr.blocks = norm_fused_blocks(L.cols);
            nr.eps = 1e-9f; nr.peak = 8.0f;
            nr.trail = g_tune[TUNE_NORM_ON_BWD] >= 2 ? 1 : 0;
        }
        DCCN_TRY(rx_bwd_fused_impl(b->x_norm, b->fft_out, b->dz, P + L.o_dense_w, b->dfft, G + L.o_dense_b, sh->batch, sh->S,
                                   sh->kin, sh->F, sh->D, ws_dbw, L.ws_dense_bw, ws_cbw, L.ws_conv_bw, nr, fin, hp, s, &ds, &fd,
                                   &fold_tilew));
    } else if (side) {
        // two-stream variant: dense dW/db on `side`, dX -> C-Conv dW on the main stream
        DCCN_HIP(hipEventRecord(ev_fork, s));
        DCCN_HIP(hipStreamWaitEvent(side, ev_fork, 0));
        DCCN_TRY(dense_bwd_w_impl(b->fft_out, b->dz, G + L.o_dense_w, G + L.o_dense_b, sh->batch, L.dK, L.dN, ws_dbw,
                                  L.ws_dense_bw, side, &ds));
        DCCN_HIP(hipEventRecord(ev_join, side));
        DCCN_TRY(dense_bwd_x_impl(b->dz, P + L.o_dense_w, b->dfft, sh->batch, L.dK, L.dN, s));
    } else {
        // default: dense dX and dW/db in one grouped launch (independent GEMMs packed on the same grid)
        // Large layers whose dW tiles are unsplit (N = 1024: 585 rows are one k range): the dense kernel's Adam update runs as a
        // launch of its own on the library's second stream next to the C-Conv weight-gradient launch.  It needs this step's
        // alpha and BER gate, so the tail's slab reduction (which otherwise rides on the C-Conv weight-gradient launch further
        // down) runs first, as a launch of its own.
        {
            const SplitPlan sp = dense_dw_plan(sh->batch, L.dK, L.dN);
            const long long bigw = (long long)ceil_div(L.dK, 128) * ceil_div(L.dN, 128);
            // round 4: the same layers, the update as a launch of its own on the library's second stream, next to the C-Conv
            // weight-gradient launch (same precondition: alpha and the BER gate must exist before it starts)
            if (g_tune[TUNE_ADAM_OVERLAP] && sp.splits == 1 && bigw >= 2 * kCUs && can_defer &&
                (L.o_dense_w % 4) == 0 && (((long long)L.dK * L.dN) % 4) == 0 && overlap_streams(&ovs)) {
                hipLaunchKernelGGL(demod_tail_finalize_kernel, dim3(tail_finalize_blocks(fin.P)), dim3(256), 0, s, fin);
                DCCN_LAUNCH_CHECK();
                fin.metrics = nullptr;
                overlap = true;
            }
        }
        DCCN_TRY(dense_bwd_grouped_impl(b->fft_out, b->dz, P + L.o_dense_w, b->dfft, G + L.o_dense_w, G + L.o_dense_b,
                                        sh->batch, L.dK, L.dN, ws_dbw, L.ws_dense_bw, s, &ds));
        if (overlap && ds.dw_slabs != nullptr) overlap = false;
        if (overlap) {
            // the optimizer kernel itself, restricted to the dense kernel's segment: same arithmetic, same results
            AdamRxArgs as;
            memset(&as, 0, sizeof(as));
            as.param = P; as.grad = G; as.m = b->adam_m; as.v = b->adam_v;
            as.reg_coef = b->reg_coef; as.reg_gate = b->reg_coef ? &b->metrics->berlin : nullptr;
            as.state = b->adam;
            as.o_dw = L.o_dense_w; as.n_dw = (long long)L.dK * L.dN; as.o_db = L.o_dense_b; as.n_db = L.dN;
            as.n = as.o_dw + as.n_dw;
            as.skip_lo = 0; as.skip_hi = as.o_dw;                 // (everything in front of the dense kernel stays with the main launch)
            as.splits = 1; as.neps = 1e-9f; as.npeak = 8.0f;
            as.reg_uniform_dw = b->reg_uniform_dense != 0 ? 1 : 0;
            as.nt = g_tune[TUNE_ADAM_OVERLAP] >= 2 ? 2 : 0;
            long long sb = ceil_div_ll(ceil_div_ll(as.n, 4), 256);
            if (sb > 8 * kCUs) sb = 8 * kCUs;                     // (2, 4, 16 per CU measured within 0.5 % of this)
            DCCN_HIP(hipEventRecord(ovs.fork, s));
            DCCN_HIP(hipStreamWaitEvent(ovs.side, ovs.fork, 0));
            ojoin.forked = true;
            hipLaunchKernelGGL(adam_rx_kernel<0>, dim3((unsigned)sb), dim3(256), 0, ovs.side, as, hp);
            DCCN_LAUNCH_CHECK();
            DCCN_HIP(hipEventRecord(ovs.join, ovs.side));
        }
    }
    // C-Conv dW/db from dX (the C-Conv input is data: no dX of its own)
    // (its fold launch also carries the tail's slab reduction: metrics, tail gradients, tx_power)
    trace.launch(5);
    if (!fuse_bw)
        DCCN_TRY(cconv_bwd_w_impl(b->x_norm, b->dfft, G + L.o_conv_w, G + L.o_conv_b, L.rows, sh->kin, sh->F, ws_cbw,
                                  L.ws_conv_bw, s, &fin, can_defer ? &fd : nullptr));
    if (side) DCCN_HIP(hipStreamWaitEvent(s, ev_join, 0));
    // (overlap: the dense kernel's update on the second stream is joined at the END of the call -- the optimizer launch below
    // leaves that segment alone (skip_lo / skip_hi), reads the same read-only step state)
    // R7 (+ BER-gated L2 term of R6), fused with the split-K reduction of the dense gradient
    if (wait_x && !ride_bw) DCCN_HIP(hipStreamWaitEvent(s, (hipEvent_t)b->x_next_ready, 0));
    trace.launch(6);
    AdamRxArgs aa;
    aa.stamp = tl_stamp;
    aa.param = P; aa.grad = G; aa.m = b->adam_m; aa.v = b->adam_v;
    aa.reg_coef = b->reg_coef; aa.reg_gate = b->reg_coef ? &b->metrics->berlin : nullptr;
    aa.state = b->adam; aa.n = L.total;
    aa.dw_slabs = ds.dw_slabs; aa.db_slabs = ds.db_slabs; aa.splits = ds.splits;
    aa.o_dw = L.o_dense_w; aa.n_dw = (long long)L.dK * L.dN; aa.o_db = L.o_dense_b; aa.n_db = L.dN;
    aa.cw_slabs = fd.slabs; aa.cw_colsum = fd.slabs ? fd.colsum : nullptr;
    aa.cw_splits = fd.slabs ? fd.splits : 0; aa.cw_slab = fd.slabs ? fd.slab : 0;
    aa.kin = sh->kin; aa.F = sh->F; aa.o_cw = L.o_conv_w; aa.cw_tilew = fd.slabs ? fold_tilew : 0;
    aa.n_conv = L.o_dense_w;                      // C-Conv kernel + bias come first in the arena
    aa.skip_lo = aa.skip_hi = 0;
    aa.reg_uniform_dw = b->reg_uniform_dense != 0 ? 1 : 0;
    aa.skip_dw_grad = b->keep_dense_grad < 0 ? 1 : 0;          // the caller never reads the summed dense gradient
    aa.nt = 0;
    if (overlap) { aa.skip_lo = L.o_dense_w; aa.skip_hi = L.o_dense_w + (long long)L.dK * L.dN; }
    aa.fold_blocks = fd.slabs ? ceil_div(sh->kin * sh->F + sh->F, fold_tilew > 0 ? kFoldLanesTiled : kRedLanes) : 0;
    long long blocks = ceil_div_ll(ceil_div_ll(L.total - (aa.fold_blocks ? aa.n_conv : 0), 4), 256);
    if (blocks > 8 * kCUs) blocks = 8 * kCUs;
    blocks += aa.fold_blocks;
    // R0 of the next batch on the leading blocks of this launch (dccn_rx_buffers.x_next)
    aa.nx = nullptr; aa.ny = nullptr; aa.npower = nullptr; aa.nbatch = 0; aa.ncols = 0; aa.norm_blocks = 0;
    aa.neps = 1e-9f; aa.npeak = 8.0f;
    aa.nv = norm_virtual_none();
    // gen_next: the next batch is produced by the fused generator launch issued at the top of this step; R0 reads it as
    // (y, noise, power partials) -- its virtual input -- instead of a materialised x_next (x_next, when given too, receives x)
    const dccn_gen_static* gen = (!ride_bw && train) ? b->gen_next : nullptr;
    const float* rin = gen ? gen->y : b->x_next;
    const bool ride = !ride_bw && rin != nullptr && kNormFusedCG == 2 && norm_fused_ok(rin, b->x_norm, sh->batch, L.cols);
    if (gen && !ride) return DCCN_ERR_INVALID_ARG;
    if (ride) {
        PowerPartials np;
        norm_power_partials(sh->batch, L.cols, ws_norm, L.ws_norm, rin, b->x_norm, &np, nslot);
        aa.nx = rin; aa.ny = b->x_norm; aa.npower = b->tx_power ? const_cast<double*>(np.partial) : nullptr;
        aa.nbatch = sh->batch; aa.ncols = L.cols; aa.norm_blocks = norm_fused_blocks(L.cols);
        blocks += aa.norm_blocks;
        if (gen) {
            aa.nv.y = gen->y; aa.nv.noise = gen->noise; aa.nv.ppart = gen->power_partial;
            aa.nv.npart = ceil_div(gen->frames, kGenFramesPerBlock);
            aa.nv.total = (double)gen->frames * (double)(gen->S * (gen->K + gen->CP));
            aa.nv.x_out = const_cast<float*>(b->x_next);
            aa.nv.npart_noise = gen->noise_partial; aa.nv.n_noise = aa.nv.npart;
            aa.nv.npow_out = gen->noise_partial ? gen->noise_power_out : nullptr;
        }
    }
    switch (ds.splits) {
        case 2: hipLaunchKernelGGL(adam_rx_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, aa, hp); break;
        case 3: hipLaunchKernelGGL(adam_rx_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, s, aa, hp); break;
        case 4: hipLaunchKernelGGL(adam_rx_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, aa, hp); break;
        case 5: hipLaunchKernelGGL(adam_rx_kernel<5>, dim3((unsigned)blocks), dim3(256), 0, s, aa, hp); break;
        case 6: hipLaunchKernelGGL(adam_rx_kernel<6>, dim3((unsigned)blocks), dim3(256), 0, s, aa, hp); break;
        default: hipLaunchKernelGGL(adam_rx_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, aa, hp); break;
    }
    DCCN_LAUNCH_CHECK();
    trace.none();
    if (b->x_next != nullptr && !ride && !ride_bw) {
        // shapes the single-pass kernel does not take: the same normalisation as launches of their own
        PowerPartials np;
        DCCN_TRY(norm_impl(b->x_next, b->x_norm, nullptr, nullptr, b->tx_power != nullptr, &np, sh->batch, L.cols, 1e-9f,
                           8.0f, nullptr, hp, ws_norm, L.ws_norm, s, nslot));
    }
    if (overlap) { ojoin.joined = true; DCCN_HIP(hipStreamWaitEvent(s, ovs.join, 0)); }
    return DCCN_OK;
}

static int eq_monitor_blocks(int B, int K) {
    long long n = ceil_div_ll((long long)B * K * 2, 256);
    if (n > 256) n = 256;
    return (int)(n < 1 ? 1 : n);
}
#include "eq_step.h"

}  // namespace dccn

using namespace dccn;

struct dccn_rx_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
    hipStream_t cap;     // capture happens on a private stream: the caller's may be the (uncapturable) null stream
    hipStream_t side;
    hipEvent_t ev_fork, ev_join;
};

struct dccn_timer {
    hipEvent_t a, b;
};

extern "C" {

const char* dccn_strerror(int status) {
    switch (status) {
        case DCCN_OK: return "ok";
        case DCCN_ERR_INVALID_ARG: return "invalid argument";
        case DCCN_ERR_WORKSPACE: return "workspace missing or too small";
        case DCCN_ERR_LAUNCH: return "HIP launch/runtime error";
        case DCCN_ERR_NO_DEVICE: return "no HIP device visible";
        case DCCN_ERR_STATE: return "object used in the wrong state";
        case DCCN_ERR_UNSUPPORTED: return "a grouped call reached a launch that cannot carry several chains";
        default: return "unknown status";
    }
}

int dccn_version(void) { return 100; }
#ifndef DCCN_BUILD_ID
#define DCCN_BUILD_ID "unknown"
#endif
const char* dccn_build_id(void) { return DCCN_BUILD_ID; }
int dccn_last_hip_error(void) { return g_last_hip_error; }

int dccn_device_info(int* cu_count, int* wavefront, size_t* hbm_bytes, char* arch, int arch_len) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return DCCN_ERR_NO_DEVICE;
    int dev = 0;
    DCCN_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    DCCN_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (wavefront) *wavefront = prop.warpSize;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return DCCN_OK;
}

size_t dccn_batch_moment_norm_workspace_size(int batch, int cols) {
    if (batch <= 0 || cols <= 0) return 0;
    return norm_ws_bytes(batch, cols);
}
int dccn_batch_moment_norm_fwd(const float* x, float* y, float* mean, float* var, int batch, int cols, float eps,
                               void* workspace, size_t workspace_bytes, dccn_stream_t stream) {
    dccn_adam_hparams hp;
    memset(&hp, 0, sizeof(hp));
    return norm_impl(x, y, mean, var, false, nullptr, batch, cols, eps, 8.0f, nullptr, hp, workspace, workspace_bytes,
                     (hipStream_t)stream);
}

size_t dccn_clip_power_workspace_size(long long n_pairs) {
    (void)n_pairs;
    return (size_t)4 * kCUs * sizeof(double);
}
int dccn_clip_power(const float* x, float* y, float* power_out, long long n_pairs, float peak, void* workspace,
                    size_t workspace_bytes, dccn_stream_t stream) {
    if (!x || !power_out || n_pairs <= 0) return DCCN_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < dccn_clip_power_workspace_size(n_pairs)) return DCCN_ERR_WORKSPACE;
    long long blocks = ceil_div_ll(n_pairs, 256);
    if (blocks > 4 * kCUs) blocks = 4 * kCUs;
    double* partial = static_cast<double*>(workspace);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(clip_power_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, y, n_pairs, peak, partial);
    DCCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, partial, (int)blocks, (double)n_pairs,
                       power_out);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

int dccn_cconv_gemm_fwd(const float* x, const float* w, const float* bias, float* out, int rows, int kin, int F,
                        dccn_stream_t stream) {
    return cconv_fwd_impl(x, w, bias, out, rows, kin, F, (hipStream_t)stream);
}
size_t dccn_cconv_gemm_bwd_w_workspace_size(int rows, int kin, int F) {
    if (rows <= 0 || kin <= 0 || F <= 0) return 0;
    return cconv_bw_ws_bytes(rows, kin, F);
}
int dccn_cconv_gemm_bwd_w(const float* x, const float* dout, float* dw, float* dbias, int rows, int kin, int F,
                          void* workspace, size_t workspace_bytes, dccn_stream_t stream) {
    return cconv_bwd_w_impl(x, dout, dw, dbias, rows, kin, F, workspace, workspace_bytes, (hipStream_t)stream);
}
int dccn_cconv_gemm_bwd_x(const float* dout, const float* w, float* dx, int rows, int kin, int F,
                          dccn_stream_t stream) {
    return cconv_bwd_x_impl(dout, w, dx, rows, kin, F, (hipStream_t)stream);
}

int dccn_dense_fwd(const float* x, const float* w, const float* bias, float* y, int M, int K, int N,
                   dccn_stream_t stream) {
    return dense_fwd_impl(x, w, bias, y, M, K, N, (hipStream_t)stream);
}
int dccn_dense_bwd_x(const float* dy, const float* w, float* dx, int M, int K, int N, dccn_stream_t stream) {
    return dense_bwd_x_impl(dy, w, dx, M, K, N, (hipStream_t)stream);
}
size_t dccn_dense_bwd_w_workspace_size(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    return splitk_ws_bytes(K, N, M);
}
int dccn_dense_bwd_w(const float* x, const float* dy, float* dw, float* dbias, int M, int K, int N, void* workspace,
                     size_t workspace_bytes, dccn_stream_t stream) {
    return dense_bwd_w_impl(x, dy, dw, dbias, M, K, N, workspace, workspace_bytes, (hipStream_t)stream);
}

int dccn_dense_bwd(const float* x, const float* dy, const float* w, float* dx, float* dw, float* dbias, int M, int K,
                   int N, void* workspace, size_t workspace_bytes, dccn_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DeferredSlabs ds;
    DCCN_TRY(dense_bwd_grouped_impl(x, dy, w, dx, dw, dbias, M, K, N, workspace, workspace_bytes, s, &ds));
    if (ds.dw_slabs) {
        const long long n = (long long)K * N;
        DCCN_TRY(launch_splitk_reduce(ds.dw_slabs, ds.splits, n, dw, n, s));
        if (dbias && ds.db_slabs) DCCN_TRY(launch_splitk_reduce(ds.db_slabs, ds.splits, (long long)N, dbias, (long long)N, s));
    }
    return DCCN_OK;
}

int dccn_dense_bwd_slabs(const float* x, const float* dy, const float* w, float* dx, float* dw, float* dbias, int M,
                         int K, int N, void* workspace, size_t workspace_bytes, int* splits, dccn_stream_t stream) {
    DeferredSlabs ds;
    DCCN_TRY(dense_bwd_grouped_impl(x, dy, w, dx, dw, dbias, M, K, N, workspace, workspace_bytes, (hipStream_t)stream,
                                    &ds));
    if (splits) *splits = ds.dw_slabs ? ds.splits : 1;
    return DCCN_OK;
}

// row[0..3] += conf, row[4] += ce_sum, row[5] += count  (one sweep-table row)
__global__ void metrics_table_add_kernel(const dccn_metrics* __restrict__ m, double* __restrict__ row) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int k = 0; k < 4; ++k) row[k] += (double)m->conf[k];
        row[4] += m->ce_sum;
        row[5] += (double)m->count;
    }
}
int dccn_metrics_table_add(const dccn_metrics* metrics, double* row6, dccn_stream_t stream) {
    if (!metrics || !row6) return DCCN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(metrics_table_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, metrics, row6);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
// the training loop's per-step monitors (ofdmreceiver_np.py:222-229 fetches ce_mean, tx_power and the noise power of every step
// and averages them per epoch): acc3 += {ce_mean, tx_power, noise_power} in ONE single-thread launch instead of three framework
// launches -- same float32 additions in the same order
__global__ void step_monitor_add_kernel(const dccn_metrics* __restrict__ m, const float* __restrict__ tx_power,
                                        const float* __restrict__ noise_power, float* __restrict__ acc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        acc[0] += m->ce_mean;
        if (tx_power) acc[1] += tx_power[0];
        if (noise_power) acc[2] += noise_power[0];
    }
}
int dccn_step_monitor_add(const dccn_metrics* metrics, const float* tx_power, const float* noise_power, float* acc3,
                          dccn_stream_t stream) {
    if (!metrics || !acc3) return DCCN_ERR_INVALID_ARG;
    DCCN_NO_CHAINS();
    hipLaunchKernelGGL(step_monitor_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, metrics, tx_power, noise_power, acc3);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
// row = the record (a one-point table: no clearing launch in front of it)
__global__ void metrics_table_set_kernel(const dccn_metrics* __restrict__ m, double* __restrict__ row) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int k = 0; k < 4; ++k) row[k] = (double)m->conf[k];
        row[4] = m->ce_sum;
        row[5] = (double)m->count;
    }
}
int dccn_metrics_table_set(const dccn_metrics* metrics, double* row6, dccn_stream_t stream) {
    if (!metrics || !row6) return DCCN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(metrics_table_set_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, metrics, row6);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

static bool tune_key_live(int key) {
    return key >= 0 && key < TUNE_COUNT && key != 15 && key != 16 && key != 22 && key != 23 && key != 26;
}
int dccn_set_tuning(int key, int value) {
    if (!tune_key_live(key) || value < 0) return DCCN_ERR_INVALID_ARG;
    g_tune.set(key, value);
    if (key == TUNE_WHOLE_K) g_whole_k_global.store(value, std::memory_order_relaxed);
    return DCCN_OK;
}
int dccn_get_tuning(int key) { return (key < 0 || key >= TUNE_COUNT) ? DCCN_ERR_INVALID_ARG : g_tune.global(key); }
int dccn_tuning_count(void) { return TUNE_COUNT; }
int dccn_tuning_snapshot(int* table, int n) {
    if (!table || n < TUNE_COUNT) return DCCN_ERR_INVALID_ARG;
    for (int k = 0; k < TUNE_COUNT; ++k) table[k] = g_tune.global(k);
    return TUNE_COUNT;
}

size_t dccn_rx_backward_workspace_size(int batch, int S, int kin, int F, int D) {
    if (batch <= 0 || S <= 0 || kin <= 0 || F <= 0 || D <= 0) return 0;
    size_t o = 0;
    o = carve_size(o, splitk_ws_bytes(S * 2 * F, 2 * D, batch));
    o = carve_size(o, rx_bwd_fused_ws_bytes(batch, S, kin, F));
    return align_up(o, 256);
}
int dccn_rx_backward(const float* x_norm, const float* fft_out, const float* dz, const float* w_dense, float* dfft,
                     float* dw_dense, float* db_dense, float* dw_conv, float* db_conv, int batch, int S, int kin, int F,
                     int D, int reduce, void* workspace, size_t workspace_bytes, dccn_stream_t stream) {
    if (!x_norm || !fft_out || !dz || !w_dense || batch <= 0 || S <= 0 || kin <= 0 || F <= 0 || D <= 0)
        return DCCN_ERR_INVALID_ARG;
    if (reduce && (!dw_dense || !dw_conv)) return DCCN_ERR_INVALID_ARG;
    if (!rx_bwd_fused_ok(batch, S, kin, F, D, x_norm, fft_out, dz, w_dense)) return DCCN_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < dccn_rx_backward_workspace_size(batch, S, kin, F, D)) return DCCN_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int dK = S * 2 * F, dN = 2 * D;
    const size_t nd = splitk_ws_bytes(dK, dN, batch), nc = rx_bwd_fused_ws_bytes(batch, S, kin, F);
    Carver c(workspace, workspace_bytes);
    void* ws_d = c.take<char>(nd);
    void* ws_c = c.take<char>(nc);
    NormRideArgs nr;
    memset(&nr, 0, sizeof(nr));
    TailFinalizeArgs fin;
    memset(&fin, 0, sizeof(fin));
    dccn_adam_hparams hp;
    memset(&hp, 0, sizeof(hp));
    DeferredSlabs ds;
    FoldDefer fd;
    int tilew = 0;
    DCCN_TRY(rx_bwd_fused_impl(x_norm, fft_out, dz, w_dense, dfft, db_dense ? db_dense : dw_dense, batch, S, kin, F, D, ws_d, nd,
                               ws_c, nc, nr, fin, hp, s, &ds, &fd, &tilew));
    if (!reduce) return DCCN_OK;
    const long long n = (long long)dK * dN;
    if (db_dense) DCCN_TRY(launch_splitk_reduce2(ds.dw_slabs, ds.splits, n, dw_dense, n, ds.db_slabs, (long long)dN, db_dense, (long long)dN, s));
    else DCCN_TRY(launch_splitk_reduce(ds.dw_slabs, ds.splits, n, dw_dense, n, s));
    const int fold_blocks = ceil_div(kin * F + F, tilew > 0 ? kFoldLanesTiled : kRedLanes);
    hipLaunchKernelGGL(cconv_fold_kernel, dim3(fold_blocks), dim3(256), 0, s, fd.slabs, fd.splits, fd.slab, fd.colsum, dw_conv,
                       db_conv, kin, F, tilew);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

size_t dccn_step_trace_bytes(int ring_steps) {
    if (ring_steps <= 0) return 0;
    return (size_t)ring_steps * kStampLaunches * kStampBlocks * kStampWords * sizeof(unsigned long long);
}
int dccn_step_trace_enable(unsigned long long* buf, size_t bytes, int ring_steps) {
    if (buf == nullptr) {
        g_step_trace.buf.store(nullptr);
        g_step_trace.ring.store(0);
        return DCCN_OK;
    }
    if (ring_steps <= 0 || bytes < dccn_step_trace_bytes(ring_steps)) return DCCN_ERR_WORKSPACE;
    g_step_trace.ring.store(ring_steps);
    g_step_trace.step.store(0);
    g_step_trace.buf.store(buf);
    return DCCN_OK;
}
long long dccn_step_trace_steps(void) { return g_step_trace.step.load(); }
void dccn_step_trace_geometry(int* launches, int* blocks, int* words) {
    if (launches) *launches = kStampLaunches;
    if (blocks) *blocks = kStampBlocks;
    if (words) *words = kStampWords;
}

#ifdef DCCN_TRACE
/* trace build only (tools/blocktrace.py): where the instrumented kernels leave their per-block time stamps */
int dccn_debug_set_trace(unsigned long long* buf) {
    DCCN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(buf)));
    return DCCN_OK;
}
#endif

int dccn_dense_tail_supported(int M, int K, int N, int nbits) { return dense_tail_shape_ok(M, K, N, nbits) ? 1 : 0; }
int dccn_rx_dense_tail_fused(const dccn_rx_shape* sh, int train) {
    if (!shape_ok(sh)) return 0;
    return dense_tail_planned(sh->nbits, train != 0, sh->batch, 2 * sh->D) &&
           dense_tail_shape_ok(sh->batch, sh->S * 2 * sh->F, 2 * sh->D, sh->nbits) ? 1 : 0;
}
int dccn_rx_norm_rides_backward(const dccn_rx_shape* sh) {
    if (!shape_ok(sh)) return 0;
    const int cols = sh->S * sh->kin * 2;
    const bool ok = g_tune[TUNE_NORM_ON_BWD] && rx_bwd_fused_ok(sh->batch, sh->S, sh->kin, sh->F, sh->D, nullptr, nullptr, nullptr, nullptr) &&
                    kNormFusedCG == 2 && norm_fused_ok(nullptr, nullptr, sh->batch, cols);
    return ok ? 1 : 0;
}
int dccn_rx_gen_next_supported(const dccn_rx_shape* sh) {
    if (!shape_ok(sh) || dccn_rx_norm_rides_backward(sh) != 0) return 0;
    return rx_gen_next_shape_ok(sh->batch, sh->S * sh->kin * 2) ? 1 : 0;
}
int dccn_rx_bwd_fused_supported(const dccn_rx_shape* sh) {
    if (!shape_ok(sh)) return 0;
    // (pointer alignment is checked again at launch time; the query assumes 16-byte aligned buffers)
    return rx_bwd_fused_ok(sh->batch, sh->S, sh->kin, sh->F, sh->D, nullptr, nullptr, nullptr, nullptr) ? 1 : 0;
}

size_t dccn_dense_tail_workspace_size(int M, int N, int nbits) {
    if (M <= 0 || N <= 0 || nbits < 1 || nbits > 4) return 0;
    return dense_tail_ws_bytes(M, N, nbits);
}
int dccn_dense_tail_fwd(const float* x, const float* w, const float* bias, float* z, const int32_t* bits,
                        const float* tailp, float* prob, dccn_metrics* metrics, int M, int K, int N, int nbits,
                        void* workspace, size_t workspace_bytes, dccn_stream_t stream) {
    return dense_tail_impl(false, x, w, bias, z, bits, tailp, prob, metrics, nullptr, nullptr, M, K, N, nbits, nullptr,
                           nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}
int dccn_dense_tail_fwd_bwd(const float* x, const float* w, const float* bias, float* z, const int32_t* bits,
                            const float* tailp, float* prob, dccn_metrics* metrics, float* dz, float* dtailp, int M,
                            int K, int N, int nbits, void* workspace, size_t workspace_bytes, dccn_stream_t stream) {
    return dense_tail_impl(true, x, w, bias, z, bits, tailp, prob, metrics, dz, dtailp, M, K, N, nbits, nullptr, nullptr,
                           workspace, workspace_bytes, (hipStream_t)stream);
}

int dccn_tail_param_count(int nbits) {
    if (nbits < 1 || nbits > 4) return DCCN_ERR_INVALID_ARG;
    return tail_param_count(nbits);
}
size_t dccn_demod_tail_workspace_size(long long cells, int nbits) {
    if (cells <= 0 || nbits < 1 || nbits > 4) return 0;
    return tail_ws_bytes(cells, nbits);
}
int dccn_demod_tail_loss_fwd(const float* z, const int32_t* bits, const float* tailp, float* prob,
                             dccn_metrics* metrics, long long cells, int nbits, void* workspace,
                             size_t workspace_bytes, dccn_stream_t stream) {
    return tail_impl(false, z, bits, tailp, prob, metrics, nullptr, nullptr, cells, nbits, nullptr, nullptr, workspace,
                     workspace_bytes, (hipStream_t)stream);
}
int dccn_demod_tail_loss_fwd_bwd(const float* z, const int32_t* bits, const float* tailp, float* prob,
                                 dccn_metrics* metrics, float* dz, float* dtailp, long long cells, int nbits,
                                 void* workspace, size_t workspace_bytes, dccn_stream_t stream) {
    return tail_impl(true, z, bits, tailp, prob, metrics, dz, dtailp, cells, nbits, nullptr, nullptr, workspace,
                     workspace_bytes, (hipStream_t)stream);
}

int dccn_adam_tf_step(float* param, const float* grad, float* m, float* v, const float* reg_coef,
                      const float* reg_gate, dccn_adam_state* state, dccn_adam_hparams hp, long long n,
                      dccn_stream_t stream) {
    return adam_impl(param, grad, m, v, reg_coef, reg_gate, state, hp, n, (hipStream_t)stream);
}

int dccn_rx_param_offsets(const dccn_rx_shape* shape, long long offsets[6]) {
    if (!shape_ok(shape) || !offsets) return DCCN_ERR_INVALID_ARG;
    const RxLayout L = rx_layout(shape);
    offsets[0] = L.o_conv_w; offsets[1] = L.o_conv_b; offsets[2] = L.o_dense_w;
    offsets[3] = L.o_dense_b; offsets[4] = L.o_tail; offsets[5] = L.total;
    return DCCN_OK;
}
size_t dccn_rx_workspace_size(const dccn_rx_shape* shape, int train) {
    // the library's second stream and the calling thread's event pair are created here, i.e. before any step and outside
    // any stream capture (every caller sizes its workspace first); rx_step_impl only looks them up
    if (train && shape_ok(shape) && g_tune[TUNE_ADAM_OVERLAP]) { OverlapStreams o; (void)overlap_streams(&o); }
    if (!shape_ok(shape)) return 0;
    return rx_ws_bytes(shape, train);
}
int dccn_rx_eval_step(const dccn_rx_shape* shape, const dccn_rx_buffers* buf, dccn_stream_t stream) {
    dccn_adam_hparams hp;
    memset(&hp, 0, sizeof(hp));
    return rx_step_impl(shape, buf, false, hp, (hipStream_t)stream, nullptr, nullptr, nullptr);
}
int dccn_rx_train_step(const dccn_rx_shape* shape, const dccn_rx_buffers* buf, dccn_adam_hparams hp,
                       dccn_stream_t stream) {
    return rx_step_impl(shape, buf, true, hp, (hipStream_t)stream, nullptr, nullptr, nullptr);
}
// R0 (+R8 partial sums) of buf->x into buf->x_norm exactly as a step would run it: primes the pipelined mode
// (dccn_rx_buffers.x_prenormalised) before the first call
int dccn_rx_normalise(const dccn_rx_shape* shape, const dccn_rx_buffers* buf, dccn_stream_t stream) {
    if (!shape_ok(shape) || !buf || !buf->x || !buf->x_norm) return DCCN_ERR_INVALID_ARG;
    if (!buf->workspace || buf->workspace_bytes < rx_ws_bytes(shape, 1)) return DCCN_ERR_WORKSPACE;
    const RxLayout L = rx_layout(shape);
    Carver c(buf->workspace, buf->workspace_bytes);
    void* ws_norm = c.take<char>(L.ws_norm);
    PowerPartials pp;
    dccn_adam_hparams hp;
    memset(&hp, 0, sizeof(hp));
    return norm_impl(buf->x, buf->x_norm, nullptr, nullptr, buf->tx_power != nullptr, &pp, shape->batch, L.cols, 1e-9f,
                     8.0f, nullptr, hp, ws_norm, L.ws_norm, (hipStream_t)stream, buf->norm_slot ? 1 : 0);
}

// mode: bit0 = train, bit1 = fork the dense weight-gradient branch onto a second stream
int dccn_rx_graph_create(const dccn_rx_shape* shape, const dccn_rx_buffers* buf, int mode, dccn_adam_hparams hp,
                         dccn_stream_t stream, dccn_rx_graph** out) {
    if (!out || !shape_ok(shape) || !buf) return DCCN_ERR_INVALID_ARG;
    const bool train = mode & 1, fork = (mode & 2) && train;
    (void)stream;
    dccn_rx_graph* g = new dccn_rx_graph();
    memset(g, 0, sizeof(*g));
    if (hipStreamCreateWithFlags(&g->cap, hipStreamNonBlocking) != hipSuccess) {
        dccn_rx_graph_destroy(g);
        return DCCN_ERR_LAUNCH;
    }
    hipStream_t s = g->cap;
    if (fork) {
        if (hipStreamCreateWithFlags(&g->side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_join, hipEventDisableTiming) != hipSuccess) {
            dccn_rx_graph_destroy(g);
            return DCCN_ERR_LAUNCH;
        }
    }
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) {
        dccn_rx_graph_destroy(g);
        return hip_fail(e);
    }
    const int st = rx_step_impl(shape, buf, train, hp, s, fork ? g->side : nullptr, g->ev_fork, g->ev_join);
    e = hipStreamEndCapture(s, &g->graph);
    if (st != DCCN_OK || e != hipSuccess) {
        dccn_rx_graph_destroy(g);
        return st != DCCN_OK ? st : hip_fail(e);
    }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        dccn_rx_graph_destroy(g);
        return hip_fail(e);
    }
    *out = g;
    return DCCN_OK;
}
// ---- fused equaliser step ---------------------------------------------------------------------------
int dccn_eq_param_offsets(const dccn_eq_shape* shape, long long* offsets) {
    if (!eq_shape_ok(shape) || !offsets) return DCCN_ERR_INVALID_ARG;
    const EqDims d = eq_dims(shape);
    for (int i = 0; i < 21; ++i) offsets[i] = d.o[i];
    return DCCN_OK;
}
size_t dccn_eq_workspace_size(const dccn_eq_shape* shape, int train) {
    if (!eq_shape_ok(shape)) return 0;
    return eq_ws_bytes(shape, train);
}
int dccn_eq_workspace_tensor(const dccn_eq_shape* shape, int train, const char* name, size_t* byte_offset, size_t* count) {
    if (!eq_shape_ok(shape) || !name || !byte_offset || !count) return DCCN_ERR_INVALID_ARG;
    const EqDims d = eq_dims(shape);
    // carve a dummy base so that every pointer is base + offset (the Carver hands out nullptr for a null base)
    char* const base = reinterpret_cast<char*>(static_cast<uintptr_t>(1) << 40);
    Carver c(base, ~static_cast<size_t>(0) >> 2);
    EqWs w;
    memset(&w, 0, sizeof(w));
    eq_carve(c, shape, d, train != 0, w);
    const size_t B = d.B, R = d.R, SK2 = d.SK2, K2 = 2 * (size_t)d.K, N2 = 2 * (size_t)d.nsc;
    struct Ent { const char* n; const float* p; size_t cnt; bool tr; };
    const Ent tab[] = {
        {"x_norm", w.x_norm, R * N2, false}, {"ln", w.ln, R * N2, false}, {"t1", w.t1, R * K2, false}, {"y", w.y, R * K2, false},
        {"d1", w.d1, B * d.Pp, false}, {"d2", w.d2, B * SK2, false}, {"d3", w.d3, B * SK2, false}, {"d4", w.d4, B * SK2, false},
        {"T", w.T, SK2 * SK2, false}, {"be", w.be, SK2, false}, {"eq", w.eq, B * SK2, false}, {"corr", w.corr, B * SK2, false},
        {"eqc", w.eqc, R * K2, false}, {"corc", w.corc, R * K2, false}, {"cat", w.cat, R * 2 * K2, false},
        {"fft", w.fft, R * 2 * (size_t)d.F, false}, {"z", w.z, B * 2 * (size_t)d.D, false},
        {"dz", w.dz, B * 2 * (size_t)d.D, true}, {"dfft", w.dfft, R * 2 * (size_t)d.F, true}, {"dout", w.dout, R * N2, true},
        {"dcat", w.dcat, R * 2 * K2, true}, {"deqc", w.deqc, R * K2, true}, {"dcorc", w.dcorc, R * K2, true},
        {"deq", w.deq, B * SK2, true}, {"dcorr", w.dcorr, B * SK2, true}, {"dy", w.dy, B * SK2, true}, {"dh", w.dh, B * SK2, true},
        {"dT", w.dT, SK2 * SK2, true}, {"dbe", w.dbe, SK2, true}, {"dd4", w.dd4, B * SK2, true}, {"dd3", w.dd3, B * SK2, true},
        {"dd2", w.dd2, B * SK2, true}, {"dd1", w.dd1, B * d.Pp, true}, {"dflat", w.dflat, B * SK2, true}, {"dt1", w.dt1, R * K2, true},
    };
    for (const Ent& e : tab) {
        if (strcmp(e.n, name) != 0) continue;
        if (e.tr && !train) return DCCN_ERR_INVALID_ARG;
        *byte_offset = (size_t)(reinterpret_cast<const char*>(e.p) - base);
        *count = e.cnt;
        return DCCN_OK;
    }
    return DCCN_ERR_INVALID_ARG;
}
size_t dccn_eq_rx_folded_floats(const dccn_eq_shape* shape) {
    if (!eq_shape_ok(shape)) return 0;
    const size_t N2 = 2 * (size_t)(shape->K + shape->CP), dN = 2 * (size_t)shape->D;
    return (size_t)shape->S * N2 * dN + dN;
}
int dccn_eq_rx_fold(const dccn_eq_shape* shape, const float* rx_params, float* out, dccn_stream_t stream) {
    if (!eq_shape_ok(shape) || !rx_params || !out) return DCCN_ERR_INVALID_ARG;
    const EqDims d = eq_dims(shape);
    dccn_rx_shape rsh;
    rsh.batch = 1; rsh.S = d.S; rsh.kin = d.cp ? d.nsc : d.K; rsh.F = d.F; rsh.D = d.D; rsh.nbits = shape->nbits;
    const RxLayout L = rx_layout(&rsh);
    const int N2 = 2 * d.nsc, rows = d.S * N2;
    hipLaunchKernelGGL(eq_rx_fold_kernel, dim3(rows + 1), dim3(256), 0, (hipStream_t)stream, rx_params + L.o_conv_w,
                       rx_params + L.o_conv_b, rx_params + L.o_dense_w, rx_params + L.o_dense_b, out, out + (size_t)rows * L.dN,
                       d.S, N2, d.win, rsh.kin, d.F, L.dN);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
int dccn_eq_eval_step(const dccn_eq_shape* shape, const dccn_eq_buffers* buf, dccn_stream_t stream) {
    return eq_step_impl(shape, buf, false, dccn_adam_hparams(), (hipStream_t)stream);
}
int dccn_eq_train_step(const dccn_eq_shape* shape, const dccn_eq_buffers* buf, dccn_adam_hparams hp,
                       dccn_stream_t stream) {
    return eq_step_impl(shape, buf, true, hp, (hipStream_t)stream);
}
// ---- chain groups: G independent equaliser chains per launch sequence (common.h ChainCtx) ------------------------------
// every device pointer of chain g must lie at ONE byte offset from chain 0's (the chains' arenas have the same layout)
struct ChainOffsetCheck {
    long long off[kMaxChains];
    bool have[kMaxChains];
    bool ok = true;
    int n;
    explicit ChainOffsetCheck(int n_) : n(n_) { for (int g = 0; g < kMaxChains; ++g) { off[g] = 0; have[g] = false; } }
    // pointers p[g] (field f of every chain's struct)
    void field(const void* const* p) {
        for (int g = 0; g < n && ok; ++g) {
            if ((p[g] == nullptr) != (p[0] == nullptr)) { ok = false; return; }
            if (p[g] == nullptr) continue;
            const long long d = (long long)(reinterpret_cast<const char*>(p[g]) - reinterpret_cast<const char*>(p[0]));
            if (!have[g]) { off[g] = d; have[g] = true; }
            else if (off[g] != d) ok = false;
        }
    }
    bool finish(ChainCtx* ctx) {
        if (!ok) return false;
        ctx->G = n;
        for (int g = 0; g < kMaxChains; ++g) { ctx->co.off[g] = 0; ctx->nbits[g] = 0; }
        for (int g = 0; g < n; ++g) {
            if (!have[g] || (off[g] & 255) != 0 || (g > 0 && off[g] == 0)) return false;
            ctx->co.off[g] = off[g];
        }
        return true;
    }
};
#define CHAIN_FIELD(chk, arr, n, member)                                             \
    do {                                                                             \
        const void* f__[kMaxChains];                                                 \
        for (int g__ = 0; g__ < (n); ++g__) f__[g__] = (const void*)((arr)[g__]->member); \
        (chk).field(f__);                                                            \
    } while (0)

static bool gen_static_same_plan(const dccn_gen_static* a, const dccn_gen_static* b) {
    if (a->frames != b->frames || a->S != b->S || a->K != b->K || a->CP != b->CP || a->D != b->D || a->n_taps != b->n_taps ||
        a->L != b->L || a->identity != b->identity || a->n_profiles != b->n_profiles || a->tap_stride != b->tap_stride ||
        a->h_rep != b->h_rep || a->pilot_re != b->pilot_re || a->pilot_im != b->pilot_im)
        return false;
    for (int i = 0; i < a->n_profiles; ++i)
        if (a->profiles[i].n_taps != b->profiles[i].n_taps || a->profiles[i].L != b->profiles[i].L ||
            a->profiles[i].identity != b->profiles[i].identity)
            return false;
    return true;
}
static void gen_static_chain_fields(ChainOffsetCheck& chk, const dccn_gen_static* const* g, int n) {
    CHAIN_FIELD(chk, g, n, bits_out); CHAIN_FIELD(chk, g, n, cell_map); CHAIN_FIELD(chk, g, n, const_tab);
    CHAIN_FIELD(chk, g, n, idft); CHAIN_FIELD(chk, g, n, coeff); CHAIN_FIELD(chk, g, n, alpha); CHAIN_FIELD(chk, g, n, snr_db);
    CHAIN_FIELD(chk, g, n, y); CHAIN_FIELD(chk, g, n, noise); CHAIN_FIELD(chk, g, n, power_partial);
    CHAIN_FIELD(chk, g, n, noise_partial); CHAIN_FIELD(chk, g, n, noise_power_out); CHAIN_FIELD(chk, g, n, tx_out);
    CHAIN_FIELD(chk, g, n, H_out);
    for (int i = 0; i < g[0]->n_profiles && chk.ok; ++i) {
        CHAIN_FIELD(chk, g, n, profiles[i].coeff);
        CHAIN_FIELD(chk, g, n, profiles[i].alpha);
    }
}

int dccn_chain_group_max(void) { return kMaxChains; }

int dccn_gen_static_frames_grouped(int n_chains, const dccn_gen_static* const* g, dccn_stream_t stream) {
    if (n_chains < 1 || n_chains > kMaxChains || !g) return DCCN_ERR_INVALID_ARG;
    for (int i = 0; i < n_chains; ++i)
        if (!g[i] || !gen_static_ok(g[i]) || !gen_static_same_plan(g[0], g[i])) return DCCN_ERR_INVALID_ARG;
    if (n_chains == 1) return gen_static_launch(g[0], (hipStream_t)stream);
    ChainOffsetCheck chk(n_chains);
    gen_static_chain_fields(chk, g, n_chains);
    ChainCtx ctx;
    if (!chk.finish(&ctx)) return DCCN_ERR_INVALID_ARG;
    GenChainScalars gc;
    memset(&gc, 0, sizeof(gc));
    gc.n = n_chains;
    for (int i = 0; i < n_chains; ++i) { gc.nbits[i] = g[i]->nbits; gc.offset[i] = g[i]->offset; gc.seed[i] = g[i]->seed; ctx.nbits[i] = g[i]->nbits; }
    ChainScope scope(ctx);
    return gen_static_launch(g[0], (hipStream_t)stream, &gc);
}

int dccn_gen_static_apply_grouped(int n_chains, const dccn_gen_static* const* g, float* const* x_out, float* const* noise_power,
                                  dccn_stream_t stream) {
    if (n_chains < 1 || n_chains > kMaxChains || !g || !x_out) return DCCN_ERR_INVALID_ARG;
    for (int i = 0; i < n_chains; ++i)
        if (!g[i] || !x_out[i] || !gen_static_ok(g[i]) || !gen_static_same_plan(g[0], g[i])) return DCCN_ERR_INVALID_ARG;
    if (n_chains == 1) return dccn_gen_static_apply(g[0], x_out[0], noise_power ? noise_power[0] : nullptr, stream);
    ChainOffsetCheck chk(n_chains);
    gen_static_chain_fields(chk, g, n_chains);
    chk.field(reinterpret_cast<const void* const*>(x_out));
    if (noise_power) chk.field(reinterpret_cast<const void* const*>(noise_power));
    ChainCtx ctx;
    if (!chk.finish(&ctx)) return DCCN_ERR_INVALID_ARG;
    ChainScope scope(ctx);
    return dccn_gen_static_apply(g[0], x_out[0], noise_power ? noise_power[0] : nullptr, stream);
}

int dccn_eq_group_supported(const dccn_eq_shape* shape) {
    if (!eq_shape_ok(shape) || g_tune[TUNE_EQ_REPLAN] != 1 || !g_tune[TUNE_FEWROW] || g_tune[TUNE_SKINNY] <= 0) return 0;
    const EqDims d = eq_dims(shape);
    // the launches that carry a chain index: the few-row plan of the fused step (<= 96 frames), the pilot bottleneck as one
    // launch per direction, the frozen receiver folded into one matrix
    return (d.B <= 96 && (d.Pp == 16 || d.Pp == 32) && dccn_eq_norm_rides(shape) == 1 && (d.S * 2 * d.nsc) % 16 == 0 &&
            d.S * 2 * d.nsc <= 1152) ? 1 : 0;
}

int dccn_eq_train_step_grouped(int n_chains, const dccn_eq_shape* const* shapes, const dccn_eq_buffers* const* bufs,
                               dccn_adam_hparams hp, dccn_stream_t stream) {
    if (n_chains < 1 || n_chains > kMaxChains || !shapes || !bufs) return DCCN_ERR_INVALID_ARG;
    for (int i = 0; i < n_chains; ++i) {
        if (!shapes[i] || !bufs[i] || !eq_shape_ok(shapes[i])) return DCCN_ERR_INVALID_ARG;
        const dccn_eq_shape *a = shapes[0], *c = shapes[i];
        // one launch plan: everything but the modulation agrees
        if (a->batch != c->batch || a->S != c->S || a->K != c->K || a->CP != c->CP || a->cp != c->cp || a->F != c->F || a->D != c->D ||
            a->pilot_size != c->pilot_size || a->P != c->P)
            return DCCN_ERR_INVALID_ARG;
        const dccn_eq_buffers *p = bufs[0], *q = bufs[i];
        if (p->workspace_bytes != q->workspace_bytes || p->reg_uniform != q->reg_uniform || p->x_prenormalised != q->x_prenormalised ||
            p->norm_slot != q->norm_slot || (p->x_next_virtual == nullptr) != (q->x_next_virtual == nullptr))
            return DCCN_ERR_INVALID_ARG;
        if (q->prob != nullptr) return DCCN_ERR_INVALID_ARG;          // (its size depends on the modulation: not part of the arena)
    }
    if (n_chains == 1) return eq_step_impl(shapes[0], bufs[0], true, hp, (hipStream_t)stream);
    if (!dccn_eq_group_supported(shapes[0])) return DCCN_ERR_UNSUPPORTED;
    ChainOffsetCheck chk(n_chains);
    CHAIN_FIELD(chk, bufs, n_chains, x); CHAIN_FIELD(chk, bufs, n_chains, bits); CHAIN_FIELD(chk, bufs, n_chains, eq_params);
    CHAIN_FIELD(chk, bufs, n_chains, eq_grads); CHAIN_FIELD(chk, bufs, n_chains, adam_m); CHAIN_FIELD(chk, bufs, n_chains, adam_v);
    CHAIN_FIELD(chk, bufs, n_chains, reg_coef); CHAIN_FIELD(chk, bufs, n_chains, adam); CHAIN_FIELD(chk, bufs, n_chains, rx_params);
    CHAIN_FIELD(chk, bufs, n_chains, out_eq); CHAIN_FIELD(chk, bufs, n_chains, chest); CHAIN_FIELD(chk, bufs, n_chains, snr_db);
    CHAIN_FIELD(chk, bufs, n_chains, pilot_carriers); CHAIN_FIELD(chk, bufs, n_chains, metrics); CHAIN_FIELD(chk, bufs, n_chains, tx_power);
    CHAIN_FIELD(chk, bufs, n_chains, workspace); CHAIN_FIELD(chk, bufs, n_chains, rx_folded); CHAIN_FIELD(chk, bufs, n_chains, x_next);
    if (bufs[0]->x_next_virtual != nullptr) {
        const dccn_gen_static* gv[kMaxChains];
        for (int i = 0; i < n_chains; ++i) {
            gv[i] = bufs[i]->x_next_virtual;
            if (!gen_static_same_plan(gv[0], gv[i])) return DCCN_ERR_INVALID_ARG;
        }
        CHAIN_FIELD(chk, gv, n_chains, y); CHAIN_FIELD(chk, gv, n_chains, noise); CHAIN_FIELD(chk, gv, n_chains, power_partial);
        CHAIN_FIELD(chk, gv, n_chains, noise_partial); CHAIN_FIELD(chk, gv, n_chains, noise_power_out);
    }
    for (int i = 0; i < n_chains; ++i)
        if ((bufs[i]->monitor == nullptr) != (bufs[0]->monitor == nullptr)) return DCCN_ERR_INVALID_ARG;
    if (bufs[0]->monitor != nullptr) {
        const dccn_eq_monitor* mm[kMaxChains];
        for (int i = 0; i < n_chains; ++i) {
            mm[i] = bufs[i]->monitor;
            if (mm[i]->chan_per_symbol != mm[0]->chan_per_symbol || mm[i]->workspace_bytes != mm[0]->workspace_bytes)
                return DCCN_ERR_INVALID_ARG;
        }
        CHAIN_FIELD(chk, mm, n_chains, chest); CHAIN_FIELD(chk, mm, n_chains, chan); CHAIN_FIELD(chk, mm, n_chains, metrics);
        CHAIN_FIELD(chk, mm, n_chains, tx_power); CHAIN_FIELD(chk, mm, n_chains, noise_power); CHAIN_FIELD(chk, mm, n_chains, acc5);
        CHAIN_FIELD(chk, mm, n_chains, rms_out); CHAIN_FIELD(chk, mm, n_chains, workspace);
    }
    ChainCtx ctx;
    if (!chk.finish(&ctx)) return DCCN_ERR_INVALID_ARG;
    if (bufs[0]->rx_folded == nullptr) return DCCN_ERR_UNSUPPORTED;
    for (int i = 0; i < n_chains; ++i) {
        ctx.nbits[i] = shapes[i]->nbits;
        if (bufs[i]->gen_next_rides != bufs[0]->gen_next_rides) return DCCN_ERR_INVALID_ARG;
        if (bufs[0]->gen_next_rides) {                     // (the generator's per-chain scalars travel with the group)
            const dccn_gen_static* gv = bufs[i]->x_next_virtual;
            if (!gv) return DCCN_ERR_INVALID_ARG;
            ctx.gen_seed[i] = gv->seed; ctx.gen_offset[i] = gv->offset; ctx.gen_nbits[i] = gv->nbits;
        }
    }
    if (bufs[0]->gen_next_rides) {
        const dccn_gen_static* gv[kMaxChains];
        for (int i = 0; i < n_chains; ++i) gv[i] = bufs[i]->x_next_virtual;
        ChainOffsetCheck chk2(n_chains);
        gen_static_chain_fields(chk2, gv, n_chains);
        ChainCtx same;
        if (!chk2.finish(&same)) return DCCN_ERR_INVALID_ARG;
        for (int i = 0; i < n_chains; ++i)
            if (same.co.off[i] != ctx.co.off[i]) return DCCN_ERR_INVALID_ARG;
    }
    ChainScope scope(ctx);
    return eq_step_impl(shapes[0], bufs[0], true, hp, (hipStream_t)stream);
}

int dccn_eq_monitor_accumulate_grouped(int n_chains, const dccn_eq_monitor* const* m, dccn_stream_t stream) {
    if (n_chains < 1 || n_chains > kMaxChains || !m) return DCCN_ERR_INVALID_ARG;
    for (int i = 0; i < n_chains; ++i) {
        if (!m[i]) return DCCN_ERR_INVALID_ARG;
        if (m[i]->chan_per_symbol != m[0]->chan_per_symbol || m[i]->B != m[0]->B || m[i]->S != m[0]->S || m[i]->K != m[0]->K ||
            m[i]->workspace_bytes != m[0]->workspace_bytes)
            return DCCN_ERR_INVALID_ARG;
    }
    const dccn_eq_monitor* a = m[0];
    if (n_chains == 1)
        return dccn_eq_monitor_accumulate(a->chest, a->chan, a->chan_per_symbol, a->B, a->S, a->K, a->metrics, a->tx_power,
                                          a->noise_power, a->acc5, a->rms_out, a->workspace, a->workspace_bytes, stream);
    ChainOffsetCheck chk(n_chains);
    CHAIN_FIELD(chk, m, n_chains, chest); CHAIN_FIELD(chk, m, n_chains, chan); CHAIN_FIELD(chk, m, n_chains, metrics);
    CHAIN_FIELD(chk, m, n_chains, tx_power); CHAIN_FIELD(chk, m, n_chains, noise_power); CHAIN_FIELD(chk, m, n_chains, acc5);
    CHAIN_FIELD(chk, m, n_chains, rms_out); CHAIN_FIELD(chk, m, n_chains, workspace);
    ChainCtx ctx;
    if (!chk.finish(&ctx)) return DCCN_ERR_INVALID_ARG;
    ChainScope scope(ctx);
    return dccn_eq_monitor_accumulate(a->chest, a->chan, a->chan_per_symbol, a->B, a->S, a->K, a->metrics, a->tx_power,
                                      a->noise_power, a->acc5, a->rms_out, a->workspace, a->workspace_bytes, stream);
}

int dccn_eq_norm_rides(const dccn_eq_shape* shape) {
    if (!eq_shape_ok(shape)) return 0;
    const EqDims d = eq_dims(shape);
    const int ncols = d.S * 2 * d.nsc;
    return (g_tune[TUNE_EQ_REPLAN] != 0 && kNormFusedCG == 2 && (ncols % 4) == 0 && d.B <= 128 * kNormFusedRPT) ? 1 : 0;
}
int dccn_eq_graph_create(const dccn_eq_shape* shape, const dccn_eq_buffers* buf, int mode, dccn_adam_hparams hp,
                         dccn_stream_t stream, dccn_rx_graph** out) {
    if (!out || !eq_shape_ok(shape) || !buf) return DCCN_ERR_INVALID_ARG;
    (void)stream;
    dccn_rx_graph* g = new dccn_rx_graph();
    memset(g, 0, sizeof(*g));
    if (hipStreamCreateWithFlags(&g->cap, hipStreamNonBlocking) != hipSuccess) {
        dccn_rx_graph_destroy(g);
        return DCCN_ERR_LAUNCH;
    }
    hipError_t e = hipStreamBeginCapture(g->cap, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) {
        dccn_rx_graph_destroy(g);
        return hip_fail(e);
    }
    const int st = eq_step_impl(shape, buf, (mode & 1) != 0, hp, g->cap);
    e = hipStreamEndCapture(g->cap, &g->graph);
    if (st != DCCN_OK || e != hipSuccess) {
        dccn_rx_graph_destroy(g);
        return st != DCCN_OK ? st : hip_fail(e);
    }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        dccn_rx_graph_destroy(g);
        return hip_fail(e);
    }
    *out = g;
    return DCCN_OK;
}

int dccn_rx_graph_launch(dccn_rx_graph* g, dccn_stream_t stream) {
    if (!g || !g->exec) return DCCN_ERR_STATE;
    DCCN_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return DCCN_OK;
}
int dccn_rx_graph_destroy(dccn_rx_graph* g) {
    if (!g) return DCCN_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    if (g->ev_fork) (void)hipEventDestroy(g->ev_fork);
    if (g->ev_join) (void)hipEventDestroy(g->ev_join);
    if (g->side) (void)hipStreamDestroy(g->side);
    if (g->cap) (void)hipStreamDestroy(g->cap);
    delete g;
    return DCCN_OK;
}

int dccn_timer_create(dccn_timer** out) {
    if (!out) return DCCN_ERR_INVALID_ARG;
    dccn_timer* t = new dccn_timer();
    if (hipEventCreate(&t->a) != hipSuccess || hipEventCreate(&t->b) != hipSuccess) {
        delete t;
        return DCCN_ERR_LAUNCH;
    }
    *out = t;
    return DCCN_OK;
}
int dccn_timer_start(dccn_timer* t, dccn_stream_t stream) {
    if (!t) return DCCN_ERR_STATE;
    DCCN_HIP(hipEventRecord(t->a, (hipStream_t)stream));
    return DCCN_OK;
}
int dccn_timer_stop(dccn_timer* t, dccn_stream_t stream) {
    if (!t) return DCCN_ERR_STATE;
    DCCN_HIP(hipEventRecord(t->b, (hipStream_t)stream));
    return DCCN_OK;
}
int dccn_timer_elapsed_ms(dccn_timer* t, float* ms) {
    if (!t || !ms) return DCCN_ERR_STATE;
    DCCN_HIP(hipEventSynchronize(t->b));
    DCCN_HIP(hipEventElapsedTime(ms, t->a, t->b));
    return DCCN_OK;
}
int dccn_timer_destroy(dccn_timer* t) {
    if (!t) return DCCN_OK;
    (void)hipEventDestroy(t->a);
    (void)hipEventDestroy(t->b);
    delete t;
    return DCCN_OK;
}
int dccn_stream_synchronize(dccn_stream_t stream) {
    DCCN_HIP(hipStreamSynchronize((hipStream_t)stream));
    return DCCN_OK;
}

// ---- equaliser stage operators ------------------------------------------------------------------
static inline unsigned ew_blocks(long long n) { return ew_blocks_n(n); }
int dccn_layer_norm_fwd(const float* x, float* y, float* mean, float* inv, int rows, int cols, float eps,
                        dccn_stream_t stream) {
    if (!x || !y || rows <= 0 || cols <= 0) return DCCN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(layer_norm_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, y, mean, inv, cols,
                       eps);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
int dccn_layer_norm_bwd(const float* dy, const float* y, const float* inv, float* dx, int rows, int cols,
                        dccn_stream_t stream) {
    if (!dy || !y || !inv || !dx || rows <= 0 || cols <= 0) return DCCN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(layer_norm_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, dy, y, inv, dx, cols);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
int dccn_tanh_fwd(const float* x, float* y, long long n, dccn_stream_t stream) {
    if (!x || !y || n <= 0) return DCCN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(tanh_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
int dccn_tanh_bwd(const float* dy, const float* y, float* dx, long long n, dccn_stream_t stream) {
    if (!dy || !y || !dx || n <= 0) return DCCN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
int dccn_equalize_fwd(const float* y, const float* h, float* eq, float* corr, long long n_pairs,
                      dccn_stream_t stream) {
    if (!y || !h || !eq || n_pairs <= 0) return DCCN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(equalize_fwd_kernel, dim3(ew_blocks(n_pairs)), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)y, (const float2*)h, (float2*)eq, (float2*)corr, n_pairs);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
int dccn_equalize_bwd(const float* y, const float* h, const float* d_eq, const float* d_corr, float* dy, float* dh,
                      long long n_pairs, dccn_stream_t stream) {
    if (!y || !h || (!d_eq && !d_corr) || (!dy && !dh) || n_pairs <= 0) return DCCN_ERR_INVALID_ARG;
    DCCN_LAUNCH_CHAINS_Z(equalize_bwd_kernel, dim3(ew_blocks(n_pairs)), dim3(256), 0, (hipStream_t)stream,
                         (const float2*)y, (const float2*)h, (const float2*)d_eq, (const float2*)d_corr, (float2*)dy,
                         (float2*)dh, n_pairs);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
int dccn_pilot_snr(const float* eq, const int* carriers, float* snr_db, int frames, int S, int K, int P,
                   dccn_stream_t stream) {
    if (!eq || !carriers || !snr_db || frames <= 0 || S <= 0 || K <= 0 || P <= 0) return DCCN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(pilot_snr_kernel, dim3(frames), dim3(64), 0, (hipStream_t)stream, (const float2*)eq, carriers,
                       snr_db, S, K, P);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
int dccn_cconv2d_same_expand(const float* w, const float* bias, float* T, float* bias_eff, int L, int W, int kL,
                             int kW, dccn_stream_t stream) {
    if (!w || !T || L <= 0 || W <= 0 || kL <= 0 || kW <= 0) return DCCN_ERR_INVALID_ARG;
    const long long n = (long long)L * W * 2;
    if (n * n > (1LL << 31)) return DCCN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(cconv2d_same_expand_kernel, dim3(ew_blocks(n * n)), dim3(256), 0, (hipStream_t)stream, w, bias,
                       T, bias_eff, L, W, kL, kW);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
int dccn_cconv2d_same_reduce(const float* dT, const float* dbias_eff, float* dw, float* dbias, int L, int W, int kL,
                             int kW, dccn_stream_t stream) {
    if (!dT || !dw || L <= 0 || W <= 0 || kL <= 0 || kW <= 0) return DCCN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(cconv2d_same_reduce_kernel, dim3(kL * kW + 1), dim3(64), 0, (hipStream_t)stream, dT, dbias_eff,
                       dw, dbias, L, W, kL, kW);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// ---- device-side input generator ------------------------------------------------------------------
int dccn_philox_fill(uint32_t* out, long long n, unsigned stream, unsigned offset, unsigned long long seed,
                     dccn_stream_t stream_handle) {
    if (!out || n <= 0) return DCCN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(philox_fill_kernel, dim3((unsigned)ceil_div_ll(n, 256)), dim3(256), 0,
                       (hipStream_t)stream_handle, out, n, stream, offset, seed);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
int dccn_ofdm_tx_frames(const int32_t* bits_in, int32_t* bits_out, const int32_t* cell_map, const float* const_tab,
                        float pilot_re, float pilot_im, const float* idft, float* grid_ws, float* tx, int frames,
                        int S, int K, int CP, int D, int nbits, unsigned long long seed, unsigned offset,
                        dccn_stream_t stream) {
    if (!cell_map || !const_tab || !idft || !grid_ws || !tx || frames <= 0 || S <= 0 || K <= 0 || CP < 0 || D <= 0 ||
        nbits < 1 || nbits > 4)
        return DCCN_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long long n_cells = (long long)frames * S * K;
    hipLaunchKernelGGL(tx_grid_kernel, dim3((unsigned)ceil_div_ll(n_cells, 256)), dim3(256), 0, s, bits_in, bits_out,
                       cell_map, (const float2*)const_tab, make_float2(pilot_re, pilot_im), (float2*)grid_ws, n_cells,
                       S * K, D, nbits, offset, seed);
    DCCN_LAUNCH_CHECK();
    return dense_fwd_impl(grid_ws, idft, nullptr, tx, frames * S, 2 * K, 2 * (K + CP), s);
}
static int chan_blocks_x(int T) { return ceil_div(T, 256); }
// persistent FIR grid: all items when they are few, else eight blocks per CU; never more than the partial slots
static int fir_blocks(int items, int cap) {
    int b = items < 8 * kCUs ? items : 8 * kCUs;
    if (b > cap) b = cap;
    return b < 1 ? 1 : b;
}
// static-channel FIR: whole frames per block (bx items each) so that a frame's taps are set up once in the launch
struct FirPlan {
    int blocks, ipb;
};
static FirPlan fir_plan(int items, int bx, int cap) {
    const int b0 = fir_blocks(items, cap);
    FirPlan p;
    p.ipb = ceil_div(ceil_div(items, b0), bx) * bx;
    p.blocks = ceil_div(items, p.ipb);
    return p;
}
size_t dccn_channel_awgn_workspace_size(int frames, int T, int L) {
    if (frames <= 0 || T <= 0 || L <= 0) return 0;
    size_t o = 0;
    o = carve_size(o, (size_t)frames * L * 2 * sizeof(float));
    o = carve_size(o, (size_t)frames * T * 2 * sizeof(float));
    o = carve_size(o, (size_t)kChanPartials * sizeof(double));
    o = carve_size(o, (size_t)frames * chan_blocks_x(T) * sizeof(double));
    o = carve_size(o, 4 * sizeof(float));
    return align_up(o, 256);
}
int dccn_channel_awgn(const float* tx, const float* taps_in, const float* coeff, const float* alpha, int n_taps,
                      int L, int identity, const float* snr_db, const float* noise_in, float* out, float* H, int nfft,
                      float* noise_power, int frames, int T, unsigned long long seed, unsigned offset,
                      void* workspace, size_t workspace_bytes, dccn_stream_t stream) {
    if (!tx || !snr_db || !out || frames <= 0 || frames > 65535 || T <= 0 || L <= 0 || L > 64) return DCCN_ERR_INVALID_ARG;
    if (!identity && (!coeff || !alpha || n_taps <= 0 || n_taps > 16)) return DCCN_ERR_INVALID_ARG;
    if (H && nfft <= 0) return DCCN_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < dccn_channel_awgn_workspace_size(frames, T, L)) return DCCN_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    Carver c(workspace, workspace_bytes);
    float* g = c.take<float>((size_t)frames * L * 2);
    float* y = c.take<float>((size_t)frames * T * 2);
    const int bx = chan_blocks_x(T);
    double* partial = c.take<double>((size_t)kChanPartials);
    double* npartial = c.take<double>((size_t)frames * bx);
    // nobody wants the frequency response and the taps are drawn here: the FIR blocks draw them themselves
    TapGen tg;
    memset(&tg, 0, sizeof(tg));
    tg.enabled = (H == nullptr && taps_in == nullptr) ? 1 : 0;
    tg.coeff = coeff; tg.alpha = alpha; tg.n_taps = n_taps; tg.identity = identity; tg.tap_stride = n_taps;
    tg.offset = offset; tg.seed = seed;
    if (!tg.enabled) {
        hipLaunchKernelGGL(channel_taps_kernel, dim3(frames), dim3(64), 0, s, taps_in, coeff, alpha, (float2*)g, (float2*)H,
                           n_taps, L, nfft, identity, offset, seed, (const int*)nullptr, n_taps, L, 1);
        DCCN_LAUNCH_CHECK();
    }
    const FirPlan fp = fir_plan(frames * bx, bx, kChanPartials);
    const int nfb = fp.blocks;
    hipLaunchKernelGGL(fir_same_kernel, dim3(nfb), dim3(256), 0, s, (const float2*)tx, (const float2*)g,
                       (float2*)y, partial, T, L, (const int*)nullptr, L, frames, 0, tg, fp.ipb);
    DCCN_LAUNCH_CHECK();
    const double total = (double)frames * (double)T;
    hipLaunchKernelGGL(awgn_kernel, dim3(bx, frames), dim3(256), 0, s, (const float2*)y, (const double*)partial, nfb, total,
                       snr_db, noise_in, (float2*)out, noise_power ? npartial : nullptr, T, offset, seed);
    DCCN_LAUNCH_CHECK();
    if (noise_power) {
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, (const double*)npartial, frames * bx, total,
                           noise_power);
        DCCN_LAUNCH_CHECK();
    }
    return DCCN_OK;
}

// ---- fused static-channel generator (datagen.h gen_static_frames_kernel) ----------------------------------------------
static_assert(sizeof(dccn_gen_static) == 200 && sizeof(dccn_rx_buffers) == 208 && sizeof(dccn_eq_buffers) == 208 &&
              sizeof(dccn_eq_monitor) == 88, "ctypes mirrors in dl_ofdm_amd/_lib.py");
int dccn_gen_static_supported(int S, int K, int CP) {
    return (S == 7 && K == 64 && CP == 16) ? 1 : 0;
}
int dccn_gen_static_partials(int frames) { return frames > 0 ? ceil_div(frames, kGenFramesPerBlock) : 0; }
int dccn_gen_static_frames(const dccn_gen_static* g, dccn_stream_t stream) { return gen_static_launch(g, (hipStream_t)stream); }
int dccn_gen_static_apply(const dccn_gen_static* g, float* x_out, float* noise_power, dccn_stream_t stream) {
    if (!gen_static_ok(g) || !x_out || !aligned16(x_out)) return DCCN_ERR_INVALID_ARG;
    const int T = g->S * (g->K + g->CP);
    const long long n4 = (long long)g->frames * T * 2 / 4;
    if (((long long)g->frames * T * 2) % 4 != 0) return DCCN_ERR_INVALID_ARG;
    const int np = dccn_gen_static_partials(g->frames);
    long long blocks = ceil_div_ll(n4, 256);
    if (blocks > 4 * kCUs) blocks = 4 * kCUs;
    DCCN_LAUNCH_CHAINS_Z(gen_static_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                         reinterpret_cast<const float4*>(g->y), reinterpret_cast<const float4*>(g->noise),
                         (const double*)g->power_partial, np, (double)g->frames * (double)T, reinterpret_cast<float4*>(x_out), n4,
                         (const double*)((noise_power && g->noise_partial) ? g->noise_partial : nullptr), np,
                         (noise_power && g->noise_partial) ? noise_power : nullptr);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

size_t dccn_channel_doppler_awgn_workspace_size(int frames, int T, int L, int S) {
    if (frames <= 0 || T <= 0 || L <= 0 || S <= 0) return 0;
    return dccn_channel_awgn_workspace_size(frames, T, L * S);
}
int dccn_channel_doppler_awgn(const float* tx, const float* theta_in, const float* coeff, const float* alpha,
                              int n_taps, int L, float Fd, float t_sym, int S, int n_sc, const float* snr_db,
                              const float* noise_in, float* out, float* H, int nfft, float* noise_power, int frames,
                              unsigned long long seed, unsigned offset, void* workspace, size_t workspace_bytes,
                              dccn_stream_t stream) {
    if (!tx || !coeff || !alpha || !snr_db || !out || frames <= 0 || frames > 65535 || S <= 0 || S > 16 || n_sc <= 0 ||
        L <= 0 || L > 64 || n_taps <= 0 || n_taps > 16 || (H && nfft <= 0))
        return DCCN_ERR_INVALID_ARG;
    const int T = S * n_sc;
    if (!workspace || workspace_bytes < dccn_channel_doppler_awgn_workspace_size(frames, T, L, S))
        return DCCN_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    Carver c(workspace, workspace_bytes);
    float* g = c.take<float>((size_t)frames * S * L * 2);
    float* y = c.take<float>((size_t)frames * T * 2);
    const int bx = chan_blocks_x(T);
    double* partial = c.take<double>((size_t)kChanPartials);
    double* npartial = c.take<double>((size_t)frames * bx);
    hipLaunchKernelGGL(doppler_taps_kernel, dim3(frames), dim3(64), 0, s, theta_in, coeff, alpha, (float2*)g, (float2*)H,
                       n_taps, L, nfft, S, Fd, t_sym, offset, seed, (const int*)nullptr, n_taps, S * L);
    DCCN_LAUNCH_CHECK();
    const int nfb = fir_blocks(frames * bx, kChanPartials);
    hipLaunchKernelGGL(fir_doppler_kernel, dim3(nfb), dim3(256), 0, s, (const float2*)tx, (const float2*)g,
                       (float2*)y, partial, T, L, n_sc, n_taps, (const int*)nullptr, S * L, frames, 0);
    DCCN_LAUNCH_CHECK();
    const double total = (double)frames * (double)T;
    hipLaunchKernelGGL(awgn_kernel, dim3(bx, frames), dim3(256), 0, s, (const float2*)y, (const double*)partial, nfb, total,
                       snr_db, noise_in, (float2*)out, noise_power ? npartial : nullptr, T, offset, seed);
    DCCN_LAUNCH_CHECK();
    if (noise_power) {
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, (const double*)npartial, frames * bx, total,
                           noise_power);
        DCCN_LAUNCH_CHECK();
    }
    return DCCN_OK;
}

size_t dccn_channel_groups_awgn_workspace_size(int frames, int T, int S) {
    if (frames <= 0 || T <= 0 || S <= 0) return 0;
    return dccn_channel_awgn_workspace_size(frames, T, 64 * S);
}
int dccn_channel_groups_awgn(const float* tx, const dccn_channel_group* groups, int n_groups, const float* taps_in,
                             const float* theta_in, float t_sym, int S, int n_sc, const float* snr_db,
                             const float* noise_in, float* out, float* H, int nfft, float* noise_power, int frames,
                             unsigned long long seed, unsigned offset, void* workspace, size_t workspace_bytes,
                             dccn_stream_t stream) {
    if (!tx || !groups || n_groups <= 0 || !snr_db || !out || frames <= 0 || frames > 65535 || S <= 0 || S > 16 ||
        n_sc <= 0 || (H && nfft <= 0))
        return DCCN_ERR_INVALID_ARG;
    const int T = S * n_sc;
    int covered = 0;
    for (int i = 0; i < n_groups; ++i) {
        const dccn_channel_group& g = groups[i];
        if (g.n_frames < 0 || (g.n_frames > 0 && !g.frames && n_groups > 1)) return DCCN_ERR_INVALID_ARG;
        if (!g.identity && (!g.coeff || !g.alpha || g.n_taps <= 0 || g.n_taps > 16 || g.L <= 0 || g.L > 64))
            return DCCN_ERR_INVALID_ARG;
        covered += g.n_frames;
    }
    if (covered != frames) return DCCN_ERR_INVALID_ARG;        // every frame belongs to exactly one group
    if (!workspace || workspace_bytes < dccn_channel_groups_awgn_workspace_size(frames, T, S)) return DCCN_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    Carver c(workspace, workspace_bytes);
    const int gstride = 64 * S;
    float* g = c.take<float>((size_t)frames * gstride * 2);
    float* y = c.take<float>((size_t)frames * T * 2);
    const int bx = chan_blocks_x(T);
    double* partial = c.take<double>((size_t)kChanPartials);
    double* npartial = c.take<double>((size_t)frames * bx);
    int live = 0;
    for (int i = 0; i < n_groups; ++i) live += groups[i].n_frames > 0 ? 1 : 0;
    const int pcap = kChanPartials / (live > 0 ? live : 1);       // partial slots per group launch
    int pbase = 0;
    for (int i = 0; i < n_groups; ++i) {
        const dccn_channel_group& q = groups[i];
        if (q.n_frames == 0) continue;
        if (q.identity || q.Fd <= 0.f) {
            const int L = q.identity ? 1 : q.L;
            TapGen tg;                          // (see dccn_channel_awgn)
            memset(&tg, 0, sizeof(tg));
            tg.enabled = (H == nullptr && taps_in == nullptr) ? 1 : 0;
            tg.coeff = q.coeff; tg.alpha = q.alpha; tg.n_taps = q.n_taps; tg.identity = q.identity; tg.tap_stride = 16;
            tg.offset = offset; tg.seed = seed;
            if (!tg.enabled) {
                hipLaunchKernelGGL(channel_taps_kernel, dim3(q.n_frames), dim3(64), 0, s, taps_in, q.coeff, q.alpha, (float2*)g,
                                   (float2*)H, q.n_taps, L, nfft, q.identity, offset, seed, q.frames, 16, gstride, S);
                DCCN_LAUNCH_CHECK();
            }
            const FirPlan fp = fir_plan(q.n_frames * bx, bx, pcap);
            const int nfb = fp.blocks;
            hipLaunchKernelGGL(fir_same_kernel, dim3(nfb), dim3(256), 0, s, (const float2*)tx, (const float2*)g,
                               (float2*)y, partial, T, L, q.frames, gstride, q.n_frames, pbase, tg, fp.ipb);
            DCCN_LAUNCH_CHECK();
            pbase += nfb;
        } else {
            hipLaunchKernelGGL(doppler_taps_kernel, dim3(q.n_frames), dim3(64), 0, s, theta_in, q.coeff, q.alpha, (float2*)g,
                               (float2*)H, q.n_taps, q.L, nfft, S, q.Fd, t_sym, offset, seed, q.frames, 16, gstride);
            DCCN_LAUNCH_CHECK();
            const int nfb = fir_blocks(q.n_frames * bx, pcap);
            hipLaunchKernelGGL(fir_doppler_kernel, dim3(nfb), dim3(256), 0, s, (const float2*)tx,
                               (const float2*)g, (float2*)y, partial, T, q.L, n_sc, q.n_taps, q.frames, gstride, q.n_frames, pbase);
            DCCN_LAUNCH_CHECK();
            pbase += nfb;
        }
    }
    const double total = (double)frames * (double)T;
    hipLaunchKernelGGL(awgn_kernel, dim3(bx, frames), dim3(256), 0, s, (const float2*)y, (const double*)partial, pbase, total,
                       snr_db, noise_in, (float2*)out, noise_power ? npartial : nullptr, T, offset, seed);
    DCCN_LAUNCH_CHECK();
    if (noise_power) {
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, (const double*)npartial, frames * bx, total,
                           noise_power);
        DCCN_LAUNCH_CHECK();
    }
    return DCCN_OK;
}

// ---- classical pilot-aided receivers (classical.h) -----------------------------------------------------------------
size_t dccn_classical_workspace_size(void) {
    size_t o = 0;
    o = carve_size(o, (size_t)kClassicalPartials * 4 * sizeof(double));
    o = carve_size(o, (size_t)kClassicalPartials * sizeof(long long));
    return align_up(o, 256);
}
static int classical_blocks(long long items) {
    long long b = items < 2 * kCUs ? items : 2 * kCUs;
    if (b > kClassicalPartials) b = kClassicalPartials;
    return (int)(b < 1 ? 1 : b);
}
int dccn_dense_fwd_ld(const float* x, int ldx, const float* w, const float* bias, float* y, int M, int K, int N,
                      dccn_stream_t stream) {
    if (ldx < K) return DCCN_ERR_INVALID_ARG;
    return dense_fwd_impl(x, w, bias, y, M, K, N, (hipStream_t)stream, ldx);
}
int dccn_classical_pilot_ls(const float* Y, const int* pil, float* gp, int n, int SK, int P, float pv_re, float pv_im,
                            dccn_stream_t stream) {
    if (!Y || !pil || !gp || n <= 0 || SK <= 0 || P <= 0 || (pv_re == 0.f && pv_im == 0.f)) return DCCN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(classical_pilot_ls_kernel, dim3((unsigned)ceil_div_ll((long long)n * P, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float2*)Y, pil, gp, n, SK, P, make_float2(pv_re, pv_im));
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
int dccn_classical_gain(const float* Y, const float* H, const float* Gls, const int* pil, int n, int SK, int P, float pv_re,
                        float pv_im, double* sums4, void* workspace, size_t workspace_bytes, dccn_stream_t stream) {
    if (!Y || !Gls || !pil || n <= 0 || SK <= 0 || P <= 0) return DCCN_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < dccn_classical_workspace_size()) return DCCN_ERR_WORKSPACE;
    Carver c(workspace, workspace_bytes);
    double* partial = c.take<double>((size_t)kClassicalPartials * 4);
    const int nblk = classical_blocks(n);       // (dccn_classical_estimate modes 1 / 3 read these partials: dccn.h order contract)
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(classical_gain_kernel, dim3(nblk), dim3(256), 0, s, (const float2*)Y, (const float2*)H, Gls, pil, partial,
                       n, SK, P, make_float2(pv_re, pv_im));
    DCCN_LAUNCH_CHECK();
    if (sums4) {
        hipLaunchKernelGGL(classical_finish_kernel, dim3(1), dim3(256), 0, s, (const long long*)nullptr, 0, (const double*)partial,
                           nblk, (long long*)nullptr, sums4);
        DCCN_LAUNCH_CHECK();
    }
    return DCCN_OK;
}
int dccn_classical_estimate(const float* Gls, const float* H, float* G, int n, int S, int K, int mode, float c_var,
                            void* workspace, size_t workspace_bytes, dccn_stream_t stream) {
    if (!Gls || !G || n <= 0 || S <= 0 || S > 32 || K <= 0 || mode < CE_LS || mode > CE_FRAME_MEAN) return DCCN_ERR_INVALID_ARG;
    if ((mode == CE_LMMSE || mode == CE_PERFECT) && !H) return DCCN_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < dccn_classical_workspace_size()) return DCCN_ERR_WORKSPACE;
    Carver c(workspace, workspace_bytes);
    double* partial = c.take<double>((size_t)kClassicalPartials * 4);
    hipLaunchKernelGGL(classical_estimate_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, Gls, (const float2*)H,
                       (const double*)partial, classical_blocks(n), (float2*)G, n, S, K, mode, c_var);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
int dccn_classical_detect(const float* Y, const float* G, const int* dat, const float* table, const int* labels,
                          const int32_t* bits, int32_t* det, long long* errors, int n, int SK, int D, int m, int nbits,
                          int g_row, int g_mod, void* workspace, size_t workspace_bytes, dccn_stream_t stream) {
    if (!Y || !G || !dat || !table || !labels || !bits || !errors || n <= 0 || SK <= 0 || D <= 0 || m <= 0 || nbits < 1 ||
        nbits > 8 || g_row <= 0)
        return DCCN_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < dccn_classical_workspace_size()) return DCCN_ERR_WORKSPACE;
    Carver c(workspace, workspace_bytes);
    c.take<double>((size_t)kClassicalPartials * 4);
    long long* ep = c.take<long long>((size_t)kClassicalPartials);
    const int nblk = classical_blocks(ceil_div_ll((long long)n * D, 256));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(classical_detect_kernel, dim3(nblk), dim3(256), 0, s, (const float2*)Y, (const float2*)G, dat,
                       (const float2*)table, labels, bits, det, ep, n, SK, D, m, nbits, g_row, g_mod);
    DCCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(classical_finish_kernel, dim3(1), dim3(256), 0, s, (const long long*)ep, nblk, (const double*)nullptr, 0,
                       errors, (double*)nullptr);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// ---- per-step monitors of the equaliser harness in one launch (equalizer.h eq_monitor_kernel) --------------------
size_t dccn_eq_monitor_workspace_size(int B, int S, int K) {
    if (B <= 0 || S <= 0 || K <= 0) return 0;
    return align_up(256 + (size_t)eq_monitor_blocks(B, K) * sizeof(double), 256);
}
int dccn_eq_monitor_accumulate(const float* chest, const float* chan, int chan_per_symbol, int B, int S, int K,
                               const dccn_metrics* metrics, const float* tx_power, const float* noise_power, float* acc5,
                               float* rms_out, void* workspace, size_t workspace_bytes, dccn_stream_t stream) {
    if (!chest || !chan || B <= 0 || S <= 0 || K <= 0 || (acc5 && !metrics) || (!acc5 && !rms_out)) return DCCN_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < dccn_eq_monitor_workspace_size(B, S, K)) return DCCN_ERR_WORKSPACE;
    EqMonitorArgs a;
    a.chest = chest; a.chan = chan; a.gt_per_symbol = chan_per_symbol ? 1 : 0; a.B = B; a.S = S; a.K = K;
    a.metrics = metrics; a.tx_power = tx_power; a.noise_power = noise_power; a.acc = acc5; a.rms_out = rms_out;
    a.counter = static_cast<unsigned*>(workspace);
    a.partial = reinterpret_cast<double*>(static_cast<char*>(workspace) + 256);
    DCCN_LAUNCH_CHAINS_Z(eq_monitor_kernel, dim3(eq_monitor_blocks(B, K)), dim3(256), 0, (hipStream_t)stream, a);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// ---- the equaliser's pilot bottleneck as one launch per direction (eq_bottleneck.h) ---------------------------
int dccn_eq_bottleneck_supported(int B, int SK2, int P) {
    return ((P == 16 || P == 32) && B > 0 && SK2 >= 64 && (SK2 % 64) == 0) ? 1 : 0;
}
size_t dccn_eq_bottleneck_workspace_size(int B, int SK2, int P) {
    if (!dccn_eq_bottleneck_supported(B, SK2, P)) return 0;
    return align_up(eq_bottleneck_part_floats(B, SK2, P) * sizeof(float), 256);
}
int dccn_eq_bottleneck_fwd(const float* y, const float* W1, const float* b1, const float* W2, const float* b2, float* d1,
                           float* d2, int B, int SK2, int P, dccn_stream_t stream) {
    if (!y || !W1 || !W2 || !d1 || !d2 || !eq_bottleneck_ok(B, SK2, P, y, W1, W2) || !aligned16(d1)) return DCCN_ERR_INVALID_ARG;
    const int q = eq_bottleneck_q(B, SK2);
    auto kern = P == 32 ? eq_bottleneck_fwd_kernel<2> : eq_bottleneck_fwd_kernel<1>;
    DCCN_LAUNCH_CHAINS_Z(kern, dim3(ceil_div(SK2 / 16, q), ceil_div(B, 16)), dim3(256), 0, (hipStream_t)stream, y, W1, b1, W2, b2,
                         d1, d2, B, SK2, q);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
int dccn_eq_bottleneck_bwd(const float* dd2, const float* d1, const float* y, const float* W1, const float* W2,
                           const float* dy_in, float* dy_out, float* dW1, float* db1, float* dW2, float* db2, int B, int SK2,
                           int P, void* workspace, size_t workspace_bytes, dccn_stream_t stream) {
    if (!dd2 || !d1 || !y || !W1 || !W2 || !dy_in || !dy_out || !dW1 || !db1 || !dW2 || !db2 ||
        !eq_bottleneck_ok(B, SK2, P, y, W1, W2) || !aligned16(dd2) || !aligned16(d1))
        return DCCN_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < dccn_eq_bottleneck_workspace_size(B, SK2, P)) return DCCN_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int tiles = ceil_div(B, 16), q = eq_bottleneck_q(B, SK2);
    float* pw2 = static_cast<float*>(workspace);
    float* pb2 = pw2 + (size_t)tiles * P * SK2;
    float* pw1 = pb2 + (size_t)tiles * SK2;
    float* pb1 = pw1 + (size_t)tiles * SK2 * P;
    auto kern = P == 32 ? eq_bottleneck_bwd_kernel<2> : eq_bottleneck_bwd_kernel<1>;
    EqRideArgs no_ride;
    memset(&no_ride, 0, sizeof(no_ride));
    dccn_adam_hparams no_hp;
    memset(&no_hp, 0, sizeof(no_hp));
    GenStaticArgs no_gen;
    GenChainScalars no_gc;
    memset(&no_gen, 0, sizeof(no_gen));
    memset(&no_gc, 0, sizeof(no_gc));
    DCCN_LAUNCH_CHAINS_Z(kern, dim3(ceil_div(SK2 / 16, q), tiles), dim3(256), 0, s, dd2, d1, y, W1, W2, dy_in, dy_out, pw2, pb2,
                         pw1, pb1, B, SK2, q, tiles, no_ride, no_hp, 0, 0, no_gen, no_gc);
    DCCN_LAUNCH_CHECK();
    // (the fused equaliser step leaves these sums to its optimizer launch)
    DCCN_TRY(launch_splitk_reduce2(pw2, tiles, (long long)P * SK2, dW2, (long long)P * SK2, pb2, (long long)SK2, db2, (long long)SK2, s));
    DCCN_TRY(launch_splitk_reduce2(pw1, tiles, (long long)SK2 * P, dW1, (long long)SK2 * P, pb1, (long long)P, db1, (long long)P, s));
    return DCCN_OK;
}

// ---- patch gather of the general-k complex convolutions -----------------------------------------------------
static bool im2col_geom_ok(const Im2colGeom& g) {
    return g.B > 0 && g.L > 0 && g.Wd > 0 && g.C > 0 && g.Lo > 0 && g.Wo > 0 && g.ntl > 0 && g.ntw > 0 && g.sL > 0 && g.sW > 0 &&
           g.tl0 >= 0 && g.tw0 >= 0 && g.pl0 >= 0 && g.pw0 >= 0;
}
int dccn_cconv_im2col(const float* x, float* rows, int B, int L, int Wd, int C, int Lo, int Wo, int ntl, int ntw, int tl0,
                      int tw0, int sL, int sW, int pl0, int pw0, dccn_stream_t stream) {
    const Im2colGeom g{B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0};
    if (!x || !rows || !im2col_geom_ok(g)) return DCCN_ERR_INVALID_ARG;
    const long long n = (long long)B * Lo * Wo * ntl * ntw * C;
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)ceil_div_ll(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)x, (float2*)rows, g, n);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
// the same convolution without the patch tensor: the GEMM's A-loader gathers the taps (gemm_f32_mfma.h OP_KPATCH)
int dccn_cconv_patch_supported(int B, int L, int Wd, int C, int Lo, int Wo, int ntl, int ntw, int F) {
    // float4 pieces must not straddle cells: 2C multiple of 4; 32-bit element offsets into x; weights on the vector loaders
    if (B <= 0 || L <= 0 || Wd <= 0 || C <= 0 || Lo <= 0 || Wo <= 0 || ntl <= 0 || ntw <= 0 || F <= 0) return 0;
    if ((C % 2) != 0 || (F % 2) != 0) return 0;
    if ((long long)B * L * Wd * C * 2 >= (1LL << 31) || (long long)B * Lo * Wo >= (1LL << 31)) return 0;
    if ((long long)ntl * ntw * C * 2 * 2 * F * 4 >= (1LL << 31)) return 0;
    return 1;
}
int dccn_cconv_patch_fwd(const float* x, const float* w, const float* bias, float* out, int B, int L, int Wd, int C, int Lo,
                         int Wo, int ntl, int ntw, int tl0, int tw0, int sL, int sW, int pl0, int pw0, int F,
                         dccn_stream_t stream) {
    const Im2colGeom g{B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0};
    if (!x || !w || !out || !im2col_geom_ok(g) || !dccn_cconv_patch_supported(B, L, Wd, C, Lo, Wo, ntl, ntw, F) ||
        !aligned16(x) || !aligned16(w))
        return DCCN_ERR_INVALID_ARG;
    const int kin = ntl * ntw * C;
    GemmParams p = gp_zero();                 // out[rows,2F] = patches[rows,2kin] . Weff[2kin,2F]
    p.A = x; p.B = w; p.C = out; p.bias = bias; p.cbias = 1;
    p.M = B * Lo * Wo; p.N = 2 * F; p.K = 2 * kin;
    p.lda = 0; p.ldb = 2 * F; p.ldc = 2 * F;
    p.klen = round_k(2 * kin);
    p.cF = F;
    p.vecA = 1; p.vecB = 1;
    p.pg.L = L; p.pg.Wd = Wd; p.pg.c2 = 2 * C; p.pg.Lo = Lo; p.pg.Wo = Wo; p.pg.ntl = ntl; p.pg.ntw = ntw;
    p.pg.sL = sL; p.pg.sW = sW; p.pg.l0 = tl0 - pl0; p.pg.w0 = tw0 - pw0;
    return launch_gemm<OP_KPATCH, OP_CCONV_W, 0, TAG_CCONV_FWD>(p, 1, (hipStream_t)stream);
}
// ---- the backward of the same convolutions without the patch tensor ---------------------------------------------
// weight gradient: dWeff[(ti,tj,c,iq), n] = sum over output positions of patch(x)^T . dout -- the k-major weight-gradient
// GEMM (gemm_kmajor.h) with its A rows gathered from x itself (APATCH); split-K slabs folded like every C-Conv's
int dccn_cconv_patch_bwd_supported(int B, int L, int Wd, int C, int Lo, int Wo, int ntl, int ntw, int sL, int sW, int F) {
    if (!dccn_cconv_patch_supported(B, L, Wd, C, Lo, Wo, ntl, ntw, F)) return 0;
    if (sL <= 0 || sW <= 0) return 0;
    if ((long long)B * Lo * Wo * F * 2 >= (1LL << 31)) return 0;          // 32-bit element offsets into dout
    int mode = 1 | 2;                  // bit 0: weight gradient; bit 1: input gradient as an implicit GEMM (any stride, round 6)
    // bit 2: ... and it is expected to beat GEMM + col2im.  Cost per INPUT position in units of 16 tile columns x 2F deep rows:
    //   few channels (2C <= 32, cconv_dx_narrow.h): ceil(2C/16) columns x the taps of the position's stride phase (ntl*ntw / (sL*sW));
    //   otherwise (64-wide tiles, inverse-stride gather): 4 ceil(2C/64) columns x ALL ntl*ntw taps (zeros where the stride skips);
    //   GEMM + col2im: 4 ceil(2 kin/64) columns per OUTPUT position (sL*sW times fewer) plus the [rows, kin, 2] round trip and the
    //   scatter launch -- which is why the implicit route may cost up to 3x (narrow) / 2x (wide) the other's tile columns.
    const long long taps = (long long)ntl * ntw, ss = (long long)sL * sW, col = 4LL * ceil_div(2 * ntl * ntw * C, 64);
    if (cconv_dx_narrow_ok(C, F, sL, sW)) {
        if ((long long)ceil_div(2 * C, 16) * taps <= 3 * col) mode |= 4;          // (both sides per input position: x ss cancels)
    } else if (4LL * ceil_div(2 * C, 64) * taps * ss * ss <= 2 * col) {
        mode |= 4;
    }
    return mode;
}
size_t dccn_cconv_patch_bwd_w_workspace_size(int B, int Lo, int Wo, int C, int ntl, int ntw, int F) {
    if (B <= 0 || Lo <= 0 || Wo <= 0 || C <= 0 || ntl <= 0 || ntw <= 0 || F <= 0) return 0;
    return cconv_bw_ws_bytes(B * Lo * Wo, ntl * ntw * C, F);
}
int dccn_cconv_patch_bwd_w(const float* x, const float* dout, float* dw, float* dbias, int B, int L, int Wd, int C, int Lo,
                           int Wo, int ntl, int ntw, int tl0, int tw0, int sL, int sW, int pl0, int pw0, int F,
                           void* workspace, size_t workspace_bytes, dccn_stream_t stream) {
    const Im2colGeom g{B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0};
    if (!x || !dout || !dw || !im2col_geom_ok(g) || !aligned16(x) || !aligned16(dout) ||
        !(dccn_cconv_patch_bwd_supported(B, L, Wd, C, Lo, Wo, ntl, ntw, sL, sW, F) & 1))
        return DCCN_ERR_INVALID_ARG;
    const int rows = B * Lo * Wo, kin = ntl * ntw * C;
    if (!workspace || workspace_bytes < cconv_bw_ws_bytes(rows, kin, F)) return DCCN_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const SplitPlan sp = plan_splitk(2 * kin, 2 * F, rows);
    Carver c(workspace, workspace_bytes);
    float* slabs = c.take<float>((size_t)sp.splits * 4 * kin * F);
    float* cs = c.take<float>((size_t)sp.splits * 2 * F);
    GemmParams p = gp_zero();                 // dWeff[2kin,2F] = patches(x)[rows,2kin]^T . dout[rows,2F]
    p.A = x; p.B = dout; p.C = slabs; p.colsum = cs;
    p.M = 2 * kin; p.N = 2 * F; p.K = rows;
    p.lda = 0; p.ldb = 2 * F; p.ldc = 2 * F;
    p.klen = sp.klen;
    p.slab = (long long)4 * kin * F;
    p.vecA = 1; p.vecB = 1;
    p.pg.L = L; p.pg.Wd = Wd; p.pg.c2 = 2 * C; p.pg.Lo = Lo; p.pg.Wo = Wo; p.pg.ntl = ntl; p.pg.ntw = ntw;
    p.pg.sL = sL; p.pg.sW = sW; p.pg.l0 = tl0 - pl0; p.pg.w0 = tw0 - pw0;
    patch_div_magic(Lo * Wo, p.pg.per_mul, p.pg.per_shift);
    patch_div_magic(Wo, p.pg.wo_mul, p.pg.wo_shift);
    if (!kmajor_ok(p)) return DCCN_ERR_INVALID_ARG;
    DCCN_TRY((launch_kmajor<1, TAG_CCONV_BWD_W, true>(p, sp.splits, s)));
    const int fold_blocks = ceil_div(kin * F + F, kRedLanes);
    hipLaunchKernelGGL(cconv_fold_kernel, dim3(fold_blocks), dim3(256), 0, s, slabs, sp.splits, p.slab, cs, dw, dbias, kin, F);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
// input gradient: dx[b,l,w,(c,iq)] = sum over taps of dout[b, (l-l0-ti)/sL, (w-w0-tj)/sW, :] . Weff[(ti,tj,c,iq), :] -- a
// convolution of dout with the tap-flipped transposed weights, i.e. the SAME implicit GEMM with dout as the gathered operand
// (geometry: rows = input positions, taps t' = nt-1-t, origin -(l0+ntl-1)) and Bt[(c,iq)][(ti',tj',n)] as a plain k-contiguous
// operand built from w by the little kernel below (4 kin F floats).  No [rows, kin, 2] gradient-of-patches tensor, no col2im.
// Strides (round 6): a tap contributes where its fine position is a multiple of the forward stride -- the loader's
// inverse-stride gather (PatchGeom::isL / isW) reads dout there and zeros elsewhere.
__global__ __launch_bounds__(256) void cconv_flip_wt_kernel(const float* __restrict__ w, float* __restrict__ bt, int C, int ntl,
                                                            int ntw, int F) {
    const long long K = (long long)ntl * ntw * 2 * F, total = 2LL * C * K;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i / K);
    const int k = (int)(i - (long long)j * K);
    const int tp = k / (2 * F), nn = k - tp * 2 * F;
    const int tip = tp / ntw, tjp = tp - tip * ntw;
    const int n = (((ntl - 1 - tip) * ntw) + (ntw - 1 - tjp)) * C + (j >> 1);
    const int iq = j & 1, f = nn >> 1, oq = nn & 1;
    const float wa = w[(size_t)n * 2 * F + f], wb = w[(size_t)n * 2 * F + F + f];
    // Weff[2n, 2f] = Wa, [2n, 2f+1] = Wb, [2n+1, 2f] = -Wb, [2n+1, 2f+1] = -Wa (gemm_f32_mfma.h OP_CCONV_W)
    bt[i] = iq == 0 ? (oq == 0 ? wa : wb) : (oq == 0 ? -wb : -wa);
}
size_t dccn_cconv_patch_bwd_x_workspace_size(int C, int ntl, int ntw, int F) {
    if (C <= 0 || ntl <= 0 || ntw <= 0 || F <= 0) return 0;
    return align_up((size_t)4 * C * ntl * ntw * F * sizeof(float), 256);
}
int dccn_cconv_patch_bwd_x(const float* dout, const float* w, float* dx, int B, int L, int Wd, int C, int Lo, int Wo, int ntl,
                           int ntw, int tl0, int tw0, int sL, int sW, int pl0, int pw0, int F, void* workspace,
                           size_t workspace_bytes, dccn_stream_t stream) {
    const Im2colGeom g{B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0};
    if (!dout || !w || !dx || !im2col_geom_ok(g) || !aligned16(dout) || !aligned16(dx) ||
        !(dccn_cconv_patch_bwd_supported(B, L, Wd, C, Lo, Wo, ntl, ntw, sL, sW, F) & 2))
        return DCCN_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < dccn_cconv_patch_bwd_x_workspace_size(C, ntl, ntw, F) || !aligned16(workspace))
        return DCCN_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* bt = reinterpret_cast<float*>(workspace);
    const long long total = 4LL * C * ntl * ntw * F;
    hipLaunchKernelGGL(cconv_flip_wt_kernel, dim3((unsigned)ceil_div_ll(total, 256)), dim3(256), 0, s, w, bt, C, ntl, ntw, F);
    DCCN_LAUNCH_CHECK();
    if (cconv_dx_narrow_ok(C, F, sL, sW)) {   // few channels: 16-column tiles, strides by phase (cconv_dx_narrow.h)
        DxNarrowArgs a;
        memset(&a, 0, sizeof(a));
        a.dout = dout; a.bt = bt; a.dx = dx;
        a.B = B; a.L = L; a.Wd = Wd; a.C2 = 2 * C; a.Lo = Lo; a.Wo = Wo; a.ntl = ntl; a.ntw = ntw; a.F2 = 2 * F;
        a.l0 = -(tl0 - pl0) - (ntl - 1); a.w0 = -(tw0 - pw0) - (ntw - 1);
        a.sL = sL; a.sW = sW;
        return launch_cconv_dx_narrow(a, s);
    }
    GemmParams p = gp_zero();                 // dx[B*L*Wd, 2C] = patches'(dout)[., ntl*ntw*2F] . Bt^T
    p.A = dout; p.B = bt; p.C = dx;
    p.M = B * L * Wd; p.N = 2 * C; p.K = ntl * ntw * 2 * F;
    p.lda = 0; p.ldb = p.K; p.ldc = 2 * C;
    p.klen = round_k(p.K);
    p.vecA = 1; p.vecB = 1;
    p.pg.L = Lo; p.pg.Wd = Wo; p.pg.c2 = 2 * F; p.pg.Lo = L; p.pg.Wo = Wd; p.pg.ntl = ntl; p.pg.ntw = ntw;
    p.pg.sL = 1; p.pg.sW = 1; p.pg.l0 = -(tl0 - pl0) - (ntl - 1); p.pg.w0 = -(tw0 - pw0) - (ntw - 1);
    p.pg.isL = sL; p.pg.isW = sW;
    patch_div_magic(sL, p.pg.il_mul, p.pg.il_shift);
    patch_div_magic(sW, p.pg.iw_mul, p.pg.iw_shift);
    return launch_gemm<OP_KPATCH, OP_KCONTIG, 0, TAG_CCONV_BWD_X>(p, 1, s);
}
// ---- few-channel 1-D C-Conv: input, weight and bias gradient in one pass over dout (cconv1d_bwd.h) -------------------------
constexpr int kConv1dMaxBlocks = 1024;
static int conv1d_bwd_pl(int ntl, int sL) { return 62 * sL - ntl + 2; }          // positions per chunk: its rows fit 64 (cconv1d_bwd.h)
int dccn_cconv1d_bwd_supported(int B, int L, int C, int Lo, int ntl, int sL, int F) {
    if (B <= 0 || L <= 0 || C <= 0 || Lo <= 0 || ntl <= 0 || sL <= 0 || F <= 0) return 0;
    if ((C % 2) != 0 || (F != 32 && F != 64) || 2 * ntl * C > 30 || conv1d_bwd_pl(ntl, sL) < 1) return 0;
    if ((long long)B * L * C * 2 >= (1LL << 31) || (long long)B * Lo * F * 2 >= (1LL << 31)) return 0;
    return 1;
}
size_t dccn_cconv1d_bwd_workspace_size(int F) {
    if (F <= 0) return 0;
    size_t o = 0;
    o = carve_size(o, (size_t)kConv1dMaxBlocks * 32 * 2 * F * sizeof(float));
    o = carve_size(o, (size_t)kConv1dMaxBlocks * 2 * F * sizeof(float));
    return align_up(o, 256);
}
int dccn_cconv1d_bwd(const float* x, const float* dout, const float* w, float* dx, float* dw, float* dbias, int B, int L, int C,
                     int Lo, int ntl, int tl0, int sL, int pl0, int F, void* workspace, size_t workspace_bytes,
                     dccn_stream_t stream) {
    if (!x || !dout || !w || !dx || !dw || !dccn_cconv1d_bwd_supported(B, L, C, Lo, ntl, sL, F) || !aligned16(dout) || !aligned16(w))
        return DCCN_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < dccn_cconv1d_bwd_workspace_size(F)) return DCCN_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    Carver c(workspace, workspace_bytes);
    Conv1dBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.dout = dout; a.w = w; a.dx = dx;
    a.slabs = c.take<float>((size_t)kConv1dMaxBlocks * 32 * 2 * F);
    a.colsum = c.take<float>((size_t)kConv1dMaxBlocks * 2 * F);
    a.B = B; a.L = L; a.C2 = 2 * C; a.Lo = Lo; a.nt = ntl; a.F2 = 2 * F; a.NC = 2 * ntl * C;
    a.o = tl0 - pl0; a.s = sL;
    a.PL = conv1d_bwd_pl(ntl, sL);
    a.nch = ceil_div(L, a.PL);
    const long long total = (long long)B * a.nch;
    int grid = 2 * kCUs;
    if (grid > kConv1dMaxBlocks) grid = kConv1dMaxBlocks;
    if (grid > total) grid = (int)total;
    DCCN_NO_CHAINS();
    if (F == 64) {
        auto kern = cconv1d_bwd_fused_kernel<128>;
        DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), conv1d_bwd_smem_bytes<128>()));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), conv1d_bwd_smem_bytes<128>(), s, a);
    } else {
        auto kern = cconv1d_bwd_fused_kernel<64>;
        DCCN_TRY(set_max_dynamic_smem(reinterpret_cast<const void*>(kern), conv1d_bwd_smem_bytes<64>()));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), conv1d_bwd_smem_bytes<64>(), s, a);
    }
    DCCN_LAUNCH_CHECK();
    const int kin = ntl * C;
    const int fold_blocks = ceil_div(kin * F + F, kRedLanes);
    hipLaunchKernelGGL(cconv_fold_kernel, dim3(fold_blocks), dim3(256), 0, s, a.slabs, grid, (long long)32 * 2 * F, a.colsum, dw, dbias,
                       kin, F);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}
int dccn_cconv_col2im(const float* drows, float* dx, int B, int L, int Wd, int C, int Lo, int Wo, int ntl, int ntw, int tl0,
                      int tw0, int sL, int sW, int pl0, int pw0, dccn_stream_t stream) {
    const Im2colGeom g{B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0};
    if (!drows || !dx || !im2col_geom_ok(g)) return DCCN_ERR_INVALID_ARG;
    const long long n = (long long)B * L * Wd * C;
    hipLaunchKernelGGL(col2im_kernel, dim3((unsigned)ceil_div_ll(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)drows, (float2*)dx, g, n);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// ---- in-graph AWGN monitor branch (`iq_tx:0`, `iq_rx:0`, `noise_power:0`) -----------------------------------
size_t dccn_ingraph_awgn_workspace_size(int frames, int pairs_per_frame) {
    if (frames <= 0 || pairs_per_frame <= 0) return 0;
    size_t o = 0;
    o = carve_size(o, (size_t)frames * pairs_per_frame * 2 * sizeof(float));                   // clipped
    o = carve_size(o, (size_t)frames * pairs_per_frame * 2 * sizeof(float));                   // re-normalised
    o = carve_size(o, norm_ws_bytes(frames, 2 * pairs_per_frame));
    o = carve_size(o, dccn_clip_power_workspace_size((long long)frames * pairs_per_frame));
    o = carve_size(o, (size_t)frames * ceil_div(pairs_per_frame, 256) * sizeof(double));
    o = carve_size(o, 256);
    return align_up(o, 256);
}
int dccn_ingraph_awgn(const float* x_norm, const float* snr_db, float* tx_signal, uint16_t* iq_tx_f16,
                      uint16_t* iq_rx_f16, float* noise_power, int frames, int pairs_per_frame, float peak,
                      unsigned long long seed, unsigned offset, void* workspace, size_t workspace_bytes,
                      dccn_stream_t stream) {
    if (!x_norm || !snr_db || !noise_power || frames <= 0 || pairs_per_frame <= 0) return DCCN_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < dccn_ingraph_awgn_workspace_size(frames, pairs_per_frame)) return DCCN_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const long long n_pairs = (long long)frames * pairs_per_frame;
    Carver c(workspace, workspace_bytes);
    float* clipped = c.take<float>((size_t)n_pairs * 2);
    float* xn = c.take<float>((size_t)n_pairs * 2);
    const size_t nws = norm_ws_bytes(frames, 2 * pairs_per_frame);
    void* ws_norm = c.take<char>(nws);
    const size_t cws = dccn_clip_power_workspace_size(n_pairs);
    void* ws_clip = c.take<char>(cws);
    const int gx = ceil_div(pairs_per_frame, 256);
    double* partial = c.take<double>((size_t)frames * gx);
    float* scratch_pw = c.take<float>(64);            // complex_clip's power output is `tx_power:0`, served elsewhere
    float* clip_dst = tx_signal ? tx_signal : clipped;
    DCCN_TRY(dccn_clip_power(x_norm, clip_dst, scratch_pw, n_pairs, peak, ws_clip, cws, stream));
    dccn_adam_hparams hp;
    memset(&hp, 0, sizeof(hp));
    DCCN_TRY(norm_impl(clip_dst, xn, nullptr, nullptr, false, nullptr, frames, 2 * pairs_per_frame, 1e-8f, peak, nullptr, hp,
                       ws_norm, nws, s));
    hipLaunchKernelGGL(ingraph_awgn_kernel, dim3(gx, frames), dim3(256), 0, s, (const float2*)clip_dst, (const float2*)xn,
                       snr_db, reinterpret_cast<__half2*>(iq_tx_f16), reinterpret_cast<__half2*>(iq_rx_f16), partial,
                       pairs_per_frame, offset, seed);
    DCCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, (const double*)partial, frames * gx, (double)n_pairs,
                       noise_power);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

// ---- CRC32C (host) --------------------------------------------------------------------------------------
static uint32_t g_crc32c_table[8][256];
static bool g_crc32c_ready = false;
static void crc32c_init() {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
        g_crc32c_table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t)
            g_crc32c_table[t][i] = (g_crc32c_table[t - 1][i] >> 8) ^ g_crc32c_table[0][g_crc32c_table[t - 1][i] & 0xffu];
    g_crc32c_ready = true;
}
uint32_t dccn_crc32c(uint32_t crc, const void* data, size_t n) {
    if (!g_crc32c_ready) crc32c_init();
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = crc ^ 0xffffffffu;
    while (n >= 8) {                                     // slicing-by-8
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = g_crc32c_table[7][lo & 0xffu] ^ g_crc32c_table[6][(lo >> 8) & 0xffu] ^ g_crc32c_table[5][(lo >> 16) & 0xffu] ^
            g_crc32c_table[4][lo >> 24] ^ g_crc32c_table[3][hi & 0xffu] ^ g_crc32c_table[2][(hi >> 8) & 0xffu] ^
            g_crc32c_table[1][(hi >> 16) & 0xffu] ^ g_crc32c_table[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = g_crc32c_table[0][(c ^ *p++) & 0xffu] ^ (c >> 8);
    return c ^ 0xffffffffu;
}

}  // extern "C"
