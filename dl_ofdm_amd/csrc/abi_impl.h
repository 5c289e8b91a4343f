// libdccn.so -- what the translation units of the C ABI share (round 6: dccn_abi.hip was one 3100-line unit):
//   dccn_abi.hip        globals, tuning table, the operators' launch planning (*_impl), the basic receiver's step and its entry points
//   dccn_abi_eq.hip     the equaliser step (eq_step.h), chain groups, equaliser stage operators, monitors
//   dccn_abi_gen.hip    device-side generator, classical receivers, in-graph AWGN branch
//   dccn_abi_conv.hip   general-k complex convolutions (patch gather, implicit GEMMs, few-channel kernels), CRC32C
// Kernels live in the headers; a non-template kernel is `static`, so a unit compiles only the kernels it launches.
#pragma once
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
extern thread_local int tl_whole_k;
extern thread_local int tl_tune_depth;
extern thread_local int tl_tune_vals[TUNE_COUNT];
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
extern TuneTable g_tune;        // (defined, with the defaults, in dccn_abi.hip)

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

// ---- shared between the units: where the operators leave work for a later launch, layouts ------------------------------
struct PowerPartials {      // where normalise left the R8 partial sums (finished by a later kernel)
    const double* partial;
    int n;
    double denom;
};

struct DeferredSlabs {
    const float* dw_slabs;
    const float* db_slabs;
    int splits;
};

struct FoldDefer {          // the fold left to the optimizer kernel (fused training step)
    const float* slabs; const float* colsum;
    int splits; long long slab;
};

struct RxLayout {
    long long o_conv_w, o_conv_b, o_dense_w, o_dense_b, o_tail, total;
    int rows, cols, dK, dN;
    long long cells;
    size_t ws_norm, ws_tail, ws_dense_bw, ws_conv_bw, ws_sync;
};

// ---- launch planning of the operators (defined in dccn_abi.hip) -------------------------------------------------------
int adam_impl(float* param, const float* grad, float* m, float* v, const float* reg_coef, const float* reg_gate, dccn_adam_state* st, dccn_adam_hparams hp, long long n, hipStream_t s, bool prep = true);
int cconv_bwd_grouped_impl(const float* x, const float* dout, const float* w, float* dx, int rows, int kin, int F, int groups, long long gx, long long gd, long long gw, void* ws, size_t ws_bytes, FoldDefer* defer, hipStream_t s);
size_t cconv_bwd_grouped_ws_bytes(int rows, int kin, int F, int groups);
int cconv_bwd_w_impl(const float* x, const float* dout, float* dw, float* dbias, int rows, int kin, int F, void* ws, size_t ws_bytes, hipStream_t s, const TailFinalizeArgs* fin = nullptr, FoldDefer* defer = nullptr);
int cconv_bwd_x_impl(const float* dout, const float* w, float* dx, int rows, int kin, int F, hipStream_t s, int ldc = 0);
int cconv_fwd_grouped_impl(const float* x, const float* w, const float* bias, float* out, int rows, int kin, int F, int groups, long long gx, long long gw, long long gb, bool join_pairs, hipStream_t s);
int cconv_fwd_impl(const float* x, const float* w, const float* bias, float* out, int rows, int kin, int F, hipStream_t s, int ldx = 0);
bool cconv_pair_ok(const float* x, const float* w, const float* o, int rows, int kin, int F, long long gx, long long gw, long long go);
int dense_bwd_grouped_impl(const float* x, const float* dy, const float* w, float* dx, float* dw, float* dbias, int M, int K, int N, void* ws, size_t ws_bytes, hipStream_t s, DeferredSlabs* defer, int actx = 1, const float* aux = nullptr, bool* act_done = nullptr, float* split_dst = nullptr, long long split_pairs_gc = 0, bool* split_done = nullptr);
int dense_bwd_w_impl(const float* x, const float* dy, float* dw, float* dbias, int M, int K, int N, void* ws, size_t ws_bytes, hipStream_t s, DeferredSlabs* defer = nullptr, int ldx = 0);
int dense_bwd_x_impl(const float* dy, const float* w, float* dx, int M, int K, int N, hipStream_t s);
int dense_fwd_impl(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, hipStream_t s, int ldx = 0, int act = 1, bool* act_done = nullptr, const float* aux = nullptr, float* out2 = nullptr, float* out3 = nullptr);
int dense_tail_impl(bool bwd, const float* x, const float* w, const float* bias, float* z, const int32_t* bits, const float* tailp, float* prob, dccn_metrics* metrics, float* dz, float* dtailp, int M, int K, int N, int nbits, const PowerPartials* pp, float* power_out, void* ws, size_t ws_bytes, hipStream_t s, TailFinalizeArgs* defer = nullptr);
bool dense_tail_ok(const float* x, const float* w, int M, int K, int N, int nbits);
bool dense_tail_planned(int nbits, bool train, int M = 0, int N = 0);
size_t dense_tail_ws_bytes(int M, int N, int nbits);
int eq_monitor_blocks(int B, int K);
int gen_static_args(const dccn_gen_static* g, GenStaticArgs* out);
int gen_static_launch(const dccn_gen_static* g, hipStream_t s, const GenChainScalars* chains = nullptr);
bool gen_static_ok(const dccn_gen_static* g);
int norm_fused_blocks(int cols);
bool norm_fused_ok(const float* x, const float* y, int batch, int cols);
int norm_impl(const float* x, float* y, float* mean, float* var, bool want_power, PowerPartials* pp, int batch, int cols, float eps, float peak, dccn_adam_state* adam, dccn_adam_hparams hp, void* ws, size_t ws_bytes, hipStream_t s, int slot = 0);
void norm_power_partials(int batch, int cols, void* ws, size_t ws_bytes, const float* x, const float* y, PowerPartials* pp, int slot = 0);
size_t norm_ws_bytes(int batch, int cols);
RxLayout rx_layout(const dccn_rx_shape* sh);
size_t splitk_ws_bytes(int Mo, int No, int Kr);
int tail_impl(bool bwd, const float* z, const int32_t* bits, const float* tailp, float* prob, dccn_metrics* metrics, float* dz, float* dtailp, long long cells, int nbits, const PowerPartials* pp, float* power_out, void* ws, size_t ws_bytes, hipStream_t s, TailFinalizeArgs* defer = nullptr);
size_t tail_ws_bytes(long long cells, int nbits);
size_t cconv_bw_ws_bytes(int rows, int kin, int F);
GemmParams gp_zero();
int round_k(int K);

}  // namespace dccn

// (the receiver's and the equaliser's captured steps share the handle type)
struct dccn_rx_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
    hipStream_t cap;     // capture happens on a private stream: the caller's may be the (uncapturable) null stream
    hipStream_t side;
    hipEvent_t ev_fork, ev_join;
};
