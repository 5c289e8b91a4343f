// libdccn.so -- C ABI over the gfx950 kernels (see include/dccn.h for the contract and the
// reference call site each entry point replaces).
#include "abi_impl.h"

namespace dccn {
thread_local int g_last_hip_error = 0;
thread_local ChainCtx tl_chain = {1, {{0, 0, 0, 0, 0, 0, 0, 0}}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
StepTraceState g_step_trace;
thread_local unsigned long long* tl_stamp = nullptr;

thread_local int tl_whole_k = -1;
thread_local int tl_tune_depth = 0;
thread_local int tl_tune_vals[TUNE_COUNT];
TuneTable g_tune = {{{9}, {7}, {7}, {7}, {0}, {0}, {0}, {1}, {1}, {1}, {1}, {1}, {3}, {1}, {14}, {0}, {0}, {1}, {0}, {2}, {1}, {1}, {0}, {0}, {1}, {2}, {0}, {1}}};
std::atomic<int> g_whole_k_global{1};

// ---------------------------------------------------------------------------------------
// R0
// ---------------------------------------------------------------------------------------
static int norm_grid_x(int cols) { return ceil_div(ceil_div(cols, 4), 64); }
static int norm_grid_y(int batch) { return ceil_div(batch, kNormRowsPerBlock); }

int norm_fused_blocks(int cols) { return ceil_div(ceil_div(cols, 4 * kNormFusedCG), 8) * 8; }
static size_t norm_power_slots(int batch, int cols) {
    const size_t a = (size_t)norm_grid_x(cols) * norm_grid_y(batch), b = (size_t)norm_fused_blocks(cols);
    return a > b ? a : b;
}
size_t norm_ws_bytes(int batch, int cols) {
    size_t o = 0;
    o = carve_size(o, (size_t)kNormRowChunks * cols * 2 * sizeof(double));
    o = carve_size(o, 2 * norm_power_slots(batch, cols) * sizeof(double));      // two slots: dccn_rx_buffers.norm_slot
    return align_up(o, 256);
}

bool norm_fused_ok(const float* x, const float* y, int batch, int cols) {
    return (cols % 4 == 0) && batch <= 128 * kNormFusedRPT && aligned16(x) && aligned16(y);
}


// want_power: also emit the per-block partial sums of the clipped power (R8); adam != nullptr: the
// optimizer bookkeeping of the fused training step rides on the first kernel
// where norm_impl leaves the R8 partial sums for a [batch, cols] input in workspace `ws` (no launch)
void norm_power_partials(int batch, int cols, void* ws, size_t ws_bytes, const float* x, const float* y,
                                PowerPartials* pp, int slot) {
    Carver c(ws, ws_bytes);
    c.take<double>((size_t)kNormRowChunks * cols * 2);
    pp->partial = c.take<double>(2 * norm_power_slots(batch, cols)) + (slot ? norm_power_slots(batch, cols) : 0);
    pp->n = norm_fused_ok(x, y, batch, cols) ? norm_fused_blocks(cols) : norm_grid_x(cols) * norm_grid_y(batch);
    pp->denom = (double)batch * (double)(cols / 2);
}

int norm_impl(const float* x, float* y, float* mean, float* var, bool want_power, PowerPartials* pp, int batch,
                     int cols, float eps, float peak, dccn_adam_state* adam, dccn_adam_hparams hp, void* ws,
                     size_t ws_bytes, hipStream_t s, int slot) {
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
GemmParams gp_zero() {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.stamp = tl_stamp;         // non-null only while a StepTraceScope names the launch being built (common.h)
    return p;
}
int round_k(int K) { return ceil_div(K, 64) * 64; }
// the fast GEMM loaders use 32-bit byte offsets from a uniform base: operand must be < 2 GiB
static bool small_enough(long long rows, long long ld) { return rows * ld * 4 < (1LL << 31); }

// ldx > 0: x is a column window of a wider row-major matrix (row stride ldx)
// act = 2: y = tanh(x.w + bias) when the launch plan has the stage (few-row 16x64 tiles); *act_done tells
// act = 5 (+ aux = the received cells, out2 / out3): the output is a channel estimate; eq and corr of model.py:431-438
// leave the same launch
int dense_fwd_impl(const float* x, const float* w, const float* bias, float* y, int M, int K, int N,
                          hipStream_t s, int ldx, int act, bool* act_done, const float* aux,
                          float* out2, float* out3) {
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

int dense_bwd_x_impl(const float* dy, const float* w, float* dx, int M, int K, int N, hipStream_t s) {
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
size_t splitk_ws_bytes(int Mo, int No, int Kr) {
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
int dense_bwd_w_impl(const float* x, const float* dy, float* dw, float* dbias, int M, int K, int N, void* ws,
                            size_t ws_bytes, hipStream_t s, DeferredSlabs* defer, int ldx) {
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
int dense_bwd_grouped_impl(const float* x, const float* dy, const float* w, float* dx, float* dw, float* dbias,
                                  int M, int K, int N, void* ws, size_t ws_bytes, hipStream_t s, DeferredSlabs* defer,
                                  int actx, const float* aux, bool* act_done,
                                  float* split_dst, long long split_pairs_gc, bool* split_done) {
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

int cconv_fwd_impl(const float* x, const float* w, const float* bias, float* out, int rows, int kin, int F,
                          hipStream_t s, int ldx) {
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

int cconv_bwd_x_impl(const float* dout, const float* w, float* dx, int rows, int kin, int F, hipStream_t s,
                            int ldc) {
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
size_t cconv_bw_ws_bytes(int rows, int kin, int F) {
    const size_t legacy = splitk_ws_bytes(2 * kin, 2 * F, rows);
    if (4LL * kin * F > 512 * 512) return legacy;
    size_t o = 0;
    o = carve_size(o, (size_t)kCconvBwMaxSplits * 4 * kin * F * sizeof(float));
    o = carve_size(o, (size_t)kCconvBwMaxSplits * 2 * F * sizeof(float));
    o = align_up(o, 256);
    return o > legacy ? o : legacy;
}


int cconv_bwd_w_impl(const float* x, const float* dout, float* dw, float* dbias, int rows, int kin, int F,
                            void* ws, size_t ws_bytes, hipStream_t s, const TailFinalizeArgs* fin,
                            FoldDefer* defer) {
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
bool cconv_pair_ok(const float* x, const float* w, const float* o, int rows, int kin, int F, long long gx, long long gw,
                          long long go) {
    return (kin % 2 == 0) && (F % 2 == 0) && (kin % 32 == 0) && (F % 32 == 0) && aligned16(x) && aligned16(w) && aligned16(o) &&
           (gx % 4 == 0) && (gw % 4 == 0) && (go % 2 == 0) && small_enough(rows, 4LL * (kin > F ? kin : F)) &&
           small_enough(kin, 2LL * F) && (long long)ceil_div(rows, 128) * ceil_div(2 * F, 128) < 2 * kCUs;
}
int cconv_fwd_grouped_impl(const float* x, const float* w, const float* bias, float* out, int rows, int kin, int F,
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
size_t cconv_bwd_grouped_ws_bytes(int rows, int kin, int F, int groups) {
    return (size_t)groups * cconv_bw_ws_bytes(rows, kin, F);
}
int cconv_bwd_grouped_impl(const float* x, const float* dout, const float* w, float* dx, int rows, int kin, int F,
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
size_t tail_ws_bytes(long long cells, int nbits) {
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
int tail_impl(bool bwd, const float* z, const int32_t* bits, const float* tailp, float* prob,
                     dccn_metrics* metrics, float* dz, float* dtailp, long long cells, int nbits,
                     const PowerPartials* pp, float* power_out, void* ws, size_t ws_bytes, hipStream_t s,
                     TailFinalizeArgs* defer) {
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
size_t dense_tail_ws_bytes(int M, int N, int nbits) {
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
bool dense_tail_ok(const float* x, const float* w, int M, int K, int N, int nbits) {
    return dense_tail_shape_ok(M, K, N, nbits) && aligned16(x) && aligned16(w);
}
// which steps take the fused launch for 8-QAM / 16-QAM (knob 13: bit 0 = the lane-per-cell forms -- nbits 3, and nbits 4
// evaluation; bit 1 = the quad-lane form of 16-QAM training).  The operator dccn_dense_tail_* itself accepts every nbits.
// bit 2: BPSK / QPSK steps of LARGE layers (>= two rounds of 128x128 tiles) run the dense forward on the 128x128x32 tile
// family and the tail as its own launch.
bool dense_tail_planned(int nbits, bool train, int M, int N) {
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
int dense_tail_impl(bool bwd, const float* x, const float* w, const float* bias, float* z, const int32_t* bits,
                           const float* tailp, float* prob, dccn_metrics* metrics, float* dz, float* dtailp, int M,
                           int K, int N, int nbits, const PowerPartials* pp, float* power_out, void* ws, size_t ws_bytes,
                           hipStream_t s, TailFinalizeArgs* defer) {
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
int adam_impl(float* param, const float* grad, float* m, float* v, const float* reg_coef,
                     const float* reg_gate, dccn_adam_state* st, dccn_adam_hparams hp, long long n, hipStream_t s,
                     bool prep) {
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

RxLayout rx_layout(const dccn_rx_shape* sh) {
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
bool gen_static_ok(const dccn_gen_static* g) {
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
int gen_static_args(const dccn_gen_static* g, GenStaticArgs* out) {
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
int gen_static_launch(const dccn_gen_static* g, hipStream_t s, const GenChainScalars* chains) {
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

int eq_monitor_blocks(int B, int K) {
    long long n = ceil_div_ll((long long)B * K * 2, 256);
    if (n > 256) n = 256;
    return (int)(n < 1 ? 1 : n);
}
}  // namespace dccn

using namespace dccn;

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
// ---- HIP-event timers on the caller's stream ------------------------------------------------------------------
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

}  // extern "C"
