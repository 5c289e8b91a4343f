// libdccn.so -- the equaliser: fused step, chain groups, stage operators, monitors (see abi_impl.h for how the library is cut into units)
#include "abi_impl.h"

namespace dccn {
#include "eq_step.h"
}  // namespace dccn

using namespace dccn;

extern "C" {

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

}  // extern "C"
