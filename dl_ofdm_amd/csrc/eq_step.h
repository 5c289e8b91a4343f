// Fused equaliser transfer-learning step: dev/py/ofdmreceiver_np_mp.py:283-330 as one
// pre-planned launch sequence (capturable into a hipGraph) over flat arenas --
//   R0 normalise -> equalizer_ofdm (model.py:349-478) -> frozen basic receiver -> loss/BER
//   -> backward to the Equalizer/* variables only -> TF Adam on the equaliser arena.
// Included inside namespace dccn of dccn_abi_eq.hip, behind abi_impl.h (the operators' *_impl launch planning it is built from).
// FLAGS.cp=False (model.py:364-366, 1236-1240): the equaliser's first dense layer and the receiver's C-Conv read
// the K-sample window behind the cyclic prefix of [.., n_sc, 2] rows -- a column window of the same buffers.

static bool eq_shape_ok(const dccn_eq_shape* sh) {
    return sh && sh->batch > 0 && sh->S > 0 && sh->K > 0 && sh->CP >= 0 && sh->F > 0 && sh->D > 0 && sh->nbits >= 1 &&
           sh->nbits <= 4 && sh->pilot_size > 0 && (sh->cp == 0 || sh->cp == 1);
}

struct EqDims {
    int B, S, K, nsc, R, SK2, Pp, F, D, cp, win;     // win: float offset of the post-CP window inside a row
    long long o[21];        // parameter offsets, TF creation order (dense, conv3d, dense_1..4, conv3d_1..3, dense_5)
    long long sz[20];       // element counts (o[i+1] - o[i] may include up to three floats of alignment padding)
};
static EqDims eq_dims(const dccn_eq_shape* sh) {
    EqDims d;
    d.B = sh->batch; d.S = sh->S; d.K = sh->K; d.nsc = sh->K + sh->CP; d.R = d.B * d.S;
    d.SK2 = d.S * d.K * 2; d.Pp = 2 * sh->pilot_size; d.F = sh->F; d.D = sh->D;
    d.cp = sh->cp; d.win = sh->cp ? 0 : 2 * sh->CP;
    const long long K2 = 2LL * d.K, N2 = 2LL * d.nsc, SK2 = d.SK2;
    const long long sizes[20] = {(sh->cp ? N2 : K2) * K2, K2,   // dense (input: whole row, or the window)
                                 (long long)d.K * K2, K2,     // conv3d     (1,K) -> K filters
                                 SK2 * d.Pp, d.Pp,            // dense_1    pilot extraction
                                 d.Pp * SK2, SK2,             // dense_2
                                 SK2 * SK2, SK2,              // dense_3
                                 SK2 * SK2, SK2,              // dense_4    (tanh)
                                 SK2, 2,                      // conv3d_1   (S,K) -> 1 filter
                                 (long long)d.K * K2, K2,     // conv3d_2   (corr)
                                 (long long)d.K * K2, K2,     // conv3d_3   (equalized)
                                 4LL * d.K * N2, N2};         // dense_5
    // every tensor starts on a 16-byte boundary (padding floats stay zero: zero gradient, zero Adam slots): the
    // two-float bias of conv3d_1 used to shift conv3d_2 / conv3d_3 / dense_5 off it, and their five GEMMs onto the
    // element-wise masked loaders -- 45 us of the 374 us step at 73 frames (profiles/r03_eq73_kernel_stats.txt)
    d.o[0] = 0;
    for (int i = 0; i < 20; ++i) { d.o[i + 1] = (d.o[i] + sizes[i] + 3) / 4 * 4; d.sz[i] = sizes[i]; }
    return d;
}

// weight-gradient workspaces, one per trainable layer (their slabs stay live until the optimizer launch reduces them)
enum EqLayer : int { EQL_DENSE = 0, EQL_CONV, EQL_DENSE1, EQL_DENSE2, EQL_DENSE3, EQL_DENSE4, EQL_SMOOTH, EQL_PAIR, EQL_DENSE5,
                     EQL_COUNT };
struct EqWs {
    void *ws_norm, *ws_tail;
    void* ws_l[EQL_COUNT];
    size_t n_norm, n_tail, n_l[EQL_COUNT];
    float *x_norm, *ln, *t1, *y, *d1, *d2, *d3, *d4, *T, *be, *eq, *corr, *eqc, *corc, *cat, *fft, *z;
    // training only
    float *dz, *dfft, *dout, *dcat, *deqc, *dcorc, *deq, *dcorr, *dy, *dh, *dT, *dbe, *dd4, *dd3, *dd2, *dd1,
        *dflat, *dt1, *dtail;
    float* bn_part;          // per-block weight-gradient partials of the fused bottleneck backward (eq_bottleneck.h)
};
static void eq_layer_ws(const EqDims& d, size_t (&n)[EQL_COUNT]) {
    n[EQL_DENSE] = splitk_ws_bytes(d.cp ? 2 * d.nsc : 2 * d.K, 2 * d.K, d.R);
    n[EQL_CONV] = cconv_bwd_grouped_ws_bytes(d.R, d.K, d.K, 1);
    n[EQL_DENSE1] = splitk_ws_bytes(d.SK2, d.Pp, d.B);
    n[EQL_DENSE2] = splitk_ws_bytes(d.Pp, d.SK2, d.B);
    n[EQL_DENSE3] = n[EQL_DENSE4] = n[EQL_SMOOTH] = splitk_ws_bytes(d.SK2, d.SK2, d.B);
    n[EQL_PAIR] = cconv_bwd_grouped_ws_bytes(d.R, d.K, d.K, 2);           // corr and eq C-Convs
    n[EQL_DENSE5] = splitk_ws_bytes(4 * d.K, 2 * d.nsc, d.R);
}
static void eq_carve(Carver& c, const dccn_eq_shape* sh, const EqDims& d, bool train, EqWs& w) {
    const size_t B = d.B, R = d.R, SK2 = d.SK2, K2 = 2 * (size_t)d.K, N2 = 2 * (size_t)d.nsc;
    w.n_norm = norm_ws_bytes(d.B, d.S * d.nsc * 2);
    // the layout does not depend on the modulation (the tail's slabs and gradient are sized for 16-QAM): chains of different
    // modulations carried by one launch sequence (dccn_eq_train_step_grouped) see the same offsets
    w.n_tail = 0;
    for (int nb = 1; nb <= 4; ++nb) {
        const size_t t = tail_ws_bytes((long long)d.B * d.D, nb), f = dense_tail_ws_bytes(d.B, 2 * d.D, nb);
        if (t > w.n_tail) w.n_tail = t;
        if (f > w.n_tail) w.n_tail = f;
    }
    w.ws_norm = c.take<char>(w.n_norm);
    w.ws_tail = c.take<char>(w.n_tail);
    eq_layer_ws(d, w.n_l);
    for (int l = 0; l < EQL_COUNT; ++l) w.ws_l[l] = train ? c.take<char>(w.n_l[l]) : nullptr;
    w.x_norm = c.take<float>(R * N2);
    w.ln = c.take<float>(R * N2);
    w.t1 = c.take<float>(R * K2);
    w.y = c.take<float>(R * K2);
    w.d1 = c.take<float>(B * d.Pp);
    w.d2 = c.take<float>(B * SK2);
    w.d3 = c.take<float>(B * SK2);
    w.d4 = c.take<float>(B * SK2);
    w.T = c.take<float>(SK2 * SK2);
    w.be = c.take<float>(SK2);
    w.eq = c.take<float>(B * SK2);
    w.corr = c.take<float>(B * SK2);
    w.eqc = c.take<float>(R * K2);
    w.corc = c.take<float>(R * K2);
    w.cat = c.take<float>(R * 2 * K2);
    w.fft = c.take<float>(R * 2 * (size_t)d.F);
    w.z = c.take<float>(B * 2 * (size_t)d.D);
    if (!train) return;
    w.dz = c.take<float>(B * 2 * (size_t)d.D);
    w.dfft = c.take<float>(R * 2 * (size_t)d.F);
    w.dout = c.take<float>(R * N2);
    w.dcat = c.take<float>(R * 2 * K2);
    w.deqc = c.take<float>(R * K2);
    w.dcorc = c.take<float>(R * K2);
    w.deq = c.take<float>(B * SK2);
    w.dcorr = c.take<float>(B * SK2);
    w.dy = c.take<float>(B * SK2);
    w.dh = c.take<float>(B * SK2);
    w.dT = c.take<float>(SK2 * SK2);
    w.dbe = c.take<float>(SK2);
    w.dd4 = c.take<float>(B * SK2);
    w.dd3 = c.take<float>(B * SK2);
    w.dd2 = c.take<float>(B * SK2);
    w.dd1 = c.take<float>(B * d.Pp);
    w.dflat = c.take<float>(B * SK2);
    w.dt1 = c.take<float>(R * K2);
    w.dtail = c.take<float>(tail_param_count(4));
    w.bn_part = c.take<float>((d.Pp == 16 || d.Pp == 32) ? eq_bottleneck_part_floats(d.B, d.SK2, d.Pp) : 0);
}
static size_t eq_ws_bytes(const dccn_eq_shape* sh, int train) {
    const EqDims d = eq_dims(sh);
    Carver c(nullptr, 0);
    EqWs w;
    eq_carve(c, sh, d, train != 0, w);
    return align_up(c.off, 256);
}

// dense backward with reduced outputs: dx (nullable: weight gradient only), dw, dbias
// actx / aux / act_done: optional element-wise stage on the dX store (dense_bwd_grouped_impl)
// keep: leave split-K slabs un-reduced and report them (the optimizer launch sums them, eq_opt.h)
// split_dst / split_gc / split_done: dx is the gradient of a concat of two IQ-pair streams; when the launch plan has the
// stage, the streams' own buffers are written instead of dx (dense_bwd_grouped_impl)
static int dense_bwd_full_impl(const float* x, const float* dy, const float* w, float* dx, float* dw, float* dbias,
                               int M, int K, int N, void* ws, size_t ws_bytes, hipStream_t s, int actx = 1,
                               const float* aux = nullptr, bool* act_done = nullptr, DeferredSlabs* keep = nullptr,
                               float* split_dst = nullptr, long long split_gc = 0, bool* split_done = nullptr) {
    if (act_done) *act_done = false;
    if (split_done) *split_done = false;
    DeferredSlabs ds;
    if (!dx) {
        DCCN_TRY(dense_bwd_w_impl(x, dy, dw, dbias, M, K, N, ws, ws_bytes, s, keep));
        return DCCN_OK;
    }
    DCCN_TRY(dense_bwd_grouped_impl(x, dy, w, dx, dw, dbias, M, K, N, ws, ws_bytes, s, &ds, actx, aux, act_done,
                                    split_dst, split_gc, split_done));
    if (keep) {
        *keep = ds;
        return DCCN_OK;
    }
    if (ds.dw_slabs) {
        const long long n = (long long)K * N;
        if (dbias && ds.db_slabs)
            DCCN_TRY(launch_splitk_reduce2(ds.dw_slabs, ds.splits, n, dw, n, ds.db_slabs, (long long)N, dbias, (long long)N, s));
        else
            DCCN_TRY(launch_splitk_reduce(ds.dw_slabs, ds.splits, n, dw, n, s));
    }
    return DCCN_OK;
}

// optimizer jobs of a dense layer (kernel variable i, bias variable i + 1)
// (uni: the caller's reg_coef is one value over each dense kernel / bias -- dccn_eq_buffers.reg_uniform)
// kernel_done: the kernel variable was updated by a rider of an earlier launch (EqRideArgs)
static void eq_opt_dense(EqOptBuilder& ob, const EqDims& d, int i, const DeferredSlabs& ds, long long N, bool uni,
                         bool kernel_done = false) {
    if (kernel_done) {}
    else if (ds.dw_slabs) ob.slabs(d.o[i], d.sz[i], ds.dw_slabs, ds.splits, d.sz[i], uni);
    else ob.plain(d.o[i], d.sz[i], uni);
    if (ds.dw_slabs && ds.db_slabs) ob.slabs(d.o[i + 1], d.sz[i + 1], ds.db_slabs, ds.splits, N, uni);
    else ob.plain(d.o[i + 1], d.sz[i + 1], uni);
}

static int eq_step_impl(const dccn_eq_shape* sh, const dccn_eq_buffers* b, bool train, dccn_adam_hparams hp,
                        hipStream_t s) {
    if (!eq_shape_ok(sh) || !b) return DCCN_ERR_INVALID_ARG;
    const TuneScope tune(b->tuning);
    if (!b->x || !b->bits || !b->eq_params || !b->rx_params || !b->out_eq || !b->chest || !b->metrics)
        return DCCN_ERR_INVALID_ARG;
    if (train && (!b->eq_grads || !b->adam_m || !b->adam_v || !b->adam)) return DCCN_ERR_INVALID_ARG;
    if (!b->workspace || b->workspace_bytes < eq_ws_bytes(sh, train ? 1 : 0)) return DCCN_ERR_WORKSPACE;
    const EqDims d = eq_dims(sh);
    Carver c(b->workspace, b->workspace_bytes);
    EqWs w;
    eq_carve(c, sh, d, train, w);
    const float* P = b->eq_params;
    float* G = b->eq_grads;
    const int B = d.B, R = d.R, K = d.K, SK2 = d.SK2, K2 = 2 * d.K, N2 = 2 * d.nsc;
    const long long nBK = (long long)B * SK2;          // floats in a [B,S,K,2] tensor
    dccn_rx_shape rsh;
    rsh.batch = B; rsh.S = d.S; rsh.kin = d.cp ? d.nsc : K; rsh.F = d.F; rsh.D = d.D; rsh.nbits = sh->nbits;
    const int kin0 = d.cp ? N2 : K2;                   // K extent of the first dense layer
    const RxLayout L = rx_layout(&rsh);
    const float* Q = b->rx_params;
    float* h = b->chest;

    // round-3 plan (TUNE_EQ_REPLAN): merged element-wise launches, the corr / eq C-Conv pair as grouped launches with the
    // concat / split of model.py:456 in the GEMM stores, ONE job-table launch for every gradient reduction + Adam
    // TUNE_EQ_REPLAN: 0 launch-per-stage plan, 1 re-plan, 2 re-plan with every dense split-K sum as its own launch
    // (debugging), 3 re-plan without the fused pilot bottleneck (bit-identical gradients to plan 0)
    const int plan = g_tune[TUNE_EQ_REPLAN];
    const bool replan = plan != 0;
    const bool keep_slabs = plan == 1 || plan == 3;
    const bool bn = plan == 1 && eq_bottleneck_ok(B, SK2, d.Pp, w.y, P + d.o[4], P + d.o[6]) && aligned16(w.d1);
    const long long g_in = w.corr - w.eq;                                  // eq -> corr stride of the pair's tensors
    const long long g_w = d.o[14] - d.o[16], g_b = d.o[15] - d.o[17];      // conv3d_3 (eq) -> conv3d_2 (corr)
    const bool pair = replan && cconv_pair_ok(w.eq, P + d.o[16], w.cat, R, K, K, g_in, g_w, 2) && aligned16(w.corr) &&
                      (g_b % 2 == 0);

    // the next batch's generator as a rider of this step (dccn_eq_buffers.gen_next_rides): on the bottleneck backward launch
    // when the plan has it, else as a launch of its own right here (same batch either way)
    bool gen_done = false;
    const bool gen_wanted = train && b->gen_next_rides != 0;
    if (gen_wanted && b->x_next_virtual == nullptr) return DCCN_ERR_INVALID_ARG;
    const bool gen_rides = gen_wanted && bn && gen_static_ok(b->x_next_virtual);
    if (gen_wanted && !gen_rides) {
        if (tl_chain.G > 1) return DCCN_ERR_UNSUPPORTED;
        DCCN_TRY(gen_static_launch(b->x_next_virtual, s));
        gen_done = true;
    }
    // `input:0` (ofdmreceiver_np.py:128-137) + tx_power partials
    PowerPartials pp;
    // (training: the optimizer's per-step bookkeeping rides on this first launch)
    // pipelined: the previous step ran this launch for us on its optimizer launch (dccn_eq_buffers.x_next)
    const int ncols = d.S * N2;
    const bool rides = replan && train && norm_fused_ok(b->x, w.x_norm, B, ncols) && kNormFusedCG == 2;
    const bool pre = rides && b->x_prenormalised != 0;
    if (b->x_prenormalised != 0 && !pre) return DCCN_ERR_INVALID_ARG;
    const int nslot = b->norm_slot ? 1 : 0;
    if (pre) norm_power_partials(B, ncols, w.ws_norm, w.n_norm, b->x, w.x_norm, &pp, nslot);
    else DCCN_TRY(norm_impl(b->x, w.x_norm, nullptr, nullptr, b->tx_power != nullptr, &pp, B, ncols, 1e-9f, 8.0f,
                            train ? b->adam : nullptr, hp, w.ws_norm, w.n_norm, s, nslot));
    // model.py:363 layer_norm, :369 dense, :378 C-Conv "DFT"; the expansion of the :428 smoothing C-Conv (S x K, same)
    // into the block-Toeplitz matrix of a dense layer depends on the parameters only and shares the launch
    if (replan) {
        DCCN_LAUNCH_CHAINS_Z(eq_prep_kernel, dim3(B + ew_blocks_n((long long)SK2 * SK2)), dim3(256), 0, s, (const float*)w.x_norm,
                             w.ln, B, d.S * N2, 1e-12f, P + d.o[12], P + d.o[13], w.T, w.be, d.S, K,
                             pre ? b->adam : (dccn_adam_state*)nullptr, hp);
        DCCN_LAUNCH_CHECK();
    } else {
        DCCN_NO_CHAINS();
        hipLaunchKernelGGL(layer_norm_fwd_kernel, dim3(B), dim3(256), 0, s, (const float*)w.x_norm, w.ln, (float*)nullptr,
                           (float*)nullptr, d.S * N2, 1e-12f);
        DCCN_LAUNCH_CHECK();
    }
    DCCN_TRY(dense_fwd_impl(w.ln + d.win, P + d.o[0], P + d.o[1], w.t1, R, kin0, K2, s, N2));
    DCCN_TRY(cconv_fwd_impl(w.t1, P + d.o[2], P + d.o[3], w.y, R, K, K, s));
    // :394-426 pilot bottleneck
    if (bn) {                               // both layers of the bottleneck in one launch (eq_bottleneck.h)
        auto kern = d.Pp == 32 ? eq_bottleneck_fwd_kernel<2> : eq_bottleneck_fwd_kernel<1>;
        const int q = eq_bottleneck_q(B, SK2);
        DCCN_LAUNCH_CHAINS_Z(kern, dim3(ceil_div(SK2 / 16, q), ceil_div(B, 16)), dim3(256), 0, s, (const float*)w.y, P + d.o[4],
                             P + d.o[5], P + d.o[6], P + d.o[7], w.d1, w.d2, B, SK2, q);
        DCCN_LAUNCH_CHECK();
    } else {
        DCCN_TRY(dense_fwd_impl(w.y, P + d.o[4], P + d.o[5], w.d1, B, SK2, d.Pp, s));
        DCCN_TRY(dense_fwd_impl(w.d1, P + d.o[6], P + d.o[7], w.d2, B, d.Pp, SK2, s));
    }
    DCCN_TRY(dense_fwd_impl(w.d2, P + d.o[8], P + d.o[9], w.d3, B, SK2, SK2, s));
    {
        bool fused = false;                 // tanh in the GEMM's store when the plan has the stage (few-row batches)
        DCCN_TRY(dense_fwd_impl(w.d3, P + d.o[10], P + d.o[11], w.d4, B, SK2, SK2, s, 0, 2, &fused));
        if (!fused) {
            DCCN_NO_CHAINS();
            hipLaunchKernelGGL(tanh_fwd_kernel, dim3(ew_blocks_n(nBK)), dim3(256), 0, s, (const float*)w.d4, w.d4, nBK);
            DCCN_LAUNCH_CHECK();
        }
    }
    // :428 smoothing C-Conv as a dense layer -> channel estimate
    if (!replan) {
        DCCN_NO_CHAINS();
        hipLaunchKernelGGL(cconv2d_same_expand_kernel, dim3(ew_blocks_n((long long)SK2 * SK2)), dim3(256), 0, s,
                           P + d.o[12], P + d.o[13], w.T, w.be, d.S, K, d.S, K);
        DCCN_LAUNCH_CHECK();
    }
    // :431-438 equalise + autocorrelation ride on the store of this GEMM when the plan has the stage; the :465-475 pilot
    // monitor then runs on the optimizer launch (training) or as its own small launch (evaluation)
    const bool want_snr = b->snr_db && b->pilot_carriers && sh->P > 0;
    bool eq_fused = false, snr_pending = false;
    if (replan) DCCN_TRY(dense_fwd_impl(w.d4, w.T, w.be, h, B, SK2, SK2, s, 0, 5, &eq_fused, w.y, w.eq, w.corr));
    else DCCN_TRY(dense_fwd_impl(w.d4, w.T, w.be, h, B, SK2, SK2, s));
    if (eq_fused) {
        if (want_snr && train) {
            snr_pending = true;
        } else if (want_snr) {
            DCCN_NO_CHAINS();
            hipLaunchKernelGGL(pilot_snr_kernel, dim3(B), dim3(64), 0, s, (const float2*)w.eq, b->pilot_carriers, b->snr_db,
                               d.S, K, sh->P);
            DCCN_LAUNCH_CHECK();
        }
    } else if (replan && want_snr) {
        const int eb = (int)ew_blocks_n(nBK / 2);
        DCCN_NO_CHAINS();
        hipLaunchKernelGGL(equalize_fwd_snr_kernel, dim3(eb + ceil_div(B, 4)), dim3(256), 0, s, (const float2*)w.y,
                           (const float2*)h, (float2*)w.eq, (float2*)w.corr, nBK / 2, eb, b->pilot_carriers, b->snr_db, B,
                           d.S, K, sh->P);
        DCCN_LAUNCH_CHECK();
    } else {
        DCCN_NO_CHAINS();
        hipLaunchKernelGGL(equalize_fwd_kernel, dim3(ew_blocks_n(nBK / 2)), dim3(256), 0, s, (const float2*)w.y,
                           (const float2*)h, (float2*)w.eq, (float2*)w.corr, nBK / 2);
        DCCN_LAUNCH_CHECK();
        if (want_snr) {
            DCCN_NO_CHAINS();
            hipLaunchKernelGGL(pilot_snr_kernel, dim3(B), dim3(64), 0, s, (const float2*)w.eq, b->pilot_carriers, b->snr_db,
                               d.S, K, sh->P);
            DCCN_LAUNCH_CHECK();
        }
    }
    // :439-449 C-Conv "IDFT" of corr and eq, :456-463 concat + dense back to the receiver's input
    if (pair) {
        // both C-Convs in one grid; their stores interleave the two IQ-pair streams into cat = [.., K, (eq, corr)]
        DCCN_TRY(cconv_fwd_grouped_impl(w.eq, P + d.o[16], P + d.o[17], w.cat, R, K, K, 2, g_in, g_w, g_b, true, s));
    } else {
        DCCN_TRY(cconv_fwd_impl(w.corr, P + d.o[14], P + d.o[15], w.corc, R, K, K, s));
        DCCN_TRY(cconv_fwd_impl(w.eq, P + d.o[16], P + d.o[17], w.eqc, R, K, K, s));
        DCCN_NO_CHAINS();
        hipLaunchKernelGGL(concat_pairs_kernel, dim3(ew_blocks_n((long long)R * K)), dim3(256), 0, s, (const float2*)w.eqc,
                           (const float2*)w.corc, (float4*)w.cat, (long long)R * K);
        DCCN_LAUNCH_CHECK();
    }
    DCCN_TRY(dense_fwd_impl(w.cat, P + d.o[18], P + d.o[19], b->out_eq, R, 4 * K, N2, s));
    // frozen basic receiver (model.py:1222-1292) + loss/BER
    // few rows (73 frames): the fused dense + tail launch is 20 tiles of 48x64 behind a 14-tile k-loop (23 us); the
    // one-latency 16x16 tiles of fewrow.h (200 blocks) followed by the stand-alone tail launch take about half of that
    const bool few_rx = g_tune[TUNE_FEWROW] && g_tune[TUNE_SKINNY] > 0 && B <= 96 && (L.dK % 16) == 0 && L.dK >= 128 &&
                        L.dK <= 1152 && (L.dN % 16) == 0 && aligned16(w.fft) && aligned16(Q + L.o_dense_w) && aligned16(w.z);
    // ... and with the receiver's C-Conv and dense layer folded into one matrix (dccn_eq_rx_fold: the receiver is frozen)
    // both run as ONE such GEMM over the flattened frame, K = S * 2 n_sc
    const float* Mf = b->rx_folded;
    const int fK = d.S * N2;
    const bool folded = few_rx && Mf != nullptr && g_tune[TUNE_FEWROW] >= 1 && aligned16(Mf) && aligned16(b->out_eq) &&
                        (fK % 16) == 0 && fK <= 1152;
    if (!folded) DCCN_TRY(cconv_fwd_impl(b->out_eq + d.win, Q + L.o_conv_w, Q + L.o_conv_b, w.fft, R, rsh.kin, d.F, s, N2));
    // the tail is the one stage whose kernels depend on the modulation: the chains of a group (common.h ChainCtx) run it class by
    // class -- one launch (pair) per distinct nbits, carrying the chains of that modulation -- everything else is one launch
    // for all of them.  fin[c] / fin_class: the metric reduction each chain's optimizer-launch job runs.
    TailFinalizeArgs fin[4];
    int fin_class[kMaxChains] = {0, 0, 0, 0, 0, 0, 0, 0}, n_class = 0;
    const bool fin_deferred = replan && train;   // training: the tail's metric reduction rides on the optimizer launch
    // the receiver's linear part as this step runs it: (input rows, weights, bias, k extent)
    const float* rxA = folded ? b->out_eq : w.fft;
    const float* rxW = folded ? Mf : Q + L.o_dense_w;
    const float* rxb = folded ? Mf + (size_t)fK * L.dN : Q + L.o_dense_b;
    const int rxK = folded ? fK : L.dK;
    auto tail_section = [&](const int nbits, TailFinalizeArgs* fin_out) -> int {
        // BPSK / QPSK: the tail rides on the few-row tiles themselves (dense_tail_impl picks the fewrow.h form for <= 96 rows)
        const bool few_tail = few_rx && nbits <= 2 && dense_tail_ok(rxA, rxW, B, rxK, L.dN, nbits);
        if ((few_tail || !few_rx) && dense_tail_planned(nbits, train, B, L.dN) &&
            dense_tail_ok(rxA, rxW, B, rxK, L.dN, nbits)) {                      // dense + tail in one launch
            DCCN_TRY(dense_tail_impl(train, rxA, rxW, rxb, nullptr, b->bits, Q + L.o_tail, b->prob,
                                     b->metrics, train ? w.dz : nullptr, train ? w.dtail : nullptr, B, rxK, L.dN, nbits,
                                     &pp, b->tx_power, w.ws_tail, w.n_tail, s, fin_deferred ? fin_out : nullptr));
        } else {
            if (folded) DCCN_TRY(dense_fwd_impl(b->out_eq, Mf, Mf + (size_t)fK * L.dN, w.z, B, fK, L.dN, s));
            else DCCN_TRY(dense_fwd_impl(w.fft, Q + L.o_dense_w, Q + L.o_dense_b, w.z, B, L.dK, L.dN, s));
            DCCN_TRY(tail_impl(train, w.z, b->bits, Q + L.o_tail, b->prob, b->metrics, train ? w.dz : nullptr,
                               train ? w.dtail : nullptr, L.cells, nbits, &pp, b->tx_power, w.ws_tail, w.n_tail, s,
                               fin_deferred ? fin_out : nullptr));
        }
        return DCCN_OK;
    };
    if (tl_chain.G == 1) {
        DCCN_TRY(tail_section(sh->nbits, &fin[0]));
        n_class = 1;
    } else {
        if (!folded || !fin_deferred) return DCCN_ERR_UNSUPPORTED;
        const ChainCtx all = tl_chain;
        bool done[kMaxChains] = {false, false, false, false, false, false, false, false};
        for (int g = 0; g < all.G; ++g) {
            if (done[g]) continue;
            if (n_class >= 4) return DCCN_ERR_INVALID_ARG;
            ChainCtx sub = all;
            sub.G = 0;
            for (int h = g; h < all.G; ++h) {
                if (all.nbits[h] != all.nbits[g]) continue;
                sub.co.off[sub.G] = all.co.off[h]; sub.nbits[sub.G] = all.nbits[h];
                ++sub.G;
                done[h] = true;
                fin_class[h] = n_class;
            }
            ChainScope scope(sub);
            DCCN_TRY(tail_section(all.nbits[g], &fin[n_class]));
            ++n_class;
        }
    }
    if (!train) return DCCN_OK;

    // ---- backward: through the frozen receiver to its input ...
    if (folded) {
        // dout = dz . Mf^T in one GEMM (the zero rows of Mf leave zeros in the cyclic-prefix samples when cp = 0)
        DCCN_TRY(dense_bwd_x_impl(w.dz, Mf, w.dout, B, fK, L.dN, s));
    } else {
        DCCN_TRY(dense_bwd_x_impl(w.dz, Q + L.o_dense_w, w.dfft, B, L.dK, L.dN, s));
        if (!d.cp) {                                        // nothing flows back into the cyclic-prefix samples
            DCCN_NO_CHAINS();
            hipLaunchKernelGGL(zero_fill_kernel, dim3(ew_blocks_n((long long)R * N2)), dim3(256), 0, s, w.dout,
                               (long long)R * N2);
            DCCN_LAUNCH_CHECK();
        }
        DCCN_TRY(cconv_bwd_x_impl(w.dfft, Q + L.o_conv_w, w.dout + d.win, R, rsh.kin, d.F, s, N2));
    }
    // ... then the equaliser, last layer first.  replan: every weight gradient stays where its GEMM left it (finished
    // in the gradient arena, or as split-K slabs in the layer's own workspace) until the optimizer launch
    DeferredSlabs ds5{}, dsT{}, ds4{}, ds3{}, ds2{}, ds1{}, ds0{};       // (null slabs: the gradient arena holds the result)
    FoldDefer fpair[2], fconv;
    fpair[0].slabs = fpair[1].slabs = fconv.slabs = nullptr;
    bool split_done = false;                // dcat's stores write deqc / dcorc themselves (when the plan has the stage)
    DCCN_TRY(dense_bwd_full_impl(w.cat, w.dout, P + d.o[18], w.dcat, G + d.o[18], G + d.o[19], R, 4 * K, N2,
                                 w.ws_l[EQL_DENSE5], w.n_l[EQL_DENSE5], s, 1, nullptr, nullptr, keep_slabs ? &ds5 : nullptr,
                                 pair ? w.deqc : nullptr, pair ? (long long)(w.dcorc - w.deqc) : 0, pair ? &split_done : nullptr));
    if (!split_done) {
        DCCN_NO_CHAINS();
        hipLaunchKernelGGL(split_pairs_kernel, dim3(ew_blocks_n((long long)R * K)), dim3(256), 0, s, (const float4*)w.dcat,
                           (float2*)w.deqc, (float2*)w.dcorc, (long long)R * K);
        DCCN_LAUNCH_CHECK();
    }
    if (pair) {
        // dX and dWeff slabs of both C-Convs in one grid (group 0 = eq, 1 = corr)
        DCCN_TRY(cconv_bwd_grouped_impl(w.eq, w.deqc, P + d.o[16], w.deq, R, K, K, 2, g_in, w.dcorc - w.deqc, g_w,
                                        w.ws_l[EQL_PAIR], w.n_l[EQL_PAIR], fpair, s));
        if (w.dcorr - w.deq != g_in) return DCCN_ERR_STATE;
    } else {
        DCCN_TRY(cconv_bwd_x_impl(w.deqc, P + d.o[16], w.deq, R, K, K, s));
        DCCN_TRY(cconv_bwd_w_impl(w.eq, w.deqc, G + d.o[16], G + d.o[17], R, K, K, w.ws_l[EQL_PAIR], w.n_l[EQL_PAIR], s));
        DCCN_TRY(cconv_bwd_x_impl(w.dcorc, P + d.o[14], w.dcorr, R, K, K, s));
        DCCN_TRY(cconv_bwd_w_impl(w.corr, w.dcorc, G + d.o[14], G + d.o[15], R, K, K, w.ws_l[EQL_PAIR], w.n_l[EQL_PAIR], s));
    }
    DCCN_LAUNCH_CHAINS_Z(equalize_bwd_kernel, dim3(ew_blocks_n(nBK / 2)), dim3(256), 0, s, (const float2*)w.y,
                         (const float2*)h, (const float2*)w.deq, (const float2*)w.dcorr, (float2*)w.dy, (float2*)w.dh,
                         nBK / 2);
    DCCN_LAUNCH_CHECK();
    bool tg_fused = false;                  // tanh gradient on the dX store: dd4 = (dh . T^T) (1 - d4^2)
    DCCN_TRY(dense_bwd_full_impl(w.d4, w.dh, w.T, w.dd4, w.dT, w.dbe, B, SK2, SK2, w.ws_l[EQL_SMOOTH], w.n_l[EQL_SMOOTH], s, 3,
                                 w.d4, &tg_fused, keep_slabs ? &dsT : nullptr));
    if (dsT.dw_slabs) {
        // the fold of this layer gathers diagonals of dT (one cache line per element): over several slabs that gather
        // thrashes the L2 (measured 53 us for the optimizer launch at 1170 frames), so the slabs are summed first
        const long long n = (long long)SK2 * SK2;
        DCCN_TRY(launch_splitk_reduce2(dsT.dw_slabs, dsT.splits, n, w.dT, n, dsT.db_slabs, (long long)SK2, w.dbe, (long long)SK2, s));
        dsT.dw_slabs = dsT.db_slabs = nullptr;
    }
    if (!replan) {
        DCCN_NO_CHAINS();
        hipLaunchKernelGGL(cconv2d_same_reduce_kernel, dim3(d.S * K + 1), dim3(64), 0, s, (const float*)w.dT,
                           (const float*)w.dbe, G + d.o[12], G + d.o[13], d.S, K, d.S, K);
        DCCN_LAUNCH_CHECK();
    }
    if (!tg_fused) {
        DCCN_NO_CHAINS();
        hipLaunchKernelGGL(tanh_bwd_kernel, dim3(ew_blocks_n(nBK)), dim3(256), 0, s, (const float*)w.dd4, (const float*)w.d4,
                           w.dd4, nBK);
        DCCN_LAUNCH_CHECK();
    }
    DCCN_TRY(dense_bwd_full_impl(w.d3, w.dd4, P + d.o[10], w.dd3, G + d.o[10], G + d.o[11], B, SK2, SK2, w.ws_l[EQL_DENSE4],
                                 w.n_l[EQL_DENSE4], s, 1, nullptr, nullptr, keep_slabs ? &ds4 : nullptr));
    DCCN_TRY(dense_bwd_full_impl(w.d2, w.dd3, P + d.o[8], w.dd2, G + d.o[8], G + d.o[9], B, SK2, SK2, w.ws_l[EQL_DENSE3],
                                 w.n_l[EQL_DENSE3], s, 1, nullptr, nullptr, keep_slabs ? &ds3 : nullptr));
    const float* dy_sum;
    bool rode_T = false;                                        // the smoothing kernel's fold rode as well
    bool rode[2] = {false, false};                              // dense_4 / dense_3 kernels updated by riders (below)
    const int bn_tiles = ceil_div(B, 16);
    float* bn_w2 = w.bn_part;                                   // [tiles][P][SK2]
    float* bn_b2 = bn_w2 + (size_t)bn_tiles * d.Pp * SK2;       // [tiles][SK2]
    float* bn_w1 = bn_b2 + (size_t)bn_tiles * SK2;              // [tiles][SK2][P]
    float* bn_b1 = bn_w1 + (size_t)bn_tiles * SK2 * d.Pp;       // [tiles][P]
    if (bn) {
        // both layers' backward in one launch: dd1, the branch's input gradient added to dy, per-block partials of the
        // four weight / bias gradients (summed by the optimizer launch)
        auto kern = d.Pp == 32 ? eq_bottleneck_bwd_kernel<2> : eq_bottleneck_bwd_kernel<1>;
        const int q = eq_bottleneck_q(B, SK2);
        const int nx = ceil_div(SK2 / 16, q);
        // riders: the Adam updates of dense_3 / dense_4 (their gradients are complete in the arena, nothing from here to the
        // end of the step touches those kernels) stream behind this launch's own blocks instead of in the optimizer launch
        EqRideArgs ride;
        memset(&ride, 0, sizeof(ride));
        if (g_tune[TUNE_EQ_RIDERS]) {
            ride.p.param = b->eq_params; ride.p.grad = G; ride.p.m = b->adam_m; ride.p.v = b->adam_v;
            ride.p.reg_coef = b->reg_coef; ride.p.state = b->adam;
            if (dsT.dw_slabs == nullptr && g_tune[TUNE_EQ_RIDERS] != 4) {
                // the smoothing kernel's fold (a gather over the diagonals of dT: latency, not bytes) goes first
                EqOptJob& J = ride.job[ride.njobs++];
                J.kind = EQJ_CONV2D_FOLD; J.block0 = ride.blocks; J.blocks = ceil_div(d.S * K + 1, 4); J.splits = 1;
                J.off = d.o[12]; J.off_b = d.o[13]; J.src = w.dT; J.src2 = w.dbe; J.slab = (long long)SK2 * SK2; J.slab2 = SK2;
                J.kin = d.S; J.F = K;
                ride.blocks += J.blocks;
                rode_T = true;
            }
            for (int li = 0; li < 2; ++li) {
                const int i = li == 0 ? 10 : 8;
                const DeferredSlabs& dsl = li == 0 ? ds4 : ds3;
                if (dsl.dw_slabs != nullptr || (d.sz[i] % 4) != 0) continue;
                if (g_tune[TUNE_EQ_RIDERS] == 2 + li) continue;          // 2: only dense_3 rides, 3: only dense_4

                EqOptJob& J = ride.job[ride.njobs++];
                J.kind = EQJ_SUM; J.block0 = ride.blocks; J.blocks = EqOptBuilder::stream_blocks(d.sz[i]); J.splits = 1;
                J.off = d.o[i]; J.n = d.sz[i]; J.vec = 1; J.reg_uniform = (b->reg_uniform != 0 && b->reg_coef != nullptr) ? 1 : 0;
                ride.blocks += J.blocks;
                rode[li] = true;
            }
        }
        // the NEXT batch's generator rides here as well (dccn_eq_buffers.gen_next_rides): its workgroups are the first grid rows
        GenStaticArgs ga;
        GenChainScalars gc;
        memset(&ga, 0, sizeof(ga));
        memset(&gc, 0, sizeof(gc));
        int gen_rows = 0, gen_blocks = 0;
        size_t gen_smem = 0;
        if (gen_rides) {
            DCCN_TRY(gen_static_args(b->x_next_virtual, &ga));
            gen_blocks = ceil_div(b->x_next_virtual->frames, kGenFramesPerBlock);
            gen_rows = ceil_div(gen_blocks, nx);
            gen_smem = gen_static_smem_bytes<7, 64, 16>();
            if (tl_chain.G > 1) {
                gc.n = tl_chain.G;
                for (int g = 0; g < tl_chain.G; ++g) {
                    gc.nbits[g] = tl_chain.gen_nbits[g]; gc.offset[g] = tl_chain.gen_offset[g]; gc.seed[g] = tl_chain.gen_seed[g];
                }
            }
            gen_done = true;
        }
        DCCN_LAUNCH_CHAINS_Z(kern, dim3(nx, gen_rows + bn_tiles + ceil_div(ride.blocks, nx)), dim3(256), gen_smem, s,
                             (const float*)w.dd2, (const float*)w.d1, (const float*)w.y, P + d.o[4], P + d.o[6], (const float*)w.dy,
                             w.dflat, bn_w2, bn_b2, bn_w1, bn_b1, B, SK2, q, bn_tiles, ride, hp, gen_rows, gen_blocks, ga, gc);
        DCCN_LAUNCH_CHECK();
        dy_sum = w.dflat;
    } else {
        DCCN_TRY(dense_bwd_full_impl(w.d1, w.dd2, P + d.o[6], w.dd1, G + d.o[6], G + d.o[7], B, d.Pp, SK2, w.ws_l[EQL_DENSE2],
                                     w.n_l[EQL_DENSE2], s, 1, nullptr, nullptr, keep_slabs ? &ds2 : nullptr));
        // dy += (gradient through the pilot branch): by the dX stores themselves when the launch plan has the stage (the sum
        // then lands in dflat), else by a launch of its own
        bool add_fused = false;
        DCCN_TRY(dense_bwd_full_impl(w.y, w.dd1, P + d.o[4], w.dflat, G + d.o[4], G + d.o[5], B, SK2, d.Pp, w.ws_l[EQL_DENSE1],
                                     w.n_l[EQL_DENSE1], s, 4, w.dy, &add_fused, keep_slabs ? &ds1 : nullptr));
        if (!add_fused) {
            DCCN_NO_CHAINS();
            hipLaunchKernelGGL(add_inplace_kernel, dim3(ew_blocks_n(nBK)), dim3(256), 0, s, w.dy, (const float*)w.dflat, nBK);
            DCCN_LAUNCH_CHECK();
        }
        dy_sum = add_fused ? w.dflat : w.dy;
    }
    const bool conv_grouped = replan && cconv_pair_ok(w.t1, P + d.o[2], w.dt1, R, K, K, 0, 0, 0) && aligned16(dy_sum);
    if (conv_grouped) {
        DCCN_TRY(cconv_bwd_grouped_impl(w.t1, dy_sum, P + d.o[2], w.dt1, R, K, K, 1, 0, 0, 0, w.ws_l[EQL_CONV], w.n_l[EQL_CONV],
                                        &fconv, s));
    } else {
        DCCN_TRY(cconv_bwd_x_impl(dy_sum, P + d.o[2], w.dt1, R, K, K, s));
        DCCN_TRY(cconv_bwd_w_impl(w.t1, dy_sum, G + d.o[2], G + d.o[3], R, K, K, w.ws_l[EQL_CONV], w.n_l[EQL_CONV], s));
    }
    DCCN_TRY(dense_bwd_w_impl(w.ln + d.win, w.dt1, G + d.o[0], G + d.o[1], R, kin0, K2, w.ws_l[EQL_DENSE], w.n_l[EQL_DENSE], s,
                              keep_slabs ? &ds0 : nullptr, N2));
    // optimizer: Equalizer/* only (ofdmreceiver_np_mp.py:330), L2 terms enter through reg_coef
    if (!replan) return adam_impl(b->eq_params, G, b->adam_m, b->adam_v, b->reg_coef, nullptr, b->adam, hp, d.o[20], s, false);
    EqOptBuilder ob;
    memset(&ob.a, 0, sizeof(ob.a));
    ob.a.param = b->eq_params; ob.a.grad = G; ob.a.m = b->adam_m; ob.a.v = b->adam_v; ob.a.reg_coef = b->reg_coef;
    ob.a.state = b->adam;
    const bool uni = b->reg_uniform != 0 && b->reg_coef != nullptr;
    const dccn_gen_static* gv = b->x_next_virtual;
    if (gv != nullptr && (!gv->y || !gv->noise || !gv->power_partial || gv->frames != B ||
                          2 * (gv->K + gv->CP) * gv->S != ncols))
        return DCCN_ERR_INVALID_ARG;
    const float* rin = gv ? gv->y : b->x_next;
    if (rides && rin != nullptr && norm_fused_ok(rin, w.x_norm, B, ncols)) {
        // `input:0` of the next batch: x_norm is read by the layer norm only, long before this launch
        PowerPartials pn;
        norm_power_partials(B, ncols, w.ws_norm, w.n_norm, rin, w.x_norm, &pn, nslot ^ 1);
        NormVirtual nv = norm_virtual_none();
        if (gv) {                                       // (y, noise, partials) of the fused generator as the input: norm_adam.h
            nv.y = gv->y; nv.noise = gv->noise; nv.ppart = gv->power_partial;
            nv.npart = ceil_div(gv->frames, kGenFramesPerBlock);
            nv.total = (double)gv->frames * (double)(gv->S * (gv->K + gv->CP));
            nv.npart_noise = gv->noise_partial; nv.n_noise = nv.npart;
            nv.npow_out = gv->noise_partial ? gv->noise_power_out : nullptr;
        }
        ob.norm_next(rin, w.x_norm, B, ncols, b->tx_power != nullptr ? const_cast<double*>(pn.partial) : nullptr,
                     norm_fused_blocks(ncols), nv);
    } else if (gv != nullptr) {
        return DCCN_ERR_INVALID_ARG;                    // nothing else would form that batch: ask dccn_eq_norm_rides first
    }
    eq_opt_dense(ob, d, 0, ds0, K2, uni);
    if (fconv.slabs) ob.cconv_fold(d.o[2], d.o[3], fconv.slabs, fconv.colsum, fconv.splits, fconv.slab, K, K);
    else { ob.plain(d.o[2], d.sz[2]); ob.plain(d.o[3], d.sz[3]); }
    if (bn) {
        ob.slabs(d.o[4], d.sz[4], bn_w1, bn_tiles, (long long)SK2 * d.Pp, uni);
        ob.slabs(d.o[5], d.sz[5], bn_b1, bn_tiles, d.Pp, uni);
        ob.slabs(d.o[6], d.sz[6], bn_w2, bn_tiles, (long long)d.Pp * SK2, uni);
        ob.slabs(d.o[7], d.sz[7], bn_b2, bn_tiles, SK2, uni);
    } else {
        eq_opt_dense(ob, d, 4, ds1, d.Pp, uni);
        eq_opt_dense(ob, d, 6, ds2, SK2, uni);
    }
    eq_opt_dense(ob, d, 8, ds3, SK2, uni, rode[1]);
    eq_opt_dense(ob, d, 10, ds4, SK2, uni, rode[0]);
    if (!rode_T)
    ob.conv2d_fold(d.o[12], d.o[13], dsT.dw_slabs ? dsT.dw_slabs : w.dT, (dsT.dw_slabs && dsT.db_slabs) ? dsT.db_slabs : w.dbe,
                   dsT.dw_slabs ? dsT.splits : 1, (long long)SK2 * SK2, SK2, d.S, K);
    for (int g = 1; g >= 0; --g) {          // arena order: conv3d_2 (corr, group 1), then conv3d_3 (eq, group 0)
        const int i = g == 1 ? 14 : 16;
        if (fpair[g].slabs) ob.cconv_fold(d.o[i], d.o[i + 1], fpair[g].slabs, fpair[g].colsum, fpair[g].splits, fpair[g].slab, K, K);
        else { ob.plain(d.o[i], d.sz[i]); ob.plain(d.o[i + 1], d.sz[i + 1]); }
    }
    eq_opt_dense(ob, d, 18, ds5, N2, uni);
    // the training loop's per-step monitors (dccn_eq_monitor_accumulate) as part of this launch: dccn_eq_buffers.monitor
    if (b->monitor != nullptr) {
        const dccn_eq_monitor* m = b->monitor;
        if (!fin_deferred || !m->chan || !m->acc5 || m->chest != b->chest || m->metrics != b->metrics || m->B != B || m->S != d.S ||
            m->K != K || !m->workspace || m->workspace_bytes < dccn_eq_monitor_workspace_size(B, d.S, K))
            return DCCN_ERR_INVALID_ARG;
        EqMonitorArgs ma;
        ma.chest = m->chest; ma.chan = m->chan; ma.gt_per_symbol = m->chan_per_symbol ? 1 : 0; ma.B = B; ma.S = d.S; ma.K = K;
        ma.metrics = m->metrics; ma.tx_power = m->tx_power; ma.noise_power = m->noise_power; ma.acc = m->acc5; ma.rms_out = m->rms_out;
        ma.counter = static_cast<unsigned*>(m->workspace);
        ma.partial = reinterpret_cast<double*>(static_cast<char*>(m->workspace) + 256);
        ob.monitor(ma, eq_monitor_blocks(B, K));
        for (int c = 0; c < n_class; ++c) { fin[c].mon_acc = m->acc5; fin[c].mon_noise = m->noise_power; }
        if (m->tx_power != b->tx_power) return DCCN_ERR_INVALID_ARG;
    }
    if (fin_deferred) ob.tail_finalize(fin, n_class, fin_class);
    if (snr_pending) ob.pilot_snr(w.eq, b->pilot_carriers, b->snr_db, B, d.S, K, sh->P);
    if (gen_wanted && !gen_done) return DCCN_ERR_STATE;
    return launch_eq_opt(ob, hp, s);
}
