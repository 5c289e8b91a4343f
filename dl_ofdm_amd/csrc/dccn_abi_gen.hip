// libdccn.so -- device-side generator, classical receivers, in-graph AWGN branch (see abi_impl.h for how the library is cut into units)
#include "abi_impl.h"

using namespace dccn;

extern "C" {

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

}  // extern "C"
