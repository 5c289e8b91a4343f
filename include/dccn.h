/*
 * dccn.h -- C ABI of libdccn.so, the MI355X (gfx950) implementation of the DCCN OFDM
 * receiver hot path.
 *
 * The reference (zhongyuanzhao/dl_ofdm) has no FFI: its boundary is the Python layer
 * API of dev/py/complex.py + dev/py/model.py sitting on TensorFlow library ops.  Each
 * entry point below replaces the TF ops behind one of those call sites (cited per
 * function as dev/py/<file>:<lines>).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (16-byte aligned), unless
 *     the parameter comment says "host";
 *   - all tensors are dense row-major float32; complex values carry a trailing
 *     {I,Q} axis of size 2 (dev/py/ofdm.py:375-377);
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous and
 *     stream-ordered, allocates nothing, and is re-entrant given distinct workspaces;
 *   - the return value is 0 (DCCN_OK) or a negative dccn_status; no C++ exception
 *     crosses the ABI.  dccn_strerror() turns a status into text;
 *   - `*_workspace_size()` return the scratch bytes the matching call needs.
 */
#ifndef DCCN_H_
#define DCCN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dccn_stream_t;

typedef enum dccn_status {
    DCCN_OK = 0,
    DCCN_ERR_INVALID_ARG = -1,   /* bad size / null pointer / unsupported nbits */
    DCCN_ERR_WORKSPACE = -2,     /* workspace too small */
    DCCN_ERR_LAUNCH = -3,        /* hipLaunch / runtime error (see dccn_last_hip_error) */
    DCCN_ERR_NO_DEVICE = -4,     /* no gfx950 device visible */
    DCCN_ERR_STATE = -5,         /* plan used in the wrong state */
    DCCN_ERR_UNSUPPORTED = -6    /* a grouped call (dccn_eq_*_grouped) reached a launch that cannot carry several chains */
} dccn_status;

const char* dccn_strerror(int status);
int dccn_version(void);
/* 16 hex digits: hash of the sources this library was built from (csrc/Makefile).  Measurements that are kept next to the code
 * (profiles/pmc_traffic.json) carry it, so a reader can tell whether they describe the library that is loaded. */
const char* dccn_build_id(void);                              /* 100*major + minor */
int dccn_last_hip_error(void);                       /* hipError_t of the last failure */
/* host out-params; returns DCCN_ERR_NO_DEVICE when no GPU is visible */
int dccn_device_info(int* cu_count, int* wavefront, size_t* hbm_bytes, char* arch, int arch_len);

/* ---- R0: input normalisation ------------------------------------------------------
 * dev/py/ofdmreceiver_np.py:128-129  tf.nn.moments(x,[0]) + batch_normalization/sqrt(2).
 * x,y [batch, cols]; mean,var [cols] (nullable).  y = (x*inv + (-mean*inv)) / sqrt(2),
 * inv = rsqrt(var + eps), biased variance over the batch axis. */
size_t dccn_batch_moment_norm_workspace_size(int batch, int cols);
int dccn_batch_moment_norm_fwd(const float* x, float* y, float* mean, float* var,
                               int batch, int cols, float eps,
                               void* workspace, size_t workspace_bytes, dccn_stream_t stream);

/* ---- R8: complex_clip --------------------------------------------------------------
 * dev/py/complex.py:21-27  clip_by_norm over the IQ axis + mean clipped power.
 * x,y [n_pairs,2] (y nullable: power only); power_out: device float[1]. */
size_t dccn_clip_power_workspace_size(long long n_pairs);
int dccn_clip_power(const float* x, float* y, float* power_out, long long n_pairs, float peak,
                    void* workspace, size_t workspace_bytes, dccn_stream_t stream);

/* ---- R1 / R1': complex convolution as one fused real GEMM ---------------------------
 * dev/py/complex.py:140-196 (layers_conv2d_complex) and :51-92 (layers_conv1d_complex):
 * conv3d/conv2d with 2F filters + 4-way combine.  GEMM form over an (im2col) patch axis:
 *   x    [rows, kin, 2]      rows = output positions, kin = taps*channels per position
 *   w    [kin, 2F]           [Wa | Wb] (the live taps of the TF kernel)
 *   bias [2F]                [ba | bb], nullable
 *   out  [rows, F, 2]        re = I.Wa - Q.Wb + (ba-bb),  im = I.Wb - Q.Wa + (bb-ba)
 * The four real products run as ONE MFMA GEMM [rows,2kin] x [2kin,2F] whose B tile is
 * expanded from w while it is staged into LDS.  Forward is bitwise deterministic. */
int dccn_cconv_gemm_fwd(const float* x, const float* w, const float* bias, float* out,
                        int rows, int kin, int F, dccn_stream_t stream);
size_t dccn_cconv_gemm_bwd_w_workspace_size(int rows, int kin, int F);
/* dw [kin,2F], dbias [2F] (nullable) from x and dout [rows,F,2] */
int dccn_cconv_gemm_bwd_w(const float* x, const float* dout, float* dw, float* dbias,
                          int rows, int kin, int F,
                          void* workspace, size_t workspace_bytes, dccn_stream_t stream);
/* dx [rows,kin,2] from dout and w */
int dccn_cconv_gemm_bwd_x(const float* dout, const float* w, float* dx,
                          int rows, int kin, int F, dccn_stream_t stream);

/* ---- R2: dense layer ---------------------------------------------------------------
 * dev/py/model.py:1268-1275  tf.layers.dense: y[M,N] = x[M,K] . w[K,N] + bias[N]. */
int dccn_dense_fwd(const float* x, const float* w, const float* bias, float* y,
                   int M, int K, int N, dccn_stream_t stream);
int dccn_dense_bwd_x(const float* dy, const float* w, float* dx,
                     int M, int K, int N, dccn_stream_t stream);
size_t dccn_dense_bwd_w_workspace_size(int M, int K, int N);
int dccn_dense_bwd_w(const float* x, const float* dy, float* dw, float* dbias,
                     int M, int K, int N,
                     void* workspace, size_t workspace_bytes, dccn_stream_t stream);

/* both gradients of the dense layer in ONE grouped launch (dx = dy.w^T and dw = x^T.dy are independent
 * GEMMs that share dy; packing their blocks on one grid fills the CUs far better than two launches);
 * workspace as dccn_dense_bwd_w_workspace_size(M,K,N). */
int dccn_dense_bwd(const float* x, const float* dy, const float* w, float* dx, float* dw, float* dbias,
                   int M, int K, int N, void* workspace, size_t workspace_bytes, dccn_stream_t stream);

/* the same grouped launch, stopping before the split-K reduction: dx is final, the weight gradient is left
 * as `*splits` slabs [splits][K*N] at the start of the workspace (bias slabs [splits][N] follow, 256-byte
 * aligned) -- this is the stage the fused training step runs (its Adam kernel sums the slabs).
 * `splits` is a host out-parameter (1 = dw/dbias were written directly). */
int dccn_dense_bwd_slabs(const float* x, const float* dy, const float* w, float* dx, float* dw, float* dbias,
                         int M, int K, int N, void* workspace, size_t workspace_bytes, int* splits,
                         dccn_stream_t stream);

/* ---- R3-R6: demodulation tail + loss + BER ----------------------------------------
 * dev/py/model.py:1278-1291 (1x1 conv2d 2->m, leaky-ReLU 0.2, concat, dense (m+2)->2b,
 * leaky-ReLU, softmax over bit pairs), dev/py/ofdmreceiver_np.py:154-169 (one_hot,
 * softmax_cross_entropy_with_logits on the softmax OUTPUT, argmax, confusion matrix),
 * dev/py/util.py:44-48 (ber_tensor).  m = 2^nbits, nbits in 1..4.
 *   z      [cells, 2]         dense output viewed per data cell
 *   bits   [cells, nbits]     int32 labels
 *   tailp  [dccn_tail_param_count(nbits)]  packed: w1[2,m] | b1[m] | w2[m+2,2b] | b2[2b]
 *   prob   [cells, nbits, 2]  (nullable)
 *   metrics: device dccn_metrics, written by the call (deterministic two-stage reduce)
 * _fwd_bwd additionally emits dz [cells,2] and dtailp (same packing), the gradient of
 * ce_mean = mean over cells*nbits of the cross entropy. */
typedef struct dccn_metrics {
    double ce_sum;             /* sum of per-bit cross entropies */
    long long conf[4];         /* confusion matrix, row = label, col = decision */
    long long count;           /* cells * nbits */
    float ce_mean;             /* ce_sum / count */
    float berlin;              /* (conf[1]+conf[2]) / count, f64 division cast to f32 */
    float log_ber;             /* logf(berlin) (-inf when error free) */
    float reserved;
} dccn_metrics;

int dccn_tail_param_count(int nbits);
size_t dccn_demod_tail_workspace_size(long long cells, int nbits);
int dccn_demod_tail_loss_fwd(const float* z, const int32_t* bits, const float* tailp,
                             float* prob, dccn_metrics* metrics, long long cells, int nbits,
                             void* workspace, size_t workspace_bytes, dccn_stream_t stream);
int dccn_demod_tail_loss_fwd_bwd(const float* z, const int32_t* bits, const float* tailp,
                                 float* prob, dccn_metrics* metrics, float* dz, float* dtailp,
                                 long long cells, int nbits,
                                 void* workspace, size_t workspace_bytes, dccn_stream_t stream);

/* R2 with R3-R6 fused into its epilogue (nbits 1..2): the demodulation tail runs on the dense output tile
 * while it is still in registers, so z never has to exist in memory (dev/py/model.py:1268-1291 +
 * dev/py/ofdmreceiver_np.py:154-169 in one launch, plus the metrics finalize).
 *   x [M,K], w [K,N], bias [N], z [M,N] (nullable: not materialised), bits [M, N/2, nbits],
 *   prob [M, N/2, nbits, 2] (nullable), metrics as above; _fwd_bwd also emits dz [M,N] and dtailp.
 * Requires 16-byte aligned x/w and K % 4 == N % 4 == 0; returns DCCN_ERR_INVALID_ARG otherwise (use the separate
 * dccn_dense_fwd + dccn_demod_tail_loss_* calls then). */
size_t dccn_dense_tail_workspace_size(int M, int N, int nbits);
int dccn_dense_tail_fwd(const float* x, const float* w, const float* bias, float* z, const int32_t* bits,
                        const float* tailp, float* prob, dccn_metrics* metrics, int M, int K, int N, int nbits,
                        void* workspace, size_t workspace_bytes, dccn_stream_t stream);
int dccn_dense_tail_fwd_bwd(const float* x, const float* w, const float* bias, float* z, const int32_t* bits,
                            const float* tailp, float* prob, dccn_metrics* metrics, float* dz, float* dtailp,
                            int M, int K, int N, int nbits, void* workspace, size_t workspace_bytes,
                            dccn_stream_t stream);

/* The equaliser's pilot bottleneck (dev/py/model.py:394-412: dense SK2 -> P, dense P -> SK2, no activation in between) as
 * ONE launch per direction, a block per 16 frames x a chunk of columns (csrc/eq_bottleneck.h).  P = 2 * pilot_size must be
 * 16 or 32, SK2 a multiple of 64 (dccn_eq_bottleneck_supported); otherwise use dccn_dense_fwd / dccn_dense_bwd twice.
 *   fwd: d1 [B,P] = y [B,SK2] . W1 [SK2,P] + b1;  d2 [B,SK2] = d1 . W2 [P,SK2] + b2
 *   bwd: given dd2 = dLoss/dd2: dy_out = dy_in + (dd2 . W2^T) . W1^T  (the branch's input gradient ADDED to dy_in),
 *        dW2 = d1^T . dd2, db2, dW1 = y^T . (dd2 . W2^T), db1   (per-block partials summed in a fixed order) */
int dccn_eq_bottleneck_supported(int B, int SK2, int P);
size_t dccn_eq_bottleneck_workspace_size(int B, int SK2, int P);
int dccn_eq_bottleneck_fwd(const float* y, const float* W1, const float* b1, const float* W2, const float* b2, float* d1,
                           float* d2, int B, int SK2, int P, dccn_stream_t stream);
int dccn_eq_bottleneck_bwd(const float* dd2, const float* d1, const float* y, const float* W1, const float* W2,
                           const float* dy_in, float* dy_out, float* dW1, float* db1, float* dW2, float* db2, int B, int SK2,
                           int P, void* workspace, size_t workspace_bytes, dccn_stream_t stream);

/* Patch gather in front of dccn_cconv_gemm_* for the general-k cases of dev/py/complex.py:51-92 / :140-196 (k taps over
 * one or two axes, strides, TF SAME / VALID geometry): x [B, L, Wd, C, 2] -> rows [B*Lo*Wo, ntl*ntw*C, 2], zeros where
 * SAME padding lies; only the live taps tl0..tl0+ntl-1 / tw0..tw0+ntw-1 (those that ever meet data) are gathered.
 * _col2im is its adjoint: dx [B, L, Wd, C, 2] = sum of the patch entries that read each input element (deterministic). */
int dccn_cconv_im2col(const float* x, float* rows, int B, int L, int Wd, int C, int Lo, int Wo, int ntl, int ntw, int tl0,
                      int tw0, int sL, int sW, int pl0, int pw0, dccn_stream_t stream);
int dccn_cconv_col2im(const float* drows, float* dx, int B, int L, int Wd, int C, int Lo, int Wo, int ntl, int ntw, int tl0,
                      int tw0, int sL, int sW, int pl0, int pw0, dccn_stream_t stream);
/* The forward of the same convolutions WITHOUT the patch tensor (dev/py/complex.py:51-92, :140-196 as an implicit GEMM):
 * out [B*Lo*Wo, F, 2] = patches(x) . Weff(w) + bias, the GEMM's operand loader gathering the taps from x [B, L, Wd, C, 2]
 * itself; w [ntl*ntw*C, 2F] = [Wa|Wb] over the live taps, bias [2F] nullable.  dccn_cconv_patch_supported: 1 when the
 * shapes qualify (even C and F, 32-bit element offsets); otherwise use dccn_cconv_im2col + dccn_cconv_gemm_fwd. */
int dccn_cconv_patch_supported(int B, int L, int Wd, int C, int Lo, int Wo, int ntl, int ntw, int F);
int dccn_cconv_patch_fwd(const float* x, const float* w, const float* bias, float* out, int B, int L, int Wd, int C, int Lo,
                         int Wo, int ntl, int ntw, int tl0, int tw0, int sL, int sW, int pl0, int pw0, int F,
                         dccn_stream_t stream);
/* The backward of the same convolutions without patch-sized tensors either (the gradients TensorFlow derives for
 * dev/py/complex.py:51-92, :140-196).  _bwd_supported: bit 0 = the weight gradient qualifies, bit 1 = the input gradient does
 * (ANY stride since round 6), bit 2 = ... and is expected to be the faster route than GEMM + col2im (2C <= 32: 16-column
 * tiles, strides by phase decomposition, csrc/cconv_dx_narrow.h -- unless the taps outnumber the channels ~10:1; wider inputs:
 * 64-column tiles whose operand loader reads dout where a tap's fine position is a multiple of the stride -- at stride 1, or
 * where the zeros a strided gather multiplies cost less than the [rows, kin, 2] round trip);
 * 0 -> dccn_cconv_im2col / _gemm_bwd_w / _gemm_bwd_x / _col2im as before.
 * _bwd_w: dw [ntl*ntw*C, 2F] (+ dbias [2F], nullable) = patches(x)^T . dout, the patch rows gathered from x [B, L, Wd, C, 2]
 *   by the weight-gradient GEMM's operand loader; dout [B*Lo*Wo, F, 2].  Deterministic (split-K slabs, fixed-order fold).
 * _bwd_x: dx [B, L, Wd, C, 2] = the convolution of dout with the tap-flipped transposed weights, as the same implicit GEMM
 *   gathering from dout (no [rows, kin, 2] intermediate, no scatter), any strides.  workspace: the flipped weights,
 *   4*C*ntl*ntw*F floats. */
int dccn_cconv_patch_bwd_supported(int B, int L, int Wd, int C, int Lo, int Wo, int ntl, int ntw, int sL, int sW, int F);
size_t dccn_cconv_patch_bwd_w_workspace_size(int B, int Lo, int Wo, int C, int ntl, int ntw, int F);
int dccn_cconv_patch_bwd_w(const float* x, const float* dout, float* dw, float* dbias, int B, int L, int Wd, int C, int Lo,
                           int Wo, int ntl, int ntw, int tl0, int tw0, int sL, int sW, int pl0, int pw0, int F,
                           void* workspace, size_t workspace_bytes, dccn_stream_t stream);
size_t dccn_cconv_patch_bwd_x_workspace_size(int C, int ntl, int ntw, int F);
int dccn_cconv_patch_bwd_x(const float* dout, const float* w, float* dx, int B, int L, int Wd, int C, int Lo, int Wo, int ntl,
                           int ntw, int tl0, int tw0, int sL, int sW, int pl0, int pw0, int F, void* workspace,
                           size_t workspace_bytes, dccn_stream_t stream);

/* Few-channel 1-D C-Convs (layers_conv1d_complex, dev/py/complex.py:51-92: x [B, L, C, 2], ntl live taps from tl0, stride sL,
 * pl0 samples of padding in front; 2*ntl*C <= 30, F = 32 or 64): dx, dw [ntl*C, 2F] and dbias [2F] (nullable) in ONE pass over
 * dout [B*Lo, F, 2] (csrc/cconv1d_bwd.h) -- the three gradients of dccn_cconv_patch_bwd_w / _bwd_x, same 1e-5 parity, for
 * about the cost of one of them on these shapes (dout is the only large tensor).  Deterministic. */
int dccn_cconv1d_bwd_supported(int B, int L, int C, int Lo, int ntl, int sL, int F);
size_t dccn_cconv1d_bwd_workspace_size(int F);
int dccn_cconv1d_bwd(const float* x, const float* dout, const float* w, float* dx, float* dw, float* dbias, int B, int L, int C,
                     int Lo, int ntl, int tl0, int sL, int pl0, int F, void* workspace, size_t workspace_bytes,
                     dccn_stream_t stream);

/* The in-graph AWGN monitor branch of the receiver graph (dev/py/radio.py:62-88 AWGN_channel, called at
 * dev/py/ofdmreceiver_np.py:136; tensors `tx_signal:0`, `iq_tx:0`, `iq_rx:0`, `noise_power:0`, :151-152,172-183):
 * tx_signal = complex_clip(x_norm, peak); xn = batch_norm(tx_signal, eps 1e-8)/sqrt(2);
 * noise = (|a| sin p, |a| cos p), a = sqrt(.5) 10^(-SNR/20) N(0,1), p = U(0, 2 pi) (Philox stream keyed by seed/offset);
 * iq_rx = fp16(xn + noise) [frames*pairs, 2], iq_tx = fp16(tx_signal), noise_power = mean |noise|^2.
 * The receiver itself never consumes this branch (`rx_iq_data = iq_tx_re`, :138).  tx_signal / iq_tx / iq_rx nullable. */
size_t dccn_ingraph_awgn_workspace_size(int frames, int pairs_per_frame);
int dccn_ingraph_awgn(const float* x_norm, const float* snr_db, float* tx_signal, uint16_t* iq_tx_f16,
                      uint16_t* iq_rx_f16, float* noise_power, int frames, int pairs_per_frame, float peak,
                      unsigned long long seed, unsigned offset, void* workspace, size_t workspace_bytes,
                      dccn_stream_t stream);

/* ---- classical pilot-aided receivers: dev/m/OFDM_Benchmark_dev.m:339-456 ----------------------
 * The estimator family the "DCCN vs LS / LMMSE" curves are drawn against, as device operators (the contractions -- DFT
 * of the FFT window, pilot interpolation, long-term LMMSE smoothing -- are dccn_dense_fwd[_ld] calls on constant
 * matrices; host mirror dl_ofdm_amd/benchmark_gpu.py, NumPy restatement = oracle: dl_ofdm_amd/benchmark.py).
 * Y [n, S*K, 2] frequency-domain frames, pil [P] / dat [D] flat cell indices (symbol*K + carrier), H [n, S*K, 2] the
 * channel's true response (Perfect / ideal LMMSE only).
 *   dccn_dense_fwd_ld        dccn_dense_fwd with a row stride on x (a column window of wider rows: the FFT window)
 *   dccn_classical_pilot_ls  gp [2][n][P] = Re / Im planes of Y[pilot] / pilot_value          (:345-352 LS at the pilots)
 *   dccn_classical_gain      block partials (left in the workspace for _estimate) of the scalar mapping the unit-gain
 *                            response onto the power-normalised frames and of sum |G_ls|^2; sums4 (nullable, device
 *                            double[4]) = {Re, Im of sum Y_p conj(H_p pv), sum |H_p pv|^2, sum |G_ls|^2}
 *   dccn_classical_estimate  mode 0 LS (planes -> interleaved), 1 ideal per-symbol LMMSE (:353-372), 2 ALMMSE (:437-446),
 *                            3 Perfect, 4 frame mean V [n, K, 2] (input of the PDP variants :399-416); Gls [2][n][S*K].
 *                            ORDER CONTRACT: modes 1 and 3 scale H by the gain whose block partials dccn_classical_gain
 *                            left in the SAME workspace -- call dccn_classical_gain(..., H != NULL, same n, same workspace)
 *                            on the same stream first; without it the partials are whatever the workspace held and G is
 *                            garbage (no error can be raised: the partials carry no tag)
 *   dccn_classical_detect    x = Y / G at the data cells, nearest point of table [m, 2], labels [m, nbits]; det (nullable)
 *                            [n, D, nbits]; errors[0] = bit errors against bits [n, D, nbits]; G rows of g_row cells per
 *                            frame, cell index taken modulo g_mod when g_mod > 0 (one estimate row per frame)      */
size_t dccn_classical_workspace_size(void);
int dccn_dense_fwd_ld(const float* x, int ldx, const float* w, const float* bias, float* y, int M, int K, int N,
                      dccn_stream_t stream);
int dccn_classical_pilot_ls(const float* Y, const int* pil, float* gp, int n, int SK, int P, float pv_re, float pv_im,
                            dccn_stream_t stream);
int dccn_classical_gain(const float* Y, const float* H, const float* Gls, const int* pil, int n, int SK, int P, float pv_re,
                        float pv_im, double* sums4, void* workspace, size_t workspace_bytes, dccn_stream_t stream);
int dccn_classical_estimate(const float* Gls, const float* H, float* G, int n, int S, int K, int mode, float c_var,
                            void* workspace, size_t workspace_bytes, dccn_stream_t stream);
int dccn_classical_detect(const float* Y, const float* G, const int* dat, const float* table, const int* labels,
                          const int32_t* bits, int32_t* det, long long* errors, int n, int SK, int D, int m, int nbits,
                          int g_row, int g_mod, void* workspace, size_t workspace_bytes, dccn_stream_t stream);

/* One row of the sweep table {c00,c01,c10,c11,ce_sum,count} (float64, device): row6 += the metrics record of the
 * last step, stream-ordered, no host round trip (dev/py/ofdmreceiver_np.py:80-85 accumulates the same on the host). */
int dccn_metrics_table_add(const dccn_metrics* metrics, double* row6, dccn_stream_t stream);
/* acc3[0] += metrics->ce_mean; acc3[1] += *tx_power; acc3[2] += *noise_power (each nullable): the per-step monitors of the
 * training loop (dev/py/ofdmreceiver_np.py:222-229) accumulated on the device in one stream-ordered launch. */
int dccn_step_monitor_add(const dccn_metrics* metrics, const float* tx_power, const float* noise_power, float* acc3,
                          dccn_stream_t stream);
/* row6 = the record (the one-point table of a single evaluation: no clearing launch needed in front of it) */
int dccn_metrics_table_set(const dccn_metrics* metrics, double* row6, dccn_stream_t stream);

/* Kernel-configuration knobs for experiments and profiling.  The table is a process-wide set of DEFAULTS; a call copies it once
 * when it begins (never mid-plan), and a plan that captured its own table (dccn_tuning_snapshot -> dccn_rx_buffers.tuning /
 * dccn_eq_buffers.tuning) is not affected by later changes at all.
 *  0 dense forward with the tail in its epilogue (> 0 on; 0: separate launches)      1 grouped dense backward (7 = k-major dW)
 *  2 C-Conv forward (7 = staged whole-k tile, default; 8-11 its other store slots / 32x128 tiles; 1-6 gemm16 tiles; 0 = 32x32x2)
 *  3 C-Conv weight gradient (7 = k-major)      4 / 5 split-K counts of the two weight gradients (0 = automatic)
 *  6 minimum LDS per block in KiB      7 single-tile launches for short k ranges (default 1)      8 skinny dispatch
 *  9 grouped backward of large layers      10 gemm16 tiles for the un-fused dense forward
 * 11 C-Conv weight gradient in the epilogue of the dense dX tiles (default 1)      12 wave priority of those tiles
 * 13 fused dense + tail launch for 8-QAM / 16-QAM steps (bit 0 lane-per-cell forms, bit 1 quad-lane training form)
 * 14 graded k ranges of the dense weight-gradient items in the fused backward launch (presets 1-24, default 14 = {9,5,2,2,1}/19)
 * 17 few-row dense backward as one grid (default 1)
 * 18 R0 of the next batch on the backward launch of double-buffered pipelined steps (default 0)
 * 19 equaliser step: element-wise stages in GEMM stores (1 few-row tiles, 2 = default: also larger batches)
 * 20 equaliser step plan (1 = default: grouped corr/eq C-Convs, concat / split in GEMM stores, ONE job-table optimizer launch;
 *    0 = launch per stage; 3 = plan 1 without the fused pilot bottleneck)
 * 21 few-row GEMMs (<= 96 rows) on one-latency 16x16 tiles (default 1)
 * 24 equaliser step: Adam updates of dense_3 / dense_4 and the smoothing kernel's fold ride behind the bottleneck backward launch
 * 25 large layers: the dense kernel's optimizer update on the library's own low-priority stream (0 off, 1 on, 2 = default: with
 *    non-temporal loads and stores)
 * 27 (default 1) large layers' fused dense + tail: a short last row tile (<= 32 of 80 rows) runs as 32x64 blocks in the same grid
 * Keys 15, 16, 22, 23, 26 were removed in round 6 (alternatives that were built, measured without gain and deleted): setting
 * them returns DCCN_ERR_INVALID_ARG.  Set knobs before workspaces are sized. */
int dccn_set_tuning(int key, int value);
int dccn_get_tuning(int key);
/* the knobs as a table (n >= dccn_tuning_count() ints; returns the count): what a plan captures to be immune to later
 * dccn_set_tuning calls -- see dccn_rx_buffers.tuning */
int dccn_tuning_count(void);
int dccn_tuning_snapshot(int* table, int n);

/* Plan queries: which launch plan the library will take for a shape under the current knobs, so that callers size
 * their buffers from the library's own rule instead of restating it.
 *   dccn_dense_tail_supported   1: dccn_dense_tail_fwd[_bwd] accepts (M, K, N, nbits) with 16-byte aligned operands
 *                               (the fused receiver step then runs R2..R6 as one launch and `z` may be NULL);
 *   dccn_rx_bwd_fused_supported 1: the training step runs its backward half as one launch (dense dX tiles with the
 *                               C-Conv weight gradient in their epilogue + dense dW items): `dfft` may be NULL. */
int dccn_dense_tail_supported(int M, int K, int N, int nbits);

/* ---- R7: optimizer -----------------------------------------------------------------
 * dev/py/ofdmreceiver_np.py:185-189  exponential_decay(1e-3, step, 500, 0.98, staircase)
 * + tf.train.AdamOptimizer (TF 1.15 ApplyAdam kernel form).  All state lives on the
 * device so a captured step can be replayed without host-side parameter changes.
 *   state: device dccn_adam_state (global_step, beta powers; advanced by the call)
 *   param/grad/m/v [n]: flat arenas
 *   reg_coef [n] (nullable): per-element L2 coefficient c; the gradient used is
 *       g + (*reg_gate) * c * param    (reg_gate: device float*, nullable -> 1.0).
 *       The receiver passes c = REG_COEFF*2*0.01 on regularised tensors and
 *       reg_gate = &metrics->berlin  (ofdmreceiver_np.py:171). */
typedef struct dccn_adam_state {
    float global_step;         /* float32 variable, ofdmreceiver_np.py:185 */
    float beta1_power;
    float beta2_power;
    float alpha;               /* lr_t used by the most recent step (output) */
} dccn_adam_state;

typedef struct dccn_adam_hparams {   /* host struct, passed by value */
    float lr0, decay_steps, decay_rate;    /* 1e-3, 500, 0.98 */
    float beta1, beta2, eps;               /* 0.9, 0.999, 1e-8 */
} dccn_adam_hparams;

int dccn_adam_tf_step(float* param, const float* grad, float* m, float* v,
                      const float* reg_coef, const float* reg_gate,
                      dccn_adam_state* state, dccn_adam_hparams hp, long long n,
                      dccn_stream_t stream);

/* ---- the fused receiver step -------------------------------------------------------
 * dev/py/ofdmreceiver_np.py:234 (session.run(train_op...)) and :80 (evaluation run):
 * R0 -> R1 -> R2 -> R3..R6 [-> backward -> R7] as one stream-ordered launch sequence
 * (optionally captured into a hipGraph and replayed).
 *
 * Parameter arena layout (floats), F2 = 2F, m = 2^nbits:
 *   conv_w [kin,F2] | conv_b [F2] | dense_w [S*F2, 2D] | dense_b [2D] | tailp
 * dccn_rx_param_offsets() fills offsets[6] = {conv_w, conv_b, dense_w, dense_b, tail, total}. */
typedef struct dccn_rx_shape {
    int batch;     /* frames (Bf) */
    int S;         /* OFDM symbols per frame (7) */
    int kin;       /* samples per symbol seen by the C-Conv (N+CP, or N when cp=False) */
    int F;         /* nfilter */
    int D;         /* data cells per frame (frame_size) */
    int nbits;     /* 1..4 */
} dccn_rx_shape;

/* The fused device-side generator of a static single-profile channel (round 5; the chain of dccn_ofdm_tx_frames +
 * dccn_channel_awgn below in ONE launch, csrc/datagen.h gen_static_frames_kernel): label bits -> resource grid -> ifft + cyclic
 * prefix -> static Rayleigh taps -> 'same' FIR (dev/py/util.py:25-29, ofdm.py:328-380, radio.py:352-372) -> y, plus the
 * frame-scaled noise of radio.py:513-526 and per-block partial sums of |y|^2 and |noise|^2.  The receiver's input is
 * x = y / sqrt(mean |y|^2 over the batch) + noise: formed by the consumer -- dccn_rx_train_step reads the descriptor as the
 * virtual input of its pipelined normalisation (dccn_rx_buffers.gen_next) -- or materialised by dccn_gen_static_apply.
 * Same Philox streams and draws as the separate launches (a batch is a pure function of (seed, offset)); the ifft runs on
 * 16x16x4 MFMA tiles instead of the 32x32x2 GEMM, so tx / y agree with that path to rounding, not bit for bit.
 * All pointers are device pointers; power_partial / noise_partial hold dccn_gen_static_partials(frames) doubles. */
typedef struct dccn_gen_static {
    int32_t* bits_out;            /* [frames, D, nbits] labels drawn here */
    const int32_t* cell_map;      /* [S*K] >= 0 data-cell index, -1 pilot, -2 empty */
    const float* const_tab;       /* [2^nbits, 2] */
    float pilot_re, pilot_im;
    const float* idft;            /* [2K, 2(K+CP)] real form of ifft + cyclic prefix */
    const float* coeff;           /* [n_taps] tap amplitudes (null when identity) */
    const float* alpha;           /* [n_taps, L] sinc interpolation */
    int n_taps, L, identity;      /* identity: AWGN channel (g = [1]) */
    const float* snr_db;          /* [frames] */
    float* y;                     /* [frames, S, K+CP, 2] channel output */
    float* noise;                 /* [frames, S, K+CP, 2] noise, already scaled per frame */
    double* power_partial;        /* [partials] */
    double* noise_partial;        /* [partials], nullable */
    float* noise_power_out;       /* device float[1], nullable: `noise_power:0`, written by the consumer that sums noise_partial */
    float* tx_out;                /* [frames, S, K+CP, 2], nullable: the transmitted frames (tests) */
    int frames, S, K, CP, D, nbits;
    unsigned long long seed;
    unsigned offset;
    /* Frame-interleaved static profiles ('mixRayleigh' without Doppler frames, dev/py/radio.py:438-452): frame f runs profile
       f % n_profiles of `profiles` (a HOST array of at most 6, read during the call); n_profiles = 0: the single profile
       (coeff, alpha, n_taps, L, identity) above.  tap_stride: taps per frame in the Philox index of the tap draws (0 = n_taps;
       16 reproduces the draws of dccn_channel_groups_awgn). */
    int n_profiles, tap_stride;
    const struct dccn_gen_profile* profiles;
    /* nullable: H [frames * h_rep, K, 2] = fft(g, K) of every frame's impulse response, h_rep copies per frame (the mixed
       channels report the response per symbol: h_rep = S) -- the values dccn_channel_awgn / _groups_awgn put into `H` */
    float* H_out;
    int h_rep;
} dccn_gen_static;
typedef struct dccn_gen_profile {
    const float* coeff;           /* [n_taps] (null when identity) */
    const float* alpha;           /* [n_taps, L] */
    int n_taps, L, identity, reserved;
} dccn_gen_profile;
int dccn_gen_static_supported(int S, int K, int CP);      /* 1: shapes the fused launch is instantiated for (N = 64) */
int dccn_gen_static_partials(int frames);
int dccn_gen_static_frames(const dccn_gen_static* g, dccn_stream_t stream);
/* x_out [frames, S, K+CP, 2] = y / sqrt(mean |y|^2) + noise; noise_power (nullable) = mean |noise|^2 */
int dccn_gen_static_apply(const dccn_gen_static* g, float* x_out, float* noise_power, dccn_stream_t stream);

typedef struct dccn_rx_buffers {
    const float* x;            /* [batch, S, kin, 2] raw input (tx_ofdm) */
    const int32_t* bits;       /* [batch, D, nbits] labels (bits_in) */
    float* params;             /* parameter arena */
    float* grads;              /* gradient arena (train only) */
    float* adam_m;             /* train only */
    float* adam_v;             /* train only */
    float* reg_coef;           /* [total] per-element L2 coefficient (train only) */
    dccn_adam_state* adam;     /* train only */
    float* x_norm;             /* [batch, S, kin, 2]  `input:0` */
    float* fft_out;            /* [batch, S, F, 2]    `receiver/fft_like/fft_out:0` */
    float* z;                  /* [batch, 2D] (nullable for nbits <= 2: the fused dense+tail launch skips it) */
    float* prob;               /* [batch, D, nbits, 2] `output:0` (nullable) */
    float* dz;                 /* [batch, 2D] train only */
    float* dfft;               /* [batch, S, F, 2] train only (nullable when dccn_rx_bwd_fused_supported: the dense dX
                                  tiles then only feed the C-Conv weight gradient in their own epilogue) */
    dccn_metrics* metrics;     /* ce_mean / conf_matrix / linear_ber / log_ber */
    float* tx_power;           /* device float[1] `tx_power:0` (nullable: skip R8) */
    void* workspace;
    size_t workspace_bytes;
    /* Software pipelining of R0 across training steps (both default 0 = every step normalises its own batch first):
       x_next != NULL  the step also normalises x_next into x_norm behind its Adam update (the leading blocks of the
                       optimizer launch; x_norm is dead by then), for the following call;
       x_prenormalised x_norm (and the R8 partial sums in the workspace) already hold this step's batch -- written
                       by the previous call through x_next -- so the step starts at R1.
       x itself is only read by R0: with both set, x and x_next may be the same buffer, refilled between calls.
       The workspace must be the same memory in both calls. */
    const float* x_next;
    int x_prenormalised;
    /* Double-buffered form of the same pipelining (dccn_rx_norm_rides_backward(shape) == 1): with x_norm_next != NULL the
       normalisation of x_next is written to x_norm_next by leading blocks of the BACKWARD launch of this step -- hidden
       behind 30 us of matrix work instead of stretching the optimizer launch -- and the caller passes that buffer as
       x_norm (and norm_slot ^ 1 as norm_slot) in the following call.  norm_slot (0/1) names the R8 partial-sum slot of
       the workspace that belongs to x_norm; x_norm is read until the backward launch ends, hence the second buffer. */
    float* x_norm_next;
    int norm_slot;
    /* Large layers (dccn_get_tuning(16), N = 1024): the dense kernel's optimizer update runs in the epilogue of its
       weight-gradient tiles and its gradient is not written to `grads` unless keep_dense_grad > 0.
       keep_dense_grad < 0 (round 5): the caller never reads the dense kernel's gradient (a training loop): the optimizer
       launch that sums its split-K slabs does not store the sum either (the other gradients are still written). */
    int keep_dense_grad;
    /* reg_coef holds ONE value over the whole dense-kernel segment (what the reference's keras l2(0.01) regulariser gives:
       dev/py/model.py:1271-1272): the optimizer then reads that value once instead of streaming a parameter-sized array
       (12 % of the optimizer launch's traffic at N = 1024).  0 = per-element coefficients everywhere (the general form). */
    int reg_uniform_dense;
    /* hipEvent_t (nullable) that the launch reading x_next waits for on `stream`: lets a producer fill x_next on ANOTHER
       stream while the forward and backward launches of this step run (the device-side generator: dl_ofdm_amd/datagen.py
       SideStreamFeeder).  Eager launches only (not inside dccn_rx_graph_create). */
    void* x_next_ready;
    /* Round 5: the next batch comes from the fused generator (host pointer to a descriptor, read during the call only): the
       step issues the generator launch as its FIRST launch and its last launch normalises (y, noise, partials) as if they were
       x_next = y / sqrt(mean |y|^2) + noise (bit-identical to normalising the materialised batch).  x_next, when non-null as
       well, receives that batch (tx_ofdm); the labels go to gen_next->bits_out (the caller's OTHER label slot).  Training
       calls on the single-buffer pipelining only (dccn_rx_norm_rides_backward(shape) == 0); one C call per generated-and-
       trained batch. */
    const dccn_gen_static* gen_next;
    /* nullable: the plan's OWN tuning table (dccn_tuning_count() ints, captured with dccn_tuning_snapshot when the plan was
       built): the call plans its launches from it instead of the process-global knobs, so dccn_set_tuning on another thread
       cannot change this plan.  NULL: the globals as they stand when the call begins (copied once: never mid-call). */
    const int* tuning;
} dccn_rx_buffers;

int dccn_rx_param_offsets(const dccn_rx_shape* shape, long long offsets[6]);
int dccn_rx_bwd_fused_supported(const dccn_rx_shape* shape);
/* 1: dccn_rx_train_step (train != 0) / dccn_rx_eval_step run R2..R6 as ONE launch for this shape -- the dense forward
 * with the demodulation tail in its epilogue -- and `z` may be NULL.  (8-QAM / 16-QAM: the tile is staged through LDS and
 * walked a lane or a quad of lanes per cell; which of those steps take the fused launch is tuning knob 13.) */
int dccn_rx_dense_tail_fused(const dccn_rx_shape* shape, int train);
/* 1: a training step given x_next + x_norm_next normalises the next batch on its backward launch (see dccn_rx_buffers) */
int dccn_rx_norm_rides_backward(const dccn_rx_shape* shape);
/* 1: a training step of this shape takes dccn_rx_buffers.gen_next (the next batch formed from the fused generator's
 * (y, noise, partials) by the pipelined R0 on the optimizer launch): single-buffer pipelining, batch <= 1536 frames, one power
 * partial per generator block.  0: generate with dccn_gen_static_frames / _apply (or the launch-per-stage generator) and hand
 * the batch over as x_next.  A step given gen_next for a shape answering 0 is refused BEFORE anything is launched. */
int dccn_rx_gen_next_supported(const dccn_rx_shape* shape);
/* The backward half of the basic receiver's training step as one launch (small layers, see the query above):
 *   dfft = dz . Wd^T                     (dev/py/model.py:1268-1275 backward; written only when dfft != NULL)
 *   dWd  = fft_out^T . dz, dbd = colsum  (k-major tiles, split-K slabs)
 *   dWeff = x_norm^T . dfft              (dev/py/complex.py:183-192 backward; contracted in the epilogue of the tiles
 *                                         that produce dfft, one [2kin,64] partial per tile)
 * reduce != 0: a second and third launch sum the slabs / fold the partials into dw_dense [S*2F, 2D], db_dense [2D]
 * (nullable), dw_conv [kin, 2F] = [Wa|Wb], db_conv [2F] (nullable); reduce == 0 leaves them in the workspace (what the
 * fused training step does: its optimizer launch reduces them).  x_norm [batch,S,kin,2], fft_out [batch,S,F,2],
 * dz [batch,2D], w_dense [S*2F, 2D]. */
size_t dccn_rx_backward_workspace_size(int batch, int S, int kin, int F, int D);
int dccn_rx_backward(const float* x_norm, const float* fft_out, const float* dz, const float* w_dense, float* dfft,
                     float* dw_dense, float* db_dense, float* dw_conv, float* db_conv, int batch, int S, int kin, int F,
                     int D, int reduce, void* workspace, size_t workspace_bytes, dccn_stream_t stream);
size_t dccn_rx_workspace_size(const dccn_rx_shape* shape, int train);
/* eager launch sequences.
 * Large layers (an unsplit dense weight gradient of >= 512 tiles of 128x128, e.g. N = 1024: a 459 MB dense kernel): the training
 * step forks the dense kernel's optimizer update onto a stream the LIBRARY owns (one per device, lowest priority, created on
 * first use) and joins it before its last launch, so that this 3.2 GB stream runs next to the MFMA-bound C-Conv
 * weight-gradient launch.  Nothing changes for the caller: every effect of the step is ordered on `stream` when the call
 * returns, results are bit-identical, and dccn_rx_graph_create captures the fork and the join with the rest. */
int dccn_rx_eval_step(const dccn_rx_shape* shape, const dccn_rx_buffers* buf, dccn_stream_t stream);
int dccn_rx_train_step(const dccn_rx_shape* shape, const dccn_rx_buffers* buf,
                       dccn_adam_hparams hp, dccn_stream_t stream);
/* R0 (+R8 partial sums) of buf->x into buf->x_norm, as the first launch of a step does it: primes the pipelined
   mode (x_prenormalised) before its first call.  Workspace = the training step's. */
int dccn_rx_normalise(const dccn_rx_shape* shape, const dccn_rx_buffers* buf, dccn_stream_t stream);
/* hipGraph-captured replay of the same sequences */
typedef struct dccn_rx_graph dccn_rx_graph;
/* mode: bit0 = train (else eval), bit1 = run the dense weight-gradient branch on a forked stream */
int dccn_rx_graph_create(const dccn_rx_shape* shape, const dccn_rx_buffers* buf, int mode,
                         dccn_adam_hparams hp, dccn_stream_t stream, dccn_rx_graph** out);
int dccn_rx_graph_launch(dccn_rx_graph* g, dccn_stream_t stream);
int dccn_rx_graph_destroy(dccn_rx_graph* g);

/* ---- measurement helpers (HIP events on the caller's stream) ------------------------ */
typedef struct dccn_timer dccn_timer;
int dccn_timer_create(dccn_timer** out);
int dccn_timer_start(dccn_timer* t, dccn_stream_t stream);
int dccn_timer_stop(dccn_timer* t, dccn_stream_t stream);
int dccn_timer_elapsed_ms(dccn_timer* t, float* ms);     /* synchronises on the stop event */
int dccn_timer_destroy(dccn_timer* t);
int dccn_stream_synchronize(dccn_stream_t stream);

/* ---- per-step monitors of the equaliser harness (dev/py/ofdmreceiver_np_mp.py:245, 325-333: `chan_rms`; :411-425: the
 * per-epoch means of ce_mean, berlin, tx_power, noise power and chan_rms) in ONE launch -------------------------------
 * chan_rms = mean((LN(chan) - LN(chest))^2), LN = LayerNormalization(axis=1, center=False, scale=False, eps 1e-3) over the
 * OFDM-symbol axis of the [B, S, K, 2] views.  chan is [B, S, K, 2] (chan_per_symbol = 1) or one row per frame [B, K, 2]
 * (static channels: its LN is 0, as in the reference).  acc5 (nullable) += {metrics->ce_mean, metrics->berlin, *tx_power,
 * *noise_power, chan_rms}; rms_out (nullable) = chan_rms.  The workspace (dccn_eq_monitor_workspace_size bytes) must be
 * ZERO before the first call and belongs to the calls of one stream; each call leaves it ready for the next. */
size_t dccn_eq_monitor_workspace_size(int B, int S, int K);
int dccn_eq_monitor_accumulate(const float* chest, const float* chan, int chan_per_symbol, int B, int S, int K,
                               const dccn_metrics* metrics, const float* tx_power, const float* noise_power, float* acc5,
                               float* rms_out, void* workspace, size_t workspace_bytes, dccn_stream_t stream);

/* ---- step timeline: in-situ start / end of the launches of dccn_rx_train_step / dccn_rx_eval_step ----------------------
 * Replaces nothing in the reference (TF1 has `RunMetadata` step stats for this: dev/py/ofdmreceiver_np.py:234 runs the
 * step without them); it exists so that a step time can be decomposed into per-launch durations and the gaps between
 * launches on whatever box the step runs on (bench.py `step.boundaries`).
 * `buf` is DEVICE memory of dccn_step_trace_bytes(ring_steps) bytes, zeroed by the caller: a ring of `ring_steps` steps x
 * `launches` slots x `blocks` workgroups x `words` 64-bit words (dccn_step_trace_geometry).  While enabled, every step call
 * takes the next ring entry and thread 0 of each workgroup of its instrumented launches writes
 *   {s_memrealtime at entry (100 MHz), s_memtime at entry (shader cycles), the same two at exit}
 * into [step % ring][slot][blockIdx.x].  Slots: 1 C-Conv forward, 2 dense forward (+ tail), 3 tail (own launch),
 * 4 backward, 5 C-Conv weight gradient (own launch), 6 optimizer; 0 and 7 unused.  buf = NULL disables (the default: a
 * disabled mark is one scalar compare per workgroup).  Process-wide switch, meant for one measuring thread. */
size_t dccn_step_trace_bytes(int ring_steps);
int dccn_step_trace_enable(unsigned long long* buf, size_t bytes, int ring_steps);
long long dccn_step_trace_steps(void);                 /* step calls recorded since the last enable */
void dccn_step_trace_geometry(int* launches, int* blocks, int* words);

/* ==== channel-equaliser stage: dev/py/model.py:349-478 =====================
 * The stage's dense / C-Conv layers use dccn_dense_* / dccn_cconv_gemm_* above; these are the
 * remaining operators.  All tensors fp32, IQ pairs interleaved on the last axis. */

/* model.py:363  tf.contrib.layers.layer_norm(center=False, scale=False, begin_norm_axis=1):
 * per-row (sample) y = x*inv + (-mean*inv), inv = rsqrt(var + eps), biased variance over `cols`.
 * mean, inv [rows] nullable (inv is what dccn_layer_norm_bwd needs). */
int dccn_layer_norm_fwd(const float* x, float* y, float* mean, float* inv, int rows, int cols, float eps,
                        dccn_stream_t stream);
int dccn_layer_norm_bwd(const float* dy, const float* y, const float* inv, float* dx, int rows, int cols,
                        dccn_stream_t stream);

/* model.py:421-426  activation=tf.nn.tanh of the channel-estimate dense layer; bwd: dx = dy*(1-y^2). */
int dccn_tanh_fwd(const float* x, float* y, long long n, dccn_stream_t stream);
int dccn_tanh_bwd(const float* dy, const float* y, float* dx, long long n, dccn_stream_t stream);

/* model.py:431-438  eq = y*conj(h)/|h| and corr = eq*conj(eq) over [n_pairs,2] (corr nullable).
 * bwd: cotangents d_eq / d_corr (either nullable) -> dy / dh (either nullable). */
int dccn_equalize_fwd(const float* y, const float* h, float* eq, float* corr, long long n_pairs,
                      dccn_stream_t stream);
int dccn_equalize_bwd(const float* y, const float* h, const float* d_eq, const float* d_corr, float* dy,
                      float* dh, long long n_pairs, dccn_stream_t stream);

/* model.py:465-475  pilot "SNR" monitor: eq [frames,S,K,2], carriers: device int32[P] -> snr_db [frames]
 * = log10(clip(mean/var of |pilot cell|^2, 1e-3, 1e4)). */
int dccn_pilot_snr(const float* eq, const int* carriers, float* snr_db, int frames, int S, int K, int P,
                   dccn_stream_t stream);

/* model.py:428  layers_conv2d_complex(chest, 1, (kL,kW), padding='same') on a one-channel L x W
 * complex image, lowered to a dense layer: expand w [kL,kW,2] (=[Wa|Wb] per tap, the TF kernel
 * [kL,kW,1,1,2]) and bias [2] (nullable) into T [L*W*2, L*W*2] and bias_eff [L*W*2] (nullable);
 * run dccn_dense_fwd/bwd with them; reduce dT / dbias_eff back onto dw [kL,kW,2] / dbias [2]
 * (dbias, dbias_eff nullable). */
int dccn_cconv2d_same_expand(const float* w, const float* bias, float* T, float* bias_eff, int L, int W,
                             int kL, int kW, dccn_stream_t stream);
int dccn_cconv2d_same_reduce(const float* dT, const float* dbias_eff, float* dw, float* dbias, int L, int W,
                             int kL, int kW, dccn_stream_t stream);

/* ---- the fused equaliser transfer-learning step ------------------------------------------------------
 * dev/py/ofdmreceiver_np_mp.py:283-330,414 (session.run(train_op...)) and :87 (evaluation run) as ONE
 * pre-planned launch sequence: R0 normalise -> equalizer_ofdm -> frozen basic receiver -> loss/BER ->
 * backward to the Equalizer/ variables only -> TF Adam on the equaliser arena.  cp = 0: the first dense layer and
 * the receiver read the K samples behind the cyclic prefix (model.py:364-366, 1236-1240).
 * Equaliser parameter arena, TF creation order (floats):
 *   dense k[2n_sc (cp=1) or 2K (cp=0), 2K] b | conv3d k[K,2K] b | dense_1 k[S*K*2,2*pilot_size] b | dense_2 | dense_3 |
 *   dense_4 | conv3d_1 k[S,K,2] b[2] | conv3d_2 k[K,2K] b | conv3d_3 | dense_5 k[4K,2n_sc] b
 * (offsets[0..19] = start of each tensor, offsets[20] = total).  rx_params: the frozen receiver in the
 * dccn_rx_param_offsets layout with kin = K + CP (cp=1) or K (cp=0). */
typedef struct dccn_eq_shape {
    int batch, S, K, CP, cp;      /* frames, OFDM symbols per frame, nfft, cyclic prefix, FLAGS.cp */
    int F, D, nbits;              /* receiver: nfilter, data cells per frame, bits per cell */
    int pilot_size, P;            /* pilot cells per frame (model.py:359), pilot carriers per symbol */
} dccn_eq_shape;

struct dccn_eq_monitor;
typedef struct dccn_eq_buffers {
    const float* x;               /* [batch,S,n_sc,2] raw `tx_ofdm` */
    const int32_t* bits;          /* [batch,D,nbits] */
    float* eq_params;             /* equaliser arena (updated by the train step) */
    float* eq_grads;              /* train: gradient arena (data term; L2 enters in the optimizer) */
    float* adam_m;
    float* adam_v;
    const float* reg_coef;        /* nullable, see dccn_adam_tf_step (gate = 1) */
    dccn_adam_state* adam;
    const float* rx_params;       /* frozen receiver arena */
    float* out_eq;                /* [batch,S,n_sc,2] equalised receiver input (model.py:463) */
    float* chest;                 /* [batch,S,K,2] channel estimate (model.py:477) */
    float* snr_db;                /* [batch] nullable (model.py:465-475) */
    const int* pilot_carriers;    /* device int32[P], nullable with snr_db */
    float* prob;                  /* [batch,D,nbits,2] nullable */
    dccn_metrics* metrics;
    float* tx_power;              /* device float[1], nullable */
    void* workspace;
    size_t workspace_bytes;
    int reg_uniform;              /* 1: reg_coef holds ONE value over each dense kernel / bias tensor (keras l2(0.01) on every dense
                                     layer, model.py:371-462) and is not consulted for the C-Conv tensors' segments other than
                                     element-wise: the optimizer launch reads one coefficient per dense tensor instead of
                                     streaming a parameter-sized array.  0: per-element coefficients everywhere. */
    const float* rx_folded;       /* nullable: dccn_eq_rx_fold(rx_params) -- the frozen receiver's C-Conv and dense layer as one
                                     matrix.  When given, few-row batches (<= 96 frames) run ONE GEMM where the step ran the two
                                     layers (and one on the way back); must be rebuilt whenever rx_params change. */
    /* Pipelined input normalisation (training, round 4; same idea as dccn_rx_buffers.x_next): the step's first launch -- the
       batch normalisation of `input:0` -- depends on nothing but x, so the PREVIOUS step can run it for this batch on leading
       workgroups of its optimizer launch.
       x_next != NULL   this step also normalises x_next (same shape as x) into the workspace; honoured when
                        dccn_eq_norm_rides(shape) is 1, otherwise ignored
       x_prenormalised  1: the workspace already holds the normalisation of x (and its R8 partial sums in slot norm_slot),
                        written by the previous call through x_next on the SAME workspace: the step starts at the layer norm
       norm_slot        0/1: the R8 partial-sum slot of THIS batch; x_next's sums go to the other one, so consecutive
                        pipelined calls alternate it.  Results are bit-identical to un-pipelined calls. */
    const float* x_next;
    int x_prenormalised;
    int norm_slot;
    /* Round 5: the next batch as the fused generator left it (host pointer to its descriptor, read during the call / at graph
       capture): instead of x_next the optimizer launch normalises y / sqrt(mean |y|^2) + noise formed in registers from
       (y, noise, power_partial) -- the values dccn_gen_static_apply would write, so results keep their bits -- and, when
       noise_partial and noise_power_out are set, finishes the noise-power monitor.  The CALLER issues dccn_gen_static_frames
       for that batch on the same stream before this call (the step itself launches nothing for it: the call stays
       graph-capturable, the generator's per-batch arguments stay outside the graph).  Honoured like x_next. */
    const dccn_gen_static* x_next_virtual;
    const int* tuning;            /* nullable: the plan's own tuning table, as dccn_rx_buffers.tuning */
    /* Round 6: 1 = the step ALSO produces the batch x_next_virtual describes (the descriptor fully armed: bits_out, snr_db,
       offset, seed, H_out ...): the generator's workgroups ride on the step's pilot-bottleneck backward launch (15 us of VALU
       work beside MFMA tiles and HBM streams instead of a launch of its own in front of the step; a launch of its own when the
       plan has no such launch).  The caller no longer issues dccn_gen_static_frames for that batch.  Same draws, same bits. */
    int gen_next_rides;
    /* Round 6, nullable: the training loop's per-step monitors (dccn_eq_monitor_accumulate: chan_rms of THIS step's channel
       estimate against `chan`, and {ce_mean, berlin, tx_power, noise_power, chan_rms} added onto acc5) as part of the step's
       optimizer launch instead of a launch of their own behind the step.  chest / metrics / tx_power must be the step's own.
       Same additions of the same values: the accumulators keep their bits. */
    const struct dccn_eq_monitor* monitor;
} dccn_eq_buffers;
/* 1: dccn_eq_train_step honours dccn_eq_buffers.x_next for this shape */
int dccn_eq_norm_rides(const dccn_eq_shape* shape);

/* The frozen receiver's linear part folded into one matrix: out [S*2n_sc*2D + 2D] floats = Mf [S*2n_sc, 2D] then bf [2D], with
 * z = out_eq_flat . Mf + bf  ==  dense(C-Conv(out_eq)) of dev/py/model.py:1246-1275 (rows of cyclic-prefix samples are zero
 * when cp = 0).  Valid as long as rx_params do not change (they are frozen: ofdmreceiver_np_mp.py:330 trains Equalizer/ only). */
size_t dccn_eq_rx_folded_floats(const dccn_eq_shape* shape);
int dccn_eq_rx_fold(const dccn_eq_shape* shape, const float* rx_params, float* out, dccn_stream_t stream);

/* offsets[i] = first float of tensor i, offsets[20] = ARENA size.  Every tensor starts on a 16-byte boundary: the arena has
 * up to three floats of PADDING behind a tensor whose element count is not a multiple of four (conv3d_1's two-float bias),
 * so offsets[i+1] - offsets[i] is NOT the tensor's size and offsets[20] is not the parameter count -- take sizes from the
 * shapes above.  The padding floats of eq_params, eq_grads, adam_m, adam_v and reg_coef must be ZERO when the arenas are
 * created (zero gradient and zero Adam state keep them zero): the optimizer launch and the merged element-wise jobs stream
 * over whole arena segments, padding included. */
int dccn_eq_param_offsets(const dccn_eq_shape* shape, long long* offsets /* [21] */);
size_t dccn_eq_workspace_size(const dccn_eq_shape* shape, int train);
/* Where a named intermediate of the fused step lives inside the caller's workspace (the step never reuses a buffer within a
 * call, so every one of them holds what its stage left there when the call returns -- the counterpart of fetching a named
 * tensor of the reference's graph, dev/py/model.py:349-478 by line):
 *   forward   "x_norm" input:0 [B,S,n_sc,2] | "ln" :363 | "t1" :371 dense [B*S,2K] | "y" :378-386 [B,S,K,2] | "d1" :394 [B,2P] |
 *             "d2" :402 | "d3" :408 | "d4" :421 (after tanh) [B,S*K*2] | "T" / "be" the :428 kernel expanded to a dense layer
 *             [SK2,SK2] / [SK2] | "eq" :435 | "corr" :438 | "cat" :456 [B*S,K,4] | "fft" / "z" receiver (unless folded) | "dz"
 *   backward  "dout" d loss / d equalized [B,S,n_sc,2] | "deqc" / "dcorc" gradients of the two :439-449 C-Conv outputs |
 *             "deq" / "dcorr" | "dy" (from the equalise stage only) | "dh" | "dd4" (through the tanh) | "dd3" | "dd2" |
 *             "dflat" (total gradient of y: "dy" + the pilot branch) | "dt1" | "dT" / "dbe"
 * Returns DCCN_OK and (*byte_offset, *count floats), or DCCN_ERR_INVALID_ARG for an unknown name / a training-only tensor of
 * an evaluation workspace.  Which buffers a given launch plan leaves un-written (e.g. "dcat" when the concat's gradient is
 * split in the GEMM's store) is the plan's business: compare only what the plan's description says it materialises. */
int dccn_eq_workspace_tensor(const dccn_eq_shape* shape, int train, const char* name, size_t* byte_offset, size_t* count);
int dccn_eq_eval_step(const dccn_eq_shape* shape, const dccn_eq_buffers* buf, dccn_stream_t stream);
int dccn_eq_train_step(const dccn_eq_shape* shape, const dccn_eq_buffers* buf, dccn_adam_hparams hp,
                       dccn_stream_t stream);
/* ---- chain groups: several independent equaliser training chains carried by ONE launch sequence ------------------------
 * The reference driver trains one (receiver -> equaliser) chain per modulation and per cp / longcp variant, each as an OS process
 * of its own (dev/py/run_local_ofdm.py:61-118, locals.py:28-38; the loop is ofdmreceiver_np_mp.py:394-466).  A 73-frame
 * equaliser step is a chain of ~21 dependent launches that keeps a few percent of an MI355X busy; here n_chains (<=
 * dccn_chain_group_max() = 8) such chains of the SAME shape share every launch: the chain index is a grid dimension and
 * every kernel adds that chain's arena offset to its pointer arguments.  The chains may differ in modulation (nbits): the
 * demodulation tail of the frozen receiver -- the only stage whose kernels depend on it -- is launched once per distinct nbits.
 *
 * Layout contract (checked; DCCN_ERR_INVALID_ARG otherwise): for every pointer field f of dccn_eq_buffers (and of the
 * dccn_gen_static behind x_next_virtual), bufs[g]->f - bufs[0]->f is the SAME byte offset off[g] for all fields, a multiple of
 * 256: each chain keeps ALL its device buffers in one arena with chain 0's internal layout (dl_ofdm_amd/equalizer.py
 * ChainArena).  `prob` must be NULL; rx_folded must be given.  Every chain computes exactly what dccn_eq_train_step would
 * compute for it alone: same kernels, same blocks, same summation orders -- bitwise equal parameters, Adam slots and
 * metrics (tests/test_gpu_chain_groups.py).  n_chains == 1 is dccn_eq_train_step.
 * dccn_eq_group_supported: 1 for the shapes whose step consists of group-capable launches only (<= 96 frames: the few-row plan);
 * a grouped call that would reach any other launch returns DCCN_ERR_UNSUPPORTED before issuing it. */
int dccn_chain_group_max(void);
int dccn_eq_group_supported(const dccn_eq_shape* shape);
int dccn_eq_train_step_grouped(int n_chains, const dccn_eq_shape* const* shapes, const dccn_eq_buffers* const* bufs,
                               dccn_adam_hparams hp, dccn_stream_t stream);
/* the fused generator launch (dccn_gen_static_frames) / the launch that materialises x (dccn_gen_static_apply) for every chain of
 * a group: per-chain seed, offset and nbits, same layout contract on the descriptors' pointers (x_out / noise_power included) */
int dccn_gen_static_frames_grouped(int n_chains, const dccn_gen_static* const* g, dccn_stream_t stream);
int dccn_gen_static_apply_grouped(int n_chains, const dccn_gen_static* const* g, float* const* x_out, float* const* noise_power,
                                  dccn_stream_t stream);
/* dccn_eq_monitor_accumulate for every chain of a group (arguments as that function's, one struct per chain) */
typedef struct dccn_eq_monitor {
    const float* chest; const float* chan;
    int chan_per_symbol, B, S, K;
    const dccn_metrics* metrics;
    const float* tx_power; const float* noise_power;
    float* acc5; float* rms_out;
    void* workspace; size_t workspace_bytes;
} dccn_eq_monitor;
int dccn_eq_monitor_accumulate_grouped(int n_chains, const dccn_eq_monitor* const* m, dccn_stream_t stream);

/* mode bit0: train.  The handle is a dccn_rx_graph: launch / destroy it with dccn_rx_graph_launch /
 * dccn_rx_graph_destroy. */
int dccn_eq_graph_create(const dccn_eq_shape* shape, const dccn_eq_buffers* buf, int mode, dccn_adam_hparams hp,
                         dccn_stream_t stream, dccn_rx_graph** out);

/* ==== device-side input generator ================================================
 * What the reference computes per batch on the host with NumPy, written straight into device buffers.
 * Random streams: Philox4x32-10, counter = (index lo, index hi, stream, offset), key = seed; a batch is
 * a pure function of (seed, offset).  Streams: 0 label bits, 1 channel taps, 2 noise. */

/* test hook: out[4*i..4*i+3] = Philox words of counter (i, stream, offset) */
int dccn_philox_fill(uint32_t* out, long long n, unsigned stream, unsigned offset, unsigned long long seed,
                     dccn_stream_t stream_handle);

/* dev/py/util.py:25-29 bit_source + dev/py/ofdm.py:328-380 ofdm_tx_frame_np.
 * bits_in [frames,D,nbits] nullable (null: draw the labels and store them to bits_out, nullable otherwise);
 * cell_map int32[S*K]: >= 0 data-cell index, -1 pilot, -2 empty; const_tab [2^nbits,2] (MSB-first index);
 * idft [2K, 2(K+CP)]: real form of ifft + cyclic prefix; grid_ws [frames,S,K,2] scratch;
 * tx [frames,S,K+CP,2] time-domain frames. */
int dccn_ofdm_tx_frames(const int32_t* bits_in, int32_t* bits_out, const int32_t* cell_map, const float* const_tab,
                        float pilot_re, float pilot_im, const float* idft, float* grid_ws, float* tx, int frames,
                        int S, int K, int CP, int D, int nbits, unsigned long long seed, unsigned offset,
                        dccn_stream_t stream);

/* dev/py/radio.py:352-372 (static Rayleigh taps, 'same' FIR over the frame) + :513-526 AWGN_channel_np.
 * tx [frames,T,2]; taps_in [frames,n_taps,2] standard normals, nullable (null: draw); coeff [n_taps];
 * alpha [n_taps,L] (L <= 64, n_taps <= 16); identity != 0: pass-through channel (AWGN only);
 * snr_db [frames]; noise_in [frames,T,2] standard normals, nullable (null: draw);
 * out [frames,T,2] = y/sqrt(mean|y|^2) + noise; H [frames,nfft,2] nullable = fft(impulse response, nfft);
 * noise_power device float[1] nullable. */
size_t dccn_channel_awgn_workspace_size(int frames, int T, int L);
int dccn_channel_awgn(const float* tx, const float* taps_in, const float* coeff, const float* alpha, int n_taps,
                      int L, int identity, const float* snr_db, const float* noise_in, float* out, float* H, int nfft,
                      float* noise_power, int frames, int T, unsigned long long seed, unsigned offset,
                      void* workspace, size_t workspace_bytes, dccn_stream_t stream);

/* dev/py/radio.py:376-407 mobile (Doppler) variant: per OFDM symbol the taps follow Jakes' sum of 48
 * sinusoids (maximum Doppler Fd [Hz], symbol period t_sym = n_sc / Fs [s]) and each symbol is filtered with its own
 * impulse response over [n_taps samples of history | the symbol].  theta_in [frames,2,48,n_taps] uniform phases,
 * nullable (null: draw, stream 3).  H [frames,S,nfft,2] nullable.  Other arguments as dccn_channel_awgn; T = S*n_sc. */
size_t dccn_channel_doppler_awgn_workspace_size(int frames, int T, int L, int S);
int dccn_channel_doppler_awgn(const float* tx, const float* theta_in, const float* coeff, const float* alpha,
                              int n_taps, int L, float Fd, float t_sym, int S, int n_sc, const float* snr_db,
                              const float* noise_in, float* out, float* H, int nfft, float* noise_power, int frames,
                              unsigned long long seed, unsigned offset, void* workspace, size_t workspace_bytes,
                              dccn_stream_t stream);

/* Frame-interleaved channels (dev/py/radio.py:438-470 'mixRayleigh' / 'mixAll'): every frame belongs to one group =
 * (profile, static | Doppler); a group lists its frame indices.  taps_in [frames,16,2] / theta_in [frames,2,48,16]
 * (padded to 16 taps; the Philox tap index also uses stride 16 here), H [frames,S,nfft,2] (static frames repeat
 * their response per symbol, as the reference does).  Fd > 0 selects the Doppler model for the group. */
typedef struct dccn_channel_group {
    const int* frames;      /* device int32[n_frames] */
    int n_frames;
    const float* coeff;     /* device [n_taps] */
    const float* alpha;     /* device [n_taps, L] */
    int n_taps, L;
    int identity;           /* pass-through slot ('mixAll' frame % 5 == 0) */
    float Fd;               /* maximum Doppler [Hz]; 0 = static taps */
} dccn_channel_group;
size_t dccn_channel_groups_awgn_workspace_size(int frames, int T, int S);
int dccn_channel_groups_awgn(const float* tx, const dccn_channel_group* groups /* host array */, int n_groups,
                             const float* taps_in, const float* theta_in, float t_sym, int S, int n_sc,
                             const float* snr_db, const float* noise_in, float* out, float* H, int nfft,
                             float* noise_power, int frames, unsigned long long seed, unsigned offset,
                             void* workspace, size_t workspace_bytes, dccn_stream_t stream);

/* ==== host utility: CRC32C (Castagnoli), the checksum of TensorFlow's tensor-bundle checkpoints =========
 * (tf.train.Saver files the reference writes at dev/py/ofdmreceiver_np.py:271).
 * Returns the CRC of (previous data ++ data) given the CRC of the previous data (0 to start).  Host code. */
uint32_t dccn_crc32c(uint32_t crc, const void* data, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* DCCN_H_ */
