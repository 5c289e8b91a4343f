// Element-wise and small-reduction kernels of the channel-equaliser stage (SURVEY.md 8(f-1);
// dev/py/model.py:349-478).  The stage's GEMM-shaped layers reuse gemm_f32_mfma.h (dense,
// C-Conv); what is new here is everything around them.  All of it is HBM/latency-bound: one pass
// over the data, float2 (IQ pair) accesses, DPP wave reductions.
#pragma once
#include "common.h"

namespace dccn {

// grid size of the grid-stride element-wise kernels below
static inline unsigned ew_blocks_n(long long n) {
    long long b = ceil_div_ll(n, 256);
    if (b > 8 * kCUs) b = 8 * kCUs;
    return (unsigned)(b < 1 ? 1 : b);
}

// sum over a 256-thread block, result in every thread; `sh` holds >= 4 elements
template <typename T>
__device__ __forceinline__ T block_sum_256(T v, T* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// tf.contrib.layers.layer_norm(center=False, scale=False, begin_norm_axis=1) (model.py:363):
// one block per sample; two-pass moments (mean, then mean of squared differences), then
// tf.nn.batch_normalization's  x*inv + (-mean*inv).
__device__ __forceinline__ void layer_norm_fwd_body(const float* __restrict__ x, float* __restrict__ y,
                                                    float* __restrict__ mean_out, float* __restrict__ inv_out,
                                                    const int cols, const float eps, const int row) {
    __shared__ float sh[4];
    const float* xr = x + (size_t)row * cols;
    float* yr = y + (size_t)row * cols;
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) s += xr[c];
    const float mean = block_sum_256(s, sh) / (float)cols;
    float q = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float d = xr[c] - mean;
        q += d * d;
    }
    const float var = block_sum_256(q, sh) / (float)cols;
    const float inv = 1.0f / sqrtf(var + eps);
    const float shift = -mean * inv;
    for (int c = threadIdx.x; c < cols; c += 256) yr[c] = xr[c] * inv + shift;
    if (threadIdx.x == 0) {
        if (mean_out) mean_out[row] = mean;
        if (inv_out) inv_out[row] = inv;
    }
}
static __global__ __launch_bounds__(256) void layer_norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             float* __restrict__ mean_out,
                                                             float* __restrict__ inv_out, int cols, float eps) {
    layer_norm_fwd_body(x, y, mean_out, inv_out, cols, eps, (int)blockIdx.x);
}

// dx = inv * (dy - mean(dy) - y * mean(dy*y))   (y = normalised output)
static __global__ __launch_bounds__(256) void layer_norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                             const float* __restrict__ inv, float* __restrict__ dx,
                                                             int cols) {
    __shared__ float sh[4];
    const size_t base = (size_t)blockIdx.x * cols;
    float s = 0.f, t = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float g = dy[base + c];
        s += g;
        t += g * y[base + c];
    }
    const float m1 = block_sum_256(s, sh) / (float)cols;
    const float m2 = block_sum_256(t, sh) / (float)cols;
    const float iv = inv[blockIdx.x];
    for (int c = threadIdx.x; c < cols; c += 256) dx[base + c] = iv * (dy[base + c] - m1 - y[base + c] * m2);
}

// activation=tf.nn.tanh of the channel-estimate dense layer (model.py:421-426)
static __global__ __launch_bounds__(256) void tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = tanhf(x[i]);
}
static __global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       float* __restrict__ dx, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float v = y[i];
        dx[i] = dy[i] * (1.0f - v * v);
    }
}

// model.py:431-438: eq = y * conj(h)/|h|,  corr = eq * conj(eq)   (IQ pairs as float2)
__device__ __forceinline__ float2 equalize_one(const float2 yv, const float2 hv) {
    const float a = sqrtf(hv.x * hv.x + hv.y * hv.y);
    const float cr = hv.x / a, ci = (-hv.y) / a;
    return make_float2(yv.x * cr - yv.y * ci, yv.x * ci + yv.y * cr);
}
__device__ __forceinline__ void equalize_fwd_body(const float2* __restrict__ y, const float2* __restrict__ h,
                                                  float2* __restrict__ eq, float2* __restrict__ corr, const long long n,
                                                  const int bx, const int nbx) {
    const long long stride = (long long)nbx * 256;
    for (long long i = (long long)bx * 256 + threadIdx.x; i < n; i += stride) {
        const float2 e = equalize_one(y[i], h[i]);
        const float er = e.x, ei = e.y;
        eq[i] = make_float2(er, ei);
        if (corr) corr[i] = make_float2(er * er - ei * (-ei), er * (-ei) + ei * er);
    }
}
static __global__ __launch_bounds__(256) void equalize_fwd_kernel(const float2* __restrict__ y, const float2* __restrict__ h,
                                                           float2* __restrict__ eq, float2* __restrict__ corr,
                                                           long long n) {
    equalize_fwd_body(y, h, eq, corr, n, (int)blockIdx.x, (int)gridDim.x);
}

// real-valued backward of the pair above; d_corr may be null.  corr = (er^2+ei^2, 0), so only its
// real cotangent reaches eq.
static __global__ __launch_bounds__(256) void equalize_bwd_kernel(const float2* y_, const float2* h_, const float2* d_eq_, const float2* d_corr_,
                                                           float2* dy_, float2* dh_, long long n, const ChainOffs co) {
    const long long coff = co.off[blockIdx.z];                     // chain groups (common.h)
    const float2* __restrict__ y = chain_at(y_, coff);
    const float2* __restrict__ h = chain_at(h_, coff);
    const float2* __restrict__ d_eq = chain_at(d_eq_, coff);
    const float2* __restrict__ d_corr = chain_at(d_corr_, coff);
    float2* __restrict__ dy = chain_at(dy_, coff);
    float2* __restrict__ dh = chain_at(dh_, coff);
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float2 yv = y[i], hv = h[i];
        const float a = sqrtf(hv.x * hv.x + hv.y * hv.y);
        const float cr = hv.x / a, ci = (-hv.y) / a;
        const float er = yv.x * cr - yv.y * ci;
        const float ei = yv.x * ci + yv.y * cr;
        float2 g = d_eq ? d_eq[i] : make_float2(0.f, 0.f);
        if (d_corr) {
            const float gc = d_corr[i].x;
            g.x += 2.0f * er * gc;
            g.y += 2.0f * ei * gc;
        }
        if (dy) dy[i] = make_float2(g.x * cr + g.y * ci, g.y * cr - g.x * ci);
        if (dh) {
            const float dcr = g.x * yv.x + g.y * yv.y;
            const float dci = g.y * yv.x - g.x * yv.y;
            const float da = -(dcr * cr + dci * ci) / a;        // through 1/|h|
            // cr = hr/a, ci = -hi/a, a = |h|:  d hr = dcr/a + da*hr/a,  d hi = -dci/a + da*hi/a
            dh[i] = make_float2(dcr / a + da * cr, -dci / a - da * ci);
        }
    }
}

// model.py:465-475 pilot "SNR" monitor: log10(clip(mean/var of |pilot|^2 over the frame's pilot
// cells)).  One wave per frame.  FROM_YH: the equalised cell is recomputed from (y, h) -- the monitor then rides on the
// launch that writes eq instead of waiting for it in a launch of its own.
template <bool FROM_YH>
__device__ __forceinline__ void pilot_snr_body(const float2* __restrict__ eq, const float2* __restrict__ y,
                                               const float2* __restrict__ h, const int* __restrict__ carriers,
                                               float* __restrict__ snr_db, const int S, const int K, const int P,
                                               const int frame, const int lane) {
    const size_t base = (size_t)frame * S * K;
    const int n = S * P;
    auto cell = [&](int i) -> float2 {
        const size_t j = base + (size_t)(i / P) * K + carriers[i % P];
        if constexpr (FROM_YH) return equalize_one(y[j], h[j]);
        else return eq[j];
    };
    float s = 0.f;
    for (int i = lane; i < n; i += 64) {
        const float2 v = cell(i);
        s += v.x * v.x + v.y * v.y;
    }
    const float mean = wave_sum(s) / (float)n;
    float q = 0.f;
    for (int i = lane; i < n; i += 64) {
        const float2 v = cell(i);
        const float d = (v.x * v.x + v.y * v.y) - mean;
        q += d * d;
    }
    const float var = wave_sum(q) / (float)n;
    const float ratio = fminf(fmaxf(mean / var, 0.001f), 10000.0f);
    if (lane == 0) snr_db[frame] = logf(ratio) / logf(10.0f);
}
static __global__ __launch_bounds__(64) void pilot_snr_kernel(const float2* __restrict__ eq, const int* __restrict__ carriers,
                                                       float* __restrict__ snr_db, int S, int K, int P) {
    pilot_snr_body<false>(eq, nullptr, nullptr, carriers, snr_db, S, K, P, (int)blockIdx.x, (int)threadIdx.x);
}
// equalise + the pilot monitor in one launch: blocks [0, eq_blocks) equalise, each later block serves four frames
static __global__ __launch_bounds__(256) void equalize_fwd_snr_kernel(const float2* __restrict__ y, const float2* __restrict__ h,
                                                               float2* __restrict__ eq, float2* __restrict__ corr,
                                                               long long n, int eq_blocks, const int* __restrict__ carriers,
                                                               float* __restrict__ snr_db, int frames, int S, int K, int P) {
    if ((int)blockIdx.x < eq_blocks) {
        equalize_fwd_body(y, h, eq, corr, n, (int)blockIdx.x, eq_blocks);
        return;
    }
    const int frame = ((int)blockIdx.x - eq_blocks) * 4 + (int)(threadIdx.x >> 6);
    if (frame < frames) pilot_snr_body<true>(nullptr, y, h, carriers, snr_db, S, K, P, frame, (int)(threadIdx.x & 63));
}

// ---- the harness's per-step monitors in one launch (dev/py/ofdmreceiver_np_mp.py:245, 325-333, 411-425) ------------------
// chan_rms = mean((LN(chan) - LN(chest))^2) with LN = tf.keras.layers.LayerNormalization(axis=1, center=False, scale=False)
// over the OFDM-symbol axis of the [B, S, K, 2] real views (moments over S only, Keras' epsilon 1e-3): a column
// (b, k, iq) is S values of each tensor.  A static channel is the same row for every symbol: gt_per_symbol = 0 reads
// chan [B, K, 2] for every s (its LN is exactly 0, as in the reference).  One thread per column, fixed-order block sums,
// per-block partials summed in order by the LAST block to arrive (no block waits for another); that block also adds the
// step's scalars onto the epoch accumulators acc[5] = {ce_mean, berlin, tx_power, noise_power, chan_rms}: what the
// harness otherwise did with ~25 small framework launches per step.
struct EqMonitorArgs {
    const float* chest;         // [B, S, K, 2]
    const float* chan;          // [B, S, K, 2] or [B, K, 2]
    int gt_per_symbol, B, S, K;
    const dccn_metrics* metrics;
    const float* tx_power;      // nullable
    const float* noise_power;   // nullable
    float* acc;                 // [5]
    float* rms_out;             // nullable: this step's chan_rms
    double* partial;            // [blocks]
    unsigned* counter;          // zero before the first launch; the last block leaves it zero again
    __device__ __forceinline__ EqMonitorArgs at_chain(const long long coff) const {      // chain groups (common.h)
        EqMonitorArgs q = *this;
        q.chest = chain_at(chest, coff); q.chan = chain_at(chan, coff); q.metrics = chain_at(metrics, coff);
        q.tx_power = chain_at(tx_power, coff); q.noise_power = chain_at(noise_power, coff); q.acc = chain_at(acc, coff);
        q.rms_out = chain_at(rms_out, coff); q.partial = chain_at(partial, coff); q.counter = chain_at(counter, coff);
        return q;
    }
};
// block of nblocks; scalars: also add {ce_mean, berlin, tx_power, noise_power} onto acc (the stand-alone launch; as a job of the
// optimizer launch those are added by the waves that produce them, tail.h demod_tail_finalize_body)
__device__ __forceinline__ void eq_monitor_body(const EqMonitorArgs& a, const unsigned block, const unsigned nblocks, const bool scalars) {
    __shared__ double red[4];
    __shared__ unsigned s_last;
    const long long cols = (long long)a.B * a.K * 2;
    const int K2 = a.K * 2;
    double sum = 0.0;
    for (long long c = (long long)block * 256 + threadIdx.x; c < cols; c += (long long)nblocks * 256) {
        const int b = (int)(c / K2), r = (int)(c % K2);
        const float* pe = a.chest + (size_t)b * a.S * K2 + r;
        const float* pg = a.gt_per_symbol ? a.chan + (size_t)b * a.S * K2 + r : a.chan + (size_t)b * K2 + r;
        const int gs = a.gt_per_symbol ? K2 : 0;
        float me = 0.f, mg = 0.f;
        for (int s = 0; s < a.S; ++s) { me += pe[(size_t)s * K2]; mg += pg[(size_t)s * gs]; }
        me /= (float)a.S; mg /= (float)a.S;
        float ve = 0.f, vg = 0.f;
        for (int s = 0; s < a.S; ++s) {
            const float de = pe[(size_t)s * K2] - me, dg = pg[(size_t)s * gs] - mg;
            ve += de * de; vg += dg * dg;
        }
        const float ie = rsqrtf(ve / (float)a.S + 1e-3f), ig = rsqrtf(vg / (float)a.S + 1e-3f);
        for (int s = 0; s < a.S; ++s) {
            const float d = (pg[(size_t)s * gs] - mg) * ig - (pe[(size_t)s * K2] - me) * ie;
            sum += (double)(d * d);
        }
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        a.partial[block] = (red[0] + red[1]) + (red[2] + red[3]);
        __threadfence();                                            // the partial is visible before the arrival is
        s_last = atomicAdd(a.counter, 1u) == nblocks - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    // the last block to arrive sums the per-block partials in block order (fixed order => deterministic).  Its threads fetch
    // them side by side first: thread 0 walking <= 256 dependent device-scope loads was 8 of this launch's 12 us
    __shared__ double part[256];
    __threadfence();
    if (threadIdx.x < nblocks) part[threadIdx.x] = __hip_atomic_load(a.partial + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x != 0) return;
    double tot = 0.0;
    for (unsigned i = 0; i < nblocks; ++i) tot += part[i];
    const float rms = (float)(tot / ((double)cols * (double)a.S));
    if (a.rms_out) *a.rms_out = rms;
    if (a.acc) {
        if (scalars) {
            a.acc[0] += a.metrics->ce_mean;
            a.acc[1] += a.metrics->berlin;
            if (a.tx_power) a.acc[2] += *a.tx_power;
            if (a.noise_power) a.acc[3] += *a.noise_power;
        }
        a.acc[4] += rms;
    }
    *a.counter = 0u;
}
static __global__ __launch_bounds__(256) void eq_monitor_kernel(const EqMonitorArgs a0, const ChainOffs co) {
    const EqMonitorArgs a = a0.at_chain(co.off[blockIdx.z]);
    eq_monitor_body(a, blockIdx.x, gridDim.x, true);
}

// ---- the FROZEN receiver's two linear layers as one matrix (equaliser training, ofdmreceiver_np_mp.py:264-330) -------------
// z[b] = sum_s (x_s . Weff) . Wd_s + ... = out_eq_flat[b] . Mf + bf with Mf[s*N2 + k][j] = sum_c Weff[k'][c] Wd[s*2F + c][j]
// (k' = k - win, rows of the cyclic prefix are zero when cp = 0) and bf = bd + sum_s cb . Wd_s.  The receiver's weights do
// not change while the equaliser trains, so Mf is built once; the step then runs ONE few-row GEMM where it ran the
// C-Conv and the dense layer, and its transpose on the way back (dout = dz . Mf^T).  Block r = row of Mf, block `rows` = bf.
static __global__ __launch_bounds__(256) void eq_rx_fold_kernel(const float* __restrict__ cw, const float* __restrict__ cb,
                                                         const float* __restrict__ wd, const float* __restrict__ bd,
                                                         float* __restrict__ Mf, float* __restrict__ bf, int S, int N2,
                                                         int win, int kin, int F, int dN) {
    const int rows = S * N2, r = (int)blockIdx.x, F2 = 2 * F;
    if (r == rows) {
        for (int j = threadIdx.x; j < dN; j += 256) {
            float acc = bd ? bd[j] : 0.f;
            for (int s = 0; s < S; ++s)
                for (int c = 0; c < F2; ++c) {
                    float b = 0.f;
                    if (cb) { const float d = cb[c >> 1] - cb[F + (c >> 1)]; b = (c & 1) ? -d : d; }
                    acc += b * wd[(size_t)(s * F2 + c) * dN + j];
                }
            bf[j] = acc;
        }
        return;
    }
    const int s = r / N2, k = r % N2 - win;
    for (int j = threadIdx.x; j < dN; j += 256) {
        float acc = 0.f;
        if (k >= 0 && k < 2 * kin)
            for (int c = 0; c < F2; ++c) acc += cconv_weff(cw, F, k, c) * wd[(size_t)(s * F2 + c) * dN + j];
        Mf[(size_t)r * dN + j] = acc;
    }
}

// ---- one-channel, one-filter complex "same" convolution as a dense layer ---------------------
// layers_conv2d_complex(chest, 1, (n_sym, K), padding='same') (model.py:428) slides a kL x kW
// complex kernel over an L x W complex image with TF's SAME zero padding.  At L x W = 7 x 64 the
// im2col matrix would be 448x the image, so instead the kernel is expanded ONCE per step into the
// block-Toeplitz matrix T [L*W*2, L*W*2] of the equivalent dense layer (row = input cell/IQ,
// column = output cell/re-im) and the layer runs on the MFMA dense GEMM.  Entry for input (s',k')
// and output (s,k): tap (a,b) = (s'-s+padL, k'-k+padW) if inside the kernel, else 0, with the
// C-Conv sign pattern  [I->re]=Wa [I->im]=Wb [Q->re]=-Wb [Q->im]=-Wa  (complex.py:185-188).
__device__ __forceinline__ void cconv2d_same_expand_body(const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ T, float* __restrict__ bias_eff,
                                                         const int L, const int W, const int kL, const int kW,
                                                         const int bx, const int nbx) {
    const int n = L * W * 2;
    const int padL = (kL - 1) / 2, padW = (kW - 1) / 2;
    const long long total = (long long)n * n;
    const long long stride = (long long)nbx * 256;
    for (long long i = (long long)bx * 256 + threadIdx.x; i < total; i += stride) {
        const int row = (int)(i / n), col = (int)(i % n);
        const int iq = row & 1, kp = (row >> 1) % W, sp = (row >> 1) / W;
        const int ri = col & 1, k = (col >> 1) % W, s = (col >> 1) / W;
        const int a = sp - s + padL, b = kp - k + padW;
        float v = 0.f;
        if (a >= 0 && a < kL && b >= 0 && b < kW) {
            const float wa = w[(a * kW + b) * 2], wb = w[(a * kW + b) * 2 + 1];
            v = iq == 0 ? (ri == 0 ? wa : wb) : (ri == 0 ? -wb : -wa);
        }
        T[i] = v;
        if (bias_eff && i < n) {
            const float d = bias ? bias[0] - bias[1] : 0.f;
            bias_eff[i] = (i & 1) ? -d : d;
        }
    }
}
static __global__ __launch_bounds__(256) void cconv2d_same_expand_kernel(const float* __restrict__ w,
                                                                  const float* __restrict__ bias,
                                                                  float* __restrict__ T, float* __restrict__ bias_eff,
                                                                  int L, int W, int kL, int kW) {
    cconv2d_same_expand_body(w, bias, T, bias_eff, L, W, kL, kW, (int)blockIdx.x, (int)gridDim.x);
}
// the equaliser step's second launch: layer_norm of the normalised frames (one block per frame) and, in the blocks behind
// them, the expansion of the smoothing kernel -- two independent 5 us launches as one
// adam != nullptr: the optimizer's per-step bookkeeping rides here (steps whose normalisation the previous step already ran)
static __global__ __launch_bounds__(256) void eq_prep_kernel(const float* x_, float* y_, int frames, int cols, float eps, const float* w_,
                                                      const float* bias_, float* T_, float* bias_eff_, int L, int W,
                                                      dccn_adam_state* adam_, dccn_adam_hparams hp, const ChainOffs co) {
    const long long coff = co.off[blockIdx.z];                     // chain groups (common.h)
    const float* __restrict__ x = chain_at(x_, coff);
    float* __restrict__ y = chain_at(y_, coff);
    const float* __restrict__ w = chain_at(w_, coff);
    const float* __restrict__ bias = chain_at(bias_, coff);
    float* __restrict__ T = chain_at(T_, coff);
    float* __restrict__ bias_eff = chain_at(bias_eff_, coff);
    dccn_adam_state* __restrict__ adam = chain_at(adam_, coff);
    if (adam != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        adam->alpha = adam_alpha(adam, hp);
        adam->beta1_power = adam->beta1_power * hp.beta1;
        adam->beta2_power = adam->beta2_power * hp.beta2;
        adam->global_step = adam->global_step + 1.0f;
    }
    if ((int)blockIdx.x < frames) layer_norm_fwd_body(x, y, nullptr, nullptr, cols, eps, (int)blockIdx.x);
    else cconv2d_same_expand_body(w, bias, T, bias_eff, L, W, L, W, (int)blockIdx.x - frames, (int)gridDim.x - frames);
}

// transpose of the expansion: one wave per tap gathers its diagonal of dT; wave kL*kW reduces the
// bias cotangent.
static __global__ __launch_bounds__(64) void cconv2d_same_reduce_kernel(const float* __restrict__ dT,
                                                                 const float* __restrict__ dbias_eff,
                                                                 float* __restrict__ dw, float* __restrict__ dbias,
                                                                 int L, int W, int kL, int kW) {
    const int n = L * W * 2;
    const int padL = (kL - 1) / 2, padW = (kW - 1) / 2;
    const int tap = blockIdx.x;
    if (tap == kL * kW) {
        if (!dbias) return;
        float d = 0.f;
        if (dbias_eff)
            for (int c = threadIdx.x; c < L * W; c += 64) d += dbias_eff[2 * c] - dbias_eff[2 * c + 1];
        d = wave_sum(d);
        if (threadIdx.x == 0) {
            dbias[0] = d;
            dbias[1] = -d;
        }
        return;
    }
    const int a = tap / kW, b = tap % kW;
    float ga = 0.f, gb = 0.f;
    for (int c = threadIdx.x; c < L * W; c += 64) {
        const int s = c / W, k = c % W;
        const int sp = s + a - padL, kp = k + b - padW;
        if (sp < 0 || sp >= L || kp < 0 || kp >= W) continue;
        const size_t r0 = (size_t)((sp * W + kp) * 2) * n + (size_t)c * 2;
        ga += dT[r0] - dT[r0 + n + 1];          // d/dWa: (I->re) - (Q->im)
        gb += dT[r0 + 1] - dT[r0 + n];          // d/dWb: (I->im) - (Q->re)
    }
    ga = wave_sum(ga);
    gb = wave_sum(gb);
    if (threadIdx.x == 0) {
        dw[tap * 2] = ga;
        dw[tap * 2 + 1] = gb;
    }
}

// ---- glue of the fused equaliser step ------------------------------------------------------------------
// tf.concat([equalized, corr_re], axis=-1) (model.py:456): two IQ-pair streams -> [n, 4]
static __global__ __launch_bounds__(256) void concat_pairs_kernel(const float2* __restrict__ a, const float2* __restrict__ b,
                                                           float4* __restrict__ out, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float2 u = a[i], v = b[i];
        out[i] = make_float4(u.x, u.y, v.x, v.y);
    }
}
static __global__ __launch_bounds__(256) void split_pairs_kernel(const float4* __restrict__ in, float2* __restrict__ a,
                                                          float2* __restrict__ b, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float4 v = in[i];
        a[i] = make_float2(v.x, v.y);
        b[i] = make_float2(v.z, v.w);
    }
}
// plain zero fill (a kernel rather than hipMemsetAsync so that graph capture sees an ordinary launch)
static __global__ __launch_bounds__(256) void zero_fill_kernel(float* __restrict__ a, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) a[i] = 0.f;
}
// a += b  (the two gradient paths into the frequency-domain input, model.py:383 and :392)
static __global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b,
                                                          long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) a[i] += b[i];
}

}  // namespace dccn
