// Device-side input generator: what the reference produces per batch on the host with
// NumPy -- label bits (dev/py/util.py:25-29), the OFDM transmitter (dev/py/ofdm.py:328-380), a static
// multipath Rayleigh channel (dev/py/radio.py:277-372, 409-470) and AWGN (radio.py:513-526) -- written
// straight into the receiver engine's resident input buffers.  Random streams are Philox4x32-10
// (counter = (index lo, index hi, stream, batch offset), key = seed), so a batch is a pure function of
// (seed, offset) and every thread draws independently.  All of it is one pass over HBM; the IFFT + cyclic
// prefix is a dense MFMA GEMM with a constant [2K, 2(K+CP)] matrix (gemm_f32_mfma.h).
#pragma once
#include <hip/hip_fp16.h>
#include "common.h"

namespace dccn {

constexpr unsigned kPhiloxM0 = 0xD2511F53u, kPhiloxM1 = 0xCD9E8D57u, kPhiloxW0 = 0x9E3779B9u, kPhiloxW1 = 0xBB67AE85u;
constexpr int kStreamBits = 0, kStreamTaps = 1, kStreamNoise = 2, kStreamDoppler = 3;

struct Philox4 {
    unsigned v[4];
};
__device__ __forceinline__ Philox4 philox4x32_10(unsigned long long index, unsigned stream, unsigned offset,
                                                 unsigned long long seed) {
    unsigned c0 = (unsigned)index, c1 = (unsigned)(index >> 32), c2 = stream, c3 = offset;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(kPhiloxM0, c0), lo0 = kPhiloxM0 * c0;
        const unsigned hi1 = __umulhi(kPhiloxM1, c2), lo1 = kPhiloxM1 * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += kPhiloxW0;
        k1 += kPhiloxW1;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}
__device__ __forceinline__ float uniform01(unsigned w) { return ((float)(w >> 8) + 0.5f) * 5.9604644775390625e-8f; }
__device__ __forceinline__ float2 box_muller(unsigned w0, unsigned w1) {
    const float r = sqrtf(-2.0f * logf(uniform01(w0)));
    float sn, cs;
    sincosf(6.2831853071795864769f * uniform01(w1), &sn, &cs);
    return make_float2(r * cs, r * sn);
}

// test hook: out[i] = the four words of counter (i, stream, offset)
__global__ __launch_bounds__(256) void philox_fill_kernel(unsigned* __restrict__ out, long long n, unsigned stream,
                                                          unsigned offset, unsigned long long seed) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Philox4 p = philox4x32_10((unsigned long long)i, stream, offset, seed);
#pragma unroll
    for (int j = 0; j < 4; ++j) out[i * 4 + j] = p.v[j];
}

// util.py:25-29 bit_source + ofdm.py:339-356 constellation / pilot / guard mapping: one thread per resource
// cell.  cell_map[S*K]: >= 0 data-cell index d, -1 pilot, -2 empty.  bits_in == nullptr: draw the label bits
// (bit j of cell (frame,d) = bit j of Philox word 0 at index frame*D+d) and store them to bits_out;
// the constellation index is MSB-first over the nbits labels (ofdm.py:121-153 const_map order).
__global__ __launch_bounds__(256) void tx_grid_kernel(const int* __restrict__ bits_in, int* __restrict__ bits_out,
                                                      const int* __restrict__ cell_map,
                                                      const float2* __restrict__ const_tab, float2 pilot,
                                                      float2* __restrict__ grid, long long n_cells, int SK, int D,
                                                      int nbits, unsigned offset, unsigned long long seed) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_cells) return;
    const long long frame = i / SK;
    const int d = cell_map[(int)(i - frame * SK)];
    float2 v = make_float2(0.f, 0.f);
    if (d == -1) {
        v = pilot;
    } else if (d >= 0) {
        const long long cell = frame * D + d;
        unsigned word = 0;
        if (!bits_in) word = philox4x32_10((unsigned long long)cell, kStreamBits, offset, seed).v[0];
        int idx = 0;
        for (int j = 0; j < nbits; ++j) {
            const int b = bits_in ? (bits_in[cell * nbits + j] & 1) : (int)((word >> j) & 1u);
            if (bits_out) bits_out[cell * nbits + j] = b;
            idx = (idx << 1) | b;
        }
        v = const_tab[idx];
    }
    grid[i] = v;
}

// radio.py:352-372 static taps: tap_k = (z0 + i z1)/sqrt(2) * coeff_k, impulse response g = taps . alpha
// ([n_taps, L] sinc-interpolation matrix), H = fft(g, nfft).  One block (64 threads) per frame.
// taps_in (standard normals [n, n_taps, 2]) == nullptr: draw them.  identity != 0: g = [1] (AWGN channel).
// frames (nullable): the frame indices this launch covers (frame-interleaved 'mix' channels run one launch per
// profile); tap_stride: taps per frame in taps_in and in the Philox index; g_stride: float2 per frame in g;
// h_rep: copies of H per frame (the mix channels report H per symbol for static frames too).
__global__ __launch_bounds__(64) void channel_taps_kernel(const float* __restrict__ taps_in,
                                                          const float* __restrict__ coeff,
                                                          const float* __restrict__ alpha, float2* __restrict__ g,
                                                          float2* __restrict__ H, int n_taps, int L, int nfft,
                                                          int identity, unsigned offset, unsigned long long seed,
                                                          const int* __restrict__ frames, int tap_stride, int g_stride,
                                                          int h_rep) {
    __shared__ float2 tap[16];
    __shared__ float2 gs[64];
    const int fr = frames ? frames[blockIdx.x] : (int)blockIdx.x, t = threadIdx.x;
    if (identity) {
        if (t < L) gs[t] = make_float2(t == 0 ? 1.f : 0.f, 0.f);
    } else {
        if (t < n_taps) {
            float2 z;
            if (taps_in) {
                z = make_float2(taps_in[((size_t)fr * tap_stride + t) * 2], taps_in[((size_t)fr * tap_stride + t) * 2 + 1]);
            } else {
                const Philox4 p = philox4x32_10((unsigned long long)fr * tap_stride + t, kStreamTaps, offset, seed);
                z = box_muller(p.v[0], p.v[1]);
            }
            const float c = coeff[t] * 0.70710678118654752440f;
            tap[t] = make_float2(z.x * c, z.y * c);
        }
        __syncthreads();
        if (t < L) {
            float2 a = make_float2(0.f, 0.f);
            for (int k = 0; k < n_taps; ++k) {
                const float w = alpha[k * L + t];
                a.x += tap[k].x * w;
                a.y += tap[k].y * w;
            }
            gs[t] = a;
        }
    }
    __syncthreads();
    if (t < L) g[(size_t)fr * g_stride + t] = gs[t];
    if (H) {
        for (int f = t; f < nfft; f += 64) {
            float2 a = make_float2(0.f, 0.f);
            for (int l = 0; l < L; ++l) {
                float sn, cs;
                sincosf(-6.2831853071795864769f * (float)((f * l) % nfft) / (float)nfft, &sn, &cs);
                a.x += gs[l].x * cs - gs[l].y * sn;
                a.y += gs[l].x * sn + gs[l].y * cs;
            }
            for (int r = 0; r < h_rep; ++r) H[((size_t)fr * h_rep + r) * nfft + f] = a;
        }
    }
}

// radio.py:360-366: y = np.convolve(frame, g, 'same') over the frame's T = n_sym*n_sc samples
// (y[t] = sum_l g[l] x[t + off - l], off = (L-1)//2, zero outside the frame) + per-block partial sums of
// |y|^2 for the AWGN stage's power normalisation.  grid = (ceil(T/256), frames).
// Persistent form: the grid is a fixed number of blocks (<= kChanPartials) that each walk a run of consecutive
// (frame, 256-sample chunk) items and leave ONE partial sum of |y|^2 each, in a fixed order -- few enough for the AWGN kernel to add
// them up itself (the separate sum_partials launch of rounds 1-2 is gone).  pbase: this launch's first slot in `partial`.
constexpr int kChanPartials = 2048;
// TapGen (enabled): the block draws the static taps of a frame itself when it moves to that frame -- the same Philox
// draws, coefficients and summation order as channel_taps_kernel, so the impulse response has the same bits; the taps
// launch disappears when nobody asked for H (the generate-and-train loop of the basic receiver).
struct TapGen {
    const float* coeff; const float* alpha;
    int enabled, n_taps, identity, tap_stride;
    unsigned offset; unsigned long long seed;
};
__global__ __launch_bounds__(256) void fir_same_kernel(const float2* __restrict__ x, const float2* __restrict__ g,
                                                       float2* __restrict__ y, double* __restrict__ partial, int T,
                                                       int L, const int* __restrict__ frames, int g_stride, int n_frames,
                                                       int pbase, const TapGen tg, const int ipb) {
    __shared__ double sh[4];
    __shared__ float2 gs[64];
    __shared__ float2 tap[16];
    const int bx = (T + 255) / 256;
    const int off = (L - 1) / 2;
    double pw = 0.0;
    // a block takes a run of consecutive items: mostly chunks of one frame, whose taps are loaded once
    // (ipb = items per block, a whole number of frames when the grid allows: a frame's taps are then set up once)
    const int items = n_frames * bx;
    int cur = -1;
    for (int item = blockIdx.x * ipb; item < min(items, ((int)blockIdx.x + 1) * ipb); ++item) {
        const int fi = item / bx, cb = item - fi * bx;
        const int fr = frames ? frames[fi] : fi;
        if (fr != cur) {                                      // (block-uniform)
            __syncthreads();                                  // previous frame's taps are no longer read
            if (!tg.enabled) {
                if (threadIdx.x < L) gs[threadIdx.x] = g[(size_t)fr * g_stride + threadIdx.x];
            } else if (tg.identity) {
                if (threadIdx.x < L) gs[threadIdx.x] = make_float2(threadIdx.x == 0 ? 1.f : 0.f, 0.f);
            } else {
                const int t = threadIdx.x;
                if (t < tg.n_taps) {
                    const Philox4 p = philox4x32_10((unsigned long long)fr * tg.tap_stride + t, kStreamTaps, tg.offset, tg.seed);
                    const float2 z = box_muller(p.v[0], p.v[1]);
                    const float c = tg.coeff[t] * 0.70710678118654752440f;
                    tap[t] = make_float2(z.x * c, z.y * c);
                }
                __syncthreads();
                if (t < L) {
                    float2 a = make_float2(0.f, 0.f);
                    for (int k = 0; k < tg.n_taps; ++k) {
                        const float w = tg.alpha[k * L + t];
                        a.x += tap[k].x * w;
                        a.y += tap[k].y * w;
                    }
                    gs[t] = a;
                }
            }
            __syncthreads();
            cur = fr;
        }
        const int t = cb * 256 + threadIdx.x;
        if (t < T) {
            const float2* xf = x + (size_t)fr * T;
            float2 a = make_float2(0.f, 0.f);
            for (int l = 0; l < L; ++l) {
                const int u = t + off - l;
                if (u < 0 || u >= T) continue;
                const float2 v = xf[u];
                a.x += gs[l].x * v.x - gs[l].y * v.y;
                a.y += gs[l].x * v.y + gs[l].y * v.x;
            }
            y[(size_t)fr * T + t] = a;
            pw += (double)a.x * a.x + (double)a.y * a.y;
        }
    }
    pw = wave_sum(pw);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = pw;
    __syncthreads();
    if (threadIdx.x == 0) partial[pbase + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// radio.py:513-526 AWGN_channel_np: out = y / sqrt(mean |y|^2 over the batch) + noise * sqrt(0.5) 10^(-SNR/20),
// SNR per frame; mean |y|^2 = sum of the FIR stage's block partials / total, added up by every block itself.
// noise_in (standard normals [n, T, 2]) == nullptr: draw them.  Also emits the per-block partial sums of the
// noise power (finished by sum_partials_kernel).  grid = (ceil(T/256), frames).
__global__ __launch_bounds__(256) void awgn_kernel(const float2* __restrict__ y, const double* __restrict__ power_partial,
                                                   int n_partial, double total,
                                                   const float* __restrict__ snr_db,
                                                   const float* __restrict__ noise_in, float2* __restrict__ out,
                                                   double* __restrict__ noise_partial, int T, unsigned offset,
                                                   unsigned long long seed) {
    __shared__ double sh[4];
    __shared__ float s_inv;
    {   // mean |y|^2 over the batch from the FIR stage's <= kChanPartials block sums: every block adds them in the
        // same fixed order (thread t: t, t+256; DPP wave sum; four wave sums), so all blocks get the same bits
        double a = 0.0;
        for (int i = threadIdx.x; i < n_partial; i += 256) a += power_partial[i];
        a = wave_sum(a);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) s_inv = 1.0f / sqrtf((float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) / total));
        __syncthreads();
    }
    const float inv_scale = s_inv;
    const int fr = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    double pw = 0.0;
    if (t < T) {
        const size_t i = (size_t)fr * T + t;
        const float std_ = 0.70710678118654752440f * exp10f(-snr_db[fr] * 0.05f);
        float2 z;
        if (noise_in) {
            z = make_float2(noise_in[2 * i], noise_in[2 * i + 1]);
        } else {
            const Philox4 p = philox4x32_10((unsigned long long)i, kStreamNoise, offset, seed);
            z = box_muller(p.v[0], p.v[1]);
        }
        z.x *= std_;
        z.y *= std_;
        const float2 v = y[i];
        out[i] = make_float2(v.x * inv_scale + z.x, v.y * inv_scale + z.y);
        pw = (double)z.x * z.x + (double)z.y * z.y;
    }
    if (noise_partial) {                                 // (kernel argument: block-uniform)
        pw = wave_sum(pw);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = pw;
        __syncthreads();
        if (threadIdx.x == 0) noise_partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    }
}

// radio.py:62-88 AWGN_channel, the *in-graph* monitor branch of the receiver graph (ofdmreceiver_np.py:136,151-152):
// xn = batch-normalised (eps 1e-8) clipped signal / sqrt(2) (computed by the caller with the R0 kernel),
// level = sqrt(.5) 10^(-SNR/20) per frame, amp = level * N(0,1), phase = U(0, 2 pi),
// noise = (|amp| sin(phase), |amp| cos(phase)); iq_rx = fp16(xn + noise), iq_tx = fp16(clipped signal),
// noise_power = mean(noise_re^2 + noise_im^2).  The receiver never consumes this branch (`rx_iq_data = iq_tx_re`);
// it only feeds the constellation dumps and the printed noise power.  grid = (ceil(T/256), frames).
__global__ __launch_bounds__(256) void ingraph_awgn_kernel(const float2* __restrict__ clipped,
                                                           const float2* __restrict__ xn,
                                                           const float* __restrict__ snr_db, __half2* __restrict__ iq_tx,
                                                           __half2* __restrict__ iq_rx, double* __restrict__ noise_partial,
                                                           int T, unsigned offset, unsigned long long seed) {
    __shared__ double sh[4];
    const int fr = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    double pw = 0.0;
    if (t < T) {
        const size_t i = (size_t)fr * T + t;
        const float level = 0.70710678118654752440f * exp10f(-snr_db[fr] * 0.05f);
        const Philox4 p = philox4x32_10((unsigned long long)i, kStreamNoise, offset, seed);
        const float amp = fabsf(level * box_muller(p.v[0], p.v[1]).x);
        float sn, cs;
        sincosf(6.2831853071795864769f * uniform01(p.v[2]), &sn, &cs);
        const float nr = amp * sn, ni = amp * cs;
        const float2 v = xn[i];
        if (iq_rx) iq_rx[i] = __floats2half2_rn(v.x + nr, v.y + ni);
        if (iq_tx) {
            const float2 c = clipped[i];
            iq_tx[i] = __floats2half2_rn(c.x, c.y);
        }
        pw = (double)nr * nr + (double)ni * ni;
    }
    pw = wave_sum(pw);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = pw;
    __syncthreads();
    if (threadIdx.x == 0) noise_partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// radio.py:376-407 Jakes sum-of-sinusoids Doppler: per OFDM symbol s (t = s * n_sc / Fs) and tap k
//   mu_re = sqrt(1/48) sum_n cos(2 pi t Fd cos(a_n + a0_k) + th_re[n,k]),  a_n = (n - 0.5) pi / (4*48), a0_k = k pi/(4*48)
//   mu_im likewise with cos(a_n - a0_k) and th_im;  tap = (mu_re + i mu_im) coeff_k;  g[s] = taps[s] . alpha;  H[s] = fft(g[s])
// theta_in [frames, 2, 48, n_taps] (uniform phases in [0, 2 pi)) == nullptr: draw them.  One block per frame.
constexpr int kSinusoids = 48;
__global__ __launch_bounds__(64) void doppler_taps_kernel(const float* __restrict__ theta_in,
                                                          const float* __restrict__ coeff,
                                                          const float* __restrict__ alpha, float2* __restrict__ g,
                                                          float2* __restrict__ H, int n_taps, int L, int nfft, int S,
                                                          float Fd, float t_sym, unsigned offset,
                                                          unsigned long long seed, const int* __restrict__ frames,
                                                          int tap_stride, int g_stride) {
    __shared__ float2 tap[16 * 16];          // [S][n_taps], S <= 16
    __shared__ float2 gs[16 * 64];           // [S][L]
    const int fr = frames ? frames[blockIdx.x] : (int)blockIdx.x;
    const float step = 3.14159265358979323846f / (4.0f * kSinusoids);
    for (int e = threadIdx.x; e < S * n_taps; e += 64) {
        const int sym = e / n_taps, k = e % n_taps;
        const float t = (float)sym * t_sym, a0 = (float)(k + 1) * step;
        float sr = 0.f, si = 0.f;
        for (int n = 0; n < kSinusoids; ++n) {
            const float an = ((float)(n + 1) - 0.5f) * step;
            float thr, thi;
            if (theta_in) {
                thr = theta_in[(((size_t)fr * 2 + 0) * kSinusoids + n) * tap_stride + k];
                thi = theta_in[(((size_t)fr * 2 + 1) * kSinusoids + n) * tap_stride + k];
            } else {
                const unsigned long long base = ((unsigned long long)fr * 2 * kSinusoids + n) * tap_stride + k;
                thr = 6.2831853071795864769f *
                      uniform01(philox4x32_10(base, kStreamDoppler, offset, seed).v[0]);
                thi = 6.2831853071795864769f *
                      uniform01(philox4x32_10(base + (unsigned long long)kSinusoids * tap_stride, kStreamDoppler, offset, seed).v[0]);
            }
            sr += cosf(6.2831853071795864769f * t * (Fd * cosf(an + a0)) + thr);
            si += cosf(6.2831853071795864769f * t * (Fd * cosf(an - a0)) + thi);
        }
        const float c = coeff[k] * 0.14433756729740644113f;          // sqrt(1/48)
        tap[sym * n_taps + k] = make_float2(sr * c, si * c);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < S * L; e += 64) {
        const int sym = e / L, l = e % L;
        float2 a = make_float2(0.f, 0.f);
        for (int k = 0; k < n_taps; ++k) {
            const float w = alpha[k * L + l];
            a.x += tap[sym * n_taps + k].x * w;
            a.y += tap[sym * n_taps + k].y * w;
        }
        gs[e] = a;
        g[(size_t)fr * g_stride + sym * L + l] = a;
    }
    __syncthreads();
    if (H) {
        for (int e = threadIdx.x; e < S * nfft; e += 64) {
            const int sym = e / nfft, f = e % nfft;
            float2 a = make_float2(0.f, 0.f);
            for (int l = 0; l < L; ++l) {
                float sn, cs;
                sincosf(-6.2831853071795864769f * (float)((f * l) % nfft) / (float)nfft, &sn, &cs);
                const float2 v = gs[sym * L + l];
                a.x += v.x * cs - v.y * sn;
                a.y += v.x * sn + v.y * cs;
            }
            H[((size_t)fr * S + sym) * nfft + f] = a;
        }
    }
}

// radio.py:385-407 per-symbol impulse response with n_taps samples of history: symbol s filters the segment
// x[s*n_sc - n_taps .. (s+1)*n_sc) with np.convolve(., g_s, 'same') and keeps its last n_sc outputs, i.e.
// y[s*n_sc + t] = sum_l g_s[l] x[s*n_sc + t + off - l] over t + off - l in [-n_taps, n_sc) (nothing beyond the
// symbol's own end, nothing before the frame).  grid = (ceil(T/256), frames); + partial sums of |y|^2.
__global__ __launch_bounds__(256) void fir_doppler_kernel(const float2* __restrict__ x, const float2* __restrict__ g,
                                                          float2* __restrict__ y, double* __restrict__ partial, int T,
                                                          int L, int n_sc, int n_taps, const int* __restrict__ frames,
                                                          int g_stride, int n_frames, int pbase) {
    __shared__ double sh[4];
    const int bx = (T + 255) / 256;
    const int off = (L - 1) / 2;
    double pw = 0.0;
    const int items = n_frames * bx, ipb = (items + (int)gridDim.x - 1) / (int)gridDim.x;
    for (int item = blockIdx.x * ipb; item < min(items, ((int)blockIdx.x + 1) * ipb); ++item) {      // persistent: see fir_same_kernel
        const int fi = item / bx, cb = item - fi * bx;
        const int fr = frames ? frames[fi] : fi;
        const int t = cb * 256 + threadIdx.x;
        if (t < T) {
            const int sym = t / n_sc, tl = t - sym * n_sc;
            const float2* xf = x + (size_t)fr * T;
            const float2* gf = g + (size_t)fr * g_stride + sym * L;
            float2 a = make_float2(0.f, 0.f);
            for (int l = 0; l < L; ++l) {
                const int r = tl + off - l;                     // position relative to the symbol start
                const int u = sym * n_sc + r;
                if (r < -n_taps || r >= n_sc || u < 0) continue;
                const float2 v = xf[u], c = gf[l];
                a.x += c.x * v.x - c.y * v.y;
                a.y += c.x * v.y + c.y * v.x;
            }
            y[(size_t)fr * T + t] = a;
            pw += (double)a.x * a.x + (double)a.y * a.y;
        }
    }
    pw = wave_sum(pw);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = pw;
    __syncthreads();
    if (threadIdx.x == 0) partial[pbase + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

}  // namespace dccn
