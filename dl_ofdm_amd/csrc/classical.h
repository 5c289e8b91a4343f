// Classical pilot-aided receivers on the device (dev/m/OFDM_Benchmark_dev.m:339-456, the estimator
// family the "DCCN vs LS / LMMSE" curves are drawn against).  The contractions run on the library's fp32 MFMA GEMM
// (N-point DFT of the aligned FFT window as one real-expanded [2K, 2K] product; pilot interpolation as [2n, P] . [P, S K];
// the K x K smoothing matrices of the long-term LMMSE variants as another [2K, 2K] product); the kernels here are the
// per-cell stages around them, all one pass, HBM/latency bound:
//   classical_pilot_ls_kernel      g_p = Y[pilot] / pilot_value, written as two real planes for the interpolation GEMM
//   classical_gain_kernel          partial sums of the scalar that maps the channel's unit-gain response onto the
//                                  power-normalised frames (Perfect / ideal LMMSE) and of mean |G_ls|^2
//   classical_estimate_kernel      a frame per block: LS copy, ideal per-symbol LMMSE (rank-one Rhh), ALMMSE, true response,
//                                  frame average (input of the PDP variants)
//   classical_detect_kernel        x = Y / G at the data cells, nearest constellation point, bit errors vs the labels
// Host mirror: dl_ofdm_amd/benchmark_gpu.py; oracle: the NumPy restatement in dl_ofdm_amd/benchmark.py.
#pragma once
#include "common.h"

namespace dccn {

enum ClassicalEstimate : int { CE_LS = 0, CE_LMMSE = 1, CE_ALMMSE = 2, CE_PERFECT = 3, CE_FRAME_MEAN = 4 };
constexpr int kClassicalPartials = 512;

// Y [n, S*K, 2] -> gp [2][n][P] (plane 0 = real parts, plane 1 = imaginary parts) = Y[pilot] / pv
static __global__ __launch_bounds__(256) void classical_pilot_ls_kernel(const float2* __restrict__ Y, const int* __restrict__ pil,
                                                                 float* __restrict__ gp, int n, int SK, int P, float2 pv) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n * P) return;
    const int f = (int)(i / P), j = (int)(i % P);
    const float2 y = Y[(size_t)f * SK + pil[j]];
    const float d = pv.x * pv.x + pv.y * pv.y;
    gp[i] = (y.x * pv.x + y.y * pv.y) / d;
    gp[(size_t)n * P + i] = (y.y * pv.x - y.x * pv.y) / d;
}

// per-block partial sums (fixed order): [0] Re, [1] Im of sum Y_p conj(H_p pv), [2] sum |H_p pv|^2 over the pilots of the
// block's frames; [3] sum |G_ls|^2 over all cells.  H [n, S*K, 2] nullable (then [0..2] stay 0); Gls planes [2][n][SK].
static __global__ __launch_bounds__(256) void classical_gain_kernel(const float2* __restrict__ Y, const float2* __restrict__ H,
                                                             const float* __restrict__ Gls, const int* __restrict__ pil,
                                                             double* __restrict__ partial, int n, int SK, int P, float2 pv) {
    __shared__ double sh[4][4];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int f = blockIdx.x; f < n; f += gridDim.x) {
        if (H != nullptr) {
            for (int j = threadIdx.x; j < P; j += 256) {
                const float2 y = Y[(size_t)f * SK + pil[j]], h = H[(size_t)f * SK + pil[j]];
                const float2 hp = make_float2(h.x * pv.x - h.y * pv.y, h.x * pv.y + h.y * pv.x);
                a0 += (double)y.x * hp.x + (double)y.y * hp.y;
                a1 += (double)y.y * hp.x - (double)y.x * hp.y;
                a2 += (double)hp.x * hp.x + (double)hp.y * hp.y;
            }
        }
        for (int c = threadIdx.x; c < SK; c += 256) {
            const float re = Gls[(size_t)f * SK + c], im = Gls[(size_t)(n + f) * SK + c];
            a3 += (double)re * re + (double)im * im;
        }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; sh[w][0] = a0; sh[w][1] = a1; sh[w][2] = a2; sh[w][3] = a3; }
    __syncthreads();
    if (threadIdx.x < 4)
        partial[(size_t)blockIdx.x * 4 + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// every consumer adds the <= kClassicalPartials block sums itself, in one fixed order
__device__ __forceinline__ void classical_sum_partials(const double* __restrict__ partial, int nblk, double (&tot)[4],
                                                       double (*sh)[4]) {
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblk; b += 256)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += partial[(size_t)b * 4 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = wave_sum(a[k]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) sh[threadIdx.x >> 6][k] = a[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) tot[k] = (sh[0][k] + sh[1][k]) + (sh[2][k] + sh[3][k]);
}

// one frame per block.  G [n, S*K, 2] (mode CE_FRAME_MEAN: V [n, K, 2]).  c = LS error variance at a pilot.
static __global__ __launch_bounds__(256) void classical_estimate_kernel(const float* __restrict__ Gls, const float2* __restrict__ H,
                                                                 const double* __restrict__ partial, int nblk,
                                                                 float2* __restrict__ G, int n, int S, int K, int mode, float c) {
    __shared__ double sh[4][4];
    __shared__ float2 s_proj[32];        // per symbol (S <= 32)
    __shared__ float s_red[4][3];
    const int f = blockIdx.x, SK = S * K;
    const float* gre = Gls + (size_t)f * SK;
    const float* gim = Gls + (size_t)(n + f) * SK;
    float2 a = make_float2(1.f, 0.f);
    if (mode == CE_LMMSE || mode == CE_PERFECT) {
        double tot[4];
        classical_sum_partials(partial, nblk, tot, sh);
        a = make_float2((float)(tot[0] / tot[2]), (float)(tot[1] / tot[2]));       // G_true = H * a
    }
    if (mode == CE_LS) {
        for (int i = threadIdx.x; i < SK; i += 256) G[(size_t)f * SK + i] = make_float2(gre[i], gim[i]);
    } else if (mode == CE_PERFECT) {
        for (int i = threadIdx.x; i < SK; i += 256) {
            const float2 h = H[(size_t)f * SK + i];
            G[(size_t)f * SK + i] = make_float2(h.x * a.x - h.y * a.y, h.x * a.y + h.y * a.x);
        }
    } else if (mode == CE_LMMSE) {
        // per symbol: proj = <h, g_ls> / (|h|^2 + c), G = h proj     (rank-one Rhh = h h^H)
        for (int s = 0; s < S; ++s) {
            float pr = 0.f, pi = 0.f, pe = 0.f;
            for (int k = threadIdx.x; k < K; k += 256) {
                const float2 h0 = H[(size_t)f * SK + s * K + k];
                const float2 h = make_float2(h0.x * a.x - h0.y * a.y, h0.x * a.y + h0.y * a.x);
                const float gr = gre[s * K + k], gi = gim[s * K + k];
                pr += h.x * gr + h.y * gi;            // conj(h) * g
                pi += h.x * gi - h.y * gr;
                pe += h.x * h.x + h.y * h.y;
            }
            pr = wave_sum(pr); pi = wave_sum(pi); pe = wave_sum(pe);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6][0] = pr; s_red[threadIdx.x >> 6][1] = pi; s_red[threadIdx.x >> 6][2] = pe; }
            __syncthreads();
            if (threadIdx.x == 0) {
                const float r = (s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0]);
                const float i = (s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1]);
                const float e = (s_red[0][2] + s_red[1][2]) + (s_red[2][2] + s_red[3][2]);
                s_proj[s] = make_float2(r / (e + c), i / (e + c));
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < SK; i += 256) {
            const float2 h0 = H[(size_t)f * SK + i], p = s_proj[i / K];
            const float2 h = make_float2(h0.x * a.x - h0.y * a.y, h0.x * a.y + h0.y * a.x);
            G[(size_t)f * SK + i] = make_float2(h.x * p.x - h.y * p.y, h.x * p.y + h.y * p.x);
        }
    } else {
        // frame average v[k] = mean_s g_ls[s,k]; ALMMSE: G = v e / (e + c), e = sum |v|^2 / S
        float pe = 0.f;
        for (int k = threadIdx.x; k < K; k += 256) {
            float vr = 0.f, vi = 0.f;
            for (int s = 0; s < S; ++s) { vr += gre[s * K + k]; vi += gim[s * K + k]; }
            vr /= (float)S; vi /= (float)S;
            if (mode == CE_FRAME_MEAN) G[(size_t)f * K + k] = make_float2(vr, vi);
            pe += vr * vr + vi * vi;
        }
        if (mode == CE_ALMMSE) {
            pe = wave_sum(pe);
            if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][0] = pe;
            __syncthreads();
            const float e = ((s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0])) / (float)S;
            const float w = e / (e + c);
            for (int k = threadIdx.x; k < K; k += 256) {
                float vr = 0.f, vi = 0.f;
                for (int s = 0; s < S; ++s) { vr += gre[s * K + k]; vi += gim[s * K + k]; }
                vr = vr / (float)S * w; vi = vi / (float)S * w;
                for (int s = 0; s < S; ++s) G[(size_t)f * SK + s * K + k] = make_float2(vr, vi);
            }
        }
    }
}

// x = Y / G at the data cells (G row stride g_sk: S*K, or K with g_mod = K when the estimate is one row per frame),
// nearest point of table[m], detected bits (nullable) and bit errors against `bits` (per-block partial, fixed order)
static __global__ __launch_bounds__(256) void classical_detect_kernel(const float2* __restrict__ Y, const float2* __restrict__ G,
                                                               const int* __restrict__ dat, const float2* __restrict__ table,
                                                               const int* __restrict__ labels, const int32_t* __restrict__ bits,
                                                               int32_t* __restrict__ det, long long* __restrict__ err_partial,
                                                               int n, int SK, int D, int m, int nbits, int g_sk, int g_mod) {
    __shared__ long long sh[4];
    long long errs = 0;
    const long long total = (long long)n * D;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int f = (int)(i / D), j = (int)(i % D);
        const int cell = dat[j];
        const float2 y = Y[(size_t)f * SK + cell];
        const float2 g = G[(size_t)f * g_sk + (g_mod > 0 ? cell % g_mod : cell)];
        const float d = g.x * g.x + g.y * g.y;
        const float2 x = make_float2((y.x * g.x + y.y * g.y) / d, (y.y * g.x - y.x * g.y) / d);
        int best = 0;
        float bd = 3.0e38f;
        for (int q = 0; q < m; ++q) {                         // first minimum, like np.argmin
            const float dx = x.x - table[q].x, dy = x.y - table[q].y;
            const float dd = dx * dx + dy * dy;
            if (dd < bd) { bd = dd; best = q; }
        }
        for (int b = 0; b < nbits; ++b) {
            const int v = labels[best * nbits + b];
            if (det != nullptr) det[i * nbits + b] = v;
            errs += (v != bits[i * nbits + b]) ? 1 : 0;
        }
    }
    errs = wave_sum(errs);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = errs;
    __syncthreads();
    if (threadIdx.x == 0) err_partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// out[0] = sum of the detect partials (errors), out[1..4] = the gain partial totals as doubles reinterpreted by the host
static __global__ __launch_bounds__(256) void classical_finish_kernel(const long long* __restrict__ err_partial, int nerr,
                                                               const double* __restrict__ partial, int nblk,
                                                               long long* __restrict__ errors, double* __restrict__ sums) {
    __shared__ double sh[4][4];
    __shared__ long long she[4];
    if (err_partial != nullptr) {
        long long e = 0;
        for (int i = threadIdx.x; i < nerr; i += 256) e += err_partial[i];
        e = wave_sum(e);
        if ((threadIdx.x & 63) == 0) she[threadIdx.x >> 6] = e;
        __syncthreads();
        if (threadIdx.x == 0) errors[0] = (she[0] + she[1]) + (she[2] + she[3]);
    }
    if (sums != nullptr) {
        double tot[4];
        classical_sum_partials(partial, nblk, tot, sh);
        if (threadIdx.x == 0)
#pragma unroll
            for (int k = 0; k < 4; ++k) sums[k] = tot[k];
    }
}

}  // namespace dccn
