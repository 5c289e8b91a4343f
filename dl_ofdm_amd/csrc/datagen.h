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
#include "gemm_f32_mfma.h"
#include "norm_adam.h"

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
static __global__ __launch_bounds__(256) void philox_fill_kernel(unsigned* __restrict__ out, long long n, unsigned stream,
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
static __global__ __launch_bounds__(256) void tx_grid_kernel(const int* __restrict__ bits_in, int* __restrict__ bits_out,
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
static __global__ __launch_bounds__(64) void channel_taps_kernel(const float* __restrict__ taps_in,
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
static __global__ __launch_bounds__(256) void fir_same_kernel(const float2* __restrict__ x, const float2* __restrict__ g,
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
static __global__ __launch_bounds__(256) void awgn_kernel(const float2* __restrict__ y, const double* __restrict__ power_partial,
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

// ---- round 5: the whole static-channel generator chain of a batch in ONE launch ----------------------------------------
// What tx_grid_kernel -> IDFT GEMM -> (channel_taps_kernel) -> fir_same_kernel -> awgn_kernel do in four or five launches
// (26-37 us of a 110 us generate-and-train step, each launch latency-bound on 1170 frames), per block of two frames:
//   1. label bits (util.py:25-29, Philox stream 0) -> constellation / pilot / guard cells (ofdm.py:339-356) into an LDS grid
//      of 2 S rows x 2K floats (rows past 2 S zero);
//   2. ifft + cyclic prefix (ofdm.py:357-362) as a 16 x 2K x 2(K+CP) product on v_mfma_f32_16x16x4_f32 with the constant
//      matrix of DeviceDataGen.idft_cp_matrix read straight from L2 into the MFMA registers (every load of a wave is issued
//      before its first MFMA), the time-domain frames land in LDS;
//   3. the frames' static taps (radio.py:352-372, Philox stream 1; the arithmetic of channel_taps_kernel);
//   4. y = np.convolve(frame, g, 'same') (radio.py:360-366; the loop of fir_same_kernel) -> y, and ONE partial sum of |y|^2
//      per block;
//   5. the frame-scaled noise of radio.py:513-526 (Philox stream 2, sqrt(.5) 10^(-SNR/20) per frame; the draws and
//      expressions of awgn_kernel) -> noise, and one partial sum of |noise|^2 per block.
// What cannot be finished here is the batch-wide 1 / sqrt(mean |y|^2): x = y * that + noise is formed by the consumer -- R0 of
// the receiver step reads (y, noise, partials) as its virtual input (norm_adam.h NormVirtual), or gen_static_apply_kernel
// materialises x where a buffer is wanted.
constexpr int kGenMaxProfiles = 6;
struct GenProfile {                 // one static fading profile: tap amplitudes, sinc interpolation [n_taps, L]; identity: g = [1]
    const float* coeff; const float* alpha; int n_taps, L, identity, pad_;
};
struct GenStaticArgs {
    int32_t* bits_out; const int* cell_map; const float2* const_tab; float2 pilot; const float* idft;
    // frame f runs profile f % n_prof (radio.py:438-452 without Doppler frames; n_prof = 1: one channel for the batch);
    // tap_stride: taps per frame in the Philox index of the tap draws (channel_taps_kernel's)
    GenProfile prof[kGenMaxProfiles]; int n_prof, tap_stride;
    float2* H; int h_rep;           // nullable: fft(g, K) per frame, h_rep copies (channel_taps_kernel's arithmetic)
    const float* snr_db;
    float2* y; float2* noise; double* power_partial; double* noise_partial; float* tx_out;
    int frames, S, K, CP, D, nbits;
    unsigned offset; unsigned long long seed;
    int abl;            // timing ablations (DCCN_GEN_ABL, experiments only: results are wrong when set): 1 no noise draws, 2 no
                        // label draws, 4 no ifft matrix loads, 8 no FIR
    __device__ __forceinline__ GenStaticArgs at_chain(const long long coff) const {     // chain groups (common.h)
        GenStaticArgs q = *this;
        q.bits_out = chain_at(bits_out, coff); q.cell_map = chain_at(cell_map, coff); q.const_tab = chain_at(const_tab, coff);
        q.idft = chain_at(idft, coff);
        // (prof[] is NOT rebased here: the kernel reads the two profiles a workgroup needs from the kernel-argument segment by
        // index and rebases those -- an indexed COPY of the table lives in scratch memory, and the compiler turns a compare chain
        // over literal indices back into an index)
        q.H = chain_at(H, coff); q.snr_db = chain_at(snr_db, coff); q.y = chain_at(y, coff); q.noise = chain_at(noise, coff);
        q.power_partial = chain_at(power_partial, coff); q.noise_partial = chain_at(noise_partial, coff); q.tx_out = chain_at(tx_out, coff);
        return q;
    }
};
// what differs between the chains of a group besides their arenas: the modulation and the Philox stream (n == 0: one chain,
// the values of GenStaticArgs stand)
struct GenChainScalars {
    int n;
    int nbits[kMaxChains];
    unsigned offset[kMaxChains];
    unsigned long long seed[kMaxChains];
};
typedef float gen_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kGenFramesPerBlock = 2;
constexpr int kGenFirPad = 64;             // >= the longest channel response the launch accepts (L <= 64)
// S, K, CP are compile-time (the N = 64 grid of the reference: 7 symbols, 64 + 16 samples): every index split is a division by a
// constant, and the per-thread loops over a block's 1024 grid cells and 1120 samples are unrolled -- a thread's cells / samples
// are INDEPENDENT chains (Philox -> Box-Muller -> store; cell map -> Philox -> constellation table) whose latencies then
// overlap instead of adding up (the first version walked them one after the other with run-time divisions: 26.8 us per launch;
// profiles/r05_e2e_kernel_stats.txt has this one)
// The body is a device function so that the generator's workgroups can also RIDE on another launch (round 6: behind the
// equaliser step's bottleneck backward launch, eq_bottleneck.h): a = the launch's argument block as gen_args_of_chain
// prepared it for this chain, block = this workgroup's index among the generator's; the body uses gen_static_smem_bytes() of
// dynamic LDS.
// The launch's argument block for one chain (arena offset applied, per-chain scalars taken), written out where the KERNEL
// PARAMETERS are in scope: handing them to a helper by reference takes their address, which turns the by-value kernel
// arguments into a private copy -- and the per-chain index into a scratch access.
#define DCCN_GEN_ARGS_OF_CHAIN(a, P0, P1, a0, gc, coff, chain, block)                                         \
    GenStaticArgs a = (a0).at_chain(coff);                                                                     \
    if ((gc).n > 0) { a.nbits = (gc).nbits[chain]; a.offset = (gc).offset[chain]; a.seed = (gc).seed[chain]; } \
    /* the profiles of the block's two frames (block-uniform; frame 1 of a one-frame block repeats frame 0) */  \
    GenProfile P0 = (a0).prof[((block) * kGenFramesPerBlock) % (a0).n_prof];                                    \
    GenProfile P1 = (a0).prof[((block) * kGenFramesPerBlock + (((a0).frames - (block) * kGenFramesPerBlock) > 1 ? 1 : 0)) % (a0).n_prof]; \
    P0.coeff = chain_at(P0.coeff, coff); P0.alpha = chain_at(P0.alpha, coff);                                   \
    P1.coeff = chain_at(P1.coeff, coff); P1.alpha = chain_at(P1.alpha, coff);
template <int S, int K, int CP>
__device__ __forceinline__ void gen_static_frames_body(const GenStaticArgs a, const GenProfile P0, const GenProfile P1, const int block) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    constexpr int K2 = 2 * K, N2 = 2 * (K + CP), T = S * (K + CP), LDG = K2 + 4;
    // the cyclic-prefix columns of the ifft matrix are bitwise copies of its last 2 CP columns (t = (t' - CP) mod K,
    // datagen.py idft_cp_matrix): only the K2 / 16 tiles behind the prefix are multiplied, the prefix is stored twice
    constexpr int G16 = K2 / 16, NTILE = N2 / 16, CPT = 2 * CP / 16, NMUL = NTILE - CPT, TPW = (NMUL + 3) / 4;
    static_assert((2 * CP) % 16 == 0 && CPT <= NMUL, "cyclic prefix: whole 16-column tiles");
    constexpr int NSMP = (kGenFramesPerBlock * T + 255) / 256;          // samples per thread (5)
    constexpr int NCELL = 16 * K / 256;                                  // grid cells per thread (4)
    static_assert(kGenFramesPerBlock * S <= 16 && K2 % 16 == 0 && N2 % 16 == 0 && (16 * K) % 256 == 0, "generator tile shape");
    // the time-domain frames sit between two runs of kGenFirPad zeros: the 'same' FIR then reads its out-of-range neighbours
    // as zeros instead of branching around them (adding +-0 leaves every partial sum as it was: same bits as the skipped form)
    constexpr int TP = T + 2 * kGenFirPad;
    float* sG = gsm;                                   // [16][LDG]   grid rows (frame, symbol), (k, iq) contiguous
    float2* sTX = reinterpret_cast<float2*>(gsm + 16 * LDG);      // [2][pad | T | pad]  time-domain frames
    __shared__ float2 gs[kGenFramesPerBlock][64];
    __shared__ float2 tap[kGenFramesPerBlock][16];
    __shared__ float2 tw[K];                           // (cos, sin) of -2 pi m / K: the frequency response's twiddles
    __shared__ double sh[2][4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int f0 = block * kGenFramesPerBlock;
    const int nfr = min(kGenFramesPerBlock, a.frames - f0);
    const int L0 = P0.identity ? 1 : P0.L, L1 = P1.identity ? 1 : P1.L;
    if (a.H != nullptr && tid < K) {
        float sn, cs;
        sincosf(-6.2831853071795864769f * (float)tid / (float)K, &sn, &cs);
        tw[tid] = make_float2(cs, sn);
    }
    static_assert(kGenFramesPerBlock * 2 * kGenFirPad == 256, "one pad cell per thread");
    sTX[(tid >> 7) * TP + ((tid >> 6) & 1) * (kGenFirPad + T) + (tid & 63)] = make_float2(0.f, 0.f);

    // 2 (early). this wave's share of the ifft matrix: tiles w, w + 4, ... ; lane (c, kq) needs idft[16 g + 4 kq + j][16 tile + c]
    const int c = lane & 15, kq = lane >> 4;
    float bfr[TPW][G16][4];
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int tile = min(CPT + w + 4 * ti, NTILE - 1);
        const float* B = a.idft + (size_t)(4 * kq) * N2 + 16 * tile + c;
#pragma unroll
        for (int g = 0; g < G16; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[ti][g][j] = (a.abl & 4) ? 0.5f : B[(size_t)(16 * g + j) * N2];
    }
    // 1 (early). cell map entries of this thread's grid cells
    int cd[NCELL];
#pragma unroll
    for (int q = 0; q < NCELL; ++q) {
        const int i = tid + 256 * q, row = i / K, k = i - row * K, fr = row / S;
        cd[q] = fr < nfr ? a.cell_map[(row - fr * S) * K + k] : -2;
    }
    // 5. the frame-scaled noise (depends on nothing else: its Philox / Box-Muller chains run under the loads above)
    double nw = 0.0;
    {
        const float std0 = 0.70710678118654752440f * exp10f(-a.snr_db[f0] * 0.05f);
        const float std1 = 0.70710678118654752440f * exp10f(-a.snr_db[f0 + (nfr > 1 ? 1 : 0)] * 0.05f);
        float2 z[NSMP];
#pragma unroll
        for (int q = 0; q < NSMP; ++q) {
            const int i = tid + 256 * q, fr = i / T;
            const size_t o = (size_t)f0 * T + i;                         // (f0 + fr) * T + (i - fr * T)
            if (a.abl & 1) { z[q] = make_float2(0.3f, 0.1f); continue; }
            const Philox4 p = philox4x32_10((unsigned long long)o, kStreamNoise, a.offset, a.seed);
            z[q] = box_muller(p.v[0], p.v[1]);
            const float sd = fr == 0 ? std0 : std1;
            z[q].x *= sd;
            z[q].y *= sd;
        }
#pragma unroll
        for (int q = 0; q < NSMP; ++q) {
            const int i = tid + 256 * q;
            if (i < nfr * T) {
                out_store2<5>(reinterpret_cast<float*>(a.noise + (size_t)f0 * T + i), z[q].x, z[q].y);
                nw += (double)z[q].x * z[q].x + (double)z[q].y * z[q].y;
            }
        }
    }
    // 3 (early). static taps of the block's frames (threads 0..n_taps-1 of waves 0 / 1 draw frame 0 / 1)
    const GenProfile Pw = w == 0 ? P0 : P1;            // (waves 0 / 1 own the taps of frames 0 / 1)
    if (w < nfr && !Pw.identity && lane < Pw.n_taps) {
        const Philox4 p = philox4x32_10((unsigned long long)(f0 + w) * a.tap_stride + lane, kStreamTaps, a.offset, a.seed);
        const float2 z = box_muller(p.v[0], p.v[1]);
        const float cf = Pw.coeff[lane] * 0.70710678118654752440f;
        tap[w][lane] = make_float2(z.x * cf, z.y * cf);
    }
    // 1. resource grid: label bits (Philox word 0 of the cell), constellation / pilot / guard value
    {
        int idx[NCELL];
#pragma unroll
        for (int q = 0; q < NCELL; ++q) {
            const int i = tid + 256 * q, row = i / K, fr = row / S;
            idx[q] = 0;
            if (cd[q] >= 0) {
                const long long cell = (long long)(f0 + fr) * a.D + cd[q];
                const unsigned word = (a.abl & 2) ? (unsigned)cell : philox4x32_10((unsigned long long)cell, kStreamBits, a.offset, a.seed).v[0];
                for (int j = 0; j < a.nbits; ++j) {
                    const int bit = (int)((word >> j) & 1u);
                    out_store<5>(a.bits_out + cell * a.nbits + j, bit);
                    idx[q] = (idx[q] << 1) | bit;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NCELL; ++q) {
            const int i = tid + 256 * q, row = i / K, k = i - row * K;
            float2 v = make_float2(0.f, 0.f);
            if (cd[q] == -1) v = a.pilot;
            else if (cd[q] >= 0) v = a.const_tab[idx[q]];
            *reinterpret_cast<float2*>(sG + row * LDG + 2 * k) = v;
        }
    }
    __syncthreads();
    if (w < kGenFramesPerBlock) {                      // g = taps . alpha (same order as channel_taps_kernel); zeros behind L:
        float2 acc = make_float2(0.f, 0.f);            // the FIR below runs both frames over the longer of the two responses
        const int Lw = w == 0 ? L0 : L1;
        if (w < nfr && lane < Lw) {
            if (Pw.identity) {
                acc = make_float2(lane == 0 ? 1.f : 0.f, 0.f);
            } else {
                for (int k = 0; k < Pw.n_taps; ++k) {
                    const float wgt = Pw.alpha[k * Pw.L + lane];
                    acc.x += tap[w][k].x * wgt;
                    acc.y += tap[w][k].y * wgt;
                }
            }
        }
        gs[w][lane] = acc;
    }
    // 2. tx[row][n] = sum_k grid[row][k] idft[k][n]
    {
        gen_f32x4 acc[TPW];
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) acc[ti] = gen_f32x4{0.f, 0.f, 0.f, 0.f};
        const float* Ar = sG + c * LDG + 4 * kq;
#pragma unroll
        for (int g = 0; g < G16; ++g) {
            const float4 av = *reinterpret_cast<const float4*>(Ar + 16 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ti = 0; ti < TPW; ++ti)
                    acc[ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(av, j), bfr[ti][g][j], acc[ti], 0, 0, 0);
        }
        // C layout: col = lane & 15, row = 4 (lane >> 4) + r
        float* txf = reinterpret_cast<float*>(sTX);
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
            const int tile = CPT + w + 4 * ti;
            if (tile < NTILE) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * kq + r, fr = row / S;
                    if (fr < nfr) {
                        float* o = txf + fr * 2 * TP + 2 * kGenFirPad + (row - fr * S) * N2 + 16 * tile + c;
                        o[0] = acc[ti][r];
                        if (tile >= NTILE - CPT) o[-K2] = acc[ti][r];              // the prefix: columns 2K.. are columns 0..
                    }
                }
            }
        }
    }
    __syncthreads();
    if (a.tx_out != nullptr)
        for (int i = tid; i < nfr * 2 * T; i += 256) {
            const int fr = i / (2 * T);
            a.tx_out[(size_t)f0 * 2 * T + i] = reinterpret_cast<const float*>(sTX)[fr * 2 * TP + 2 * kGenFirPad + (i - fr * 2 * T)];
        }
    // 4. 'same' FIR (the loop of fir_same_kernel) and its power
    // frequency response of the block's frames (channel_taps_kernel's sum, its twiddles from the table: same arguments, same bits)
    if (a.H != nullptr) {
        for (int idx = tid; idx < nfr * K; idx += 256) {
            const int fr = idx / K, f = idx - fr * K, Lf = fr == 0 ? L0 : L1;
            float2 acc = make_float2(0.f, 0.f);
            for (int l = 0; l < Lf; ++l) {
                const float2 t2 = tw[(f * l) % K], gl = gs[fr][l];
                acc.x += gl.x * t2.x - gl.y * t2.y;
                acc.y += gl.x * t2.y + gl.y * t2.x;
            }
            for (int r = 0; r < a.h_rep; ++r) a.H[((size_t)(f0 + fr) * a.h_rep + r) * K + f] = acc;
        }
    }
    const int off0 = (L0 - 1) / 2, off1 = (L1 - 1) / 2, Lmax = max(L0, nfr > 1 ? L1 : 0);
    double pw = 0.0;
    {
        float2 yv[NSMP];
#pragma unroll
        for (int q = 0; q < NSMP; ++q) {
            const int i = min(tid + 256 * q, nfr * T - 1), fr = i / T, t = i - fr * T;
            const float2* xf = sTX + fr * TP + kGenFirPad + t + (fr == 0 ? off0 : off1);
            float2 acc = make_float2(0.f, 0.f);
            for (int l = 0; l < ((a.abl & 8) ? 1 : Lmax); ++l) {
                const float2 v = xf[-l], gl = gs[fr][l];
                acc.x += gl.x * v.x - gl.y * v.y;
                acc.y += gl.x * v.y + gl.y * v.x;
            }
            yv[q] = acc;
        }
#pragma unroll
        for (int q = 0; q < NSMP; ++q) {
            const int i = tid + 256 * q;
            if (i < nfr * T) {
                out_store2<5>(reinterpret_cast<float*>(a.y + (size_t)f0 * T + i), yv[q].x, yv[q].y);
                pw += (double)yv[q].x * yv[q].x + (double)yv[q].y * yv[q].y;
            }
        }
    }
    pw = wave_sum(pw);
    nw = wave_sum(nw);
    if (lane == 0) { sh[0][w] = pw; sh[1][w] = nw; }
    __syncthreads();
    if (tid == 0) {
        a.power_partial[block] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        if (a.noise_partial != nullptr) a.noise_partial[block] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}
template <int S, int K, int CP>
constexpr size_t gen_static_smem_bytes() {
    return (size_t)16 * (2 * K + 4) * sizeof(float) + (size_t)kGenFramesPerBlock * (S * (K + CP) + 2 * kGenFirPad) * sizeof(float2);
}
template <int S, int K, int CP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void gen_static_frames_kernel(const GenStaticArgs a0, const GenChainScalars gc,
                                                                                                           const ChainOffs co) {
    DCCN_GEN_ARGS_OF_CHAIN(a, P0, P1, a0, gc, co.off[blockIdx.z], blockIdx.z, (int)blockIdx.x)
    gen_static_frames_body<S, K, CP>(a, P0, P1, (int)blockIdx.x);
}
// x = y / sqrt(mean |y|^2) + noise where a buffer is wanted (the first batch of a pipelined loop, tests, iq dumps): the
// expression of awgn_kernel on the generator's y and noise.  grid: any; 256 threads.
static __global__ __launch_bounds__(256) void gen_static_apply_kernel(const float4* y_, const float4* noise_, const double* ppart_, int npart,
                                                               double total, float4* x_, long long n4, const double* npart_noise_,
                                                               int n_noise, float* npow_out_, const ChainOffs co) {
    const long long coff = co.off[blockIdx.z];                     // chain groups (common.h)
    const float4* __restrict__ y = chain_at(y_, coff);
    const float4* __restrict__ noise = chain_at(noise_, coff);
    const double* __restrict__ ppart = chain_at(ppart_, coff);
    float4* __restrict__ x = chain_at(x_, coff);
    const double* __restrict__ npart_noise = chain_at(npart_noise_, coff);
    float* __restrict__ npow_out = chain_at(npow_out_, coff);
    __shared__ double sh4[4];
    __shared__ float s_inv;
    const float inv = batch_power_inv_scale(ppart, npart, total, sh4, &s_inv);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = y[i], z = noise[i];
        x[i] = make_float4(v.x * inv + z.x, v.y * inv + z.y, v.z * inv + z.z, v.w * inv + z.w);
    }
    if (npow_out != nullptr && blockIdx.x == 0) {
        __syncthreads();
        double a = 0.0;
        for (int i = threadIdx.x; i < n_noise; i += 256) a += npart_noise[i];
        a = wave_sum(a);
        if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) npow_out[0] = (float)(((sh4[0] + sh4[1]) + (sh4[2] + sh4[3])) / total);
    }
}

// radio.py:62-88 AWGN_channel, the *in-graph* monitor branch of the receiver graph (ofdmreceiver_np.py:136,151-152):
// xn = batch-normalised (eps 1e-8) clipped signal / sqrt(2) (computed by the caller with the R0 kernel),
// level = sqrt(.5) 10^(-SNR/20) per frame, amp = level * N(0,1), phase = U(0, 2 pi),
// noise = (|amp| sin(phase), |amp| cos(phase)); iq_rx = fp16(xn + noise), iq_tx = fp16(clipped signal),
// noise_power = mean(noise_re^2 + noise_im^2).  The receiver never consumes this branch (`rx_iq_data = iq_tx_re`);
// it only feeds the constellation dumps and the printed noise power.  grid = (ceil(T/256), frames).
static __global__ __launch_bounds__(256) void ingraph_awgn_kernel(const float2* __restrict__ clipped,
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
static __global__ __launch_bounds__(64) void doppler_taps_kernel(const float* __restrict__ theta_in,
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
static __global__ __launch_bounds__(256) void fir_doppler_kernel(const float2* __restrict__ x, const float2* __restrict__ g,
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
