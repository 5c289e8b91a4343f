// HBM-bound pieces of the receiver step: R0 batch-moment normalisation (+ fused R8 clip
// power), R7 TF-Adam over a flat parameter arena.  All are float4-vectorised,
// deterministic (two-stage reductions in a fixed order, fp64 moment accumulation).
#pragma once
#include "common.h"

namespace dccn {

constexpr int kNormRowChunks = 32;      // partial-moment slabs per column
constexpr int kNormRowsPerBlock = 32;   // rows handled by one normalise block
// single-pass kernel: column groups per block / rows per thread (overridable for experiments; 128*CG threads per block,
// the optimizer kernel's ride needs CG = 2)
#ifndef DCCN_NORM_CG
#define DCCN_NORM_CG 2
#endif
#ifndef DCCN_NORM_RPT
#define DCCN_NORM_RPT 12
#endif
constexpr int kNormFusedCG = DCCN_NORM_CG, kNormFusedRPT = DCCN_NORM_RPT;

__device__ __forceinline__ float adam_alpha(const dccn_adam_state* st, const dccn_adam_hparams& hp) {
    const float lr = hp.lr0 * powf(hp.decay_rate, floorf(st->global_step / hp.decay_steps));
    return lr * sqrtf(1.0f - st->beta2_power) / (1.0f - st->beta1_power);
}

// ---- R0 stage 1: per-column partial sum / sum of squares in fp64 ------------------------
// grid (ceil(cols/4/64), kNormRowChunks), block (64,4).  partial[(chunk*cols + c)*2 + {0,1}]
// In the fused training step the first thread of the first block also does the optimizer's
// per-step bookkeeping (it runs long before the Adam kernel of this step and after the one of the
// previous step): alpha for THIS step from the current global_step / beta powers, then advance them.
static __global__ __launch_bounds__(256) void moments_kernel(const float* __restrict__ x, int batch, int cols,
                                                      double* __restrict__ partial,
                                                      dccn_adam_state* __restrict__ adam, dccn_adam_hparams hp) {
    __shared__ double red[4][64][8];
    const int tx = threadIdx.x, ty = threadIdx.y;
    if (adam != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tx == 0 && ty == 0) {
        adam->alpha = adam_alpha(adam, hp);
        adam->beta1_power = adam->beta1_power * hp.beta1;
        adam->beta2_power = adam->beta2_power * hp.beta2;
        adam->global_step = adam->global_step + 1.0f;
    }
    const int c4 = (blockIdx.x * 64 + tx) * 4;
    const int rows_per_chunk = (batch + kNormRowChunks - 1) / kNormRowChunks;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(batch, r0 + rows_per_chunk);
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (c4 < cols) {
        const bool vec = (c4 + 3 < cols) && ((cols & 3) == 0);
        if (vec) {
            // batches of 4 rows: the four 16-byte loads are issued together (clamped row + weight 0/1
            // instead of a branch), then accumulated in row order
            for (int rb = r0 + ty; rb < r1; rb += 16) {
                float4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    t[u] = *reinterpret_cast<const float4*>(x + (size_t)min(rb + 4 * u, batch - 1) * cols + c4);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (rb + 4 * u < r1) {
                        const double d0 = t[u].x, d1 = t[u].y, d2 = t[u].z, d3 = t[u].w;
                        s[0] += d0; q[0] += d0 * d0;
                        s[1] += d1; q[1] += d1 * d1;
                        s[2] += d2; q[2] += d2 * d2;
                        s[3] += d3; q[3] += d3 * d3;
                    }
                }
            }
        } else {
            for (int r = r0 + ty; r < r1; r += 4) {
                const float* p = x + (size_t)r * cols + c4;
                for (int e = 0; e < 4; ++e) {
                    if (c4 + e < cols) {
                        const double d = (double)p[e];
                        s[e] += d;
                        q[e] += d * d;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[ty][tx][e] = s[e];
        red[ty][tx][4 + e] = q[e];
    }
    __syncthreads();
    if (ty == 0 && c4 < cols) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (c4 + e < cols) {
                const double ss = (red[0][tx][e] + red[1][tx][e]) + (red[2][tx][e] + red[3][tx][e]);
                const double qq = (red[0][tx][4 + e] + red[1][tx][4 + e]) + (red[2][tx][4 + e] + red[3][tx][4 + e]);
                double* o = partial + ((size_t)blockIdx.y * cols + c4 + e) * 2;
                o[0] = ss;
                o[1] = qq;
            }
        }
    }
}

// ---- R0 stage 2+3: combine the slabs (each block for its own 256 columns: mean, inv = rsqrt(var+eps),
// shift = -mean*inv), then y = (x*inv + shift)/sqrt(2)   (+ R8: per-block partial sums of the clipped power)
// grid (ceil(cols/4/64), ceil(batch/kNormRowsPerBlock)), block (64,4)
// power_partial[blockIdx.y*gridDim.x + blockIdx.x] = sum over the block of |clip(y)|^2 (fp64)
static __global__ __launch_bounds__(256) void normalise_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const double* __restrict__ partial, int batch, int cols,
                                                        float eps, float peak, double* __restrict__ power_partial,
                                                        float* __restrict__ mean_out, float* __restrict__ var_out) {
    __shared__ float2 sstat[256];
    __shared__ double red[4];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int lin = ty * 64 + tx;
    {
        const int c = blockIdx.x * 256 + lin;
        float2 st = make_float2(0.f, 0.f);
        if (c < cols) {
            // all chunk loads are issued before the first add (the sums are then formed in a fixed order)
            double2 pv[kNormRowChunks];
#pragma unroll
            for (int z = 0; z < kNormRowChunks; ++z)
                pv[z] = *reinterpret_cast<const double2*>(partial + ((size_t)z * cols + c) * 2);
            __builtin_amdgcn_sched_barrier(0);
            double sm = 0.0, sq = 0.0;
#pragma unroll
            for (int z = 0; z < kNormRowChunks; ++z) {
                sm += pv[z].x;
                sq += pv[z].y;
            }
            const double mean = sm / (double)batch;
            double var = sq / (double)batch - mean * mean;    // biased variance; fp64, so no cancellation issue
            if (var < 0.0) var = 0.0;
            const float meanf = (float)mean, varf = (float)var;
            const float inv = 1.0f / sqrtf(varf + eps);
            st = make_float2(inv, -meanf * inv);
            if (blockIdx.y == 0) {
                if (mean_out) mean_out[c] = meanf;
                if (var_out) var_out[c] = varf;
            }
        }
        sstat[lin] = st;
    }
    __syncthreads();
    const int c4 = (blockIdx.x * 64 + tx) * 4;
    const int r0 = blockIdx.y * kNormRowsPerBlock;
    const int r1 = min(batch, r0 + kNormRowsPerBlock);
    const float rs2 = 1.41421356237309515f;            // float32(np.sqrt(2))
    double pw = 0.0;
    if (c4 < cols) {
        const bool vec = (c4 + 3 < cols) && ((cols & 3) == 0);
        float inv[4], sh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float2 st = sstat[tx * 4 + e];
            inv[e] = st.x;
            sh[e] = st.y;
        }
#pragma unroll 8
        for (int r = r0 + ty; r < r1; r += 4) {
            const size_t off = (size_t)r * cols + c4;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (vec) {
                const float4 t = *reinterpret_cast<const float4*>(x + off);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
                for (int e = 0; e < 4; ++e)
                    if (c4 + e < cols) v[e] = x[off + e];
            }
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[e] * inv[e] + sh[e]) / rs2;
            if (vec) {
                *reinterpret_cast<float4*>(y + off) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                for (int e = 0; e < 4; ++e)
                    if (c4 + e < cols) y[off + e] = o[e];
            }
            if (power_partial != nullptr) {
                // complex_clip (dev/py/complex.py:24-26): pairs (I,Q) = (o0,o1), (o2,o3); cols is even
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    if (c4 + e + 1 < cols) {
                        const float l2 = sqrtf(o[e] * o[e] + o[e + 1] * o[e + 1]);
                        const float sc = peak / fmaxf(l2, peak);
                        const float ci = o[e] * sc, cq = o[e + 1] * sc;
                        pw += (double)(ci * ci + cq * cq);
                    }
                }
            }
        }
    }
    if (power_partial != nullptr) {
        pw = wave_sum(pw);
        if ((lin & 63) == 0) red[lin >> 6] = pw;
        __syncthreads();
        if (lin == 0) power_partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// ---- R0 in ONE pass for batches that fit in registers -----------------------------------------------
// A block owns CG float4 column groups (4*CG columns) for ALL rows: thread t = (row slot t/CG, group t%CG)
// keeps its rows slot, slot+128, ... (at most RPT of them) in registers, so x is read once and the two-pass
// moments (mean, then mean of squared differences -- what tf.nn.moments computes) cost no second trip to HBM.
// Column sums: xor-shuffles over the row slots of a wave, then one LDS stage over the waves (fixed order).
// grid = ceil(cols / (4*CG)) rounded up to a multiple of 8, block = 128*CG threads; needs cols % 4 == 0 and
// batch <= 128*RPT.
// power_partial[blockIdx.x] = sum over the block of |clip(y)|^2; adam: see moments_kernel.
// (a __device__ body: the kernel below runs it, and so do the leading blocks of the optimizer kernel when the step
// normalises the NEXT batch behind its Adam update -- bidx / nblk are the block's index and count inside that group)
// Virtual input (round 5, the fused device generator of datagen.h): the batch to normalise is not in memory as x but as the
// two terms of dev/py/radio.py:513-526 AWGN_channel_np, x = y / sqrt(mean |y|^2 over the batch) + noise -- y the channel
// output, noise already scaled per frame, the batch power as <= 2048 per-block partial sums of the generator launch.  R0 forms
// x in registers on the way in (the expression and the order of awgn_kernel: the same bits as the materialised input), so
// the AWGN launch and a 5 MB round trip of x disappear.  y == nullptr: plain input.
struct NormVirtual {
    const float* y; const float* noise;           // [batch, cols] each
    const double* ppart; int npart; double total; // partial sums of |y|^2, batch * T (complex samples)
    float* x_out;                                 // nullable: also store x (tx_ofdm) -- tests, iq dumps
    const double* npart_noise; int n_noise; float* npow_out;      // nullable: `noise_power:0` monitor, summed by block 0
    __device__ __forceinline__ NormVirtual at_chain(const long long coff) const {        // chain groups (common.h)
        NormVirtual q = *this;
        q.y = chain_at(y, coff); q.noise = chain_at(noise, coff); q.ppart = chain_at(ppart, coff); q.x_out = chain_at(x_out, coff);
        q.npart_noise = chain_at(npart_noise, coff); q.npow_out = chain_at(npow_out, coff);
        return q;
    }
};
__device__ __host__ inline NormVirtual norm_virtual_none() {
    NormVirtual v;
    v.y = nullptr; v.noise = nullptr; v.ppart = nullptr; v.npart = 0; v.total = 1.0; v.x_out = nullptr;
    v.npart_noise = nullptr; v.n_noise = 0; v.npow_out = nullptr;
    return v;
}
// 1 / sqrt(mean |y|^2): every block of a 256-thread launch adds the partial sums in the same fixed order (thread t: t, t + 256,
// ...; DPP wave sum; the four wave sums as (0 + 1) + (2 + 3)), so all blocks -- and awgn_kernel, and R0 -- get the same bits
__device__ __forceinline__ float batch_power_inv_scale(const double* __restrict__ ppart, const int npart, const double total,
                                                       double* sh4, float* s_inv) {
    double a = 0.0;
    for (int i = threadIdx.x; i < npart; i += 256) a += ppart[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) *s_inv = 1.0f / sqrtf((float)(((sh4[0] + sh4[1]) + (sh4[2] + sh4[3])) / total));
    __syncthreads();
    return *s_inv;
}

template <int CG, int RPT>
__device__ __forceinline__ void norm_fused_body(const float* __restrict__ x, float* __restrict__ y, int batch, int cols,
                                                float eps, float peak, double* __restrict__ power_partial,
                                                float* __restrict__ mean_out, float* __restrict__ var_out,
                                                dccn_adam_state* __restrict__ adam, dccn_adam_hparams hp,
                                                const int bidx, const int nblk, const NormVirtual nv = norm_virtual_none()) {
    constexpr int NW = 2 * CG;                        // waves per block
    constexpr int RS = 64 / CG;                       // row slots per wave
    __shared__ double red[NW][CG][8];
    __shared__ double stat[CG][8];
    __shared__ double pred[NW > 4 ? NW : 4];
    __shared__ float s_inv;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool virt = NW == 4 && nv.y != nullptr;     // (kernel argument: block-uniform; 256-thread blocks only)
    float inv_scale = 1.0f;
    if (virt) inv_scale = batch_power_inv_scale(nv.ppart, nv.npart, nv.total, pred, &s_inv);
    if (adam != nullptr && bidx == 0 && t == 0) {
        adam->alpha = adam_alpha(adam, hp);
        adam->beta1_power = adam->beta1_power * hp.beta1;
        adam->beta2_power = adam->beta2_power * hp.beta2;
        adam->global_step = adam->global_step + 1.0f;
    }
    // XCD-aware strip order: workgroups go round-robin over the 8 XCDs (each with its own L2), and strips that
    // share 128-byte lines of x are neighbours -- so XCD k takes the k-th run of consecutive strips
    const int per_xcd = nblk / 8;
    const int strip = (bidx % 8) * per_xcd + bidx / 8;
    const int cg = t % CG, slot = t / CG;
    const int c4 = (strip * CG + cg) * 4;
    const bool live = c4 < cols;
    float4 v[RPT];
    const int c4c = live ? c4 : 0;                    // clamped addresses: every load issues, none branches
    if (virt) {
        float4 nz[RPT];
#pragma unroll
        for (int p = 0; p < RPT; ++p) {
            const size_t o = (size_t)min(slot + 128 * p, batch - 1) * cols + c4c;
            v[p] = *reinterpret_cast<const float4*>(nv.y + o);
            nz[p] = *reinterpret_cast<const float4*>(nv.noise + o);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < RPT; ++p) {               // radio.py:524-526, as awgn_kernel evaluates it
            v[p] = make_float4(v[p].x * inv_scale + nz[p].x, v[p].y * inv_scale + nz[p].y, v[p].z * inv_scale + nz[p].z,
                               v[p].w * inv_scale + nz[p].w);
            if (nv.x_out != nullptr && live && slot + 128 * p < batch)
                *reinterpret_cast<float4*>(nv.x_out + (size_t)(slot + 128 * p) * cols + c4) = v[p];
        }
    } else {
#pragma unroll
        for (int p = 0; p < RPT; ++p)
            v[p] = *reinterpret_cast<const float4*>(x + (size_t)min(slot + 128 * p, batch - 1) * cols + c4c);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < RPT; ++p)
        if (!live || slot + 128 * p >= batch) v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    // per-thread sum and sum of squares in fp64 (var = E[x^2] - mean^2 is safe in double), then ONE reduction
    // over the block's rows: xor-shuffles over the row slots of a wave, LDS over the waves, fixed order
    double sq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < RPT; ++p) {
        const double d0 = v[p].x, d1 = v[p].y, d2 = v[p].z, d3 = v[p].w;      // rows past the batch hold zeros
        sq[0] += d0; sq[1] += d1; sq[2] += d2; sq[3] += d3;
        sq[4] += d0 * d0; sq[5] += d1 * d1; sq[6] += d2 * d2; sq[7] += d3 * d3;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int m = CG; m < 64; m <<= 1) sq[e] += __shfl_xor(sq[e], m, 64);
    if (lane < CG) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave][cg][e] = sq[e];
    }
    __syncthreads();
    if (t < 8 * CG) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) a += red[w][t >> 3][t & 7];
        stat[t >> 3][t & 7] = a;
    }
    __syncthreads();
    double mean[4], q[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        mean[e] = stat[cg][e] / (double)batch;
        double var = stat[cg][4 + e] / (double)batch - mean[e] * mean[e];
        q[e] = (var < 0.0 ? 0.0 : var) * (double)batch;
    }
    (void)RS;
    float inv[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float meanf = (float)mean[e], varf = (float)(q[e] / (double)batch);
        inv[e] = 1.0f / sqrtf(varf + eps);
        sh[e] = -meanf * inv[e];
        if (slot == 0 && live) {
            if (mean_out) mean_out[c4 + e] = meanf;
            if (var_out) var_out[c4 + e] = varf;
        }
    }
    const float rs2 = 1.41421356237309515f;            // float32(np.sqrt(2))
    double pw = 0.0;
#pragma unroll
    for (int p = 0; p < RPT; ++p) {
        const int r = slot + 128 * p;
        if (live && r < batch) {
            // t / sqrt(2) as q = t*r, then one FMA residual correction (q + fma(-q, c, t)*r): the correctly rounded
            // quotient for a constant divisor at a third of the instructions of the generic IEEE division sequence
            auto div_rs2 = [&](float t) {
                const float r = 0.70710678118654752440f;
                const float q = t * r;
                return __builtin_fmaf(__builtin_fmaf(-q, rs2, t), r, q);
            };
            float o[4] = {div_rs2(v[p].x * inv[0] + sh[0]), div_rs2(v[p].y * inv[1] + sh[1]),
                          div_rs2(v[p].z * inv[2] + sh[2]), div_rs2(v[p].w * inv[3] + sh[3])};
            out_store4<3>(y + (size_t)r * cols + c4, o[0], o[1], o[2], o[3]);
            if (power_partial != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    // |y| <= peak (8 sigma of a unit-variance input: practically always): the clip factor is exactly
                    // peak / peak = 1, so the clipped power is the sum of squares already at hand -- same bits as the
                    // general form below, without its square root and division
                    const float sq = o[e] * o[e] + o[e + 1] * o[e + 1];
                    if (sq <= peak * peak) {
                        pw += (double)sq;
                    } else {
                        const float l2 = sqrtf(sq);
                        const float sc = peak / fmaxf(l2, peak);
                        const float ci = o[e] * sc, cq = o[e + 1] * sc;
                        pw += (double)(ci * ci + cq * cq);
                    }
                }
            }
        }
    }
    if (power_partial != nullptr) {
        pw = wave_sum(pw);
        if (lane == 0) pred[wave] = pw;
        __syncthreads();
        if (t == 0) {
            double a = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) a += pred[w];
            power_partial[bidx] = a;              // (idle strips past the last column contribute 0)
        }
    }
    if (virt && nv.npow_out != nullptr && bidx == 0) {          // `noise_power:0`: mean |noise|^2 from the generator's partials
        __syncthreads();
        double a = 0.0;
        for (int i = t; i < nv.n_noise; i += 256) a += nv.npart_noise[i];
        a = wave_sum(a);
        if (lane == 0) pred[wave] = a;
        __syncthreads();
        if (t == 0) nv.npow_out[0] = (float)(((pred[0] + pred[1]) + (pred[2] + pred[3])) / nv.total);
    }
}

template <int CG, int RPT>
__global__ __launch_bounds__(128 * CG) void norm_fused_kernel(const float* x_, float* y_, int batch, int cols, float eps, float peak,
                                                              double* power_partial_, float* mean_out_, float* var_out_,
                                                              dccn_adam_state* adam_, dccn_adam_hparams hp, const ChainOffs co) {
    const long long coff = co.off[blockIdx.z];                     // chain groups (common.h)
    const float* __restrict__ x = chain_at(x_, coff);
    float* __restrict__ y = chain_at(y_, coff);
    double* __restrict__ power_partial = chain_at(power_partial_, coff);
    float* __restrict__ mean_out = chain_at(mean_out_, coff);
    float* __restrict__ var_out = chain_at(var_out_, coff);
    dccn_adam_state* __restrict__ adam = chain_at(adam_, coff);
    norm_fused_body<CG, RPT>(x, y, batch, cols, eps, peak, power_partial, mean_out, var_out, adam, hp, (int)blockIdx.x,
                             (int)gridDim.x);
}

// out[0] = (float)(sum(partial[0..n)) / denom)
static __global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partial, int n, double denom,
                                                           float* __restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1] + red[2] + red[3]) / denom);
}

// ---- R8 standalone: clip + power over [n_pairs,2] ----------------------------------------
static __global__ __launch_bounds__(256) void clip_power_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         long long n_pairs, float peak,
                                                         double* __restrict__ partial) {
    __shared__ double red[4];
    double pw = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs; i += stride) {
        const float2 v = *reinterpret_cast<const float2*>(x + 2 * i);
        const float l2 = sqrtf(v.x * v.x + v.y * v.y);
        const float sc = peak / fmaxf(l2, peak);
        const float ci = v.x * sc, cq = v.y * sc;
        if (y != nullptr) *reinterpret_cast<float2*>(y + 2 * i) = make_float2(ci, cq);
        pw += (double)(ci * ci + cq * cq);
    }
    pw = wave_sum(pw);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pw;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ---- R7: TF Adam -----------------------------------------------------------------------
// prep: alpha = lr*sqrt(1-b2p)/(1-b1p) with lr = lr0*rate^floor(step/decay_steps); then the
// state advances (beta powers *= beta, global_step += 1) -- exactly one thread.
static __global__ void adam_prep_kernel(dccn_adam_state* __restrict__ st, dccn_adam_hparams hp) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float step = st->global_step;
        const float lr = hp.lr0 * powf(hp.decay_rate, floorf(step / hp.decay_steps));
        const float b1p = st->beta1_power, b2p = st->beta2_power;
        st->alpha = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
        st->beta1_power = b1p * hp.beta1;
        st->beta2_power = b2p * hp.beta2;
        st->global_step = step + 1.0f;
    }
}

// m += (g-m)*(1-b1); v += (g*g-v)*(1-b2); p -= m*alpha/(sqrt(v)+eps)   (TF ApplyAdam form)
// g = grad + gate*reg_coef*param
static __global__ __launch_bounds__(256) void adam_apply_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         const float* __restrict__ reg_coef,
                                                         const float* __restrict__ reg_gate,
                                                         const dccn_adam_state* __restrict__ st,
                                                         dccn_adam_hparams hp, long long n) {
    const float alpha = st->alpha;
    const float gate = reg_gate ? reg_gate[0] : 1.0f;
    const float omb1 = 1.0f - hp.beta1, omb2 = 1.0f - hp.beta2;
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            float4 p4 = *reinterpret_cast<float4*>(param + i);
            const float4 g4 = *reinterpret_cast<const float4*>(grad + i);
            float4 m4 = *reinterpret_cast<float4*>(m + i);
            float4 v4 = *reinterpret_cast<float4*>(v + i);
            float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (reg_coef) c4 = *reinterpret_cast<const float4*>(reg_coef + i);
            float* pp = reinterpret_cast<float*>(&p4);
            const float* gg = reinterpret_cast<const float*>(&g4);
            float* mm = reinterpret_cast<float*>(&m4);
            float* vv = reinterpret_cast<float*>(&v4);
            const float* cc = reinterpret_cast<const float*>(&c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g = gg[e] + (gate * cc[e]) * pp[e];
                mm[e] += (g - mm[e]) * omb1;
                vv[e] += (g * g - vv[e]) * omb2;
                pp[e] -= (mm[e] * alpha) / (sqrtf(vv[e]) + hp.eps);
            }
            *reinterpret_cast<float4*>(param + i) = p4;
            *reinterpret_cast<float4*>(m + i) = m4;
            *reinterpret_cast<float4*>(v + i) = v4;
        } else {
            for (long long j = i; j < n; ++j) {
                const float c = reg_coef ? reg_coef[j] : 0.f;
                const float g = grad[j] + (gate * c) * param[j];
                float mj = m[j], vj = v[j];
                mj += (g - mj) * omb1;
                vj += (g * g - vj) * omb2;
                param[j] -= (mj * alpha) / (sqrtf(vj) + hp.eps);
                m[j] = mj;
                v[j] = vj;
            }
        }
    }
}


// ---- R7 fused into the receiver step: one launch does the split-K reduction of the dense weight
// gradient and the Adam update over the whole arena.  alpha comes from the optimizer state, which the
// first kernel of the step (moments_kernel) has already advanced. ----
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
struct AdamRxArgs {
    float* param; float* grad; float* m; float* v;
    const float* reg_coef; const float* reg_gate;
    const dccn_adam_state* state;
    long long n;
    // dense kernel / bias gradient still in split-K slabs (splits > 1), else nullptr
    const float* dw_slabs; const float* db_slabs;
    int splits;
    long long o_dw, n_dw, o_db, n_db;      // arena segments of dense kernel / bias
    // C-Conv weight gradient still in split-K slabs of dWeff (fold_blocks > 0): the first fold_blocks blocks fold
    // them (cconv_fold_body) and update those parameters; the other blocks start at element n_conv
    const float* cw_slabs; const float* cw_colsum;
    int cw_splits, kin, F, fold_blocks;
    int cw_tilew;                          // > 0: the slabs are the column-tiled dX-epilogue partials (rx_bwd.h)
    long long cw_slab, o_cw, n_conv;       // o_cw: arena offset of the C-Conv kernel (its bias follows)
    // R0 of the NEXT batch riding on this launch (norm_blocks > 0): the leading blocks normalise nx -> ny, which no
    // kernel of the current step reads any more (software pipelining across steps: one boundary less per step)
    const float* nx; float* ny; double* npower;
    int nbatch, ncols, norm_blocks;
    float neps, npeak;
    NormVirtual nv;                        // y != nullptr: the next batch comes as (y, noise, power partials) instead of nx
    // [skip_lo, skip_hi): elements whose update already happened in the epilogue of their weight-gradient GEMM
    long long skip_lo, skip_hi;
    unsigned long long* stamp;             // step timeline stamps (common.h stamp_mark), nullptr = none
    int reg_uniform_dw;                    // reg_coef is one value over [o_dw, o_dw + n_dw): read once, not streamed
    int nt;                                // streaming hints: the gradient is loaded non-temporal (read exactly once)
    int skip_dw_grad;                      // 1: the sum of the dense kernel's split-K slabs feeds the update only and is not
                                           //    written to the gradient arena (dccn_rx_buffers.keep_dense_grad < 0: 2.3 MB of
                                           //    the launch's 14 MB of stores at C2)
};

// C-Conv parameters: fold the dWeff slabs (Appendix A.2) and apply the update right here.
// PUBLISH: the updated parameters are handed to other workgroups of the SAME launch (the C-Conv forward of the next
// batch, rx_update_prefetch_kernel): write-through (agent-scope relaxed atomic = sc1) stores, every wave drains its
// stores, one lane bumps the arrival counter (cdna_hip_programming.md Guideline 16, recipe R1).
// Hand-off words: ONE arrival counter for the producers; the last arriver fans the news out to one flag word per
// consumer block, 256 bytes apart (256 blocks polling one address serialise on that address's memory channel: measured
// 37 us for the launch instead of 13).  Flag value = the step's epoch (bits of the advanced global_step), so the flags
// never need clearing; the counter is reset by the backward launch of the same step.
struct HandoffWords {
    unsigned* counter;
    unsigned* flags;        // [n_flags] at a stride of kFlagStride words
    unsigned expected;      // producers
    int n_flags;
};
constexpr int kFlagStride = 64;
template <bool PUBLISH>
__device__ __forceinline__ void adam_fold_role(const AdamRxArgs& a, const dccn_adam_hparams& hp, const int bx,
                                               const HandoffWords hw) {
    const float alpha = a.state->alpha;
    const float gate = a.reg_gate ? a.reg_gate[0] : 1.0f;
    const float omb1 = 1.0f - hp.beta1, omb2 = 1.0f - hp.beta2;
    long long idx[2];
    float gv[2];
    // the two parameters this thread will update (group 0 of the fold: the same (n, f) / bias mapping as cconv_fold_body) are
    // known before the fold: their Adam operands are requested now and arrive under the fold's own loads
    float pre_p[2] = {0.f, 0.f}, pre_m[2] = {0.f, 0.f}, pre_v[2] = {0.f, 0.f}, pre_c[2] = {0.f, 0.f};
    {
        const int LANES = a.cw_tilew > 0 ? kFoldLanesTiled : kRedLanes;
        const int lane = threadIdx.x % LANES, grp = threadIdx.x / LANES;
        const int e = bx * LANES + lane, total = a.kin * a.F, N2 = 2 * a.F;
        if (grp == 0 && e < total + a.F) {
            const bool is_w = e < total;
            const int n = is_w ? e / a.F : a.kin, f = is_w ? e % a.F : e - total;
            const long long j0 = a.o_cw + (long long)n * N2 + f, j1 = j0 + a.F;
            pre_p[0] = a.param[j0]; pre_p[1] = a.param[j1];
            pre_m[0] = a.m[j0]; pre_m[1] = a.m[j1];
            pre_v[0] = a.v[j0]; pre_v[1] = a.v[j1];
            if (a.reg_coef) { pre_c[0] = a.reg_coef[j0]; pre_c[1] = a.reg_coef[j1]; }
        }
    }
    if (a.cw_tilew > 0)
        cconv_fold_body<kFoldLanesTiled>(a.cw_slabs, a.cw_splits, a.cw_slab, a.cw_colsum, a.grad + a.o_cw,
                                         a.grad + a.o_cw + (long long)a.kin * 2 * a.F, a.kin, a.F, bx, idx, gv, a.cw_tilew);
    else
        cconv_fold_body<kRedLanes>(a.cw_slabs, a.cw_splits, a.cw_slab, a.cw_colsum, a.grad + a.o_cw,
                                   a.grad + a.o_cw + (long long)a.kin * 2 * a.F, a.kin, a.F, bx, idx, gv, 0);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (idx[e] < 0) continue;
        const long long j = a.o_cw + idx[e];
        float p = pre_p[e], mm = pre_m[e], vv = pre_v[e];
        const float ge = gv[e] + (gate * pre_c[e]) * p;
        mm += (ge - mm) * omb1;
        vv += (ge * ge - vv) * omb2;
        p -= (mm * alpha) / (sqrtf(vv) + hp.eps);
        if constexpr (PUBLISH) __hip_atomic_store(a.param + j, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else a.param[j] = p;
        a.m[j] = mm; a.v[j] = vv;
    }
    if constexpr (PUBLISH) {
        __shared__ unsigned s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // EVERY storing wave drains its write-through stores
        __syncthreads();
        if (threadIdx.x == 0)
            s_last = __hip_atomic_fetch_add(hw.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == hw.expected - 1u ? 1u : 0u;
        __syncthreads();
        if (s_last) {
            // every producer drained before it counted itself in: all parameters are at the coherence point now
            const unsigned epoch = __builtin_bit_cast(unsigned, a.state->global_step);
            for (int i = threadIdx.x; i < hw.n_flags; i += blockDim.x)
                __hip_atomic_store(hw.flags + (size_t)i * kFlagStride, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// the rest of the arena (dense kernel / bias with their split-K slabs, tail weights): block bx of nbx streams it
template <int SPLITS>     // 0: runtime count
__device__ __forceinline__ void adam_stream_role(const AdamRxArgs& a, const dccn_adam_hparams& hp, const int bx, const int nbx) {
    const float alpha = a.state->alpha;
    const float gate = a.reg_gate ? a.reg_gate[0] : 1.0f;
    const float omb1 = 1.0f - hp.beta1, omb2 = 1.0f - hp.beta2;
    const bool seg4 = ((a.o_dw | a.n_dw | a.o_db | a.n_db) & 3) == 0;
    const bool uni = a.reg_uniform_dw != 0 && a.reg_coef != nullptr && seg4;
    const float cdw = uni ? a.reg_coef[a.o_dw] : 0.f;          // (uniform address: one scalar load)
    const int splits = SPLITS > 0 ? SPLITS : a.splits;
    const long long first = a.fold_blocks > 0 ? a.n_conv : 0;
    const long long stride = (long long)nbx * blockDim.x * 4;
    for (long long i = first + ((long long)bx * blockDim.x + threadIdx.x) * 4; i < a.n; i += stride) {
        if (i >= a.skip_lo && i + 4 <= a.skip_hi) continue;       // (segment bounds are multiples of 4)
        const int cnt = (int)((a.n - i) < 4 ? (a.n - i) : 4);
        float g[4] = {0.f, 0.f, 0.f, 0.f}, p[4], mm[4], vv[4], cc[4] = {0.f, 0.f, 0.f, 0.f};
        const bool full = cnt == 4;
        // everything this element needs is requested up front (independent loads), summed afterwards
        float4 p4, m4, v4, c4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (full && a.nt >= 2) {
            // (nt = 2: the launch shares the chip with an MFMA-bound GEMM on another stream -- nothing it touches should
            // displace that kernel's operand panels from the caches)
            const nt_f32x4 tp = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(a.param + i));
            const nt_f32x4 tm = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(a.m + i));
            const nt_f32x4 tv = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(a.v + i));
            p4 = make_float4(tp[0], tp[1], tp[2], tp[3]);
            m4 = make_float4(tm[0], tm[1], tm[2], tm[3]);
            v4 = make_float4(tv[0], tv[1], tv[2], tv[3]);
            if (uni && i >= a.o_dw && i < a.o_dw + a.n_dw) c4 = make_float4(cdw, cdw, cdw, cdw);
            else if (a.reg_coef) c4 = *reinterpret_cast<const float4*>(a.reg_coef + i);
        } else if (full) {
            p4 = *reinterpret_cast<const float4*>(a.param + i);
            m4 = *reinterpret_cast<const float4*>(a.m + i);
            v4 = *reinterpret_cast<const float4*>(a.v + i);
            if (uni && i >= a.o_dw && i < a.o_dw + a.n_dw) c4 = make_float4(cdw, cdw, cdw, cdw);
            else if (a.reg_coef) c4 = *reinterpret_cast<const float4*>(a.reg_coef + i);
        }
        // ---- gradient: plain arena, or the fixed-order sum of the split-K slabs ----
        if (full && seg4 && a.dw_slabs && i >= a.o_dw && i < a.o_dw + a.n_dw) {
            const float* q = a.dw_slabs + (i - a.o_dw);
            float4 s = *reinterpret_cast<const float4*>(q);
            if constexpr (SPLITS > 1) {
                float4 t[SPLITS - 1];
#pragma unroll
                for (int z = 1; z < SPLITS; ++z) t[z - 1] = *reinterpret_cast<const float4*>(q + (size_t)z * a.n_dw);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int z = 1; z < SPLITS; ++z) { s.x += t[z - 1].x; s.y += t[z - 1].y; s.z += t[z - 1].z; s.w += t[z - 1].w; }
            } else {
                for (int z = 1; z < splits; ++z) {
                    const float4 t = *reinterpret_cast<const float4*>(q + (size_t)z * a.n_dw);
                    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
                }
            }
            g[0] = s.x; g[1] = s.y; g[2] = s.z; g[3] = s.w;
            if (!a.skip_dw_grad) *reinterpret_cast<float4*>(a.grad + i) = s;
        } else if (full && !(a.dw_slabs && i < a.o_dw + a.n_dw && i + 4 > a.o_dw) &&
                   !(a.db_slabs && i < a.o_db + a.n_db && i + 4 > a.o_db)) {
            float4 s;
            if (a.nt) {
                const nt_f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(a.grad + i));
                s = make_float4(t[0], t[1], t[2], t[3]);
            } else {
                s = *reinterpret_cast<const float4*>(a.grad + i);
            }
            g[0] = s.x; g[1] = s.y; g[2] = s.z; g[3] = s.w;
        } else {
            for (int e = 0; e < cnt; ++e) {
                const long long j = i + e;
                float t;
                if (a.dw_slabs && j >= a.o_dw && j < a.o_dw + a.n_dw) {
                    t = 0.f;
                    for (int z = 0; z < splits; ++z) t += a.dw_slabs[(size_t)z * a.n_dw + (j - a.o_dw)];
                    a.grad[j] = t;
                } else if (a.db_slabs && j >= a.o_db && j < a.o_db + a.n_db) {
                    t = 0.f;
                    for (int z = 0; z < splits; ++z) t += a.db_slabs[(size_t)z * a.n_db + (j - a.o_db)];
                    a.grad[j] = t;
                } else {
                    t = a.grad[j];
                }
                g[e] = t;
            }
        }
        if (full) {
            p[0] = p4.x; p[1] = p4.y; p[2] = p4.z; p[3] = p4.w;
            mm[0] = m4.x; mm[1] = m4.y; mm[2] = m4.z; mm[3] = m4.w;
            vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
            cc[0] = c4.x; cc[1] = c4.y; cc[2] = c4.z; cc[3] = c4.w;
        } else {
            for (int e = 0; e < 4; ++e) {
                const bool in = e < cnt;
                p[e] = in ? a.param[i + e] : 0.f; mm[e] = in ? a.m[i + e] : 0.f; vv[e] = in ? a.v[i + e] : 0.f;
                cc[e] = (in && a.reg_coef) ? a.reg_coef[i + e] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = g[e] + (gate * cc[e]) * p[e];
            mm[e] += (ge - mm[e]) * omb1;
            vv[e] += (ge * ge - vv[e]) * omb2;
            p[e] -= (mm[e] * alpha) / (sqrtf(vv[e]) + hp.eps);
        }
        if (full && a.nt >= 2) {
            nt_f32x4 sp = {p[0], p[1], p[2], p[3]}, sm = {mm[0], mm[1], mm[2], mm[3]}, sv = {vv[0], vv[1], vv[2], vv[3]};
            __builtin_nontemporal_store(sp, reinterpret_cast<nt_f32x4*>(a.param + i));
            __builtin_nontemporal_store(sm, reinterpret_cast<nt_f32x4*>(a.m + i));
            __builtin_nontemporal_store(sv, reinterpret_cast<nt_f32x4*>(a.v + i));
        } else if (full) {
            out_store4<4>(a.param + i, p[0], p[1], p[2], p[3]);
            out_store4<4>(a.m + i, mm[0], mm[1], mm[2], mm[3]);
            out_store4<4>(a.v + i, vv[0], vv[1], vv[2], vv[3]);
        } else {
            for (int e = 0; e < cnt; ++e) { a.param[i + e] = p[e]; a.m[i + e] = mm[e]; a.v[i + e] = vv[e]; }
        }
    }
}

template <int SPLITS>     // 0: runtime count
__global__ __launch_bounds__(256) void adam_rx_kernel(const AdamRxArgs a, const dccn_adam_hparams hp) {
    stamp_mark(a.stamp, 0);
    if ((int)blockIdx.x < a.norm_blocks) {
        norm_fused_body<kNormFusedCG, kNormFusedRPT>(a.nx, a.ny, a.nbatch, a.ncols, a.neps, a.npeak, a.npower, nullptr, nullptr,
                                           nullptr, hp, (int)blockIdx.x, a.norm_blocks, a.nv);
        stamp_mark(a.stamp, 1);
        return;
    }
    const int bx = (int)blockIdx.x - a.norm_blocks, nbx = (int)gridDim.x - a.norm_blocks;
    if (bx < a.fold_blocks) {
        adam_fold_role<false>(a, hp, bx, HandoffWords{nullptr, nullptr, 0u, 0});
        stamp_mark(a.stamp, 1);
        return;
    }
    adam_stream_role<SPLITS>(a, hp, bx - a.fold_blocks, nbx - a.fold_blocks);
    stamp_mark(a.stamp, 1);
}

}  // namespace dccn
