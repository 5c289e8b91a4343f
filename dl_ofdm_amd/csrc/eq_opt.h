// The equaliser step's LAST launch: every gradient reduction that used to be its own 5 us kernel (three C-Conv folds,
// the split-K sums of the dense layers, the fold of the block-Toeplitz smoothing layer, the tail's metric reduction)
// and the TF-Adam update of all 20 Equalizer/* variables (ofdmreceiver_np_mp.py:330) as ONE grid driven by a job table.
// A job is a contiguous run of blocks that produces the gradient of one arena segment from where the backward GEMMs
// left it and applies the update to those elements right away -- each parameter is read and written once, by the
// thread that reduced its gradient; no job depends on another, so there is nothing to order inside the launch.
//
//   EQJ_SUM           g[i] = sum_z src[z*slab + i]  (fixed order), or the gradient arena itself (src == nullptr)
//   EQJ_CCONV_FOLD    Appendix A.2 fold of the dWeff slabs of a (1,K) C-Conv into [Wa|Wb] and its bias pair
//   EQJ_CONV2D_FOLD   transpose of the block-Toeplitz expansion of the (S,K) smoothing C-Conv (one wave per tap)
//   EQJ_TAIL_FINALIZE the demodulation tail's per-block metrics -> dccn_metrics (no parameters: the receiver is frozen)
//   EQJ_PILOT_SNR     the pilot monitor of model.py:465-475 (no parameters either: it only has to run once eq exists)
#pragma once
#include "gemm_f32_mfma.h"
#include "tail.h"
#include "equalizer.h"

namespace dccn {

enum EqOptKind : int { EQJ_SUM = 0, EQJ_CCONV_FOLD = 1, EQJ_CONV2D_FOLD = 2, EQJ_TAIL_FINALIZE = 3, EQJ_PILOT_SNR = 4,
                        EQJ_NORM_NEXT = 5, EQJ_MONITOR = 6 };
struct EqOptJob {
    int kind, block0, blocks, splits;
    long long off, n;        // arena segment of the (first) variable
    long long off_b;         // folds: arena offset of the bias variable
    const float* src;        // slabs (SUM: nullptr = the gradient arena holds the finished gradient)
    const float* src2;       // folds: column-sum slabs
    long long slab, slab2;   // elements between consecutive slabs of src / src2
    int vec;                 // SUM: float4 loads of the slabs are legal (slab % 4 == 0, 16-byte aligned base)
    int reg_uniform;         // SUM: reg_coef holds one value over the job's segment (a dense kernel or bias: keras l2(0.01) on
                             // every element, model.py:371-462; a C-Conv kernel: zero) -- read once instead of streamed
    int kin, F;              // CCONV_FOLD: kin, F; CONV2D_FOLD: L, W
    __device__ __forceinline__ EqOptJob at_chain(const long long coff) const {          // chain groups (common.h)
        EqOptJob q = *this;
        q.src = chain_at(src, coff); q.src2 = chain_at(src2, coff);
        return q;
    }
};
constexpr int kEqOptJobs = 24;
struct EqOptPtrs {
    float* param; float* grad; float* m; float* v;
    const float* reg_coef;
    const dccn_adam_state* state;
    __device__ __forceinline__ void move_to_chain(const long long coff) {
        param = chain_at(param, coff); grad = chain_at(grad, coff); m = chain_at(m, coff); v = chain_at(v, coff);
        reg_coef = chain_at(reg_coef, coff); state = chain_at(state, coff);
    }
};
struct EqOptArgs : EqOptPtrs {
    int njobs;
    EqOptJob job[kEqOptJobs];
    // EQJ_TAIL_FINALIZE: chain g of a group (common.h) runs fin[fin_class[g]] -- chains of different modulations leave
    // different slab counts / gradient lengths; one chain: fin[0]
    TailFinalizeArgs fin[4];
    int fin_class[kMaxChains];
    // EQJ_PILOT_SNR
    const float2* ps_eq; const int* ps_carriers; float* ps_out;
    int ps_frames, ps_S, ps_K, ps_P;
    // EQJ_NORM_NEXT: `input:0` of the NEXT batch (dccn_eq_buffers.x_next), norm_adam.h norm_fused_body
    const float* nx; float* ny; double* npower;
    int nbatch, ncols;
    NormVirtual nv;               // nx as (y, noise, power partials) of the fused generator (dccn_eq_buffers.x_next_virtual)
    // EQJ_MONITOR: the training loop's chan_rms monitor (equalizer.h eq_monitor_body) as a job of this launch: it reads the
    // step's channel estimate and the generator's truth, both long complete; the scalar monitors are added by the finalize job
    EqMonitorArgs mon;
};
static_assert(sizeof(EqOptArgs) + sizeof(dccn_adam_hparams) + sizeof(ChainOffs) <= 4096, "kernel argument block");

struct AdamCoef {
    float alpha, omb1, omb2, eps;
};
__device__ __forceinline__ void eq_adam_one(const EqOptPtrs& a, const AdamCoef& k, const long long j, const float g) {
    float p = a.param[j], mm = a.m[j], vv = a.v[j];
    const float ge = g + (a.reg_coef ? a.reg_coef[j] : 0.f) * p;
    mm += (ge - mm) * k.omb1;
    vv += (ge * ge - vv) * k.omb2;
    p -= (mm * k.alpha) / (sqrtf(vv) + k.eps);
    a.param[j] = p; a.m[j] = mm; a.v[j] = vv;
}

__device__ __forceinline__ void eq_opt_sum(const EqOptPtrs& a, const EqOptJob& J, const AdamCoef& k, const int bx) {
    const long long stride = (long long)J.blocks * 256 * 4;
    const float creg = (J.reg_uniform && a.reg_coef) ? a.reg_coef[J.off] : 0.f;      // (uniform address: one scalar load)
    for (long long i = ((long long)bx * 256 + threadIdx.x) * 4; i < J.n; i += stride) {
        const long long j = J.off + i;
        if (i + 4 <= J.n && J.vec) {               // (segments start on 16-byte boundaries: eq_dims)
            const float4 p4 = *reinterpret_cast<const float4*>(a.param + j);
            const float4 m4 = *reinterpret_cast<const float4*>(a.m + j);
            const float4 v4 = *reinterpret_cast<const float4*>(a.v + j);
            float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (J.reg_uniform) c4 = make_float4(creg, creg, creg, creg);
            else if (a.reg_coef) c4 = *reinterpret_cast<const float4*>(a.reg_coef + j);
            float4 g4;
            if (J.src) {
                // the summation order of splitk_reduce_body (four interleaved runs, then 0+1+2+3): the step's gradients
                // are bit-identical to those of the launch-per-stage plan
                float4 run[kRedGroups];
                if (J.splits <= 2 * kRedGroups) {
                    // (uniform count) every slab's load is issued before the first add
                    float4 u[2 * kRedGroups];
#pragma unroll
                    for (int q = 0; q < 2 * kRedGroups; ++q) {
                        u[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (q < J.splits) u[q] = *reinterpret_cast<const float4*>(J.src + (size_t)q * J.slab + i);
                    }
#pragma unroll
                    for (int r = 0; r < kRedGroups; ++r) {
                        run[r] = u[r];
                        if (r + kRedGroups < J.splits) {
                            run[r].x += u[r + kRedGroups].x; run[r].y += u[r + kRedGroups].y;
                            run[r].z += u[r + kRedGroups].z; run[r].w += u[r + kRedGroups].w;
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < kRedGroups; ++r) {
                        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                        for (int zb = r; zb < J.splits; zb += 8 * kRedGroups) {       // batches of 8 loads, then 8 adds
                            float4 u[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q)
                                if (zb + q * kRedGroups < J.splits)
                                    u[q] = *reinterpret_cast<const float4*>(J.src + (size_t)(zb + q * kRedGroups) * J.slab + i);
#pragma unroll
                            for (int q = 0; q < 8; ++q)
                                if (zb + q * kRedGroups < J.splits) { t.x += u[q].x; t.y += u[q].y; t.z += u[q].z; t.w += u[q].w; }
                        }
                        run[r] = t;
                    }
                }
                g4 = run[0];
#pragma unroll
                for (int r = 1; r < kRedGroups; ++r) { g4.x += run[r].x; g4.y += run[r].y; g4.z += run[r].z; g4.w += run[r].w; }
                *reinterpret_cast<float4*>(a.grad + j) = g4;
            } else {
                g4 = *reinterpret_cast<const float4*>(a.grad + j);
            }
            float p[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
            const float g[4] = {g4.x, g4.y, g4.z, g4.w}, cc[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = g[e] + cc[e] * p[e];
                mm[e] += (ge - mm[e]) * k.omb1;
                vv[e] += (ge * ge - vv[e]) * k.omb2;
                p[e] -= (mm[e] * k.alpha) / (sqrtf(vv[e]) + k.eps);
            }
            *reinterpret_cast<float4*>(a.param + j) = make_float4(p[0], p[1], p[2], p[3]);
            *reinterpret_cast<float4*>(a.m + j) = make_float4(mm[0], mm[1], mm[2], mm[3]);
            *reinterpret_cast<float4*>(a.v + j) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
            // (a thread's own quad only: the grid stride hands [i, i+4) to exactly one thread -- unaligned / odd-sized
            // segments take this path for every quad, not just the last one)
            const long long e1 = (i + 4 < J.n) ? i + 4 : J.n;
            for (long long e = i; e < e1; ++e) {
                float g;
                if (J.src) {
                    float run[kRedGroups];
                    for (int r = 0; r < kRedGroups; ++r) {
                        float t = 0.f;
                        for (int z = r; z < J.splits; z += kRedGroups) t += J.src[(size_t)z * J.slab + e];
                        run[r] = t;
                    }
                    g = run[0];
                    for (int r = 1; r < kRedGroups; ++r) g += run[r];
                    a.grad[J.off + e] = g;
                } else {
                    g = a.grad[J.off + e];
                }
                eq_adam_one(a, k, J.off + e, g);
            }
        }
    }
}

// element i of a split-K result, summed in splitk_reduce_body's order
__device__ __forceinline__ float slab_sum(const float* __restrict__ src, const size_t i, const int splits, const long long slab) {
    if (splits == 1) return src[i];
    float run[kRedGroups];
    if (splits <= 2 * kRedGroups) {
        float u[2 * kRedGroups];
#pragma unroll
        for (int q = 0; q < 2 * kRedGroups; ++q) {
            u[q] = 0.f;
            if (q < splits) u[q] = src[(size_t)q * slab + i];
        }
#pragma unroll
        for (int r = 0; r < kRedGroups; ++r) {
            run[r] = u[r];
            if (r + kRedGroups < splits) run[r] += u[r + kRedGroups];
        }
    } else {
#pragma unroll
        for (int r = 0; r < kRedGroups; ++r) {
            float t = 0.f;
            for (int z = r; z < splits; z += kRedGroups) t += src[(size_t)z * slab + i];
            run[r] = t;
        }
    }
    return ((run[0] + run[1]) + run[2]) + run[3];
}

// dev/py/complex.py:185-188 with a (S,K) kernel, 'same' padding, one filter, expanded to T[(s',k',iq)][(s,k,re/im)]
// (equalizer.h cconv2d_same_expand_kernel); its transpose: tap (a,b) gathers its diagonal of dT
__device__ __forceinline__ void eq_opt_conv2d(const EqOptPtrs& a, const EqOptJob& J, const AdamCoef& k, const int bx) {
    const int L = J.kin, W = J.F, n = L * W * 2;
    const int padL = (L - 1) / 2, padW = (W - 1) / 2;
    const int tap = bx * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tap > L * W) return;
    if (tap == L * W) {                              // bias pair: d = sum_c (dbe[2c] - dbe[2c+1])
        float d = 0.f;
        for (int c = lane; c < L * W; c += 64)
            d += slab_sum(J.src2, 2 * c, J.splits, J.slab2) - slab_sum(J.src2, 2 * c + 1, J.splits, J.slab2);
        d = wave_sum(d);
        if (lane == 0) {
            a.grad[J.off_b] = d;
            a.grad[J.off_b + 1] = -d;
            eq_adam_one(a, k, J.off_b, d);
            eq_adam_one(a, k, J.off_b + 1, -d);
        }
        return;
    }
    const int ta = tap / W, tb = tap % W;
    float ga = 0.f, gb = 0.f;
    if (J.splits == 1 && L * W <= 8 * 64) {
        // the usual case (one finished dT): a lane's <= 8 cells x 4 gathers are requested together -- one memory latency
        // instead of eight dependent ones on the long pole of the optimizer launch; same order of additions
        float v[8][4];
        bool ok[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = lane + 64 * u;
            const int cc = min(c, L * W - 1);
            const int s = cc / W, kk = cc % W;
            const int sp = s + ta - padL, kp = kk + tb - padW;
            ok[u] = c < L * W && sp >= 0 && sp < L && kp >= 0 && kp < W;
            const size_t r0 = ok[u] ? (size_t)((sp * W + kp) * 2) * n + (size_t)cc * 2 : 0;
            v[u][0] = J.src[r0]; v[u][1] = J.src[r0 + n + 1]; v[u][2] = J.src[r0 + 1]; v[u][3] = J.src[r0 + n];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (ok[u]) { ga += v[u][0] - v[u][1]; gb += v[u][2] - v[u][3]; }
        }
    } else
    for (int c = lane; c < L * W; c += 64) {
        const int s = c / W, kk = c % W;
        const int sp = s + ta - padL, kp = kk + tb - padW;
        if (sp < 0 || sp >= L || kp < 0 || kp >= W) continue;
        const size_t r0 = (size_t)((sp * W + kp) * 2) * n + (size_t)c * 2;
        // d/dWa: (I->re) - (Q->im), d/dWb: (I->im) - (Q->re); each dT element = the split-K sum of its slabs
        ga += slab_sum(J.src, r0, J.splits, J.slab) - slab_sum(J.src, r0 + n + 1, J.splits, J.slab);
        gb += slab_sum(J.src, r0 + 1, J.splits, J.slab) - slab_sum(J.src, r0 + n, J.splits, J.slab);
    }
    ga = wave_sum(ga);
    gb = wave_sum(gb);
    if (lane == 0) {
        a.grad[J.off + tap * 2] = ga;
        a.grad[J.off + tap * 2 + 1] = gb;
        eq_adam_one(a, k, J.off + tap * 2, ga);
        eq_adam_one(a, k, J.off + tap * 2 + 1, gb);
    }
}

// Optimizer work riding BEHIND another launch's own workgroups (round 4): the Adam update of a dense kernel whose gradient is
// already complete in the arena is a pure stream (7 arrays, ~22 MB for 896 x 896), and the launches of the backward chain
// are latency-bound with the memory system idle -- so the two big kernels of the equaliser are updated by extra workgroups
// of a LATER backward launch that neither reads nor writes them, instead of in the optimizer launch at the end of the chain
// (which was 16 us, 9 of them these two streams).  Same arithmetic as EQJ_SUM (eq_opt_sum): bit-identical parameters.
struct EqRideArgs {
    EqOptPtrs p;
    int njobs, blocks;
    EqOptJob job[3];         // EQJ_SUM (no slabs) / EQJ_CONV2D_FOLD
    __device__ __forceinline__ EqRideArgs at_chain(const long long coff) const {
        EqRideArgs q = *this;
        q.p.move_to_chain(coff);
#pragma unroll
        for (int i = 0; i < 3; ++i) q.job[i] = job[i].at_chain(coff);
        return q;
    }
};
// coff: the chain's arena offset (common.h); the table itself stays in the kernel-argument segment (indexing a rebased COPY of
// it would put the copy into scratch memory), only the job this block runs is rebased
__device__ __forceinline__ void eq_ride_body(const EqRideArgs& r, const dccn_adam_hparams& hp, const int bx, const long long coff = 0) {
    if (bx >= r.blocks) return;
    int j = 0;
    while (j + 1 < r.njobs && bx >= r.job[j + 1].block0) ++j;
    const EqOptJob J = r.job[j].at_chain(coff);
    EqOptPtrs p = r.p;
    p.move_to_chain(coff);
    AdamCoef k;
    k.alpha = p.state->alpha; k.omb1 = 1.0f - hp.beta1; k.omb2 = 1.0f - hp.beta2; k.eps = hp.eps;
    if (J.kind == EQJ_CONV2D_FOLD) eq_opt_conv2d(p, J, k, bx - J.block0);
    else eq_opt_sum(p, J, k, bx - J.block0);
}

static __global__ __launch_bounds__(256) void eq_opt_kernel(const EqOptArgs a0, const dccn_adam_hparams hp, const ChainOffs co) {
    const int chain = (int)blockIdx.z;                               // chain groups (common.h)
    const long long coff = co.off[chain];
    int j = 0;
    while (j + 1 < a0.njobs && (int)blockIdx.x >= a0.job[j + 1].block0) ++j;      // (block-uniform scan of the table)
    const EqOptJob J = a0.job[j].at_chain(coff);
    const int bx = (int)blockIdx.x - J.block0;
    if (J.kind == EQJ_NORM_NEXT) {
        norm_fused_body<kNormFusedCG, kNormFusedRPT>(chain_at(a0.nx, coff), chain_at(a0.ny, coff), a0.nbatch, a0.ncols, 1e-9f, 8.0f,
                                                     chain_at(a0.npower, coff), nullptr, nullptr, nullptr, hp, bx, J.blocks,
                                                     a0.nv.at_chain(coff));
        return;
    }
    if (J.kind == EQJ_TAIL_FINALIZE) {
        demod_tail_finalize_body(a0.fin[a0.fin_class[chain]].at_chain(coff), bx);
        return;
    }
    if (J.kind == EQJ_MONITOR) {
        const EqMonitorArgs m = a0.mon.at_chain(coff);
        eq_monitor_body(m, (unsigned)bx, (unsigned)J.blocks, false);
        return;
    }
    if (J.kind == EQJ_PILOT_SNR) {
        const int frame = bx * 4 + (int)(threadIdx.x >> 6);
        if (frame < a0.ps_frames)
            pilot_snr_body<false>(chain_at(a0.ps_eq, coff), nullptr, nullptr, chain_at(a0.ps_carriers, coff), chain_at(a0.ps_out, coff),
                                  a0.ps_S, a0.ps_K, a0.ps_P, frame, (int)(threadIdx.x & 63));
        return;
    }
    EqOptPtrs a = a0;
    a.move_to_chain(coff);
    AdamCoef k;
    k.alpha = a.state->alpha; k.omb1 = 1.0f - hp.beta1; k.omb2 = 1.0f - hp.beta2; k.eps = hp.eps;
    if (J.kind == EQJ_SUM) {
        eq_opt_sum(a, J, k, bx);
    } else if (J.kind == EQJ_CCONV_FOLD) {
        long long idx[2];
        float gv[2];
        cconv_fold_body<kRedLanes>(J.src, J.splits, J.slab, J.src2, a.grad + J.off, a.grad + J.off_b, J.kin, J.F, bx, idx, gv, 0);
        const long long nw = (long long)J.kin * 2 * J.F;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (idx[e] < 0) continue;
            eq_adam_one(a, k, idx[e] < nw ? J.off + idx[e] : J.off_b + (idx[e] - nw), gv[e]);
        }
    } else {
        eq_opt_conv2d(a, J, k, bx);
    }
}

// host side: the table is built per step from what the backward GEMMs report
struct EqOptBuilder {
    EqOptArgs a;
    int blocks = 0;
    int status = DCCN_OK;
    EqOptJob* add(int kind, int nblocks) {
        if (a.njobs >= kEqOptJobs) { status = DCCN_ERR_STATE; return nullptr; }
        EqOptJob* J = &a.job[a.njobs++];
        memset(J, 0, sizeof(*J));
        J->kind = kind; J->block0 = blocks; J->blocks = nblocks;
        blocks += nblocks;
        return J;
    }
    static int stream_blocks(long long n) {
        long long b = ceil_div_ll(ceil_div_ll(n, 4), 256);
        if (b > 8 * kCUs) b = 8 * kCUs;             // (one float4 per thread up to 2 M elements: no dependent second round)
        return (int)(b < 1 ? 1 : b);
    }
    // [off, off+n): finished gradient in the arena (adjacent segments merge into one job)
    // uniform: reg_coef is one value over [off, off + n) (see EqOptJob::reg_uniform); merged segments must agree on it AND
    // on the value, which only the caller knows: kernel + bias of one layer share their coefficient, different layers' values
    // are equal in this model, so segments merge when both are uniform (or both are not)
    void plain(long long off, long long n, bool uniform = false) {
        if (a.njobs > 0) {
            EqOptJob& P = a.job[a.njobs - 1];
            if (P.kind == EQJ_SUM && P.src == nullptr && P.off + P.n <= off && off - (P.off + P.n) < 4 &&
                (P.reg_uniform != 0) == uniform && !uniform) {
                blocks -= P.blocks;                     // (the gap is alignment padding: zero gradient, zero state)
                P.n = off + n - P.off;
                P.blocks = stream_blocks(P.n);
                blocks += P.blocks;
                return;
            }
        }
        EqOptJob* J = add(EQJ_SUM, stream_blocks(n));
        if (J) { J->off = off; J->n = n; J->splits = 1; J->vec = 1; J->reg_uniform = uniform ? 1 : 0; }
    }
    void slabs(long long off, long long n, const float* src, int splits, long long slab, bool uniform = false) {
        EqOptJob* J = add(EQJ_SUM, stream_blocks(n));
        // odd K or K + CP (e.g. CP = 9: a 146-float bias slab) leaves slabs that are not multiples of four floats apart:
        // those jobs take the element-wise branch
        if (J) {
            J->off = off; J->n = n; J->src = src; J->splits = splits; J->slab = slab;
            J->vec = (slab % 4 == 0 && aligned16(src)) ? 1 : 0;
            J->reg_uniform = uniform ? 1 : 0;
        }
    }
    void cconv_fold(long long off, long long off_b, const float* slabs_, const float* colsum, int splits, long long slab,
                    int kin, int F) {
        EqOptJob* J = add(EQJ_CCONV_FOLD, ceil_div(kin * F + F, kRedLanes));
        if (J) { J->off = off; J->off_b = off_b; J->src = slabs_; J->src2 = colsum; J->splits = splits; J->slab = slab; J->kin = kin; J->F = F; }
    }
    void conv2d_fold(long long off, long long off_b, const float* dT, const float* dbe, int splits, long long slab,
                     long long slab2, int L, int W) {
        EqOptJob* J = add(EQJ_CONV2D_FOLD, ceil_div(L * W + 1, 4));
        if (J) { J->off = off; J->off_b = off_b; J->src = dT; J->src2 = dbe; J->splits = splits; J->slab = slab; J->slab2 = slab2; J->kin = L; J->F = W; }
    }
    void pilot_snr(const float* eq, const int* carriers, float* out, int frames, int S, int K, int P) {
        EqOptJob* J = add(EQJ_PILOT_SNR, ceil_div(frames, 4));
        if (J) {
            a.ps_eq = reinterpret_cast<const float2*>(eq); a.ps_carriers = carriers; a.ps_out = out;
            a.ps_frames = frames; a.ps_S = S; a.ps_K = K; a.ps_P = P;
        }
    }
    // (first job of the table: its blocks are dispatched first -- a latency chain, not a stream)
    void norm_next(const float* x, float* y, int batch, int cols, double* power_partial, int nblocks,
                   const NormVirtual nv = norm_virtual_none()) {
        EqOptJob* J = add(EQJ_NORM_NEXT, nblocks);
        if (J) { a.nx = x; a.ny = y; a.nbatch = batch; a.ncols = cols; a.npower = power_partial; a.nv = nv; }
    }
    void monitor(const EqMonitorArgs& m, int nblocks) {
        EqOptJob* J = add(EQJ_MONITOR, nblocks);
        if (J) a.mon = m;
    }
    void tail_finalize(const TailFinalizeArgs* fin, int n_class, const int* fin_class) {
        int blocks_ = 0;
        for (int c = 0; c < n_class; ++c) blocks_ = tail_finalize_blocks(fin[c].P) > blocks_ ? tail_finalize_blocks(fin[c].P) : blocks_;
        EqOptJob* J = add(EQJ_TAIL_FINALIZE, blocks_);
        if (J) {
            for (int c = 0; c < n_class; ++c) a.fin[c] = fin[c];
            for (int g = 0; g < kMaxChains; ++g) a.fin_class[g] = fin_class[g];
        }
    }
};

static int launch_eq_opt(EqOptBuilder& b, dccn_adam_hparams hp, hipStream_t s) {
    if (b.status != DCCN_OK) return b.status;
    if (b.blocks <= 0) return DCCN_OK;
    DCCN_LAUNCH_CHAINS_Z(eq_opt_kernel, dim3((unsigned)b.blocks), dim3(256), 0, s, b.a, hp);
    DCCN_LAUNCH_CHECK();
    return DCCN_OK;
}

}  // namespace dccn
