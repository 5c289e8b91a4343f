// Patch gather for the general-k complex convolutions (dev/py/complex.py:51-92 layers_conv1d_complex and the k > 1 /
// strided uses of layers_conv2d_complex, :140-196): x [B, L, Wd, C, 2] -> rows [B*Lo*Wo, ntl*ntw*C, 2], the A operand
// of the fused C-Conv GEMM (gemm_f32_mfma.h), zero where TensorFlow's SAME padding lies.  Only the kernel taps that ever
// meet data are gathered (a contiguous range [t0, t0+nt) per axis: the other taps see nothing but padding, SURVEY.md
// Appendix A.2).  One pass over HBM each way; the backward is a gather too (every input element sums the patch entries
// that read it, in a fixed order) -- no atomics, deterministic.
#pragma once
#include "common.h"

namespace dccn {

struct Im2colGeom {
    int B, L, Wd, C;          // input [B, L, Wd, C, 2]
    int Lo, Wo;               // output positions
    int ntl, ntw, tl0, tw0;   // live taps per axis: tl0 .. tl0+ntl-1
    int sL, sW, pl0, pw0;     // strides, padding before
};

static __global__ __launch_bounds__(256) void im2col_kernel(const float2* __restrict__ x, float2* __restrict__ rows, Im2colGeom g,
                                                     long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int kc = g.ntl * g.ntw * g.C;
    const long long r = i / kc;
    int col = (int)(i - r * kc);
    const int c = col % g.C;
    col /= g.C;
    const int tj = col % g.ntw, ti = col / g.ntw;
    const int wo = (int)(r % g.Wo);
    const long long r2 = r / g.Wo;
    const int lo = (int)(r2 % g.Lo), b = (int)(r2 / g.Lo);
    const int l = lo * g.sL + g.tl0 + ti - g.pl0, w = wo * g.sW + g.tw0 + tj - g.pw0;
    float2 v = make_float2(0.f, 0.f);
    if (l >= 0 && l < g.L && w >= 0 && w < g.Wd) v = x[(((size_t)b * g.L + l) * g.Wd + w) * g.C + c];
    rows[i] = v;
}

static __global__ __launch_bounds__(256) void col2im_kernel(const float2* __restrict__ drows, float2* __restrict__ dx, Im2colGeom g,
                                                     long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % g.C);
    long long t = i / g.C;
    const int w = (int)(t % g.Wd);
    t /= g.Wd;
    const int l = (int)(t % g.L), b = (int)(t / g.L);
    const int kc = g.ntl * g.ntw * g.C;
    float2 acc = make_float2(0.f, 0.f);
    for (int ti = 0; ti < g.ntl; ++ti) {
        const int nl = l + g.pl0 - g.tl0 - ti;
        if (nl < 0 || nl % g.sL != 0) continue;
        const int lo = nl / g.sL;
        if (lo >= g.Lo) continue;
        for (int tj = 0; tj < g.ntw; ++tj) {
            const int nw = w + g.pw0 - g.tw0 - tj;
            if (nw < 0 || nw % g.sW != 0) continue;
            const int wo = nw / g.sW;
            if (wo >= g.Wo) continue;
            const float2 v = drows[(((size_t)b * g.Lo + lo) * g.Wo + wo) * kc + (size_t)(ti * g.ntw + tj) * g.C + c];
            acc.x += v.x;
            acc.y += v.y;
        }
    }
    dx[i] = acc;
}

}  // namespace dccn
