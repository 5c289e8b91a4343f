"""Tensor-level operators over libdccn.so: raw calls + ``torch.autograd.Function`` wrappers.

torch is plumbing here (device memory, streams, autograd bookkeeping); all arithmetic
happens in the HIP kernels.  Every op raises if its input is not a CUDA float32 tensor --
there is deliberately no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import Metrics, check

__all__ = ["batch_moment_norm", "clip_power", "cconv_gemm", "dense", "demod_tail_loss",
           "demod_tail_eval", "read_metrics", "tail_param_count", "pack_tail_params",
           "unpack_tail_params", "workspace"]


# ---- plumbing -----------------------------------------------------------------------------
def _need_cuda(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.DccnError("dl_ofdm_amd ops need CUDA (ROCm) tensors; got a %s tensor -- there is no "
                                 "CPU fallback" % t.device)
        if not t.is_contiguous():
            raise _lib.DccnError("dl_ofdm_amd ops need contiguous tensors")


def _f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise TypeError("expected float32, got %s" % t.dtype)


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_ws_cache = {}


def workspace(nbytes: int, device, key: str = "default") -> torch.Tensor:
    """Grow-only per-(device,key) scratch buffer (stream-ordered reuse on the current stream)."""
    k = (str(device), key)
    t = _ws_cache.get(k)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[k] = t
    return t


def tail_param_count(nbits: int) -> int:
    m = 2 ** nbits
    return 2 * m + m + (m + 2) * 2 * nbits + 2 * nbits


def pack_tail_params(w1, b1, w2, b2) -> torch.Tensor:
    """[2,m] | [m] | [m+2,2b] | [2b] -> flat (dccn.h tail packing)."""
    return torch.cat([w1.reshape(-1), b1.reshape(-1), w2.reshape(-1), b2.reshape(-1)]).contiguous()


def unpack_tail_params(flat: torch.Tensor, nbits: int):
    m = 2 ** nbits
    o = 0
    w1 = flat[o:o + 2 * m].view(2, m); o += 2 * m
    b1 = flat[o:o + m]; o += m
    w2 = flat[o:o + (m + 2) * 2 * nbits].view(m + 2, 2 * nbits); o += (m + 2) * 2 * nbits
    b2 = flat[o:o + 2 * nbits]
    return w1, b1, w2, b2


# ---- R0 / R8 ------------------------------------------------------------------------------
def batch_moment_norm(x: torch.Tensor, eps: float = 1e-9, return_moments: bool = False):
    """ofdmreceiver_np.py:128-129: moments over axis 0 + batch_normalization / sqrt(2)."""
    _need_cuda(x); _f32(x)
    lib = _lib.load()
    batch = x.shape[0]
    cols = x.numel() // batch
    y = torch.empty_like(x)
    mean = torch.empty(cols, dtype=torch.float32, device=x.device) if return_moments else None
    var = torch.empty(cols, dtype=torch.float32, device=x.device) if return_moments else None
    nws = lib.dccn_batch_moment_norm_workspace_size(batch, cols)
    ws = workspace(nws, x.device)
    check(lib.dccn_batch_moment_norm_fwd(_p(x), _p(y), _p(mean), _p(var), batch, cols, eps, _p(ws), nws, _stream()),
          "dccn_batch_moment_norm_fwd")
    if return_moments:
        return y, mean.view(x.shape[1:]), var.view(x.shape[1:])
    return y


def clip_power(x: torch.Tensor, peak: float = 1.0, want_clipped: bool = True):
    """complex.py:21-27 ``complex_clip``: (clip_by_norm over IQ, mean clipped power)."""
    _need_cuda(x); _f32(x)
    if x.shape[-1] != 2:
        raise AssertionError("last axis must be IQ (2)")
    lib = _lib.load()
    n_pairs = x.numel() // 2
    y = torch.empty_like(x) if want_clipped else None
    power = torch.empty(1, dtype=torch.float32, device=x.device)
    nws = lib.dccn_clip_power_workspace_size(n_pairs)
    ws = workspace(nws, x.device)
    check(lib.dccn_clip_power(_p(x), _p(y), _p(power), n_pairs, peak, _p(ws), nws, _stream()), "dccn_clip_power")
    return y, power[0]


# ---- R1 -----------------------------------------------------------------------------------
class _CConvGemm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        _need_cuda(x, w, bias); _f32(x, w, bias)
        lib = _lib.load()
        rows, kin, _ = x.shape
        F = w.shape[1] // 2
        out = torch.empty(rows, F, 2, dtype=torch.float32, device=x.device)
        check(lib.dccn_cconv_gemm_fwd(_p(x), _p(w), _p(bias), _p(out), rows, kin, F, _stream()),
              "dccn_cconv_gemm_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        lib = _lib.load()
        dout = dout.contiguous()
        rows, kin, _ = x.shape
        F = w.shape[1] // 2
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(lib.dccn_cconv_gemm_bwd_x(_p(dout), _p(w), _p(dx), rows, kin, F, _stream()),
                  "dccn_cconv_gemm_bwd_x")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            db = torch.empty(2 * F, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            nws = lib.dccn_cconv_gemm_bwd_w_workspace_size(rows, kin, F)
            ws = workspace(nws, x.device)
            check(lib.dccn_cconv_gemm_bwd_w(_p(x), _p(dout), _p(dw), _p(db), rows, kin, F, _p(ws), nws, _stream()),
                  "dccn_cconv_gemm_bwd_w")
        return dx, dw, db


class _Im2col(torch.autograd.Function):
    """dccn_cconv_im2col / _col2im: the patch gather in front of the C-Conv GEMM and its adjoint."""

    @staticmethod
    def forward(ctx, x, geom):
        _need_cuda(x); _f32(x)
        B, L, Wd, C, _ = x.shape
        Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0 = geom
        rows = torch.empty(B * Lo * Wo, ntl * ntw * C, 2, dtype=torch.float32, device=x.device)
        check(_lib.load().dccn_cconv_im2col(_p(x), _p(rows), B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0,
                                            _stream()), "dccn_cconv_im2col")
        ctx.geom, ctx.shape = geom, (B, L, Wd, C)
        return rows

    @staticmethod
    def backward(ctx, drows):
        B, L, Wd, C = ctx.shape
        Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0 = ctx.geom
        drows = drows.contiguous()
        dx = torch.empty(B, L, Wd, C, 2, dtype=torch.float32, device=drows.device)
        check(_lib.load().dccn_cconv_col2im(_p(drows), _p(dx), B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0,
                                            _stream()), "dccn_cconv_col2im")
        return dx, None


def cconv_im2col(x: torch.Tensor, Lo: int, Wo: int, taps_l, taps_w, strides, pads) -> torch.Tensor:
    """x [B, L, Wd, C, 2] -> patch rows [B*Lo*Wo, len(taps_l)*len(taps_w)*C, 2] (zeros in the SAME padding); taps_* are
    the contiguous live-tap ranges, strides (sL, sW), pads (pad_before_L, pad_before_W).  Differentiable.

    Restriction (stated here because the layer API relies on it): the live taps of each axis -- the taps that meet data
    for at least one output position -- must be a contiguous range.  That holds for every stride / 'valid' / 'same'
    combination of the reference's layers (complex.py:51-92,140-196: undilated kernels, the taps reachable from output o
    are an interval that slides by the stride); a dilated kernel would break it and is rejected."""
    tl, tw = list(taps_l), list(taps_w)
    if tl != list(range(tl[0], tl[0] + len(tl))) or tw != list(range(tw[0], tw[0] + len(tw))):
        raise ValueError("live taps must form a contiguous range (undilated kernels only: see the docstring)")
    geom = (int(Lo), int(Wo), len(tl), len(tw), int(tl[0]), int(tw[0]), int(strides[0]), int(strides[1]), int(pads[0]),
            int(pads[1]))
    return _Im2col.apply(x.contiguous(), geom)


_PATCH_BWD_FUSED1D = True        # few-channel 1-D C-Convs: all three gradients in one pass over dout (dccn_cconv1d_bwd)
_PATCH_BWD_IM2COL = False        # tools/convbench.py, tests: force the backward onto the im2col / col2im operators
_PATCH_BWD_DX_ALWAYS = False     # ... or the input gradient onto the implicit GEMM wherever it qualifies (bit 1), not only
                                 # where dccn_cconv_patch_bwd_supported expects it to be the faster route (bit 2)


class _CConvPatch(torch.autograd.Function):
    """dccn_cconv_patch_fwd: the general-k complex convolution as an implicit GEMM (the operand loader gathers the taps;
    no patch tensor in the forward).  The backward is implicit too where dccn_cconv_patch_bwd_supported says so: the weight
    gradient's GEMM gathers its patch rows from x (dccn_cconv_patch_bwd_w), the input gradient is the same implicit GEMM
    over dout with the taps flipped (dccn_cconv_patch_bwd_x: any stride; 16-column tiles for few channels).  Only where
    that is expected to lose (taps outnumbering the channels ~10:1, wide strided inputs): GEMM into the gradient of the
    patches, scattered back by dccn_cconv_col2im."""

    @staticmethod
    def forward(ctx, x, w, bias, geom):
        _need_cuda(x, w, bias); _f32(x, w, bias)
        B, L, Wd, C, _ = x.shape
        Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0 = geom
        F = w.shape[1] // 2
        out = torch.empty(B * Lo * Wo, F, 2, dtype=torch.float32, device=x.device)
        check(_lib.load().dccn_cconv_patch_fwd(_p(x), _p(w), _p(bias), _p(out), B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW,
                                               pl0, pw0, F, _stream()), "dccn_cconv_patch_fwd")
        ctx.save_for_backward(x, w)
        ctx.geom, ctx.has_bias = geom, bias is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        lib = _lib.load()
        B, L, Wd, C, _ = x.shape
        Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0 = ctx.geom
        dout = dout.contiguous()
        rows, kin, F = B * Lo * Wo, ntl * ntw * C, w.shape[1] // 2
        dx = dw = db = None
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        mode = 0 if _PATCH_BWD_IM2COL else lib.dccn_cconv_patch_bwd_supported(B, L, Wd, C, Lo, Wo, ntl, ntw, sL, sW, F)
        if (_PATCH_BWD_FUSED1D and not _PATCH_BWD_IM2COL and ctx.needs_input_grad[0] and need_w and Wd == 1 and Wo == 1 and ntw == 1
                and sW == 1 and lib.dccn_cconv1d_bwd_supported(B, L, C, Lo, ntl, sL, F)):
            # one pass over dout for dx, dw and dbias (csrc/cconv1d_bwd.h)
            dx, dw = torch.empty_like(x), torch.empty_like(w)
            db = torch.empty(2 * F, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            nws = lib.dccn_cconv1d_bwd_workspace_size(F)
            ws = workspace(nws, x.device)
            check(lib.dccn_cconv1d_bwd(_p(x), _p(dout), _p(w), _p(dx), _p(dw), _p(db), B, L, C, Lo, ntl, tl0, sL, pl0, F, _p(ws), nws,
                                       _stream()), "dccn_cconv1d_bwd")
            return dx, dw, db, None
        if ctx.needs_input_grad[0] and (mode & (2 if _PATCH_BWD_DX_ALWAYS else 4)):    # implicit GEMM over dout, taps flipped
            dx = torch.empty_like(x)
            nws = lib.dccn_cconv_patch_bwd_x_workspace_size(C, ntl, ntw, F)
            ws = workspace(nws, x.device)
            check(lib.dccn_cconv_patch_bwd_x(_p(dout), _p(w), _p(dx), B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0,
                                             F, _p(ws), nws, _stream()), "dccn_cconv_patch_bwd_x")
        elif ctx.needs_input_grad[0]:
            drows = torch.empty(rows, kin, 2, dtype=torch.float32, device=x.device)
            check(lib.dccn_cconv_gemm_bwd_x(_p(dout), _p(w), _p(drows), rows, kin, F, _stream()), "dccn_cconv_gemm_bwd_x")
            dx = torch.empty_like(x)
            check(lib.dccn_cconv_col2im(_p(drows), _p(dx), B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0, _stream()),
                  "dccn_cconv_col2im")
        if need_w and (mode & 1):                           # the weight-gradient GEMM gathers its patch rows from x
            dw = torch.empty_like(w)
            db = torch.empty(2 * F, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            nws = lib.dccn_cconv_patch_bwd_w_workspace_size(B, Lo, Wo, C, ntl, ntw, F)
            ws = workspace(nws, x.device)
            check(lib.dccn_cconv_patch_bwd_w(_p(x), _p(dout), _p(dw), _p(db), B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW,
                                             pl0, pw0, F, _p(ws), nws, _stream()), "dccn_cconv_patch_bwd_w")
        elif need_w:
            patches = torch.empty(rows, kin, 2, dtype=torch.float32, device=x.device)
            check(lib.dccn_cconv_im2col(_p(x), _p(patches), B, L, Wd, C, Lo, Wo, ntl, ntw, tl0, tw0, sL, sW, pl0, pw0, _stream()),
                  "dccn_cconv_im2col")
            dw = torch.empty_like(w)
            db = torch.empty(2 * F, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            nws = lib.dccn_cconv_gemm_bwd_w_workspace_size(rows, kin, F)
            ws = workspace(nws, x.device)
            check(lib.dccn_cconv_gemm_bwd_w(_p(patches), _p(dout), _p(dw), _p(db), rows, kin, F, _p(ws), nws, _stream()),
                  "dccn_cconv_gemm_bwd_w")
        return dx, dw, db, None


def cconv_patch_supported(x: torch.Tensor, Lo: int, Wo: int, ntl: int, ntw: int, F: int) -> bool:
    B, L, Wd, C, _ = x.shape
    return bool(_lib.load().dccn_cconv_patch_supported(B, L, Wd, C, int(Lo), int(Wo), int(ntl), int(ntw), int(F)))


def cconv_patch(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], Lo: int, Wo: int, taps_l, taps_w, strides,
                pads) -> torch.Tensor:
    """General-k complex convolution, x [B, L, Wd, C, 2], w [ntl*ntw*C, 2F] over the live taps -> [B*Lo*Wo, F, 2]: what
    ``cconv_gemm(cconv_im2col(x, ...), w, bias)`` computes, with the patches gathered inside the GEMM (same geometry
    arguments and the same contiguous-live-taps restriction as :func:`cconv_im2col`)."""
    tl, tw = list(taps_l), list(taps_w)
    if tl != list(range(tl[0], tl[0] + len(tl))) or tw != list(range(tw[0], tw[0] + len(tw))):
        raise ValueError("live taps must form a contiguous range (undilated kernels only)")
    geom = (int(Lo), int(Wo), len(tl), len(tw), int(tl[0]), int(tw[0]), int(strides[0]), int(strides[1]), int(pads[0]),
            int(pads[1]))
    return _CConvPatch.apply(x.contiguous(), w.contiguous(), None if bias is None else bias.contiguous(), geom)


def cconv_gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """Complex convolution in GEMM form (complex.py:140-196 after im2col).

    x [rows, kin, 2], w [kin, 2F] = [Wa|Wb], bias [2F] = [ba|bb] -> [rows, F, 2] with
    re = I.Wa - Q.Wb + (ba-bb), im = I.Wb - Q.Wa + (bb-ba)."""
    if x.dim() != 3 or x.shape[2] != 2 or w.dim() != 2 or w.shape[0] != x.shape[1] or w.shape[1] % 2:
        raise TypeError("cconv_gemm: x [rows,kin,2], w [kin,2F]")
    return _CConvGemm.apply(x.contiguous(), w.contiguous(), None if bias is None else bias.contiguous())


# ---- R2 -----------------------------------------------------------------------------------
class _Dense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        _need_cuda(x, w, bias); _f32(x, w, bias)
        lib = _lib.load()
        M, K = x.shape
        N = w.shape[1]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        check(lib.dccn_dense_fwd(_p(x), _p(w), _p(bias), _p(y), M, K, N, _stream()), "dccn_dense_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        lib = _lib.load()
        dy = dy.contiguous()
        M, K = x.shape
        N = w.shape[1]
        dx = dw = db = None
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        if ctx.needs_input_grad[0] and need_w:            # both gradients: one grouped launch
            dx, dw = torch.empty_like(x), torch.empty_like(w)
            db = torch.empty(N, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            nws = lib.dccn_dense_bwd_w_workspace_size(M, K, N)
            ws = workspace(nws, x.device)
            check(lib.dccn_dense_bwd(_p(x), _p(dy), _p(w), _p(dx), _p(dw), _p(db), M, K, N, _p(ws), nws, _stream()),
                  "dccn_dense_bwd")
            return dx, dw, db
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(lib.dccn_dense_bwd_x(_p(dy), _p(w), _p(dx), M, K, N, _stream()), "dccn_dense_bwd_x")
        if need_w:
            dw = torch.empty_like(w)
            db = torch.empty(N, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            nws = lib.dccn_dense_bwd_w_workspace_size(M, K, N)
            ws = workspace(nws, x.device)
            check(lib.dccn_dense_bwd_w(_p(x), _p(dy), _p(dw), _p(db), M, K, N, _p(ws), nws, _stream()),
                  "dccn_dense_bwd_w")
        return dx, dw, db


def dense(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """tf.layers.dense on the last axis (model.py:1268-1275): x[..., K] . w[K,N] + bias."""
    lead = x.shape[:-1]
    y = _Dense.apply(x.reshape(-1, x.shape[-1]).contiguous(), w.contiguous(),
                     None if bias is None else bias.contiguous())
    return y.view(*lead, w.shape[1])


# ---- R3-R6 --------------------------------------------------------------------------------
def _new_metrics(device) -> torch.Tensor:
    return torch.zeros(_lib.METRICS_BYTES, dtype=torch.uint8, device=device)


def read_metrics(buf: torch.Tensor) -> dict:
    """Device dccn_metrics -> dict(ce_mean, conf[2][2], berlin, log_ber, ce_sum, count).  Synchronises."""
    raw = buf.cpu().numpy().tobytes()
    m = Metrics.from_buffer_copy(raw)
    return dict(ce_sum=m.ce_sum, conf=[[m.conf[0], m.conf[1]], [m.conf[2], m.conf[3]]], count=m.count,
                ce_mean=m.ce_mean, berlin=m.berlin, log_ber=m.log_ber)


class _TailLoss(torch.autograd.Function):
    """ce_mean of the demodulation tail; forward already produces dz / dtailp (fused kernel)."""

    @staticmethod
    def forward(ctx, z, tailp, bits, nbits, prob_out, metrics_buf):
        _need_cuda(z, tailp, bits); _f32(z, tailp)
        lib = _lib.load()
        cells = z.numel() // 2
        nws = lib.dccn_demod_tail_workspace_size(cells, nbits)
        ws = workspace(nws, z.device)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        if need:
            dz = torch.empty_like(z)
            dt = torch.empty_like(tailp)
            check(lib.dccn_demod_tail_loss_fwd_bwd(_p(z), _p(bits), _p(tailp), _p(prob_out), _p(metrics_buf), _p(dz),
                                                   _p(dt), cells, nbits, _p(ws), nws, _stream()),
                  "dccn_demod_tail_loss_fwd_bwd")
            ctx.save_for_backward(dz, dt)
        else:
            check(lib.dccn_demod_tail_loss_fwd(_p(z), _p(bits), _p(tailp), _p(prob_out), _p(metrics_buf), cells,
                                               nbits, _p(ws), nws, _stream()), "dccn_demod_tail_loss_fwd")
        # ce_mean lives at a fixed offset inside dccn_metrics (after ce_sum, conf[4], count)
        ce = metrics_buf[48:52].view(torch.float32).clone().reshape(())
        return ce

    @staticmethod
    def backward(ctx, g):
        dz, dt = ctx.saved_tensors
        return dz * g, dt * g, None, None, None, None


def demod_tail_loss(z: torch.Tensor, tailp: torch.Tensor, bits: torch.Tensor, nbits: int,
                    want_prob: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor]:
    """model.py:1278-1291 + ofdmreceiver_np.py:154-169.

    z [..., D, 2] (dense output per data cell), tailp packed tail weights, bits int32
    [..., D, nbits].  Returns (ce_mean scalar tensor with grad, prob [...,D,nbits,2] or None,
    metrics buffer -> :func:`read_metrics`)."""
    if bits.dtype != torch.int32:
        raise TypeError("bits must be int32")
    if nbits < 1 or nbits > 4:
        raise ValueError("nbits must be in 1..4")
    cells = z.numel() // 2
    if bits.numel() != cells * nbits or tailp.numel() != tail_param_count(nbits):
        raise ValueError("shape mismatch between z, bits and tail params")
    z = z.contiguous()
    prob = torch.empty(*z.shape[:-1], nbits, 2, dtype=torch.float32, device=z.device) if want_prob else None
    mbuf = _new_metrics(z.device)
    ce = _TailLoss.apply(z, tailp.contiguous(), bits.contiguous(), nbits, prob, mbuf)
    return ce, prob, mbuf


class _DenseTailLoss(torch.autograd.Function):
    """ce_mean of dense -> demodulation tail in ONE launch (dccn_dense_tail_fwd_bwd): the tail runs in the dense
    GEMM's epilogue, dz never leaves the kernel un-consumed; the backward finishes with the grouped dX + dW launch."""

    @staticmethod
    def forward(ctx, x, w, bias, tailp, bits, nbits, prob_out, metrics_buf):
        _need_cuda(x, w, bias, tailp, bits); _f32(x, w, bias, tailp)
        lib = _lib.load()
        M, K = x.shape
        N = w.shape[1]
        nws = lib.dccn_dense_tail_workspace_size(M, N, nbits)
        ws = workspace(nws, x.device)
        need = any(ctx.needs_input_grad[:4])
        if need:
            dz = torch.empty(M, N, dtype=torch.float32, device=x.device)
            dt = torch.empty_like(tailp)
            check(lib.dccn_dense_tail_fwd_bwd(_p(x), _p(w), _p(bias), None, _p(bits), _p(tailp), _p(prob_out),
                                              _p(metrics_buf), _p(dz), _p(dt), M, K, N, nbits, _p(ws), nws, _stream()),
                  "dccn_dense_tail_fwd_bwd")
            ctx.save_for_backward(x, w, dz, dt)
            ctx.has_bias = bias is not None
        else:
            check(lib.dccn_dense_tail_fwd(_p(x), _p(w), _p(bias), None, _p(bits), _p(tailp), _p(prob_out),
                                          _p(metrics_buf), M, K, N, nbits, _p(ws), nws, _stream()), "dccn_dense_tail_fwd")
        return metrics_buf[48:52].view(torch.float32).clone().reshape(())

    @staticmethod
    def backward(ctx, g):
        x, w, dz, dt = ctx.saved_tensors
        lib = _lib.load()
        M, K = x.shape
        N = w.shape[1]
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        db = torch.empty(N, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        nws = lib.dccn_dense_bwd_w_workspace_size(M, K, N)
        ws = workspace(nws, x.device)
        check(lib.dccn_dense_bwd(_p(x), _p(dz), _p(w), _p(dx), _p(dw), _p(db), M, K, N, _p(ws), nws, _stream()),
              "dccn_dense_bwd")
        return dx * g, dw * g, (db * g if db is not None else None), dt * g, None, None, None, None


def dense_tail_supported(x: torch.Tensor, w: torch.Tensor, nbits: int) -> bool:
    """The fused launch needs vector-legal operands and a small enough layer (else use dense() + demod_tail_loss())."""
    M, K = x.shape[0], x.shape[-1]
    N = w.shape[1]
    # the shape rule is the library's own (dccn_dense_tail_supported); the pointers are checked here as the launch does
    return bool(_lib.load().dccn_dense_tail_supported(int(M), int(K), int(N), int(nbits))
                and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)


def dense_demod_tail_loss(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], tailp: torch.Tensor,
                          bits: torch.Tensor, nbits: int, want_prob: bool = True):
    """model.py:1268-1291 + ofdmreceiver_np.py:154-169 as one fused operator: x [M,K] . w [K,2D] + bias, then the
    demodulation tail and loss per data cell.  Returns (ce_mean with grad, prob [M,D,nbits,2] or None, metrics)."""
    if bits.dtype != torch.int32:
        raise TypeError("bits must be int32")
    x = x.reshape(-1, x.shape[-1]).contiguous()
    w = w.contiguous()
    M, N = x.shape[0], w.shape[1]
    if bits.numel() != M * (N // 2) * nbits or tailp.numel() != tail_param_count(nbits):
        raise ValueError("shape mismatch between x, w, bits and tail params")
    if not dense_tail_supported(x, w, nbits):
        raise ValueError("fused dense+tail needs 16-byte aligned operands with K, N multiples of 4 (dccn_dense_tail_supported)")
    prob = torch.empty(M, N // 2, nbits, 2, dtype=torch.float32, device=x.device) if want_prob else None
    mbuf = _new_metrics(x.device)
    ce = _DenseTailLoss.apply(x, w, None if bias is None else bias.contiguous(), tailp.contiguous(), bits.contiguous(),
                              nbits, prob, mbuf)
    return ce, prob, mbuf


def demod_tail_eval(z, tailp, bits, nbits, want_prob=True):
    with torch.no_grad():
        return demod_tail_loss(z, tailp, bits, nbits, want_prob)


# ---- equaliser stage (model.py:349-478) --------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        _need_cuda(x); _f32(x)
        lib = _lib.load()
        rows, cols = x.shape[0], x.numel() // x.shape[0]
        y = torch.empty_like(x)
        inv = torch.empty(rows, dtype=torch.float32, device=x.device)
        check(lib.dccn_layer_norm_fwd(_p(x), _p(y), None, _p(inv), rows, cols, eps, _stream()), "dccn_layer_norm_fwd")
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        lib = _lib.load()
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        rows, cols = y.shape[0], y.numel() // y.shape[0]
        check(lib.dccn_layer_norm_bwd(_p(dy), _p(y), _p(inv), _p(dx), rows, cols, _stream()), "dccn_layer_norm_bwd")
        return dx, None


def layer_norm(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """tf.contrib.layers.layer_norm(x, center=False, scale=False, begin_norm_axis=1) (model.py:363):
    per-sample moments over every non-batch axis."""
    if x.dim() < 2:
        raise TypeError("layer_norm: expected [batch, ...]")
    return _LayerNorm.apply(x.contiguous(), float(eps))


class _Tanh(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x); _f32(x)
        y = torch.empty_like(x)
        check(_lib.load().dccn_tanh_fwd(_p(x), _p(y), x.numel(), _stream()), "dccn_tanh_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        check(_lib.load().dccn_tanh_bwd(_p(dy), _p(y), _p(dx), y.numel(), _stream()), "dccn_tanh_bwd")
        return dx


def tanh(x: torch.Tensor) -> torch.Tensor:
    """tf.nn.tanh (activation of the channel-estimate dense layer, model.py:421-426)."""
    return _Tanh.apply(x.contiguous())


class _Equalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, h):
        _need_cuda(y, h); _f32(y, h)
        eq, corr = torch.empty_like(y), torch.empty_like(y)
        check(_lib.load().dccn_equalize_fwd(_p(y), _p(h), _p(eq), _p(corr), y.numel() // 2, _stream()),
              "dccn_equalize_fwd")
        ctx.save_for_backward(y, h)
        return eq, corr

    @staticmethod
    def backward(ctx, d_eq, d_corr):
        y, h = ctx.saved_tensors
        d_eq = None if d_eq is None else d_eq.contiguous()
        d_corr = None if d_corr is None else d_corr.contiguous()
        dy = torch.empty_like(y) if ctx.needs_input_grad[0] else None
        dh = torch.empty_like(h) if ctx.needs_input_grad[1] else None
        if dy is None and dh is None:
            return None, None
        check(_lib.load().dccn_equalize_bwd(_p(y), _p(h), _p(d_eq), _p(d_corr), _p(dy), _p(dh), y.numel() // 2,
                                            _stream()), "dccn_equalize_bwd")
        return dy, dh


def equalize(y: torch.Tensor, h: torch.Tensor):
    """model.py:431-438 on IQ-pair tensors [..., 2]: (eq, corr) = (y*conj(h)/|h|, eq*conj(eq))."""
    if y.shape != h.shape or y.shape[-1] != 2:
        raise TypeError("equalize: y and h must both be [..., 2]")
    return _Equalize.apply(y.contiguous(), h.contiguous())


def pilot_snr(eq_freq: torch.Tensor, pilot_carriers) -> torch.Tensor:
    """model.py:465-475 monitor: eq_freq [B,S,K,2] -> log10(clip(mean/var of |pilot|^2)) [B,1]."""
    _need_cuda(eq_freq); _f32(eq_freq)
    B, S, K, _ = eq_freq.shape
    car = torch.as_tensor(np.asarray(pilot_carriers, dtype=np.int32), device=eq_freq.device)
    out = torch.empty(B, 1, dtype=torch.float32, device=eq_freq.device)
    eqc = eq_freq.detach().contiguous()
    check(_lib.load().dccn_pilot_snr(_p(eqc), _p(car), _p(out), B, S, K, int(car.numel()), _stream()),
          "dccn_pilot_snr")
    return out


class _SameConvExpand(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, bias, L, W):
        _need_cuda(w, bias); _f32(w, bias)
        kL, kW, _ = w.shape
        n = L * W * 2
        T = torch.empty(n, n, dtype=torch.float32, device=w.device)
        be = torch.empty(n, dtype=torch.float32, device=w.device)
        check(_lib.load().dccn_cconv2d_same_expand(_p(w), _p(bias), _p(T), _p(be), L, W, kL, kW, _stream()),
              "dccn_cconv2d_same_expand")
        ctx.geom = (L, W, kL, kW, bias is not None)
        return T, be

    @staticmethod
    def backward(ctx, dT, dbe):
        L, W, kL, kW, has_bias = ctx.geom
        dT = dT.contiguous()
        dbe = None if dbe is None else dbe.contiguous()
        dw = torch.empty(kL, kW, 2, dtype=torch.float32, device=dT.device)
        db = torch.empty(2, dtype=torch.float32, device=dT.device) if has_bias else None
        check(_lib.load().dccn_cconv2d_same_reduce(_p(dT), _p(dbe), _p(dw), _p(db), L, W, kL, kW, _stream()),
              "dccn_cconv2d_same_reduce")
        return dw, db, None, None


def cconv2d_same(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """One-channel, one-filter ``layers_conv2d_complex(x, 1, (kL,kW), padding='same')`` (model.py:428):
    x [B,L,W,2], w [kL,kW,2] (= TF kernel [kL,kW,1,1,2]), bias [2] -> [B,L,W,2].  The kernel is expanded
    into the block-Toeplitz matrix of the equivalent dense layer and run on the MFMA dense GEMM."""
    if x.dim() != 4 or x.shape[-1] != 2 or w.dim() != 3 or w.shape[-1] != 2:
        raise TypeError("cconv2d_same: x [B,L,W,2], w [kL,kW,2]")
    B, L, W, _ = x.shape
    T, be = _SameConvExpand.apply(w.contiguous(), None if bias is None else bias.contiguous(), L, W)
    return dense(x.reshape(B, L * W * 2), T, be).view(B, L, W, 2)
