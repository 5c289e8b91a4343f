"""Complex-valued layer API on real tensors with a trailing IQ axis -- the host-side mirror of
dev/py/complex.py (same function names, argument order, shape/dtype rules and error types).

The reference builds each layer from a TF conv op with 2F filters and then combines four real
sub-convolutions (complex.py:185-188).  Here every layer lowers to ONE fused kernel call:
im2col over the *live* taps (taps that can ever meet data under the TF padding rule) followed by
``ops.cconv_gemm`` -- a single fp32 MFMA GEMM [rows, 2*kin] x [2*kin, 2F] whose weight tile is
expanded from ``[Wa|Wb]`` on the fly and whose epilogue applies the ``(ba-bb, bb-ba)`` bias pair.

TF creates variables implicitly inside ``tf.layers``; the equivalent here is a
:class:`VariableStore` passed as ``scope=``: it hands out parameters under TF's auto-generated
names (``conv3d/kernel``, ``conv3d_1/bias`` ...), so checkpoints keep the reference's variable names.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops

__all__ = ["VariableStore", "complex_clip", "nn_conv1d_complex", "layers_conv1d_complex",
           "layers_conv2d_complex", "tf_padding"]


# --------------------------------------------------------------------------------------------
# variables
# --------------------------------------------------------------------------------------------
class VariableStore(torch.nn.Module):
    """tf.variable_scope + tf.layers auto-naming for the layer functions below.

    ``with store.scope('fft_like') as sc: y = layers_conv2d_complex(x, F, (1, K), scope=sc)``
    creates (first call) or reuses ``fft_like/conv3d/kernel`` and ``fft_like/conv3d/bias``.
    Kernels are glorot-uniform over the FULL TensorFlow kernel shape (fans include dead taps,
    SURVEY.md Appendix A.7), biases zero -- the tf.layers defaults.
    """

    def __init__(self, seed: int = 1, device="cuda"):
        super().__init__()
        self._vars = torch.nn.ParameterDict()
        self.meta: Dict[str, dict] = {}
        self._rng = np.random.RandomState(seed)
        self._device = torch.device(device)
        self._prefix: List[str] = []
        self._counters: List[Dict[str, int]] = [{}]

    @staticmethod
    def _key(name: str) -> str:
        return name.replace("/", "|").replace(".", "_")

    # -- scoping ------------------------------------------------------------------------------
    def scope(self, name: str):
        store = self

        class _Ctx:
            def __enter__(self_inner):
                store._prefix.append(name)
                store._counters.append({})
                return store

            def __exit__(self_inner, *exc):
                store._prefix.pop()
                store._counters.pop()
                return False
        return _Ctx()

    def begin(self):
        """Call at the start of every forward pass: resets the per-scope layer counters."""
        self._prefix, self._counters = [], [{}]

    def layer_name(self, base: str) -> str:
        c = self._counters[-1]
        n = c.get(base, 0)
        c[base] = n + 1
        local = base if n == 0 else "%s_%d" % (base, n)
        return "/".join(self._prefix + [local])

    # -- variables ------------------------------------------------------------------------------
    def get(self, name: str, shape, *, fan_in: Optional[int] = None, fan_out: Optional[int] = None,
            zeros: bool = False, meta: Optional[dict] = None) -> torch.nn.Parameter:
        k = self._key(name)
        if k in self._vars:
            p = self._vars[k]
            if tuple(p.shape) != tuple(shape):
                raise ValueError("variable %s exists with shape %s, requested %s" % (name, tuple(p.shape), tuple(shape)))
            return p
        if zeros:
            val = np.zeros(shape, np.float32)
        else:
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            val = self._rng.uniform(-lim, lim, size=shape).astype(np.float32)
        p = torch.nn.Parameter(torch.from_numpy(val).to(self._device))
        self._vars[k] = p
        self.meta[name] = dict(meta or {}, shape=tuple(shape))
        return p

    def names(self):
        return list(self.meta)

    def tensor(self, name: str) -> torch.nn.Parameter:
        return self._vars[self._key(name)]

    def set(self, name: str, value):
        with torch.no_grad():
            self.tensor(name).copy_(torch.as_tensor(np.asarray(value, dtype=np.float32)).reshape(self.tensor(name).shape))


# --------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------
def tf_padding(in_size: int, k: int, stride: int, padding: str) -> Tuple[int, int, int]:
    """TensorFlow SAME / VALID geometry -> (out_size, pad_before, pad_after)."""
    pad = padding.lower()
    if pad == "same":
        out = -(-in_size // stride)
        total = max((out - 1) * stride + k - in_size, 0)
        return out, total // 2, total - total // 2
    if pad == "valid":
        return -(-(in_size - k + 1) // stride), 0, 0
    raise ValueError("padding must be 'same' or 'valid'")


def _live_taps(in_size: int, k: int, stride: int, out: int, pad_before: int) -> List[int]:
    """Kernel taps that meet real (non-padding) data for at least one output position."""
    return [a for a in range(k) if any(0 <= o * stride + a - pad_before < in_size for o in range(out))]


# general-k convolutions run as an implicit GEMM when the shapes qualify (ops.cconv_patch_supported); False forces the
# im2col + GEMM route (the fallback, and the reference point of tools/convbench.py)
IMPLICIT_GEMM = True


def _split_complex(inputs: torch.Tensor):
    return torch.stack([inputs.real, inputs.imag], dim=-1).to(torch.float32)


# --------------------------------------------------------------------------------------------
# the API
# --------------------------------------------------------------------------------------------
def complex_clip(inputs: torch.Tensor, peak: float = 1.0):
    """dev/py/complex.py:21-27: ``tf.clip_by_norm(inputs, peak, axes=[-1])`` and the mean clipped power."""
    assert inputs.shape[-1] == 2
    return ops.clip_power(inputs.contiguous(), float(peak))


def nn_conv1d_complex(inputs: torch.Tensor, filter: torch.Tensor) -> torch.Tensor:
    """dev/py/complex.py:30-48: canonical complex 1-D convolution, SAME padding.

    inputs [B, L, C, 2], filter [k, C, 1, 2] -> [B, L, 2] (real | imaginary on axis 2).
    Unlike the layers below this one IS the canonical product (re = I*fr - Q*fi, im = I*fi + Q*fr)."""
    assert inputs.shape[-1] == 2 and filter.shape[-1] == 2
    assert inputs.shape[2] == filter.shape[1] and filter.shape[2] == 1
    B, L, C, _ = inputs.shape
    k = filter.shape[0]
    out, p0, _ = tf_padding(L, k, 1, "same")
    taps = list(range(k))
    g = ops.cconv_im2col(inputs.reshape(B, L, 1, C, 2), out, 1, taps, [0], (1, 1), (p0, 0))      # [B*L, k*C, 2]
    rows = g.reshape(B * L, k * C * 2)                                    # interleaved (tap, c, iq)
    fr, fi = filter[..., 0, 0].reshape(k * C), filter[..., 0, 1].reshape(k * C)
    w = torch.stack([torch.stack([fr, fi], dim=-1), torch.stack([-fi, fr], dim=-1)], dim=1)   # [kC, iq_in, 2]
    y = ops.dense(rows, w.reshape(k * C * 2, 2).contiguous(), None)
    return y.view(B, L, 2)


def _cconv_lower(x5: torch.Tensor, filters: int, ksize, strides, padding: str, scope: VariableStore, base: str,
                 tf_kernel_shape) -> torch.Tensor:
    """Shared lowering: x5 real [B, L, Wd, C, 2], taps over (L, Wd) -> [B, L', W', F, 2]."""
    B, L, Wd, C, _ = x5.shape
    (kL, kW), (sL, sW) = ksize, strides
    Lo, pl0, _ = tf_padding(L, kL, sL, padding)
    Wo, pw0, _ = tf_padding(Wd, kW, sW, padding)
    tl, tw = _live_taps(L, kL, sL, Lo, pl0), _live_taps(Wd, kW, sW, Wo, pw0)
    name = scope.layer_name(base)
    kvol = kL * kW
    kern = scope.get(name + "/kernel", (len(tl), len(tw), C, 2 * filters), fan_in=kvol * C, fan_out=kvol * 2 * filters,
                     meta=dict(tf_shape=tuple(tf_kernel_shape), live_taps=(tuple(tl), tuple(tw))))
    bias = scope.get(name + "/bias", (2 * filters,), zeros=True)
    if (C == 1 and filters == 1 and (sL, sW) == (1, 1) and padding.lower() == "same" and len(tl) == kL
            and len(tw) == kW and kvol >= 64 and L * Wd * 2 <= 4096):
        # big one-channel 'same' kernel (equaliser smoothing conv, model.py:428): im2col would blow the
        # image up kL*kW-fold, so run it as a block-Toeplitz dense layer instead
        return ops.cconv2d_same(x5[:, :, :, 0, :], kern.view(kL, kW, 2), bias).view(B, L, Wd, 1, 2)
    if len(tl) == 1 and len(tw) == 1 and (sL, sW) == (1, 1) and Lo == L and Wo == Wd and tl[0] == pl0 and tw[0] == pw0:
        rows = x5.reshape(B * L * Wd, C, 2)                               # one live tap over the input itself: no gather
    elif IMPLICIT_GEMM and ops.cconv_patch_supported(x5, Lo, Wo, len(tl), len(tw), filters):
        # implicit GEMM: the operand loader gathers the taps, the k-inflated patch tensor never exists in the forward
        out = ops.cconv_patch(x5, kern.reshape(-1, 2 * filters), bias, Lo, Wo, tl, tw, (sL, sW), (pl0, pw0))
        return out.view(B, Lo, Wo, filters, 2)
    else:
        rows = ops.cconv_im2col(x5, Lo, Wo, tl, tw, (sL, sW), (pl0, pw0))  # HIP patch gather [B*Lo*Wo, tl*tw*C, 2]
    out = ops.cconv_gemm(rows, kern.reshape(-1, 2 * filters), bias)
    return out.view(B, Lo, Wo, filters, 2)


def layers_conv1d_complex(inputs: torch.Tensor, filters: int, kernal, strides=1, padding="valid", *,
                          scope: VariableStore) -> torch.Tensor:
    """dev/py/complex.py:51-92.  inputs: real [batch, size, channel, 2] or complex64 [batch, size, channel];
    returns [batch, size', filters, 2] (or complex64 [batch, size', filters])."""
    assert type(kernal) == int
    complex_flag = False
    if inputs.dim() == 3 and inputs.dtype == torch.complex64:
        inputs = _split_complex(inputs)
        complex_flag = True
    elif inputs.dim() == 4 and inputs.shape[-1] == 2:
        pass
    else:
        raise NameError("Check input tensor dtypes or shape")
    B, L, C, _ = inputs.shape
    x5 = inputs.reshape(B, L, 1, C, 2)
    s = int(strides)
    out = _cconv_lower(x5, filters, (kernal, 1), (s, 1), padding, scope, "conv2d", (kernal, 1, C, 2 * filters))
    out = out.reshape(B, out.shape[1], filters, 2)
    if complex_flag:
        out = torch.complex(out[..., 0], out[..., 1])
    return out


def layers_conv2d_complex(inputs: torch.Tensor, filters: int, kernal, strides=1, padding="valid", *,
                          scope: VariableStore) -> torch.Tensor:
    """dev/py/complex.py:140-196.  inputs: real [batch, length, width, channel, 2] or complex64
    [batch, length, width, channel]; ``kernal`` int or (kL, kW); returns [batch, L', W', filters, 2]
    (or complex64).  re = I*Wa - Q*Wb + (ba-bb), im = I*Wb - Q*Wa + (bb-ba) -- as the reference computes it."""
    assert (isinstance(kernal, int) or len(kernal) < 3)
    complex_flag = False
    if inputs.dim() == 4 and inputs.dtype == torch.complex64:
        inputs = _split_complex(inputs)
        complex_flag = True
    elif inputs.dim() == 5 and inputs.shape[-1] == 2:
        pass
    else:
        raise TypeError("Check input tensor dtypes or shape")
    if type(kernal) is int:
        ksize = (kernal, kernal)
    elif type(kernal) is tuple and len(kernal) == 2:
        ksize = kernal
    else:
        raise NameError("Unacceptable Kernal Size")
    if type(strides) is int:
        st = (strides, strides)
    elif type(strides) is tuple and len(strides) == 2:
        st = strides
    else:
        raise NameError("Unacceptable Kernal Size")
    C = inputs.shape[3]
    out = _cconv_lower(inputs, filters, ksize, st, padding, scope, "conv3d", (ksize[0], ksize[1], 1, C, 2 * filters))
    if complex_flag:
        out = torch.complex(out[..., 0], out[..., 1])
    return out
