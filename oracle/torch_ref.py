"""Literal torch-CPU restatement of the reference's TF1 graph for the basic receiver.

TEST INFRASTRUCTURE ONLY (see oracle/dccn_oracle.py header; same import rule).
PARITY STATUS: parity unpinned against TensorFlow numerics (TF 1.15 is not available).

Purpose
  1. an *independent second formulation* of the hot path: the C-Conv is evaluated the
     way TensorFlow evaluates it -- a zero-padded NDHWC ``conv3d`` with the full
     ``[1, K, 1, K, 2F]`` kernel (dev/py/complex.py:168-192, SAME padding) -- and the
     backward comes from autograd, so it cross-checks both the centre-tap GEMM claim and
     the hand-derived backward of ``dccn_oracle``;
  2. the ``cpu_baseline`` of bench.py ("port": reference-equivalent CPU graph, timed on
     the host cores), in the literal conv3d form and in GEMM form.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as Fnn

from . import dccn_oracle as O


def tf_same_pad(in_size: int, k: int, stride: int = 1):
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    return total // 2, total - total // 2


def conv2d_complex_literal(inputs: torch.Tensor, kernel: torch.Tensor, bias: torch.Tensor,
                           padding: str = "same") -> torch.Tensor:
    """layers_conv2d_complex (complex.py:140-196), real 5-D input [B,L,Wd,C,2], TF kernel
    [kL,kW,1,C,2F], strides 1."""
    B, L, Wd, C, _ = inputs.shape
    kL, kW, _, _, F2 = kernel.shape
    F = F2 // 2
    conv = inputs.permute(0, 1, 2, 4, 3)                      # NDHWC [B,L,Wd,2,C]  (:168)
    x = conv.permute(0, 4, 1, 2, 3)                           # NCDHW for torch
    if padding.lower() == "same":
        pl = tf_same_pad(L, kL)
        pw = tf_same_pad(Wd, kW)
        x = Fnn.pad(x, (0, 0, pw[0], pw[1], pl[0], pl[1]))
    w = kernel.permute(4, 3, 0, 1, 2)                         # [Cout,Cin,kD,kH,kW]
    y = Fnn.conv3d(x, w, bias)                                # [B,2F,L',W',2]
    y = y.permute(0, 2, 3, 4, 1)                              # NDHWC [B,L',W',2,2F]
    Lo, Wo = y.shape[1], y.shape[2]
    y = y.reshape(B, Lo, Wo, 4, F)                            # (:185)
    re = y[:, :, :, 0, :] - y[:, :, :, 3, :]                  # (:187)
    im = y[:, :, :, 1, :] - y[:, :, :, 2, :]                  # (:188)
    return torch.stack([re, im], dim=-1)                      # [B,L',W',F,2]  (:191-192)


def layers_conv2d_complex_literal_t(inputs: torch.Tensor, kernel: torch.Tensor, bias, strides=(1, 1),
                                    padding: str = "valid") -> torch.Tensor:
    """torch twin of ``dccn_oracle.layers_conv2d_complex_literal`` (complex.py:140-196; any strides, SAME / VALID): the same
    tap-by-tap evaluation on torch tensors, so that autograd yields the gradients TensorFlow derives for the layer --
    the reference for the backward of the general-k convolutions.  inputs [B,L,Wd,C,2], kernel [kL,kW,1,C,2F]."""
    B, L, Wd, C, _ = inputs.shape
    kL, kW, _, _, F2 = kernel.shape
    F = F2 // 2
    sL, sW = strides
    Lo, pl0, pl1 = O._tf_pad(L, kL, sL, padding)
    Wo, pw0, pw1 = O._tf_pad(Wd, kW, sW, padding)
    x = inputs.permute(0, 1, 2, 4, 3)                                           # [B,L,Wd,2,C]  (:168)
    x = Fnn.pad(x, (0, 0, 0, 0, pw0, pw1, pl0, pl1))
    conv = torch.zeros(B, Lo, Wo, 2, F2, dtype=inputs.dtype)
    for a in range(kL):
        for b in range(kW):
            patch = x[:, a:a + (Lo - 1) * sL + 1:sL, b:b + (Wo - 1) * sW + 1:sW]
            if patch.shape[1] != Lo or patch.shape[2] != Wo:
                continue
            conv = conv + patch @ kernel[a, b, 0]
    if bias is not None:
        conv = conv + bias
    conv = conv.reshape(B, Lo, Wo, 4, F)                                        # (:185)
    return torch.stack([conv[:, :, :, 0] - conv[:, :, :, 3], conv[:, :, :, 1] - conv[:, :, :, 2]], dim=-1)   # (:187-192)


def layers_conv1d_complex_literal_t(inputs: torch.Tensor, kernel: torch.Tensor, bias, strides: int = 1,
                                    padding: str = "valid") -> torch.Tensor:
    """``layers_conv1d_complex`` (complex.py:51-92) on torch tensors: [B,L,C,2], kernel [k,1,C,2F] -> [B,L',F,2]
    (the 2-D layer over a width-1 second axis: identical arithmetic, :78-85)."""
    y = layers_conv2d_complex_literal_t(inputs[:, :, None], kernel[:, :, None], bias, (strides, 1), padding)
    return y[:, :, 0]


def full_kernel_from_live(w_live: np.ndarray, kin: int, seed: int = 0) -> np.ndarray:
    """Embed the live tap [kin,2F] at (K-1)//2 of a TF-shaped [1,K,1,K,2F] kernel whose
    other (dead) taps hold arbitrary values -- they must not influence anything."""
    rng = np.random.RandomState(seed)
    full = rng.uniform(-0.02, 0.02, size=(1, kin, 1, kin, w_live.shape[1])).astype(w_live.dtype)
    full[0, (kin - 1) // 2, 0] = w_live
    return full


class LiteralRx:
    """ofdmreceiver_np.py:121-189 as a torch-CPU autograd graph."""

    def __init__(self, params: Dict[str, np.ndarray], cfg: O.RxConfig, dtype=torch.float32,
                 literal_conv: bool = True):
        self.cfg = cfg
        self.dtype = dtype
        self.literal_conv = literal_conv
        self.p = {}
        for k, v in params.items():
            if k == "fft_like/conv3d/kernel" and literal_conv:
                v = full_kernel_from_live(v, cfg.kin)
            self.p[k] = torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True)
        self.m = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.b1p, self.b2p, self.step = O.ADAM_BETA1, O.ADAM_BETA2, 0.0

    # -- graph pieces ------------------------------------------------------------------
    def normalise(self, x):
        mean = x.mean(dim=0)
        var = ((x - mean) ** 2).mean(dim=0)
        inv = torch.rsqrt(var + O.NORM_EPS)
        return (x * inv + (-mean * inv)) / math.sqrt(2.0)

    def receiver(self, x_norm):
        c = self.cfg
        Bf = x_norm.shape[0]
        if self.literal_conv:
            conv = x_norm.reshape(Bf, c.S, 1, c.kin, 2)
            out = conv2d_complex_literal(conv, self.p["fft_like/conv3d/kernel"],
                                         self.p["fft_like/conv3d/bias"], "same")
        else:
            w, b = self.p["fft_like/conv3d/kernel"], self.p["fft_like/conv3d/bias"]
            xi, xq = x_norm[..., 0].reshape(-1, c.kin), x_norm[..., 1].reshape(-1, c.kin)
            wa, wb = w[:, :c.F], w[:, c.F:]
            re = xi @ wa - xq @ wb + (b[:c.F] - b[c.F:])
            im = xi @ wb - xq @ wa + (b[c.F:] - b[:c.F])
            out = torch.stack([re, im], dim=-1)
        fft_out = out.reshape(Bf, c.S, c.F, 2)
        a = fft_out.reshape(Bf, c.S * c.F * 2)
        z = a @ self.p["demodulation/dense/kernel"] + self.p["demodulation/dense/bias"]
        iq = z.reshape(Bf, 1, c.D, 2)
        h1 = Fnn.leaky_relu(iq @ self.p["demodulation/conv2d/kernel"]
                            + self.p["demodulation/conv2d/bias"], O.LEAKY_ALPHA)
        cc = torch.cat([h1, iq], dim=-1)
        u = Fnn.leaky_relu(cc @ self.p["demodulation/dense_1/kernel"]
                           + self.p["demodulation/dense_1/bias"], O.LEAKY_ALPHA)
        prob = torch.softmax(u.reshape(Bf, c.D, c.nbits, 2), dim=-1)
        return prob, fft_out, z

    def losses(self, prob, bits):
        pr = prob.reshape(-1, 2)
        y = torch.as_tensor(np.asarray(bits).reshape(-1), dtype=torch.long)
        ce = Fnn.cross_entropy(pr, y, reduction="none")     # softmax applied twice
        ce_mean = ce.mean()
        pred = (pr[:, 1] > pr[:, 0]).long()                  # argmax, ties -> 0
        conf = torch.zeros(2, 2, dtype=torch.long)
        conf.view(-1).index_add_(0, y * 2 + pred, torch.ones_like(y))
        berlin = float(conf[0, 1] + conf[1, 0]) / float(conf.sum())
        reg = sum(O.REG_L2 * (self.p[n] ** 2).sum() for n in O.REGULARIZED)
        cost_grad_part = ce_mean + np.float32(berlin) * O.REG_COEFF * reg
        return ce_mean, conf, berlin, cost_grad_part

    # -- one training step -------------------------------------------------------------
    def forward_backward(self, x_raw: np.ndarray, bits: np.ndarray):
        x = torch.as_tensor(x_raw, dtype=self.dtype)
        for v in self.p.values():
            v.grad = None
        x_norm = self.normalise(x)
        prob, fft_out, z = self.receiver(x_norm)
        ce_mean, conf, berlin, loss = self.losses(prob, bits)
        loss.backward()
        grads = {k: v.grad.detach().numpy().copy() for k, v in self.p.items()}
        return grads, dict(ce_mean=float(ce_mean.detach()), conf=conf.numpy(), berlin=berlin,
                           prob=prob.detach().numpy(), x_norm=x_norm.detach().numpy(),
                           fft_out=fft_out.detach().numpy(), z=z.detach().numpy())

    @torch.no_grad()
    def adam(self):
        lr = O.LR0 * O.LR_DECAY ** math.floor(self.step / O.LR_DECAY_STEPS)
        alpha = lr * math.sqrt(1.0 - self.b2p) / (1.0 - self.b1p)
        for k, v in self.p.items():
            g = v.grad
            self.m[k] += (g - self.m[k]) * (1.0 - O.ADAM_BETA1)
            self.v[k] += (g * g - self.v[k]) * (1.0 - O.ADAM_BETA2)
            v -= (self.m[k] * alpha) / (self.v[k].sqrt() + O.ADAM_EPS)
        self.b1p *= O.ADAM_BETA1
        self.b2p *= O.ADAM_BETA2
        self.step += 1.0

    def train_step(self, x_raw, bits):
        out = self.forward_backward(x_raw, bits)
        self.adam()
        return out

    def live_params(self) -> Dict[str, np.ndarray]:
        c = self.cfg
        out = {}
        for k, v in self.p.items():
            a = v.detach().numpy()
            if k == "fft_like/conv3d/kernel" and self.literal_conv:
                a = a[0, (c.kin - 1) // 2, 0]
            out[k] = a.copy()
        return out


class LiteralEqualizer:
    """``equalizer_ofdm`` (dev/py/model.py:349-478) + the transfer-learning loss of
    dev/py/ofdmreceiver_np_mp.py:283-330 as a torch-CPU autograd graph: batch-moment norm ->
    equaliser (trainable) -> frozen basic receiver -> ce_mean + 1e-3 * sum(reg)."""

    def __init__(self, eq_params: Dict[str, np.ndarray], rx: "LiteralRx", ecfg, dtype=torch.float64):
        from . import equalizer_oracle as E
        self.E, self.c, self.rx, self.dtype = E, ecfg, rx, dtype
        self.p = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in eq_params.items()}

    def _conv(self, x5, name, padding):
        return conv2d_complex_literal(x5, self.p[name + "/kernel"], self.p[name + "/bias"], padding)

    def equalizer(self, x):
        c, p = self.c, self.p
        B, S, K = x.shape[0], c.S, c.K
        mean = x.mean(dim=(1, 2, 3), keepdim=True)
        var = ((x - mean) ** 2).mean(dim=(1, 2, 3), keepdim=True)
        inv = torch.rsqrt(var + self.E.LN_EPS)
        chest = x * inv + (-mean * inv)
        if not c.cp:
            chest = chest[:, :, c.CP:c.CP + K, :]
        t1 = chest.reshape(B, S, -1) @ p["Equalizer/dense/kernel"] + p["Equalizer/dense/bias"]
        y5 = self._conv(t1.reshape(B, S, K, 1, 2), "Equalizer/conv3d", "valid").permute(0, 1, 3, 2, 4)
        y = torch.complex(y5[:, :, :, :, 0].contiguous(), y5[:, :, :, :, 1].contiguous())      # [B,S,K,1]
        d = y5.reshape(B, S * K * 2)
        d = d @ p["Equalizer/dense_1/kernel"] + p["Equalizer/dense_1/bias"]
        d = d @ p["Equalizer/dense_2/kernel"] + p["Equalizer/dense_2/bias"]
        d = d @ p["Equalizer/dense_3/kernel"] + p["Equalizer/dense_3/bias"]
        d = torch.tanh(d @ p["Equalizer/dense_4/kernel"] + p["Equalizer/dense_4/bias"])
        h5 = self._conv(d.reshape(B, S, K, 1, 2), "Equalizer/conv3d_1", "same")
        h = torch.complex(h5[:, :, :, :, 0].contiguous(), h5[:, :, :, :, 1].contiguous())
        hc = torch.conj(h)
        ha = torch.abs(h)
        hc = torch.complex(hc.real / ha, hc.imag / ha)
        eq = y * hc
        corr = eq * torch.conj(eq)
        corr5 = self._conv(torch.view_as_real(corr).reshape(B, S, K, 1, 2), "Equalizer/conv3d_2", "valid")
        corr_re = corr5[:, :, 0, :, :]
        e5 = self._conv(torch.view_as_real(eq).reshape(B, S, K, 1, 2), "Equalizer/conv3d_3", "valid")
        equalized = e5[:, :, 0, :, :]
        cat = torch.cat([equalized, corr_re], dim=-1).reshape(B, S, 4 * K)
        out = cat @ p["Equalizer/dense_5/kernel"] + p["Equalizer/dense_5/bias"]
        return out.reshape(B, S, c.n_sc, 2), torch.view_as_real(h)[:, :, :, 0, :], torch.view_as_real(eq)[:, :, :, 0, :]

    def forward_backward(self, x_raw: np.ndarray, bits: np.ndarray):
        x = torch.as_tensor(x_raw, dtype=self.dtype)
        for v in self.p.values():
            v.grad = None
        x_norm = self.rx.normalise(x)
        out_eq, h, eq = self.equalizer(x_norm)
        out_eq.retain_grad()
        prob, fft_out, z = self.rx.receiver(out_eq)
        ce_mean, conf, berlin, _ = self.rx.losses(prob, bits)
        reg = sum(O.REG_L2 * (self.p[n] ** 2).sum() for n in self.E.regularized(self.c))
        reg = reg + sum(O.REG_L2 * (self.rx.p[n] ** 2).sum() for n in O.REGULARIZED)
        loss = ce_mean + self.E.EQ_REG_COEFF * reg
        loss.backward()
        grads = {k: v.grad.detach().numpy().copy() for k, v in self.p.items()}
        return grads, dict(ce_mean=float(ce_mean.detach()), conf=conf.numpy(), berlin=berlin, loss=float(loss.detach()),
                           out_eq=out_eq.detach().numpy(), d_out_eq=out_eq.grad.numpy().copy(),
                           chest=h.detach().numpy(), eq=eq.detach().numpy(), prob=prob.detach().numpy(),
                           x_norm=x_norm.detach().numpy())
