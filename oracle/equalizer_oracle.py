"""CPU oracle of the DCCN channel-equaliser stage.

TEST INFRASTRUCTURE ONLY -- same import rule as oracle/dccn_oracle.py: nothing under
dl_ofdm_amd/ may import this file; it is the checker of tests/, never the product.
PARITY STATUS: parity unpinned against TensorFlow numerics (TF 1.15 is not available in this
image; the reference holds no golden tensors for this stage).  What pins it instead: two
independent formulations that must agree -- the NumPy restatement below (literal, tap-by-tap
conv3d) and the torch-CPU autograd graph in oracle/torch_ref.py (``LiteralEqualizer``, padded
``conv3d``), forward in fp64 to 1e-12 and, through autograd, the gradients the HIP path is
compared with.

Restates ``equalizer_ofdm`` (dev/py/model.py:349-478) and the transfer-learning loss of
dev/py/ofdmreceiver_np_mp.py:283-347: the equaliser sits between ``input:0`` (the batch-moment
normalised IQ, ofdmreceiver_np.py:128-129,137) and the frozen basic receiver; the loss is
``ce_mean + 1e-3 * sum(regularization_losses)`` (:319-324) and only ``Equalizer/*`` variables are
trained (:330).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Tuple

import numpy as np

from . import dccn_oracle as O

LN_EPS = 1e-12          # tf.contrib.layers.layer_norm variance_epsilon
EQ_REG_COEFF = 1e-3     # ofdmreceiver_np_mp.py:321


@dataclass
class EqConfig:
    """Shapes of ``equalizer_ofdm`` (model.py:349-366)."""
    S: int = 7                # nsymbol
    K: int = 64               # nfft
    CP: int = 16
    cp: bool = True           # FLAGS.cp: the input still carries the cyclic prefix
    pilot_size: int = 16      # ofdmobj.pilot_size (pilot cells per frame)
    pilot_carriers: Tuple[int, ...] = field(default_factory=tuple)   # ofdmobj.pilotCarriers

    @property
    def n_sc(self):
        """width of `input:0` -- the CP samples are there whether or not FLAGS.cp uses them (:362,:366)"""
        return self.K + self.CP


# TF auto-naming inside ``with tf.variable_scope('Equalizer')`` (ofdmreceiver_np_mp.py:283-286),
# in creation order.
def param_shapes(c: EqConfig) -> Dict[str, Tuple[int, ...]]:
    SK2 = c.S * c.K * 2
    kin0 = (c.n_sc if c.cp else c.K) * 2
    return {
        "Equalizer/dense/kernel": (kin0, 2 * c.K), "Equalizer/dense/bias": (2 * c.K,),            # :371-376
        "Equalizer/conv3d/kernel": (1, c.K, 1, 1, 2 * c.K), "Equalizer/conv3d/bias": (2 * c.K,),  # :379
        "Equalizer/dense_1/kernel": (SK2, 2 * c.pilot_size), "Equalizer/dense_1/bias": (2 * c.pilot_size,),  # :394
        "Equalizer/dense_2/kernel": (2 * c.pilot_size, SK2), "Equalizer/dense_2/bias": (SK2,),    # :402
        "Equalizer/dense_3/kernel": (SK2, SK2), "Equalizer/dense_3/bias": (SK2,),                 # :408
        "Equalizer/dense_4/kernel": (SK2, SK2), "Equalizer/dense_4/bias": (SK2,),                 # :421 (tanh)
        "Equalizer/conv3d_1/kernel": (c.S, c.K, 1, 1, 2), "Equalizer/conv3d_1/bias": (2,),        # :428
        "Equalizer/conv3d_2/kernel": (1, c.K, 1, 1, 2 * c.K), "Equalizer/conv3d_2/bias": (2 * c.K,),  # :439 (corr)
        "Equalizer/conv3d_3/kernel": (1, c.K, 1, 1, 2 * c.K), "Equalizer/conv3d_3/bias": (2 * c.K,),  # :443
        "Equalizer/dense_5/kernel": (4 * c.K, 2 * c.n_sc), "Equalizer/dense_5/bias": (2 * c.n_sc,),   # :458
    }


def regularized(c: EqConfig):
    """every tf.layers.dense of the stage carries l2(0.01) on kernel and bias; the convs none."""
    return tuple(n for n in param_shapes(c) if "/dense" in n)


def init_params(c: EqConfig, seed: int = 3, dtype=np.float32, bias_scale: float = 0.0) -> Dict[str, np.ndarray]:
    """glorot-uniform kernels (fans as tf.layers computes them: receptive field x channels),
    zero biases; ``bias_scale`` > 0 draws non-zero biases so that tests exercise the bias paths."""
    rng = np.random.RandomState(seed)
    p = {}
    for name, shp in param_shapes(c).items():
        if name.endswith("bias"):
            p[name] = (bias_scale * rng.standard_normal(shp)).astype(dtype)
        else:
            if len(shp) == 2:
                fi, fo = shp
            else:
                rf = shp[0] * shp[1] * shp[2]
                fi, fo = rf * shp[3], rf * shp[4]
            lim = math.sqrt(6.0 / (fi + fo))
            p[name] = rng.uniform(-lim, lim, size=shp).astype(dtype)
    return p


def layer_norm(x: np.ndarray, eps: float = LN_EPS) -> np.ndarray:
    """tf.contrib.layers.layer_norm(center=False, scale=False, begin_norm_axis=1) (model.py:363):
    per-sample moments over every non-batch axis, then tf.nn.batch_normalization's
    ``x*inv + (-mean*inv)`` with inv = rsqrt(var + 1e-12)."""
    ax = tuple(range(1, x.ndim))
    mean = x.mean(axis=ax, keepdims=True, dtype=x.dtype)
    var = np.mean((x - mean) ** 2, axis=ax, keepdims=True, dtype=x.dtype)
    inv = (1.0 / np.sqrt(var + x.dtype.type(eps))).astype(x.dtype)
    return x * inv + (-mean * inv)


def _cconv(x5, p, name, padding):
    return O.layers_conv2d_complex_literal(x5, p[name + "/kernel"], p[name + "/bias"], (1, 1), padding)


def equalize(y: np.ndarray, h: np.ndarray):
    """model.py:431-437: eq = y * conj(h)/|h| and corr = eq * conj(eq) on [..., 2] IQ arrays."""
    yr, yi, hr, hi = y[..., 0], y[..., 1], h[..., 0], h[..., 1]
    a = np.sqrt(hr * hr + hi * hi)                        # tf.abs
    cr, ci = hr / a, (-hi) / a                            # conj / abs (:433)
    er = yr * cr - yi * ci                                # complex multiply (:435)
    ei = yr * ci + yi * cr
    corr_r = er * er - ei * (-ei)                         # eq * conj(eq) (:438)
    corr_i = er * (-ei) + ei * er
    return np.stack([er, ei], -1), np.stack([corr_r, corr_i], -1)


def pilot_snr(eq_freq: np.ndarray, pilot_carriers) -> np.ndarray:
    """model.py:466-475: log10(clip(mean/var of |pilot|^2 over the frame's S*P pilot cells))."""
    B = eq_freq.shape[0]
    pil = eq_freq[:, :, list(pilot_carriers), :]
    pw = (pil[..., 0] ** 2 + pil[..., 1] ** 2).reshape(B, -1)
    m = pw.mean(axis=1, keepdims=True)
    v = np.mean((pw - m) ** 2, axis=1, keepdims=True)
    ratio = np.clip(m / v, 0.001, 10000.0)
    return (np.log(ratio) / np.log(10.0)).reshape(-1, 1).astype(eq_freq.dtype)


def equalizer_forward(p: Dict[str, np.ndarray], x: np.ndarray, c: EqConfig, keep: bool = False):
    """``equalizer_ofdm`` (model.py:349-478) on ``input:0`` x [B,S,n_sc,2].
    Returns (equalized [B,S,n_sc,2], snr_db [B,1], chest [B,S,K,2] as IQ pairs)."""
    B = x.shape[0]
    S, K = c.S, c.K
    chest = layer_norm(x)                                                              # :363
    if not c.cp:
        chest = chest[:, :, c.CP:c.CP + K, :]                                          # :366
    t0 = chest.reshape(B, S, -1)
    t1 = t0 @ p["Equalizer/dense/kernel"] + p["Equalizer/dense/bias"]                  # :371
    y5 = _cconv(t1.reshape(B, S, K, 1, 2), p, "Equalizer/conv3d", "valid")             # :378-379 [B,S,1,K,2]
    y5 = np.transpose(y5, (0, 1, 3, 2, 4))                                             # :380     [B,S,K,1,2]
    y = y5[:, :, :, 0, :]                                                              # inputs_complex (:383)
    flat = y5.reshape(B, S * K * 2)                                                    # :392
    d1 = flat @ p["Equalizer/dense_1/kernel"] + p["Equalizer/dense_1/bias"]            # :394 pilot
    d2 = d1 @ p["Equalizer/dense_2/kernel"] + p["Equalizer/dense_2/bias"]              # :402
    d3 = d2 @ p["Equalizer/dense_3/kernel"] + p["Equalizer/dense_3/bias"]              # :408
    d4 = np.tanh(d3 @ p["Equalizer/dense_4/kernel"] + p["Equalizer/dense_4/bias"])     # :421
    h5 = _cconv(d4.reshape(B, S, K, 1, 2), p, "Equalizer/conv3d_1", "same")            # :427-428 [B,S,K,1,2]
    h = h5[:, :, :, 0, :]                                                              # chest (:429-430)
    eq, corr = equalize(y, h)                                                          # :432-438
    corr5 = _cconv(corr.reshape(B, S, K, 1, 2), p, "Equalizer/conv3d_2", "valid")      # :439 [B,S,1,K,2]
    corr_re = corr5[:, :, 0, :, :]                                                     # :440-441 [B,S,K,2]
    e5 = _cconv(eq.reshape(B, S, K, 1, 2), p, "Equalizer/conv3d_3", "valid")           # :443
    equalized = e5[:, :, 0, :, :]                                                      # :444-449 [B,S,K,2]
    cat = np.concatenate([equalized, corr_re], axis=-1).reshape(B, S, 4 * K)           # :456-457
    out = cat @ p["Equalizer/dense_5/kernel"] + p["Equalizer/dense_5/bias"]            # :458
    out = out.reshape(B, S, c.n_sc, 2)                                                 # :463
    snr_db = pilot_snr(eq, c.pilot_carriers) if len(c.pilot_carriers) else None        # :465-475
    if keep:
        return out, snr_db, h, dict(ln=t0, t1=t1, y=y, d1=d1, d2=d2, d3=d3, d4=d4, h=h, eq=eq, corr=corr,
                                    corr_re=corr_re, equalized=equalized)
    return out, snr_db, h


def reg_sum(p: Dict[str, np.ndarray], c: EqConfig):
    return sum(O.REG_L2 * np.sum(np.square(p[n].astype(np.float64))) for n in regularized(c))
