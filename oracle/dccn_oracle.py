"""CPU oracle for the DCCN receiver hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product package
``dl_ofdm_amd`` never does (its ops fail loudly when the HIP library is missing).

PARITY STATUS: **pinned to TensorFlow's graph; unpinned only in TensorFlow's fp32 kernel rounding.**
The reference runs its arithmetic in TensorFlow 1.15, which is neither vendored in /root/reference nor
installable here (no network, Python 3.10), so no tensor TensorFlow ever computed is available.  What the
reference does ship is TensorFlow's complete *graph* of the v1 receiver, eight ``test_v1/model/*.meta``
files: forward ops, the loss/BER assembly, the ``gradients/...`` subgraph of TF's autodiff and the Adam
ops, with every constant.  They are committed as node lists (tests/golden/v1_graph/*.json.gz, made by
tests/golden/make_graph_golden.py) and executed op by op in NumPy float64 by oracle/tf_graph.py:
  * tests/test_graph_golden.py: the functions below -- R0 normalisation, C-Conv GEMM form and its
    hand-derived backward, dense, demodulation tail + double-softmax CE and their hand-derived backward,
    confusion/BER, cost, clip/power -- composed in the v1 topology reproduce every fetched tensor and
    every gradient feeding ``ApplyAdam`` of all eight graphs (nbits 1-4, cp on/off) to 1e-11, and every
    constant hard-coded here equals the graph's;
  * tests/test_gpu_graph_golden.py runs the HIP operators against the same graph evaluation (1e-5).
What is NOT in those graphs and therefore rests on the cited source lines alone: the dev-version details
that differ from v1 (7 symbols and the LTE pilot grid, one 1x1 conv instead of two, keras l2(0.01)
instead of contrib l2_regularizer), the inside of the ``ApplyAdam`` kernel (TF 1.15 training_ops.cc
formula, Appendix A.6) and the equaliser stage (oracle/equalizer_oracle.py, no archived graph).
Also still in place:
  * ``oracle/torch_ref.py`` re-derives the same graph in the *literal* TF form
    (zero-padded NDHWC conv3d, autograd backward) and must agree with the GEMM-form
    forward and the hand-derived backward below (tests/test_oracle.py);
  * structural fixtures parsed from the reference's own checkpoints
    (tests/golden/v1_index_manifest.json) pin variable names / shapes;
  * the reference's importable NumPy substrate (ofdm.py / radio.py / util.py) pins the
    input tensors and label layout through tests/golden/*.npz.

All functions are pure NumPy and work in the dtype of their inputs (float32 to
mirror the reference graph, float64 to serve as a tight "truth" in tolerance tests).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np

LEAKY_ALPHA = 0.2          # tf.nn.leaky_relu default, model.py:1280,1287
NORM_EPS = 1e-9            # ofdmreceiver_np.py:129
CLIP_PEAK = 8.0            # ofdmreceiver_np.py:131
REG_L2 = 0.01              # tf.keras.regularizers.l2(l=0.01), model.py:1271-1272,1285-1286
REG_COEFF = 1e-4           # ofdmreceiver_np.py:162
BER_COEFF = 1.0            # ofdmreceiver_np.py:163
ADAM_BETA1, ADAM_BETA2, ADAM_EPS = 0.9, 0.999, 1e-8   # tf.train.AdamOptimizer defaults
LR0, LR_DECAY_STEPS, LR_DECAY = 1e-3, 500.0, 0.98      # ofdmreceiver_np.py:186-187


# --------------------------------------------------------------------------------------
# R0  input normalisation   (dev/py/ofdmreceiver_np.py:128-129)
# --------------------------------------------------------------------------------------
def batch_moment_norm(x: np.ndarray, eps: float = NORM_EPS):
    """``tf.nn.moments(x,[0])`` + ``tf.nn.batch_normalization(..., eps)/np.sqrt(2)``.

    moments = mean over the batch axis and the *biased* variance
    mean((x-mean)^2); batch_normalization with offset=scale=None evaluates
    ``x*inv + (-mean*inv)`` with ``inv = rsqrt(var+eps)`` (TF 1.15 nn_impl.py).
    Returns (y, mean, var).
    """
    dt = x.dtype
    mean = x.mean(axis=0, dtype=dt)
    var = np.square(x - mean).mean(axis=0, dtype=dt)
    inv = (1.0 / np.sqrt(var + dt.type(eps))).astype(dt)
    y = x * inv + (-mean * inv)
    y = y / dt.type(math.sqrt(2.0))
    return y.astype(dt), mean, var


# --------------------------------------------------------------------------------------
# R8  complex_clip   (dev/py/complex.py:21-27)
# --------------------------------------------------------------------------------------
def complex_clip(x: np.ndarray, peak: float = 1.0):
    """``tf.clip_by_norm(x, peak, axes=[-1])`` and the mean power of the clipped IQ.

    clip_by_norm: x * peak / max(||x||_2, peak) over the IQ axis.
    """
    assert x.shape[-1] == 2
    dt = x.dtype
    l2 = np.sqrt(np.sum(x * x, axis=-1, keepdims=True))
    clipped = (x * dt.type(peak) / np.maximum(l2, dt.type(peak))).astype(dt)
    power = np.mean(np.square(clipped[..., 0]) + np.square(clipped[..., 1]), dtype=dt)
    return clipped, power


# --------------------------------------------------------------------------------------
# R1  C-Conv   (dev/py/complex.py:140-196)  -- GEMM form of the receiver use
# --------------------------------------------------------------------------------------
def cconv_gemm_fwd(x: np.ndarray, w: np.ndarray, bias: Optional[np.ndarray]):
    """x [rows, kin, 2], w [kin, 2F] (= the live tap ``kernel[0,(K-1)//2,0,:,:]``),
    bias [2F] -> out [rows, F, 2].

    complex.py:185-188: with Wa = w[:, :F], Wb = w[:, F:],
        re = I.Wa - Q.Wb + (ba - bb),   im = I.Wb - Q.Wa + (bb - ba)
    (the imaginary part is *not* the canonical complex product; preserved as written).
    """
    F = w.shape[1] // 2
    xi, xq = x[..., 0], x[..., 1]
    wa, wb = w[:, :F], w[:, F:]
    re = xi @ wa - xq @ wb
    im = xi @ wb - xq @ wa
    if bias is not None:
        ba, bb = bias[:F], bias[F:]
        re = re + (ba - bb)
        im = im + (bb - ba)
    return np.stack([re, im], axis=-1).astype(x.dtype)


def cconv_gemm_bwd(x: np.ndarray, w: np.ndarray, dout: np.ndarray):
    """Hand-derived backward of :func:`cconv_gemm_fwd`.

    Returns (dx [rows,kin,2], dw [kin,2F], dbias [2F]).
    """
    F = w.shape[1] // 2
    xi, xq = x[..., 0], x[..., 1]
    wa, wb = w[:, :F], w[:, F:]
    dre, dim = dout[..., 0], dout[..., 1]
    dwa = xi.T @ dre - xq.T @ dim
    dwb = xi.T @ dim - xq.T @ dre
    dw = np.concatenate([dwa, dwb], axis=1)
    dba = (dre - dim).sum(axis=0)
    dbias = np.concatenate([dba, -dba])
    dxi = dre @ wa.T + dim @ wb.T
    dxq = -(dre @ wb.T) - dim @ wa.T
    dx = np.stack([dxi, dxq], axis=-1)
    return dx.astype(x.dtype), dw.astype(x.dtype), dbias.astype(x.dtype)


def _tf_pad(in_size: int, k: int, stride: int, padding: str) -> Tuple[int, int, int]:
    """TensorFlow SAME/VALID geometry -> (out_size, pad_before, pad_after)."""
    padding = padding.lower()
    if padding == "same":
        out = -(-in_size // stride)
        total = max((out - 1) * stride + k - in_size, 0)
        return out, total // 2, total - total // 2
    if padding == "valid":
        return -(-(in_size - k + 1) // stride), 0, 0
    raise ValueError(padding)


def layers_conv2d_complex_literal(inputs: np.ndarray, kernel: np.ndarray,
                                  bias: Optional[np.ndarray], strides=(1, 1),
                                  padding: str = "valid") -> np.ndarray:
    """Literal restatement of ``layers_conv2d_complex`` (complex.py:140-196) for a real
    5-D input [B, L, Wd, C, 2] and a TF conv3d kernel [kL, kW, 1, C, 2F].

    transpose to NDHWC [B,L,Wd,2,C] (:168) -> conv3d (kL,kW,1) (:183) -> reshape
    [.., 4, F] (:185) -> re=j0-j3, im=j1-j2 (:187-188) -> [B,L',W',F,2] (:191-192).
    The conv3d is evaluated tap by tap with TensorFlow's SAME/VALID padding rule.
    """
    B, L, Wd, C, two = inputs.shape
    assert two == 2
    kL, kW, k1, Cin, F2 = kernel.shape
    assert k1 == 1 and Cin == C
    F = F2 // 2
    sL, sW = strides
    Lo, pl0, pl1 = _tf_pad(L, kL, sL, padding)
    Wo, pw0, pw1 = _tf_pad(Wd, kW, sW, padding)
    x = np.transpose(inputs, (0, 1, 2, 4, 3))               # [B,L,Wd,2,C]
    x = np.pad(x, ((0, 0), (pl0, pl1), (pw0, pw1), (0, 0), (0, 0)))
    conv = np.zeros((B, Lo, Wo, 2, F2), dtype=inputs.dtype)
    for a in range(kL):
        for b in range(kW):
            patch = x[:, a:a + (Lo - 1) * sL + 1:sL, b:b + (Wo - 1) * sW + 1:sW]
            if patch.shape[1] != Lo or patch.shape[2] != Wo:
                continue
            conv += patch @ kernel[a, b, 0]
    if bias is not None:
        conv = conv + bias
    conv = conv.reshape(B, Lo, Wo, 4, F)
    re = conv[:, :, :, 0, :] - conv[:, :, :, 3, :]
    im = conv[:, :, :, 1, :] - conv[:, :, :, 2, :]
    return np.stack([re, im], axis=-1).astype(inputs.dtype)


def layers_conv1d_complex_literal(inputs: np.ndarray, kernel: np.ndarray,
                                  bias: Optional[np.ndarray], strides: int = 1,
                                  padding: str = "valid") -> np.ndarray:
    """``layers_conv1d_complex`` (complex.py:51-92): [B, L, C, 2] with a conv2d kernel
    [k, 1, C, 2F] -> [B, L', F, 2]; same 4-way combine (:84-85)."""
    B, L, C, two = inputs.shape
    k, k1, Cin, F2 = kernel.shape
    assert two == 2 and k1 == 1 and Cin == C
    F = F2 // 2
    Lo, p0, p1 = _tf_pad(L, k, strides, padding)
    x = np.transpose(inputs, (0, 1, 3, 2))                   # [B,L,2,C]  (:78)
    x = np.pad(x, ((0, 0), (p0, p1), (0, 0), (0, 0)))
    conv = np.zeros((B, Lo, 2, F2), dtype=inputs.dtype)
    for a in range(k):
        conv += x[:, a:a + (Lo - 1) * strides + 1:strides] @ kernel[a, 0]
    if bias is not None:
        conv = conv + bias
    conv = conv.reshape(B, Lo, 4, F)
    re = conv[:, :, 0, :] - conv[:, :, 3, :]
    im = conv[:, :, 1, :] - conv[:, :, 2, :]
    return np.stack([re, im], axis=-1).astype(inputs.dtype)


def nn_conv1d_complex(inputs: np.ndarray, filt: np.ndarray) -> np.ndarray:
    """``nn_conv1d_complex`` (complex.py:30-48): canonical complex product, SAME padding.
    inputs [B, L, C, 2], filt [k, C, 1, 2] -> [B, L, 2] (re | im concatenated on axis 2)."""
    B, L, C, _ = inputs.shape
    k = filt.shape[0]
    _, p0, p1 = _tf_pad(L, k, 1, "same")

    def conv(x, f):                                           # x [B,L,C], f [k,C,1]
        xp = np.pad(x, ((0, 0), (p0, p1), (0, 0)))
        out = np.zeros((B, L, 1), dtype=inputs.dtype)
        for a in range(k):
            out += xp[:, a:a + L] @ f[a]
        return out
    xi, xq = inputs[..., 0], inputs[..., 1]
    fr, fi = filt[..., 0], filt[..., 1]
    re = conv(xi, fr) - conv(xq, fi)
    im = conv(xi, fi) + conv(xq, fr)
    return np.concatenate([re, im], axis=2).astype(inputs.dtype)


# --------------------------------------------------------------------------------------
# R2-R6  dense + demodulation tail + loss + BER
# --------------------------------------------------------------------------------------
def leaky(x):
    """tf.nn.leaky_relu = max(alpha*x, x)."""
    return np.maximum(x.dtype.type(LEAKY_ALPHA) * x, x)


def softmax_pairs(u):
    m = u.max(axis=-1, keepdims=True)
    e = np.exp(u - m)
    return e / e.sum(axis=-1, keepdims=True)


@dataclass
class RxConfig:
    """Shapes of ``ofdm_dense_rx`` (model.py:1222-1292)."""
    S: int = 7            # nsymbol
    kin: int = 80         # samples per symbol seen by the C-Conv (N+CP if cp else N)
    F: int = 64           # nfilter
    D: int = 320          # frame_size (data cells per frame)
    nbits: int = 2

    @property
    def m(self):
        return 2 ** self.nbits


PARAM_NAMES = ("fft_like/conv3d/kernel", "fft_like/conv3d/bias",
               "demodulation/dense/kernel", "demodulation/dense/bias",
               "demodulation/conv2d/kernel", "demodulation/conv2d/bias",
               "demodulation/dense_1/kernel", "demodulation/dense_1/bias")
REGULARIZED = ("demodulation/dense/kernel", "demodulation/dense/bias",
               "demodulation/dense_1/kernel", "demodulation/dense_1/bias")


def param_shapes(cfg: RxConfig) -> Dict[str, Tuple[int, ...]]:
    """Live parameter shapes (dead conv3d taps dropped)."""
    return {
        "fft_like/conv3d/kernel": (cfg.kin, 2 * cfg.F),
        "fft_like/conv3d/bias": (2 * cfg.F,),
        "demodulation/dense/kernel": (2 * cfg.S * cfg.F, 2 * cfg.D),
        "demodulation/dense/bias": (2 * cfg.D,),
        "demodulation/conv2d/kernel": (2, cfg.m),
        "demodulation/conv2d/bias": (cfg.m,),
        "demodulation/dense_1/kernel": (cfg.m + 2, 2 * cfg.nbits),
        "demodulation/dense_1/bias": (2 * cfg.nbits,),
    }


def init_params(cfg: RxConfig, seed: int = 1, dtype=np.float32) -> Dict[str, np.ndarray]:
    """glorot-uniform kernels / zero biases with the reference's fan computation
    (conv3d fans include the K dead taps)."""
    rng = np.random.RandomState(seed)
    fans = {
        "fft_like/conv3d/kernel": (cfg.kin * cfg.kin, cfg.kin * 2 * cfg.F),
        "demodulation/dense/kernel": (2 * cfg.S * cfg.F, 2 * cfg.D),
        "demodulation/conv2d/kernel": (2, cfg.m),
        "demodulation/dense_1/kernel": (cfg.m + 2, 2 * cfg.nbits),
    }
    p = {}
    for name, shp in param_shapes(cfg).items():
        if name.endswith("bias"):
            p[name] = np.zeros(shp, dtype=dtype)
        else:
            fi, fo = fans[name]
            lim = math.sqrt(6.0 / (fi + fo))
            p[name] = rng.uniform(-lim, lim, size=shp).astype(dtype)
    return p


def rx_forward(p: Dict[str, np.ndarray], x_norm: np.ndarray, cfg: RxConfig, keep=False):
    """``ofdm_dense_rx`` forward on the already normalised input [Bf,S,kin,2]
    (model.py:1246-1291).  Returns probabilities [Bf, D, nbits, 2] (+ saved tensors)."""
    Bf = x_norm.shape[0]
    rows = x_norm.reshape(Bf * cfg.S, cfg.kin, 2)
    fft = cconv_gemm_fwd(rows, p["fft_like/conv3d/kernel"], p["fft_like/conv3d/bias"])
    a = fft.reshape(Bf, cfg.S * cfg.F * 2)                                    # :1268
    z = a @ p["demodulation/dense/kernel"] + p["demodulation/dense/bias"]     # :1269-1274
    iq = z.reshape(Bf * cfg.D, 2)                                             # :1276
    pre1 = iq @ p["demodulation/conv2d/kernel"] + p["demodulation/conv2d/bias"]   # :1278
    h1 = leaky(pre1)                                                          # :1280
    c = np.concatenate([h1, iq], axis=-1)                                     # :1282
    pre2 = c @ p["demodulation/dense_1/kernel"] + p["demodulation/dense_1/bias"]  # :1283
    u = leaky(pre2)                                                           # activation
    prob = softmax_pairs(u.reshape(Bf * cfg.D, cfg.nbits, 2))                 # :1290-1291
    out = prob.reshape(Bf, cfg.D, cfg.nbits, 2)
    if keep:
        return out, dict(rows=rows, fft=fft, a=a, z=z, iq=iq, pre1=pre1, h1=h1, c=c,
                         pre2=pre2, u=u, prob=prob)
    return out


def loss_ber(prob: np.ndarray, bits: np.ndarray):
    """ofdmreceiver_np.py:154-169 + util.py:44-48.

    ce = softmax_cross_entropy_with_logits(one_hot(bits), logits=prob) -- the softmax is
    applied a second time to the probabilities; decision = argmax (first index on ties);
    conf[label, pred]; berlin = (c01+c10)/sum computed in float64 and cast to float32.
    Returns dict(ce_mean, conf[2,2] int64, berlin, log_ber).
    """
    dt = prob.dtype
    pr = prob.reshape(-1, 2)
    y = bits.reshape(-1).astype(np.int64)
    mx = pr.max(axis=-1, keepdims=True)
    lse = np.log(np.exp(pr - mx).sum(axis=-1)) + mx[:, 0]
    ce = lse - pr[np.arange(pr.shape[0]), y]
    ce_mean = ce.mean(dtype=dt)
    pred = np.argmax(pr, axis=-1)
    conf = np.zeros((2, 2), dtype=np.int64)
    np.add.at(conf, (y, pred), 1)
    berlin64 = float(conf[0, 1] + conf[1, 0]) / float(conf.sum())
    berlin = np.float32(berlin64)
    with np.errstate(divide="ignore"):
        log_ber = np.log(np.float64(berlin64))
    return dict(ce_mean=ce_mean, conf=conf, berlin=berlin, log_ber=log_ber, ce=ce)


def reg_sum(p: Dict[str, np.ndarray]):
    """sum(tf.GraphKeys.REGULARIZATION_LOSSES): l2(0.01) on demodulation dense/dense_1
    kernel+bias only (model.py:1271-1272,1285-1286)."""
    return sum(REG_L2 * np.sum(np.square(p[n]), dtype=p[n].dtype) for n in REGULARIZED)


def total_loss(p, lb):
    """ofdmreceiver_np.py:171."""
    return (lb["ce_mean"] + lb["berlin"] * np.float32(REG_COEFF) * reg_sum(p)
            + np.float32(BER_COEFF) * np.float32(lb["log_ber"]))


def tail_forward_backward(z: np.ndarray, bits: np.ndarray, w1, b1, w2, b2, nbits: int):
    """Demodulation tail + loss on its own (model.py:1278-1291, ofdmreceiver_np.py:154-169) with
    the hand-derived backward of ce_mean.

    z [cells,2], bits [cells,nbits].  Returns dict(prob, loss_ber fields, dz, grads{w1,b1,w2,b2},
    pre1, pre2) -- pre1/pre2 let tests locate cells sitting on the leaky-ReLU kink."""
    dt = z.dtype
    m = 2 ** nbits
    cells = z.shape[0]
    pre1 = z @ w1 + b1
    h1 = leaky(pre1)
    c = np.concatenate([h1, z], axis=-1)
    pre2 = c @ w2 + b2
    u = leaky(pre2)
    prob = softmax_pairs(u.reshape(cells, nbits, 2))
    lb = loss_ber(prob, bits)
    pr = prob.reshape(-1, 2)
    ncls = pr.shape[0]
    y = bits.reshape(-1).astype(np.int64)
    g = softmax_pairs(pr)                       # d ce / d p = softmax(p) - onehot
    g[np.arange(ncls), y] -= 1.0
    g = g / dt.type(ncls)                       # mean
    du = pr * (g - (g * pr).sum(axis=-1, keepdims=True))          # through the first softmax
    du = du.reshape(cells, 2 * nbits)
    dpre2 = du * np.where(pre2 > 0, dt.type(1.0), dt.type(LEAKY_ALPHA))
    gW2 = c.T @ dpre2
    gb2 = dpre2.sum(axis=0)
    dc = dpre2 @ w2.T
    dpre1 = dc[:, :m] * np.where(pre1 > 0, dt.type(1.0), dt.type(LEAKY_ALPHA))
    gW1 = z.T @ dpre1
    gb1 = dpre1.sum(axis=0)
    dz = dpre1 @ w1.T + dc[:, m:]
    out = dict(lb)
    out.update(prob=prob, dz=dz, grads=dict(w1=gW1, b1=gb1, w2=gW2, b2=gb2), pre1=pre1, pre2=pre2)
    return out


def rx_forward_backward(p: Dict[str, np.ndarray], x_norm: np.ndarray, bits: np.ndarray,
                        cfg: RxConfig, need_dx: bool = False):
    """Forward + hand-derived backward of ce_mean + berlin*REG_COEFF*sum(reg)
    (the log(berlin) term carries no gradient).  SURVEY.md Appendix A.2-A.4.

    Returns (grads dict, info dict with ce_mean/conf/berlin/..., and dx if asked)."""
    dt = x_norm.dtype
    Bf = x_norm.shape[0]
    prob4, t = rx_forward(p, x_norm, cfg, keep=True)
    tl = tail_forward_backward(t["iq"], bits.reshape(Bf * cfg.D, cfg.nbits),
                               p["demodulation/conv2d/kernel"], p["demodulation/conv2d/bias"],
                               p["demodulation/dense_1/kernel"], p["demodulation/dense_1/bias"], cfg.nbits)
    lb = {k: tl[k] for k in ("ce_mean", "conf", "berlin", "log_ber", "ce")}
    dz = tl["dz"].reshape(Bf, 2 * cfg.D)
    Wd = p["demodulation/dense/kernel"]
    gWd = t["a"].T @ dz
    gbd = dz.sum(axis=0)
    da = dz @ Wd.T
    dfft = da.reshape(Bf * cfg.S, cfg.F, 2)
    dx, gWc, gbc = cconv_gemm_bwd(t["rows"], p["fft_like/conv3d/kernel"], dfft)
    grads = {
        "fft_like/conv3d/kernel": gWc, "fft_like/conv3d/bias": gbc,
        "demodulation/dense/kernel": gWd, "demodulation/dense/bias": gbd,
        "demodulation/conv2d/kernel": tl["grads"]["w1"], "demodulation/conv2d/bias": tl["grads"]["b1"],
        "demodulation/dense_1/kernel": tl["grads"]["w2"], "demodulation/dense_1/bias": tl["grads"]["b2"],
    }
    # + berlin * REG_COEFF * d/dw (0.01 * sum w^2)
    rs = lb["berlin"].astype(dt) * dt.type(REG_COEFF) * dt.type(2.0 * REG_L2)
    for n in REGULARIZED:
        grads[n] = grads[n] + rs * p[n]
    grads = {k: v.astype(dt) for k, v in grads.items()}
    info = dict(lb)
    info["cost"] = total_loss(p, lb)
    info["prob"] = prob4
    info["dz"] = dz
    info["dfft"] = dfft
    info["saved"] = t
    info["pre1"], info["pre2"] = tl["pre1"], tl["pre2"]
    if need_dx:
        return grads, info, dx.reshape(x_norm.shape)
    return grads, info


# --------------------------------------------------------------------------------------
# R7  optimizer   (dev/py/ofdmreceiver_np.py:185-189; TF 1.15 training_ops ApplyAdam)
# --------------------------------------------------------------------------------------
@dataclass
class AdamState:
    """tf.train.AdamOptimizer slots + the float32 global_step variable."""
    m: Dict[str, np.ndarray] = field(default_factory=dict)
    v: Dict[str, np.ndarray] = field(default_factory=dict)
    beta1_power: np.float32 = np.float32(ADAM_BETA1)
    beta2_power: np.float32 = np.float32(ADAM_BETA2)
    global_step: np.float32 = np.float32(0.0)


def adam_init(p) -> AdamState:
    st = AdamState()
    st.m = {k: np.zeros_like(v) for k, v in p.items()}
    st.v = {k: np.zeros_like(v) for k, v in p.items()}
    return st


def learning_rate(global_step: np.float32) -> np.float32:
    """tf.train.exponential_decay(0.001, step, 500, 0.98, staircase=True), float32."""
    e = np.floor(np.float32(global_step) / np.float32(LR_DECAY_STEPS))
    return np.float32(LR0) * np.power(np.float32(LR_DECAY), e, dtype=np.float32)


def adam_tf_step(p, grads, st: AdamState):
    """One ApplyAdam per variable, TF kernel form (training_ops.cc):
        alpha = lr*sqrt(1-beta2_power)/(1-beta1_power)
        m += (g-m)*(1-beta1);  v += (g*g-v)*(1-beta2);  var -= m*alpha/(sqrt(v)+eps)
    then beta powers *= beta, global_step += 1.  In place."""
    f = np.float32
    lr = learning_rate(st.global_step)
    alpha = f(lr * np.sqrt(f(1.0) - st.beta2_power, dtype=f) / (f(1.0) - st.beta1_power))
    for k in p:
        g = grads[k].astype(f)
        st.m[k] += (g - st.m[k]) * (f(1.0) - f(ADAM_BETA1))
        st.v[k] += (g * g - st.v[k]) * (f(1.0) - f(ADAM_BETA2))
        p[k] -= (st.m[k] * alpha) / (np.sqrt(st.v[k]) + f(ADAM_EPS))
    st.beta1_power = f(st.beta1_power * f(ADAM_BETA1))
    st.beta2_power = f(st.beta2_power * f(ADAM_BETA2))
    st.global_step = f(st.global_step + f(1.0))
    return alpha


def rx_train_step(p, st: AdamState, x_raw: np.ndarray, bits: np.ndarray, cfg: RxConfig):
    """One ``session.run(train_op, ...)`` of the basic receiver
    (ofdmreceiver_np.py:128-189,234): normalise -> fwd -> loss/BER -> bwd -> Adam."""
    x_norm, _, _ = batch_moment_norm(x_raw.reshape(x_raw.shape[0], -1))
    x_norm = x_norm.reshape(x_raw.shape)
    grads, info = rx_forward_backward(p, x_norm, bits, cfg)
    info["alpha"] = adam_tf_step(p, grads, st)
    info["grads"] = grads
    return info


def rx_eval(p, x_raw: np.ndarray, bits: np.ndarray, cfg: RxConfig):
    """Inference-side session.run([conf_matrix, berlin, ce_mean, ...]) (ofdmreceiver_np.py:80)."""
    x_norm, _, _ = batch_moment_norm(x_raw.reshape(x_raw.shape[0], -1))
    x_norm = x_norm.reshape(x_raw.shape)
    prob = rx_forward(p, x_norm, cfg)
    lb = loss_ber(prob, bits)
    lb["prob"] = prob
    lb["cost"] = total_loss(p, lb)
    _, lb["tx_power"] = complex_clip(x_norm, CLIP_PEAK)
    return lb
