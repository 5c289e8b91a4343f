"""Pins the CPU oracle (tests' source of truth): two independent formulations must agree.

 * GEMM-form C-Conv + hand-derived backward (oracle/dccn_oracle.py) vs the literal TF-style graph
   (zero-padded NDHWC conv3d, autograd; oracle/torch_ref.py), in float64;
 * the general layers_conv2d_complex / conv1d literal restatements vs the GEMM form on im2col patches;
 * structural facts of the reference's own checkpoints (tests/golden/v1_index_manifest.json).
TF numerics themselves are unpinned (TensorFlow 1.15 is unavailable) -- see the oracle header."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import dccn_oracle as O
from oracle.torch_ref import LiteralRx


def _case(nbits, Bf=10, kin=80, F=64, D=320, seed=0):
    cfg = O.RxConfig(S=7, kin=kin, F=F, D=D, nbits=nbits)
    rng = np.random.RandomState(seed)
    p = O.init_params(cfg, seed=seed + 1, dtype=np.float64)
    for k in p:
        if k.endswith("bias"):
            p[k] = rng.uniform(-0.1, 0.1, p[k].shape)
    x = rng.randn(Bf, 7, kin, 2) * 0.7 + 0.05
    bits = rng.randint(0, 2, (Bf, D, nbits))
    return cfg, p, x, bits


@pytest.mark.parametrize("nbits,kin", [(1, 80), (2, 80), (3, 64), (4, 80)])
def test_gemm_form_and_manual_backward_match_literal_autograd(nbits, kin):
    cfg, p, x, bits = _case(nbits, kin=kin)
    lit = LiteralRx(p, cfg, dtype=torch.float64, literal_conv=True)
    g_l, info_l = lit.forward_backward(x, bits)
    xn, _, _ = O.batch_moment_norm(x.reshape(x.shape[0], -1))
    xn = xn.reshape(x.shape)
    g, info = O.rx_forward_backward(p, xn, bits, cfg)
    assert np.abs(xn - info_l["x_norm"]).max() < 1e-12
    assert np.abs(info["prob"] - info_l["prob"]).max() < 1e-12
    assert abs(info["ce_mean"] - info_l["ce_mean"]) < 1e-12
    assert np.array_equal(info["conf"], info_l["conf"])
    t0 = (kin - 1) // 2
    for k in g:
        gl = g_l[k]
        if k == "fft_like/conv3d/kernel":
            dead = gl.copy()
            dead[0, t0, 0] = 0
            assert np.abs(dead).max() == 0.0            # only the centre tap ever receives gradient
            gl = gl[0, t0, 0]
        assert np.abs(g[k] - gl).max() <= 1e-9 * max(np.abs(gl).max(), 1e-30), k


def test_adam_matches_literal_over_steps():
    cfg, p, x, bits = _case(2, Bf=6, D=40, F=16, kin=20)
    lit = LiteralRx(p, cfg, dtype=torch.float64, literal_conv=False)
    p2 = {k: v.copy() for k, v in p.items()}
    st = O.adam_init(p2)
    rng = np.random.RandomState(3)
    for _ in range(4):
        xi = x + 0.1 * rng.randn(*x.shape)
        lit.train_step(xi, bits)
        O.rx_train_step(p2, st, xi, bits, cfg)
    lp = lit.live_params()
    for k in p2:
        # the oracle keeps the Adam scalars in float32 like TF does; the literal model in Python floats
        assert np.abs(lp[k] - p2[k]).max() <= 1e-6 * max(np.abs(p2[k]).max(), 1e-30), k


def test_learning_rate_schedule():
    assert O.learning_rate(np.float32(0)) == np.float32(1e-3)
    assert O.learning_rate(np.float32(499)) == np.float32(1e-3)
    assert abs(O.learning_rate(np.float32(500)) - 0.98e-3) < 1e-9
    assert abs(O.learning_rate(np.float32(1700)) - 1e-3 * 0.98 ** 3) < 1e-9


@pytest.mark.parametrize("shape,kshape,padding", [
    ((3, 7, 1, 80, 2), (1, 80, 1, 80, 16), "same"),      # receiver use: only tap 39 is live
    ((3, 7, 64, 1, 2), (1, 64, 1, 1, 12), "valid"),      # equalizer (1,K) VALID -> GEMM over K
    ((2, 7, 16, 1, 2), (7, 16, 1, 1, 2), "same"),        # equalizer (S,K) SAME, F=1 -> true 2-D correlation
    ((2, 5, 6, 3, 2), (3, 2, 1, 3, 8), "valid"),
])
def test_conv2d_complex_literal_equals_gemm_on_patches(shape, kshape, padding):
    rng = np.random.RandomState(0)
    x = rng.randn(*shape)
    k = rng.randn(*kshape)
    b = rng.randn(kshape[-1])
    ref = O.layers_conv2d_complex_literal(x, k, b, (1, 1), padding)
    # im2col by hand, then the GEMM form
    B, L, Wd, C, _ = shape
    kL, kW = kshape[0], kshape[1]
    Lo, pl0, pl1 = O._tf_pad(L, kL, 1, padding)
    Wo, pw0, pw1 = O._tf_pad(Wd, kW, 1, padding)
    xp = np.pad(x, ((0, 0), (pl0, pl1), (pw0, pw1), (0, 0), (0, 0)))
    patches = np.stack([xp[:, a:a + Lo, c:c + Wo] for a in range(kL) for c in range(kW)], axis=3)
    rows = patches.reshape(B * Lo * Wo, kL * kW * C, 2)
    out = O.cconv_gemm_fwd(rows, k.reshape(kL * kW * C, -1), b).reshape(ref.shape)
    assert np.abs(out - ref).max() < 1e-10
    if shape[2] == 1 and padding == "same":               # centre-tap claim of SURVEY.md Appendix A.2
        t0 = (kW - 1) // 2
        live = O.cconv_gemm_fwd(x.reshape(B * L, C, 2), k[0, t0, 0], b).reshape(ref.shape)
        assert np.abs(live - ref).max() < 1e-10


def test_conv1d_and_nn_conv1d_literals():
    rng = np.random.RandomState(1)
    x = rng.randn(2, 9, 4, 2)
    k = rng.randn(3, 1, 4, 10)
    out = O.layers_conv1d_complex_literal(x, k, None, 1, "same")
    assert out.shape == (2, 9, 5, 2)
    # the layer uses im = I.Wb - Q.Wa; nn_conv1d_complex the canonical product (complex.py:44-45)
    f = rng.randn(3, 4, 1, 2)
    canon = O.nn_conv1d_complex(x, f)
    xc = x[..., 0] + 1j * x[..., 1]
    fc = f[..., 0, 0] + 1j * f[..., 0, 1]
    xp = np.pad(xc, ((0, 0), (1, 1), (0, 0)))
    want = sum(xp[:, a:a + 9] @ fc[a] for a in range(3))
    assert np.abs(canon[..., 0] - want.real).max() < 1e-12 and np.abs(canon[..., 1] - want.imag).max() < 1e-12


@pytest.mark.parametrize("shape,F,kern,strides,padding", [
    ((2, 9, 10, 3, 2), 5, (3, 2), (2, 1), "valid"),
    ((3, 8, 12, 2, 2), 4, (3, 5), (1, 2), "same"),
    ((2, 11, 1, 4, 2), 6, (5, 1), (1, 1), "same"),
])
def test_torch_twin_of_the_literal_convolution(shape, F, kern, strides, padding):
    """oracle/torch_ref.layers_conv2d_complex_literal_t (the autograd reference of the general-k backward) is the numpy
    literal, and its gradients satisfy the adjoint identity <g, J v> = <J^T g, v> of a linear map."""
    from oracle.torch_ref import layers_conv2d_complex_literal_t, layers_conv1d_complex_literal_t
    rng = np.random.RandomState(3)
    x = rng.randn(*shape)
    k = rng.randn(kern[0], kern[1], 1, shape[3], 2 * F)
    b = rng.randn(2 * F)
    xt, kt = torch.tensor(x, requires_grad=True), torch.tensor(k, requires_grad=True)
    y = layers_conv2d_complex_literal_t(xt, kt, torch.tensor(b), strides, padding)
    ref = O.layers_conv2d_complex_literal(x, k, b, strides, padding)
    assert y.shape == ref.shape and np.abs(y.detach().numpy() - ref).max() <= 1e-12 * np.abs(ref).max()
    g = rng.randn(*ref.shape)
    y.backward(torch.tensor(g))
    v = rng.randn(*shape)
    jv = O.layers_conv2d_complex_literal(v, k, None, strides, padding)             # the layer is linear in x
    assert abs((g * jv).sum() - (xt.grad.numpy() * v).sum()) <= 1e-10 * np.abs(g * jv).sum()
    if kern[1] == 1:
        y1 = layers_conv1d_complex_literal_t(torch.tensor(x[:, :, 0]), torch.tensor(k[:, :, 0]), torch.tensor(b), strides[0], padding)
        ref1 = O.layers_conv1d_complex_literal(x[:, :, 0], k[:, :, 0], b, strides[0], padding)
        assert np.abs(y1.numpy() - ref1).max() <= 1e-12 * np.abs(ref1).max()


def test_loss_ber_semantics():
    prob = np.array([[[[0.5, 0.5]], [[0.2, 0.8]], [[0.9, 0.1]]]])      # [1,3,1,2]
    bits = np.array([[[1], [1], [1]]])
    lb = O.loss_ber(prob, bits)
    assert np.array_equal(lb["conf"], [[0, 0], [2, 1]])                 # tie -> decision 0 (argmax)
    assert abs(float(lb["berlin"]) - 2 / 3) < 1e-7
    want = np.mean([np.log(np.exp(a) + np.exp(b)) - b for a, b in [(0.5, 0.5), (0.2, 0.8), (0.9, 0.1)]])
    assert abs(lb["ce_mean"] - want) < 1e-12
    lb0 = O.loss_ber(np.array([[[[0.1, 0.9]]]]), np.array([[[1]]]))
    assert lb0["log_ber"] == -np.inf


def test_v1_checkpoint_manifest_structure(golden_dir):
    """Variable names / shapes of the reference's own v1 checkpoints."""
    man = json.load(open(os.path.join(golden_dir, "v1_index_manifest.json")))
    assert len(man) == 8
    q = man["OFDM_Dense3_2mod_snr6_cpTrue"]
    assert q["fft_like/conv3d/kernel"]["shape"] == [1, 80, 1, 80, 128]
    assert man["OFDM_Dense3_2mod_snr6_cpFalse"]["fft_like/conv3d/kernel"]["shape"] == [1, 64, 1, 64, 128]
    assert q["fft_like/conv3d/bias"]["shape"] == [128]
    assert q["demodulation/dense/kernel"]["shape"] == [1024, 736]
    assert q["demodulation/conv2d/kernel"]["shape"] == [1, 1, 2, 4]
    assert q["demodulation/dense_1/kernel"]["shape"] == [6, 4]
    assert q["global_step"]["shape"] == [] and q["global_step"]["dtype"] == 1        # DT_FLOAT
    for v in ("fft_like/conv3d/kernel", "demodulation/dense/kernel", "demodulation/dense_1/bias"):
        assert v + "/Adam" in q and v + "/Adam_1" in q
    assert "beta1_power" in q and "beta2_power" in q
    for b in (1, 2, 3, 4):
        e = man["OFDM_Dense3_%dmod_snr%d_cpTrue" % (b, 3 * b)]
        assert e["demodulation/conv2d/kernel"]["shape"] == [1, 1, 2, 2 ** b]
        assert e["demodulation/dense_1/kernel"]["shape"] == [2 ** b + 2, 2 * b]
    # the oracle's live shapes are those shapes with the dead conv3d taps dropped
    shp = O.param_shapes(O.RxConfig(S=8, kin=80, F=64, D=368, nbits=2))
    assert shp["fft_like/conv3d/kernel"] == (80, 128) and shp["demodulation/dense/kernel"] == (1024, 736)
