"""HIP kernels against TensorFlow's own graph (``-m gpu``).

The reference's archived v1 graphs (tests/golden/v1_graph, see tests/test_graph_golden.py) are evaluated by
oracle/tf_graph.py in float64; the same receiver is run through the C-ABI operators (R0 normalisation, C-Conv GEMM,
dense, demodulation tail with its loss -- fused with the dense layer for nbits <= 2) on the GPU, forward and backward.
The v1 tail has two stacked 1x1 convolutions with no activation in between, i.e. one affine map w1.w1b, b1.w1b + b1b:
the kernels get the folded weights, TensorFlow's gradients of the two factors follow from the kernel's by the chain rule.
Tolerance: 1e-5 relative (north_star), confusion counts exact.
"""
import glob
import math
import os

import numpy as np
import pytest
import torch

from oracle import tf_graph as T

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GRAPHS = sorted(glob.glob(os.path.join(HERE, "golden", "v1_graph", "*.json.gz")))


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def rel(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.mark.parametrize("path", GRAPHS, ids=[os.path.basename(p)[:-8] for p in GRAPHS])
def test_hip_operators_match_tensorflows_graph(path):
    from dl_ofdm_amd import ops
    base = os.path.basename(path)
    nbits, cp = int(base.split("_")[2][0]), "cpTrue" in base
    g = T.Graph(T.load_manifest(path), dtype=np.float64)
    rng = np.random.RandomState(7 + 10 * nbits + cp)
    B, S, D1, F = 4, 8, 46, 64
    kin = 80 if cp else 64
    shapes, gmap = g.variables(), g.trainable_gradients()
    # float32-representable values everywhere, so that both sides start from identical numbers
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)          # noqa: E731
    vars_ = {}
    for k in gmap:
        sc = 0.5 if ("conv2d" in k or "dense_1" in k) else (0.06 if "dense" in k else 0.05)
        vars_[k] = f32(rng.randn(*shapes[k]) * sc)
    w1, b1 = vars_["demodulation/conv2d/kernel"][0, 0], vars_["demodulation/conv2d/bias"]
    w1b, b1b = vars_["demodulation/conv2d_1/kernel"][0, 0], vars_["demodulation/conv2d_1/bias"]
    w_eff, b_eff = f32(w1 @ w1b), f32(b1 @ w1b + b1b)
    # make the folded weights exactly representable: give TensorFlow's graph a factorisation of the float32 w_eff
    w1b_inv = np.linalg.inv(w1b)
    vars_["demodulation/conv2d/kernel"] = (w_eff @ w1b_inv)[None, None]
    vars_["demodulation/conv2d/bias"] = (b_eff - b1b) @ w1b_inv
    w1, b1 = vars_["demodulation/conv2d/kernel"][0, 0], vars_["demodulation/conv2d/bias"]
    names = sorted(gmap)
    for attempt in range(200):
        x = f32(rng.randn(B, 8, 80, 2) * rng.uniform(0.5, 2.0, (8, 80, 2)) + rng.randn(8, 80, 2))
        bits = rng.randint(0, 2, (B * 8, D1, nbits))
        feed = {"tx_ofdm": x, "bits_in": bits, "SNR": np.full((B, 1), 10.0)}
        fetch = ["input", "receiver/fft_like/fft_out", "output", "ce_mean", "conf_matrix", "tx_power",
                 "receiver/demodulation/conv2d_1/BiasAdd", "receiver/demodulation/dense_1/BiasAdd"] + [gmap[n] for n in names]
        exact = {"transmitter/div/y": math.sqrt(2.0), "transmitter/batchnorm/add/y": 1e-9}
        tf = dict(zip(fetch, g.run(fetch, feed, vars_, const_override=exact)))
        prob = tf["output"].reshape(-1, 2)
        # keep away from the leaky-ReLU kinks and from decision ties (a float32-vs-float64 sign flip there is not a
        # rounding-level effect)
        if (np.abs(tf["receiver/demodulation/conv2d_1/BiasAdd"]).min() > 2e-4 and
                np.abs(tf["receiver/demodulation/dense_1/BiasAdd"]).min() > 2e-4 and
                np.abs(prob[:, 1] - prob[:, 0]).min() > 2e-5):
            break
    else:
        raise AssertionError("could not draw a well-conditioned case")

    # ---- the same receiver through the HIP operators ----
    xt = dev(x)
    xn = ops.batch_moment_norm(xt)
    assert rel(xn.cpu().numpy(), tf["input"]) <= 1e-5
    _, pw = ops.clip_power(xn, peak=8.0, want_clipped=False)
    assert abs(float(pw) - float(tf["tx_power"])) <= 1e-5 * float(tf["tx_power"])
    xr = xn if cp else xn[:, :, 16:16 + 64, :]
    t0 = (kin - 1) // 2
    wc = dev(vars_["fft_like/conv3d/kernel"][0, t0, 0]).requires_grad_()
    bc = dev(vars_["fft_like/conv3d/bias"]).requires_grad_()
    Wd = dev(vars_["demodulation/dense/kernel"]).requires_grad_()
    bd = dev(vars_["demodulation/dense/bias"]).requires_grad_()
    w2, b2 = vars_["demodulation/dense_1/kernel"], vars_["demodulation/dense_1/bias"]
    tailp = ops.pack_tail_params(dev(w_eff), dev(b_eff), dev(w2), dev(b2)).requires_grad_()
    fft = ops.cconv_gemm(xr.reshape(B * S, kin, 2).contiguous(), wc, bc)
    assert rel(fft.detach().cpu().numpy().reshape(B, S, F, 2), tf["receiver/fft_like/fft_out"]) <= 1e-5
    flat = fft.reshape(B, S * F * 2)
    labels = dev(bits.reshape(B, S * D1, nbits), torch.int32)
    if ops.dense_tail_supported(flat, Wd, nbits):
        ce, p, mbuf = ops.dense_demod_tail_loss(flat, Wd, bd, tailp, labels, nbits)
    else:
        z = ops.dense(flat, Wd, bd)
        ce, p, mbuf = ops.demod_tail_loss(z.view(B, S * D1, 2), tailp, labels, nbits)
    ce.backward()
    m = ops.read_metrics(mbuf)
    assert rel(p.cpu().numpy().reshape(-1, 2), prob) <= 1e-5
    assert abs(m["ce_mean"] - float(tf["ce_mean"])) <= 1e-5 * float(tf["ce_mean"])
    assert np.array_equal(np.array(m["conf"]), tf["conf_matrix"])
    # gradients of ce_mean: TensorFlow's include berlin*1e-4*scale*w on the four regularised variables -- remove it
    ber32 = float(np.float32((m["conf"][0][1] + m["conf"][1][0]) / float(np.sum(m["conf"]))))
    scales = {n.replace("receiver/", "").replace("/Regularizer/l2_regularizer/scale", ""): float(g.const(n))
              for n in g.order if n.endswith("l2_regularizer/scale")}
    def tfgrad(n):
        gr = tf[gmap[n]]
        if n in scales:
            gr = gr - ber32 * float(np.float32(1e-4)) * scales[n] * vars_[n]
        return gr
    tol = 1e-5
    assert rel(wc.grad.cpu().numpy(), tfgrad("fft_like/conv3d/kernel")[0, t0, 0]) <= tol
    assert rel(bc.grad.cpu().numpy(), tfgrad("fft_like/conv3d/bias")) <= tol
    assert rel(Wd.grad.cpu().numpy(), tfgrad("demodulation/dense/kernel")) <= tol
    assert rel(bd.grad.cpu().numpy(), tfgrad("demodulation/dense/bias")) <= tol
    gw1e, gb1e, gw2, gb2 = (t.cpu().numpy().astype(np.float64) for t in ops.unpack_tail_params(tailp.grad, nbits))
    assert rel(gw1e @ w1b.T, tfgrad("demodulation/conv2d/kernel")[0, 0]) <= tol
    assert rel(gb1e @ w1b.T, tfgrad("demodulation/conv2d/bias")) <= tol
    assert rel(w1.T @ gw1e + np.outer(b1, gb1e), tfgrad("demodulation/conv2d_1/kernel")[0, 0]) <= tol
    assert rel(gb1e, tfgrad("demodulation/conv2d_1/bias")) <= tol
    assert rel(gw2, tfgrad("demodulation/dense_1/kernel")) <= tol
    assert rel(gb2, tfgrad("demodulation/dense_1/bias")) <= tol
