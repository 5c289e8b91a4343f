"""Pins oracle/dccn_oracle.py to the TensorFlow graphs the reference ships (test_v1/model/*.meta).

The fixtures tests/golden/v1_graph/*.json.gz are the node lists of those graphs (tests/golden/make_graph_golden.py);
oracle/tf_graph.py executes them op by op in NumPy.  TensorFlow itself cannot run here, but its *graph* can: forward,
loss/BER assembly and the ``gradients/...`` subgraph TF's autodiff generated are evaluated in float64 on seeded inputs
and compared with the oracle's building blocks (R0 normalisation, C-Conv GEMM form + hand-derived backward, dense,
demodulation tail + double-softmax CE + hand-derived backward, confusion/BER, cost) composed in the v1 topology
(8 symbols, 46 data carriers per symbol, two 1x1 convolutions in front of the leaky-ReLU).  Agreement is to 1e-12,
i.e. the oracle states the same mathematics as TensorFlow's graph; what remains unpinned is only the fp32 rounding of
TF's kernels.  Every constant the oracle hard-codes is read back from the graphs as well.
"""
import glob
import math
import os

import numpy as np
import pytest

from oracle import dccn_oracle as O
from oracle import tf_graph as T

HERE = os.path.dirname(os.path.abspath(__file__))
GRAPHS = sorted(glob.glob(os.path.join(HERE, "golden", "v1_graph", "*.json.gz")))


def _cfg_of(path):
    base = os.path.basename(path)
    nbits = int(base.split("_")[2][0])
    return nbits, "cpTrue" in base


def test_fixtures_present():
    assert len(GRAPHS) == 8
    assert sorted(_cfg_of(p) for p in GRAPHS) == sorted((b, c) for b in (1, 2, 3, 4) for c in (False, True))


@pytest.mark.parametrize("path", GRAPHS, ids=[os.path.basename(p)[:-8] for p in GRAPHS])
def test_constants_of_the_reference_graph_match_the_oracle(path):
    g = T.Graph(T.load_manifest(path), dtype=np.float32)
    nbits, cp = _cfg_of(path)
    m = 2 ** nbits
    kin = 80 if cp else 64
    c = lambda name: g.const(name)                                    # noqa: E731
    # R0: rsqrt(var + 1e-9), then / sqrt(2) in float32 (ofdmreceiver_np.py:128-129)
    assert float(c("transmitter/batchnorm/add/y")) == np.float32(O.NORM_EPS)
    assert float(c("transmitter/div/y")) == np.float32(math.sqrt(2.0))
    assert g.nodes["transmitter/moments/mean"]["op"] == "Mean" and list(c("transmitter/moments/mean/reduction_indices")) == [0]
    # R8: clip_by_norm(., 8) over the IQ axis
    assert float(c("transmitter/clip_by_norm/mul_1/y")) == O.CLIP_PEAK == float(c("transmitter/clip_by_norm/Maximum/y"))
    assert list(np.atleast_1d(c("transmitter/clip_by_norm/Sum/reduction_indices"))) in ([-1], [3])
    # R1: NDHWC conv3d, SAME, unit strides, [1,K,1,K,2F] kernel; IQ moved in front of the channel axis and back
    conv = g.nodes["receiver/fft_like/conv3d/Conv3D"]
    assert g.attr(conv, "data_format") == "NDHWC" and g.attr(conv, "padding") == "SAME"
    assert g.attr(conv, "strides")["i"] == [1, 1, 1, 1, 1]
    assert g.variables()["fft_like/conv3d/kernel"] == (1, kin, 1, kin, 128)
    assert list(c("receiver/fft_like/transpose/perm")) == [0, 1, 2, 4, 3] == list(c("receiver/fft_like/transpose_1/perm"))
    assert list(c("receiver/fft_like/Reshape/shape")) == [-1, 8, 1, kin, 2]
    assert list(c("receiver/fft_like/Reshape_1/shape")) == [-1, 8, 1, 4, 64]          # j = 2*iq + (c // F)
    # re = j0 - j3, im = j1 - j2 (complex.py:187-188)
    sub, sub1 = g.nodes["receiver/fft_like/sub"], g.nodes["receiver/fft_like/sub_1"]
    pick = lambda ref: int(c(ref + "/stack")[3])                      # noqa: E731  index on the j axis
    assert [pick(r) for r in sub["inputs"]] == [0, 3] and [pick(r) for r in sub1["inputs"]] == [1, 2]
    # glorot-uniform limit of the conv3d kernel counts the dead taps: fan_in = K*K, fan_out = K*2F (Appendix A.7)
    lim = math.sqrt(6.0 / (kin * kin + kin * 128))
    assert abs(float(c("fft_like/conv3d/kernel/Initializer/random_uniform/max")) - lim) < 1e-7
    assert abs(float(c("demodulation/dense/kernel/Initializer/random_uniform/max")) - math.sqrt(6.0 / (1024 + 736))) < 1e-7
    # R2/R3: dense [1024,736], 1x1 convs, leaky-ReLU 0.2 = max(alpha*x, x), concat [hidden, iq], softmax over pairs
    assert g.variables()["demodulation/dense/kernel"] == (1024, 736)
    assert g.variables()["demodulation/conv2d/kernel"] == (1, 1, 2, m)
    assert g.variables()["demodulation/dense_1/kernel"] == (m + 2, 2 * nbits)
    for a in ("receiver/demodulation/LeakyRelu", "receiver/demodulation/dense_1/LeakyRelu"):
        assert float(c(a + "/alpha")) == np.float32(O.LEAKY_ALPHA) and g.nodes[a]["op"] == "Maximum"
    cat = g.nodes["receiver/demodulation/concat"]
    assert cat["inputs"][:2] == ["receiver/demodulation/LeakyRelu", "receiver/demodulation/Reshape_1"]
    assert list(c("receiver/Reshape/shape")) == [-1, 46, nbits, 2]
    # R4/R5: one_hot depth 2, CE *with logits* applied to the softmax output, argmax over axis 1
    assert int(c("one_hot/depth")) == 2 and int(c("ArgMax/dimension")) == 1
    xent = g.nodes["softmax_cross_entropy_with_logits"]
    assert xent["op"] == "SoftmaxCrossEntropyWithLogits"
    assert g.nodes["Reshape_3"]["inputs"][0] == "receiver/Reshape_2" and g.nodes["receiver/Reshape_2"]["inputs"][0] == "receiver/Softmax"
    # R6: cost = ce_mean + berlin*1e-4*sum(reg) + 1.0*log(berlin)
    assert float(c("mul/y")) == np.float32(O.REG_COEFF) and float(c("mul_2/x")) == np.float32(O.BER_COEFF)
    regs = [n for n in g.order if n.endswith("l2_regularizer") and g.nodes[n]["op"] == "Mul"]
    assert sorted(r.split("/")[2] + "/" + r.split("/")[3] for r in regs) == ["dense/bias", "dense/kernel", "dense_1/bias", "dense_1/kernel"]
    # R7: Adam(0.9, 0.999, 1e-8), lr = 1e-3 * 0.98^floor(step/500), float32 global_step += 1.0
    assert (float(c("Adam/beta1")), float(c("Adam/beta2")), float(c("Adam/epsilon"))) == \
           (np.float32(O.ADAM_BETA1), np.float32(O.ADAM_BETA2), np.float32(O.ADAM_EPS))
    assert float(c("ExponentialDecay/learning_rate")) == np.float32(O.LR0)
    assert float(c("ExponentialDecay/Cast/x")) == O.LR_DECAY_STEPS and float(c("ExponentialDecay/Cast_1/x")) == np.float32(O.LR_DECAY)
    assert g.nodes["ExponentialDecay/Floor"]["op"] == "Floor"
    assert g.attr(g.nodes["global_step"], "dtype") == "float32" and float(c("Adam/value")) == 1.0
    assert set(g.trainable_gradients()) == {n for n in g.variables() if "/Adam" not in n and n not in
                                            ("global_step", "beta1_power", "beta2_power")}
    for step in (0.0, 499.0, 500.0, 1234.0):
        lr = g.run(["ExponentialDecay"], {}, {"global_step": np.float32(step)})[0]
        assert abs(float(lr) - float(O.learning_rate(np.float32(step)))) <= 1e-10


def _v1_oracle(vars_, x, bits, nbits, cp, scales, reg_coeff):
    """The oracle's building blocks composed in the v1 topology; everything float64."""
    B, S, D1 = x.shape[0], 8, 46
    kin = 80 if cp else 64
    F = 64
    xn, mean, var = O.batch_moment_norm(x.reshape(B, -1))
    xn = xn.reshape(x.shape)
    xr = xn if cp else xn[:, :, 16:16 + 64, :]
    K = vars_["fft_like/conv3d/kernel"]
    t0 = (kin - 1) // 2                                                # the only tap SAME padding lets meet data
    w_live, b_c = K[0, t0, 0], vars_["fft_like/conv3d/bias"]
    rows = xr.reshape(B * S, kin, 2)
    fft = O.cconv_gemm_fwd(rows, w_live, b_c)
    a = fft.reshape(B, S * F * 2)
    Wd, bd = vars_["demodulation/dense/kernel"], vars_["demodulation/dense/bias"]
    z = a @ Wd + bd
    cells = z.reshape(B * S * D1, 2)
    w1, b1 = vars_["demodulation/conv2d/kernel"][0, 0], vars_["demodulation/conv2d/bias"]
    w1b, b1b = vars_["demodulation/conv2d_1/kernel"][0, 0], vars_["demodulation/conv2d_1/bias"]
    w2, b2 = vars_["demodulation/dense_1/kernel"], vars_["demodulation/dense_1/bias"]
    # two stacked 1x1 convolutions without an activation between them are one affine map
    w_eff, b_eff = w1 @ w1b, b1 @ w1b + b1b
    t = O.tail_forward_backward(cells, bits.reshape(-1, nbits), w_eff, b_eff, w2, b2, nbits)
    dz = t["dz"].reshape(B, -1)
    da = dz @ Wd.T
    dx, gWc, gbc = O.cconv_gemm_bwd(rows, w_live, da.reshape(B * S, F, 2))
    gK = np.zeros_like(K)
    gK[0, t0, 0] = gWc                                                # every other tap: zero gradient, forever
    g = {
        "fft_like/conv3d/kernel": gK, "fft_like/conv3d/bias": gbc,
        "demodulation/dense/kernel": a.T @ dz, "demodulation/dense/bias": dz.sum(0),
        # chain rule through w_eff = w1.w1b, b_eff = b1.w1b + b1b
        "demodulation/conv2d/kernel": (t["grads"]["w1"] @ w1b.T)[None, None],
        "demodulation/conv2d/bias": t["grads"]["b1"] @ w1b.T,
        "demodulation/conv2d_1/kernel": (w1.T @ t["grads"]["w1"] + np.outer(b1, t["grads"]["b1"]))[None, None],
        "demodulation/conv2d_1/bias": t["grads"]["b1"],
        "demodulation/dense_1/kernel": t["grads"]["w2"], "demodulation/dense_1/bias": t["grads"]["b2"],
    }
    # cost = ce_mean + berlin*reg_coeff*sum_i scale_i*L2Loss(w_i) + log(berlin): d/dw = berlin*reg_coeff*scale*w
    ber32 = np.float64(np.float32(t["berlin"]))                        # the graph casts BER to float32 first
    reg = 0.0
    for n, sc in scales.items():
        g[n] = g[n] + ber32 * reg_coeff * sc * vars_[n]
        reg += sc * np.sum(np.square(vars_[n])) / 2
    clipped, power = O.complex_clip(xn, O.CLIP_PEAK)
    cost = t["ce_mean"] + ber32 * reg_coeff * reg + O.BER_COEFF * np.float64(np.float32(t["log_ber"]))   # Cast_4
    return dict(input=xn, fft_out=fft.reshape(B, S, F, 2), output=t["prob"].reshape(B * S, D1, nbits, 2),
                ce_mean=t["ce_mean"], conf=t["conf"], berlin=t["berlin"], log_ber=t["log_ber"], cost=cost,
                tx_power=power, tx_signal=clipped, grads=g)


@pytest.mark.parametrize("path", GRAPHS, ids=[os.path.basename(p)[:-8] for p in GRAPHS])
def test_oracle_forward_and_backward_equal_tensorflows_graph(path):
    nbits, cp = _cfg_of(path)
    g = T.Graph(T.load_manifest(path), dtype=np.float64)
    rng = np.random.RandomState(100 + 10 * nbits + cp)
    B = 3
    shapes = g.variables()
    vars_ = {}
    for k in g.trainable_gradients():
        sc = 0.5 if "conv2d" in k or "dense_1" in k else (0.1 if "dense" in k else 0.05)
        vars_[k] = rng.randn(*shapes[k]) * sc
    x = rng.randn(B, 8, 80, 2) * rng.uniform(0.5, 2.0, (8, 80, 2)) + rng.randn(8, 80, 2)
    bits = rng.randint(0, 2, (B * 8, 46, nbits))
    feed = {"tx_ofdm": x, "bits_in": bits, "SNR": np.full((B, 1), 10.0)}
    gmap = g.trainable_gradients()
    names = sorted(gmap)
    fetch = ["input", "receiver/fft_like/fft_out", "output", "ce_mean", "conf_matrix", "linear_ber", "log_ber", "cost",
             "tx_power", "tx_signal"] + [gmap[n] for n in names]
    # the float32 constants of the graph equal float32(oracle constant) (test above); this float64 run uses the
    # un-rounded values so that graph and oracle state exactly the same arithmetic
    exact = {"transmitter/div/y": math.sqrt(2.0), "transmitter/batchnorm/add/y": O.NORM_EPS,
             "receiver/demodulation/LeakyRelu/alpha": O.LEAKY_ALPHA, "receiver/demodulation/dense_1/LeakyRelu/alpha": O.LEAKY_ALPHA,
             "mul/y": O.REG_COEFF}
    out = g.run(fetch, feed, vars_, const_override=exact)
    tf = dict(zip(fetch, out))
    scales = {n.replace("receiver/", "").replace("/Regularizer/l2_regularizer/scale", ""): float(g.const(n))
              for n in g.order if n.endswith("l2_regularizer/scale")}
    assert set(scales) == set(O.REGULARIZED)
    orc = _v1_oracle(vars_, x, bits, nbits, cp, scales, O.REG_COEFF)

    def close(a, b, what, tol=1e-12):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        assert a.shape == b.shape, (what, a.shape, b.shape)
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
        assert err <= tol, "%s: %.3e" % (what, err)

    close(orc["input"], tf["input"], "input:0 (R0)")
    close(orc["fft_out"], tf["receiver/fft_like/fft_out"], "fft_out (R1)")
    close(orc["output"], tf["output"], "output:0 (R2-R3)")
    close(orc["ce_mean"], tf["ce_mean"], "ce_mean (R4)")
    assert np.array_equal(orc["conf"], tf["conf_matrix"]), "confusion matrix (R5)"
    assert np.float32(orc["berlin"]) == np.float32(tf["linear_ber"])
    close(orc["log_ber"], tf["log_ber"], "log_ber")                   # float64 in the graph as well
    close(orc["tx_power"], tf["tx_power"], "tx_power (R8)")
    close(orc["tx_signal"], tf["tx_signal"], "tx_signal (R8)")
    close(orc["cost"], tf["cost"], "cost (R6)")
    for n in names:
        close(orc["grads"][n], tf[gmap[n]], "gradient of " + n, 1e-11)
    # under SAME padding of the size-1 axis only the centre tap ever gets a gradient
    gk = tf[gmap["fft_like/conv3d/kernel"]]
    kin = gk.shape[1]
    dead = np.delete(gk, (kin - 1) // 2, axis=1)
    assert np.all(dead == 0.0) and np.abs(gk[0, (kin - 1) // 2]).max() > 0


def test_graph_evaluator_conv_backprop_against_finite_differences():
    """The evaluator's own conv kernels (forward / filter / input backprop) are mutually consistent."""
    rng = np.random.RandomState(3)
    x = rng.randn(2, 3, 1, 2, 5)
    w = rng.randn(1, 4, 1, 5, 6)
    y = T.conv_nd(x, w, (1, 1, 1), "SAME")
    dy = rng.randn(*y.shape)
    dw = T.conv_nd_backprop_filter(x, w.shape, dy, (1, 1, 1), "SAME")
    dx = T.conv_nd_backprop_input(x.shape, w, dy, (1, 1, 1), "SAME")
    eps = 1e-6
    for _ in range(5):
        i = tuple(rng.randint(0, s) for s in w.shape)
        wp = w.copy()
        wp[i] += eps
        num = ((T.conv_nd(x, wp, (1, 1, 1), "SAME") - y) * dy).sum() / eps
        assert abs(num - dw[i]) < 1e-5 * max(1.0, abs(dw[i]))
        j = tuple(rng.randint(0, s) for s in x.shape)
        xp = x.copy()
        xp[j] += eps
        num = ((T.conv_nd(xp, w, (1, 1, 1), "SAME") - y) * dy).sum() / eps
        assert abs(num - dx[j]) < 1e-5 * max(1.0, abs(dx[j]))
