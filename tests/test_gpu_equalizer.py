"""GPU parity of the channel-equaliser stage (dev/py/model.py:349-478,
dev/py/ofdmreceiver_np_mp.py:283-330) against oracle/equalizer_oracle.py (NumPy, literal conv3d)
and oracle/torch_ref.py::LiteralEqualizer (fp64 autograd).

Tolerances (fp32 kernels vs fp64 oracle): every operator, forward and backward, 1e-5 of the reference tensor's scale
(north_star's figure).  What runs END TO END through the stage's twelve layers -- fp32 against fp64 -- is checked for
direction only (``aligned``: cosine >= 1 - 1e-6 for gradients, >= 1 - 1e-8 for activations): its max-norm distance measures
the conditioning of the chain (the 1/|h| of the equalise stage, the tanh, the cancelling bias pairs), not a kernel; the
1e-5 bar for the whole stage is held stage by stage, each on the GPU's own inputs, by tests/test_gpu_equalizer_stages.py
(all three launch plans).  Two launch plans with different summation orders are compared the same way.
"""
import numpy as np
import pytest
import torch

from oracle import dccn_oracle as O
from oracle import equalizer_oracle as E
from oracle.torch_ref import LiteralEqualizer, LiteralRx, conv2d_complex_literal

pytestmark = pytest.mark.gpu


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a), dtype=dtype).cuda()


def close(got, want, tol, what=""):
    got = got.detach().cpu().numpy().astype(np.float64) if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = max(np.abs(want).max(), 1e-30)
    err = np.abs(got - want).max() / scale
    assert err <= tol, "%s: rel err %.3e > %.1e" % (what, err, tol)


def aligned(got, want, what="", min_cos=1 - 1e-6):
    """end-to-end check: same direction (and, through it, the same scale: cos >= 1 - c bounds the relative L2 distance
    by sqrt(2c))"""
    got = got.detach().cpu().numpy().astype(np.float64) if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    got, want = got.ravel(), np.asarray(want, np.float64).ravel()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    cos = float(got @ want / max(np.linalg.norm(got) * np.linalg.norm(want), 1e-300))
    assert cos >= min_cos, "%s: cosine %.10f < %.10f" % (what, cos, min_cos)
    ratio = float(np.linalg.norm(got) / max(np.linalg.norm(want), 1e-300))
    assert abs(ratio - 1.0) <= 1e-3, "%s: norm ratio %.8f" % (what, ratio)       # (a gross scale error; cos is blind to it)


# ---- operators -------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(3, 7, 80, 2), (1, 5), (37, 1120), (2, 3000)])
def test_layer_norm(shape):
    from dl_ofdm_amd import ops
    rng = np.random.RandomState(1)
    x = (rng.standard_normal(shape) * 3 + 0.7).astype(np.float32)
    xt = dev(x).requires_grad_(True)
    y = ops.layer_norm(xt)
    close(y, E.layer_norm(x.astype(np.float64)), 1e-5, "layer_norm fwd")
    g = rng.standard_normal(shape).astype(np.float32)
    y.backward(dev(g))
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ax = tuple(range(1, xr.dim()))
    m = xr.mean(dim=ax, keepdim=True)
    v = ((xr - m) ** 2).mean(dim=ax, keepdim=True)
    ((xr - m) * torch.rsqrt(v + E.LN_EPS)).backward(torch.tensor(g, dtype=torch.float64))
    close(xt.grad, xr.grad.numpy(), 1e-5, "layer_norm bwd")


def test_tanh():
    from dl_ofdm_amd import ops
    rng = np.random.RandomState(2)
    x = (rng.standard_normal((33, 129)) * 2).astype(np.float32)
    g = rng.standard_normal(x.shape).astype(np.float32)
    xt = dev(x).requires_grad_(True)
    y = ops.tanh(xt)
    close(y, np.tanh(x.astype(np.float64)), 1e-6, "tanh fwd")
    y.backward(dev(g))
    close(xt.grad, g * (1 - np.tanh(x.astype(np.float64)) ** 2), 2e-6, "tanh bwd")


def test_equalize():
    from dl_ofdm_amd import ops
    rng = np.random.RandomState(3)
    y = rng.standard_normal((6, 7, 64, 2)).astype(np.float32)
    h = rng.standard_normal((6, 7, 64, 2)).astype(np.float32)
    h[np.abs(h).sum(-1) < 0.2] += 0.5                          # keep |h| away from the 1/|h| pole
    eq_o, corr_o = E.equalize(y.astype(np.float64), h.astype(np.float64))
    yt, ht = dev(y).requires_grad_(True), dev(h).requires_grad_(True)
    eq, corr = ops.equalize(yt, ht)
    close(eq, eq_o, 2e-6, "eq")
    close(corr, corr_o, 2e-6, "corr")
    assert float(corr.detach()[..., 1].abs().max()) == 0.0               # imaginary part cancels exactly
    g1 = rng.standard_normal(y.shape).astype(np.float32)
    g2 = rng.standard_normal(y.shape).astype(np.float32)
    (eq * dev(g1)).sum().add((corr * dev(g2)).sum()).backward()
    yr = torch.tensor(y, dtype=torch.float64, requires_grad=True)
    hr = torch.tensor(h, dtype=torch.float64, requires_grad=True)
    yc, hc = torch.view_as_complex(yr), torch.view_as_complex(hr)
    cj = torch.conj(hc)
    a = torch.abs(hc)
    e = yc * torch.complex(cj.real / a, cj.imag / a)
    c = e * torch.conj(e)
    ((torch.view_as_real(e) * torch.tensor(g1, dtype=torch.float64)).sum()
     + (torch.view_as_real(c) * torch.tensor(g2, dtype=torch.float64)).sum()).backward()
    close(yt.grad, yr.grad.numpy(), 1e-5, "d y")
    close(ht.grad, hr.grad.numpy(), 1e-5, "d h")


def test_pilot_snr():
    from dl_ofdm_amd import ops
    rng = np.random.RandomState(4)
    eq = rng.standard_normal((9, 7, 64, 2)).astype(np.float32)
    car = (4, 12, 20, 28, 35, 43, 51, 59)
    close(ops.pilot_snr(dev(eq), car), E.pilot_snr(eq.astype(np.float64), car), 1e-5, "pilot snr")


@pytest.mark.parametrize("geom", [(7, 64, 7, 64), (3, 8, 3, 8), (4, 10, 3, 6), (5, 6, 2, 5)])
def test_cconv2d_same_toeplitz(geom):
    """block-Toeplitz dense lowering == literal zero-padded conv3d, forward and all gradients"""
    from dl_ofdm_amd import ops
    L, W, kL, kW = geom
    rng = np.random.RandomState(5)
    B = 4
    x = rng.standard_normal((B, L, W, 2)).astype(np.float32)
    w = (rng.standard_normal((kL, kW, 2)) / np.sqrt(kL * kW)).astype(np.float32)
    b = rng.standard_normal(2).astype(np.float32)
    g = rng.standard_normal((B, L, W, 2)).astype(np.float32)
    xt, wt, bt = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    y = ops.cconv2d_same(xt, wt, bt)
    y.backward(dev(g))
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wr = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    br = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    yr = conv2d_complex_literal(xr.reshape(B, L, W, 1, 2), wr.reshape(kL, kW, 1, 1, 2), br, "same")[:, :, :, 0, :]
    yr.backward(torch.tensor(g, dtype=torch.float64))
    lit = O.layers_conv2d_complex_literal(x.astype(np.float64).reshape(B, L, W, 1, 2),
                                          w.astype(np.float64).reshape(kL, kW, 1, 1, 2), b.astype(np.float64),
                                          (1, 1), "same")[:, :, :, 0, :]
    close(yr, lit, 1e-12, "torch literal vs numpy literal")
    close(y, lit, 1e-5, "fwd")
    close(xt.grad, xr.grad.numpy(), 1e-5, "dx")
    close(wt.grad, wr.grad.numpy(), 1e-5, "dw")
    close(bt.grad, br.grad.numpy(), 1e-5, "db")


# ---- the stage -------------------------------------------------------------------------------------
class _Flags:
    nfft, nsymbol, nbits, npilot, nguard, nfilter = 64, 7, 2, 8, 8, 64
    cp, longcp, pilot, channel = True, True, "lte", "EPA"


def _store_shape(name, shp):
    return (shp[0], shp[1], shp[3], shp[4]) if len(shp) == 5 else shp


def build_stage(nbits=2, cp=True, seed=11):
    from dl_ofdm_amd.complex import VariableStore
    from dl_ofdm_amd.model import OfdmDenseRx, equalizer_ofdm
    from dl_ofdm_amd.ofdm import ofdm_tx
    F = _Flags()
    F.nbits, F.cp = nbits, cp
    tx = ofdm_tx(F)
    ecfg = E.EqConfig(S=7, K=tx.K, CP=tx.CP, cp=cp, pilot_size=tx.pilot_size,
                      pilot_carriers=tuple(int(v) for v in tx.pilotCarriers))
    rcfg = O.RxConfig(S=7, kin=ecfg.n_sc, F=64, D=tx.frame_size, nbits=nbits)
    pe = E.init_params(ecfg, seed=seed, bias_scale=0.05)
    pr = O.init_params(rcfg, seed=seed + 1)
    rng = np.random.RandomState(seed)
    for n in pr:                                            # non-zero biases in the frozen receiver too
        if n.endswith("bias"):
            pr[n] = (0.05 * rng.standard_normal(pr[n].shape)).astype(np.float32)
    rx = OfdmDenseRx(F, tx, seed=1)
    rx.import_params(pr)
    for p_ in rx.store.parameters():
        p_.requires_grad_(False)
    eq_store = VariableStore(seed=2)

    def run_eq(x_norm):
        eq_store.begin()
        with eq_store.scope("Equalizer"):
            return equalizer_ofdm(x_norm, F, tx, scope=eq_store)
    run_eq(torch.randn(2, 7, 64 + 16, 2, device="cuda"))                 # create the variables
    assert [n for n in eq_store.names()] == list(E.param_shapes(ecfg).keys())
    for n, shp in E.param_shapes(ecfg).items():
        assert tuple(eq_store.tensor(n).shape) == _store_shape(n, shp), n
        eq_store.set(n, pe[n])
    return F, tx, ecfg, rcfg, pe, pr, rx, eq_store, run_eq


@pytest.mark.parametrize("nbits,B", [(2, 6), (1, 3), (4, 9)])
def test_equalizer_forward_and_gradients(nbits, B):
    from dl_ofdm_amd import ops
    from dl_ofdm_amd.model import ofdm_dense_rx, regularization_loss
    F, tx, ecfg, rcfg, pe, pr, rx, eq_store, run_eq = build_stage(nbits)
    rng = np.random.RandomState(100 + nbits)
    x = (rng.standard_normal((B, 7, 80, 2)) * 2).astype(np.float32)
    bits = rng.randint(0, 2, (B, tx.frame_size, nbits)).astype(np.int32)

    lit_rx = LiteralRx({k: v.astype(np.float64) for k, v in pr.items()}, rcfg, dtype=torch.float64, literal_conv=False)
    lit = LiteralEqualizer({k: v.astype(np.float64) for k, v in pe.items()}, lit_rx, ecfg)
    g_ref, info = lit.forward_backward(x.astype(np.float64), bits)
    out_np, snr_np, h_np = E.equalizer_forward({k: v.astype(np.float64) for k, v in pe.items()}, info["x_norm"], ecfg)
    close(out_np, info["out_eq"], 1e-11, "numpy vs torch oracle")

    x_norm = ops.batch_moment_norm(dev(x))
    out_eq, snr_db, chest = run_eq(x_norm)
    aligned(out_eq, info["out_eq"], "equalized", 1 - 1e-8)          # end to end (stage by stage: test_gpu_equalizer_stages.py)
    aligned(torch.view_as_real(chest), info["chest"], "chest", 1 - 1e-8)
    aligned(snr_db, snr_np, "snr_db", 1 - 1e-8)
    rx.store.begin()
    prob, ce, mbuf, _ = ofdm_dense_rx(out_eq, F, tx, rx.outshape, scope=rx.store, bits=dev(bits, torch.int32))
    aligned(prob, info["prob"], "prob", 1 - 1e-8)
    loss = ce + E.EQ_REG_COEFF * (regularization_loss(eq_store, "Equalizer") + regularization_loss(rx.store))
    assert abs(float(loss.detach()) - info["loss"]) <= 2e-6 * abs(info["loss"])
    assert abs(float(ce.detach()) - info["ce_mean"]) <= 2e-6 * abs(info["ce_mean"])
    loss.backward()
    m = ops.read_metrics(mbuf)
    assert np.array_equal(np.asarray(m["conf"]).reshape(2, 2), info["conf"])
    for n in E.param_shapes(ecfg):
        got = eq_store.tensor(n).grad.detach().cpu().numpy().astype(np.float64).ravel()
        want = g_ref[n].ravel()
        cos = float(got @ want / (np.linalg.norm(got) * np.linalg.norm(want)))
        assert cos >= 1 - 1e-6, (n, cos)
    assert all(p_.grad is None for p_ in rx.store.parameters())           # receiver stays frozen


def test_equalizer_without_cp():
    """FLAGS.cp=False: the stage drops the cyclic prefix itself (model.py:364-366) but still maps back
    to the input width (:458)."""
    from dl_ofdm_amd.complex import VariableStore
    from dl_ofdm_amd.model import equalizer_ofdm
    from dl_ofdm_amd.ofdm import ofdm_tx
    F = _Flags()
    tx = ofdm_tx(F)
    F.cp = False
    st = VariableStore(seed=5)
    x = dev(np.random.RandomState(41).standard_normal((3, 7, 80, 2)) * 1.5)       # fixed input (not torch's global RNG)
    with st.scope("Equalizer"):
        out, snr, chest = equalizer_ofdm(x, F, tx, scope=st)
    assert out.shape == (3, 7, 80, 2) and snr.shape == (3, 1) and chest.shape == (3, 7, 64)
    c = E.EqConfig(S=7, K=64, CP=16, cp=False, pilot_size=tx.pilot_size,
                   pilot_carriers=tuple(int(v) for v in tx.pilotCarriers))
    p = {n: st.tensor(n).detach().cpu().numpy().astype(np.float64).reshape(s) for n, s in E.param_shapes(c).items()}
    o_out, o_snr, o_h = E.equalizer_forward(p, x.cpu().numpy().astype(np.float64), c)
    aligned(out, o_out, "equalized (no cp)", 1 - 1e-8)
    aligned(torch.view_as_real(chest), o_h, "chest (no cp)", 1 - 1e-8)


# ---- the transfer-learning step (ofdmreceiver_np_mp.py:319-330) -----------------------------------------
def _trainer(nbits=2, seed=21, cp=True, nfft=64, longcp=True):
    from dl_ofdm_amd.equalizer import EqualizerTrainer
    from dl_ofdm_amd.ofdm import ofdm_tx
    F = _Flags()
    F.nbits, F.opt, F.init_learning, F.cp = nbits, 0, 1e-3, cp
    F.nfft, F.nfilter, F.longcp = nfft, nfft, longcp
    tx = ofdm_tx(F)
    ecfg = E.EqConfig(S=7, K=tx.K, CP=tx.CP, cp=cp, pilot_size=tx.pilot_size,
                      pilot_carriers=tuple(int(v) for v in tx.pilotCarriers))
    rcfg = O.RxConfig(S=7, kin=tx.K + tx.CP if cp else tx.K, F=nfft, D=tx.frame_size, nbits=nbits)
    pe = E.init_params(ecfg, seed=seed, bias_scale=0.05)
    pr = O.init_params(rcfg, seed=seed + 1)
    tr = EqualizerTrainer(F, tx, pr, seed=3)
    assert tr.names == list(E.param_shapes(ecfg).keys())
    if nfft == 64 and longcp:
        assert tr.n_params == (1_753_282 if cp else 1_753_282 - 32 * 128)
    tr.load_params(pe)
    return F, tx, ecfg, rcfg, pe, pr, tr


@pytest.mark.parametrize("mode", ["fused-graph", "fused-eager", "composed"])
def test_trainer_step_gradients_and_adam(mode):
    fused, graph = mode != "composed", mode == "fused-graph"
    F, tx, ecfg, rcfg, pe, pr, tr = _trainer()
    rng = np.random.RandomState(7)
    B = 8
    st = O.adam_init(pe)
    p_or = {k: v.copy() for k, v in pe.items()}
    for step in range(3):
        x = (rng.standard_normal((B, 7, 80, 2)) * 2).astype(np.float32)
        bits = rng.randint(0, 2, (B, tx.frame_size, 2)).astype(np.int32)
        p_before = tr.get_params()
        lit_rx = LiteralRx({k: v.astype(np.float64) for k, v in pr.items()}, rcfg, dtype=torch.float64,
                           literal_conv=False)
        lit = LiteralEqualizer({k: v.astype(np.float64).reshape(pe[k].shape) for k, v in p_before.items()}, lit_rx, ecfg)
        g_ref, info = lit.forward_backward(x.astype(np.float64), bits)
        m = tr.train_step(x, bits, fused=fused, graph=graph)
        assert abs(m["ce_mean"] - info["ce_mean"]) <= 3e-6 * abs(info["ce_mean"])
        assert np.array_equal(np.asarray(m["conf"]).reshape(2, 2), info["conf"])
        assert abs(tr.total_loss(m) - info["loss"]) <= 5e-3 * abs(info["loss"])      # reg term read after the update: loose
        g_gpu = tr.get_grads()
        used = {}
        for n in tr.names:
            reg = E.EQ_REG_COEFF * 2 * O.REG_L2 * p_before[n].astype(np.float64) if "/dense" in n else 0.0
            got = (g_gpu[n].astype(np.float64) + reg).ravel()
            want = g_ref[n].ravel()
            cos = float(got @ want / (np.linalg.norm(got) * np.linalg.norm(want)))
            assert cos >= 1 - 1e-6, (step, n, cos)
            used[n] = (g_gpu[n] + np.float32(E.EQ_REG_COEFF * 2 * O.REG_L2) * p_before[n]).astype(np.float32) \
                if "/dense" in n else g_gpu[n]
        # the optimizer itself: TF ApplyAdam on the GPU's own gradients, from the GPU's own parameters
        for k in p_or:
            p_or[k] = p_before[k].reshape(pe[k].shape).copy()
        alpha = O.adam_tf_step(p_or, {k: used[k].reshape(pe[k].shape) for k in p_or}, st)
        a = tr.adam()
        assert a["global_step"] == step + 1 and abs(a["alpha"] - float(alpha)) <= 1e-6 * float(alpha)
        p_after = tr.get_params()
        for n in tr.names:
            d_gpu = (p_after[n] - p_before[n]).ravel().astype(np.float64)
            d_or = (p_or[n].ravel() - p_before[n].ravel()).astype(np.float64)
            # |update| <= ~lr; agreement to 1e-3 of lr except where g ~ 0 flips m/sqrt(v) (measure by quantile)
            err = np.abs(d_gpu - d_or)
            assert np.quantile(err, 0.999) <= 2e-6, (step, n, np.quantile(err, 0.999))
            m_or, m_gpu = st.m[n].ravel(), tr.view(n, tr.adam_m).detach().cpu().numpy().ravel()
            assert np.abs(m_or - m_gpu).max() <= 1e-6 * max(np.abs(m_or).max(), 1e-30) + 1e-12


def test_trainer_checkpoint_roundtrip_and_unsupported_variant(tmp_path):
    from dl_ofdm_amd import receiver_mp as H
    from dl_ofdm_amd.equalizer import EqualizerTrainer
    F, tx, ecfg, rcfg, pe, pr, tr = _trainer()
    rng = np.random.RandomState(8)
    x = (rng.standard_normal((5, 7, 80, 2)) * 2).astype(np.float32)
    bits = rng.randint(0, 2, (5, tx.frame_size, 2)).astype(np.int32)
    tr.train_step(x, bits)
    hf = H.Flags(nbits=2, nfilter=64, token="T", channel="EPA", save_dir=str(tmp_path))
    path = H.save_checkpoint(str(tmp_path / H.save_model_name(hf)), tr, hf)
    z = np.load(path + ".npz")
    for key in ("Equalizer/dense_3/kernel", "optimizer/Equalizer/dense_3/kernel/Adam",
                "optimizer/Equalizer/conv3d_1/bias/Adam_1", "optimizer/global_step", "optimizer/beta1_power"):
        assert key in z.files
    e1 = tr.eval_step(x, bits)
    tr2 = EqualizerTrainer(F, tx, pr, seed=99)
    H.load_checkpoint(path, tr2)
    e2 = tr2.eval_step(x, bits)
    assert e1["ce_sum"] == e2["ce_sum"] and e1["conf"] == e2["conf"]
    assert tr2.adam()["global_step"] == 1.0
    assert torch.equal(tr.adam_v, tr2.adam_v)
    F.opt = 3
    with pytest.raises(NotImplementedError):
        EqualizerTrainer(F, tx, pr)


def test_equalizer_learns_a_flat_fading_channel():
    """End to end through the harness pieces: a receiver trained on AWGN cannot demodulate a random
    phase rotation (flat Rayleigh tap); a few hundred equaliser steps must cut its BER."""
    from dl_ofdm_amd import receiver as R, receiver_mp as H
    from dl_ofdm_amd import ofdm
    base = R.Flags(nbits=2, nfilter=64, channel="AWGN", SNR=10.0, msg_length=7 * 4096, batch_size=512,
                   max_epoch_num=6, early_stop=100, token="B", save_dir="/tmp/_eq_test/", seed=5)
    res = R.train(base, verbose=False, run_test=False)
    hf = H.Flags(nbits=2, nfilter=64, channel="Flat", msg_length=7 * 2048, batch_size=512, max_epoch_num=5,
                 early_stop=100, token="B", save_dir="/tmp/_eq_test/", seed=6, eval_frames=2048)
    out = H.train(hf, verbose=False, run_test=False, rx_params=res["params"])
    hist = out["history"]
    assert hist[-1]["train_loss"] < hist[0]["train_loss"] - 0.02, hist
    tr = out["trainer"]
    tx = ofdm.ofdm_tx(hf)
    np.random.seed(11)
    xs, ys, _, _ = H.make_batch(hf, tx, H.RayleighChanParallel(hf, tx.Fs), 2048, 20.0)
    with_eq = tr.eval_step(xs, ys)["berlin"]
    from dl_ofdm_amd.engine import RxEngine
    eng = RxEngine(R.rx_dims(base, tx), 2048, train=False, params=res["params"], want_prob=False)
    eng.eval_step(xs, ys)
    without = eng.metrics()["berlin"]
    assert without > 0.25, without                       # random phase: the bare receiver is lost
    assert with_eq < without - 0.05, (with_eq, without)
    sw = H.test_model_cross(H.Flags(**{**hf.__dict__, "test_frames": 256, "snr_lo": 0, "snr_hi": 20, "snr_step": 10}),
                            tr, tx, out_dir="/tmp/_eq_test", verbose=False, channels=("Flat", "EPA"))
    import os
    assert os.path.basename(sw["EPA"][3]) == "Test_DCCN_B_Equalizer0_Flat_test_chan_EPA.csv" and os.path.exists(sw["EPA"][3])
    assert len(sw["Flat"][1]) == 3


@pytest.mark.parametrize("cp", [True, False])
def test_fused_step_equals_composed_step(cp):
    """the planned launch sequence and the autograd-composed one compute the same step on the same arenas
    (cp=False: both read the post-CP window, model.py:364-366 / 1236-1240).  Since round 3 the planned step runs the pilot
    bottleneck as one launch per direction (eq_bottleneck.h, its own summation order) where the composed path runs GEMMs:
    the channel-estimator branch's gradients are ill-conditioned in fp32 -- either path is 5e-5..1e-4 of scale away from the
    fp64 autograd reference -- so the two are compared at IDENTICAL parameters every step, by direction
    (with the fused bottleneck switched off, tuning key 20 = 3, they agree to the last bit:
    test_round3_launch_plan_equals_round2_plan)"""
    F, tx, ecfg, rcfg, pe, pr, tr_a = _trainer(seed=31, cp=cp)
    _, _, _, _, _, _, tr_b = _trainer(seed=31, cp=cp)
    rng = np.random.RandomState(9)
    for step in range(4):
        x = (rng.standard_normal((12, 7, 80, 2)) * 2).astype(np.float32)
        bits = rng.randint(0, 2, (12, tx.frame_size, 2)).astype(np.int32)
        chan = (rng.standard_normal((12, 7, 64)) + 1j * rng.standard_normal((12, 7, 64))).astype(np.complex64)
        with torch.no_grad():                    # same parameters and optimizer state on both sides before the step
            for name in ("params", "adam_m", "adam_v", "adam_state"):
                getattr(tr_b, name).copy_(getattr(tr_a, name))
        ma = tr_a.train_step(x, bits, chan, fused=True, graph=(step % 2 == 0))
        mb = tr_b.train_step(x, bits, chan, fused=False)
        assert ma["conf"] == mb["conf"] and abs(ma["ce_mean"] - mb["ce_mean"]) <= 1e-6
        assert abs(ma["chan_rms"] - mb["chan_rms"]) <= 1e-5 * abs(mb["chan_rms"])
        assert abs(ma["tx_power"] - mb["tx_power"]) <= 1e-6 * abs(mb["tx_power"])
        ga, gb = tr_a.grads.cpu().numpy(), tr_b.grads.cpu().numpy()
        aligned(ga, gb, "gradient arena, fused vs composed")
    with torch.no_grad():
        for name in ("params", "adam_m", "adam_v", "adam_state"):
            getattr(tr_b, name).copy_(getattr(tr_a, name))
    ea, eb = tr_a.eval_step(x, bits, fused=True), tr_b.eval_step(x, bits, fused=False)
    assert ea["conf"] == eb["conf"] and abs(ea["ce_mean"] - eb["ce_mean"]) <= 1e-6
    pl = tr_a._plan(12)
    out_b = tr_b._forward(x, bits)
    # (end to end, two summation orders: the planned step runs the row-strip and bottleneck kernels where the composed path runs
    # GEMM launches; each is held to 1e-5 per stage by test_gpu_equalizer_stages.py)
    aligned(pl.out_eq, out_b[5].detach().cpu().numpy(), "out_eq", 1 - 1e-8)
    aligned(pl.snr_db, out_b[3].detach().cpu().numpy(), "snr_db", 1 - 1e-8)
    aligned(pl.chest, torch.view_as_real(out_b[4]).cpu().numpy(), "chest", 1 - 1e-8)


@pytest.mark.parametrize("plan", [1, 3])
@pytest.mark.parametrize("B,cp", [(12, True), (73, True), (73, False), (200, True), (200, False)])
def test_round3_launch_plan_equals_round2_plan(B, cp, plan):
    """dccn_set_tuning(20, .): the re-planned step (grouped corr/eq C-Convs, concat / split in GEMM stores, merged
    element-wise launches, one job-table optimizer launch) computes what the launch-per-stage plan computes -- at the
    few-row batch (direct weight gradients), the reference's 73 frames and a batch whose dense gradients are split-K slabs.
    plan 3 = the re-plan without the fused pilot bottleneck: same GEMM plans and summation orders, gradients of the first
    step bit-identical; plan 1 (default) also runs the bottleneck as one launch per direction (its own summation order)"""
    from dl_ofdm_amd import _lib
    lib = _lib.load()
    F, tx, ecfg, rcfg, pe, pr, tr_a = _trainer(seed=41, cp=cp)
    _, _, _, _, _, _, tr_b = _trainer(seed=41, cp=cp)
    rng = np.random.RandomState(19)
    try:
        for step in range(3):
            x = (rng.standard_normal((B, 7, 80, 2)) * 2).astype(np.float32)
            bits = rng.randint(0, 2, (B, tx.frame_size, 2)).astype(np.int32)
            with torch.no_grad():                # same parameters and optimizer state on both sides before the step
                for name in ("params", "adam_m", "adam_v", "adam_state"):
                    getattr(tr_b, name).copy_(getattr(tr_a, name))
            lib.dccn_set_tuning(20, plan)
            ma = tr_a.train_step(x, bits, fused=True, graph=(step == 1))
            lib.dccn_set_tuning(20, 0)
            mb = tr_b.train_step(x, bits, fused=True, graph=False)
            assert ma["conf"] == mb["conf"] and abs(ma["ce_mean"] - mb["ce_mean"]) <= 1e-6 * abs(mb["ce_mean"])
            assert abs(ma["tx_power"] - mb["tx_power"]) <= 1e-6 * abs(mb["tx_power"])
            ga, gb = tr_a.get_grads(), tr_b.get_grads()
            if plan == 3:                        # same GEMM plans, same summation orders -> same bits, every step
                diff = {n: float(np.abs(ga[n] - gb[n]).max() / max(np.abs(gb[n]).max(), 1e-30)) for n in tr_a.names
                        if not np.array_equal(ga[n], gb[n])}
                assert not diff, " ".join("%s=%.1e" % (k.split("/", 1)[1], v) for k, v in diff.items())
            # plan 1: the fused bottleneck sums in its own order; the channel-estimator branch is ill-conditioned in fp32 (each
            # plan is held to 1e-5 per stage by test_gpu_equalizer_stages.py; the two are compared by direction here)
            # (two float32 evaluation orders of the same chain -- row-strip and bottleneck kernels against GEMM launches -- each
            # within cos 1 - 1e-6 of the float64 oracle per parameter (test_trainer_step_gradients_and_adam); against each other
            # the distances add: measured 1 - 6.3e-6 on the concatenated gradient at 73 frames)
            aligned(np.concatenate([ga[n].ravel() for n in tr_a.names]), np.concatenate([gb[n].ravel() for n in tr_a.names]),
                    "gradients, plan %d vs plan 0" % plan, 1 - 2e-5)
            pa, pb = tr_a.get_params(), tr_b.get_params()
            for n in tr_a.names:
                d = np.abs(pa[n] - pb[n]).ravel()
                assert np.quantile(d, 0.99) <= (2e-6 if plan == 3 else 1e-5), (step, n, d.max())       # (lr = 1e-3)
        pl_a, pl_b = tr_a._plan(B), tr_b._plan(B)
        for name in ("out_eq", "snr_db", "chest"):
            if plan == 3:
                close(getattr(pl_a, name), getattr(pl_b, name).cpu().numpy(), 2e-6, name)
            else:
                aligned(getattr(pl_a, name), getattr(pl_b, name).cpu().numpy(), name, 1 - 1e-8)
        with torch.no_grad():
            tr_b.params.copy_(tr_a.params)
        lib.dccn_set_tuning(20, plan)
        ea = tr_a.eval_step(x, bits, fused=True)
        lib.dccn_set_tuning(20, 0)
        eb = tr_b.eval_step(x, bits, fused=True)
        assert ea["conf"] == eb["conf"]
    finally:
        lib.dccn_set_tuning(20, 1)


@pytest.mark.parametrize("B", [12, 200])
def test_launch_plan_with_unaligned_gradient_slabs(B):
    """nfft=128 with the short prefix (CP = 9: rows of 2(K+CP) = 274 floats, 274 % 4 = 2): split-K bias / kernel slabs that
    are not 16 bytes apart take the optimizer launch's element-wise branch (eq_opt.h eq_opt_sum, J.vec == 0).  Every element
    must be summed and Adam-updated exactly once: the re-planned step against the launch-per-stage plan, three steps, and the
    step must repeat bit for bit (a quad walked by more than one thread would race)."""
    from dl_ofdm_amd import _lib
    lib = _lib.load()
    F, tx, ecfg, rcfg, pe, pr, tr_a = _trainer(seed=43, nfft=128, longcp=False)
    _, _, _, _, _, _, tr_b = _trainer(seed=43, nfft=128, longcp=False)
    _, _, _, _, _, _, tr_c = _trainer(seed=43, nfft=128, longcp=False)
    assert tx.CP == 9 and (2 * (tx.K + tx.CP)) % 4 == 2
    rng = np.random.RandomState(23)
    try:
        for step in range(3):
            x = (rng.standard_normal((B, 7, tx.K + tx.CP, 2)) * 2).astype(np.float32)
            bits = rng.randint(0, 2, (B, tx.frame_size, 2)).astype(np.int32)
            with torch.no_grad():
                for name in ("params", "adam_m", "adam_v", "adam_state"):
                    getattr(tr_b, name).copy_(getattr(tr_a, name))
                    getattr(tr_c, name).copy_(getattr(tr_a, name))
            lib.dccn_set_tuning(20, 3)
            tr_a.train_step(x, bits, fused=True, graph=False)
            tr_c.train_step(x, bits, fused=True, graph=False)
            lib.dccn_set_tuning(20, 0)
            tr_b.train_step(x, bits, fused=True, graph=False)
            ga, gb, gc = tr_a.get_grads(), tr_b.get_grads(), tr_c.get_grads()
            pa, pb, pc = tr_a.get_params(), tr_b.get_params(), tr_c.get_params()
            for n in tr_a.names:
                assert np.array_equal(ga[n], gc[n]) and np.array_equal(pa[n], pc[n]), (step, n, "not repeatable")
                assert np.array_equal(ga[n], gb[n]), (step, n, float(np.abs(ga[n] - gb[n]).max()))
                assert float(np.abs(pa[n] - pb[n]).max()) <= 2e-6, (step, n, float(np.abs(pa[n] - pb[n]).max()))   # (lr = 1e-3)
            dm = (tr_a.adam_m - tr_b.adam_m).abs().max() / tr_b.adam_m.abs().max()
            assert float(dm) <= 1e-6, float(dm)
    finally:
        lib.dccn_set_tuning(20, 1)


@pytest.mark.parametrize("B,SK2,P", [(73, 896, 32), (12, 896, 32), (200, 896, 16), (33, 128, 32), (16, 64, 16)])
def test_pilot_bottleneck_one_launch_per_direction(B, SK2, P):
    """dccn_eq_bottleneck_fwd / _bwd (model.py:394-412: dense SK2 -> P -> SK2 without an activation) against fp64 NumPy on
    well-conditioned data: 1e-5 of scale, forward and all six backward outputs (ragged last row tile, one or two MFMA tiles
    across P, a single row tile)"""
    from dl_ofdm_amd import _lib
    lib = _lib.load()
    assert lib.dccn_eq_bottleneck_supported(B, SK2, P) == 1 and lib.dccn_eq_bottleneck_supported(B, SK2, 24) == 0
    rng = np.random.RandomState(B + P)
    y = rng.standard_normal((B, SK2)); W1 = rng.standard_normal((SK2, P)) / np.sqrt(SK2); b1 = rng.standard_normal(P) * 0.1
    W2 = rng.standard_normal((P, SK2)) / np.sqrt(P); b2 = rng.standard_normal(SK2) * 0.1
    dd2 = rng.standard_normal((B, SK2)); dy = rng.standard_normal((B, SK2))
    t = {k: dev(v) for k, v in dict(y=y, W1=W1, b1=b1, W2=W2, b2=b2, dd2=dd2, dy=dy).items()}
    d1 = torch.empty(B, P, device="cuda"); d2 = torch.empty(B, SK2, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    p = lambda a: a.data_ptr()
    _lib.check(lib.dccn_eq_bottleneck_fwd(p(t["y"]), p(t["W1"]), p(t["b1"]), p(t["W2"]), p(t["b2"]), p(d1), p(d2), B, SK2, P, st))
    d1r = y @ W1 + b1
    close(d1, d1r, 1e-5, "d1")
    close(d2, d1r @ W2 + b2, 1e-5, "d2")
    nws = lib.dccn_eq_bottleneck_workspace_size(B, SK2, P)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    out = {k: torch.empty(*shp, device="cuda") for k, shp in dict(dy=(B, SK2), dW1=(SK2, P), db1=(P,), dW2=(P, SK2), db2=(SK2,)).items()}
    _lib.check(lib.dccn_eq_bottleneck_bwd(p(t["dd2"]), p(d1), p(t["y"]), p(t["W1"]), p(t["W2"]), p(t["dy"]), p(out["dy"]),
                                          p(out["dW1"]), p(out["db1"]), p(out["dW2"]), p(out["db2"]), B, SK2, P, p(ws), nws, st))
    d1f = d1.cpu().numpy().astype(np.float64)
    dd1 = dd2 @ W2.T
    close(out["dy"], dy + dd1 @ W1.T, 1e-5, "dy")
    close(out["dW2"], d1f.T @ dd2, 1e-5, "dW2")
    close(out["db2"], dd2.sum(0), 1e-5, "db2")
    close(out["dW1"], y.T @ dd1, 1e-5, "dW1")
    close(out["db1"], dd1.sum(0), 1e-5, "db1")


def test_chan_rms_monitor_is_keras_layer_normalization_over_the_symbol_axis():
    """ofdmreceiver_np_mp.py:245, 325-333: LayerNormalization(axis=1, center=False, scale=False), epsilon 1e-3"""
    F, tx, ecfg, rcfg, pe, pr, tr = _trainer(seed=3, cp=True)
    rng = np.random.RandomState(4)
    gt = (rng.standard_normal((6, 7, 64)) + 1j * rng.standard_normal((6, 7, 64))).astype(np.complex64)
    est = (rng.standard_normal((6, 7, 64)) + 1j * rng.standard_normal((6, 7, 64))).astype(np.complex64)

    def keras_ln(c):
        t = np.stack([c.real, c.imag], -1).astype(np.float64)                  # [B, S, K, 2]
        mean = t.mean(axis=1, keepdims=True)
        var = t.var(axis=1, keepdims=True)
        return (t - mean) / np.sqrt(var + 1e-3)
    want = float(((keras_ln(gt) - keras_ln(est)) ** 2).mean())
    got = float(tr.chan_rms(torch.as_tensor(est).cuda(), gt))
    assert abs(got - want) <= 1e-5 * want


@pytest.mark.parametrize("cp", [True, False])
def test_frozen_receiver_folds_into_one_matrix(cp):
    """dccn_eq_rx_fold: Mf / bf with out_eq_flat . Mf + bf == dense(C-Conv(out_eq)) of the frozen receiver (model.py:1246-1275)
    against float64 NumPy; cp = False reads the window behind the cyclic prefix (zero rows for the prefix samples)."""
    import ctypes as C
    from dl_ofdm_amd import _lib
    lib = _lib.load()
    S, K, CP, F, D = 7, 64, 16, 64, 320
    kin, N2, F2, dN = (K + CP if cp else K), 2 * (K + CP), 2 * F, 2 * D
    rng = np.random.RandomState(3 + cp)
    cw = rng.standard_normal((kin, F2)).astype(np.float32)
    cb = rng.standard_normal(F2).astype(np.float32)
    wd = (rng.standard_normal((S * F2, dN)) / 30).astype(np.float32)
    bd = rng.standard_normal(dN).astype(np.float32)
    tail = np.zeros(40, np.float32)
    arena = dev(np.concatenate([cw.ravel(), cb, wd.ravel(), bd, tail]))
    sh = _lib.EqShape(5, S, K, CP, 1 if cp else 0, F, D, 2, 8, 8)
    n = lib.dccn_eq_rx_folded_floats(C.byref(sh))
    assert n == S * N2 * dN + dN
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    _lib.check(lib.dccn_eq_rx_fold(C.byref(sh), arena.data_ptr(), out.data_ptr(),
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)), "fold")
    Mf = out[:S * N2 * dN].view(S * N2, dN).cpu().numpy().astype(np.float64)
    bf = out[S * N2 * dN:].cpu().numpy().astype(np.float64)
    # reference: push a random frame through the two layers in float64
    Wa, Wb = cw[:, :F].astype(np.float64), cw[:, F:].astype(np.float64)
    Weff = np.zeros((2 * kin, F2))
    Weff[0::2, 0::2], Weff[0::2, 1::2], Weff[1::2, 0::2], Weff[1::2, 1::2] = Wa, Wb, -Wb, -Wa
    ba, bb = cb[:F].astype(np.float64), cb[F:].astype(np.float64)
    cbe = np.empty(F2)
    cbe[0::2], cbe[1::2] = ba - bb, bb - ba
    x = rng.standard_normal((4, S, N2))
    win = 0 if cp else 2 * CP
    fft = x[:, :, win:win + 2 * kin] @ Weff + cbe                      # [4, S, 2F]
    z = fft.reshape(4, S * F2) @ wd.astype(np.float64) + bd
    got = x.reshape(4, S * N2) @ Mf + bf
    assert np.abs(got - z).max() <= 2e-6 * np.abs(z).max()
    if not cp:
        assert not Mf.reshape(S, N2, dN)[:, :win].any()


@pytest.mark.parametrize("B,per_symbol", [(6, 1), (73, 0), (73, 1), (1024, 0)])
def test_monitor_launch_equals_the_framework_monitors(B, per_symbol):
    """dccn_eq_monitor_accumulate: chan_rms (same Keras LayerNormalization as above, fp64 NumPy as the judge; a static
    channel is one row per frame whose normalisation is exactly zero) and the epoch accumulators
    {ce_mean, berlin, tx_power, noise_power, chan_rms} over three calls, in one launch per step."""
    import ctypes as C
    from dl_ofdm_amd import _lib
    lib = _lib.load()
    S, K = 7, 64
    rng = np.random.RandomState(B + per_symbol)
    est = (rng.standard_normal((B, S, K, 2)) * 0.7 + 0.1).astype(np.float32)
    gt = (rng.standard_normal((B, S, K, 2) if per_symbol else (B, K, 2))).astype(np.float32)

    def keras_ln(t):
        t = t.astype(np.float64)
        return (t - t.mean(axis=1, keepdims=True)) / np.sqrt(t.var(axis=1, keepdims=True) + 1e-3)
    gfull = gt if per_symbol else np.broadcast_to(gt[:, None], (B, S, K, 2))
    want = float(((keras_ln(gfull) - keras_ln(est)) ** 2).mean())
    nws = lib.dccn_eq_monitor_workspace_size(B, S, K)
    ws = torch.zeros(nws, dtype=torch.uint8, device="cuda")
    acc = torch.zeros(5, dtype=torch.float32, device="cuda")
    rms = torch.zeros(1, dtype=torch.float32, device="cuda")
    m = np.zeros(16, np.float32)
    m[12], m[13] = 0.625, 0.03125                                             # dccn_metrics.ce_mean / berlin
    mbuf = torch.as_tensor(m).cuda()
    txp, npw = torch.tensor([1.5], device="cuda"), torch.tensor([0.25], device="cuda")
    e, g = dev(est), dev(gt)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        _lib.check(lib.dccn_eq_monitor_accumulate(e.data_ptr(), g.data_ptr(), per_symbol, B, S, K, mbuf.data_ptr(),
                                                  txp.data_ptr(), npw.data_ptr(), acc.data_ptr(), rms.data_ptr(),
                                                  ws.data_ptr(), nws, st), "monitor")
    torch.cuda.synchronize()
    assert abs(float(rms) - want) <= 2e-6 * max(want, 1e-3)
    got = acc.cpu().numpy()
    np.testing.assert_allclose(got, 3 * np.array([0.625, 0.03125, 1.5, 0.25, want]), rtol=3e-6)
    assert int(ws[:4].view(torch.int32)) == 0                                  # arrival counter left ready for the next call
    # the framework-side monitor the host-data path keeps using gives the same number
    F, tx, ecfg, rcfg, pe, pr, tr = _trainer(seed=3, cp=True)
    ref = float(tr.chan_rms(torch.view_as_complex(e), torch.view_as_complex(dev(np.ascontiguousarray(gfull)))))
    assert abs(ref - float(rms)) <= 1e-5 * max(want, 1e-3)


def test_generator_on_a_side_stream_trains_the_same_equaliser():
    """receiver_mp.DeviceEpochLoop: two resident buffer sets and the generator on its own HIP stream (batch i+1 produced while
    step i runs) against the one-stream loop: same seeds -> history and trained parameters identical bit for bit over 3 epochs
    of 28 steps (a missed dependency trains on a half-written batch, stale labels or the wrong SNR row)."""
    from dl_ofdm_amd import receiver as R, receiver_mp as H
    from dl_ofdm_amd.engine import glorot_init
    from dl_ofdm_amd import ofdm
    out = []
    for overlap in (False, True):
        hf = H.Flags(nbits=2, nfilter=64, channel="EPA", msg_length=7 * 2048, batch_size=512, max_epoch_num=3,
                     early_stop=100, token="OV", save_dir="/tmp/_eq_overlap_%d/" % overlap, seed=6, eval_frames=512,
                     device_data=True, overlap_generator=overlap)
        o = ofdm.ofdm_tx(hf)
        res = H.train(hf, verbose=False, run_test=False, rx_params=glorot_init(R.rx_dims(hf, o), 1))
        out.append((res["history"], res["trainer"].params.detach().clone()))
    assert out[0][0] == out[1][0]
    assert torch.equal(out[0][1], out[1][1])


def test_next_batch_normalised_on_the_optimizer_launch_trains_the_same_equaliser():
    """include/dccn.h dccn_eq_buffers.x_next / x_prenormalised / norm_slot: step i normalises batch i+1 on leading workgroups
    of its optimizer launch and step i+1 starts at the layer norm (receiver_mp.DeviceEpochLoop pipeline, twin plans sharing
    one workspace).  (1) the C ABI directly: two pipelined steps == two plain steps, bit for bit (parameters, Adam state,
    metrics, tx_power of both steps); (2) the harness: 3 epochs of 28 steps (an odd and an even step count both start every
    epoch on plan 0) with and without the pipeline -> identical history and parameters."""
    from dl_ofdm_amd import receiver as R, receiver_mp as H
    from dl_ofdm_amd.engine import glorot_init
    import ctypes as C
    from dl_ofdm_amd.equalizer import _FusedPlan
    from dl_ofdm_amd import ofdm
    F, tx, ecfg, rcfg, pe, pr, tr = _trainer(seed=23)
    assert tr.lib.dccn_eq_norm_rides(C.byref(tr.resident(9).shape)) == 1
    rng = np.random.RandomState(5)
    xs = [(rng.standard_normal((9, 7, 80, 2)) * 1.5).astype(np.float32) for _ in range(3)]
    bs = [rng.randint(0, 2, (9, tx.frame_size, 2)).astype(np.int32) for _ in range(3)]
    start = {n: getattr(tr, n).clone() for n in ("params", "adam_m", "adam_v", "adam_state")}

    def restore():
        for n, v in start.items():
            getattr(tr, n).copy_(v)

    def snap(pl):
        torch.cuda.synchronize()
        return (tr.params.clone(), tr.adam_m.clone(), tr.adam_v.clone(), tr.adam_state.clone(), pl.metrics_buf.clone(),
                pl.tx_power.clone(), pl.chest.clone())

    plain = []
    pl = tr.resident(9)
    for k in range(3):
        pl.set_batch(xs[k], bs[k])
        pl.run(True, graph=False)
        plain.append(snap(pl))
    for graph in (False, True):
        restore()
        a = _FusedPlan(tr, 9)
        b = _FusedPlan(tr, 9, twin_of=a)
        a.pipe_with(b, 0)
        b.pipe_with(a, 1)
        a.set_batch(xs[0], bs[0])
        b.set_batch(xs[1], bs[1])
        a.run(True, graph=graph, pipe=0)
        got = [snap(a)]
        a.set_batch(xs[2], bs[2])                     # (refilled only after the step that read it was issued: same stream)
        b.run(True, graph=graph, pipe=1)
        got.append(snap(b))
        a.run(True, graph=graph, pipe=1)
        got.append(snap(a))
        for k in range(3):
            for u, v in zip(plain[k], got[k]):
                assert torch.equal(u, v), (graph, k)
        a.close()
        b.close()
    out = []
    for pipe in (False, True):
        hf = H.Flags(nbits=2, nfilter=64, channel="EPA", msg_length=7 * 2048, batch_size=512, max_epoch_num=3,
                     early_stop=100, token="PN", save_dir="/tmp/_eq_pipe_%d/" % pipe, seed=6, eval_frames=512,
                     device_data=True, pipeline_norm=pipe)
        o = ofdm.ofdm_tx(hf)
        res = H.train(hf, verbose=False, run_test=False, rx_params=glorot_init(R.rx_dims(hf, o), 1))
        out.append((res["history"], res["trainer"].params.detach().clone()))
    assert out[0][0] == out[1][0]
    assert torch.equal(out[0][1], out[1][1])


def test_virtual_next_batch_trains_the_same_equaliser_on_interleaved_profiles():
    """include/dccn.h dccn_eq_buffers.x_next_virtual on the equaliser's own training channel (mixRayleigh: four static profiles
    frame by frame): the pipelined loop whose optimizer launch reads the next batch as the fused generator's (y, noise, power
    partials) -- x never written but for an epoch's first batch, the noise-power monitor finished by that launch -- against the
    same loop materialising every batch (dccn_gen_static_apply): identical history (losses, BER, tx / noise power, channel
    RMS) and parameters over 3 epochs of 28 steps -- with the step issued eagerly (the loop's default) and as hipGraph replays."""
    from dl_ofdm_amd import receiver as R, receiver_mp as H
    from dl_ofdm_amd.engine import glorot_init
    from dl_ofdm_amd import ofdm
    out = []
    for virt, graph in ((False, False), (True, False), (True, True)):
        hf = H.Flags(nbits=2, nfilter=64, channel="mixRayleigh", msg_length=7 * 2048, batch_size=512, max_epoch_num=3,
                     early_stop=100, token="VN", save_dir="/tmp/_eq_virt_%d%d/" % (virt, graph), seed=8, eval_frames=512,
                     device_data=True, virtual_next=virt, step_graph=graph)
        o = ofdm.ofdm_tx(hf)
        res = H.train(hf, verbose=False, run_test=False, rx_params=glorot_init(R.rx_dims(hf, o), 1))
        out.append((res["history"], res["trainer"].params.detach().clone()))
    for o in out[1:]:
        assert out[0][0] == o[0]
        assert torch.equal(out[0][1], o[1])
    assert all(np.isfinite(h["train_loss"]) and 0.0 < h["train_ber"] < 0.6 for h in out[0][0])


def test_equalizer_harness_on_device_generated_data():
    """receiver_mp.train(device_data=True): bits, frames, fading, noise and the true channel response all come
    from the GPU generator; same learning criterion as the host-data test above."""
    from dl_ofdm_amd import receiver as R, receiver_mp as H
    base = R.Flags(nbits=2, nfilter=64, channel="AWGN", SNR=10.0, msg_length=7 * 4096, batch_size=512,
                   max_epoch_num=6, early_stop=100, token="B2", save_dir="/tmp/_eq_test2/", seed=5, device_data=True)
    res = R.train(base, verbose=False, run_test=False)
    hf = H.Flags(nbits=2, nfilter=64, channel="Flat", msg_length=7 * 2048, batch_size=512, max_epoch_num=5,
                 early_stop=100, token="B2", save_dir="/tmp/_eq_test2/", seed=6, eval_frames=2048, device_data=True,
                 test_frames=512, snr_lo=0, snr_hi=20, snr_step=10)
    out = H.train(hf, verbose=False, run_test=True, rx_params=res["params"])
    hist = out["history"]
    assert hist[-1]["train_loss"] < hist[0]["train_loss"] - 0.02, hist
    assert np.isfinite(hist[-1]["chan_rms"]) and hist[-1]["chan_rms"] > 0
    sw = out["sweep"]
    assert set(sw) == set(H.TEST_CHANNELS) and all(len(v[1]) == 3 for v in sw.values())
    assert sw["Flat"][1][-1] < sw["Flat"][1][0]                 # BER falls with SNR on the training channel


def test_fused_step_without_cp_matches_oracle():
    """cp=False through the planned step against the fp64 autograd oracle (window reads, zero gradient into the CP)"""
    F, tx, ecfg, rcfg, pe, pr, tr = _trainer(cp=False, seed=41)
    rng = np.random.RandomState(17)
    x = (rng.standard_normal((7, 7, 80, 2)) * 2).astype(np.float32)
    bits = rng.randint(0, 2, (7, tx.frame_size, 2)).astype(np.int32)
    lit_rx = LiteralRx({k: v.astype(np.float64) for k, v in pr.items()}, rcfg, dtype=torch.float64, literal_conv=False)

    class _Win(LiteralEqualizer):                       # the frozen receiver crops the prefix itself (model.py:1236-1240)
        def forward_backward(self, x_raw, bits_):
            rec = self.rx.receiver
            self.rx.receiver = lambda v: rec(v[:, :, 16:80, :])
            try:
                return super().forward_backward(x_raw, bits_)
            finally:
                self.rx.receiver = rec
    g_ref, info = _Win({k: v.astype(np.float64) for k, v in pe.items()}, lit_rx, ecfg).forward_backward(x.astype(np.float64), bits)
    m = tr.train_step(x, bits, fused=True, graph=False)
    assert abs(m["ce_mean"] - info["ce_mean"]) <= 3e-6 * abs(info["ce_mean"])
    assert np.array_equal(np.asarray(m["conf"]).reshape(2, 2), info["conf"])
    g = tr.get_grads()
    p0 = {k: v for k, v in pe.items()}
    for n in tr.names:
        reg = E.EQ_REG_COEFF * 2 * O.REG_L2 * p0[n].astype(np.float64).ravel() if "/dense" in n else 0.0
        got, want = g[n].astype(np.float64).ravel() + reg, g_ref[n].ravel()
        cos = float(got @ want / (np.linalg.norm(got) * np.linalg.norm(want)))
        assert cos >= 1 - 1e-6, (n, cos)


def test_equalizer_forward_at_nfft_128():
    """the stage is not tied to N=64: K=128 (7x128 smoothing kernel -> a 1792x1792 Toeplitz layer) vs the oracle"""
    from dl_ofdm_amd.complex import VariableStore
    from dl_ofdm_amd.model import equalizer_ofdm
    from dl_ofdm_amd.ofdm import ofdm_tx
    F = _Flags()
    F.nfft, F.nfilter = 128, 128
    tx = ofdm_tx(F)
    st = VariableStore(seed=6)
    x = dev(np.random.RandomState(42).standard_normal((3, 7, tx.K + tx.CP, 2)) * 1.3)
    with st.scope("Equalizer"):
        out, snr, chest = equalizer_ofdm(x, F, tx, scope=st)
    c = E.EqConfig(S=7, K=tx.K, CP=tx.CP, cp=True, pilot_size=tx.pilot_size,
                   pilot_carriers=tuple(int(v) for v in tx.pilotCarriers))
    p = {n: st.tensor(n).detach().cpu().numpy().astype(np.float64).reshape(s) for n, s in E.param_shapes(c).items()}
    o_out, o_snr, o_h = E.equalizer_forward(p, x.cpu().numpy().astype(np.float64), c)
    aligned(out, o_out, "equalized (N=128)", 1 - 1e-8)
    aligned(torch.view_as_real(chest), o_h, "chest (N=128)", 1 - 1e-8)
    aligned(snr, o_snr, "snr (N=128)", 1 - 1e-8)
