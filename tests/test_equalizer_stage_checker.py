"""CPU check of the stage-by-stage checker of tests/test_gpu_equalizer_stages.py: fed with the float64 intermediates and
gradients of the end-to-end autograd oracle (oracle/torch_ref.py::LiteralEqualizer with every intermediate retained) in the
place of the GPU's workspace tensors, every stage must agree to rounding (1e-11) -- i.e. the per-stage formulas, reshapes and
upstream-gradient layouts the GPU test relies on are themselves consistent with the whole-graph oracle; and a perturbed
intermediate must be caught by exactly the stages that consume or produce it."""
import numpy as np
import torch

from oracle import dccn_oracle as O
from oracle import equalizer_oracle as E
from oracle.torch_ref import LiteralRx, conv2d_complex_literal
from test_gpu_equalizer_stages import Stage, check_stages


def oracle_values(B=5, nbits=2, seed=3):
    rng = np.random.RandomState(seed)
    pc = (3, 9, 15, 21, 42, 48, 54, 60)
    c = E.EqConfig(S=7, K=64, CP=16, cp=True, pilot_size=16, pilot_carriers=pc)
    rc = O.RxConfig(S=7, kin=80, F=64, D=320, nbits=nbits)
    pe = {k: v.astype(np.float64) for k, v in E.init_params(c, seed=seed, bias_scale=0.05).items()}
    pr = {k: v.astype(np.float64) for k, v in O.init_params(rc, seed=seed + 1).items()}
    x = rng.standard_normal((B, 7, 80, 2)) * 2
    bits = rng.randint(0, 2, (B, 320, nbits)).astype(np.int32)
    lit_rx = LiteralRx(pr, rc, dtype=torch.float64, literal_conv=False)
    P = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in pe.items()}
    S, K = 7, 64
    keep = {}

    def k(name, t):
        t.retain_grad()
        keep[name] = t
        return t
    conv = lambda x5, n, pad: conv2d_complex_literal(x5, P[n + "/kernel"], P[n + "/bias"], pad)      # noqa: E731
    xt = torch.tensor(x, dtype=torch.float64)
    x_norm = k("x_norm", lit_rx.normalise(xt).clone().requires_grad_(True))
    mean = x_norm.mean(dim=(1, 2, 3), keepdim=True)
    var = ((x_norm - mean) ** 2).mean(dim=(1, 2, 3), keepdim=True)
    inv = torch.rsqrt(var + E.LN_EPS)
    ln = k("ln", x_norm * inv + (-mean * inv))
    t1 = k("t1", ln.reshape(B, S, -1) @ P["Equalizer/dense/kernel"] + P["Equalizer/dense/bias"])
    y = k("y", conv(t1.reshape(B, S, K, 1, 2), "Equalizer/conv3d", "valid").permute(0, 1, 3, 2, 4).reshape(B, S, K, 2))
    d1 = k("d1", y.reshape(B, S * K * 2) @ P["Equalizer/dense_1/kernel"] + P["Equalizer/dense_1/bias"])
    d2 = k("d2", d1 @ P["Equalizer/dense_2/kernel"] + P["Equalizer/dense_2/bias"])
    d3 = k("d3", d2 @ P["Equalizer/dense_3/kernel"] + P["Equalizer/dense_3/bias"])
    pre4 = k("pre4", d3 @ P["Equalizer/dense_4/kernel"] + P["Equalizer/dense_4/bias"])
    d4 = k("d4", torch.tanh(pre4))
    h = k("h", conv(d4.reshape(B, S, K, 1, 2), "Equalizer/conv3d_1", "same").reshape(B, S, K, 2))
    yc, hc = torch.view_as_complex(y.contiguous()), torch.view_as_complex(h.contiguous())
    e = yc * (torch.conj(hc) / torch.abs(hc))
    eq = k("eq", torch.view_as_real(e))
    corr = k("corr", torch.view_as_real(torch.view_as_complex(eq.contiguous()) * torch.conj(torch.view_as_complex(eq.contiguous()))))
    eqc = k("eqc", conv(eq.reshape(B, S, K, 1, 2), "Equalizer/conv3d_3", "valid")[:, :, 0, :, :])
    corc = k("corc", conv(corr.reshape(B, S, K, 1, 2), "Equalizer/conv3d_2", "valid")[:, :, 0, :, :])
    cat = k("cat", torch.cat([eqc, corc], dim=-1))
    out_eq = k("out_eq", (cat.reshape(B, S, 4 * K) @ P["Equalizer/dense_5/kernel"] + P["Equalizer/dense_5/bias"]).reshape(B, S, 80, 2))
    prob, _, z = lit_rx.receiver(out_eq)
    z.retain_grad()
    ce_mean, conf, _, _ = lit_rx.losses(prob, bits)
    ce_mean.backward(retain_graph=True)
    v = {n: t.detach().numpy().copy() for n, t in keep.items() if n not in ("pre4", "eqc", "corc")}
    grads = {n: t.grad.numpy().copy() for n, t in keep.items()}      # (copied now: retained gradients accumulate on later calls)
    G = {n: t.grad.numpy().copy() for n, t in P.items()}
    dz = z.grad.numpy().copy()
    g = lambda n: grads[n]      # noqa: E731
    # "deq" of the step is the gradient through the :443 C-Conv alone (eq also feeds the autocorrelation; the equalise
    # stage's backward adds that path itself)
    (deq_direct,) = torch.autograd.grad(eqc, eq, grad_outputs=torch.tensor(grads["eqc"]), retain_graph=True)
    v.update(x=x, snr_db=E.pilot_snr(eq.detach().numpy(), pc), dz=dz, dout=g("out_eq"), deqc=g("eqc"),
             dcorc=g("corc"), deq=deq_direct.numpy().copy(), dcorr=g("corr"), dh=g("h"), dd4=g("pre4"), dd3=g("d3"), dd2=g("d2"),
             dflat=g("y"), dt1=g("t1"))
    # dy = the equalise stage's own contribution to dy (the total, dflat, also carries the pilot branch)
    v["dy"] = v["dflat"] - (g("d1") @ pe["Equalizer/dense_1/kernel"].T).reshape(B, S, K, 2)
    return v, pe, G, c, lit_rx, bits, float(ce_mean.detach()), conf.numpy()


def test_stage_checker_agrees_with_the_whole_graph_oracle():
    v, pe, G, c, lit_rx, bits, ce, conf = oracle_values()
    st = Stage()
    ce2, conf2, _ = check_stages(v, pe, G, c, lit_rx, bits, st)
    worst = max(e for _, e, _ in st.rows)
    assert worst <= 1e-11, sorted(st.rows, key=lambda r: -r[1])[:3]
    assert len(st.rows) >= 45 and abs(ce - ce2) <= 1e-14 and np.array_equal(conf, conf2)


def test_stage_checker_localises_a_wrong_intermediate():
    v, pe, G, c, lit_rx, bits, _, _ = oracle_values(seed=4)
    v = dict(v)
    v["d3"] = v["d3"] * (1 + 1e-4)                     # one stage's output off by 1e-4
    st = Stage()
    check_stages(v, pe, G, c, lit_rx, bits, st)
    bad = {n for n, e, tol in st.rows if e > tol}
    assert bad == {":408 dense_3", ":421 dense_4 + tanh", "dense_4 dW", "dense_3 dX"} or \
        bad == {":408 dense_3", ":421 dense_4 + tanh", "dense_4 dW"}, bad
