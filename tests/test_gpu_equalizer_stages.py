"""Stage-by-stage parity of the FUSED equaliser training step (dccn_eq_train_step: what receiver_mp / config 5 run) at
north_star's tolerance: every stage of ``equalizer_ofdm`` (dev/py/model.py:349-478) is compared with the float64 oracle
evaluated on the GPU's OWN inputs to that stage -- forward, input gradient and parameter gradients -- to **1e-5 of the
reference tensor's scale**, the way tests/test_gpu_engine.py checks the basic receiver (an end-to-end fp32-vs-fp64 comparison
through twelve layers measures the conditioning of the chain, 5e-5..3e-4 here, not any kernel; test_gpu_equalizer.py keeps the
end-to-end direction check, cosine >= 1 - 1e-6).

The GPU's stage inputs are the step's own intermediates, read out of its workspace through dccn_eq_workspace_tensor
(include/dccn.h); the oracle side is oracle/equalizer_oracle.py (NumPy, literal tap-by-tap C-Conv) for the forward stages and
float64 autograd over oracle/torch_ref.py::conv2d_complex_literal (the zero-padded conv3d formulation) for the gradients.

Stated margins (everything else is held to 1e-5 flat):
  * equalise (:431-438) divides by |h|: cells with |h| below 2 % of the batch's median |h| are left out of the comparison of
    that stage's outputs and gradients (d(1/|h|) ~ 1/|h|^2 amplifies the GPU's own 6e-8 input rounding past any fixed bar);
  * Equalizer/conv3d_1/bias is d = sum_c (dbe[2c] - dbe[2c+1]), a difference of two nearly equal sums: it is held to 1e-5 of
    the SUM OF MAGNITUDES of its terms (the forward error bound of a float32 sum), not of the cancelled result.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import dccn_oracle as O
from oracle import equalizer_oracle as E
from oracle.torch_ref import LiteralRx, conv2d_complex_literal

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel(got, want, scale=None):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    s = float(np.abs(want).max()) if scale is None else float(scale)
    return float(np.abs(got - want).max()) / max(s, 1e-300)


class Stage:
    """collects (name, error) and asserts them together, so one run shows every stage's margin"""

    def __init__(self):
        self.rows = []

    def check(self, name, got, want, scale=None, tol=TOL):
        self.rows.append((name, rel(got, want, scale), tol))

    def finish(self):
        bad = ["%s %.2e > %.0e" % r for r in self.rows if not r[1] <= r[2]]
        print("\n".join("%-34s %.2e" % (n, e) for n, e, _ in self.rows))
        assert not bad, "; ".join(bad)


def _trainer(nbits, B, seed):
    from test_gpu_equalizer import _trainer as mk
    return mk(nbits=nbits, seed=seed)


def _ws(tr, pl, name):
    off, cnt = C.c_size_t(), C.c_size_t()
    rc = tr.lib.dccn_eq_workspace_tensor(C.byref(pl.shape), 1, name.encode(), C.byref(off), C.byref(cnt))
    assert rc == 0, (name, rc)
    t = pl.ws[off.value:off.value + 4 * cnt.value].view(torch.float32)
    return t.cpu().numpy().astype(np.float64)


def _t(a, grad=False):
    return torch.tensor(np.asarray(a, np.float64), dtype=torch.float64, requires_grad=grad)


def _vjp(fn, inputs, upstream):
    """float64 autograd: gradients of sum(fn(*inputs) * upstream) with respect to every input"""
    ts = [_t(a, True) for a in inputs]
    out = fn(*ts)
    gs = torch.autograd.grad(out, ts, grad_outputs=_t(upstream), allow_unused=True)
    return [None if g is None else g.numpy() for g in gs]


def check_stages(v, P, G, ecfg, lit_rx, bits, st):
    """every stage of the step on the values `v` holds for its inputs (v: name -> float64 array, the names of
    dccn_eq_workspace_tensor + "x", "h", "out_eq", "snr_db"); P / G: parameters the step ran with / gradients it left
    (TF shapes).  Returns the oracle's (ce_mean, confusion matrix, number of near-tie decisions) of the frozen receiver on
    v["out_eq"]."""
    S, K, nsc = ecfg.S, ecfg.K, ecfg.n_sc
    B = v["x"].shape[0]
    R, SK2, K2, N2 = B * S, S * K * 2, 2 * K, 2 * nsc

    # ---- forward, each stage on the GPU's own input ---------------------------------------------------------------
    x_norm = v["x_norm"].reshape(B, S, nsc, 2)
    st.check("R0 batch-moment norm", x_norm, O.batch_moment_norm(v["x"].astype(np.float64))[0])
    ln = v["ln"].reshape(B, S, nsc, 2)
    st.check(":363 layer_norm", ln, E.layer_norm(x_norm))
    t1 = v["t1"].reshape(R, K2)
    st.check(":371 dense", t1, ln.reshape(R, N2) @ P["Equalizer/dense/kernel"] + P["Equalizer/dense/bias"])
    y = v["y"].reshape(B, S, K, 2)

    def cconv_1k(inp, name):            # (1,K) valid C-Conv of [B,S,K,2] -> [B,S,K(filters),2]  (:378-380, :439-449)
        o = O.layers_conv2d_complex_literal(inp.reshape(B, S, K, 1, 2), P[name + "/kernel"], P[name + "/bias"], (1, 1), "valid")
        return o[:, :, 0, :, :]
    st.check(":378 C-Conv (1,K)", y, cconv_1k(t1.reshape(B, S, K, 2), "Equalizer/conv3d"))
    d1, d2 = v["d1"].reshape(B, -1), v["d2"].reshape(B, SK2)
    st.check(":394 dense_1 (pilots)", d1, y.reshape(B, SK2) @ P["Equalizer/dense_1/kernel"] + P["Equalizer/dense_1/bias"])
    st.check(":402 dense_2", d2, d1 @ P["Equalizer/dense_2/kernel"] + P["Equalizer/dense_2/bias"])
    d3, d4 = v["d3"].reshape(B, SK2), v["d4"].reshape(B, SK2)
    st.check(":408 dense_3", d3, d2 @ P["Equalizer/dense_3/kernel"] + P["Equalizer/dense_3/bias"])
    st.check(":421 dense_4 + tanh", d4, np.tanh(d3 @ P["Equalizer/dense_4/kernel"] + P["Equalizer/dense_4/bias"]))
    h = v["h"].reshape(B, S, K, 2)
    h_ref = O.layers_conv2d_complex_literal(d4.reshape(B, S, K, 1, 2), P["Equalizer/conv3d_1/kernel"],
                                            P["Equalizer/conv3d_1/bias"], (1, 1), "same")[:, :, :, 0, :]
    st.check(":428 C-Conv (S,K) same", h, h_ref)
    eq, corr = v["eq"].reshape(B, S, K, 2), v["corr"].reshape(B, S, K, 2)
    habs = np.sqrt((h ** 2).sum(-1))
    ok = habs >= 0.02 * np.median(habs)                   # stated margin: cells next to the 1/|h| pole
    # (how many cells the margin leaves out: printed so that the number is on record -- DESIGN.md section 4 quotes it)
    print("equalise margin: %d of %d cells (%.4f %%) below 2 %% of the median |h| left out [B=%d]"
          % (int((~ok).sum()), ok.size, 100.0 * float((~ok).mean()), B))
    assert ok.mean() > 0.97
    eq_ref, corr_ref = E.equalize(y, h)
    st.check(":431-436 equalise", eq[ok], eq_ref[ok])
    st.check(":438 autocorrelation", corr[ok], corr_ref[ok])
    cat = v["cat"].reshape(B, S, K, 4)
    st.check(":443 C-Conv of eq -> cat", cat[..., 0:2], cconv_1k(eq, "Equalizer/conv3d_3"))
    st.check(":439 C-Conv of corr -> cat", cat[..., 2:4], cconv_1k(corr, "Equalizer/conv3d_2"))
    out_eq = v["out_eq"].reshape(B, S, nsc, 2)
    st.check(":458 dense_5", out_eq.reshape(R, N2), cat.reshape(R, 4 * K) @ P["Equalizer/dense_5/kernel"] + P["Equalizer/dense_5/bias"])
    snr_ref = E.pilot_snr(eq, ecfg.pilot_carriers)
    st.check(":465-475 pilot SNR", v["snr_db"].reshape(B, 1), snr_ref, scale=max(np.abs(snr_ref).max(), 1.0))

    # ---- the frozen receiver: loss and the gradient handed to the equaliser ----------------------------------------
    o_t = _t(out_eq, True)
    prob, _, z = lit_rx.receiver(o_t)
    ce_mean, conf, _, _ = lit_rx.losses(prob, bits)
    dout = v["dout"].reshape(B, S, nsc, 2)
    (dout_ref,) = torch.autograd.grad(z, o_t, grad_outputs=_t(v["dz"].reshape(z.shape)))      # the receiver's linear part, on the GPU's dz
    st.check("receiver dX (dz -> d equalized)", dout, dout_ref.numpy())

    # ---- backward, last stage first: each on the GPU's own upstream gradient --------------------------------------
    do2 = dout.reshape(R, N2)
    dcat_ref = (do2 @ P["Equalizer/dense_5/kernel"].T).reshape(B, S, K, 4)
    deqc, dcorc = v["deqc"].reshape(B, S, K, 2), v["dcorc"].reshape(B, S, K, 2)
    st.check("dense_5 dX -> d(C-Conv eq)", deqc, dcat_ref[..., 0:2], scale=np.abs(dcat_ref).max())
    st.check("dense_5 dX -> d(C-Conv corr)", dcorc, dcat_ref[..., 2:4], scale=np.abs(dcat_ref).max())
    st.check("dense_5 dW", G["Equalizer/dense_5/kernel"], cat.reshape(R, 4 * K).T @ do2)
    st.check("dense_5 db", G["Equalizer/dense_5/bias"], do2.sum(0))

    def conv_fn(padding):
        return lambda inp, kern, bias: conv2d_complex_literal(inp, kern, bias, padding)

    def bias_scale(gb, up):
        # the bias pair (ba - bb, bb - ba) of complex.py:187-188: db = +-sum(d re - d im), a signed sum over every row --
        # held to 1e-5 of its value or of 1e-3 of the sum of magnitudes, whichever is larger (cancellation)
        return max(np.abs(gb).max(), np.abs(up).sum() * 1e-3)
    deq, dcorr = v["deq"].reshape(B, S, K, 2), v["dcorr"].reshape(B, S, K, 2)
    for tag, name, inp, up, dx_gpu in (("eq", "Equalizer/conv3d_3", eq, deqc, deq), ("corr", "Equalizer/conv3d_2", corr, dcorc, dcorr)):
        # input [B,S,K,1,2] -> output [B,S,1,K,2]: the upstream gradient is indexed by filter on the last-but-one axis
        gx, gk, gb = _vjp(conv_fn("valid"), [inp.reshape(B, S, K, 1, 2), P[name + "/kernel"], P[name + "/bias"]],
                          up.reshape(B, S, 1, K, 2))
        st.check("C-Conv(%s) dX" % tag, dx_gpu, gx.reshape(B, S, K, 2))
        st.check("C-Conv(%s) dW" % tag, G[name + "/kernel"], gk)
        st.check("C-Conv(%s) db" % tag, G[name + "/bias"], gb, scale=bias_scale(gb, up))

    def equalize_t(yt, ht):
        yc, hc = torch.view_as_complex(yt.contiguous()), torch.view_as_complex(ht.contiguous())
        hn = torch.conj(hc) / torch.abs(hc)
        e = yc * hn
        return torch.cat([torch.view_as_real(e), torch.view_as_real(e * torch.conj(e))], dim=-1)
    dy, dh = v["dy"].reshape(B, S, K, 2), v["dh"].reshape(B, S, K, 2)
    gy, gh = _vjp(equalize_t, [y, h], np.concatenate([deq, dcorr], axis=-1))
    st.check("equalise dy", dy[ok], gy[ok])
    st.check("equalise dh", dh[ok], gh[ok])

    dd4 = v["dd4"].reshape(B, SK2)
    gd4, gk1, gb1 = _vjp(conv_fn("same"), [d4.reshape(B, S, K, 1, 2), P["Equalizer/conv3d_1/kernel"], P["Equalizer/conv3d_1/bias"]],
                         dh.reshape(B, S, K, 1, 2))
    st.check("C-Conv (S,K) dX . tanh'", dd4, gd4.reshape(B, SK2) * (1.0 - d4 ** 2))
    st.check("C-Conv (S,K) dW", G["Equalizer/conv3d_1/kernel"], gk1)
    st.check("C-Conv (S,K) db", G["Equalizer/conv3d_1/bias"], gb1, scale=np.abs(dh).sum())      # stated margin (cancellation)
    dd3, dd2 = v["dd3"].reshape(B, SK2), v["dd2"].reshape(B, SK2)
    st.check("dense_4 dX", dd3, dd4 @ P["Equalizer/dense_4/kernel"].T)
    st.check("dense_4 dW", G["Equalizer/dense_4/kernel"], d3.T @ dd4)
    st.check("dense_4 db", G["Equalizer/dense_4/bias"], dd4.sum(0))
    st.check("dense_3 dX", dd2, dd3 @ P["Equalizer/dense_3/kernel"].T)
    st.check("dense_3 dW", G["Equalizer/dense_3/kernel"], d2.T @ dd3)
    st.check("dense_3 db", G["Equalizer/dense_3/bias"], dd3.sum(0))
    # pilot bottleneck (dense_2 then dense_1 backwards, one launch per direction in the default plan); its input gradient
    # lands on top of the equalise stage's dy: dflat = total gradient of y
    dd1_ref = dd2 @ P["Equalizer/dense_2/kernel"].T
    dflat = v["dflat"].reshape(B, S, K, 2)
    st.check("bottleneck dX + dy", dflat.reshape(B, SK2), dy.reshape(B, SK2) + dd1_ref @ P["Equalizer/dense_1/kernel"].T)
    st.check("dense_2 dW", G["Equalizer/dense_2/kernel"], d1.T @ dd2)
    st.check("dense_2 db", G["Equalizer/dense_2/bias"], dd2.sum(0))
    st.check("dense_1 dW", G["Equalizer/dense_1/kernel"], y.reshape(B, SK2).T @ dd1_ref)
    st.check("dense_1 db", G["Equalizer/dense_1/bias"], dd1_ref.sum(0))
    dt1 = v["dt1"].reshape(R, K2)
    gt1, gk0, gb0 = _vjp(conv_fn("valid"), [t1.reshape(B, S, K, 1, 2), P["Equalizer/conv3d/kernel"], P["Equalizer/conv3d/bias"]],
                         dflat.reshape(B, S, 1, K, 2))
    st.check("C-Conv (1,K) dX", dt1, gt1.reshape(R, K2))
    st.check("C-Conv (1,K) dW", G["Equalizer/conv3d/kernel"], gk0)
    st.check("C-Conv (1,K) db", G["Equalizer/conv3d/bias"], gb0, scale=bias_scale(gb0, dflat))
    st.check("dense dW", G["Equalizer/dense/kernel"], ln.reshape(R, N2).T @ dt1)
    st.check("dense db", G["Equalizer/dense/bias"], dt1.sum(0))
    pr = prob.detach().numpy().reshape(-1, 2)
    return float(ce_mean.detach()), conf.numpy(), int((np.abs(pr[:, 1] - pr[:, 0]) < 2e-5).sum())


WS_NAMES = ("x_norm", "ln", "t1", "y", "d1", "d2", "d3", "d4", "eq", "corr", "cat", "dz", "dout", "deqc", "dcorc", "deq", "dcorr",
            "dy", "dh", "dd4", "dd3", "dd2", "dflat", "dt1")


@pytest.mark.parametrize("nbits,B,plan", [(2, 73, 1), (2, 12, 1), (2, 200, 1), (4, 73, 1), (2, 73, 0), (2, 73, 3), (2, 200, 0),
                                          (1, 73, 1)])
def test_fused_step_stage_by_stage_at_1e5(nbits, B, plan):
    """plan = tuning key 20: 1 the shipped launch plan (few-row tiles, fused pilot bottleneck, grouped C-Conv pairs, job-table
    optimizer launch), 0 the launch-per-stage plan, 3 the re-plan without the fused bottleneck; 73 frames = the reference's
    training batch (few-row kernels, folded receiver), 200 = split-K gradients, 12 = direct ones."""
    from dl_ofdm_amd import _lib
    lib = _lib.load()
    F, tx, ecfg, rcfg, pe, pr, tr = _trainer(nbits, B, seed=51 + B)
    rng = np.random.RandomState(300 + B + nbits)
    x = (rng.standard_normal((B, ecfg.S, ecfg.n_sc, 2)) * 2).astype(np.float32)
    bits = rng.randint(0, 2, (B, tx.frame_size, nbits)).astype(np.int32)
    shp = E.param_shapes(ecfg)
    P = {n: a.astype(np.float64).reshape(shp[n]) for n, a in tr.get_params().items()}     # the step's own parameters
    try:
        assert lib.dccn_set_tuning(20, plan) == 0
        tr.train_step(x, bits, fused=True, graph=False)
        torch.cuda.synchronize()
    finally:
        lib.dccn_set_tuning(20, 1)
    pl = tr._plan(B)
    G = {n: a.astype(np.float64).reshape(shp[n]) for n, a in tr.get_grads().items()}      # without the L2 term (reg_coef)
    v = {n: _ws(tr, pl, n) for n in WS_NAMES}
    v["x"] = x
    v["h"] = pl.chest.detach().cpu().numpy().astype(np.float64)
    v["out_eq"] = pl.out_eq.detach().cpu().numpy().astype(np.float64)
    v["snr_db"] = pl.snr_db.detach().cpu().numpy().astype(np.float64)
    if plan != 1:
        # without the fused bottleneck the pilot branch's input gradient is a GEMM of its own: it lands in "dflat" and is
        # then either added onto "dy" in place (dy = total, dflat = branch) or the sum rides on that GEMM's store (dflat =
        # total, dy untouched: the layout of the shipped plan).  Bring the first form to the second.
        SK2 = ecfg.S * ecfg.K * 2
        branch = (v["dd2"].reshape(B, SK2) @ P["Equalizer/dense_2/kernel"].T) @ P["Equalizer/dense_1/kernel"].T
        if rel(v["dflat"].reshape(B, SK2), branch) < 1e-3:
            total = v["dy"]
            v["dy"] = total - v["dflat"]
            v["dflat"] = total
    lit_rx = LiteralRx({k: a.astype(np.float64) for k, a in pr.items()}, rcfg, dtype=torch.float64, literal_conv=False)
    st = Stage()
    ce_mean, conf, ties = check_stages(v, P, G, ecfg, lit_rx, bits, st)
    st.finish()
    m = tr.last
    assert abs(m["ce_mean"] - ce_mean) <= TOL * abs(ce_mean), (m["ce_mean"], ce_mean)
    # hard decisions: exact outside a 1e-5 probability margin (|p1 - p0| < 2e-5: float32 and float64 may fall on either side
    # of a tie; such a cell moves one count between two cells of a row of the confusion matrix)
    got = np.asarray(m["conf"]).reshape(2, 2)
    assert got.sum() == conf.sum() and np.array_equal(got.sum(1), conf.sum(1))
    assert np.abs(got - conf).sum() <= 2 * ties, (got, conf, ties)
