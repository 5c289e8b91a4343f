"""GPU parity of every C-ABI operator against the CPU oracle (``-m gpu``).

Tolerance (north_star): 1e-5 relative fp32, measured as max|gpu - ref64| / max|ref64| where
ref64 is the oracle evaluated in float64 on the same float32 inputs.  Integer results
(confusion counts, decisions outside the stated margin) must be exact.
"""
import numpy as np
import pytest
import torch

from oracle import dccn_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def relerr(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    scale = max(float(np.abs(ref).max()), 1e-30)
    return float(np.abs(got - ref).max()) / scale


def assert_close(got, ref, what, tol=RTOL):
    e = relerr(got, ref)
    assert e <= tol, "%s: rel err %.3e > %.1e" % (what, e, tol)


@pytest.fixture(scope="module")
def ops():
    from dl_ofdm_amd import ops as _ops
    from dl_ofdm_amd import _lib
    cu, wf, hbm, arch = _lib.device_info()
    assert wf == 64 and arch.startswith("gfx950"), (cu, wf, arch)
    return _ops


# ---- R0 -------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(36, 7, 80, 2), (1170, 7, 80, 2), (5, 3, 5, 2), (64, 7, 64, 2), (2, 1, 3, 2)])
def test_batch_moment_norm(ops, shape):
    rng = np.random.RandomState(1)
    x = (rng.randn(*shape) * rng.uniform(0.1, 3.0, size=shape[1:]) + rng.randn(*shape[1:])).astype(np.float32)
    y, mean, var = ops.batch_moment_norm(dev(x), return_moments=True)
    x2 = x.reshape(shape[0], -1).astype(np.float64)
    yr, mr, vr = O.batch_moment_norm(x2)
    assert_close(mean.cpu().numpy().reshape(-1), mr, "mean")
    assert_close(var.cpu().numpy().reshape(-1), vr, "var")
    assert_close(y.cpu().numpy().reshape(shape[0], -1), yr, "normalised x")


def test_batch_moment_norm_batch1_degenerate(ops):
    # batch 1 -> var 0 -> output exactly 0 * rsqrt(1e-9) = 0
    x = np.random.RandomState(2).randn(1, 7, 80, 2).astype(np.float32)
    y = ops.batch_moment_norm(dev(x))
    assert float(y.abs().max()) <= 1e-3


# ---- R8 -------------------------------------------------------------------------------------
def test_clip_power(ops):
    rng = np.random.RandomState(3)
    x = (rng.randn(50, 7, 80, 2) * 4.0).astype(np.float32)      # plenty of samples above the peak
    y, pw = ops.clip_power(dev(x), peak=8.0)
    yr, pr = O.complex_clip(x.astype(np.float64), 8.0)
    assert_close(y.cpu().numpy(), yr, "clipped")
    assert abs(float(pw) - float(pr)) <= 1e-5 * float(pr)
    assert float(np.abs(np.linalg.norm(y.cpu().numpy(), axis=-1)).max()) <= 8.0 * (1 + 1e-6)


# ---- R1 -------------------------------------------------------------------------------------
CCONV_SHAPES = [(252, 80, 64), (8190, 80, 64), (37, 5, 3), (100, 64, 64), (129, 33, 17), (70, 1096, 256),
                (1000, 34, 18), (511, 64, 64)]     # ragged k-major tiles; the equaliser's (1,K) C-Convs


@pytest.mark.parametrize("rows,kin,F", CCONV_SHAPES)
def test_cconv_gemm_fwd_bwd(ops, rows, kin, F):
    rng = np.random.RandomState(rows + kin + F)
    x = rng.randn(rows, kin, 2).astype(np.float32)
    w = (rng.randn(kin, 2 * F) / np.sqrt(kin)).astype(np.float32)
    b = rng.randn(2 * F).astype(np.float32)
    dout = rng.randn(rows, F, 2).astype(np.float32)
    xt, wt, bt = dev(x).requires_grad_(), dev(w).requires_grad_(), dev(b).requires_grad_()
    out = ops.cconv_gemm(xt, wt, bt)
    out.backward(dev(dout))
    ref = O.cconv_gemm_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    dx, dw, db = O.cconv_gemm_bwd(x.astype(np.float64), w.astype(np.float64), dout.astype(np.float64))
    assert_close(out.detach().cpu().numpy(), ref, "cconv fwd")
    assert_close(xt.grad.cpu().numpy(), dx, "cconv dx")
    assert_close(wt.grad.cpu().numpy(), dw, "cconv dw")
    assert_close(bt.grad.cpu().numpy(), db, "cconv dbias")


def test_cconv_gemm_no_bias_and_determinism(ops):
    rng = np.random.RandomState(9)
    x, w = rng.randn(500, 80, 2).astype(np.float32), rng.randn(80, 128).astype(np.float32)
    a = ops.cconv_gemm(dev(x), dev(w), None)
    b = ops.cconv_gemm(dev(x), dev(w), None)
    assert torch.equal(a, b)
    assert_close(a.cpu().numpy(), O.cconv_gemm_fwd(x.astype(np.float64), w.astype(np.float64), None), "no-bias fwd")


@pytest.mark.parametrize("rows,kin,F", [(8190, 80, 64), (511, 80, 64), (8190, 64, 64), (70, 64, 64), (4095, 80, 64), (37, 80, 96),
                                        (300, 80, 33)])
def test_cconv_fwd_staged_is_bitwise_the_whole_k_tile(ops, rows, kin, F):
    """csrc/cconv_fwd.h: the N = 64 C-Conv forward with its k range cut into stages of 32 that are consumed as they land
    (tuning key 2 = 7 .. 11: 64 x 64 and 32 x 128 tiles, three LDS-store slots) writes the same values to the same LDS positions and issues the same MFMA chain as the
    whole-k tile of gemm_f32_mfma.h (key 2 = 0): every output bit is the same -- ragged last row tile, K = 128 (no
    cyclic prefix), column counts that are not a multiple of 64 and an odd F (which must fall back) included."""
    import ctypes as C
    from dl_ofdm_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(rows + kin + F)
    x = dev(rng.randn(rows, kin, 2).astype(np.float32))
    w = dev((rng.randn(kin, 2 * F) / np.sqrt(kin)).astype(np.float32))
    b = dev(rng.randn(2 * F).astype(np.float32))
    xv = x
    default = lib.dccn_get_tuning(2)
    outs = {}
    try:
        for v in (0, 7, 8, 9, 10, 11):
            assert lib.dccn_set_tuning(2, v) == 0
            o = torch.full((rows, F, 2), float("nan"), device="cuda")
            _lib.check(lib.dccn_cconv_gemm_fwd(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()),
                                               C.c_void_p(o.data_ptr()), rows, kin, F, None), "cconv_fwd")
            torch.cuda.synchronize()
            outs[v] = o
    finally:
        lib.dccn_set_tuning(2, default)
    ref = O.cconv_gemm_fwd(xv.cpu().numpy().astype(np.float64), w.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64))
    assert_close(outs[0].cpu().numpy(), ref, "whole-k tile vs oracle")
    for v in (7, 8, 9, 10, 11):
        assert torch.equal(outs[v], outs[0]), (v, float((outs[v] - outs[0]).abs().max()))


def test_cconv_known_answer_dft(ops):
    """Wa=cos_k, Wb=-sin_k gives re = Re{X_k}, im = -Im{X_{N-k}}."""
    N = 64
    n = np.arange(N)[:, None]
    k = np.arange(N)[None, :]
    wa, wb = np.cos(2 * np.pi * n * k / N), -np.sin(2 * np.pi * n * k / N)
    w = np.concatenate([wa, wb], axis=1).astype(np.float32)
    rng = np.random.RandomState(4)
    xc = rng.randn(33, N) + 1j * rng.randn(33, N)
    x = np.stack([xc.real, xc.imag], axis=-1).astype(np.float32)
    out = ops.cconv_gemm(dev(x), dev(w), None).cpu().numpy()
    X = np.fft.fft(xc, axis=1)
    assert_close(out[..., 0], X.real, "Re X_k", tol=2e-5)
    assert_close(out[..., 1], -X[:, (-np.arange(N)) % N].imag, "-Im X_{N-k}", tol=2e-5)


# ---- backward half of the step as one launch (dense dX + C-Conv dWeff in its epilogue + dense dW) ----------------
@pytest.mark.parametrize("batch,kin,D", [(1170, 80, 320), (64, 64, 320), (37, 80, 50), (200, 64, 322)])
def test_rx_backward_fused(ops, batch, kin, D):
    """dccn_rx_backward against the float64 oracle of its three contractions (model.py:1268-1275 and
    complex.py:183-192 backward), with and without the dfft store."""
    import ctypes as C
    from dl_ofdm_amd import _lib
    lib = _lib.load()
    S, F = 7, 64
    rng = np.random.RandomState(batch + kin + D)
    xn = rng.randn(batch, S, kin, 2).astype(np.float32)
    a = rng.randn(batch, S * 2 * F).astype(np.float32)
    dz = rng.randn(batch, 2 * D).astype(np.float32)
    w = (rng.randn(S * 2 * F, 2 * D) / 30).astype(np.float32)
    shape = _lib.RxShape(batch, S, kin, F, D, 2)
    assert lib.dccn_rx_bwd_fused_supported(C.byref(shape)) == 1
    nws = lib.dccn_rx_backward_workspace_size(batch, S, kin, F, D)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    xt, at, dzt, wt = dev(xn), dev(a), dev(dz), dev(w)
    outs = []
    for want_dfft in (True, False):
        dfft = torch.zeros(batch, S * 2 * F, device="cuda")
        dw, db = torch.zeros_like(wt), torch.zeros(2 * D, device="cuda")
        dcw, dcb = torch.zeros(kin, 2 * F, device="cuda"), torch.zeros(2 * F, device="cuda")
        _lib.check(lib.dccn_rx_backward(xt.data_ptr(), at.data_ptr(), dzt.data_ptr(), wt.data_ptr(),
                                        dfft.data_ptr() if want_dfft else None, dw.data_ptr(), db.data_ptr(), dcw.data_ptr(),
                                        dcb.data_ptr(), batch, S, kin, F, D, 1, ws.data_ptr(), nws, None), "dccn_rx_backward")
        torch.cuda.synchronize()
        outs.append((dfft, dw, db, dcw, dcb))
    dfft, dw, db, dcw, dcb = outs[0]
    for u, v in zip(outs[0][1:], outs[1][1:]):
        assert torch.equal(u, v)
    assert float(outs[1][0].abs().max()) == 0.0
    dz64, a64, w64 = dz.astype(np.float64), a.astype(np.float64), w.astype(np.float64)
    assert_close(dfft.cpu().numpy(), dz64 @ w64.T, "dfft")
    assert_close(dw.cpu().numpy(), a64.T @ dz64, "dense dW")
    assert_close(db.cpu().numpy(), dz64.sum(0), "dense dbias")
    # the C-Conv gradient on the GPU's own dfft (the stage's input)
    _, gw, gb = O.cconv_gemm_bwd(xn.astype(np.float64).reshape(batch * S, kin, 2), np.zeros((kin, 2 * F)),
                                 dfft.cpu().numpy().astype(np.float64).reshape(batch * S, F, 2))
    assert_close(dcw.cpu().numpy(), gw, "C-Conv dW")
    assert_close(dcb.cpu().numpy(), gb, "C-Conv dbias")


# ---- R2 -------------------------------------------------------------------------------------
DENSE_SHAPES = [(36, 896, 640), (1170, 896, 640), (7, 13, 5), (300, 2048, 1024), (65, 130, 67), (1, 896, 640),
                (600, 260, 132), (2000, 64, 64),       # k-major weight gradient: ragged tiles / ragged last k range
                (73, 896, 896), (90, 512, 300),        # <= 96 rows: the skinny 16x64 tiles (forward and dX)
                (96, 896, 896), (5, 640, 896), (50, 256, 128), (73, 1120, 640)]   # fewrow.h: one-latency 16x16 / 64x64 tiles


@pytest.mark.parametrize("M,K,N", DENSE_SHAPES)
def test_dense_fwd_bwd(ops, M, K, N):
    rng = np.random.RandomState(M + K + N)
    x = rng.randn(M, K).astype(np.float32)
    w = (rng.randn(K, N) / np.sqrt(K)).astype(np.float32)
    b = rng.randn(N).astype(np.float32)
    dy = rng.randn(M, N).astype(np.float32)
    xt, wt, bt = dev(x).requires_grad_(), dev(w).requires_grad_(), dev(b).requires_grad_()
    y = ops.dense(xt, wt, bt)
    y.backward(dev(dy))
    x6, w6, b6, d6 = (a.astype(np.float64) for a in (x, w, b, dy))
    assert_close(y.detach().cpu().numpy(), x6 @ w6 + b6, "dense fwd")
    assert_close(xt.grad.cpu().numpy(), d6 @ w6.T, "dense dx")
    assert_close(wt.grad.cpu().numpy(), x6.T @ d6, "dense dw")
    assert_close(bt.grad.cpu().numpy(), d6.sum(0), "dense dbias")


def test_dense_asymmetric_identity(ops):
    """A = I with an asymmetric B catches a transposed C write (guide section 3)."""
    K = N = 96
    bmat = (np.arange(K)[:, None] * 1.0 + 0.001 * np.arange(N)[None, :] ** 2).astype(np.float32)
    y = ops.dense(dev(np.eye(K, dtype=np.float32)), dev(bmat), None).cpu().numpy()
    assert np.array_equal(y, bmat)


# ---- R3-R6 ----------------------------------------------------------------------------------
def _tail_params(nbits, rng, dtype=np.float64):
    m = 2 ** nbits
    return dict(w1=rng.uniform(-1, 1, (2, m)).astype(dtype), b1=rng.uniform(-.3, .3, m).astype(dtype),
                w2=rng.uniform(-1, 1, (m + 2, 2 * nbits)).astype(dtype),
                b2=rng.uniform(-.3, .3, 2 * nbits).astype(dtype))


def _tail_case(nbits, cells, rng):
    """Random tail weights / inputs with every pre-activation kept away from the leaky-ReLU kink
    (the derivative jumps there, so an fp32-vs-fp64 sign difference would not be a rounding-level
    effect) and every probability pair away from a tie."""
    tp = {k: v.astype(np.float32).astype(np.float64) for k, v in _tail_params(nbits, rng).items()}
    z = (rng.randn(cells, 2) * 2.0).astype(np.float32)
    bits = rng.randint(0, 2, (cells, nbits)).astype(np.int32)
    for _ in range(50):
        r = O.tail_forward_backward(z.astype(np.float64), bits, tp["w1"], tp["b1"], tp["w2"], tp["b2"], nbits)
        pr = r["prob"].reshape(cells, -1, 2)
        bad = (np.abs(r["pre1"]).min(1) < 1e-4) | (np.abs(r["pre2"]).min(1) < 1e-4) | \
              (np.abs(pr[..., 1] - pr[..., 0]).min(1) < 1e-5)
        if not bad.any():
            return tp, z, bits, r
        z[bad] = (rng.randn(int(bad.sum()), 2) * 2.0).astype(np.float32)
    raise AssertionError("could not build a well-conditioned tail case")


@pytest.mark.parametrize("nbits,cells", [(1, 36 * 320), (2, 36 * 320), (3, 36 * 320), (4, 36 * 320),
                                         (2, 1170 * 320), (4, 1170 * 320), (2, 1), (3, 77)])
def test_demod_tail_loss(ops, nbits, cells):
    rng = np.random.RandomState(10 * nbits + cells % 7)
    tp, z, bits, r = _tail_case(nbits, cells, rng)
    flat = np.concatenate([tp[k].reshape(-1) for k in ("w1", "b1", "w2", "b2")]).astype(np.float32)

    zt, ft = dev(z).requires_grad_(), dev(flat).requires_grad_()
    ce, prob, mbuf = ops.demod_tail_loss(zt, ft, dev(bits, torch.int32), nbits)
    ce.backward()
    m = ops.read_metrics(mbuf)
    assert_close(prob.cpu().numpy(), r["prob"], "prob")
    assert abs(m["ce_mean"] - r["ce_mean"]) <= 1e-5 * abs(r["ce_mean"])
    assert abs(float(ce.detach()) - r["ce_mean"]) <= 1e-5 * abs(r["ce_mean"])
    # hard decisions and the confusion matrix: bit-exact (no cell is within 1e-5 of a tie)
    pg = prob.cpu().numpy().reshape(-1, 2)
    pr = r["prob"].reshape(-1, 2)
    assert np.array_equal(pg[:, 1] > pg[:, 0], pr[:, 1] > pr[:, 0])
    assert np.array_equal(np.array(m["conf"]), r["conf"])
    assert m["count"] == cells * nbits
    assert abs(m["berlin"] - float(r["berlin"])) <= 1e-7
    assert_close(zt.grad.cpu().numpy(), r["dz"], "dz")
    gref = np.concatenate([r["grads"][k].reshape(-1) for k in ("w1", "b1", "w2", "b2")])
    assert_close(ft.grad.cpu().numpy(), gref, "tail param grads")
    # the inference variant is a different template instantiation of the same kernel: bitwise equal
    ce2, prob2, mbuf2 = ops.demod_tail_eval(dev(z), dev(flat), dev(bits, torch.int32), nbits)
    assert torch.equal(prob2, prob)
    assert ops.read_metrics(mbuf2)["conf"] == m["conf"]


# ---- R2 + R3-R6 in one launch (gemm16 EPI_TAIL) -----------------------------------------------------
@pytest.mark.parametrize("nbits,M,K,N", [(2, 36, 896, 640), (1, 36, 896, 640), (2, 1170, 896, 640), (2, 53, 100, 36),
                                         (1, 130, 64, 132), (2, 585, 128, 64),
                                         # 8-QAM / 16-QAM: tile staged through LDS, lane-per-cell / quad-lane tail
                                         (3, 36, 896, 640), (4, 36, 896, 640), (4, 300, 896, 640), (3, 300, 896, 640),
                                         (4, 53, 100, 36), (3, 130, 64, 132)])
def test_dense_tail_fused(ops, nbits, M, K, N):
    """dccn_dense_tail_fwd_bwd vs the float64 oracle (dense, then tail forward/backward): prob, ce_mean, confusion
    counts, dz through its consumers dx/dw/db, and the tail-weight gradients.  Cells whose pre-activations sit within
    rounding of a leaky-ReLU kink (derivative jumps 1 <-> 0.2) are excluded from the element-wise dz-dependent
    comparisons by construction of the inputs: z is re-drawn until every cell is well conditioned."""
    rng = np.random.RandomState(100 * nbits + M)
    D = N // 2
    tp = {k: v.astype(np.float32).astype(np.float64) for k, v in _tail_params(nbits, rng).items()}
    flat = np.concatenate([tp[k].reshape(-1) for k in ("w1", "b1", "w2", "b2")]).astype(np.float32)
    w = (rng.randn(K, N) / np.sqrt(K)).astype(np.float32)
    b = (rng.randn(N) * 0.5).astype(np.float32)
    x = (rng.randn(M, K) * 2.0).astype(np.float32)
    bits = rng.randint(0, 2, (M, D, nbits)).astype(np.int32)
    w6, b6 = w.astype(np.float64), b.astype(np.float64)
    for _ in range(60):
        z6 = x.astype(np.float64) @ w6 + b6
        r = O.tail_forward_backward(z6.reshape(-1, 2), bits.reshape(-1, nbits), tp["w1"], tp["b1"], tp["w2"], tp["b2"], nbits)
        pr = r["prob"].reshape(M * D, -1, 2)
        bad = (np.abs(r["pre1"]).min(1) < 2e-4) | (np.abs(r["pre2"]).min(1) < 2e-4) | \
              (np.abs(pr[..., 1] - pr[..., 0]).min(1) < 2e-5)
        rows = np.unique(np.nonzero(bad)[0] // D)
        if rows.size == 0:
            break
        x[rows] = (rng.randn(rows.size, K) * 2.0).astype(np.float32)
    else:
        raise AssertionError("could not build a well-conditioned case")
    xt, wt, bt, ft = (dev(a).requires_grad_() for a in (x, w, b, flat))
    ce, prob, mbuf = ops.dense_demod_tail_loss(xt, wt, bt, ft, dev(bits, torch.int32), nbits)
    ce.backward()
    m = ops.read_metrics(mbuf)
    assert_close(prob.cpu().numpy().reshape(-1, nbits, 2), r["prob"].reshape(-1, nbits, 2), "prob")
    assert abs(m["ce_mean"] - r["ce_mean"]) <= 1e-5 * abs(r["ce_mean"])
    assert np.array_equal(np.array(m["conf"]), r["conf"]) and m["count"] == M * D * nbits
    dz6 = r["dz"].reshape(M, N)
    assert_close(xt.grad.cpu().numpy(), dz6 @ w6.T, "dx", tol=2e-5)
    assert_close(wt.grad.cpu().numpy(), x.astype(np.float64).T @ dz6, "dw", tol=2e-5)
    assert_close(bt.grad.cpu().numpy(), dz6.sum(0), "db", tol=2e-5)
    gref = np.concatenate([r["grads"][k].reshape(-1) for k in ("w1", "b1", "w2", "b2")])
    assert_close(ft.grad.cpu().numpy(), gref, "tail param grads", tol=2e-5)
    # inference instantiation: same probabilities bit for bit
    with torch.no_grad():
        ce2, prob2, mbuf2 = ops.dense_demod_tail_loss(dev(x), dev(w), dev(b), dev(flat), dev(bits, torch.int32), nbits)
    assert torch.equal(prob2, prob) and ops.read_metrics(mbuf2)["conf"] == m["conf"]
    # and the two-launch path (dense, then tail) agrees to rounding
    z = ops.dense(dev(x), dev(w), dev(b))
    ce3, prob3, _ = ops.demod_tail_eval(z.view(M, D, 2), dev(flat), dev(bits, torch.int32), nbits)
    assert_close(prob3.cpu().numpy().reshape(-1), prob.cpu().numpy().reshape(-1), "fused vs separate prob", tol=5e-6)


# ---- R7 -------------------------------------------------------------------------------------
def test_adam_tf_steps(ops):
    import ctypes as C
    from dl_ofdm_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(5)
    n = 10007
    p0 = rng.randn(n).astype(np.float32)
    reg = np.zeros(n, np.float32)
    reg[1000:6000] = 2e-6
    p = {"w": p0.copy()}
    st = O.adam_init(p)
    pt, mt, vt = dev(p0), dev(np.zeros(n, np.float32)), dev(np.zeros(n, np.float32))
    state = dev(np.array([0.0, 0.9, 0.999, 0.0], np.float32))
    gate = dev(np.array([0.25], np.float32))
    hp = _lib.AdamHParams.default()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for it in range(7):
        g = (rng.randn(n) * 0.1).astype(np.float32)
        geff = g + (np.float32(0.25) * reg) * p["w"]
        O.adam_tf_step(p, {"w": geff}, st)
        gt = dev(g)
        _lib.check(lib.dccn_adam_tf_step(pt.data_ptr(), gt.data_ptr(), mt.data_ptr(), vt.data_ptr(),
                                         dev(reg).data_ptr(), gate.data_ptr(), state.data_ptr(), hp, n, s))
        torch.cuda.synchronize()
    assert_close(pt.cpu().numpy(), p["w"], "adam params", tol=2e-6)
    assert_close(mt.cpu().numpy(), st.m["w"], "adam m", tol=2e-6)
    assert_close(vt.cpu().numpy(), st.v["w"], "adam v", tol=2e-6)
    sv = state.cpu().numpy()
    assert sv[0] == 7.0 and abs(sv[1] - st.beta1_power) < 1e-7 and abs(sv[2] - st.beta2_power) < 1e-7
