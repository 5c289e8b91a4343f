"""GPU parity of the fused receiver step (R0..R7 in one launch sequence / hipGraph) against
the CPU oracle, at the BASELINE.json configurations (``-m gpu``)."""
import numpy as np
import pytest
import torch

from oracle import dccn_oracle as O

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def relerr(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.abs(got - ref).max()) / max(float(np.abs(ref).max()), 1e-30)


def make_case(batch, nbits, kin=80, F=64, D=320, S=7, seed=0):
    from dl_ofdm_amd.engine import RxDims
    rng = np.random.RandomState(seed)
    dims = RxDims(S=S, kin=kin, F=F, D=D, nbits=nbits)
    cfg = O.RxConfig(S=S, kin=kin, F=F, D=D, nbits=nbits)
    x = (rng.randn(batch, S, kin, 2) * rng.uniform(0.5, 2.0, (S, kin, 2)) + 0.1 * rng.randn(S, kin, 2)).astype(np.float32)
    bits = rng.randint(0, 2, (batch, D, nbits)).astype(np.int32)
    p = O.init_params(cfg, seed=seed + 1)
    # non-zero biases so the bias paths are exercised
    for k in p:
        if k.endswith("bias"):
            p[k] = rng.uniform(-0.05, 0.05, p[k].shape).astype(np.float32)
    # scale the tail so decisions are not all coin flips
    p["demodulation/dense_1/kernel"] = (p["demodulation/dense_1/kernel"] * 1.0).astype(np.float32)
    return dims, cfg, x, bits, p


CASES = [  # (name, batch frames, nbits, kin, F, D)
    ("C1_bpsk_256sym", 36, 1, 80, 64, 320),
    ("C2_qpsk_8192sym", 1170, 2, 80, 64, 320),
    ("C3_16qam", 1170, 4, 80, 64, 320),
    ("qam8_nocp", 100, 3, 64, 64, 320),
    ("ragged", 13, 2, 20, 12, 50),
]


def staged_checks(eng, p, x, bits, cfg, rtol=RTOL, grad_rtol=RTOL, cos_tol=1e-5):
    """End-to-end forward parity against the float64 oracle + stage-by-stage backward parity.

    The leaky-ReLU derivative jumps at 0, so an end-to-end fp32-vs-fp64 comparison of gradients is
    ill-conditioned whenever some pre-activation sits within rounding of the kink (a few cells out of
    ~10^6 always do).  Each backward stage is therefore checked against the oracle evaluated in float64
    ON THE GPU's OWN INPUTS to that stage (z, dz, dfft ...): per-stage errors stay at rounding level,
    and the composition of exact stages is the exact backward."""
    batch = x.shape[0]
    F, D, nb = cfg.F, cfg.D, cfg.nbits
    m = eng.metrics()
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    xn, _, _ = O.batch_moment_norm(x.reshape(batch, -1).astype(np.float64))
    xn = xn.reshape(x.shape)
    grads_e2e, info = O.rx_forward_backward(p64, xn, bits, cfg)
    sv = info["saved"]
    # ---- forward, end to end -------------------------------------------------------------------
    assert relerr(eng.x_norm.cpu().numpy(), xn) <= rtol
    assert relerr(eng.fft_out.cpu().numpy().reshape(-1, F, 2), sv["fft"]) <= rtol
    assert relerr(eng.z.cpu().numpy(), sv["z"]) <= rtol
    assert relerr(eng.prob.cpu().numpy(), info["prob"]) <= rtol
    assert abs(m["ce_mean"] - info["ce_mean"]) <= rtol * abs(info["ce_mean"])
    _, pw = O.complex_clip(xn, 8.0)
    assert abs(m["tx_power"] - pw) <= rtol * pw
    # hard decisions: bit-exact outside a 1e-5 probability margin; the few sub-margin cells may flip
    pg, pr = eng.prob.cpu().numpy().reshape(-1, 2), info["prob"].reshape(-1, 2)
    safe = np.abs(pr[:, 1] - pr[:, 0]) > 1e-5
    assert np.array_equal((pg[:, 1] > pg[:, 0])[safe], (pr[:, 1] > pr[:, 0])[safe])
    n_unsafe = int((~safe).sum())
    assert n_unsafe <= 5e-3 * pr.shape[0] + 4            # informational bound; flips are checked below
    assert np.abs(np.array(m["conf"]) - info["conf"]).sum() <= 2 * n_unsafe
    assert np.array(m["conf"]).sum() == batch * D * nb == m["count"]
    # ---- tail: oracle on the GPU's z ----------------------------------------------------------------
    zg = eng.z.cpu().numpy().astype(np.float64).reshape(batch * D, 2)
    tl = O.tail_forward_backward(zg, bits.reshape(batch * D, nb), p64["demodulation/conv2d/kernel"],
                                 p64["demodulation/conv2d/bias"], p64["demodulation/dense_1/kernel"],
                                 p64["demodulation/dense_1/bias"], nb)
    pg4 = eng.prob.cpu().numpy().reshape(-1, 2)
    pt = tl["prob"].reshape(-1, 2)
    tie = np.abs(pt[:, 1] - pt[:, 0]) < 1e-6
    assert np.array_equal((pg4[:, 1] > pg4[:, 0])[~tie], (pt[:, 1] > pt[:, 0])[~tie])
    assert np.abs(np.array(m["conf"]) - tl["conf"]).sum() <= 2 * int(tie.sum())
    kink = (np.abs(tl["pre1"]).min(1) < 2e-6) | (np.abs(tl["pre2"]).min(1) < 2e-6)     # ~0-3 cells
    dzg = eng.dz.cpu().numpy().reshape(batch * D, 2)
    assert relerr(dzg[~kink], tl["dz"][~kink]) <= rtol
    g = eng.get_grads()
    if not kink.any():
        for name, key in (("demodulation/conv2d/kernel", "w1"), ("demodulation/conv2d/bias", "b1"),
                          ("demodulation/dense_1/kernel", "w2"), ("demodulation/dense_1/bias", "b2")):
            assert relerr(g[name], tl["grads"][key]) <= grad_rtol, name
    # ---- dense backward: oracle on the GPU's dz / fft_out ------------------------------------------
    dz64 = eng.dz.cpu().numpy().astype(np.float64)
    a64 = eng.fft_out.cpu().numpy().astype(np.float64).reshape(batch, -1)
    assert relerr(g["demodulation/dense/kernel"], a64.T @ dz64) <= grad_rtol
    assert relerr(g["demodulation/dense/bias"], dz64.sum(0)) <= grad_rtol
    assert relerr(eng.dfft.cpu().numpy().reshape(batch, -1), dz64 @ p64["demodulation/dense/kernel"].T) <= grad_rtol
    # ---- C-Conv weight gradient: oracle on the GPU's dfft / x_norm ---------------------------------
    xr = eng.x_norm.cpu().numpy().astype(np.float64).reshape(batch * cfg.S, cfg.kin, 2)
    _, gw, gb = O.cconv_gemm_bwd(xr, p64["fft_like/conv3d/kernel"],
                                 eng.dfft.cpu().numpy().astype(np.float64).reshape(-1, F, 2))
    assert relerr(g["fft_like/conv3d/kernel"], gw) <= grad_rtol
    assert relerr(g["fft_like/conv3d/bias"], gb) <= grad_rtol
    # ---- end-to-end gradient sanity (direction), tolerant of kink flips ------------------------------
    rs = float(info["berlin"]) * O.REG_COEFF * 2.0 * O.REG_L2
    ge = np.concatenate([(grads_e2e[k] - (rs * p64[k] if k in O.REGULARIZED else 0.0)).reshape(-1) for k in g])
    gg = np.concatenate([g[k].reshape(-1).astype(np.float64) for k in g])
    cos = float(ge @ gg / (np.linalg.norm(ge) * np.linalg.norm(gg)))
    assert cos > 1.0 - cos_tol, cos
    return m, g, info


@pytest.mark.parametrize("name,batch,nbits,kin,F,D", CASES)
def test_train_step_matches_oracle(name, batch, nbits, kin, F, D):
    from dl_ofdm_amd.engine import RxEngine
    dims, cfg, x, bits, p = make_case(batch, nbits, kin, F, D)
    eng = RxEngine(dims, batch, params=p, train=True)
    eng.train_step(x, bits)
    torch.cuda.synchronize()
    m, g, info = staged_checks(eng, p, x, bits, cfg)
    # parameters after the Adam step: oracle Adam (float32) on the gradient the kernel used
    # (ce gradient from the arena + the BER-gated L2 term of ofdmreceiver_np.py:171)
    p32 = {k: v.copy() for k, v in p.items()}
    st = O.adam_init(p32)
    berl = np.float32(m["berlin"])
    geff = {k: (g[k] + (berl * np.float32(2e-6)) * p[k] if k in O.REGULARIZED else g[k]) for k in g}
    O.adam_tf_step(p32, geff, st)
    newp = eng.get_params()
    for k in newp:
        assert relerr(newp[k], p32[k]) <= 2e-6, k
    a = eng.adam()
    assert a["global_step"] == 1.0 and abs(a["beta1_power"] - 0.81) < 1e-6


def test_trajectory_five_steps_c2():
    """5 consecutive train steps (fresh data each step): parameters track the float64 oracle."""
    from dl_ofdm_amd.engine import RxEngine
    dims, cfg, x, bits, p = make_case(256, 2, seed=3)
    eng = RxEngine(dims, 256, params=p, train=True)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    st = O.adam_init(p64)
    rng = np.random.RandomState(7)
    for it in range(5):
        xi = (x + 0.3 * rng.randn(*x.shape)).astype(np.float32)
        bi = rng.randint(0, 2, bits.shape).astype(np.int32)
        eng.train_step(xi, bi)
        info = O.rx_train_step(p64, st, xi.astype(np.float64), bi, cfg)
        m = eng.metrics()
        assert abs(m["ce_mean"] - info["ce_mean"]) <= 1e-5 * abs(info["ce_mean"])
    # Adam normalises by sqrt(v): an element whose gradient sits at fp32 rounding level can move by up
    # to one lr step differently, so parameters are compared by quantile (loss path above is strict)
    newp = eng.get_params()
    diff = np.concatenate([np.abs(newp[k] - p64[k]).reshape(-1) for k in newp])
    assert np.quantile(diff, 0.999) <= 5e-5 and diff.max() <= 5e-3, (np.quantile(diff, 0.999), diff.max())


def test_graph_replay_equals_eager():
    from dl_ofdm_amd.engine import RxEngine
    dims, cfg, x, bits, p = make_case(300, 2, seed=5)
    outs = []
    for graph, fork in ((False, False), (True, False), (True, True)):
        eng = RxEngine(dims, 300, params=p, train=True)
        for _ in range(3):
            eng.train_step(x, bits, graph=graph, fork=fork)
        torch.cuda.synchronize()
        outs.append((eng.params.clone(), eng.prob.clone(), eng.metrics()))
    # captured replay = the same launches: bitwise.  The forked two-stream graph runs the round-2 plan (dX, dW and the
    # C-Conv weight gradient as separate launches instead of the fused backward launch): fp32 sums regrouped only.
    assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1])
    assert outs[1][2]["conf"] == outs[0][2]["conf"]
    assert float((outs[2][0] - outs[0][0]).abs().max()) <= 2e-6 * float(outs[0][0].abs().max())
    assert float((outs[2][1] - outs[0][1]).abs().max()) <= 2e-6
    e = RxEngine(dims, 300, params=p, train=True)             # and the forked graph is deterministic in itself
    for _ in range(3):
        e.train_step(x, bits, graph=True, fork=True)
    torch.cuda.synchronize()
    assert torch.equal(e.params, outs[2][0]) and torch.equal(e.prob, outs[2][1])


def test_eval_step_deterministic_and_matches_train_forward():
    from dl_ofdm_amd.engine import RxEngine
    dims, cfg, x, bits, p = make_case(500, 2, seed=6)
    e1 = RxEngine(dims, 500, params=p, train=False)
    e1.eval_step(x, bits)
    a = e1.prob.clone()
    e1.eval_step(x, bits)
    assert torch.equal(a, e1.prob)
    e1.eval_step(x, bits, graph=True)
    assert torch.equal(a, e1.prob)
    e2 = RxEngine(dims, 500, params=p, train=True)
    e2.train_step(x, bits)
    assert torch.equal(a, e2.prob)
    assert e1.metrics()["conf"] == e2.metrics()["conf"]


def test_large_fft_config_c4_slice():
    """BASELINE config 4 geometry (N=1024, CP=72, F=1024, D=4000) on a reduced batch: exercises the
    big-K GEMMs, the 128x128 tile path and the split-K reductions."""
    from dl_ofdm_amd.engine import RxEngine
    batch = 40
    dims, cfg, x, bits, p = make_case(batch, 2, kin=1096, F=1024, D=4000, seed=8)
    eng = RxEngine(dims, batch, params=p, train=True, want_prob=True)
    eng.train_step(x, bits)
    torch.cuda.synchronize()
    staged_checks(eng, p, x, bits, cfg, rtol=2e-5, grad_rtol=2e-5, cos_tol=1e-4)    # K up to 14336: a little more fp32 rounding


def test_fused_backward_without_dfft_and_against_the_separate_launches():
    """The backward half as one launch (rx_bwd.h): with or without the dfft store the step is the same bits; against
    the round-2 plan (tuning key 11 = 0: dX, then the C-Conv weight gradient as its own split-K launch) it regroups
    fp32 sums only.  Shapes: QPSK with the cyclic prefix (2kin = 160: the odd 32-row unit), 8-QAM without (2kin = 128),
    a last row tile of 18 frames (1170 = 18*64 + 18) and of 36."""
    from dl_ofdm_amd import _lib
    from dl_ofdm_amd.engine import RxEngine
    lib = _lib.load()
    for batch, nbits, kin in ((1170, 2, 80), (100, 3, 64), (36, 1, 80)):
        dims, cfg, x, bits, p = make_case(batch, nbits, kin=kin, seed=21)
        assert lib.dccn_rx_bwd_fused_supported(__import__("ctypes").byref(RxEngine(dims, batch, params=p).shape)) == 1

        def run(**kw):
            e = RxEngine(dims, batch, params=p, train=True, want_prob=True, **kw)
            for _ in range(2):
                e.train_step(x, bits)
            torch.cuda.synchronize()
            return e
        a, b = run(want_dfft=True), run(want_dfft=False)
        assert b.dfft is None
        assert torch.equal(a.params, b.params) and torch.equal(a.grads, b.grads) and torch.equal(a.adam_m, b.adam_m)
        try:
            assert lib.dccn_set_tuning(11, 0) == 0
            c = run(want_dfft=True)
        finally:
            lib.dccn_set_tuning(11, 1)
        # (step 2 runs on parameters that already differ in the last bits)
        assert float((a.dfft - c.dfft).abs().max()) <= 2e-5 * float(c.dfft.abs().max())
        for name in ("fft_like/conv3d/kernel", "fft_like/conv3d/bias", "demodulation/dense/kernel"):
            ga, gc = a.view(name, a.grads), c.view(name, c.grads)
            assert float((ga - gc).abs().max()) <= 2e-6 * float(gc.abs().max()), (batch, name)
        assert float((a.params - c.params).abs().max()) <= 2e-6 * float(c.params.abs().max())


PIPE_CASES = [  # (frames, nbits, kin, F, D): N=64 shapes + the geometries bench.py times (BASELINE configs[1..3])
    (1170, 2, 80, 64, 320), (300, 2, 80, 64, 320), (64, 4, 80, 64, 320), (2000, 1, 80, 64, 320),
    (1170, 4, 80, 64, 320),            # C3: 16-QAM, 1170 frames (separate tail launch)
    (300, 2, 64, 64, 320),             # cp=False: the C-Conv sees the 64 samples behind the prefix
    (585, 2, 1096, 1024, 4000),        # C4: N=1024/CP=72, 585 frames (128x128x32 tiles, R0 over 30 688 columns)
]


@pytest.mark.parametrize("frames,nbits,kin,F,D", PIPE_CASES)
def test_pipelined_normalisation_is_bitwise_the_plain_step(frames, nbits, kin, F, D):
    """dccn_rx_buffers.x_next / x_prenormalised: R0 of batch t+1 rides on the Adam launch of step t.  Same kernels, same
    order of arithmetic -> parameters, gradients, probabilities, metrics and the optimizer state are bit-identical to
    plain steps fed the same batches (2000 frames: the two-kernel normalisation; step 3-4: hipGraph replay; the label
    slots alternate like in receiver.train's device-data loop; last=True ends the pipeline)."""
    from dl_ofdm_amd.engine import RxDims, RxEngine
    dims = RxDims(S=7, kin=kin, F=F, D=D, nbits=nbits)
    rng = np.random.RandomState(5)
    nb = 6 if F <= 64 else 4                                  # the N=1024 case: four batches of 180 MB are enough
    xs = [rng.standard_normal((frames, 7, kin, 2)).astype(np.float32) * (1 + 0.1 * t) for t in range(nb)]
    bs = [rng.randint(0, 2, (frames, D, nbits)).astype(np.int32) for t in range(nb)]
    a = RxEngine(dims, frames, train=True, seed=3, want_prob=True, want_tx_power=True)
    # the pipelined engine as bench.py builds it: no z, no dfft
    b = RxEngine(dims, frames, train=True, seed=3, want_prob=True, want_tx_power=True, want_z=False, want_dfft=False)
    b.prime(xs[0])
    for t in range(nb):
        a.train_step(xs[t], bs[t])
        if t < nb - 3:   # labels through the alternating slots, next input copied by the call
            b.train_step_pipelined(next_x=xs[t + 1], bits=bs[t], slot=t & 1)
        elif t < nb - 1:  # captured replay (slot 0)
            b.train_step_pipelined(next_x=xs[t + 1], bits=bs[t], graph=True)
        else:            # end of the epoch: nothing is prefetched, the next call would prime again
            b.train_step_pipelined(bits=bs[t], last=True)
        torch.cuda.synchronize()
        assert torch.equal(a.params, b.params) and torch.equal(a.grads, b.grads), t
        assert torch.equal(a.prob, b.prob) and torch.equal(a.adam_state, b.adam_state), t
        assert a.metrics() == b.metrics(), t
    assert not b._norm_ready and not b._prefetch_pending
    a.train_step(xs[1], bs[1])
    b.train_step_pipelined(next_x=xs[2], bits=bs[1])          # primes itself from eng.x ...
    torch.cuda.synchronize()
    assert not torch.equal(a.params, b.params)                # ... which still holds the last batch: a different one, by design
    # a prefetched batch is pending now: any other kind of step on this engine would silently clobber it -> loud
    from dl_ofdm_amd._lib import DccnError
    for bad in (lambda: b.eval_step(), lambda: b.train_step(), lambda: b.set_batch(xs[0], bs[0])):
        with pytest.raises(DccnError):
            bad()
    b.drop_prefetch()
    b.train_step(xs[0], bs[0])


@pytest.mark.parametrize("placement", [1, 2])
def test_double_buffered_normalisation_on_the_backward_launch_is_bitwise(placement):
    """Tuning knob 18: R0 of the next batch rides on the backward launch into a second x_norm buffer instead of the optimizer
    launch (1: on the leading workgroups of that grid, 2: on its closing ones, in the slots the last dW items free); eager
    double-buffered steps, captured single-buffer replays of either parity and the closing step interleave freely and stay
    bit-identical to plain steps."""
    from dl_ofdm_amd import _lib
    from dl_ofdm_amd.engine import RxDims, RxEngine
    lib = _lib.load()
    default = lib.dccn_get_tuning(18)
    try:
        assert lib.dccn_set_tuning(18, placement) == 0
        frames, nbits = 300, 2
        dims = RxDims(S=7, kin=80, F=64, D=320, nbits=nbits)
        rng = np.random.RandomState(10)
        xs = [rng.standard_normal((frames, 7, 80, 2)).astype(np.float32) for _ in range(8)]
        bs = [rng.randint(0, 2, (frames, 320, nbits)).astype(np.int32) for _ in range(8)]
        a = RxEngine(dims, frames, train=True, seed=4, want_prob=True)
        b = RxEngine(dims, frames, train=True, seed=4, want_prob=True, want_z=False, want_dfft=False)
        assert b._ride == 1 and b._norm_bufs[1] is not None
        b.prime(xs[0])
        modes = ["e", "g", "g", "e", "e", "g", "e", "last"]       # eager flips the buffer parity, replays keep it
        for t, mode in enumerate(modes):
            a.train_step(xs[t], bs[t])
            if mode == "last":
                b.train_step_pipelined(bits=bs[t], last=True)
            else:
                b.train_step_pipelined(next_x=xs[t + 1], bits=bs[t], graph=(mode == "g"))
            torch.cuda.synchronize()
            assert torch.equal(a.params, b.params) and torch.equal(a.prob, b.prob) and a.metrics() == b.metrics(), (t, mode)
    finally:
        lib.dccn_set_tuning(18, default)


def test_large_layer_optimizer_stream_overlap_is_bitwise_neutral():
    """Tuning knob 25 (include/dccn.h "eager launch sequences"): large layers run the dense kernel's Adam update on the
    library's own low-priority stream next to the C-Conv weight-gradient launch, with non-temporal loads and stores (2) or
    plain ones (1).  Same kernel on the same operands: three steps (plain and pipelined) must leave parameters, both Adam
    moments, probabilities and metrics bit-identical to the one-stream order (0) -- a missing fork / join edge shows up as a
    half-updated kernel, a stale BER gate or a wrong alpha.  N = 512 / CP = 40 / D = 2000 with 48 frames: 1792 dW tiles of
    128x128, one k range."""
    from dl_ofdm_amd import _lib
    from dl_ofdm_amd.engine import RxDims, RxEngine
    lib = _lib.load()
    dims = RxDims(S=7, kin=552, F=512, D=2000, nbits=2)
    frames = 48
    rng = np.random.RandomState(5)
    xs = [rng.standard_normal((frames, 7, 552, 2)).astype(np.float32) for _ in range(3)]
    bs = [rng.randint(0, 2, (frames, 2000, 2)).astype(np.int32) for _ in range(3)]

    def run(pipelined):
        e = RxEngine(dims, frames, train=True, seed=4, want_prob=True)
        if pipelined:
            e.prime(xs[0])
            for k in range(3):
                e.train_step_pipelined(next_x=xs[(k + 1) % 3], bits=bs[k], last=(k == 2))
        else:
            for k in range(3):
                e.train_step(xs[k], bs[k])
        torch.cuda.synchronize()
        return e.params.clone(), e.adam_m.clone(), e.adam_v.clone(), e.prob.clone(), e.metrics()
    default = lib.dccn_get_tuning(25)
    try:
        out = {}
        for v in (0, 1, 2):
            assert lib.dccn_set_tuning(25, v) == 0
            out[v] = (run(False), run(True))
    finally:
        lib.dccn_set_tuning(25, default)
    for v in (1, 2):
        for mode in (0, 1):
            for a, b in zip(out[0][mode][:4], out[v][mode][:4]):
                assert torch.equal(a, b), (v, mode)
            assert out[0][mode][4] == out[v][mode][4]
    for a, b in zip(out[0][0][:3], out[0][1][:3]):          # and the pipelined order equals the plain one
        assert torch.equal(a, b)


@pytest.mark.parametrize("key,value", [(0, 0), (1, 0), (1, 3), (2, 1), (3, 0), (3, 1), (4, 3), (5, 32), (7, 0), (8, 0),
                                       (9, 0), (10, 0), (11, 0), (12, 0), (14, 0), (14, 3)])
def test_every_tuning_setting_computes_the_same_step(key, value):
    """dccn_set_tuning only selects tile configurations: two training steps under any setting agree with the default
    ones to rounding (the settings that keep the summation order are bitwise equal; the others regroup fp32 sums)."""
    from dl_ofdm_amd import _lib
    from dl_ofdm_amd.engine import RxDims, RxEngine
    lib = _lib.load()
    dims = RxDims(S=7, kin=80, F=64, D=320, nbits=2)
    rng = np.random.RandomState(11)
    x = rng.standard_normal((300, 7, 80, 2)).astype(np.float32)
    bits = rng.randint(0, 2, (300, 320, 2)).astype(np.int32)

    def run():
        e = RxEngine(dims, 300, train=True, seed=2, want_prob=True)
        for _ in range(2):
            e.train_step(x, bits)
        torch.cuda.synchronize()
        return e.params.clone(), e.prob.clone(), e.metrics()
    default = lib.dccn_get_tuning(key)
    p0, q0, m0 = run()
    try:
        assert lib.dccn_set_tuning(key, value) == 0
        p1, q1, m1 = run()
    finally:
        lib.dccn_set_tuning(key, default)
    assert float((p1 - p0).abs().max()) <= 2e-6 * float(p0.abs().max())
    assert float((q1 - q0).abs().max()) <= 2e-6
    assert abs(m1["ce_mean"] - m0["ce_mean"]) <= 1e-6 and abs(sum(sum(r) for r in m1["conf"]) - sum(sum(r) for r in m0["conf"])) == 0


def test_a_pinned_plan_is_immune_to_knobs_flipped_by_another_thread():
    """The tuning knobs are process-global defaults; a plan that captured its own table (RxEngine.pin_tuning ->
    dccn_rx_buffers.tuning) keeps planning from it whatever dccn_set_tuning calls another thread issues meanwhile: thread B
    flips the knobs that change the launch plan and the summation order (fused / grouped backward, graded dW ranges, C-Conv
    tiles, dense tile family) as fast as it can while thread A trains -- A's parameters are the bits of a quiet run.  An
    UNPINNED engine under the same fire still computes a valid step every call (a call copies the table once, never reads it
    mid-plan): within rounding of the quiet run."""
    import threading
    from dl_ofdm_amd import _lib
    from dl_ofdm_amd.engine import RxDims, RxEngine
    lib = _lib.load()
    dims = RxDims(S=7, kin=80, F=64, D=320, nbits=2)
    rng = np.random.RandomState(3)
    x = rng.standard_normal((300, 7, 80, 2)).astype(np.float32)
    bits = rng.randint(0, 2, (300, 320, 2)).astype(np.int32)

    def run(pin, steps=40):
        e = RxEngine(dims, 300, train=True, seed=2, want_prob=False)
        if pin:
            e.pin_tuning()
        e.set_batch(x, bits)
        for _ in range(steps):
            e.train_step()
        torch.cuda.synchronize()
        return e.params.clone(), e.adam_m.clone(), e.adam_v.clone()

    quiet = run(True)
    quiet2 = run(True, steps=2)
    flips = [(11, (0, 1)), (14, (0, 14)), (12, (0, 3)), (2, (1, 7)), (0, (2, 9)), (1, (0, 7)), (7, (0, 1))]
    defaults = {k: lib.dccn_get_tuning(k) for k, _ in flips}
    stop = threading.Event()
    count = [0]

    def fire():
        i = 0
        while not stop.is_set():
            k, vals = flips[i % len(flips)]
            lib.dccn_set_tuning(k, vals[(i // len(flips)) & 1])
            i += 1
        count[0] = i

    th = threading.Thread(target=fire)
    pinned = None
    try:
        e = RxEngine(dims, 300, train=True, seed=2, want_prob=False).pin_tuning()          # (captured BEFORE the fire starts)
        e.set_batch(x, bits)
        th.start()
        for _ in range(40):
            e.train_step()
        torch.cuda.synchronize()
        pinned = (e.params.clone(), e.adam_m.clone(), e.adam_v.clone())
        loose = run(False, steps=2)
    finally:
        stop.set()
        th.join()
        for k, v in defaults.items():
            lib.dccn_set_tuning(k, v)
    assert count[0] > 1000
    for a, b in zip(quiet, pinned):
        assert torch.equal(a, b)
    # (two steps, as test_every_tuning_setting_computes_the_same_step: the plans regroup fp32 sums, and Adam's first updates are
    # ~lr * sign(g) -- over tens of steps a rounding-level difference in a near-zero gradient becomes an lr-sized one)
    assert float((loose[0] - quiet2[0]).abs().max()) <= 2e-6 * float(quiet2[0].abs().max())
    assert all(lib.dccn_get_tuning(k) == v for k, v in defaults.items())
