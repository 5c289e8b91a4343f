"""SURVEY.md 8(f-4) on the GPU (``-m gpu``): the classical pilot-aided receivers as libdccn launches
(dl_ofdm_amd.benchmark_gpu, include/dccn.h "classical pilot-aided receivers") against their NumPy restatement
(dl_ofdm_amd.benchmark = the oracle of this path; itself pinned by closed forms in tests/test_benchmark.py), on the same
received frames -- produced by the device-side generator, so nothing crosses PCIe in the product path.

Tolerance: the detected bits are a hard decision on x = Y / G; fp32 (GPU) and fp64 (oracle) agree except for cells within
rounding of a decision boundary: <= 3e-4 of the bits may differ and the two error counts within 5e-4 of the bit count."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [(2, "EPA", 15.0), (4, "EVA", 25.0), (1, "ETU", 10.0), (3, "AWGN", 8.0), (2, "Flat", 12.0)]


@pytest.mark.parametrize("nbits,channel,snr", CASES)
def test_every_gpu_estimator_matches_the_numpy_receiver(nbits, channel, snr):
    from dl_ofdm_amd import benchmark as B, benchmark_gpu as G, ofdm, receiver as R
    from dl_ofdm_amd.datagen import DeviceDataGen
    F = R.Flags(nbits=nbits, channel=channel, nfilter=64)
    o = ofdm.ofdm_tx(F)
    rx = G.ClassicalReceiverGPU(F, o)
    host = rx.host
    gen = DeviceDataGen(F, o, device="cuda", seed=7 + nbits)
    n = 600
    x, bits, _, H = gen.make_batch(n, snr, want_H=True)
    xh, bh = x.cpu().numpy(), bits.cpu().numpy()
    Hh = H.cpu().numpy()
    Hs = np.repeat(Hh[:, None, :], o.nSymbol, axis=1) if Hh.ndim == 2 else Hh
    adv = 0 if channel == "AWGN" else host.advance_of(rx.fading)
    assert adv == rx.advance
    total = n * o.frame_size * nbits
    for method in G.GPU_METHODS:
        if channel == "AWGN" and method in ("LMMSE-Fast", "LMMSE-UniPDP", "LMMSE-ExpPDP"):
            continue                                        # no power-delay profile to smooth with
        if method == "LMMSE-Fast":
            R_long = host.long_term_correlation(rx.fading)
        elif method in ("LMMSE-UniPDP", "LMMSE-ExpPDP"):
            R_long = host.pdp_correlation(rx.fading, uniform=(method == "LMMSE-UniPDP"))
        else:
            R_long = None
        # the cyclic-prefix methods solve two least-squares systems per symbol: the NumPy side takes seconds per hundred
        # frames, and the solve amplifies the fp32 rounding of the starting estimate -- 150 frames, 1e-3 of the bits
        m = 150 if method.endswith("-CP") else n
        tol = 1e-3 if method.endswith("-CP") else 3e-4
        ref = host.receive(xh[:m], method, snr, H_true=Hs[:m], R_long=R_long, advance=adv)
        err, cnt, det = rx.receive(x[:m], bits[:m], method, snr, H_true=H[:m], want_bits=True)
        det = det.cpu().numpy()
        tot_m = m * o.frame_size * nbits
        assert cnt == tot_m and det.shape == ref.shape
        assert err == int(np.count_nonzero(det != bh[:m]))                # the device's own count is exact
        differ = float(np.mean(det != ref))
        assert differ <= tol, (method, differ)
        assert abs(err - int(np.count_nonzero(ref != bh[:m]))) <= (tol + 2e-4) * tot_m + 2, method
    # perfect channel knowledge is the lower bound of the family
    e_perf = rx.receive(x, bits, "Perfect", snr, H_true=H)[0]
    e_ls = rx.receive(x, bits, "LS-Spline", snr)[0]
    assert e_perf <= e_ls + 2e-4 * total


def test_unaligned_window_and_curve_points():
    """aligned=False keeps radio.py's centred filter timing (the round-1 tables); CurvePointsGPU = one SNR point of a
    curve as an independent, seeded unit (what config 5 deals to the ranks)."""
    from dl_ofdm_amd import benchmark_gpu as G, ofdm, receiver as R
    from dl_ofdm_amd.datagen import DeviceDataGen
    F = R.Flags(nbits=2, channel="EVA", nfilter=64)
    o = ofdm.ofdm_tx(F)
    rx = G.ClassicalReceiverGPU(F, o)
    gen = DeviceDataGen(F, o, device="cuda", seed=3)
    x, bits, _, H = gen.make_batch(400, 30.0, want_H=True)
    Hs = np.repeat(H.cpu().numpy()[:, None, :], o.nSymbol, axis=1)
    for method in ("Perfect", "LS-Spline", "ALMMSE"):
        ref = rx.host.receive(x.cpu().numpy(), method, 30.0, H_true=Hs, advance=rx.advance, aligned=False)
        err, cnt, det = rx.receive(x, bits, method, 30.0, H_true=H, aligned=False, want_bits=True)
        assert float(np.mean(det.cpu().numpy() != ref)) <= 3e-4, method
    # at 30 dB on EVA the centred filter's pre-cursor ISI leaves an error floor that the aligned window removes
    e_un = rx.receive(x, bits, "Perfect", 30.0, H_true=H, aligned=False)[0]
    e_al = rx.receive(x, bits, "Perfect", 30.0, H_true=H, aligned=True)[0]
    assert e_al < e_un
    cp = G.CurvePointsGPU(F, "LS-Spline", n_frames=300, seed=5)
    a, b = cp.point(2, 10.0), cp.point(2, 10.0)
    assert a == b and a[1] == 300 * 320 * 2 and 0 < a[0] < a[1] // 2                     # seeded: reproducible
    assert cp.point(3, 25.0)[0] < a[0]                                                     # BER falls with SNR
    with pytest.raises(NotImplementedError):
        rx.receive(x, bits, "LS-Cubic", 10.0)
    with pytest.raises(ValueError):
        rx.receive(x, bits, "LS-CP", 10.0, aligned=False)                                  # the prefix equations are causal


def test_interpolation_and_correlation_matrices_stand_on_their_own():
    """The constant matrices of the device path are built in benchmark_gpu.py, not borrowed from the NumPy receiver they are
    checked against (VERDICT r03 Weak 10): (1) biharmonic-spline weights reproduce the pilot values exactly, are exact for the
    spline's own basis functions, and agree with the host's independent construction; linear weights are a partition of unity
    that reproduces affine functions inside the pilots' hull; (2) the long-term correlation E[h h^H] built from the profile's tap
    matrices equals the SAMPLE covariance of the device generator's own channel responses (20 000 draws), i.e. formula and
    generator agree without either being the other's input."""
    from dl_ofdm_amd import benchmark_gpu as G, ofdm, receiver as R
    from dl_ofdm_amd.datagen import DeviceDataGen
    F = R.Flags(nbits=2, channel="EVA", nfilter=64)
    o = ofdm.ofdm_tx(F)
    rx = G.ClassicalReceiverGPU(F, o)
    S, K, pil = rx.S, rx.K, rx.host.pil
    W = rx.w_spline64.cpu().numpy()                                  # [S K, P]
    assert np.abs(W[pil] - np.eye(len(pil))).max() <= 1e-8
    pts = np.stack([pil % K, pil // K], -1).astype(np.float64)
    cells = np.arange(S * K)
    grid = np.stack([cells % K, cells // K], -1).astype(np.float64)
    rng = np.random.RandomState(0)
    w = rng.standard_normal(len(pil))                                # a function in the span of the Green's functions ...
    def green(r):
        return np.where(r > 0, r * r * (np.log(np.maximum(r, 1e-300)) - 1.0), 0.0)
    f_grid = green(np.linalg.norm(grid[:, None] - pts[None], axis=-1)) @ w
    f_pil = green(np.linalg.norm(pts[:, None] - pts[None], axis=-1)) @ w
    assert np.abs(W @ f_pil - f_grid).max() <= 1e-7 * np.abs(f_grid).max()      # ... is interpolated exactly
    assert np.abs(W - rx.host.W_spline).max() <= 1e-7 * np.abs(W).max()         # two independent constructions agree
    Wl = G.linear_weights(pil, S, K)
    assert np.abs(Wl.sum(1) - 1.0).max() <= 1e-12 and Wl.min() >= -1e-12
    inside = np.abs(Wl).max(1) < 1.0 - 1e-9                                      # cells that really are interpolated
    aff = 0.3 * grid[:, 0] - 1.7 * grid[:, 1] + 2.0
    assert inside.sum() > 50 and np.abs((Wl @ (0.3 * pts[:, 0] - 1.7 * pts[:, 1] + 2.0) - aff)[inside]).max() <= 1e-9
    # (2) correlation from the tap matrices vs the generator's own draws
    Rl = rx.long_term_correlation()
    gen = DeviceDataGen(F, o, device="cuda", seed=11)
    _, _, _, H = gen.make_batch(20000, 30.0, want_H=True)
    Hc = H.cpu().numpy().astype(np.complex128) * rx.ramp()[None, :]             # same phase reference (centre tap)
    Rs = (Hc.T @ Hc.conj()) / Hc.shape[0]
    assert np.abs(Rs - Rl).max() <= 0.04 * np.abs(Rl).max()                     # 20 000 draws: ~1 % per entry, 4 % worst
    assert abs(np.trace(Rs).real / np.trace(Rl).real - 1.0) <= 0.02
    assert np.abs(Rl - rx.host.long_term_correlation(rx.fading)).max() <= 1e-9 * np.abs(Rl).max()
