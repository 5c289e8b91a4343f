"""Classical LS / LMMSE baseline receivers (dl_ofdm_amd/benchmark.py): closed-form pins.
Unpinned against MATLAB (the reference's dev/m scripts need MATLAB + Communications Toolbox)."""
import math
import os

import numpy as np
import pytest

from dl_ofdm_amd import benchmark as B
from dl_ofdm_amd import ofdm
from dl_ofdm_amd.receiver import Flags


def qfunc(x):
    return 0.5 * math.erfc(x / math.sqrt(2.0))


def test_interpolation_matrices():
    F = Flags(nbits=2, channel="EPA")
    r = B.ClassicalReceiver(F)
    P = len(r.pil)
    assert r.W_spline.shape == (7 * 64, P) and r.W_linear.shape == (7 * 64, P)
    assert np.abs(r.W_spline[r.pil] - np.eye(P)).max() < 1e-8          # both interpolants pass through the pilots
    assert np.abs(r.W_linear[r.pil] - np.eye(P)).max() < 1e-12
    assert np.abs(r.W_linear.sum(axis=1) - 1.0).max() < 1e-12          # convex combinations / nearest pilot
    assert abs(r.beta - 1.0) < 1e-12                                   # QPSK: constant modulus
    assert abs(B.ClassicalReceiver(Flags(nbits=4)).beta - 17.0 / 9.0) < 1e-6
    assert B.EST_NAMES[0] == "Perfect" and len(B.EST_NAMES) == 10


@pytest.mark.parametrize("nbits,snr", [(1, 0.0), (2, 3.0)])
def test_perfect_csi_awgn_matches_closed_form(nbits, snr):
    F = Flags(nbits=nbits, channel="AWGN")
    ber = B.ber_curve(F, "Perfect", [snr], n_frames=1500, seed=3)[0]
    sigma2 = 10.0 ** (-snr / 10.0)
    theory = qfunc(math.sqrt((64.0 / 24.0 if nbits == 1 else 64.0 / 48.0) / sigma2))
    n_bits = 1500 * 320 * nbits
    assert abs(ber - theory) <= 4 * math.sqrt(theory * (1 - theory) / n_bits) + 0.02 * theory, (ber, theory)
    # the pilot-based estimators see the same flat response: LS costs only the estimation noise
    ls = B.ber_curve(F, "LS-Spline", [snr], n_frames=300, seed=3)[0]
    assert theory * 0.9 <= ls <= 3.5 * theory + 0.01


def test_perfect_csi_flat_rayleigh_matches_closed_form():
    """coherent BPSK over flat Rayleigh fading: Pb = (1 - sqrt(g/(1+g)))/2 with g the mean per-bit SNR"""
    F = Flags(nbits=1, channel="Flat")
    snr = 10.0
    ber = B.ber_curve(F, "Perfect", [snr], n_frames=6000, seed=5)[0]
    g = (64.0 / 48.0) / 10.0 ** (-snr / 10.0)
    theory = 0.5 * (1.0 - math.sqrt(g / (1.0 + g)))
    assert abs(ber - theory) <= 0.1 * theory, (ber, theory)


def test_estimator_ordering_on_a_multipath_channel():
    F = Flags(nbits=2, channel="EPA")
    snr = [20.0]
    ber = {m: B.ber_curve(F, m, snr, n_frames=800, seed=9)[0] for m in B.EST_NAMES}
    assert all(0.0 <= v < 0.3 for v in ber.values()), ber
    assert ber["Perfect"] <= ber["LS-Spline"] + 1e-4 and ber["Perfect"] <= ber["LS-Linear"] + 1e-4, ber
    assert ber["LMMSE"] <= ber["LS-Spline"] + 1e-4, ber                # knowing h exactly can only help
    assert ber["LMMSE-Fast"] <= 1.5 * ber["LS-Spline"] + 1e-3, ber


def test_gray_qam_tables_are_gray_and_separate_from_the_dccn_tables():
    """MATLAB-style qammod(.,'gray') constellations: nearest neighbours differ in exactly one bit; they are NOT ofdm.py's
    tables (keep the two mappings apart)."""
    for nb in (1, 2, 3, 4):
        pts, lab = B.gray_qam_table(nb)
        assert len(pts) == 2 ** nb and len(set(np.round(pts, 6))) == 2 ** nb
        assert [int("".join(map(str, r)), 2) for r in lab] == list(range(2 ** nb))
        d = np.abs(pts[:, None] - pts[None, :])
        near = np.isclose(d, 2.0)
        assert near.any()
        assert np.all((lab[:, None, :] != lab[None, :, :]).sum(-1)[near] == 1)
    assert not np.allclose(B.gray_qam_table(4)[0], ofdm.const_map(4))
    r = B.ClassicalReceiver(Flags(nbits=4), mapping="gray")
    assert abs(r.pilot_value - 3.0 * np.sqrt(2.0) * np.sqrt(0.5) * (1 + 1j)) < 1e-12 and r.papr_clip == 8.0
    bits = np.random.RandomState(0).randint(0, 2, (5, 320, 4))
    t = r.transmit(bits)
    pw = np.abs(t.reshape(-1, 80)) ** 2
    assert t.shape == (5, 7, 80) and np.all(pw.max(1) <= 8.0 * pw.mean(1) * 1.3)       # clipped (mean taken before clipping)
    assert np.allclose(t[:, :, :16], t[:, :, 64:80])                                    # cyclic prefix


def test_pdp_correlations_match_their_defining_integrals():
    """dev/m/mmse_pdp.m closed forms vs numerical quadrature of R[m,n] = int p(t) exp(-2 pi i (m-n) t / N) dt."""
    N, L, Trms = 64, 7, 1.3
    t = np.linspace(0.0, L, 200001)
    d = np.arange(-5, 6)
    for uniform, p in ((True, np.ones_like(t) / L), (False, np.exp(-t / Trms) / (Trms * (1 - np.exp(-L / Trms))))):
        R = B.mmse_pdp(L, N, Trms, uniform)
        assert np.allclose(R, R.conj().T) and np.allclose(np.diag(R), 1.0)
        num = np.array([np.trapezoid(p * np.exp(-2j * np.pi * k * t / N), t) for k in d])
        assert np.abs(num - np.array([R[(k) % N if k >= 0 else 0, 0 if k >= 0 else -k] for k in d])).max() < 1e-6
    trms, tmean = B.rms_delay_spread([0.0, 100e-9], [0.0, 0.0])
    assert abs(tmean - 50e-9) < 1e-15 and abs(trms - 50e-9) < 1e-15


def test_cp_enhanced_recovers_the_symbols_of_a_causal_channel():
    """dev/m/cpenhanced.m on a noise-free causal 5-tap channel with the exact response: the joint least-squares solve
    returns the transmitted grid (first symbol exactly: no previous symbol; later ones up to the script's one-diagonal-short
    ISI term)."""
    rng = np.random.RandomState(2)
    n, S, N, L = 6, 7, 64, 16
    X = (rng.choice([-1, 1], (n, S, N)) + 1j * rng.choice([-1, 1], (n, S, N))).astype(np.complex128)
    h = (rng.randn(n, 5) + 1j * rng.randn(n, 5)) * np.array([1.0, 0.5, 0.3, 0.2, 0.1])
    x = np.fft.ifft(X, axis=-1)
    xcp = np.concatenate([x[..., N - L:], x], axis=-1).reshape(n, S * (N + L))
    y = np.stack([np.convolve(xcp[i], h[i])[:S * (N + L)] for i in range(n)]).reshape(n, S, N + L)
    Y = np.fft.fft(y[..., L:], axis=-1)
    G = np.repeat(np.fft.fft(h, N, axis=-1)[:, None, :], S, axis=1)
    assert np.abs(Y / G - X).max() < 1e-9                                  # sanity: the prefix covers the channel
    Xe = B.cp_enhanced(Y, G, y, N, L)
    assert np.abs(Xe[:, 0] - X[:, 0]).max() < 1e-8
    assert np.abs(Xe - X).max() < 0.2 and np.array_equal(np.sign(Xe.real), np.sign(X.real))


def test_aligned_window_removes_the_precursor_floor_and_all_estimators_run():
    """radio.py's centred channel filter leaks (L-1)/2 pre-cursor taps into the next symbol's prefix: with the FFT
    window where radio.py puts it even perfect CSI floors; aligned to the causal response (MATLAB's situation) it does not."""
    F = Flags(nbits=4, channel="EVA")
    raw = B.ber_curve(F, "Perfect", [40.0], n_frames=150, seed=4, aligned=False)[0]
    ali = B.ber_curve(F, "Perfect", [40.0], n_frames=150, seed=4, aligned=True)[0]
    assert raw > 5e-3 and ali < 0.2 * raw, (raw, ali)
    F = Flags(nbits=2, channel="ETU")
    ber = {m: B.ber_curve(F, m, [25.0], n_frames=200, seed=6)[0] for m in B.EST_NAMES}
    assert all(0.0 <= v < 0.25 for v in ber.values()), ber
    assert ber["Perfect"] <= min(ber["LS-Spline"], ber["LS-Linear"], ber["ALMMSE"], ber["LS-CP"]) + 1e-4, ber
    assert ber["LMMSE"] <= ber["LS-Spline"] + 1e-4 and ber["LMMSE-UniPDP"] <= 2.0 * ber["LS-Spline"] + 1e-3, ber
    assert ber["ALMMSE-CP"] <= 1.2 * ber["ALMMSE"] + 1e-3 and ber["LS-CP"] <= 1.2 * ber["LS-Spline"] + 1e-3, ber
    with pytest.raises(ValueError):
        B.ber_curve(F, "LS-CP", [25.0], n_frames=10, aligned=False)
    g = B.ber_curve(Flags(nbits=4, channel="EPA"), "LS-Spline", [20.0], n_frames=150, seed=6, mapping="gray")[0]
    t = B.ber_curve(Flags(nbits=4, channel="EPA"), "LS-Spline", [20.0], n_frames=150, seed=6, mapping="table")[0]
    assert 0.0 < g < 0.2 and 0.0 < t < 0.2


def test_run_benchmark_writes_matlab_style_tables(tmp_path):
    F = Flags(nbits=1, channel="Flat")
    paths = B.run_benchmark(F, methods=("Perfect", "ALMMSE"), snrs=(0, 10, 20), n_frames=60, out_dir=str(tmp_path))
    assert os.path.basename(paths["ALMMSE"]) == "BER_OFDM_Flat_ALMMSE_lte_64_Table.csv"
    t = np.loadtxt(paths["Perfect"], delimiter=",")
    assert t.shape == (5, 3) and list(t[0]) == [0, 10, 20]
    assert np.all(t[1:, 2] <= t[1:, 0] + 1e-9)                         # BER falls with SNR for every modulation
    assert np.all(t[1, :] <= t[4, :] + 1e-9)                           # BPSK beats 16-QAM
