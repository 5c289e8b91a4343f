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
    assert B.EST_NAMES[0] == "Perfect" and len(B.EST_NAMES) == 6


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


def test_run_benchmark_writes_matlab_style_tables(tmp_path):
    F = Flags(nbits=1, channel="Flat")
    paths = B.run_benchmark(F, methods=("Perfect", "ALMMSE"), snrs=(0, 10, 20), n_frames=60, out_dir=str(tmp_path))
    assert os.path.basename(paths["ALMMSE"]) == "BER_OFDM_Flat_ALMMSE_lte_64_Table.csv"
    t = np.loadtxt(paths["Perfect"], delimiter=",")
    assert t.shape == (5, 3) and list(t[0]) == [0, 10, 20]
    assert np.all(t[1:, 2] <= t[1:, 0] + 1e-9)                         # BER falls with SNR for every modulation
    assert np.all(t[1, :] <= t[4, :] + 1e-9)                           # BPSK beats 16-QAM
