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
        ref = host.receive(xh, method, snr, H_true=Hs, R_long=R_long, advance=adv)
        err, cnt, det = rx.receive(x, bits, method, snr, H_true=H, want_bits=True)
        det = det.cpu().numpy()
        assert cnt == total and det.shape == ref.shape
        assert err == int(np.count_nonzero(det != bh))                    # the device's own count is exact
        differ = float(np.mean(det != ref))
        assert differ <= 3e-4, (method, differ)
        assert abs(err - int(np.count_nonzero(ref != bh))) <= 5e-4 * total + 2, method
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
        rx.receive(x, bits, "LS-CP", 10.0)
