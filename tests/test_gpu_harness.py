"""End-to-end behaviour on the GPU (-m gpu): closed-form AWGN BER through the whole chain
(NumPy transmitter -> AWGN -> fused GPU receiver), short training run, checkpoint round trip, sweep CSV."""
import math
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def flags(**kw):
    from dl_ofdm_amd.receiver import Flags
    f = Flags(channel="AWGN", nfilter=64, nbits=1, SNR=5.0)
    for k, v in kw.items():
        setattr(f, k, v)
    return f


def qfunc(x):
    return 0.5 * math.erfc(x / math.sqrt(2.0))


def dft_receiver_params(dims, ofdmobj, nbits, gain=4.0):
    """Analytic receiver: filter k = (cos_k, -sin_k) over the 64 post-CP
    samples gives re_k = Re X_k and im_k = -Im X_{N-k}; the dense layer picks Re/Im of every data cell; the
    tail slices signs (BPSK: bit = Re>0; QPSK per dev/py/ofdm.py:45-50: b1 = Re>0, b0 = Im<0)."""
    N, CP, F, D = ofdmobj.K, ofdmobj.CP, dims.F, dims.D
    assert F == N and dims.kin == N + CP
    n = np.arange(N)[:, None]
    k = np.arange(N)[None, :]
    w = np.zeros((dims.kin, 2 * F), np.float32)
    w[CP:, :F] = np.cos(2 * np.pi * n * k / N)
    w[CP:, F:] = -np.sin(2 * np.pi * n * k / N)
    dense = np.zeros((dims.S * F * 2, 2 * D), np.float32)
    for d, sc in enumerate(ofdmobj.dataSc):
        s, c = divmod(int(sc), N)
        dense[(s * F + c) * 2 + 0, 2 * d] = 1.0                      # Re X_c   = re of filter c
        dense[(s * F + (N - c) % N) * 2 + 1, 2 * d + 1] = -1.0       # Im X_c   = -im of filter N-c
    m = 2 ** nbits
    w1 = np.zeros((2, m), np.float32)
    w2 = np.zeros((m + 2, 2 * nbits), np.float32)
    if nbits == 1:
        w2[m + 0] = [-gain, gain]                                    # bit = Re > 0
    else:
        w2[m + 1, 0:2] = [gain, -gain]                               # b0 = Im < 0
        w2[m + 0, 2:4] = [-gain, gain]                               # b1 = Re > 0
    z = lambda *s: np.zeros(s, np.float32)
    return {"fft_like/conv3d/kernel": w, "fft_like/conv3d/bias": z(2 * F),
            "demodulation/dense/kernel": dense, "demodulation/dense/bias": z(2 * D),
            "demodulation/conv2d/kernel": w1, "demodulation/conv2d/bias": z(m),
            "demodulation/dense_1/kernel": w2, "demodulation/dense_1/bias": z(2 * nbits)}


@pytest.mark.parametrize("nbits,snr_db", [(1, -2.0), (1, 1.0), (2, 1.0), (2, 4.0)])
def test_awgn_ber_matches_closed_form(nbits, snr_db):
    from dl_ofdm_amd import ofdm, radio, receiver
    from dl_ofdm_amd.engine import RxEngine
    F = flags(nbits=nbits)
    o = ofdm.ofdm_tx(F)
    dims = receiver.rx_dims(F, o)
    frames = 6000
    np.random.seed(1234 + nbits)
    fading = radio.rayleigh_chan_lte(F, o.Fs)
    xs, ys, _ = receiver.make_batch(F, o, fading, frames, snr_db)
    eng = RxEngine(dims, frames, train=False, params=dft_receiver_params(dims, o, nbits), want_prob=False)
    eng.eval_step(xs, ys)
    m = eng.metrics()
    sigma2 = 10.0 ** (-snr_db / 10.0)
    # unit mean IQ power over 48 occupied carriers of energy 18 -> per-dimension SNR after the 64-point DFT
    per_dim = (64.0 / 24.0 if nbits == 1 else 64.0 / 48.0) / sigma2
    theory = qfunc(math.sqrt(per_dim))
    n_bits = frames * o.frame_size * nbits
    assert m["count"] == n_bits
    tol = 4.0 * math.sqrt(theory * (1 - theory) / n_bits) + 0.02 * theory     # 4 sigma + 2 % model error
    assert abs(m["berlin"] - theory) <= tol, (m["berlin"], theory)


def test_short_training_run_learns_and_sweeps(tmp_path):
    from dl_ofdm_amd import receiver
    F = flags(nbits=1, SNR=5.0, msg_length=7 * 2048, batch_size=7 * 64, max_epoch_num=12, early_stop=50,
              save_dir=str(tmp_path / "ckpt"), token="OFDM_t", test_frames=1500, eval_frames=512, snr_lo=0, snr_hi=6,
              iq_dump=True)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        res = receiver.train(F, verbose=False)
    finally:
        os.chdir(cwd)
    h = res["history"]
    assert h[-1]["train_loss"] < h[0]["train_loss"] - 0.02, (h[0], h[-1])
    assert h[-1]["test_ber"] < 0.35 and h[0]["test_ber"] > h[-1]["test_ber"]
    snrs, ber, loss, csvfile = res["sweep"]
    csvfile = os.path.join(str(tmp_path), os.path.basename(csvfile))
    assert os.path.basename(csvfile) == "Test_DCCN_OFDM_t_AWGN.csv" and os.path.isfile(csvfile)
    lines = open(csvfile).read().splitlines()
    assert lines[0] == "SNR,BER,Loss" and len(lines) == 8
    assert ber[-1] < ber[0]                                            # BER falls with SNR
    # checkpoint holds the reference's variable names incl. Adam slots
    z = np.load(res["best_path"] + ".npz")
    for n in ("fft_like/conv3d/kernel", "demodulation/dense/kernel/Adam", "demodulation/dense_1/bias/Adam_1",
              "global_step", "beta1_power", "beta2_power"):
        assert n in z
    assert float(z["global_step"]) > 0
    # constellation dumps of the graph's monitor branch (ofdmreceiver_np.py:264-265): 2048 fp16 IQ pairs each
    for suffix in ("txiq", "rxiq"):
        d = np.loadtxt(os.path.join(str(tmp_path / "ckpt"), "OFDM_t_%s.csv" % suffix), delimiter=",")     # save_dir, on request
        assert d.shape == (2048, 2) and np.isfinite(d).all()
        assert np.array_equal(d.astype(np.float16).astype(np.float64), d)            # values are fp16-representable


def test_checkpoint_round_trip(tmp_path):
    import torch
    from dl_ofdm_amd import receiver
    from dl_ofdm_amd.engine import RxDims, RxEngine
    dims = RxDims(7, 80, 64, 320, 2)
    rng = np.random.RandomState(0)
    x, bits = rng.randn(64, 7, 80, 2).astype(np.float32), rng.randint(0, 2, (64, 320, 2)).astype(np.int32)
    a = RxEngine(dims, 64, train=True, seed=3)
    for _ in range(3):
        a.train_step(x, bits)
    path = receiver.save_checkpoint(str(tmp_path / "m"), a, receiver.Flags())
    b = RxEngine(dims, 64, train=True, seed=9)
    receiver.load_checkpoint(path, b)
    a.train_step(x, bits)
    b.train_step(x, bits)
    torch.cuda.synchronize()
    assert torch.equal(a.params, b.params) and torch.equal(a.adam_v, b.adam_v)
    assert a.adam() == b.adam()


def test_tf_bundle_checkpoint_roundtrip_through_the_harness(tmp_path):
    """--tf_checkpoint writes the tf.train.Saver files next to the .npz; a directory holding ONLY the TF bundle
    (as a reference-trained model would) loads into the engine with identical parameters and Adam state."""
    from dl_ofdm_amd import ofdm, receiver, tf_bundle
    from dl_ofdm_amd.engine import PARAM_NAMES, RxEngine
    F = flags(nbits=2, tf_checkpoint=True, token="TFCK", save_dir=str(tmp_path) + "/")
    o = ofdm.ofdm_tx(F)
    dims = receiver.rx_dims(F, o)
    eng = RxEngine(dims, 16, train=True, seed=4)
    rng = np.random.RandomState(2)
    for _ in range(3):
        eng.train_step(rng.standard_normal((16, 7, 80, 2)).astype(np.float32), rng.randint(0, 2, (16, o.frame_size, 2)))
    stem = os.path.join(F.save_dir, F.token)
    receiver.save_checkpoint(stem, eng, F)
    ent = tf_bundle.read_index(stem + ".index")
    assert len(ent) == 3 * 8 + 3 and ent["fft_like/conv3d/kernel"]["shape"] == [1, 80, 1, 80, 128]
    assert ent["demodulation/conv2d/kernel/Adam_1"]["shape"] == [1, 1, 2, 4]
    os.remove(stem + ".npz")                                            # only the TensorFlow files remain
    eng2 = RxEngine(dims, 16, train=True, seed=99)
    receiver.load_checkpoint(stem, eng2)
    import torch
    assert torch.equal(eng.params, eng2.params) and torch.equal(eng.adam_m, eng2.adam_m)
    assert torch.equal(eng.adam_v, eng2.adam_v) and eng2.adam()["global_step"] == 3.0
    z = receiver.read_checkpoint_file(stem)
    assert all(np.array_equal(z[n], eng.get_params()[n]) for n in PARAM_NAMES)


def test_sweep_driver_both_stages_on_device_data(tmp_path, monkeypatch):
    """dev/py/run_local_ofdm.py end to end, scaled down: 16 AWGN receivers, then the --opt=0 equaliser on
    mixRayleigh for cp in (True, False) and both CP lengths, every batch generated on the GPU; the result
    directories hold the reference's CSV names."""
    from dl_ofdm_amd import run_local_ofdm
    monkeypatch.chdir(tmp_path)
    run_local_ofdm.main(["--awgn=True", "--equalizer=True", "--device_data=True", "--max_epoch_scale=0.0005",
                         "--msg_length=7168", "--test_frames=500"])
    for cpdir in ("short", "long"):
        d = tmp_path / ("test_ext_64_%s_cross_mobile" % cpdir)
        names = sorted(os.listdir(d))
        assert len([n for n in names if n.endswith("_AWGN.csv")]) == 8
        for cp in ("True", "False"):
            for ch in ("ETU", "EVA", "EPA", "Flat", "Custom"):
                assert "Test_DCCN_OFDM_Dense3_1mod_snr5_cp%s_Equalizer0_mixRayleigh_test_chan_%s.csv" % (cp, ch) in names
        rows = open(d / "Test_DCCN_OFDM_Dense3_1mod_snr5_cpTrue_Equalizer0_mixRayleigh_test_chan_EPA.csv").read().splitlines()
        assert rows[0] == "SNR,BER,Loss" and len(rows) == 10
    # second invocation: everything is already there, nothing is recomputed
    import time
    t0 = time.time()
    run_local_ofdm.main(["--awgn=True", "--equalizer=True", "--device_data=True", "--max_epoch_scale=0.0005"])
    assert time.time() - t0 < 2.0
