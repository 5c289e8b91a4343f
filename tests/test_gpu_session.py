"""Graph-name API: ``load_model_np`` + ``Session.run`` serve every tensor name the reference's
loaders fetch (dev/py/model.py:58-71, dev/py/ofdmreceiver_np_mp.py:273-285), with the oracle's values (``-m gpu``)."""
import numpy as np
import pytest

from oracle import dccn_oracle as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("nbits,batch,cp", [(1, 36, True), (2, 130, True), (2, 64, False), (4, 48, True)])
def test_every_graph_tensor_name_resolves_and_matches_the_oracle(tmp_path, nbits, batch, cp):
    from dl_ofdm_amd import receiver
    from dl_ofdm_amd.engine import RxDims, RxEngine
    from dl_ofdm_amd.session import PLACEHOLDERS, TENSOR_NAMES, Session, load_model_np
    kin = 80 if cp else 64
    cfg = O.RxConfig(S=7, kin=kin, F=64, D=320, nbits=nbits)
    p = O.init_params(cfg, seed=5)
    rng = np.random.RandomState(nbits)
    p["demodulation/dense/bias"] = (rng.randn(640) * 0.1).astype(np.float32)
    eng = RxEngine(RxDims(7, kin, 64, 320, nbits), 8, params=p, train=True)
    F = receiver.Flags(nbits=nbits, cp=cp, nfilter=64)
    path = receiver.save_checkpoint(str(tmp_path / "OFDM_x"), eng, F)
    sess = Session(seed=3)
    tup = load_model_np(path, sess)
    assert [t.name for t in tup] == ["bits_in:0", "tx_ofdm:0", "input:0", "output:0", "cost:0", "log_ber:0", "linear_ber:0",
                                     "conf_matrix:0", "tx_power:0", "noise_power:0", "iq_rx:0", "iq_tx:0", "ce_mean:0", "SNR:0"]
    with pytest.raises(KeyError):
        sess.get_tensor_by_name("Norm:0")
    x = (rng.randn(batch, 7, kin, 2) * rng.uniform(0.3, 3.0, (7, kin, 2))).astype(np.float32)
    bits = rng.randint(0, 2, (batch, 320, nbits)).astype(np.int32)
    snr_db = 7.0
    snr = np.full((batch, 1), snr_db, np.float32)
    names = [n for n in TENSOR_NAMES]
    g = sess.get_tensor_by_name
    vals = dict(zip(names, sess.run([g(n) for n in names], {g("tx_ofdm:0"): x, g("bits_in:0"): bits, g("SNR:0"): snr})))
    assert set(vals) == set(TENSOR_NAMES) and all(v is not None for v in vals.values())
    ref = O.rx_eval({k: v.astype(np.float64) for k, v in p.items()}, x.astype(np.float64), bits, cfg)
    xn, _, _ = O.batch_moment_norm(x.astype(np.float64).reshape(batch, -1))
    xn = xn.reshape(x.shape)
    assert np.array_equal(vals["bits_in:0"], bits) and np.array_equal(vals["tx_ofdm:0"], x)
    assert _rel(vals["input:0"], xn) <= 1e-5
    assert _rel(vals["output:0"], ref["prob"]) <= 1e-5 and vals["output:0"].shape == (batch, 320, nbits, 2)
    fft = O.cconv_gemm_fwd(xn.reshape(batch * 7, kin, 2), p["fft_like/conv3d/kernel"].astype(np.float64),
                           p["fft_like/conv3d/bias"].astype(np.float64)).reshape(batch, 7, 64, 2)
    assert _rel(vals["receiver/fft_like/fft_out:0"], fft) <= 1e-5
    assert abs(float(vals["ce_mean:0"]) - float(ref["ce_mean"])) <= 1e-5 * float(ref["ce_mean"])
    # decisions bit-exact unless a probability pair is within rounding of a tie
    pr = ref["prob"].reshape(-1, 2)
    if np.abs(pr[:, 1] - pr[:, 0]).min() > 1e-5:
        assert np.array_equal(vals["conf_matrix:0"], ref["conf"])
    assert int(vals["conf_matrix:0"].sum()) == batch * 320 * nbits
    c = vals["conf_matrix:0"].astype(np.float64)
    ber = (c[0, 1] + c[1, 0]) / c.sum()
    assert abs(float(vals["linear_ber:0"]) - ber) <= 1e-7 and abs(float(vals["log_ber:0"]) - np.log(ber)) <= 1e-6
    assert abs(float(vals["cost:0"]) - float(ref["cost"])) <= 2e-5 * abs(float(ref["cost"])) + 1e-5
    clipped, power = O.complex_clip(xn, 8.0)
    assert abs(float(vals["tx_power:0"]) - float(power)) <= 1e-5 * float(power)
    assert _rel(vals["tx_signal:0"], clipped) <= 1e-5
    # iq_tx = fp16([-1,2]) of tx_signal, round-to-nearest like tf.cast
    assert vals["iq_tx:0"].dtype == np.float16 and vals["iq_tx:0"].shape == (batch * 7 * kin, 2)
    assert np.array_equal(vals["iq_tx:0"], vals["tx_signal:0"].reshape(-1, 2).astype(np.float16))
    # iq_rx = fp16(batchnorm(tx_signal, 1e-8)/sqrt(2) + noise), noise power 0.5 * 10^(-SNR/10) (radio.py:62-88)
    x2, _, _ = O.batch_moment_norm(clipped.reshape(batch, -1), eps=1e-8)
    noise = vals["iq_rx:0"].astype(np.float64) - x2.reshape(-1, 2)
    level2 = 0.5 * 10.0 ** (-snr_db / 10.0)
    n = noise.shape[0]
    assert abs((noise ** 2).sum(1).mean() - level2) <= 6.0 * level2 * np.sqrt(2.0 / n) + 2e-3 * level2 + 1e-3
    assert abs(noise.mean()) <= 6.0 * np.sqrt(level2 / 2 / n) + 1e-3
    assert abs(float(vals["noise_power:0"]) - (noise ** 2).sum(1).mean()) <= 0.02 * level2 + 1e-3
    assert PLACEHOLDERS == ("bits_in:0", "tx_ofdm:0", "SNR:0")
    sess.close()
