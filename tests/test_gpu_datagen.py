"""Device-side input generator against the host substrate (dl_ofdm_amd/ofdm.py, radio.py --
themselves bit-pinned to the reference by tests/test_golden_substrate.py) and oracle/datagen_oracle.py.

Deterministic stages: fed the same bits / tap draws / noise draws, fp32 kernels vs the fp64 host chain agree to
1e-5 of the signal scale.  Random streams: Philox words bit-exact vs the oracle (which carries the Random123
known-answer vectors); label bits bit-exact; normals to 2e-6 (libm differences in log/sincos); moments within
5 sigma.  End to end: closed-form AWGN BER through the analytic DFT receiver."""
import math

import numpy as np
import pytest
import torch

from oracle import datagen_oracle as G

pytestmark = pytest.mark.gpu


def flags(**kw):
    from dl_ofdm_amd.receiver import Flags
    f = Flags(channel="EPA", nfilter=64, nbits=2, SNR=5.0)
    for k, v in kw.items():
        setattr(f, k, v)
    return f


def test_philox_words_bit_exact():
    from dl_ofdm_amd import _lib
    import ctypes as C
    lib = _lib.load()
    n = 5000
    for stream, off, seed in ((0, 0, 1), (2, 7, 0xDEADBEEFCAFE1234), (1, 0xFFFFFFFF, 42)):
        out = torch.zeros(n, 4, dtype=torch.int32, device="cuda")
        _lib.check(lib.dccn_philox_fill(out.data_ptr(), n, stream, off, seed, None), "philox")
        want = G.philox4x32_10(G.counters(np.arange(n), stream, off), G.key_of(seed))
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want)
    for c, k, o in G.KAT:                                       # the oracle itself is pinned by the published vectors
        assert tuple(int(v) for v in G.philox4x32_10(np.array(c, dtype=np.uint32), k)) == o


@pytest.mark.parametrize("nbits", [1, 2, 3, 4])
def test_transmitter_matches_host(nbits):
    from dl_ofdm_amd import ofdm
    from dl_ofdm_amd.datagen import DeviceDataGen
    F = flags(nbits=nbits)
    o = ofdm.ofdm_tx(F)
    gen = DeviceDataGen(F, o, seed=3)
    rng = np.random.RandomState(nbits)
    bits = rng.randint(0, 2, (37, o.frame_size, nbits))
    _, want, _ = o.ofdm_tx_frame_np(bits)
    tx, b = gen.transmit(37, bits=bits)
    assert np.abs(tx.cpu().numpy() - want).max() <= 1e-5 * np.abs(want).max()
    # drawn labels: bit-exact vs the oracle stream, and the frames are the modulation of exactly those labels
    tx2, b2 = gen.transmit(11, offset=5)
    want_bits = G.bits(3, 5, 11, o.frame_size, nbits)
    assert np.array_equal(b2.cpu().numpy(), want_bits)
    _, want2, _ = o.ofdm_tx_frame_np(want_bits)
    assert np.abs(tx2.cpu().numpy() - want2).max() <= 1e-5 * np.abs(want2).max()


@pytest.mark.parametrize("chan", ["EPA", "EVA", "ETU", "Flat", "Custom", "AWGN"])
def test_channel_and_awgn_match_host_given_the_same_draws(chan):
    from dl_ofdm_amd import ofdm, radio, util
    from dl_ofdm_amd.datagen import DeviceDataGen
    F = flags(channel=chan)
    o = ofdm.ofdm_tx(F)
    gen = DeviceDataGen(F, o, seed=9)
    n = 23
    np.random.seed(77)
    bits = util.bit_source(2, o.frame_size, n)
    iq, _, _ = o.ofdm_tx_frame_np(bits)
    fading = radio.rayleigh_chan_lte(F, o.Fs)
    snr = np.linspace(-3, 25, n).reshape(n, 1)
    np.random.seed(123)
    y_host, H_host = fading.run(iq)
    out_host, npow_host = radio.AWGN_channel_np(y_host, snr)
    # replay numpy's draws: per frame normal(size=[n_taps,2]) (radio.py static taps), then randn(*shape)
    np.random.seed(123)
    taps = None
    if chan != "AWGN":
        taps = np.stack([np.random.normal(loc=0.0, scale=1.0, size=[gen.n_taps, 2]) for _ in range(n)])
        # the host draws with scale 1/sqrt(2); standard normals * 1/sqrt(2) are the same stream scaled
    noise = np.random.randn(n, 7, 80, 2)
    tx, _ = gen.transmit(n, bits=bits)
    out, npow, H = gen.channel(tx, snr, taps=taps, noise=noise.reshape(n, -1, 2), want_H=True)
    scale = np.abs(out_host).max()
    assert np.abs(out.cpu().numpy() - out_host).max() <= 2e-5 * scale
    assert abs(float(npow) - npow_host) <= 1e-5 * npow_host
    Hh = H_host[:, 0, :]
    assert np.abs(H.cpu().numpy() - Hh).max() <= 2e-5 * max(np.abs(Hh).max(), 1.0)


@pytest.mark.parametrize("chan,mix", [("EPA", False), ("ETU", False), ("AWGN", False), ("mixRayleigh", False)])
def test_taps_drawn_inside_the_fir_launch_are_the_same_taps(chan, mix):
    """want_H=False: no taps launch, the FIR blocks draw the static taps of their frames themselves (datagen.h TapGen) --
    same Philox draws, same coefficients, same order of the sums: the received batch has the same bits as with the taps
    kernel in front (want_H=True), for the plain static channels and the frame-interleaved mix channels"""
    from dl_ofdm_amd import ofdm
    from dl_ofdm_amd.datagen import DeviceDataGen
    F = flags(channel=chan)
    o = ofdm.ofdm_tx(F)
    n = 301
    snr = np.linspace(0, 20, n)
    outs = []
    for want_H in (True, False):
        gen = DeviceDataGen(F, o, seed=13, mix=mix)
        tx, _ = gen.transmit(n)
        res = gen.channel(tx, snr, want_H=want_H, offset=5)
        outs.append((res[0].clone(), float(res[1])))
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]


def test_random_streams_statistics():
    from dl_ofdm_amd import ofdm
    from dl_ofdm_amd.datagen import DeviceDataGen
    F = flags(channel="ETU")
    o = ofdm.ofdm_tx(F)
    gen = DeviceDataGen(F, o, seed=11)
    n = 4000
    x, bits, npow, H = gen.make_batch(n, 10.0, want_H=True)
    b = bits.cpu().numpy()
    N = b.size
    assert abs(b.mean() - 0.5) <= 5 * 0.5 / math.sqrt(N)
    assert abs(float(npow) - 0.1) <= 5 * 0.1 / math.sqrt(n * 560)            # E|n|^2 = 10^(-SNR/10)
    xp = x.cpu().numpy()
    sig_plus_noise = (xp ** 2).sum(-1).mean()
    assert abs(sig_plus_noise - 1.1) <= 0.02                                  # unit signal power + noise
    Hn = H.cpu().numpy()
    coeff, alpha = gen.coeff.cpu().numpy().astype(np.float64), gen.alpha.cpu().numpy().astype(np.float64)
    expect = float(((coeff[:, None] * alpha) ** 2).sum())          # Parseval: mean_f |H|^2 = sum_l E|g_l|^2
    assert abs((np.abs(Hn) ** 2).mean() - expect) <= 0.05 * expect
    # the drawn normals are the oracle's stream: isolate them as (noisy - noiseless) on an AWGN pass-through
    gen2 = DeviceDataGen(flags(channel="AWGN"), o, seed=11)
    tx1, _ = gen2.transmit(1, offset=0)
    o1 = gen2.channel(tx1, 0.0, offset=0)[0].clone()
    o2 = gen2.channel(tx1, 0.0, offset=0, noise=np.zeros((1, 560, 2), np.float32))[0].clone()
    drawn = (o1 - o2).cpu().numpy().reshape(-1, 2)[:64] / math.sqrt(0.5)
    assert np.abs(drawn - G.noise_normals(11, 0, 64)).max() <= 2e-5
    # consecutive batches differ; the same (seed, offset) repeats exactly
    a = gen2.make_batch(3, 5.0)[0].clone()
    b2 = gen2.make_batch(3, 5.0)[0].clone()
    assert not torch.equal(a, b2)
    gen3 = DeviceDataGen(flags(channel="AWGN"), o, seed=11)
    assert torch.equal(gen3.make_batch(3, 5.0)[0], a)


@pytest.mark.parametrize("nbits,snr_db", [(1, 1.0), (2, 4.0)])
def test_awgn_ber_closed_form_with_device_generated_data(nbits, snr_db):
    from test_gpu_harness import dft_receiver_params, qfunc
    from dl_ofdm_amd import ofdm, receiver
    from dl_ofdm_amd.datagen import DeviceDataGen
    from dl_ofdm_amd.engine import RxEngine
    F = flags(nbits=nbits, channel="AWGN")
    o = ofdm.ofdm_tx(F)
    dims = receiver.rx_dims(F, o)
    frames = 6000
    eng = RxEngine(dims, frames, train=False, params=dft_receiver_params(dims, o, nbits), want_prob=False)
    gen = DeviceDataGen(F, o, seed=2024 + nbits)
    gen.make_batch(frames, snr_db, out_x=eng.x, out_bits=eng.bits)             # straight into the engine's buffers
    eng.eval_step()
    m = eng.metrics()
    sigma2 = 10.0 ** (-snr_db / 10.0)
    per_dim = (64.0 / 24.0 if nbits == 1 else 64.0 / 48.0) / sigma2
    theory = qfunc(math.sqrt(per_dim))
    n_bits = frames * o.frame_size * nbits
    tol = 4.0 * math.sqrt(theory * (1 - theory) / n_bits) + 0.02 * theory
    assert abs(m["berlin"] - theory) <= tol, (m["berlin"], theory)


@pytest.mark.parametrize("chan,mobile,mix", [("mixRayleigh", False, False), ("mixRayleigh", True, True),
                                             ("mixAll", True, True), ("mixAll", False, False)])
def test_mix_channels_match_host_given_the_same_draws(chan, mobile, mix):
    """frame-interleaved profiles (radio.py:438-470): profile = frame % 4 (or 5), Doppler every 3rd (4th) frame"""
    from dl_ofdm_amd import ofdm, radio, util
    from dl_ofdm_amd.datagen import DeviceDataGen
    F = flags(channel=chan)
    o = ofdm.ofdm_tx(F)
    gen = DeviceDataGen(F, o, seed=13, mobile=mobile, mix=mix)
    n = 27
    np.random.seed(8)
    bits = util.bit_source(2, o.frame_size, n)
    iq, _, _ = o.ofdm_tx_frame_np(bits)
    fading = radio.rayleigh_chan_lte(F, o.Fs, mobile=mobile, mix=mix)
    snr = np.linspace(2, 22, n).reshape(n, 1)
    np.random.seed(99)
    y_host, H_host = fading.run(iq)
    out_host, npow_host = radio.AWGN_channel_np(y_host, snr)
    np.random.seed(99)                                              # replay numpy's draws frame by frame
    tn = np.zeros((n, 16, 2), np.float32)
    th = np.zeros((n, 2, 48, 16), np.float32)
    plan = gen.frame_plan(n)
    assert any(d for _, d in plan) == (mobile and mix)
    for fr, (pi, dop) in enumerate(plan):
        pr = gen.profiles[pi]
        if pr["identity"]:
            continue
        if dop:
            for c in range(2):
                th[fr, c, :, :pr["n_taps"]] = np.random.uniform(0, 2 * np.pi, size=(48, pr["n_taps"]))
        else:
            tn[fr, :pr["n_taps"]] = np.random.normal(loc=0.0, scale=1.0, size=[pr["n_taps"], 2])
    noise = np.random.randn(n, 7, 80, 2)
    tx, _ = gen.transmit(n, bits=bits)
    out, npow, H = gen.channel(tx, snr, taps=(tn, th), noise=noise.reshape(n, -1, 2), want_H=True)
    assert np.abs(out.cpu().numpy() - out_host).max() <= 5e-5 * np.abs(out_host).max()
    assert abs(float(npow) - npow_host) <= 1e-5 * npow_host
    assert H.shape == (n, 7, 64) and np.abs(H.cpu().numpy() - H_host).max() <= 5e-5 * max(np.abs(H_host).max(), 1.0)
    # drawn path: finite, unit output power + noise, different profiles give different responses
    x, b, npw, Hd = gen.make_batch(400, 15.0, want_H=True)
    assert torch.isfinite(x).all() and abs(float((x ** 2).sum(-1).mean()) - (1 + 10 ** -1.5)) < 0.05


@pytest.mark.parametrize("chan", ["ETU", "EVA", "EPA", "Flat", "Custom"])
def test_doppler_channel_matches_host_given_the_same_phases(chan):
    """mobile=True (radio.py:376-407): Jakes taps per symbol + per-symbol FIR with history, fed numpy's own draws"""
    from dl_ofdm_amd import ofdm, radio, util
    from dl_ofdm_amd.datagen import DeviceDataGen
    F = flags(channel=chan)
    o = ofdm.ofdm_tx(F)
    gen = DeviceDataGen(F, o, seed=9, mobile=True)
    assert gen.doppler
    n = 9
    np.random.seed(5)
    bits = util.bit_source(2, o.frame_size, n)
    iq, _, _ = o.ofdm_tx_frame_np(bits)
    fading = radio.rayleigh_chan_lte(F, o.Fs, mobile=True)
    snr = np.linspace(0, 20, n).reshape(n, 1)
    np.random.seed(321)
    y_host, H_host = fading.run(iq)
    out_host, npow_host = radio.AWGN_channel_np(y_host, snr)
    np.random.seed(321)                                             # replay: per frame th_re then th_im
    th = np.stack([np.stack([np.random.uniform(0, 2 * np.pi, size=(48, gen.n_taps)) for _ in range(2)])
                   for _ in range(n)])
    noise = np.random.randn(n, 7, 80, 2)
    tx, _ = gen.transmit(n, bits=bits)
    out, npow, H = gen.channel(tx, snr, taps=th, noise=noise.reshape(n, -1, 2), want_H=True)
    assert np.abs(out.cpu().numpy() - out_host).max() <= 5e-5 * np.abs(out_host).max()
    assert abs(float(npow) - npow_host) <= 1e-5 * npow_host
    assert H.shape == (n, 7, 64) and np.abs(H.cpu().numpy() - H_host).max() <= 5e-5 * max(np.abs(H_host).max(), 1.0)
    # drawn phases are the oracle's stream: the channel built from them equals the channel given them explicitly
    th_dev = G.doppler_thetas(9, 4, n, gen.n_taps)
    a = gen.channel(tx, snr, noise=noise.reshape(n, -1, 2), offset=4)[0].clone()
    b = gen.channel(tx, snr, taps=th_dev, noise=noise.reshape(n, -1, 2), offset=4)[0]
    assert np.abs((a - b).cpu().numpy()).max() <= 5e-5 * float(b.abs().max())


def test_harness_trains_on_device_generated_data(tmp_path):
    """receiver.train(device_data=True): data never touches the host; the receiver still learns AWGN QPSK"""
    from dl_ofdm_amd import receiver as R
    F = R.Flags(nbits=2, nfilter=64, channel="AWGN", SNR=10.0, msg_length=7 * 8192, batch_size=512, max_epoch_num=8,
                early_stop=100, token="D", save_dir=str(tmp_path) + "/", seed=5, device_data=True,
                test_frames=2000, snr_lo=0, snr_hi=10)
    res = R.train(F, verbose=False, run_test=True)
    h = res["history"]
    assert h[-1]["train_loss"] < h[0]["train_loss"] - 0.05, h
    assert h[-1]["test_ber"] < 0.2, h
    snrs, ber, loss, csv = res["sweep"]
    assert ber[-1] < ber[0] and len(snrs) == 11


@pytest.mark.parametrize("nfft,longcp", [(1024, False), (128, True)])
def test_transmitter_and_awgn_at_other_fft_sizes(nfft, longcp):
    """config-4 geometry (N=1024, CP=72) and N=128: grid tables, IDFT+CP GEMM and the AWGN stage vs the host chain"""
    from dl_ofdm_amd import ofdm, radio
    from dl_ofdm_amd.datagen import DeviceDataGen
    F = flags(channel="AWGN", nfft=nfft, longcp=longcp, nfilter=nfft)
    o = ofdm.ofdm_tx(F)
    gen = DeviceDataGen(F, o, seed=5)
    rng = np.random.RandomState(nfft)
    n = 6
    bits = rng.randint(0, 2, (n, o.frame_size, 2))
    iq, want, _ = o.ofdm_tx_frame_np(bits)
    tx, _ = gen.transmit(n, bits=bits)
    assert tx.shape == want.shape and np.abs(tx.cpu().numpy() - want).max() <= 2e-5 * np.abs(want).max()
    np.random.seed(3)
    y_host, H_host = radio.rayleigh_chan_lte(F, o.Fs).run(iq)
    out_host, npow_host = radio.AWGN_channel_np(y_host, 7.0 * np.ones((n, 1)))
    np.random.seed(3)
    noise = np.random.randn(*want.shape)
    out, npow, H = gen.channel(tx, 7.0, noise=noise.reshape(n, -1, 2), want_H=True)
    assert np.abs(out.cpu().numpy() - out_host).max() <= 3e-5 * np.abs(out_host).max()
    assert abs(float(npow) - npow_host) <= 1e-5 * npow_host
    assert np.abs(H.cpu().numpy() - H_host[:, 0, :]).max() <= 1e-5


def test_generator_on_a_side_stream_trains_the_same_model():
    """datagen.SideStreamFeeder + dccn_rx_buffers.x_next_ready: batch i+1 is generated on a second HIP stream while the forward
    and backward launches of step i run; the optimizer launch (which normalises that batch) waits for the generator's event
    and the generator waits for the previous step.  Same seeds -> parameters, optimizer state and metrics are bit-identical to
    the one-stream loop after 40 steps (a missed dependency would train on a half-written batch or on stale labels)."""
    from dl_ofdm_amd import ofdm, receiver as R
    from dl_ofdm_amd.datagen import DeviceDataGen, SideStreamFeeder
    from dl_ofdm_amd.engine import RxEngine
    F = R.Flags(nbits=2, nfilter=64, channel="EPA", SNR=10.0)
    o = ofdm.ofdm_tx(F)
    frames, steps = 1170, 40

    def run(overlap):
        eng = RxEngine(R.rx_dims(F, o), frames, train=True, seed=3, want_prob=False)
        gen = DeviceDataGen(F, o, seed=5)
        gen.want_noise_power = False
        make = lambda slot: gen.make_batch(frames, F.SNR, out_x=eng.x, out_bits=eng.label_slot(slot))      # noqa: E731
        if overlap:
            feed = SideStreamFeeder(eng, make)
            feed.first(0)
            for i in range(steps):
                last = i + 1 == steps
                if not last:
                    feed.next((i + 1) & 1)
                eng.train_step_pipelined(slot=i & 1, last=last, x_ready=None if last else feed.ready)
                feed.step_issued()
        else:
            make(0)
            eng.prime()
            for i in range(steps):
                last = i + 1 == steps
                if not last:
                    make((i + 1) & 1)
                eng.train_step_pipelined(slot=i & 1, last=last)
        torch.cuda.synchronize()
        return eng.params.clone(), eng.adam_m.clone(), eng.adam_state.clone(), eng.metrics()

    a, b = run(False), run(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert a[3] == b[3] and 0.0 < a[3]["berlin"] < 0.5



# ---- round 5: the whole static-channel chain in one launch (csrc/datagen.h gen_static_frames_kernel) ------------------------
@pytest.mark.parametrize("chan,nbits,n,cp", [("EPA", 2, 1170, True), ("EVA", 4, 37, True), ("AWGN", 1, 64, True), ("ETU", 3, 9, True),
                                              ("EPA", 2, 73, False)])
def test_fused_static_generator_matches_the_launch_per_stage_chain(chan, nbits, n, cp):
    """dccn_gen_static_frames + dccn_gen_static_apply against dccn_ofdm_tx_frames + dccn_channel_awgn at the same (seed,
    offset): the label bits are the same bits, the transmitted frames and the received batch agree to rounding (the ifft runs
    on 16x16x4 MFMA tiles instead of the 32x32x2 GEMM: another summation order), the noise power monitor is the mean of
    |noise|^2 -- odd frame counts (a block holding one frame), every modulation, the AWGN channel (no taps), no cyclic prefix
    in the receiver's view (the generator always emits it)."""
    from dl_ofdm_amd import ofdm
    from dl_ofdm_amd.datagen import DeviceDataGen, FusedStaticGen
    F = flags(nbits=nbits, channel=chan, cp=cp)
    o = ofdm.ofdm_tx(F)
    snr = np.linspace(0.0, 20.0, n).astype(np.float32)
    ga, gb = DeviceDataGen(F, o, seed=11), DeviceDataGen(F, o, seed=11)
    ga.offset = gb.offset = 3
    assert FusedStaticGen.supported(gb)
    tx_a, bits_a = ga.transmit(n)
    x_a, npow_a, _ = ga.channel(tx_a, snr)
    fg = FusedStaticGen(gb, n, snr, want_noise_power=True)
    x_b = torch.empty_like(x_a)
    bits_b = torch.empty_like(bits_a)
    tx_b = torch.empty_like(tx_a)
    _, _, npow_b = fg.make_batch(x_b, bits_b, slot=0, tx_out=tx_b)
    torch.cuda.synchronize()
    assert gb.offset == 4
    assert torch.equal(bits_a, bits_b)
    sc = float(tx_a.abs().max())
    assert float((tx_a - tx_b).abs().max()) <= 2e-6 * sc
    # the cyclic prefix is a copy of the symbol's tail, to the bit (the fused launch multiplies the 2K columns behind the prefix
    # only and stores the last 2 CP of them twice; the ifft matrix's prefix columns are bitwise copies, so the GEMM route agrees)
    ncp = gb.CP
    for t in (tx_a, tx_b):
        v = t.view(n, gb.S, gb.K + ncp, 2)
        assert torch.equal(v[:, :, :ncp], v[:, :, gb.K:])
    assert float((x_a - x_b).abs().max()) <= 1e-5 * float(x_a.abs().max())
    assert abs(float(npow_a) - float(npow_b)) <= 1e-6 * float(npow_a)
    want_npow = float((fg.noise.double() ** 2).sum() / (n * gb.T))
    assert abs(float(npow_b) - want_npow) <= 1e-6 * want_npow
    # x = y / sqrt(mean |y|^2) + noise, in float32 with the float64 batch power
    inv = np.float32(1.0) / np.sqrt(np.float32(float((fg.y.double() ** 2).sum()) / (n * gb.T)))
    want_x = fg.y.cpu().numpy() * inv + fg.noise.cpu().numpy()
    assert np.abs(x_b.cpu().numpy() - want_x).max() <= 2e-7 * np.abs(want_x).max()


@pytest.mark.parametrize("chan,nbits,n", [("mixRayleigh", 2, 73), ("mixRayleigh", 4, 146), ("mixAll", 1, 9), ("EPA", 2, 73)])
def test_fused_generator_with_interleaved_profiles_and_frequency_response(chan, nbits, n):
    """The frame-interleaved static profiles of the equaliser's training channel (radio.py:438-452 without Doppler frames: frame f
    runs profile f % n_profiles -- flat / ETU / EVA / EPA, responses of different lengths side by side in one block) and the
    frequency response H the equaliser's monitor reads, from the ONE fused launch: against dccn_ofdm_tx_frames +
    dccn_channel_groups_awgn (or dccn_channel_awgn) at the same (seed, offset) with per-frame SNRs taken from a caller's tensor.
    Same bits for the labels and for H (same tap draws, same sums, same twiddle arguments); frames to rounding."""
    from dl_ofdm_amd import ofdm
    from dl_ofdm_amd.datagen import DeviceDataGen, FusedStaticGen
    F = flags(nbits=nbits, channel=chan)
    o = ofdm.ofdm_tx(F)
    snr = torch.linspace(-3.0, 27.0, n, device="cuda")
    ga, gb = DeviceDataGen(F, o, seed=21), DeviceDataGen(F, o, seed=21)
    ga.offset = gb.offset = 6
    assert FusedStaticGen.supported(gb) and gb.mixed == chan.startswith("mix")
    tx_a, bits_a = ga.transmit(n)
    x_a, npow_a, H_a = ga.channel(tx_a, snr, want_H=True)
    fg = FusedStaticGen(gb, n, 0.0, want_noise_power=True)
    hshape = (n, gb.S, gb.K, 2) if gb.mixed else (n, gb.K, 2)
    x_b, bits_b, tx_b = torch.empty_like(x_a), torch.empty_like(bits_a), torch.empty_like(tx_a)
    H_b = torch.full(hshape, float("nan"), device="cuda")
    _, _, npow_b = fg.make_batch(x_b, bits_b, slot=1, tx_out=tx_b, out_H=H_b, snr=snr)
    torch.cuda.synchronize()
    assert gb.offset == 7 and torch.equal(bits_a, bits_b)
    assert torch.equal(torch.view_as_real(H_a).reshape(hshape), H_b)
    assert float((tx_a - tx_b).abs().max()) <= 2e-6 * float(tx_a.abs().max())
    assert float((x_a - x_b).abs().max()) <= 1e-5 * float(x_a.abs().max())
    assert abs(float(npow_a) - float(npow_b)) <= 1e-6 * float(npow_a)
    if gb.mixed:                             # neighbouring frames really run different channels
        Hc = torch.view_as_complex(H_b)[:, 0]
        assert float((Hc[0].abs() - Hc[1].abs()).abs().max()) > 1e-3


@pytest.mark.parametrize("frames,nbits,chan", [(1170, 2, "EPA"), (73, 4, "EPA"), (300, 2, "mixRayleigh")])
def test_generated_steps_equal_pipelined_steps_on_the_materialised_batches(frames, nbits, chan):
    """RxEngine.train_step_generated (ONE C call per batch: generator launch + the four step launches, R0 reading (y, noise,
    power partials) as its virtual input) against the same generator materialising x for train_step_pipelined: the same
    batches in the same order => the same bits in every parameter, Adam slot and metric after six steps -- and the
    materialised copy the virtual path can leave behind (keep_x) is the batch the other path trained on."""
    from dl_ofdm_amd import ofdm, receiver as R
    from dl_ofdm_amd.datagen import DeviceDataGen, FusedStaticGen
    from dl_ofdm_amd.engine import RxEngine
    F = flags(nbits=nbits, channel=chan)
    o = ofdm.ofdm_tx(F)
    dims = R.rx_dims(F, o)
    engs, gens, fgs = [], [], []
    for _ in range(2):
        engs.append(RxEngine(dims, frames, train=True, seed=5, want_prob=False, want_z=False))
        gens.append(DeviceDataGen(F, o, seed=21))
        fgs.append(FusedStaticGen(gens[-1], frames, 7.0, want_noise_power=True))
    ea, eb = engs
    n = 6
    xs = []
    for i in range(n):
        ea.train_step_generated(fgs[0], slot=i & 1, last=(i + 1 == n), keep_x=True)
        xs.append(ea.x.clone())                       # (the batch step i normalised ahead: batch i + 1)
    # the reference loop: batch 0, then every step materialises the next batch and normalises it behind its Adam update
    fgs[1].make_batch(eb.x, eb.label_slot(0), 0)
    eb.prime()
    for i in range(n):
        last = i + 1 == n
        if not last:
            fgs[1].make_batch(eb.x, eb.label_slot((i + 1) & 1), (i + 1) & 1)
            assert torch.equal(eb.x, xs[i]), i
        eb.train_step_pipelined(slot=i & 1, last=last)
    torch.cuda.synchronize()
    assert gens[0].offset == gens[1].offset == n
    for name in ("params", "adam_m", "adam_v", "adam_state"):
        assert torch.equal(getattr(ea, name), getattr(eb, name)), name
    ma, mb = ea.metrics(), eb.metrics()
    assert ma["conf"] == mb["conf"] and ma["ce_mean"] == mb["ce_mean"] and ma["tx_power"] == mb["tx_power"]
    assert torch.equal(fgs[0].npow, fgs[1].npow)


def test_generated_step_refuses_the_double_buffered_plan_and_apply_takes_no_noise_monitor():
    """edge cases of the fused-generator ABI: (1) dccn_rx_buffers.gen_next together with x_norm_next (the double-buffered
    pipelining normalises on the backward launch, which has no virtual-input form) is refused, nothing is launched and the
    parameters stay as they were; (2) dccn_gen_static_apply on a generator armed WITHOUT the noise-power monitor leaves the
    monitor output alone."""
    import ctypes as C
    from dl_ofdm_amd import _lib, ofdm, receiver as R
    from dl_ofdm_amd.datagen import DeviceDataGen, FusedStaticGen
    from dl_ofdm_amd.engine import RxEngine
    F = flags(nbits=2, channel="EPA")
    o = ofdm.ofdm_tx(F)
    eng = RxEngine(R.rx_dims(F, o), 73, device="cuda", train=True, seed=1, want_prob=False, want_z=False, want_dfft=False)
    gen = DeviceDataGen(F, o, seed=3)
    fg = FusedStaticGen(gen, 73, 10.0)
    eng.train_step_generated(fg, slot=0)
    torch.cuda.synchronize()
    before = eng.params.clone()
    d = fg.arm(eng.label_slot(0), 0)
    good = eng._pipe_buffers(1, False, 0, False, 1, 0, C.addressof(d), False)
    vals = {f: getattr(good, f) for f, _ in _lib.RxBuffers._fields_}
    vals["x_norm_next"] = vals["x_norm"]
    bad = _lib.RxBuffers(*[vals[f] for f, _ in _lib.RxBuffers._fields_])
    rc = eng.lib.dccn_rx_train_step(C.byref(eng.shape), C.byref(bad), eng.hp, eng._stream())
    torch.cuda.synchronize()
    assert rc == -1 and torch.equal(before, eng.params)            # DCCN_ERR_INVALID_ARG (include/dccn.h)
    # (2)
    assert fg.npart is None and fg.npow is None
    x = torch.empty(73, gen.S, gen.K + gen.CP, 2, device="cuda")
    bits = torch.empty(73, o.frame_size, 2, dtype=torch.int32, device="cuda")
    _, _, npow = fg.make_batch(x, bits, slot=0)
    torch.cuda.synchronize()
    assert npow is None and torch.isfinite(x).all() and float(x.abs().max()) > 0


def test_generated_step_beyond_the_single_pass_normalisation_is_refused_before_anything_runs():
    """receiver.train grows its batch with falling BER (ideal_batch_size: 2045 frames once BER < ~3e-4 at BPSK); the pipelined
    normalisation forms a batch from the fused generator's output for <= 1536 frames only.  The query says so
    (dccn_rx_gen_next_supported, FusedStaticGen.supported(gen, eng)), a step handed gen_next anyway returns
    DCCN_ERR_INVALID_ARG with NOTHING launched -- parameters, Adam slots and the step counter untouched -- and the harness's
    loop trains such batches through the materialised path."""
    import ctypes as C
    from dl_ofdm_amd import ofdm, receiver as R
    from dl_ofdm_amd.datagen import DeviceDataGen, FusedStaticGen
    from dl_ofdm_amd.engine import RxEngine
    F = flags(nbits=1, channel="AWGN")
    o = ofdm.ofdm_tx(F)
    dims = R.rx_dims(F, o)
    small = RxEngine(dims, 1536, train=True, seed=1, want_prob=False, want_z=False)
    big = RxEngine(dims, 2045, train=True, seed=1, want_prob=False, want_z=False)
    gen = DeviceDataGen(F, o, seed=3)
    assert FusedStaticGen.supported(gen) and FusedStaticGen.supported(gen, small) and not FusedStaticGen.supported(gen, big)
    fg = FusedStaticGen(gen, 2045, 10.0)
    fg.make_batch(big.x, big.label_slot(0), 0)
    big.prime()
    torch.cuda.synchronize()
    before = [t.clone() for t in (big.params, big.adam_m, big.adam_v, big.adam_state)]
    d = fg.arm(big.label_slot(1), 1)
    bufs = big._pipe_buffers(0, False, 0, False, 1, 0, C.addressof(d), False)
    rc = big.lib.dccn_rx_train_step(C.byref(big.shape), C.byref(bufs), big.hp, big._stream())
    torch.cuda.synchronize()
    assert rc == -1                                                 # DCCN_ERR_INVALID_ARG
    for a, b in zip(before, (big.params, big.adam_m, big.adam_v, big.adam_state)):
        assert torch.equal(a, b)
    # the harness at a batch beyond the range: BPSK on AWGN at 10 dB reaches 2045 frames within a few epochs
    Ft = R.Flags(nbits=1, nfilter=64, channel="AWGN", SNR=10.0, max_epoch_num=6, early_stop=200, token="big_batch",
                 save_dir="/tmp/dccn_big_batch/", device_data=True, seed=1)
    res = R.train(Ft, device="cuda", verbose=False, run_test=False)
    assert len(res["history"]) == 6 and np.isfinite(res["history"][-1]["train_loss"])
    assert max(h.get("batch_size", 0) for h in res["history"]) > 1536 or res["history"][-1]["test_ber"] > 3e-4
