"""Host-side channel simulation feeding the receiver: LTE multipath Rayleigh fading (static taps or
Jakes sum-of-sinusoids Doppler) and AWGN.  Mirror of the NumPy part of dev/py/radio.py:277-526
(class / method names kept); arithmetic is batched over frames instead of the reference's per-frame
Python loop, drawing from ``np.random`` in the reference's order so seeded runs reproduce its
outputs (tests/test_golden_substrate.py).
"""
from __future__ import annotations

import json
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "lte_tap_interp.json")

# 3GPP TS 36.101/36.104 power-delay profiles: (tap delays [ns], tap powers [dB], Doppler [Hz] when mobile)
_PROFILES = {
    "etu": ([0, 50, 120, 200, 230, 500, 1600, 2300, 5000], [-1.0, -1.0, -1.0, 0.0, 0.0, 0.0, -3.0, -5.0, -7.0], 300.0),
    "epa": ([0, 30, 70, 90, 110, 190, 410], [0.0, -1.0, -2.0, -3.0, -8.0, -17.2, -20.8], 5.0),
    "eva": ([0, 30, 150, 310, 370, 710, 1090, 1730, 2510], [0.0, -1.5, -1.4, -3.6, -0.6, -9.1, -7.0, -12.0, -16.9], 70.0),
    "custom": ([0, 70, 200, 230, 500, 1600, 2700, 3000], [0.0, -1.4, -1.4, -1.0, -3.0, -9.1, -15.0, -19.0], 80.0),
}
_N_SINUSOIDS = 48


def _alpha_matrices():
    with open(_DATA) as f:
        return {k: np.asarray(v, dtype=np.float64) for k, v in json.load(f)["matrices"].items()}


class _Profile:
    def __init__(self, name: str, mobile: bool, alphas):
        name = name.lower()
        if name in _PROFILES:
            delay, powdb, fd = _PROFILES[name]
            self.alpha = alphas[name]
            self.Fd = fd if mobile else 0.0
        else:                                       # 'flat', 'awgn', anything else: single tap
            delay, powdb = [0], [0]
            self.alpha = np.ones((1, 1), dtype=np.float64)
            self.Fd = 5.0 if mobile else 0.0
        self.tap_delay = np.asarray(delay)
        self.tap_powdB = np.asarray(powdb, dtype=np.float64)
        self.n_taps = len(delay)
        lin = 10.0 ** (self.tap_powdB / 10.0)
        self.ch_coeff = lin * (1.0 / np.sqrt(np.sum(lin)))    # power-weighted (as the reference does)


def _convolve_same(tx: np.ndarray, g: np.ndarray) -> np.ndarray:
    """Row-wise ``np.convolve(tx[i], g[i], 'same')`` for tx [n, T] and g [n, L], L <= T."""
    n, T = tx.shape
    L = g.shape[1]
    off = (L - 1) // 2
    pad = np.zeros((n, T + L - 1), dtype=np.complex128)
    pad[:, L - 1 - off:L - 1 - off + T] = tx
    out = np.zeros((n, T), dtype=np.complex128)
    for l in range(L):                                      # y[t] = sum_l g[l] * tx[t + off - l]
        out += g[:, l:l + 1] * pad[:, L - 1 - l:L - 1 - l + T]
    return out


class rayleigh_chan_lte:
    """``fading = rayleigh_chan_lte(FLAGS, Fs[, mobile, mix]); y, H = fading.run(tx_complex)``.

    tx [n_frames, n_sym, n_sc] complex -> y float32 [n_frames, n_sym, n_sc, 2] and the per-symbol
    channel frequency response H complex64 [n_frames, n_sym, nfft].
    channel: 'AWGN' (pass-through), 'Flat', 'EPA', 'EVA', 'ETU', 'Custom', 'mixRayleigh'
    (flat/ETU/EVA/EPA by frame index mod 4, Doppler on every third frame when ``mix``), 'mixAll'.
    """

    def __init__(self, FLAGS, sample_rate=0.96e6, mobile=False, mix=False):
        self.nSymbol = int(FLAGS.nsymbol)
        self.chan = FLAGS.channel.lower()
        self.sample_rate = sample_rate
        self.nfft = int(FLAGS.nfft)
        self.mobile, self.mix = bool(mobile), bool(mix)
        self.ss = _N_SINUSOIDS
        self.const1 = np.sqrt(1.0 / self.ss)
        alphas = _alpha_matrices()
        if self.chan == "mixrayleigh":
            self.profiles = [_Profile(n, mobile, alphas) for n in ("flat", "etu", "eva", "epa")]
        elif self.chan == "mixall":
            self.profiles = [_Profile(n, mobile, alphas) for n in ("awgn", "flat", "etu", "eva", "epa")]
        else:
            self.profiles = [_Profile(self.chan, mobile, alphas)]
        p = self.profiles[-1] if len(self.profiles) == 1 else None
        if p is not None:
            self.Fd, self.n_taps, self.ch_coeff, self.alpha_matrix = p.Fd, p.n_taps, p.ch_coeff, p.alpha
            self.tap_delay, self.tap_powdB = p.tap_delay, p.tap_powdB

    # -- per-frame random draws, in the reference's order ------------------------------------
    def _static_taps(self, prof: _Profile):
        z = np.random.normal(loc=0.0, scale=1.0 / np.sqrt(2), size=[prof.n_taps, 2])
        return (z[:, 0] + 1j * z[:, 1]) * prof.ch_coeff

    def _doppler_taps(self, prof: _Profile, n_sym: int, n_sc: int):
        """Jakes sum-of-sinusoids tap gains per OFDM symbol: [n_sym, n_taps] (radio.py:376-407)."""
        k = np.arange(1, prof.n_taps + 1)
        n = (np.arange(1, self.ss + 1).reshape(self.ss, 1) - 0.5) * np.pi / (4 * self.ss)
        a0 = k * np.pi / (4 * self.ss)
        f_re, f_im = prof.Fd * np.cos(n + a0), prof.Fd * np.cos(n - a0)
        th_re = np.random.uniform(0, 2 * np.pi, size=(self.ss, prof.n_taps))
        th_im = np.random.uniform(0, 2 * np.pi, size=(self.ss, prof.n_taps))
        t = (np.arange(n_sym) * (n_sc / self.sample_rate)).reshape(n_sym, 1, 1)
        mu_re = self.const1 * np.sum(np.cos(2 * np.pi * t * f_re + th_re), axis=1)
        mu_im = self.const1 * np.sum(np.cos(2 * np.pi * t * f_im + th_im), axis=1)
        return (mu_re + 1j * mu_im) * prof.ch_coeff

    def _apply_doppler(self, tx: np.ndarray, taps: np.ndarray, prof: _Profile, n_sym: int, n_sc: int):
        """One frame, per-symbol impulse response, carrying n_taps samples of history
        (radio.py:385-407: convolve 'same' over [n_taps + n_sc] and drop the first n_taps)."""
        nt = prof.n_taps
        pre = np.zeros(nt + n_sym * n_sc, dtype=np.complex64)
        pre[nt:] = tx
        g = taps @ prof.alpha                                         # [n_sym, L]
        seg = np.stack([pre[n_sc * i:nt + n_sc * (i + 1)] for i in range(n_sym)])
        y = _convolve_same(seg, g)[:, nt:]
        return y.reshape(-1), np.fft.fft(g, self.nfft, axis=1)

    # ---------------------------------------------------------------------------------------
    def run(self, inputs: np.ndarray):
        if not np.iscomplexobj(inputs):
            raise AssertionError("complex baseband input expected")
        n_fr, n_sym, n_sc = inputs.shape
        T = n_sym * n_sc
        H = np.zeros((n_fr, n_sym, self.nfft), dtype=np.complex64)
        if self.chan == "awgn":
            y = inputs
            H[:] = 1.0
        else:
            y = np.zeros(inputs.shape, dtype=np.complex64)
            flat_in = inputs.reshape(n_fr, T)
            n_prof = len(self.profiles)
            # draw all random numbers frame by frame (keeps the reference's RNG order), then batch
            static = {i: ([], []) for i in range(n_prof)}
            for fr in range(n_fr):
                pi = fr % n_prof if n_prof > 1 else 0
                prof = self.profiles[pi]
                if self.chan == "mixall" and pi == 0:                  # AWGN slot: identity channel
                    y[fr] = inputs[fr]
                    H[fr] = np.fft.fft(np.array([1 + 0j]), self.nfft)
                    continue
                if n_prof > 1:
                    period = 3 if self.chan == "mixrayleigh" else 4
                    doppler = (fr % period == 0) and prof.Fd > 0.1 and self.mix
                else:
                    doppler = prof.Fd > 0.1
                if doppler:
                    taps = self._doppler_taps(prof, n_sym, n_sc)
                    yy, hh = self._apply_doppler(flat_in[fr], taps, prof, n_sym, n_sc)
                    y[fr] = yy.reshape(n_sym, n_sc)
                    H[fr] = hh
                else:
                    static[pi][0].append(fr)
                    static[pi][1].append(self._static_taps(prof))
            for pi, (frames, taps) in static.items():
                if not frames:
                    continue
                prof = self.profiles[pi]
                g = np.asarray(taps) @ prof.alpha                       # [n, L]
                yy = _convolve_same(flat_in[frames].astype(np.complex128), g)
                y[frames] = yy.reshape(len(frames), n_sym, n_sc)
                H[frames] = np.fft.fft(g, self.nfft, axis=1)[:, None, :]
        y_out = np.stack([np.real(y), np.imag(y)], axis=-1)
        return y_out, H

    def __call__(self, inputs):
        return self.run(inputs)


def AWGN_channel_np(inputs: np.ndarray, SNR):
    """Normalise the batch to unit mean IQ power, add complex white noise at ``SNR`` dB
    (per frame, [n,1]); returns (noisy float64 [n,S,n_sc,2], mean noise power)  (radio.py:513-526)."""
    sig = np.square(inputs[:, :, :, 0:1]) + np.square(inputs[:, :, :, 1:])
    scale = np.sqrt(np.nanmean(sig))
    noise = np.random.randn(*inputs.shape)
    std = np.sqrt(0.5) * np.power(10.0, -np.asarray(SNR, dtype=np.float64) / 20.0)
    noise = noise * np.reshape(std, [-1, 1, 1, 1])
    out = inputs / scale + noise
    npow = np.mean(np.square(noise[:, :, :, 0:1]) + np.square(noise[:, :, :, 1:]))
    return out, npow
