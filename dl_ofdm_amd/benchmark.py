"""Classical OFDM receivers for "DCCN vs LS / LMMSE" curves -- a NumPy restatement of the
estimator family of dev/m/OFDM_Benchmark_dev.m:339-456 on THIS repo's own substrate (ofdm.py frames, radio.py
channels), so the baseline curves can be drawn without MATLAB.  Host code, not part of the GPU hot path.

Estimators (MATLAB ``eq_idx`` in brackets):
  Perfect     [1]  true channel response
  LS-Spline   [2]  LS at the pilots, biharmonic-spline interpolation over the (subcarrier, symbol) grid
                   (``griddata(...,'v4')``: Green's function r^2 (ln r - 1), Sandwell 1987)
  LS-Linear   [3]  LS at the pilots, piecewise-linear scattered interpolation (``scatteredInterpolant``;
                   outside the pilots' convex hull the nearest pilot is used -- MATLAB extrapolates linearly)
  LMMSE       [4]  ideal per-symbol LMMSE: W = Rhh (Rhh + c I)^-1 with the true rank-one Rhh = h h^H
  ALMMSE      [7]  approximate LMMSE from the frame-averaged LS estimate
  LMMSE-Fast [10]  LMMSE with the long-term channel correlation of the power-delay profile
(The CP-enhanced variants [5,6] and the uniform/exponential-PDP ones [8,9] are not restated.)

What differs from the MATLAB script, on purpose: the transmitter is ofdm.py's (its constellation tables, pilot
value 3+3i, no PAPR clipping), the channel is radio.py's, SNR is radio.AWGN_channel_np's definition (unit mean
sample power), and the LMMSE noise term is the LS error variance at the pilots, c = N sigma^2 / |pilot|^2.
UNPINNED against MATLAB output (no MATLAB/Octave here, no result files in the reference); pinned instead by closed
forms: perfect-CSI BER on AWGN and on flat Rayleigh fading (tests/test_benchmark.py).
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import numpy as np

from . import ofdm, radio, util

EST_NAMES = ("Perfect", "LS-Spline", "LS-Linear", "LMMSE", "ALMMSE", "LMMSE-Fast")
MOD_NAMES = ("BPSK", "QPSK", "8QAM", "16QAM")


def _green(r: np.ndarray) -> np.ndarray:
    """biharmonic Green's function in 2-D: r^2 (ln r - 1), 0 at r = 0"""
    out = np.zeros_like(r)
    nz = r > 0
    out[nz] = r[nz] ** 2 * (np.log(r[nz]) - 1.0)
    return out


class ClassicalReceiver:
    def __init__(self, FLAGS, ofdmobj=None):
        self.o = o = ofdmobj or ofdm.ofdm_tx(FLAGS)
        self.K, self.S, self.CP, self.nbits = o.K, o.nSymbol, o.CP, int(FLAGS.nbits)
        self.channel = FLAGS.channel
        K, S = self.K, self.S
        self.pil = np.asarray(o.pilotSc)                    # flat symbol*K + carrier
        self.dat = np.asarray(o.dataSc)
        pf, pt = (self.pil % K).astype(np.float64) + 1, (self.pil // K).astype(np.float64) + 1     # MATLAB 1-based grid
        gt, gf = np.meshgrid(np.arange(1, S + 1, dtype=np.float64), np.arange(1, K + 1, dtype=np.float64), indexing="ij")
        grid = np.stack([gf.ravel(), gt.ravel()], -1)       # [S*K, 2], flat index = symbol*K + carrier
        pts = np.stack([pf, pt], -1)
        # griddata 'v4': weights solve G_pp w = v, value(x) = sum_j w_j g(|x - x_j|)  ->  one [S*K, P] matrix
        Gpp = _green(np.linalg.norm(pts[:, None, :] - pts[None, :, :], axis=-1))
        Gxp = _green(np.linalg.norm(grid[:, None, :] - pts[None, :, :], axis=-1))
        self.W_spline = Gxp @ np.linalg.inv(Gpp)
        self.W_linear = self._linear_matrix(pts, grid)
        self.table = ofdm.const_map(self.nbits).astype(np.complex128)
        m = len(self.table)
        self.labels = ((np.arange(m)[:, None] >> np.arange(self.nbits - 1, -1, -1)[None, :]) & 1).astype(np.int32)
        self.beta = float(np.mean(np.abs(self.table) ** 2) * np.mean(1.0 / np.abs(self.table) ** 2))
        self.pilot_value = complex(o.pilotValue)

    @staticmethod
    def _linear_matrix(pts, grid):
        from scipy.interpolate import LinearNDInterpolator, NearestNDInterpolator
        P = len(pts)
        W = np.zeros((len(grid), P))
        eye = np.eye(P)
        lin = LinearNDInterpolator(pts, eye)
        near = NearestNDInterpolator(pts, eye)
        v = lin(grid)
        bad = np.isnan(v).any(axis=1)
        v[bad] = near(grid[bad])
        W[:] = v
        return W

    # ---- pieces ------------------------------------------------------------------------------------------
    def to_frequency(self, rx: np.ndarray) -> np.ndarray:
        """rx [n,S,n_sc,2] (or complex [n,S,n_sc]) -> Y [n, S*K] after CP removal and an N-point FFT"""
        if not np.iscomplexobj(rx):
            rx = rx[..., 0] + 1j * rx[..., 1]
        return np.fft.fft(rx[:, :, self.CP:self.CP + self.K], axis=-1).reshape(rx.shape[0], self.S * self.K)

    @staticmethod
    def advance_of(fading: radio.rayleigh_chan_lte) -> int:
        """radio.py filters with np.convolve(..., 'same'): the L-tap impulse response is centred, i.e. the received
        frame is ADVANCED by (L-1)//2 samples relative to the response ``fading.run`` reports -- a linear phase
        ramp exp(+2 pi i k adv / N) over the subcarriers that a receiver with "perfect" knowledge must include."""
        return (int(fading.profiles[0].alpha.shape[1]) - 1) // 2

    def ramp(self, advance: int) -> np.ndarray:
        return np.exp(2j * np.pi * np.arange(self.K) * advance / self.K)

    def long_term_correlation(self, fading: radio.rayleigh_chan_lte) -> np.ndarray:
        """E[h h^H] of the channel's frequency response from its power-delay profile (unit-gain units)"""
        prof = fading.profiles[0]
        A = np.asarray(prof.alpha, dtype=np.float64)                      # [n_taps, L]
        Rgg = (A * (np.asarray(prof.ch_coeff) ** 2)[:, None]).T @ A      # E[g g^H], taps independent, unit variance
        F = np.exp(-2j * np.pi * np.outer(np.arange(self.K), np.arange(A.shape[1])) / self.K)
        F = self.ramp(self.advance_of(fading))[:, None] * F
        return F @ Rgg @ F.conj().T

    def estimate(self, Y: np.ndarray, method: str, noise_var: float, G_true: Optional[np.ndarray] = None,
                 R_long: Optional[np.ndarray] = None) -> np.ndarray:
        """channel estimate on the whole grid [n, S*K].  noise_var: variance of the FFT-domain noise per cell."""
        n, K, S = Y.shape[0], self.K, self.S
        if method == "Perfect":
            return G_true
        g_p = Y[:, self.pil] / self.pilot_value                           # LS at the pilots
        if method == "LS-Linear":
            return g_p @ self.W_linear.T
        G_ls = g_p @ self.W_spline.T
        if method == "LS-Spline":
            return G_ls
        c = noise_var / abs(self.pilot_value) ** 2                        # LS error variance at a pilot
        if method == "LMMSE":                                             # rank-one Rhh = h h^H per symbol
            h = G_true.reshape(n, S, K)
            gl = G_ls.reshape(n, S, K)
            proj = np.sum(np.conj(h) * gl, axis=-1, keepdims=True) / (np.sum(np.abs(h) ** 2, axis=-1, keepdims=True) + c)
            return (h * proj).reshape(n, S * K)
        if method == "ALMMSE":                                            # Rhh = v v^H / S, v = frame-averaged LS estimate
            v = G_ls.reshape(n, S, K).mean(axis=1)
            e = np.sum(np.abs(v) ** 2, axis=-1, keepdims=True) / S
            return np.repeat((v * (e / (e + c)))[:, None, :], S, axis=1).reshape(n, S * K)
        if method == "LMMSE-Fast":
            if R_long is None:
                raise ValueError("LMMSE-Fast needs the long-term correlation (long_term_correlation(fading))")
            # scale the unit-gain correlation to the received amplitude (the AWGN stage normalises the sample power)
            gain = max(float(np.mean(np.abs(G_ls) ** 2)) - c, 1e-12) / float(np.real(np.trace(R_long)) / K)
            R = gain * R_long
            Wf = R @ np.linalg.inv(R + c * np.eye(K))
            return np.einsum("kl,nsl->nsk", Wf, G_ls.reshape(n, S, K)).reshape(n, S * K)
        raise ValueError("unknown estimator %r (one of %s)" % (method, EST_NAMES))

    def demap(self, x_hat: np.ndarray) -> np.ndarray:
        """nearest constellation point -> label bits [n, D, nbits]"""
        idx = np.argmin(np.abs(x_hat[..., None] - self.table[None, None, :]), axis=-1)
        return self.labels[idx]

    def receive(self, rx: np.ndarray, method: str, snr_db, H_true: Optional[np.ndarray] = None,
                R_long: Optional[np.ndarray] = None, advance: int = 0) -> np.ndarray:
        """rx frames after channel + AWGN (radio.AWGN_channel_np) -> detected bits [n, D, nbits].
        H_true: the channel response returned by ``fading.run`` ([n,S,K]); needed by Perfect / LMMSE;
        advance: ``advance_of(fading)``."""
        Y = self.to_frequency(rx)
        sigma2 = float(np.mean(10.0 ** (-np.asarray(snr_db, dtype=np.float64) / 10.0)))
        noise_var = self.K * sigma2                                       # unnormalised N-point FFT of CN(0, sigma2)
        G_true = None
        if H_true is not None:
            Ht = (np.asarray(H_true).reshape(Y.shape[0], self.S, self.K) * self.ramp(advance)).reshape(Y.shape[0], -1)
            # the AWGN stage divided the frames by sqrt(mean power): fold that scalar into the true response
            a = np.sum(Y[:, self.pil] * np.conj(Ht[:, self.pil] * self.pilot_value)) / \
                np.sum(np.abs(Ht[:, self.pil] * self.pilot_value) ** 2)
            G_true = Ht * a
        G = self.estimate(Y, method, noise_var, G_true, R_long)
        return self.demap(Y[:, self.dat] / G[:, self.dat])


def ber_curve(FLAGS, method: str, snrs: Sequence[float], n_frames: int = 2000, seed: int = 1, mobile: bool = False):
    """BER of one estimator over an SNR list on FLAGS.channel / FLAGS.nbits (bits -> ofdm.py TX -> radio.py)."""
    o = ofdm.ofdm_tx(FLAGS)
    rxr = ClassicalReceiver(FLAGS, o)
    fading = radio.rayleigh_chan_lte(FLAGS, o.Fs, mobile=mobile)
    R_long = rxr.long_term_correlation(fading) if (method == "LMMSE-Fast" and FLAGS.channel.lower() != "awgn") else \
        np.ones((o.K, o.K), dtype=np.complex128)
    out = []
    for i, snr in enumerate(snrs):
        np.random.seed(seed + 7919 * i)
        bits = util.bit_source(FLAGS.nbits, o.frame_size, n_frames)
        iq, _, _ = o.ofdm_tx_frame_np(bits)
        y, H = fading.run(iq)
        rx, _ = radio.AWGN_channel_np(y, snr * np.ones((n_frames, 1)))
        det = rxr.receive(rx, method, snr, H_true=H, R_long=R_long,
                          advance=0 if FLAGS.channel.lower() == "awgn" else rxr.advance_of(fading))
        out.append(float(np.mean(det != bits)))
    return np.asarray(out)


def run_benchmark(FLAGS, methods: Sequence[str] = EST_NAMES, snrs: Sequence[float] = tuple(range(-10, 31, 5)),
                  n_frames: int = 2000, out_dir: str = ".", mobile: bool = False) -> Dict[str, str]:
    """one CSV per estimator, laid out like the MATLAB script's ``berofdm_all`` (row 0 = SNRs, then one row per
    modulation BPSK, QPSK, 8QAM, 16QAM): ``BER_OFDM_<channel>_<estimator>_lte_<N>_Table[_mobile][_shortcp].csv``
    ("Table": ofdm.py's constellation tables instead of MATLAB's Gray qammod)."""
    import copy
    os.makedirs(out_dir, exist_ok=True)
    paths = {}
    for method in methods:
        table = np.zeros((5, len(snrs)))
        table[0] = snrs
        for nb in (1, 2, 3, 4):
            fl = copy.copy(FLAGS)
            fl.nbits = nb
            table[nb] = ber_curve(fl, method, snrs, n_frames=n_frames, mobile=mobile)
        name = "BER_OFDM_%s_%s_%s_%d_Table%s%s.csv" % (FLAGS.channel, method, FLAGS.pilot, FLAGS.nfft,
                                                       "_mobile" if mobile else "", "" if FLAGS.longcp else "_shortcp")
        paths[method] = os.path.join(out_dir, name)
        np.savetxt(paths[method], table, delimiter=",")
    return paths
