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
  LS-CP       [5]  LS + cyclic-prefix exploitation (dev/m/cpenhanced.m, Quadeer 2010): the received prefix samples give
                   L extra equations per symbol -- a time-domain LS re-estimate of the channel taps from the prefix, then
                   a joint least-squares solve of [diag(G); H_L Q_cp] X = [Y; y_cp - ISI] for the symbol's N carriers
  ALMMSE-CP   [6]  the same on top of the ALMMSE estimate
  LMMSE-UniPDP [8] / LMMSE-ExpPDP [9]  LMMSE with the correlation of a uniform / exponential power-delay profile of the
                   channel's tap count and RMS delay spread (dev/m/mmse_pdp.m, rms_delay_spread.m; Hung & Lin 2010)

Symbol mapping: ``mapping="table"`` uses ofdm.py's constellation tables (what the DCCN experiments transmit);
``mapping="gray"`` uses the Gray-coded rectangular QAM of MATLAB's ``qammod(.,M,'gray')`` that the MATLAB script
transmits (OFDM_Benchmark_dev.m:243-254: pilot = peak amplitude * (1+i)/sqrt(2), PAPR clipped to 8) -- kept HERE,
separate from ofdm.py, as two different transmitters.

What differs from the MATLAB script, on purpose: the default transmitter is ofdm.py's (its constellation tables, pilot
value 3+3i, no PAPR clipping), the channel is radio.py's, SNR is radio.AWGN_channel_np's definition (unit mean
sample power), and the LMMSE noise term is the LS error variance at the pilots, c = N sigma^2 / |pilot|^2.
Timing: radio.py's channel filter is a centred ('same') convolution, MATLAB's ``filter`` is causal; the receivers here
therefore align their FFT window to the causal response first (``aligned=True``, see ClassicalReceiver.receive).
UNPINNED against MATLAB output (no MATLAB/Octave here, no result files in the reference); pinned instead by closed
forms: perfect-CSI BER on AWGN and on flat Rayleigh fading (tests/test_benchmark.py).
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import numpy as np

from . import ofdm, radio, util

EST_NAMES = ("Perfect", "LS-Spline", "LS-Linear", "LMMSE", "ALMMSE", "LMMSE-Fast", "LS-CP", "ALMMSE-CP", "LMMSE-UniPDP",
             "LMMSE-ExpPDP")
MOD_NAMES = ("BPSK", "QPSK", "8QAM", "16QAM")


def gray_qam_table(nbits: int):
    """(constellation [2^nbits], label bits [2^nbits, nbits]) of ``qammod(0:M-1, M, 'gray')`` on the odd-integer lattice:
    2: -1, +1; 4: 2x2; 8: 4 (real) x 2 (imag); 16: 4x4.  The high bits Gray-index the column (real axis, left to right),
    the low bits Gray-index the row (imaginary axis, top to bottom).  Restated from the toolbox documentation; BER only
    depends on the Gray property and the geometry, not on which of its symmetric labelings MATLAB uses."""
    if nbits == 1:
        return np.array([-1.0, 1.0], dtype=np.complex128), np.array([[0], [1]], dtype=np.int32)
    nx, ny = {2: (1, 1), 3: (2, 1), 4: (2, 2)}[nbits]                     # bits on the real / imaginary axis
    gray_inv = lambda g, n: int(np.argwhere([(i ^ (i >> 1)) == g for i in range(1 << n)])[0, 0])   # noqa: E731
    M = 1 << nbits
    pts = np.empty(M, dtype=np.complex128)
    for d in range(M):
        cx, cy = d >> ny, d & ((1 << ny) - 1)
        x = -(2 ** nx - 1) + 2 * gray_inv(cx, nx)
        y = (2 ** ny - 1) - 2 * gray_inv(cy, ny)
        pts[d] = complex(x, y)
    labels = ((np.arange(M)[:, None] >> np.arange(nbits - 1, -1, -1)[None, :]) & 1).astype(np.int32)
    return pts, labels


def rms_delay_spread(tau, pdb):
    """dev/m/rms_delay_spread.m: (Trms, Tmean) of a power-delay profile (delays tau, powers in dB)."""
    tau, pli = np.asarray(tau, dtype=np.float64), 10.0 ** (np.asarray(pdb, dtype=np.float64) / 10.0)
    tmean = float(tau @ pli / pli.sum())
    return float(np.sqrt(((tau - tmean) ** 2) @ pli / pli.sum())), tmean


def mmse_pdp(L: int, N: int, Trms: float, uniform: bool) -> np.ndarray:
    """dev/m/mmse_pdp.m: channel frequency-correlation matrix [N,N] of a uniform or exponential power-delay profile of
    length L samples (Trms in samples); unit diagonal."""
    d = np.arange(N)[:, None] - np.arange(N)[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        if uniform:
            a = 2j * np.pi * L * d / N
            R = (1.0 - np.exp(-a)) / a
        else:
            b = (1.0 / Trms) + 2j * np.pi * d / N
            R = (1.0 - np.exp(-L * b)) / (Trms * (1.0 - np.exp(-L / Trms)) * b)
    R = np.asarray(R, dtype=np.complex128)
    R[np.arange(N), np.arange(N)] = 1.0
    return R


def _conv_matrix(upper: np.ndarray, lower: np.ndarray) -> np.ndarray:
    """cpenhanced.m ``circshift_comb``: [.., L, L] Toeplitz matrices T[r, c] = lower[r - c] (r >= c) / upper[L + r - c]
    (r < c) -- the convolution matrix of the sequence [upper | lower] restricted to the L outputs of ``lower``."""
    L = lower.shape[-1]
    r, c = np.arange(L)[:, None], np.arange(L)[None, :]
    seq = np.concatenate([upper, lower], axis=-1)                          # [.., 2L]
    return seq[..., L + r - c]


def cp_enhanced(Y: np.ndarray, G: np.ndarray, y_time: np.ndarray, N: int, L: int) -> np.ndarray:
    """dev/m/cpenhanced.m, batched over frames.  Y, G [n, S, N] (received / estimated channel per symbol and carrier),
    y_time [n, S, N+L] received samples incl. the cyclic prefix.  Returns the equalised symbols X [n, S, N]."""
    n, S, _ = Y.shape
    Qinv = np.conj(np.fft.fft(np.eye(N))) / N                             # conj(dftmtx(N))/N
    Q_cp = Qinv[N - L:N, :]                                               # rows producing the prefix samples
    x_ls = np.fft.ifft(Y / G, axis=-1)                                    # time-domain LS symbols
    X = np.zeros_like(Y)
    prev = np.zeros((n, L), dtype=np.complex128)
    eyeL = np.eye(L)
    for j in range(S):
        cur = x_ls[:, j, N - L:N]
        T = _conv_matrix(prev, cur)                                       # [n, L, L]
        ycp = y_time[:, j, :L]
        # prefix-based LS re-estimate of the L channel taps: h = (T^H T)^-1 T^H y_cp (pinv when ill-conditioned)
        A = np.conj(np.swapaxes(T, 1, 2)) @ T
        rhs = (np.conj(np.swapaxes(T, 1, 2)) @ ycp[..., None])[..., 0]
        h = np.empty((n, L), dtype=np.complex128)
        cond_ok = 1.0 / np.linalg.cond(A) >= 1e-10
        if cond_ok.any():
            h[cond_ok] = np.linalg.solve(A[cond_ok], rhs[cond_ok][..., None])[..., 0]
        if (~cond_ok).any():
            h[~cond_ok] = (np.linalg.pinv(A[~cond_ok]) @ rhs[~cond_ok][..., None])[..., 0]
        H_L = _conv_matrix(np.zeros_like(h), h)                           # lower-triangular Toeplitz of the taps
        # ISI of the previous symbol: [0 | triu(T(:,2:L),1)] (the script's own, one-diagonal-short, upper part)
        r, c = np.arange(L)[:, None], np.arange(L)[None, :]
        Tu = T * (c - r >= 2)
        B = np.concatenate([G[:, j, :, None] * np.eye(N)[None], H_L @ Q_cp[None]], axis=1)       # [n, N+L, N]
        C = np.concatenate([Y[:, j], ycp - (Tu @ h[..., None])[..., 0]], axis=1)                  # [n, N+L]
        BhB = np.conj(np.swapaxes(B, 1, 2)) @ B
        BhC = (np.conj(np.swapaxes(B, 1, 2)) @ C[..., None])[..., 0]
        ok = 1.0 / np.linalg.cond(BhB) >= 1e-10
        if ok.any():
            X[ok, j] = np.linalg.solve(BhB[ok], BhC[ok][..., None])[..., 0]
        if (~ok).any():
            X[~ok, j] = (np.linalg.pinv(BhB[~ok]) @ BhC[~ok][..., None])[..., 0]
        prev = cur
    del eyeL
    return X


def _green(r: np.ndarray) -> np.ndarray:
    """biharmonic Green's function in 2-D: r^2 (ln r - 1), 0 at r = 0"""
    out = np.zeros_like(r)
    nz = r > 0
    out[nz] = r[nz] ** 2 * (np.log(r[nz]) - 1.0)
    return out


class ClassicalReceiver:
    def __init__(self, FLAGS, ofdmobj=None, mapping: str = "table"):
        self.o = o = ofdmobj or ofdm.ofdm_tx(FLAGS)
        self.mapping = mapping
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
        if mapping == "table":                                         # ofdm.py's tables, pilot 3+3i
            self.table = ofdm.const_map(self.nbits).astype(np.complex128)
            m = len(self.table)
            self.labels = ((np.arange(m)[:, None] >> np.arange(self.nbits - 1, -1, -1)[None, :]) & 1).astype(np.int32)
            self.pilot_value = complex(o.pilotValue)
        elif mapping == "gray":                                        # MATLAB qammod(.,'gray'), pilot at the peak amplitude
            self.table, self.labels = gray_qam_table(self.nbits)
            self.pilot_value = complex(np.abs(self.table).max() * np.sqrt(0.5) * (1 + 1j))       # :250-251
        else:
            raise ValueError("mapping must be 'table' or 'gray'")
        self.beta = float(np.mean(np.abs(self.table) ** 2) * np.mean(1.0 / np.abs(self.table) ** 2))
        self.papr_clip = 8.0 if mapping == "gray" else None            # :261-268
        self._pdp = {}

    # ---- transmitter of the chosen mapping -----------------------------------------------------------------
    def transmit(self, bits: np.ndarray) -> np.ndarray:
        """bits [n, D, nbits] -> complex frames [n, S, K+CP].  'table': exactly ofdm.py's transmitter; 'gray': the
        MATLAB script's (Gray QAM, peak-amplitude pilot, PAPR clipped to 8 per OFDM symbol)."""
        if self.mapping == "table":
            return self.o.ofdm_tx_frame_np(bits)[0]
        n = bits.shape[0]
        w = (1 << np.arange(self.nbits - 1, -1, -1)).astype(np.int64)
        grid = np.zeros((n, self.S * self.K), dtype=np.complex128)
        grid[:, self.dat] = self.table[bits.astype(np.int64) @ w]
        grid[:, self.pil] = self.pilot_value
        t = np.fft.ifft(grid.reshape(n * self.S, self.K), axis=-1)
        t = np.concatenate([t[:, self.K - self.CP:], t], axis=1)
        pw = np.abs(t) ** 2
        lim = self.papr_clip * pw.mean(axis=1, keepdims=True)
        over = pw > lim
        t = np.where(over, t / np.maximum(np.abs(t), 1e-30) * np.sqrt(lim), t)
        return t.reshape(n, self.S, self.K + self.CP)

    def pdp_correlation(self, fading: radio.rayleigh_chan_lte, uniform: bool, aligned: bool = True) -> np.ndarray:
        """Rhh_uni / Rhh_exp of OFDM_Benchmark_dev.m:197-200: L = number of taps of the profile, Trms in samples
        (mmse_pdp.m), with the model profile's mean delay moved onto the channel's (Hung & Lin 2010 estimate that
        shift from the pilots; here it is known): a correlation matrix only fixes delays relative to its own origin."""
        key = (bool(uniform), bool(aligned))
        if key not in self._pdp:
            prof = fading.profiles[0]
            trms, tmean = rms_delay_spread(np.asarray(prof.tap_delay, dtype=np.float64) * 1e-9, prof.tap_powdB)
            trms_s, tmean_s = max(trms * self.o.Fs, 1e-6), tmean * self.o.Fs
            L = int(prof.n_taps)
            R = mmse_pdp(L, self.K, trms_s, uniform)
            mu_model = L / 2.0 if uniform else trms_s - L * np.exp(-L / trms_s) / (1.0 - np.exp(-L / trms_s))
            mu_true = tmean_s                                              # relative to the centre tap = phase reference
            d = np.exp(-2j * np.pi * np.arange(self.K) * (mu_true - mu_model) / self.K)
            self._pdp[key] = d[:, None] * R * np.conj(d)[None, :]
        return self._pdp[key]

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

    def long_term_correlation(self, fading: radio.rayleigh_chan_lte, aligned: bool = True) -> np.ndarray:
        """E[h h^H] of the channel's frequency response from its power-delay profile (unit-gain units)"""
        prof = fading.profiles[0]
        A = np.asarray(prof.alpha, dtype=np.float64)                      # [n_taps, L]
        Rgg = (A * (np.asarray(prof.ch_coeff) ** 2)[:, None]).T @ A      # E[g g^H], taps independent, unit variance
        F = np.exp(-2j * np.pi * np.outer(np.arange(self.K), np.arange(A.shape[1])) / self.K)
        F = self.ramp(self.advance_of(fading))[:, None] * F                # phase reference = the centre tap (see receive)
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
        if method in ("LMMSE-UniPDP", "LMMSE-ExpPDP"):                   # W from a model power-delay profile, applied to
            if R_long is None:                                            # the frame-averaged LS estimate (:399-416)
                raise ValueError("%s needs the profile correlation (pdp_correlation(fading, uniform))" % method)
            v = G_ls.reshape(n, S, K).mean(axis=1)
            gain = max(float(np.mean(np.abs(G_ls) ** 2)) - c, 1e-12)      # R_long has a unit diagonal
            R = gain * R_long
            Wp = R @ np.linalg.inv(R + c * np.eye(K))
            return np.repeat((v @ Wp.T)[:, None, :], S, axis=1).reshape(n, S * K)
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
                R_long: Optional[np.ndarray] = None, advance: int = 0, aligned: bool = True) -> np.ndarray:
        """rx frames after channel + AWGN (radio.AWGN_channel_np) -> detected bits [n, D, nbits].
        H_true: the channel response returned by ``fading.run`` ([n,S,K]); needed by Perfect / LMMSE;
        advance: ``advance_of(fading)``.

        aligned (default): the receiver first delays the frame by ``advance`` samples, i.e. places its FFT window where
        the channel is causal -- the timing MATLAB's causal ``filter(h, .)`` gives the script's receivers for free -- and
        removes the linear phase that window position implies (the response is referenced to the centre tap either way).
        radio.py filters with a centred 'same' convolution, so without that step the (L-1)/2 pre-cursor taps reach into
        the NEXT symbol's prefix: inter-symbol interference no cyclic prefix covers, an error floor at high SNR (1.4 % for
        16-QAM on EVA even with perfect channel knowledge).  ``aligned=False`` keeps the frame as radio.py delivers it
        (the round-1 tables)."""
        n = rx.shape[0]
        r = rx[..., 0] + 1j * rx[..., 1] if not np.iscomplexobj(rx) else rx
        rot = np.tile(self.ramp(advance), self.S)[None, :]                # phase reference = the centre tap
        if aligned and advance > 0:
            flat = r.reshape(n, -1)
            flat = np.concatenate([np.zeros((n, advance), dtype=flat.dtype), flat[:, :flat.shape[1] - advance]], axis=1)
            r = flat.reshape(r.shape)
            # the window now starts `advance` samples inside the prefix: a known linear phase over the carriers, taken
            # out again so that the response stays as smooth for the pilot interpolation as the centred one
            Y = self.to_frequency(r) * rot
        else:
            Y = self.to_frequency(r)
        sigma2 = float(np.mean(10.0 ** (-np.asarray(snr_db, dtype=np.float64) / 10.0)))
        noise_var = self.K * sigma2                                       # unnormalised N-point FFT of CN(0, sigma2)
        G_true = None
        if H_true is not None:
            Ht = (np.asarray(H_true).reshape(n, self.S, self.K) * self.ramp(advance)).reshape(n, -1)
            # the AWGN stage divided the frames by sqrt(mean power): fold that scalar into the true response
            a = np.sum(Y[:, self.pil] * np.conj(Ht[:, self.pil] * self.pilot_value)) / \
                np.sum(np.abs(Ht[:, self.pil] * self.pilot_value) ** 2)
            G_true = Ht * a
        if method in ("LS-CP", "ALMMSE-CP"):
            if not aligned:
                raise ValueError("the cyclic-prefix methods model a causal channel: use aligned=True")
            if method == "LS-CP":                                         # :417-423 frame-averaged LS-spline estimate
                G_ls = (Y[:, self.pil] / self.pilot_value) @ self.W_spline.T
                G0 = np.repeat(G_ls.reshape(n, self.S, self.K).mean(axis=1, keepdims=True), self.S, axis=1)
            else:                                                         # :424-436
                G0 = self.estimate(Y, "ALMMSE", noise_var).reshape(n, self.S, self.K)
            # cpenhanced.m's time-domain equations are those of the causal response: back to the window's own phase
            back = np.conj(rot).reshape(1, self.S, self.K)
            X = cp_enhanced(Y.reshape(n, self.S, self.K) * back, G0 * back, r, self.K, self.CP)
            return self.demap(X.reshape(n, -1)[:, self.dat])
        G = self.estimate(Y, method, noise_var, G_true, R_long)
        return self.demap(Y[:, self.dat] / G[:, self.dat])


class CurvePoints:
    """One estimator on one (modulation, channel): SNR point ``i`` of a curve as an independent unit -- its draws depend
    only on (seed, i), so the points of a curve may be evaluated in any order, on any rank (config5 deals them round-robin
    over the GPUs' host processes).  ``point`` returns (bit errors, bits)."""

    def __init__(self, FLAGS, method: str, n_frames: int = 2000, seed: int = 1, mobile: bool = False, mapping: str = "table",
                 aligned: bool = True):
        self.FLAGS, self.method, self.n_frames, self.seed, self.aligned = FLAGS, method, int(n_frames), int(seed), aligned
        self.o = ofdm.ofdm_tx(FLAGS)
        self.rxr = ClassicalReceiver(FLAGS, self.o, mapping=mapping)
        self.fading = radio.rayleigh_chan_lte(FLAGS, self.o.Fs, mobile=mobile)
        self.awgn = FLAGS.channel.lower() == "awgn"
        if method == "LMMSE-Fast" and not self.awgn:
            self.R_long = self.rxr.long_term_correlation(self.fading, aligned)
        elif method in ("LMMSE-UniPDP", "LMMSE-ExpPDP") and not self.awgn:
            self.R_long = self.rxr.pdp_correlation(self.fading, uniform=(method == "LMMSE-UniPDP"), aligned=aligned)
        else:
            self.R_long = np.ones((self.o.K, self.o.K), dtype=np.complex128)

    def point(self, i: int, snr: float):
        np.random.seed(self.seed + 7919 * i)
        bits = util.bit_source(self.FLAGS.nbits, self.o.frame_size, self.n_frames)
        iq = self.rxr.transmit(bits)
        y, H = self.fading.run(iq)
        rx, _ = radio.AWGN_channel_np(y, snr * np.ones((self.n_frames, 1)))
        det = self.rxr.receive(rx, self.method, snr, H_true=H, R_long=self.R_long,
                               advance=0 if self.awgn else self.rxr.advance_of(self.fading), aligned=self.aligned)
        return int(np.count_nonzero(det != bits)), int(bits.size)


def ber_curve(FLAGS, method: str, snrs: Sequence[float], n_frames: int = 2000, seed: int = 1, mobile: bool = False,
              mapping: str = "table", aligned: bool = True):
    """BER of one estimator over an SNR list on FLAGS.channel / FLAGS.nbits (bits -> transmitter of the chosen mapping
    -> radio.py channel + AWGN)."""
    cp = CurvePoints(FLAGS, method, n_frames, seed, mobile, mapping, aligned)
    out = []
    for i, snr in enumerate(snrs):
        e, n = cp.point(i, snr)
        out.append(e / n)
    return np.asarray(out)


def run_benchmark(FLAGS, methods: Sequence[str] = EST_NAMES, snrs: Sequence[float] = tuple(range(-10, 31, 5)),
                  n_frames: int = 2000, out_dir: str = ".", mobile: bool = False, mapping: str = "table",
                  aligned: bool = True) -> Dict[str, str]:
    """one CSV per estimator, laid out like the MATLAB script's ``berofdm_all`` (row 0 = SNRs, then one row per
    modulation BPSK, QPSK, 8QAM, 16QAM): ``BER_OFDM_<channel>_<estimator>_lte_<N>_Table[_mobile][_shortcp].csv``
    ("Table": ofdm.py's constellation tables; "Gray": MATLAB's Gray qammod -- the ``mapping`` argument)."""
    import copy
    os.makedirs(out_dir, exist_ok=True)
    paths = {}
    for method in methods:
        table = np.zeros((5, len(snrs)))
        table[0] = snrs
        for nb in (1, 2, 3, 4):
            fl = copy.copy(FLAGS)
            fl.nbits = nb
            table[nb] = ber_curve(fl, method, snrs, n_frames=n_frames, mobile=mobile, mapping=mapping, aligned=aligned)
        name = "BER_OFDM_%s_%s_%s_%d_%s%s%s.csv" % (FLAGS.channel, method, FLAGS.pilot, FLAGS.nfft,
                                                    "Table" if mapping == "table" else "Gray",
                                                    "_mobile" if mobile else "", "" if FLAGS.longcp else "_shortcp")
        paths[method] = os.path.join(out_dir, name)
        np.savetxt(paths[method], table, delimiter=",")
    return paths
