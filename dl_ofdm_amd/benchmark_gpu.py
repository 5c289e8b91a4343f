"""Classical pilot-aided receivers on the GPU: the estimator family of
dev/m/OFDM_Benchmark_dev.m:339-456 -- the curves the paper draws DCCN against -- as a launch sequence over libdccn
(include/dccn.h "classical pilot-aided receivers"), fed by device-resident frames (the device-side generator's, or any
[n, S, K+CP, 2] batch).  The NumPy restatement in :mod:`dl_ofdm_amd.benchmark` is the oracle of this path
(tests/test_gpu_benchmark.py).  Round 4: the interpolation weights, the window's phase ramp and the long-term
correlation are built HERE from first principles (torch float64 on the device: biharmonic Green's-function solve, the
profile's tap matrices), not taken from the host object this path is checked against; and the two cyclic-prefix methods
(LS-CP / ALMMSE-CP, dev/m/cpenhanced.m: per symbol an L x L and an N x N least-squares solve) run on the device as batched
complex128 linear algebra (torch.linalg on the GPU -- this is the baseline tool, not the DCCN hot path).

Per batch:  FFT window (aligned to the causal response, phase ramp folded into the DFT matrix) as ONE real-expanded
[n S, 2K] . [2K, 2K] GEMM -> LS at the pilots -> interpolation over the (subcarrier, symbol) grid as a
[2n, P] . [P, S K] GEMM (biharmonic-spline 'v4' or linear weights, precomputed once) -> estimator stage
(LS / ideal LMMSE / ALMMSE / perfect / long-term LMMSE smoothing as another [., 2K] . [2K, 2K] GEMM) -> one-tap
equalisation, nearest-point demapping and the bit-error count.  No CPU fallback: a CUDA (ROCm) device is required.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib, radio
from ._lib import check
from .benchmark import ClassicalReceiver

GPU_METHODS = ("Perfect", "LS-Spline", "LS-Linear", "LMMSE", "ALMMSE", "LMMSE-Fast", "LMMSE-UniPDP", "LMMSE-ExpPDP",
               "LS-CP", "ALMMSE-CP")
_MODE = {"LS": 0, "LMMSE": 1, "ALMMSE": 2, "Perfect": 3, "FrameMean": 4}


def _cexpand(Mc: np.ndarray) -> np.ndarray:
    """complex [a, b] (acting as y = x . M on interleaved-IQ rows) -> real [2a, 2b]"""
    a, b = Mc.shape
    R = np.zeros((2 * a, 2 * b), dtype=np.float64)
    R[0::2, 0::2], R[0::2, 1::2] = Mc.real, Mc.imag
    R[1::2, 0::2], R[1::2, 1::2] = -Mc.imag, Mc.real
    return R


def spline_weights(pil: np.ndarray, S: int, K: int, device) -> torch.Tensor:
    """MATLAB griddata(..., 'v4') (OFDM_Benchmark_dev.m:345-352) as ONE matrix [S*K, P]: biharmonic-spline interpolation
    value(x) = sum_j w_j g(|x - x_j|), g(r) = r^2 (ln r - 1), with the weights solving g(|x_i - x_j|) w = v at the pilots.
    Built on the device in float64 with a linear solve (W = Gxp . Gpp^-1  <=>  Gpp^T W^T = Gxp^T)."""
    f64 = dict(dtype=torch.float64, device=device)
    pil = torch.as_tensor(np.asarray(pil, dtype=np.int64), device=device)
    pts = torch.stack([(pil % K).to(torch.float64), (pil // K).to(torch.float64)], -1)             # (carrier, symbol)
    cells = torch.arange(S * K, device=device)
    grid = torch.stack([(cells % K).to(torch.float64), (cells // K).to(torch.float64)], -1)

    def green(a, b):
        r = torch.cdist(a, b)
        return torch.where(r > 0, r * r * (torch.log(torch.clamp(r, min=1e-300)) - 1.0), torch.zeros((), **f64))
    return torch.linalg.solve(green(pts, pts).T, green(grid, pts).T).T


def linear_weights(pil: np.ndarray, S: int, K: int) -> np.ndarray:
    """griddata(..., 'linear') weights [S*K, P]: barycentric interpolation on the Delaunay triangulation of the pilot cells,
    nearest pilot outside their convex hull (scipy's triangulation, evaluated on unit vectors)."""
    from scipy.interpolate import LinearNDInterpolator, NearestNDInterpolator
    pil = np.asarray(pil, dtype=np.int64)
    pts = np.stack([pil % K, pil // K], -1).astype(np.float64)
    cells = np.arange(S * K)
    grid = np.stack([cells % K, cells // K], -1).astype(np.float64)
    eye = np.eye(len(pil))
    W = LinearNDInterpolator(pts, eye)(grid)
    out = np.isnan(W).any(axis=1)
    W[out] = NearestNDInterpolator(pts, eye)(grid[out])
    return W


class ClassicalReceiverGPU:
    def __init__(self, FLAGS, ofdmobj=None, mapping: str = "table", device="cuda", mobile: bool = False):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.DccnError("ClassicalReceiverGPU needs a CUDA (ROCm) device; the host receivers are dl_ofdm_amd.benchmark")
        self.host = h = ClassicalReceiver(FLAGS, ofdmobj, mapping=mapping)
        self.FLAGS, self.o = FLAGS, h.o
        self.K, self.S, self.CP, self.nbits = h.K, h.S, h.CP, h.nbits
        self.n_sc = self.K + self.CP
        self.P, self.D = len(h.pil), len(h.dat)
        self.fading = radio.rayleigh_chan_lte(FLAGS, h.o.Fs, mobile=mobile)
        self.awgn = FLAGS.channel.lower() == "awgn"
        self.advance = 0 if self.awgn else h.advance_of(self.fading)
        f32 = dict(dtype=torch.float32, device=self.device)
        i32 = dict(dtype=torch.int32, device=self.device)
        self.pil = torch.as_tensor(h.pil.astype(np.int32), **i32)
        self.dat = torch.as_tensor(h.dat.astype(np.int32), **i32)
        self.table = torch.as_tensor(np.stack([h.table.real, h.table.imag], -1).astype(np.float32), **f32)
        self.labels = torch.as_tensor(np.ascontiguousarray(h.labels, dtype=np.int32), **i32)
        # interpolation weights [P, S K], built here (not the host receiver's: that one is this path's checker)
        self.w_spline64 = spline_weights(h.pil, self.S, self.K, self.device)
        self.w_spline = self.w_spline64.T.contiguous().to(torch.float32)
        self.w_linear = torch.as_tensor(np.ascontiguousarray(linear_weights(h.pil, self.S, self.K).T, dtype=np.float32), **f32)
        self._dft = {}
        self._R = {}
        self.nws = int(self.lib.dccn_classical_workspace_size())
        self.ws = torch.empty(self.nws, dtype=torch.uint8, device=self.device)
        self._sums = torch.zeros(4, dtype=torch.float64, device=self.device)
        self._err = torch.zeros(1, dtype=torch.int64, device=self.device)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dft_matrix(self, aligned: bool) -> torch.Tensor:
        """x_window [., 2K] . M = Y [., 2K]: the N-point DFT with the phase ramp of the window position folded in"""
        key = bool(aligned)
        if key not in self._dft:
            K = self.K
            F = np.exp(-2j * np.pi * np.outer(np.arange(K), np.arange(K)) / K)          # F[t, k]
            if aligned and self.advance > 0:
                F = F * self.ramp()[None, :]
            self._dft[key] = torch.as_tensor(_cexpand(F).astype(np.float32), device=self.device)
        return self._dft[key]

    def ramp(self) -> np.ndarray:
        """exp(+2 pi i k adv / K): the linear phase over the carriers of a response referenced to the centre tap of radio.py's
        'same' convolution ((L-1)//2 samples early)"""
        return np.exp(2j * np.pi * np.arange(self.K) * self.advance / self.K)

    def long_term_correlation(self) -> np.ndarray:
        """E[h h^H] of the frequency response, unit-gain units, from the profile's own tap matrices: the impulse response is
        g = sum_t z_t coeff_t alpha[t, :] with independent unit-variance z_t, so E[g g^H] = alpha^T diag(coeff^2) alpha and
        R = F E[g g^H] F^H with F[k, l] = exp(-2 pi i k l / K) and the centre-tap phase reference."""
        prof = self.fading.profiles[0]
        A = torch.as_tensor(np.asarray(prof.alpha, dtype=np.float64), device=self.device)          # [n_taps, L]
        c2 = torch.as_tensor(np.asarray(prof.ch_coeff, dtype=np.float64) ** 2, device=self.device)
        Rgg = (A * c2[:, None]).T @ A
        k = torch.arange(self.K, device=self.device, dtype=torch.float64)
        l = torch.arange(A.shape[1], device=self.device, dtype=torch.float64)
        F = torch.exp(-2j * np.pi * torch.outer(k, l) / self.K) * torch.as_tensor(self.ramp(), device=self.device)[:, None]
        return (F @ Rgg.to(torch.complex128) @ F.conj().T).cpu().numpy()

    def _long_term(self, method: str, aligned: bool) -> np.ndarray:
        key = (method, bool(aligned))
        if key not in self._R:
            if method == "LMMSE-Fast":
                self._R[key] = self.long_term_correlation()
            else:
                self._R[key] = self.host.pdp_correlation(self.fading, uniform=(method == "LMMSE-UniPDP"), aligned=aligned)
        return self._R[key]

    def receive(self, rx: torch.Tensor, bits: torch.Tensor, method: str, snr_db, H_true: Optional[torch.Tensor] = None,
                aligned: bool = True, want_bits: bool = False):
        """rx float32 [n, S, K+CP, 2] (device), bits int32 [n, D, nbits] (the transmitted labels), H_true complex
        [n, K] / [n, S, K] or float [.., 2] (the generator's channel response; Perfect / LMMSE) ->
        (bit errors, bits counted[, detected bits int32 [n, D, nbits]]).  Same arithmetic as ClassicalReceiver.receive."""
        if method not in GPU_METHODS:
            raise NotImplementedError("%s is not an estimator of this path (%s)" % (method, ", ".join(GPU_METHODS)))
        if method in ("LS-CP", "ALMMSE-CP") and not aligned:
            raise ValueError("the cyclic-prefix methods model a causal channel: use aligned=True")
        lib, s = self.lib, self._stream()
        n, S, K, SK, P, D = rx.shape[0], self.S, self.K, self.S * self.K, self.P, self.D
        rx = rx.contiguous()
        bits = bits.to(torch.int32).contiguous()
        f32 = dict(dtype=torch.float32, device=self.device)
        adv = self.advance if aligned else 0
        if adv > self.CP:
            # the delayed FFT window would start before the first symbol's cyclic prefix (the host receiver zero-fills
            # there); no profile of radio.py has a centre tap that far out
            raise NotImplementedError("FFT-window advance %d exceeds the cyclic prefix %d: use dl_ofdm_amd.benchmark" % (adv, self.CP))
        # FFT window: the frame delayed by `adv` samples = the K samples from CP - adv of every symbol row (the first
        # symbol's window then starts adv samples inside its own prefix; ClassicalReceiver shifts the flat frame instead,
        # which reads the same samples for every symbol: CP >= adv)
        Y = torch.empty(n * S, 2 * K, **f32)
        xwin = rx.view(n * S, 2 * self.n_sc)
        off = 2 * (self.CP - adv)
        check(lib.dccn_dense_fwd_ld(xwin.data_ptr() + 4 * off, 2 * self.n_sc, self._dft_matrix(aligned and adv > 0).data_ptr(),
                                    None, Y.data_ptr(), n * S, 2 * K, 2 * K, s), "dccn_dense_fwd_ld")
        pv = complex(self.host.pilot_value)
        gp = torch.empty(2, n, P, **f32)
        check(lib.dccn_classical_pilot_ls(Y.data_ptr(), self.pil.data_ptr(), gp.data_ptr(), n, SK, P, pv.real, pv.imag, s),
              "dccn_classical_pilot_ls")
        W = self.w_linear if method == "LS-Linear" else self.w_spline
        Gls = torch.empty(2 * n, SK, **f32)
        check(lib.dccn_dense_fwd(gp.data_ptr(), W.data_ptr(), None, Gls.data_ptr(), 2 * n, P, SK, s), "dccn_dense_fwd")
        sigma2 = float(np.mean(10.0 ** (-np.asarray(snr_db, dtype=np.float64) / 10.0)))
        c = K * sigma2 / abs(pv) ** 2                                    # LS error variance at a pilot
        H = None
        if method in ("Perfect", "LMMSE"):
            if H_true is None:
                raise ValueError("%s needs the channel's true response" % method)
            Hc = torch.view_as_real(H_true) if torch.is_complex(H_true) else H_true
            Hc = Hc.to(torch.float32)
            if Hc.dim() == 3:                                            # static channel: one response per frame
                Hc = Hc[:, None, :, :].expand(n, S, K, 2)
            if self.advance > 0:                                          # (aligned or not: the host does the same)
                r = self.ramp()                                           # phase reference = the centre tap
                rr = torch.as_tensor(np.stack([r.real, r.imag], -1).astype(np.float32), device=self.device)
                Hc = torch.stack([Hc[..., 0] * rr[:, 0] - Hc[..., 1] * rr[:, 1],
                                  Hc[..., 0] * rr[:, 1] + Hc[..., 1] * rr[:, 0]], -1)
            H = Hc.contiguous().view(n, SK, 2)
        need_gain = method in ("Perfect", "LMMSE", "LMMSE-Fast", "LMMSE-UniPDP", "LMMSE-ExpPDP")
        if need_gain:
            want_sums = method.startswith("LMMSE-")
            check(lib.dccn_classical_gain(Y.data_ptr(), None if H is None else H.data_ptr(), Gls.data_ptr(), self.pil.data_ptr(),
                                          n, SK, P, pv.real, pv.imag, self._sums.data_ptr() if want_sums else None,
                                          self.ws.data_ptr(), self.nws, s), "dccn_classical_gain")
        g_row, g_mod = SK, 0
        if method in ("LS-CP", "ALMMSE-CP"):
            # dev/m/cpenhanced.m (OFDM_Benchmark_dev.m:409,421): starting estimate G0 = the frame-averaged LS-spline estimate
            # (:417-423) or the ALMMSE estimate (:424-436); per symbol the received prefix re-estimates the L channel taps and
            # the symbol is the least-squares solution of [diag(G0); H_L Q_cp] X = [Y; y_cp - ISI]
            if method == "LS-CP":
                V = torch.empty(n, K, 2, **f32)
                check(lib.dccn_classical_estimate(Gls.data_ptr(), None, V.data_ptr(), n, S, K, 4, float(c), self.ws.data_ptr(),
                                                  self.nws, s), "dccn_classical_estimate")
                G0 = torch.view_as_complex(V)[:, None, :].expand(n, S, K)
            else:
                Ga = torch.empty(n, SK, 2, **f32)
                check(lib.dccn_classical_estimate(Gls.data_ptr(), None, Ga.data_ptr(), n, S, K, 2, float(c), self.ws.data_ptr(),
                                                  self.nws, s), "dccn_classical_estimate")
                G0 = torch.view_as_complex(Ga).view(n, S, K)
            X = self._cp_enhanced(torch.view_as_complex(Y.view(n, S, K, 2)), G0, rx, adv)
            Xr = torch.view_as_real(X.to(torch.complex64)).contiguous().view(n, SK, 2)
            G = torch.zeros(n, SK, 2, **f32)
            G[..., 0] = 1.0                                               # x = X / 1 at the data cells
            Y = Xr
        elif method in ("Perfect", "LMMSE", "ALMMSE", "LS-Spline", "LS-Linear"):
            mode = {"Perfect": 3, "LMMSE": 1, "ALMMSE": 2}.get(method, 0)
            G = torch.empty(n, SK, 2, **f32)
            check(lib.dccn_classical_estimate(Gls.data_ptr(), None if H is None else H.data_ptr(), G.data_ptr(), n, S, K, mode,
                                              float(c), self.ws.data_ptr(), self.nws, s), "dccn_classical_estimate")
        else:
            # long-term LMMSE: W = R (R + c I)^-1 with R = gain * R_long; gain from the batch's mean |G_ls|^2 (one scalar
            # read back: the 64 x 64 inverse is host work, like the constant matrices above)
            mean_g2 = float(self._sums.cpu()[3]) / (n * SK)
            R_long = self._long_term(method, aligned)
            if method == "LMMSE-Fast":
                gain = max(mean_g2 - c, 1e-12) / float(np.real(np.trace(R_long)) / K)
            else:
                gain = max(mean_g2 - c, 1e-12)
            R = gain * R_long
            Wm = R @ np.linalg.inv(R + c * np.eye(K))                    # G = Wm . g  ->  rows: g . Wm^T
            Wt = torch.as_tensor(_cexpand(Wm.T).astype(np.float32), device=self.device)
            if method == "LMMSE-Fast":
                Gi = torch.empty(n, SK, 2, **f32)
                check(lib.dccn_classical_estimate(Gls.data_ptr(), None, Gi.data_ptr(), n, S, K, 0, float(c), self.ws.data_ptr(),
                                                  self.nws, s), "dccn_classical_estimate")
                G = torch.empty(n * S, 2 * K, **f32)
                check(lib.dccn_dense_fwd(Gi.data_ptr(), Wt.data_ptr(), None, G.data_ptr(), n * S, 2 * K, 2 * K, s), "dccn_dense_fwd")
            else:
                V = torch.empty(n, K, 2, **f32)
                check(lib.dccn_classical_estimate(Gls.data_ptr(), None, V.data_ptr(), n, S, K, 4, float(c), self.ws.data_ptr(),
                                                  self.nws, s), "dccn_classical_estimate")
                G = torch.empty(n, 2 * K, **f32)
                check(lib.dccn_dense_fwd(V.data_ptr(), Wt.data_ptr(), None, G.data_ptr(), n, 2 * K, 2 * K, s), "dccn_dense_fwd")
                g_row, g_mod = K, K
        det = torch.empty(n, D, self.nbits, dtype=torch.int32, device=self.device) if want_bits else None
        check(lib.dccn_classical_detect(Y.data_ptr(), G.data_ptr(), self.dat.data_ptr(), self.table.data_ptr(),
                                        self.labels.data_ptr(), bits.data_ptr(), None if det is None else det.data_ptr(),
                                        self._err.data_ptr(), n, SK, D, int(self.table.shape[0]), self.nbits, g_row, g_mod,
                                        self.ws.data_ptr(), self.nws, s), "dccn_classical_detect")
        errors = int(self._err.item())
        total = n * D * self.nbits
        return (errors, total, det) if want_bits else (errors, total)


    def _cp_enhanced(self, Yc: torch.Tensor, G0: torch.Tensor, rx: torch.Tensor, adv: int) -> torch.Tensor:
        """cpenhanced.m batched over frames AND symbols (the previous symbol enters only through its LS estimate): complex128
        on the device.  Yc / G0 [n, S, K] in the window's phase reference; rx [n, S, K+CP, 2] as received."""
        n, S, K, L = Yc.shape[0], self.S, self.K, self.CP
        dev = self.device
        c128 = torch.complex128
        back = torch.as_tensor(np.conj(self.ramp()), device=dev)[None, None, :]       # to the causal response's own phase
        Y = Yc.to(c128) * back
        G = G0.to(c128) * back
        T = S * (K + L)
        flat = torch.view_as_complex(rx.contiguous().view(n, T, 2)).to(c128)
        if adv > 0:                                                                    # the aligned (delayed) frame
            flat = torch.cat([torch.zeros(n, adv, dtype=c128, device=dev), flat[:, :T - adv]], dim=1)
        y_time = flat.view(n, S, K + L)
        k = torch.arange(K, device=dev, dtype=torch.float64)
        Qinv = torch.exp(2j * np.pi * torch.outer(k, k) / K) / K                        # conj(dftmtx(N)) / N
        Q_cp = Qinv[K - L:K, :]
        x_ls = torch.fft.ifft(Y / G, dim=-1)
        cur = x_ls[:, :, K - L:K]                                                       # [n, S, L]
        prev = torch.cat([torch.zeros(n, 1, L, dtype=c128, device=dev), cur[:, :-1]], dim=1)
        r, cidx = torch.arange(L, device=dev)[:, None], torch.arange(L, device=dev)[None, :]
        idx = (L + r - cidx).reshape(-1)                                                # circshift_comb as a gather

        def conv_matrix(upper, lower):
            seq = torch.cat([upper, lower], dim=-1)
            return seq[..., idx].reshape(*seq.shape[:-1], L, L)

        def solve_ls(A, b):
            """inv(A) b, pinv where rcond(A) < 1e-10 (cpenhanced.m:41-45, 49-53)"""
            sv = torch.linalg.svdvals(A)
            ok = (sv[..., -1] / sv[..., 0].clamp_min(1e-300)) >= 1e-10
            x = torch.empty_like(b)
            if ok.any():
                x[ok] = torch.linalg.solve(A[ok], b[ok].unsqueeze(-1)).squeeze(-1)
            if (~ok).any():
                x[~ok] = (torch.linalg.pinv(A[~ok]) @ b[~ok].unsqueeze(-1)).squeeze(-1)
            return x
        Tm = conv_matrix(prev, cur)                                                     # [n, S, L, L]
        ycp = y_time[:, :, :L]
        Th = Tm.conj().transpose(-1, -2)
        h = solve_ls(Th @ Tm, (Th @ ycp.unsqueeze(-1)).squeeze(-1))
        H_L = conv_matrix(torch.zeros_like(h), h)
        Tu = Tm * ((cidx - r) >= 2)                                                     # the script's (one-diagonal-short) ISI part
        HQ = H_L @ Q_cp                                                                 # [n, S, L, K]
        BhB = torch.diag_embed((G.conj() * G)) + HQ.conj().transpose(-1, -2) @ HQ
        rhs2 = ycp - (Tu @ h.unsqueeze(-1)).squeeze(-1)
        BhC = G.conj() * Y + (HQ.conj().transpose(-1, -2) @ rhs2.unsqueeze(-1)).squeeze(-1)
        return solve_ls(BhB, BhC)


class CurvePointsGPU:
    """benchmark.CurvePoints on the device: SNR point i of one estimator's curve, frames from the device-side generator
    (Philox streams keyed by (seed, i)), receiver = ClassicalReceiverGPU.  ``point`` returns (bit errors, bits)."""

    def __init__(self, FLAGS, method: str, n_frames: int = 2000, seed: int = 1, device="cuda", aligned: bool = True):
        from .datagen import DeviceDataGen
        self.FLAGS, self.method, self.n_frames, self.seed, self.aligned = FLAGS, method, int(n_frames), int(seed), aligned
        self.rx = ClassicalReceiverGPU(FLAGS, device=device)
        self.gen = DeviceDataGen(FLAGS, self.rx.o, device=device, seed=seed)
        self.gen.want_noise_power = False

    def point(self, i: int, snr: float):
        self.gen.seed, self.gen.offset = self.seed + 7919 * i, 0
        need_h = self.method in ("Perfect", "LMMSE")
        out = self.gen.make_batch(self.n_frames, float(snr), want_H=need_h)
        x, bits = out[0], out[1]
        H = out[3] if need_h else None
        return self.rx.receive(x, bits, self.method, snr, H_true=H, aligned=self.aligned)
