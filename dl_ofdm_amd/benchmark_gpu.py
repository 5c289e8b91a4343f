"""Classical pilot-aided receivers on the GPU: the estimator family of
dev/m/OFDM_Benchmark_dev.m:339-456 -- the curves the paper draws DCCN against -- as a launch sequence over libdccn
(include/dccn.h "classical pilot-aided receivers"), fed by device-resident frames (the device-side generator's, or any
[n, S, K+CP, 2] batch).  The NumPy restatement in :mod:`dl_ofdm_amd.benchmark` is the oracle of this path
(tests/test_gpu_benchmark.py) and keeps the two cyclic-prefix methods (LS-CP / ALMMSE-CP: a per-symbol joint
least-squares solve) that are not built here.

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

GPU_METHODS = ("Perfect", "LS-Spline", "LS-Linear", "LMMSE", "ALMMSE", "LMMSE-Fast", "LMMSE-UniPDP", "LMMSE-ExpPDP")
_MODE = {"LS": 0, "LMMSE": 1, "ALMMSE": 2, "Perfect": 3, "FrameMean": 4}


def _cexpand(Mc: np.ndarray) -> np.ndarray:
    """complex [a, b] (acting as y = x . M on interleaved-IQ rows) -> real [2a, 2b]"""
    a, b = Mc.shape
    R = np.zeros((2 * a, 2 * b), dtype=np.float64)
    R[0::2, 0::2], R[0::2, 1::2] = Mc.real, Mc.imag
    R[1::2, 0::2], R[1::2, 1::2] = -Mc.imag, Mc.real
    return R


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
        self.w_spline = torch.as_tensor(np.ascontiguousarray(h.W_spline.T, dtype=np.float32), **f32)      # [P, S K]
        self.w_linear = torch.as_tensor(np.ascontiguousarray(h.W_linear.T, dtype=np.float32), **f32)
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
                F = F * self.host.ramp(self.advance)[None, :]
            self._dft[key] = torch.as_tensor(_cexpand(F).astype(np.float32), device=self.device)
        return self._dft[key]

    def _long_term(self, method: str, aligned: bool) -> np.ndarray:
        key = (method, bool(aligned))
        if key not in self._R:
            if method == "LMMSE-Fast":
                self._R[key] = self.host.long_term_correlation(self.fading, aligned)
            else:
                self._R[key] = self.host.pdp_correlation(self.fading, uniform=(method == "LMMSE-UniPDP"), aligned=aligned)
        return self._R[key]

    def receive(self, rx: torch.Tensor, bits: torch.Tensor, method: str, snr_db, H_true: Optional[torch.Tensor] = None,
                aligned: bool = True, want_bits: bool = False):
        """rx float32 [n, S, K+CP, 2] (device), bits int32 [n, D, nbits] (the transmitted labels), H_true complex
        [n, K] / [n, S, K] or float [.., 2] (the generator's channel response; Perfect / LMMSE) ->
        (bit errors, bits counted[, detected bits int32 [n, D, nbits]]).  Same arithmetic as ClassicalReceiver.receive."""
        if method not in GPU_METHODS:
            raise NotImplementedError("%s is a host-only estimator (dl_ofdm_amd.benchmark)" % method)
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
                r = self.host.ramp(self.advance)                          # phase reference = the centre tap
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
        if method in ("Perfect", "LMMSE", "ALMMSE", "LS-Spline", "LS-Linear"):
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
