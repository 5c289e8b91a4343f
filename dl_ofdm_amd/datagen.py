"""Device-side input generator: label bits -> OFDM frames -> static Rayleigh / flat /
AWGN channel -> noise, produced on the GPU straight into the receiver engine's resident buffers.

The reference generates every epoch on the host with NumPy (dev/py/ofdmreceiver_np.py:220-229:
``bit_source`` -> ``ofdm_tx_frame_np`` -> ``rayleigh_chan_lte.run`` -> ``AWGN_channel_np``); that host path is
kept in ofdm.py / radio.py (bit-pinned to the reference) and is ~2000x slower than the fused GPU step.
``DeviceDataGen.make_batch`` is the same chain as libdccn kernels (include/dccn.h: "device-side input
generator"): same constellation tables, grid layout, ifft + cyclic prefix (as one MFMA GEMM), tap model,
'same' FIR, power normalisation and noise scaling; its random streams are Philox4x32-10 instead of NumPy's
Mersenne Twister, so agreement with the host path is exact for the deterministic stages (given the same
bits / tap draws / noise draws -- tests/test_gpu_datagen.py) and statistical for the draws themselves.

Scope: every channel of ``rayleigh_chan_lte``: the single-profile ones ('AWGN', 'Flat', 'EPA', 'EVA', 'ETU',
'Custom'), static or mobile (Jakes Doppler, radio.py:376-407: per-symbol taps, per-symbol FIR with n_taps samples
of history), and the frame-interleaved 'mixRayleigh' / 'mixAll' (radio.py:438-470: profile = frame index modulo 4
or 5, Doppler on every 3rd / 4th frame when ``mix``) -- those run one launch pair per (profile, static|Doppler)
group over a frame-index list.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib, ofdm, radio
from ._lib import check


class DeviceDataGen:
    def __init__(self, FLAGS, ofdmobj=None, device="cuda", seed: int = 1, mobile: bool = False, mix: bool = False):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.DccnError("DeviceDataGen needs a CUDA (ROCm) device; use ofdm.py/radio.py on the host")
        self.FLAGS, self.o = FLAGS, ofdmobj or ofdm.ofdm_tx(FLAGS)
        chan = FLAGS.channel.lower()
        self.mixed = chan in ("mixrayleigh", "mixall")
        self.mix = bool(mix)
        o = self.o
        self.S, self.K, self.CP, self.D, self.nbits = o.nSymbol, o.K, o.CP, o.frame_size, int(FLAGS.nbits)
        self.n_sc, self.T = o.K + o.CP, o.nSymbol * (o.K + o.CP)
        self.seed, self.offset = int(seed) & 0xFFFFFFFFFFFFFFFF, 0
        self.want_noise_power = True
        # extension (not in the reference): receiver-side timing alignment.  radio.py filters with np.convolve(.., 'same'), i.e.
        # the L-tap response is centred and the received frame arrives (L-1)//2 samples EARLY: the FFT window of the reference
        # then catches the head of the next symbol (inter-symbol interference the cyclic prefix cannot absorb; the high-SNR
        # floor of profiles/r02_ber_floor).  align_window delays every generated frame by that advance (single-profile
        # channels), which is what a receiver with timing synchronisation sees.
        self.align_window = bool(getattr(FLAGS, "align_window", False))
        dev = self.device
        cell = np.full(self.S * self.K, -2, dtype=np.int32)
        cell[o.dataSc] = np.arange(self.D, dtype=np.int32)
        cell[o.pilotSc] = -1
        self.cell_map = torch.from_numpy(cell).to(dev)
        tab = ofdm.const_map(self.nbits)
        self.const_tab = torch.from_numpy(np.stack([tab.real, tab.imag], -1).astype(np.float32)).to(dev)
        self.pilot = complex(o.pilotValue)
        self.idft = torch.from_numpy(self.idft_cp_matrix(self.K, self.CP)).to(dev)
        self.identity = chan == "awgn"
        if self.mixed:                                               # radio.py:438-446 profile rotation
            names = ("flat", "etu", "eva", "epa") if chan == "mixrayleigh" else ("awgn", "flat", "etu", "eva", "epa")
            alphas = radio._alpha_matrices()
            self.period = 3 if chan == "mixrayleigh" else 4
            self.profiles = []
            for nm in names:
                pr = radio._Profile(nm, bool(mobile), alphas)
                self.profiles.append(dict(identity=(nm == "awgn"), Fd=float(pr.Fd), n_taps=int(pr.n_taps),
                                          L=int(pr.alpha.shape[1]),
                                          coeff=torch.from_numpy(np.asarray(pr.ch_coeff, dtype=np.float32)).to(dev),
                                          alpha=torch.from_numpy(np.ascontiguousarray(pr.alpha, dtype=np.float32)).to(dev)))
            self._groups = {}
        prof = radio._Profile("flat" if self.mixed else chan, bool(mobile), radio._alpha_matrices())
        self.Fd = float(prof.Fd)
        self.doppler = (not self.identity) and self.Fd > 0.1          # radio.py: `doppler = prof.Fd > 0.1`
        self.t_sym = float(self.n_sc) / float(o.Fs)
        self.n_taps, self.L = int(prof.n_taps), int(prof.alpha.shape[1])
        self.coeff = torch.from_numpy(np.asarray(prof.ch_coeff, dtype=np.float32)).to(dev)
        self.alpha = torch.from_numpy(np.ascontiguousarray(prof.alpha, dtype=np.float32)).to(dev)
        if self.identity:
            self.L = 1
        self._ws = {}

    @staticmethod
    def idft_cp_matrix(K: int, CP: int) -> np.ndarray:
        """real [2K, 2(K+CP)] form of ``np.fft.ifft`` followed by the cyclic-prefix copy (ofdm.py:357-362):
        row (k, iq) -> column (t', iq'), t = (t' - CP) mod K, W = exp(+2 pi i k t / K) / K."""
        k = np.arange(K)[:, None]
        t = (np.arange(K + CP)[None, :] - CP) % K
        w = np.exp(2j * np.pi * k * t / K) / K
        m = np.empty((K, 2, K + CP, 2), dtype=np.float64)
        m[:, 0, :, 0], m[:, 0, :, 1] = w.real, w.imag
        m[:, 1, :, 0], m[:, 1, :, 1] = -w.imag, w.real
        return m.reshape(2 * K, 2 * (K + CP)).astype(np.float32)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, n: int):
        if n not in self._ws:
            f32 = dict(dtype=torch.float32, device=self.device)
            nws = max(self.lib.dccn_channel_doppler_awgn_workspace_size(n, self.T, self.L, self.S),
                      self.lib.dccn_channel_groups_awgn_workspace_size(n, self.T, self.S) if self.mixed else 0)
            self._ws[n] = dict(grid=torch.empty(n, self.S, self.K, 2, **f32), tx=torch.empty(n, self.S, self.n_sc, 2, **f32),
                               ws=torch.empty(nws, dtype=torch.uint8, device=self.device), nws=nws,
                               snr=torch.empty(n, **f32), npow=torch.zeros(1, **f32))
        return self._ws[n]

    @staticmethod
    def _p(t):
        return None if t is None else t.data_ptr()

    def transmit(self, n_frames: int, bits: Optional[torch.Tensor] = None, out_bits: Optional[torch.Tensor] = None,
                 offset: Optional[int] = None):
        """(tx [n,S,n_sc,2], bits int32 [n,D,nbits]); ``bits`` given: modulate those (no draw)."""
        w = self._workspace(n_frames)
        off = self.offset if offset is None else int(offset)
        if bits is not None:
            bits = torch.as_tensor(bits).to(device=self.device, dtype=torch.int32).contiguous()
            out_bits = bits
        elif out_bits is None:
            out_bits = torch.empty(n_frames, self.D, self.nbits, dtype=torch.int32, device=self.device)
        check(self.lib.dccn_ofdm_tx_frames(self._p(bits), None if bits is not None else self._p(out_bits),
                                           self._p(self.cell_map), self._p(self.const_tab), self.pilot.real,
                                           self.pilot.imag, self._p(self.idft), self._p(w["grid"]), self._p(w["tx"]),
                                           n_frames, self.S, self.K, self.CP, self.D, self.nbits, self.seed, off,
                                           self._stream()), "dccn_ofdm_tx_frames")
        return w["tx"], out_bits

    def channel(self, tx: torch.Tensor, snr_db, out_x: Optional[torch.Tensor] = None, taps: Optional[torch.Tensor] = None,
                noise: Optional[torch.Tensor] = None, want_H: bool = False, offset: Optional[int] = None,
                out_H: Optional[torch.Tensor] = None, out_npow: Optional[torch.Tensor] = None):
        """fading + AWGN on [n,S,n_sc,2] frames; ``taps`` / ``noise``: external draws instead of the Philox streams
        (static: standard normals [n,n_taps,2]; mobile: uniform phases [n,2,48,n_taps]; noise: normals [n,T,2]).
        H: complex [n,K] (static) or [n,S,K] (mobile)."""
        n = tx.shape[0]
        w = self._workspace(n)
        off = self.offset if offset is None else int(offset)
        snr_t = w["snr"]
        if (isinstance(snr_db, torch.Tensor) and snr_db.device == self.device and snr_db.dtype == torch.float32 and
                snr_db.is_contiguous() and snr_db.numel() == n):
            snr_t = snr_db              # per-frame SNRs already on the device (a row of the epoch's table): no copy, no launch
        elif isinstance(snr_db, torch.Tensor):
            w["snr"].copy_(snr_db.reshape(-1).to(torch.float32))
            w["snr_scalar"] = None
        elif np.isscalar(snr_db):
            if w.get("snr_scalar") != float(snr_db):           # a training run keeps one SNR: fill once, not per batch
                w["snr"].fill_(float(snr_db))
                w["snr_scalar"] = float(snr_db)
        else:
            w["snr"].copy_(torch.as_tensor(np.asarray(snr_db, dtype=np.float32).reshape(-1)))
            w["snr_scalar"] = None
        if out_x is None:
            out_x = torch.empty(n, self.S, self.n_sc, 2, dtype=torch.float32, device=self.device)
        hshape = (n, self.S, self.K, 2) if (self.doppler or self.mixed) else (n, self.K, 2)
        if out_H is not None:
            if tuple(out_H.shape) != hshape or out_H.dtype != torch.float32 or not out_H.is_contiguous():
                raise ValueError("out_H must be a contiguous float32 tensor of shape %s" % (hshape,))
            H, want_H = out_H, True
        else:
            H = torch.empty(*hshape, dtype=torch.float32, device=self.device) if want_H else None
        if taps is not None and not self.mixed:
            taps = torch.as_tensor(taps, dtype=torch.float32).to(self.device).contiguous()
        if noise is not None:
            noise = torch.as_tensor(noise, dtype=torch.float32).to(self.device).contiguous()
        # the noise-power monitor (`noise_power:0` of the reference's log line) costs one more reduction launch per batch:
        # sweeps and benchmarks that never read it switch it off (want_noise_power = False -> None is returned)
        npw_t = (out_npow if out_npow is not None else w["npow"]) if self.want_noise_power else None
        npw = self._p(npw_t)
        if self.mixed:
            # taps: (normals [n,16,2] for the static frames, phases [n,2,48,16] for the Doppler frames), or None
            tn, th = (None, None) if taps is None else taps
            if tn is not None:
                tn = torch.as_tensor(tn, dtype=torch.float32).to(self.device).contiguous()
            if th is not None:
                th = torch.as_tensor(th, dtype=torch.float32).to(self.device).contiguous()
            arr, keep = self._frame_groups(n)
            check(self.lib.dccn_channel_groups_awgn(self._p(tx), arr, len(arr), self._p(tn), self._p(th), self.t_sym,
                                                    self.S, self.n_sc, self._p(snr_t), self._p(noise), self._p(out_x),
                                                    self._p(H), self.K, npw, n, self.seed, off,
                                                    self._p(w["ws"]), w["nws"], self._stream()), "dccn_channel_groups_awgn")
            return out_x, npw_t, (torch.view_as_complex(H) if want_H else None)
        if self.doppler:
            check(self.lib.dccn_channel_doppler_awgn(self._p(tx), self._p(taps), self._p(self.coeff), self._p(self.alpha),
                                                     self.n_taps, self.L, self.Fd, self.t_sym, self.S, self.n_sc,
                                                     self._p(snr_t), self._p(noise), self._p(out_x), self._p(H),
                                                     self.K, npw, n, self.seed, off, self._p(w["ws"]),
                                                     w["nws"], self._stream()), "dccn_channel_doppler_awgn")
            self._align(out_x)
            return out_x, npw_t, (torch.view_as_complex(H) if want_H else None)
        check(self.lib.dccn_channel_awgn(self._p(tx), self._p(taps), self._p(self.coeff), self._p(self.alpha),
                                         self.n_taps, self.L, 1 if self.identity else 0, self._p(snr_t),
                                         self._p(noise), self._p(out_x), self._p(H), self.K, npw, n,
                                         self.T, self.seed, off, self._p(w["ws"]), w["nws"], self._stream()),
              "dccn_channel_awgn")
        self._align(out_x)
        return out_x, npw_t, (torch.view_as_complex(H) if want_H else None)

    def _align(self, out_x: torch.Tensor):
        adv = (self.L - 1) // 2 if (self.align_window and not self.mixed and not self.identity) else 0
        if adv > 0:
            flat = out_x.view(out_x.shape[0], self.T, 2)
            out_x.copy_(torch.roll(flat, adv, dims=1).view_as(out_x))

    def frame_plan(self, n: int):
        """per frame (profile index, doppler?) exactly as radio.py:438-452 decides it"""
        n_prof = len(self.profiles)
        plan = []
        for fr in range(n):
            pi = fr % n_prof
            pr = self.profiles[pi]
            dop = (fr % self.period == 0) and pr["Fd"] > 0.1 and self.mix and not pr["identity"]
            plan.append((pi, bool(dop)))
        return plan

    def _frame_groups(self, n: int):
        if n not in self._groups:
            from ._lib import ChannelGroup
            plan = self.frame_plan(n)
            keys = sorted(set(plan))
            arr = (ChannelGroup * len(keys))()
            keep = []
            for i, (pi, dop) in enumerate(keys):
                ids = torch.as_tensor(np.asarray([f for f, k in enumerate(plan) if k == (pi, dop)], dtype=np.int32)).to(self.device)
                pr = self.profiles[pi]
                keep.append(ids)
                arr[i] = ChannelGroup(ids.data_ptr(), int(ids.numel()), pr["coeff"].data_ptr(), pr["alpha"].data_ptr(),
                                      pr["n_taps"], pr["L"], 1 if pr["identity"] else 0, pr["Fd"] if dop else 0.0)
            self._groups[n] = (arr, keep)
        return self._groups[n]

    def make_batch(self, n_frames: int, snr_db, out_x: Optional[torch.Tensor] = None,
                   out_bits: Optional[torch.Tensor] = None, want_H: bool = False):
        """receiver.make_batch on the device: (x float32 [n,S,n_sc,2], bits int32 [n,D,nbits], noise power[, H]);
        advances the batch offset so consecutive calls draw fresh data."""
        tx, bits = self.transmit(n_frames, out_bits=out_bits)
        x, npow, H = self.channel(tx, snr_db, out_x=out_x, want_H=want_H)
        self.offset = (self.offset + 1) & 0xFFFFFFFF
        return (x, bits, npow, H) if want_H else (x, bits, npow)


class FusedStaticGen:
    """``DeviceDataGen.make_batch`` of a static single-profile channel as ONE launch (include/dccn.h dccn_gen_static,
    csrc/datagen.h gen_static_frames_kernel): bits -> grid -> ifft + CP -> taps -> 'same' FIR -> (y, frame-scaled noise, power
    partials).  The receiver's input x = y / sqrt(mean |y|^2) + noise is formed by the consumer: ``RxEngine.
    train_step_generated`` hands the descriptor to ``dccn_rx_train_step``, which issues the generator launch itself and reads
    (y, noise, partials) as the virtual input of its pipelined normalisation -- one C call and five launches per generated-and-
    trained batch instead of two or three calls and eight or nine launches; ``make_batch`` materialises x (one more launch)
    where a buffer is wanted.  Same Philox streams, draws and batch offsets as ``DeviceDataGen`` (it advances ``gen.offset``)."""

    def __init__(self, gen: DeviceDataGen, n_frames: int, snr_db, want_noise_power: bool = False, arena=None):
        if not self.supported(gen):
            raise _lib.DccnError("FusedStaticGen: static N = 64 channels only (one profile, or the frame-interleaved profiles of "
                                 "mixRayleigh / mixAll without Doppler frames)")
        from . import arena as A
        self.gen, self.n = gen, int(n_frames)
        self.arena = arena
        dev, f32 = gen.device, dict(dtype=torch.float32, device=gen.device)
        self.snr = A.empty(arena, self.n, **f32)
        self.set_snr(snr_db)
        self.y = A.empty(arena, self.n, gen.S, gen.n_sc, 2, **f32)
        self.noise = A.empty(arena, self.n, gen.S, gen.n_sc, 2, **f32)
        npart = int(gen.lib.dccn_gen_static_partials(self.n))
        self.ppart = A.zeros(arena, npart, dtype=torch.float64, device=dev)
        self.npart = A.zeros(arena, npart, dtype=torch.float64, device=dev) if want_noise_power else None
        self.npow = A.zeros(arena, 2, 1, **f32) if want_noise_power else None       # one slot per label slot
        p = DeviceDataGen._p
        # (chain groups: the generator's tables travel with the chain -- copies inside its arena, the constellation set aside
        # for 16-QAM)
        self._tabs = [A.place(arena, gen.cell_map), A.place(arena, gen.const_tab, reserve=32), A.place(arena, gen.idft),
                      A.place(arena, gen.coeff), A.place(arena, gen.alpha)]
        self.desc = _lib.GenStatic(0, p(self._tabs[0]), p(self._tabs[1]), float(gen.pilot.real), float(gen.pilot.imag),
                                   p(self._tabs[2]), p(self._tabs[3]), p(self._tabs[4]), gen.n_taps, gen.L, 1 if gen.identity else 0,
                                   p(self.snr), p(self.y), p(self.noise), p(self.ppart), p(self.npart), None, None,
                                   self.n, gen.S, gen.K, gen.CP, gen.D, gen.nbits, gen.seed, 0)
        if gen.mixed:
            # radio.py:438-452: frame f runs profile f % n_profiles; the launch-per-stage path draws 16 tap slots per frame
            self._ptabs = [(A.place(arena, pr["coeff"]), A.place(arena, pr["alpha"])) for pr in gen.profiles]
            self._profiles = (_lib.GenProfile * len(gen.profiles))(*[
                _lib.GenProfile(p(tc), p(ta), pr["n_taps"], pr["L"], 1 if pr["identity"] else 0, 0)
                for pr, (tc, ta) in zip(gen.profiles, self._ptabs)])
            self.desc.n_profiles, self.desc.tap_stride = len(gen.profiles), 16
            self.desc.profiles = C.addressof(self._profiles)

    @staticmethod
    def supported(gen: DeviceDataGen, eng=None) -> bool:
        """``eng``: the training engine whose steps would take this generator as ``gen_next`` (train_step_generated): its batch
        must be one the pipelined normalisation can form from the generator's output (dccn_rx_gen_next_supported: the batch grows
        with falling BER in receiver.train and leaves that range beyond 1536 frames)"""
        if eng is not None and not bool(gen.lib.dccn_rx_gen_next_supported(C.byref(eng.shape))):
            return False
        if gen.doppler or gen.align_window or not bool(gen.lib.dccn_gen_static_supported(gen.S, gen.K, gen.CP)):
            return False
        if gen.mixed:                       # static frames only: no (profile, Doppler) pair in any frame plan
            return (not gen.mix) and len(gen.profiles) <= 6 and all(pr["L"] <= 64 and pr["n_taps"] <= 16 for pr in gen.profiles)
        return True

    def set_snr(self, snr_db):
        if np.isscalar(snr_db):
            self.snr.fill_(float(snr_db))
        else:
            self.snr.copy_(torch.as_tensor(np.asarray(snr_db, dtype=np.float32).reshape(-1)) if not isinstance(snr_db, torch.Tensor)
                           else snr_db.reshape(-1).to(torch.float32))

    def arm(self, out_bits: torch.Tensor, slot: int = 0, tx_out: Optional[torch.Tensor] = None,
            out_H: Optional[torch.Tensor] = None, snr: Optional[torch.Tensor] = None, into=None) -> "_lib.GenStatic":
        """the descriptor for the NEXT launch: labels go to ``out_bits``, the batch offset is the generator's current one (which
        is advanced: call once per batch).  ``out_H`` float32 [n, K, 2] or [n, S, K, 2]: the frequency response per frame (per
        symbol); ``snr``: a float32 device tensor of n per-frame SNRs to read instead of the generator's own copy."""
        # ``into``: a copy of the descriptor to arm instead of the generator's own (the one a step's buffers point to when the
        # step produces the batch itself: dccn_eq_buffers.gen_next_rides)
        g, d = self.gen, (self.desc if into is None else into)
        d.bits_out = out_bits.data_ptr()
        if out_H is not None:
            per = (self.n, g.S, g.K, 2)
            if tuple(out_H.shape) not in (per, (self.n, g.K, 2)) or out_H.dtype != torch.float32 or not out_H.is_contiguous():
                raise ValueError("out_H must be a contiguous float32 tensor of shape %s or %s" % (per, (self.n, g.K, 2)))
            d.H_out, d.h_rep = out_H.data_ptr(), (g.S if out_H.dim() == 4 else 1)
        else:
            d.H_out, d.h_rep = None, 0
        d.snr_db = self.snr.data_ptr() if snr is None else snr.data_ptr()
        d.offset = g.offset & 0xFFFFFFFF
        d.seed = g.seed
        d.noise_power_out = self.npow[slot & 1].data_ptr() if self.npow is not None else None
        d.tx_out = None if tx_out is None else tx_out.data_ptr()
        g.offset = (g.offset + 1) & 0xFFFFFFFF
        return d

    def make_batch(self, out_x: torch.Tensor, out_bits: torch.Tensor, slot: int = 0, tx_out: Optional[torch.Tensor] = None,
                   out_H: Optional[torch.Tensor] = None, snr: Optional[torch.Tensor] = None):
        """generate + materialise: (x, bits, noise power or None), two launches (the first batch of a pipelined loop, the
        equaliser's epoch loop, tests)"""
        d = self.arm(out_bits, slot, tx_out, out_H, snr)
        st = self.gen._stream()
        check(self.gen.lib.dccn_gen_static_frames(C.byref(d), st), "dccn_gen_static_frames")
        npw = self.npow[slot & 1] if self.npow is not None else None
        check(self.gen.lib.dccn_gen_static_apply(C.byref(d), out_x.data_ptr(), None if npw is None else npw.data_ptr(), st),
              "dccn_gen_static_apply")
        return out_x, out_bits, npw


class SideStreamFeeder:
    """The generator on its own HIP stream: batch i+1 is produced while the forward and backward launches of step i run.

    The pipelined receiver step reads ``eng.x`` only in its LAST launch (R0 of the next batch rides on the optimizer launch) and
    reads the labels of the batch it trains on from one of two slots, so the generation of batch i+1 into ``eng.x`` / the other
    slot depends on nothing step i's first three launches touch.  Two events order the rest: the step's x_next_ready event
    (include/dccn.h) makes the optimizer launch wait for the generator, and the generator waits for the previous step (whose
    optimizer launch read the batch that ``eng.x`` held).  The generator's four latency-bound launches (~35 us of a mostly
    idle chip) then cost the loop nothing but the join.

        feed = SideStreamFeeder(eng, lambda slot: gen.make_batch(n, snr, out_x=eng.x, out_bits=eng.label_slot(slot)))
        feed.first()                                   # batch 0 on the caller's stream + eng.prime()
        for i in range(steps):
            feed.next((i + 1) & 1)                     # batch i+1 on the side stream
            eng.train_step_pipelined(slot=i & 1, x_ready=feed.ready)
            feed.step_issued()
    """

    def __init__(self, eng, make):
        self.eng, self.make = eng, make
        self.stream = torch.cuda.Stream(device=eng.device)
        self.ready = torch.cuda.Event()
        self._step = torch.cuda.Event()
        self._have_step = False

    def rebind(self, make):
        """a new epoch on the same engine: new producer, same stream and events; the first next() waits for everything the
        main stream has issued (the end-of-epoch evaluation included)"""
        self.make = make
        self._have_step = False

    def first(self, slot: int = 0):
        self.make(slot)
        self.eng.prime()

    def next(self, slot: int):
        main = torch.cuda.current_stream(self.eng.device)
        if self._have_step:
            self.stream.wait_event(self._step)         # eng.x was read by the previous step's last launch
        else:
            self.stream.wait_stream(main)              # (first call: everything issued so far, incl. prime())
        with torch.cuda.stream(self.stream):
            out = self.make(slot)
            self.ready.record(self.stream)
        return out

    def step_issued(self):
        self._step.record(torch.cuda.current_stream(self.eng.device))
        self._have_step = True

    def drain(self):
        torch.cuda.current_stream(self.eng.device).wait_stream(self.stream)

