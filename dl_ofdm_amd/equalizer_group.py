"""Several equaliser transfer-learning chains trained in lock-step through ONE launch sequence.

The reference driver runs one (receiver -> equaliser) chain per modulation and per cp / longcp variant, each as an OS process
(dev/py/run_local_ofdm.py:61-118; the loop is dev/py/ofdmreceiver_np_mp.py:394-466).  On an MI355X such a chain is a sequence of
73-frame steps of ~21 dependent launches that keeps a few percent of the chip busy, and the host spends as long issuing a step
as the GPU spends running it.  Here G chains of the same shape share every launch (include/dccn.h "chain groups": the chain
index is a grid dimension): per group step ONE C call carries all G chains' batches -- the step, with the next batch's generator
riding on its bottleneck backward launch and the loop's monitors on its optimizer launch (three calls without those riders).  Every chain keeps its own seeds, draws, early stopping and best-model snapshot: the trained model of a chain
is bit for bit the one :func:`dl_ofdm_amd.receiver_mp.train` produces for it alone (tests/test_gpu_chain_groups.py).

    results = train_group([flags_bpsk, flags_qpsk, ...], [rx_params_bpsk, rx_params_qpsk, ...])
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib, ofdm
from . import receiver_mp as H
from ._lib import check
from .arena import ChainArena, check_same_layout


def _ptr_array(structs):
    """a C array of pointers to the given ctypes structures (kept alive by the caller)"""
    arr = (C.c_void_p * len(structs))()
    for i, s in enumerate(structs):
        arr[i] = C.addressof(s)
    return arr


class _Chain:
    """one chain's state: what :func:`receiver_mp._train_on_device` keeps in local variables"""

    def __init__(self, FLAGS, rx_params, device, arena_bytes: int):
        from .datagen import DeviceDataGen, FusedStaticGen
        from .equalizer import EqualizerTrainer, _FusedPlan
        self.F = FLAGS
        self.o = ofdm.ofdm_tx(FLAGS)
        self.arena = ChainArena(arena_bytes, device)
        np.random.seed(FLAGS.seed)                                          # as receiver_mp.train does (nothing draws from it here)
        self.tr = EqualizerTrainer(FLAGS, self.o, rx_params, device=device, seed=FLAGS.seed, arena=self.arena).pin_tuning()
        self.B = FLAGS.batch_size // FLAGS.nsymbol
        self.steps = (FLAGS.msg_length // FLAGS.nsymbol) // self.B
        self.gen = DeviceDataGen(FLAGS, self.o, device=self.tr.device, seed=FLAGS.seed, mobile=FLAGS.mobile, mix=FLAGS.mobile)
        if not FusedStaticGen.supported(self.gen):
            raise _lib.DccnError("chain groups need the fused static-channel generator (no Doppler frames)")
        self.pl = _FusedPlan(self.tr, self.B, arena=self.arena)
        self.tr._plans[self.B] = self.pl
        self.ev = self.tr.resident(FLAGS.eval_frames)                      # evaluation runs per chain, outside the group
        self.loop = H.DeviceEpochLoop(FLAGS, self.o, self.tr, self.gen, self.pl, self.steps, arena=self.arena)
        if not (self.loop.pipeline and self.loop.virt is not None):
            raise _lib.DccnError("chain groups need the pipelined loop with the virtual next batch (dccn_eq_norm_rides)")
        self.best = H.BestSnapshot(self.tr, os.path.join(FLAGS.save_dir, H.save_model_name(FLAGS)), FLAGS)
        self.loss_min, self.epoch_min, self.best_path, self.history = 100.0, 0, "", []
        self.epoch, self.done = 0, False
        self.rs = None
        lp = self.loop
        self.monitors = []
        for q in range(2):
            pl = lp.pls[q]
            npow = lp.npow[q] if self.gen.want_noise_power else None
            self.monitors.append(_lib.EqMonitor(pl.chest.data_ptr(), lp.H[q].data_ptr(), lp.per_symbol, pl.batch, FLAGS.nsymbol,
                                                self.o.K, pl.metrics_buf.data_ptr(), pl.tx_power.data_ptr(),
                                                None if npow is None else npow.data_ptr(), lp.acc.data_ptr(), None,
                                                lp.ws.data_ptr(), lp.nws))

    # the pieces of receiver_mp._train_on_device, per chain -------------------------------------------------------------------
    def begin_epoch(self):
        F = self.F
        self.rs = np.random.RandomState((F.seed + 1000003 * (self.epoch + 1)) & 0xFFFFFFFF)
        self.loop.begin_epoch(self.rs.choice(H.TRAIN_SNR_GRID, [self.steps, self.B], p=H.TRAIN_SNR_PROB))

    def end_epoch(self, verbose: bool):
        F, gen, ev, tr = self.F, self.gen, self.ev, self.tr
        a = self.loop.epoch_means()
        train_loss_epoch = float(a[0])
        snr = self.rs.choice(H.TRAIN_SNR_GRID, [F.eval_frames], p=H.TRAIN_SNR_PROB)           # ofdmreceiver_np_mp.py:438
        tx, _ = gen.transmit(F.eval_frames, out_bits=ev.bits)
        gen.channel(tx, snr, out_x=ev.x)
        gen.offset += 1
        ev.run(False)
        em = tr._metrics(ev.metrics_buf, ev.tx_power)
        self.history.append(dict(epoch=self.epoch, train_loss=train_loss_epoch, train_ber=float(a[1]), chan_rms=float(a[4]),
                                 test_loss=em["ce_mean"], test_ber=em["berlin"]))
        if verbose:
            print("[%s] Epoch: %d  Train Loss: %f  Tx Power: %f  Noise Power: %f  SNR MSE: %f | Test Loss: %f  Test BER: %.8f"
                  % (F.token, self.epoch, train_loss_epoch, a[2], a[3], a[4], em["ce_mean"], em["berlin"]))
        if train_loss_epoch < self.loss_min:
            self.epoch_min, self.loss_min = self.epoch, train_loss_epoch
            self.best.take()
        else:
            self.best.maybe_flush()
        if self.epoch - F.early_stop > self.epoch_min or self.epoch + 1 >= F.max_epoch_num:
            self.done = True
        self.epoch += 1

    def result(self) -> dict:
        return dict(history=self.history, best_path=self.best_path, trainer=self.tr)


def arena_bytes_for(FLAGS, lib=None) -> int:
    """a chain's arena: five parameter-sized arrays, the fused step's workspace, the folded receiver, the loop's buffers"""
    lib = lib or _lib.load()
    o = ofdm.ofdm_tx(FLAGS)
    B = FLAGS.batch_size // FLAGS.nsymbol
    shape = _lib.EqShape(B, FLAGS.nsymbol, o.K, o.CP, 1 if FLAGS.cp else 0, FLAGS.nfilter, o.frame_size, 4, o.pilot_size,
                         len(o.pilotCarriers))
    offs = (C.c_longlong * 21)()
    check(lib.dccn_eq_param_offsets(C.byref(shape), offs), "dccn_eq_param_offsets")
    steps = (FLAGS.msg_length // FLAGS.nsymbol) // B
    n = 5 * 4 * int(offs[20]) + int(lib.dccn_eq_workspace_size(C.byref(shape), 1)) + 4 * int(lib.dccn_eq_rx_folded_floats(C.byref(shape)))
    n += 4 * (2 * FLAGS.nsymbol * FLAGS.nfilter * 2 * o.frame_size + (o.K + o.CP) * 2 * FLAGS.nfilter) + 4096      # frozen receiver
    n += 12 * B * FLAGS.nsymbol * (o.K + o.CP) * 2 * 4          # inputs, outputs, generator frames, channel truth
    n += 2 * B * o.frame_size * 4 * 4 + steps * B * 4 + (1 << 20)
    return (n + (4 << 20)) // 256 * 256


class EqualizerChainGroup:
    def __init__(self, flags_list: Sequence, rx_params_list: Sequence[Dict[str, np.ndarray]], device="cuda"):
        self.lib = _lib.load()
        if not (1 <= len(flags_list) <= int(self.lib.dccn_chain_group_max())):
            raise ValueError("1..%d chains per group" % int(self.lib.dccn_chain_group_max()))
        nbytes = max(arena_bytes_for(F, self.lib) for F in flags_list)
        self.chains: List[_Chain] = [_Chain(F, rx, device, nbytes) for F, rx in zip(flags_list, rx_params_list)]
        check_same_layout([c.arena for c in self.chains])
        c0 = self.chains[0]
        for c in self.chains[1:]:
            if (c.B, c.steps) != (c0.B, c0.steps):
                raise ValueError("the chains of a group step through the same epoch schedule (batch_size, msg_length)")
        if not bool(self.lib.dccn_eq_group_supported(C.byref(c0.pl.shape))):
            raise _lib.DccnError("dccn_eq_group_supported: this batch shape has launches that cannot carry several chains")
        self.device = c0.tr.device
        self.hp = c0.tr.hp
        self._tables = {}

    # ---- one lock-step training step of the chains in `act` -------------------------------------------------------------------
    def _table(self, act: List[_Chain]):
        """pointer tables of an active set: [pipe code][q] -> (shapes, buffers), [q] -> monitors"""
        key = tuple(id(c) for c in act)
        t = self._tables.get(key)
        if t is None:
            t = dict(n=len(act),
                     shapes=[_ptr_array([c.loop.pls[q].shape for c in act]) for q in range(2)],
                     bufs={(pipe, q): _ptr_array([c.loop.pls[q].pipe_buffers[pipe] for c in act]) for pipe in range(4) for q in range(2)},
                     mon=[_ptr_array([c.monitors[q] for c in act]) for q in range(2)],
                     desc=_ptr_array([c.loop.fg.desc for c in act]),
                     x=[(C.c_void_p * len(act))(*[c.loop.pls[q].x.data_ptr() for c in act]) for q in range(2)],
                     npow=[(C.c_void_p * len(act))(*[c.loop.fg.npow[q].data_ptr() for c in act]) if act[0].loop.fg.npow is not None
                           else None for q in range(2)])
            self._tables[key] = t
        return t

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _generate(self, act, t, i: int, q: int, materialise: bool):
        """batch i of the epoch for every chain: ONE fused generator launch (+ one that forms x for an epoch's first batch) -- or
        none at all: the group step that normalises the batch produces it too (dccn_eq_buffers.gen_next_rides), the descriptors
        its buffers hold are armed here"""
        if not materialise and all(getattr(c.loop, "ride_gen", False) for c in act):
            for c in act:
                lp = c.loop
                lp.fg.arm(lp.pls[q].bits, q, None, lp.H[q], lp.snr_rows[i], into=lp.virt[q ^ 1])
            return
        for c in act:
            lp = c.loop
            lp.fg.arm(lp.pls[q].bits, q, None, lp.H[q], lp.snr_rows[i])                    # (advances the chain's gen.offset)
        st = self._stream()
        check(self.lib.dccn_gen_static_frames_grouped(t["n"], t["desc"], st), "dccn_gen_static_frames_grouped")
        if materialise:
            check(self.lib.dccn_gen_static_apply_grouped(t["n"], t["desc"], t["x"][q], t["npow"][q], st),
                  "dccn_gen_static_apply_grouped")

    def step(self, act: List[_Chain], i: int):
        t = self._table(act)
        steps = act[0].steps
        q = i & 1
        if i == 0:
            self._generate(act, t, 0, 0, True)
        if i + 1 < steps:
            self._generate(act, t, i + 1, q ^ 1, False)              # before step i: its optimizer launch normalises it
        last = i + 1 == steps
        pipe = (3 if last else 0) if i == 0 else (2 if last else 1)
        for c in act:                                                # (_FusedPlan.run's bookkeeping of the shared workspace)
            pl = c.loop.pls[q]
            pl._ahead()[pl.ws.data_ptr()] = pl._partner if pipe in (0, 1) else None
        st = self._stream()
        check(self.lib.dccn_eq_train_step_grouped(t["n"], t["shapes"][q], t["bufs"][(pipe, q)], self.hp, st),
              "dccn_eq_train_step_grouped")
        if not all(c.loop.mon is not None for c in act):             # (else the step's optimizer launch carried the monitors)
            check(self.lib.dccn_eq_monitor_accumulate_grouped(t["n"], t["mon"][q], st), "dccn_eq_monitor_accumulate_grouped")

    # ---- the epoch loop of receiver_mp._train_on_device for all chains ---------------------------------------------------------
    def train(self, verbose: bool = False) -> List[dict]:
        try:
            act = [c for c in self.chains if not c.done]
            while act:
                for c in act:
                    c.begin_epoch()
                for i in range(act[0].steps):
                    self.step(act, i)
                for c in act:
                    c.end_epoch(verbose)
                act = [c for c in act if not c.done]
        finally:
            for c in self.chains:
                c.best_path = c.best.flush()                          # also on exceptions / KeyboardInterrupt
        return [c.result() for c in self.chains]


def train_group(flags_list: Sequence, rx_params_list: Sequence[Dict[str, np.ndarray]], device="cuda",
                verbose: bool = False) -> List[dict]:
    """:func:`dl_ofdm_amd.receiver_mp.train` (device_data, static channels) for several chains at once; returns one result dict
    per chain (history, best_path, trainer) -- the bits the per-chain function returns"""
    return EqualizerChainGroup(flags_list, rx_params_list, device=device).train(verbose=verbose)
