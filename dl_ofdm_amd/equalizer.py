"""Transfer-learning step engine of the DCCN channel equaliser.

dev/py/ofdmreceiver_np_mp.py:264-347 re-wires the trained basic-receiver graph so that
``input:0 -> Equalizer/* -> receiver`` and trains ONLY the ``Equalizer/*`` variables with a fresh Adam
(``optimizer/*``) on ``ce_mean + 1e-3 * sum(regularization_losses)``.  Here:

  * the receiver is a frozen :class:`~dl_ofdm_amd.model.OfdmDenseRx` (its kernels run forward and
    backward-to-input only);
  * the equaliser's 20 variables live in ONE flat fp32 arena (parameter / gradient / Adam m / Adam v /
    L2 coefficient), the :class:`~dl_ofdm_amd.complex.VariableStore` parameters being views into it, so
    the optimizer is a single ``dccn_adam_tf_step`` launch over the arena -- the L2 terms enter there
    as ``g + 1e-3 * 2 * 0.01 * w`` exactly like the receiver's fused step;
  * every operator is a libdccn kernel (ops.py); torch only owns memory and the autograd tape.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, ops
from ._lib import AdamHParams, EqBuffers, EqShape, check
from .complex import VariableStore
from .engine import PARAM_NAMES, param_layout
from .model import OfdmDenseRx, equalizer_ofdm, ofdm_dense_rx

EQ_REG_COEFF = 1e-3          # ofdmreceiver_np_mp.py:321
REG_L2 = 0.01                # tf.keras.regularizers.l2(l=0.01) on every dense kernel/bias (model.py:371-462)
SUPPORTED_OPTS = (0, 9, 10)  # the variants that build ``equalizer_ofdm`` (ofdmreceiver_np_mp.py:285,301-304)


class _FusedPlan:
    """Resident buffers + captured hipGraphs of ``dccn_eq_train_step`` / ``dccn_eq_eval_step`` for one batch
    size (include/dccn.h: "the fused equaliser transfer-learning step")."""

    def __init__(self, tr: "EqualizerTrainer", batch: int, twin_of: "Optional[_FusedPlan]" = None, arena=None):
        """twin_of: a plan of the same batch size whose workspace and outputs this one shares -- only the input frames and
        labels are its own.  Two twins hold consecutive batches of a training loop: while one's step runs, the generator
        fills the other's input, and the running step normalises it on its optimizer launch (``pipe_with``).
        arena: a :class:`~dl_ofdm_amd.arena.ChainArena` to place every buffer in (chain groups: equalizer_group.py)."""
        from . import arena as A
        F, o, dev = tr.FLAGS, tr.ofdmobj, tr.device
        self.arena = arena
        self.tr, self.batch = tr, int(batch)
        self.pipe_buffers: Dict[int, EqBuffers] = {}
        self.shape = EqShape(self.batch, F.nsymbol, o.K, o.CP, 1 if F.cp else 0, F.nfilter, o.frame_size, F.nbits, o.pilot_size,
                             len(o.pilotCarriers))
        offs = (C.c_longlong * 21)()
        check(tr.lib.dccn_eq_param_offsets(C.byref(self.shape), offs), "dccn_eq_param_offsets")
        assert offs[20] == tr.arena_size and [offs[i] for i in range(20)] == [tr.layout[n][0] for n in tr.names]
        f32 = dict(dtype=torch.float32, device=dev)
        B, S, n_sc = self.batch, F.nsymbol, o.K + o.CP
        self.x = A.zeros(arena, B, S, n_sc, 2, **f32)
        # (the label buffer is set aside for 16-QAM whatever this chain's modulation: the arena layout must not depend on it)
        self.bits = A.zeros(arena, B, o.frame_size, F.nbits, dtype=torch.int32, device=dev, reserve=B * o.frame_size * 4)
        if twin_of is not None:
            assert twin_of.batch == self.batch
            for n in ("out_eq", "chest", "snr_db", "metrics_buf", "tx_power", "ws", "nws"):
                setattr(self, n, getattr(twin_of, n))
        else:
            self.out_eq = A.empty(arena, B, S, n_sc, 2, **f32)
            self.chest = A.empty(arena, B, S, o.K, 2, **f32)
            self.snr_db = A.empty(arena, B, 1, **f32)
            self.metrics_buf = A.zeros(arena, _lib.METRICS_BYTES, dtype=torch.uint8, device=dev)
            self.tx_power = A.zeros(arena, 1, **f32)
            self.nws = tr.lib.dccn_eq_workspace_size(C.byref(self.shape), 1)
            self.ws = A.empty(arena, self.nws, dtype=torch.uint8, device=dev)
        self.buffers = self._buffers()
        self._partner = None
        self.graphs: Dict[object, C.c_void_p] = {}

    def _buffers(self, x_next=None, pre: int = 0, slot: int = 0, virt: int = 0, gen_rides: bool = False, monitor: int = 0) -> EqBuffers:
        tr = self.tr
        p = lambda t: t.data_ptr()          # noqa: E731
        return EqBuffers(p(self.x), p(self.bits), p(tr.params), p(tr.grads), p(tr.adam_m), p(tr.adam_v),
                         p(tr.reg_coef), p(tr.adam_state), p(tr.rx_arena), p(self.out_eq), p(self.chest),
                         p(self.snr_db), p(tr.pilot_carriers), None, p(self.metrics_buf), p(self.tx_power),
                         p(self.ws), self.nws, 1, p(tr.rx_folded(self.shape)),        # reg_uniform: _flatten() fills one value per dense tensor
                         None if x_next is None else p(x_next), int(pre), int(slot), (virt or None) if x_next is not None else None,
                         C.addressof(tr._tune) if getattr(tr, "_tune", None) is not None else None,
                         1 if (gen_rides and virt and x_next is not None) else 0, monitor or None)

    def pipe_with(self, other: "_FusedPlan", slot: int, virt=None, gen_rides: bool = False, monitor=None):
        """Training steps of this plan normalise `other`'s input on their optimizer launch (include/dccn.h
        dccn_eq_buffers.x_next); run(True, pipe=0) starts a chain (own normalisation), pipe=1 continues one.
        ``virt`` (a ``_lib.GenStatic`` the caller keeps alive): that input is never written -- the launch reads the fused
        generator's (y, noise, power partials) instead (dccn_eq_buffers.x_next_virtual); the caller issues the generator launch
        of the batch before the step, and only a chain's FIRST batch has to sit in ``x``."""
        assert other.ws is self.ws
        self._virt = virt
        other._x_virtual = virt is not None               # `other.x` is stale for every batch but a chain's first
        for key in [k for k in self.graphs if isinstance(k, tuple)]:      # captured with the previous partner's pointers
            self.tr.lib.dccn_rx_graph_destroy(self.graphs.pop(key))
        # 0 starts a chain, 1 continues it, 2 ends it (consumes the batch normalised ahead, normalises nothing: the last
        # step of an epoch), 3 = a one-step chain is the plain step
        va = C.addressof(virt) if virt is not None else 0
        # gen_rides: the steps that normalise the next batch also PRODUCE it (dccn_eq_buffers.gen_next_rides; the caller arms
        # `virt` per batch instead of launching the generator)
        # monitor (a ``_lib.EqMonitor`` the caller keeps alive): the loop's per-step monitors ride on the optimizer launch
        # (dccn_eq_buffers.monitor) instead of following the step as a launch of their own
        self._monitor = monitor
        ma = C.addressof(monitor) if monitor is not None else 0
        self.pipe_buffers = {0: self._buffers(other.x, 0, slot, va, gen_rides, ma), 1: self._buffers(other.x, 1, slot, va, gen_rides, ma),
                             2: self._buffers(None, 1, slot, monitor=ma), 3: self._buffers(None, 0, slot, monitor=ma)}
        self._partner = other

    def _ahead(self) -> dict:
        """which plan's input the shared workspace currently holds normalised (twins share one workspace): any run that is
        not a link of the chain clears it, and a link that does not find its own batch there starts a new chain instead"""
        return self.tr.__dict__.setdefault("_normalised_ahead", {})

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.tr.device).cuda_stream)

    def set_batch(self, x, bits):
        self.x.copy_(torch.as_tensor(x, dtype=torch.float32).reshape(self.x.shape), non_blocking=True)
        self.bits.copy_(torch.as_tensor(bits).to(torch.int32).reshape(self.bits.shape), non_blocking=True)

    def run(self, train: bool, graph: bool = True, pipe: Optional[int] = None):
        lib, tr = self.tr.lib, self.tr
        assert pipe is None or train
        ahead, wskey = self._ahead(), self.ws.data_ptr()
        if pipe in (1, 2) and ahead.get(wskey) is not self:      # an eval / plain run used the workspace in between
            if getattr(self, "_x_virtual", False):
                raise _lib.DccnError("the batch normalised ahead was lost and this plan's input was never materialised "
                                     "(x_next_virtual): start the chain again from a materialised batch (pipe=0)")
            pipe = 0 if pipe == 1 else 3
        bufs = self.buffers if pipe is None else self.pipe_buffers[pipe]
        ahead[wskey] = self._partner if pipe in (0, 1) else None
        if not graph:
            if train:
                check(lib.dccn_eq_train_step(C.byref(self.shape), C.byref(bufs), tr.hp, self._stream()),
                      "dccn_eq_train_step")
            else:
                check(lib.dccn_eq_eval_step(C.byref(self.shape), C.byref(bufs), self._stream()),
                      "dccn_eq_eval_step")
            return
        mode = 1 if train else 0
        key = mode if pipe is None else (mode, pipe)
        if key not in self.graphs:
            g = C.c_void_p(0)
            torch.cuda.synchronize(tr.device)
            check(lib.dccn_eq_graph_create(C.byref(self.shape), C.byref(bufs), mode, tr.hp, self._stream(),
                                           C.byref(g)), "dccn_eq_graph_create")
            self.graphs[key] = g
        check(lib.dccn_rx_graph_launch(self.graphs[key], self._stream()), "dccn_rx_graph_launch")

    def close(self):
        for g in self.graphs.values():
            self.tr.lib.dccn_rx_graph_destroy(g)
        self.graphs = {}


class EqualizerTrainer:
    """``session.run([train_op, power_tx, ce_mean, berlin, chan_rms], feed)`` of the equaliser harness.

    rx_params: the trained basic receiver (name -> array, engine layout); FLAGS/ofdmobj as in the
    reference.  ``train_step(x, bits[, chan_gt])`` / ``eval_step(x, bits)`` take the raw `tx_ofdm` batch
    [frames, n_sym, n_sc, 2] (host or device) and return a metrics dict."""

    def __init__(self, FLAGS, ofdmobj, rx_params: Dict[str, np.ndarray], device="cuda", seed: int = 1,
                 lr0: Optional[float] = None, arena=None):
        """arena: a :class:`~dl_ofdm_amd.arena.ChainArena` that receives every device buffer the fused step reads or writes
        (parameter / gradient / Adam arenas, the frozen receiver, pilot table): chain groups, equalizer_group.py."""
        from . import arena as A
        self.arena = arena
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.DccnError("EqualizerTrainer needs a CUDA (ROCm) device; there is no CPU fallback")
        opt = int(getattr(FLAGS, "opt", 0))
        if opt not in SUPPORTED_OPTS:
            raise NotImplementedError("equaliser variant opt=%d is outside this implementation's scope "
                                      "(equalizer_ofdm only: opt in %s)" % (opt, SUPPORTED_OPTS))
        self.FLAGS, self.ofdmobj = FLAGS, ofdmobj
        self.rx = OfdmDenseRx(FLAGS, ofdmobj, seed=seed, device=device)
        self.rx.import_params(rx_params)
        for p in self.rx.store.parameters():
            p.requires_grad_(False)
        self.store = VariableStore(seed=seed, device=device)
        n_sc = ofdmobj.K + ofdmobj.CP
        with torch.no_grad():                                         # first pass creates the variables
            self._equalizer(torch.randn(2, FLAGS.nsymbol, n_sc, 2, device=self.device))
        self.names = [n for n in self.store.names() if n.startswith("Equalizer/")]
        self._flatten()
        self.hp = AdamHParams.default(float(lr0 if lr0 is not None else getattr(FLAGS, "init_learning", 1e-3)))
        self.adam_state = A.place(arena, torch.tensor([0.0, 0.9, 0.999, 0.0], dtype=torch.float32, device=self.device))
        self.last: dict = {}
        # frozen receiver as one arena (dccn_rx_param_offsets layout) + pilot carriers for the fused step
        lay, total = param_layout(self.rx.dims())
        self.rx_arena = A.zeros(arena, total, dtype=torch.float32, device=self.device, reserve=lay["demodulation/conv2d/kernel"][0] + 256)   # (room for the 200 tail weights of 16-QAM: one layout for every modulation)
        for n in PARAM_NAMES:
            o_, shp = lay[n]
            self.rx_arena[o_:o_ + int(np.prod(shp))] = torch.as_tensor(
                np.asarray(rx_params[n], dtype=np.float32).reshape(-1)).to(self.device)
        self.pilot_carriers = A.place(arena, torch.as_tensor(np.asarray(ofdmobj.pilotCarriers, dtype=np.int32)).to(self.device))
        self.fused_ok = True
        self._plans: Dict[int, _FusedPlan] = {}
        self._rx_folded = None

    def pin_tuning(self):
        """Capture the library's tuning table NOW for every plan built from here on (dccn_eq_buffers.tuning): later
        ``dccn_set_tuning`` calls no longer reach this trainer's steps.  Call before the first plan is created."""
        if self._plans:
            raise _lib.DccnError("pin_tuning() after plans were built: their buffers hold the old (global) table")
        n = int(self.lib.dccn_tuning_count())
        self._tune = (C.c_int * n)()
        check(self.lib.dccn_tuning_snapshot(self._tune, n) - n, "dccn_tuning_snapshot")
        return self

    def rx_folded(self, shape) -> torch.Tensor:
        """The frozen receiver's C-Conv + dense layer as one matrix (include/dccn.h dccn_eq_rx_fold): built once -- the
        receiver's weights do not change while the equaliser trains (ofdmreceiver_np_mp.py:330)."""
        if self._rx_folded is None:
            n = int(self.lib.dccn_eq_rx_folded_floats(C.byref(shape)))
            from . import arena as A
            self._rx_folded = A.empty(self.arena, n, dtype=torch.float32, device=self.device)
            check(self.lib.dccn_eq_rx_fold(C.byref(shape), self.rx_arena.data_ptr(), self._rx_folded.data_ptr(),
                                           C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "dccn_eq_rx_fold")
        return self._rx_folded

    # ---- arena ---------------------------------------------------------------------------------
    def _flatten(self):
        f32 = dict(dtype=torch.float32, device=self.device)
        sizes = [self.store.tensor(n).numel() for n in self.names]
        self.n_params = int(sum(sizes))
        # every tensor starts on a 16-byte boundary of the arena (dccn_eq_param_offsets): vector loads for every layer
        total = int(sum((sz + 3) // 4 * 4 for sz in sizes))
        self.arena_size = total
        from . import arena as A
        self.params, self.grads = A.zeros(self.arena, total, **f32), A.zeros(self.arena, total, **f32)
        self.adam_m, self.adam_v = A.zeros(self.arena, total, **f32), A.zeros(self.arena, total, **f32)
        self.reg_coef = A.zeros(self.arena, total, **f32)
        self.layout, o = {}, 0
        for n, sz in zip(self.names, sizes):
            p = self.store.tensor(n)
            self.params[o:o + sz].copy_(p.detach().reshape(-1))
            p.data = self.params[o:o + sz].view(p.shape)              # parameter now aliases the arena
            p.grad = self.grads[o:o + sz].view(p.shape)               # autograd accumulates in place
            if "/dense" in n:
                self.reg_coef[o:o + sz] = EQ_REG_COEFF * 2.0 * REG_L2
            self.layout[n] = (o, tuple(p.shape))
            o += (sz + 3) // 4 * 4

    def view(self, name: str, arena: Optional[torch.Tensor] = None) -> torch.Tensor:
        o, shp = self.layout[name]
        a = self.params if arena is None else arena
        return a[o:o + int(np.prod(shp))].view(*shp)

    def load_params(self, params: Dict[str, np.ndarray]):
        with torch.no_grad():
            for n in self.names:
                self.view(n).copy_(torch.as_tensor(np.asarray(params[n], dtype=np.float32)).reshape(self.layout[n][1]))

    def get_params(self) -> Dict[str, np.ndarray]:
        return {n: self.view(n).detach().cpu().numpy().copy() for n in self.names}

    def get_grads(self) -> Dict[str, np.ndarray]:
        return {n: self.view(n, self.grads).detach().cpu().numpy().copy() for n in self.names}

    # ---- graph ---------------------------------------------------------------------------------
    def _equalizer(self, x_norm):
        self.store.begin()
        with self.store.scope("Equalizer"):
            return equalizer_ofdm(x_norm, self.FLAGS, self.ofdmobj, scope=self.store)

    def _forward(self, x, bits):
        x = torch.as_tensor(x, dtype=torch.float32).to(self.device, non_blocking=True)
        bits = torch.as_tensor(bits).to(device=self.device, dtype=torch.int32, non_blocking=True)
        x_norm, tx_power = ops.batch_moment_norm(x), None                       # `input:0` (ofdmreceiver_np.py:128-137)
        _, tx_power = ops.clip_power(x_norm, peak=8.0, want_clipped=False)      # `tx_power:0` monitor (:131)
        out_eq, snr_db, chest = self._equalizer(x_norm)
        self.rx.store.begin()
        prob, ce, mbuf, _ = ofdm_dense_rx(out_eq, self.FLAGS, self.ofdmobj, self.rx.outshape, scope=self.rx.store,
                                          bits=bits)
        return ce, mbuf, tx_power, snr_db, chest, out_eq, prob

    def _metrics(self, mbuf, tx_power, chan_rms=None) -> dict:
        m = ops.read_metrics(mbuf)
        m["tx_power"] = float(tx_power)
        if chan_rms is not None:
            m["chan_rms"] = float(chan_rms)
        self.last = m
        return m

    def chan_rms(self, chest: torch.Tensor, chan_gt) -> torch.Tensor:
        """ofdmreceiver_np_mp.py:245, 325-333 monitor: MSE between the true channel and the estimate, each passed through
        tf.keras.layers.LayerNormalization(axis=1, center=False, scale=False) -- moments over the OFDM-symbol axis only,
        Keras' default epsilon 1e-3 (no gradient: it is fetched, never part of total_loss)."""
        def ln_axis1(t):
            mean = t.mean(dim=1, keepdim=True)
            var = ((t - mean) ** 2).mean(dim=1, keepdim=True)
            return (t - mean) * torch.rsqrt(var + 1e-3)
        with torch.no_grad():
            g = torch.view_as_real(torch.as_tensor(chan_gt).to(self.device).to(torch.complex64)).contiguous()
            a = ln_axis1(g)
            b = ln_axis1(torch.view_as_real(chest).contiguous())
            return ((a - b) ** 2).mean()

    def _adam_step(self):
        check(self.lib.dccn_adam_tf_step(self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(),
                                         self.adam_v.data_ptr(), self.reg_coef.data_ptr(), None,
                                         self.adam_state.data_ptr(), self.hp, self.arena_size,
                                         C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
              "dccn_adam_tf_step")

    def _plan(self, batch: int) -> _FusedPlan:
        if batch not in self._plans:
            self._plans[batch] = _FusedPlan(self, batch)
        return self._plans[batch]

    def _fused_step(self, x, bits, train: bool, graph: bool, chan_gt=None) -> dict:
        pl = self._plan(int(np.shape(x)[0]) if not isinstance(x, torch.Tensor) else int(x.shape[0]))
        pl.set_batch(x, bits)
        pl.run(train, graph)
        rms = None
        if chan_gt is not None:
            rms = self.chan_rms(torch.view_as_complex(pl.chest), chan_gt)
        return self._metrics(pl.metrics_buf, pl.tx_power, rms)

    def resident(self, batch: int) -> _FusedPlan:
        """the fused plan of this batch size: fill ``.x`` / ``.bits`` in place (e.g. from
        :class:`~dl_ofdm_amd.datagen.DeviceDataGen`), then ``.run(train)``; ``.metrics_buf`` / ``.tx_power`` /
        ``.chest`` / ``.out_eq`` / ``.snr_db`` hold the step's outputs on the device."""
        return self._plan(batch)

    def __del__(self):
        try:
            for pl in self._plans.values():
                pl.close()
        except Exception:
            pass

    def train_step(self, x, bits, chan_gt=None, fused: bool = True, graph: bool = True) -> dict:
        """fused=True: the pre-planned ``dccn_eq_train_step`` sequence (hipGraph replay); fused=False: the
        same kernels composed through the layer API and the autograd tape."""
        if fused and self.fused_ok:
            return self._fused_step(x, bits, True, graph, chan_gt)
        self.grads.zero_()
        ce, mbuf, tx_power, snr_db, chest, _, _ = self._forward(x, bits)
        ce.backward()
        self._adam_step()
        rms = self.chan_rms(chest, chan_gt) if chan_gt is not None else None
        return self._metrics(mbuf, tx_power, rms)

    @torch.no_grad()
    def eval_step(self, x, bits, fused: bool = True, graph: bool = True) -> dict:
        if fused and self.fused_ok:
            return self._fused_step(x, bits, False, graph)
        ce, mbuf, tx_power, _, _, _, _ = self._forward(x, bits)
        return self._metrics(mbuf, tx_power)

    def adam(self) -> dict:
        s = self.adam_state.cpu().numpy()
        return dict(global_step=float(s[0]), beta1_power=float(s[1]), beta2_power=float(s[2]), alpha=float(s[3]))

    def total_loss(self, m: Optional[dict] = None) -> float:
        """ofdmreceiver_np_mp.py:322-323: ce_mean + 1e-3 * sum(REGULARIZATION_LOSSES) -- the collection also
        holds the frozen receiver's two dense layers."""
        m = m or self.last
        reg = sum(REG_L2 * float((self.view(n) ** 2).sum()) for n in self.names if "/dense" in n)
        reg += sum(REG_L2 * float((self.rx.store.tensor(n) ** 2).sum()) for n in self.rx.store.names() if "/dense" in n)
        return m["ce_mean"] + EQ_REG_COEFF * reg

    # ---- checkpoints (TF variable names) ------------------------------------------------------------
    def state_dict_tf(self) -> Dict[str, np.ndarray]:
        """what tf.train.Saver stores for the edited graph: Equalizer/* plus, under the ``optimizer`` scope,
        global_step, the beta powers and the Adam slots (ofdmreceiver_np_mp.py:319-330)."""
        out = {}
        for n in self.names:
            out[n] = self.view(n).detach().cpu().numpy()
            out["optimizer/" + n + "/Adam"] = self.view(n, self.adam_m).detach().cpu().numpy()
            out["optimizer/" + n + "/Adam_1"] = self.view(n, self.adam_v).detach().cpu().numpy()
        a = self.adam()
        out["optimizer/global_step"] = np.float32(a["global_step"])
        out["optimizer/beta1_power"] = np.float32(a["beta1_power"])
        out["optimizer/beta2_power"] = np.float32(a["beta2_power"])
        return out

    def load_state_dict_tf(self, z, with_optimizer: bool = True):
        self.load_params({n: z[n] for n in self.names})
        if with_optimizer and ("optimizer/global_step" in z):
            with torch.no_grad():
                for n in self.names:
                    self.view(n, self.adam_m).copy_(torch.as_tensor(z["optimizer/" + n + "/Adam"]))
                    self.view(n, self.adam_v).copy_(torch.as_tensor(z["optimizer/" + n + "/Adam_1"]))
                self.adam_state.copy_(torch.tensor([float(z["optimizer/global_step"]),
                                                    float(z["optimizer/beta1_power"]),
                                                    float(z["optimizer/beta2_power"]), 0.0]))
