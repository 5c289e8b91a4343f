"""The fused basic-receiver step: one ``session.run`` of the reference as one HIP launch
sequence (optionally a replayed hipGraph).

Replaces dev/py/ofdmreceiver_np.py:121-189 (graph) + :234 (training run) + :80 (evaluation
run).  All state -- parameters, Adam slots, ``global_step``/beta powers, activations the
reference exposes by tensor name -- lives in flat HBM arenas owned by this object; the
library only launches kernels into them.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import AdamHParams, RxBuffers, RxShape, check
from .ops import read_metrics

# checkpoint variable names -> (arena segment, live shape)
PARAM_NAMES = ("fft_like/conv3d/kernel", "fft_like/conv3d/bias",
               "demodulation/dense/kernel", "demodulation/dense/bias",
               "demodulation/conv2d/kernel", "demodulation/conv2d/bias",
               "demodulation/dense_1/kernel", "demodulation/dense_1/bias")
REGULARIZED = ("demodulation/dense/kernel", "demodulation/dense/bias",
               "demodulation/dense_1/kernel", "demodulation/dense_1/bias")
REG_COEFF = 1e-4      # ofdmreceiver_np.py:162
REG_L2 = 0.01         # model.py:1271-1272,1285-1286


@dataclass(frozen=True)
class RxDims:
    S: int      # nsymbol
    kin: int    # samples per symbol seen by the C-Conv
    F: int      # nfilter
    D: int      # frame_size
    nbits: int

    @property
    def m(self):
        return 2 ** self.nbits


def param_layout(d: RxDims):
    """name -> (offset, shape) inside the flat parameter arena (dccn_rx_param_offsets)."""
    F2, m, b = 2 * d.F, d.m, d.nbits
    out, o = {}, 0
    for name, shape in (("fft_like/conv3d/kernel", (d.kin, F2)), ("fft_like/conv3d/bias", (F2,)),
                        ("demodulation/dense/kernel", (d.S * F2, 2 * d.D)), ("demodulation/dense/bias", (2 * d.D,)),
                        ("demodulation/conv2d/kernel", (2, m)), ("demodulation/conv2d/bias", (m,)),
                        ("demodulation/dense_1/kernel", (m + 2, 2 * b)), ("demodulation/dense_1/bias", (2 * b,))):
        n = int(np.prod(shape))
        out[name] = (o, shape)
        o += n
    return out, o


def glorot_init(d: RxDims, seed: int = 1) -> Dict[str, np.ndarray]:
    """glorot-uniform kernels / zero biases with the reference's fans (the conv3d fans count
    the K dead taps of the [1,K,1,K,2F] TF kernel -- SURVEY.md Appendix A.7)."""
    rng = np.random.RandomState(seed)
    fans = {"fft_like/conv3d/kernel": (d.kin * d.kin, d.kin * 2 * d.F),
            "demodulation/dense/kernel": (2 * d.S * d.F, 2 * d.D),
            "demodulation/conv2d/kernel": (2, d.m),
            "demodulation/dense_1/kernel": (d.m + 2, 2 * d.nbits)}
    lay, _ = param_layout(d)
    p = {}
    for name, (_, shape) in lay.items():
        if name.endswith("bias"):
            p[name] = np.zeros(shape, np.float32)
        else:
            fi, fo = fans[name]
            lim = np.sqrt(6.0 / (fi + fo))
            p[name] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
    return p


class RxEngine:
    """Fixed-shape receiver step engine on one GPU.

    train_step(x, bits) == session.run([train_op, power_tx, ce_mean, berlin], feed)
    eval_step(x, bits)  == session.run([conf_matrix, berlin, power_tx, ce_mean, ...], feed)
    """

    def __init__(self, dims: RxDims, batch: int, device="cuda", train: bool = True, seed: int = 1,
                 params: Optional[Dict[str, np.ndarray]] = None, lr0: float = 1e-3, want_prob: bool = True,
                 want_tx_power: bool = True, want_z: bool = True, want_dfft: bool = True, want_grads: bool = True):
        self.lib = _lib.load()
        self.dims, self.batch, self.train = dims, int(batch), bool(train)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.DccnError("RxEngine needs a CUDA (ROCm) device; there is no CPU fallback")
        self.shape = RxShape(self.batch, dims.S, dims.kin, dims.F, dims.D, dims.nbits)
        offs = (C.c_longlong * 6)()
        check(self.lib.dccn_rx_param_offsets(C.byref(self.shape), offs), "dccn_rx_param_offsets")
        self.layout, total = param_layout(dims)
        assert total == offs[5] and self.layout["demodulation/conv2d/kernel"][0] == offs[4]
        self.n_params = total
        f32 = dict(dtype=torch.float32, device=self.device)
        B, d = self.batch, dims
        self.params = torch.zeros(total, **f32)
        self.x = torch.zeros(B, d.S, d.kin, 2, **f32)
        self.bits = torch.zeros(B, d.D, d.nbits, dtype=torch.int32, device=self.device)
        self.x_norm = torch.empty(B, d.S, d.kin, 2, **f32)
        self.fft_out = torch.empty(B, d.S, d.F, 2, **f32)
        # the dense output: the tail runs inside the dense launch and z only exists when asked for
        # (the library's own plan queries decide which intermediate buffers a step can do without)
        fused_tail = bool(self.lib.dccn_rx_dense_tail_fused(C.byref(self.shape), 1 if train else 0))
        self.z = torch.empty(B, 2 * d.D, **f32) if (want_z or not fused_tail) else None
        self.prob = torch.empty(B, d.D, d.nbits, 2, **f32) if want_prob else None
        self.metrics_buf = torch.zeros(_lib.METRICS_BYTES, dtype=torch.uint8, device=self.device)
        self.tx_power = torch.zeros(1, **f32) if want_tx_power else None
        if train:
            self.grads = torch.zeros(total, **f32)
            self.adam_m = torch.zeros(total, **f32)
            self.adam_v = torch.zeros(total, **f32)
            self.reg_coef = torch.zeros(total, **f32)
            for n in REGULARIZED:
                o, shp = self.layout[n]
                self.reg_coef[o:o + int(np.prod(shp))] = REG_COEFF * 2.0 * REG_L2
            self.adam_state = torch.tensor([0.0, 0.9, 0.999, 0.0], **f32)     # dccn_adam_state
            self.dz = torch.empty(B, 2 * d.D, **f32)
            # the gradient w.r.t. fft_out: its only consumer is the C-Conv weight gradient, which the fused backward
            # launch computes in the epilogue of the tiles that produce it -- materialised only on request then
            fused_bwd = bool(self.lib.dccn_rx_bwd_fused_supported(C.byref(self.shape)))
            self.dfft = torch.empty(B, d.S, d.F, 2, **f32) if (want_dfft or not fused_bwd) else None
        else:
            self.grads = self.adam_m = self.adam_v = self.reg_coef = self.adam_state = self.dz = self.dfft = None
        nws = self.lib.dccn_rx_workspace_size(C.byref(self.shape), 1 if train else 0)
        self.ws = torch.empty(nws, dtype=torch.uint8, device=self.device)
        self.hp = AdamHParams.default(lr0)
        self._graph = None
        self._graph_mode = None
        p = lambda t: 0 if t is None else t.data_ptr()   # noqa: E731
        self.buffers = RxBuffers(p(self.x), p(self.bits), p(self.params), p(self.grads), p(self.adam_m),
                                 p(self.adam_v), p(self.reg_coef), p(self.adam_state), p(self.x_norm),
                                 p(self.fft_out), p(self.z), p(self.prob), p(self.dz), p(self.dfft),
                                 p(self.metrics_buf), p(self.tx_power), p(self.ws), nws, 0, 0, 0, 0, 1 if want_grads else -1,
                                 1,       # reg_uniform_dense: self.reg_coef is one value over the dense kernel (above)
                                 0)       # x_next_ready
        # pipelined training, double-buffered: the next batch is normalised into the OTHER x_norm buffer by leading blocks of
        # the backward launch (dccn.h: x_norm_next / norm_slot); `x_norm` stays the buffer plain steps use (parity 0)
        self._norm_bufs = [self.x_norm, None]
        self._norm_parity = 0
        self._ride = int(self.lib.dccn_rx_norm_rides_backward(C.byref(self.shape))) if train else 0
        if self._ride:
            self._norm_bufs[1] = torch.empty_like(self.x_norm)
        # the same buffers in the pipelined mode (dccn.h: x_next / x_prenormalised), built on demand per (label slot, last)
        self._pipe_bufs = {}
        self.bits_alt = None                 # second label buffer: the generator fills it while a step reads the first
        self._norm_ready = False
        self._prefetch_pending = False       # a pipelined call has normalised a batch that no step has consumed yet
        self.load_params(params if params is not None else glorot_init(dims, seed))

    # ---- parameters ----------------------------------------------------------------------
    def view(self, name: str, arena: Optional[torch.Tensor] = None) -> torch.Tensor:
        o, shp = self.layout[name]
        a = self.params if arena is None else arena
        return a[o:o + int(np.prod(shp))].view(*shp)

    def load_params(self, params: Dict[str, np.ndarray]):
        for n in PARAM_NAMES:
            self.view(n).copy_(torch.as_tensor(np.asarray(params[n], dtype=np.float32)).reshape(self.layout[n][1]))

    def get_params(self) -> Dict[str, np.ndarray]:
        return {n: self.view(n).detach().cpu().numpy().copy() for n in PARAM_NAMES}

    def get_grads(self) -> Dict[str, np.ndarray]:
        return {n: self.view(n, self.grads).detach().cpu().numpy().copy() for n in PARAM_NAMES}

    # ---- steps ---------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _no_prefetch_pending(self, what: str):
        """train_step_pipelined leaves the NEXT batch normalised in x_norm; another kind of step on this engine would
        overwrite it and the following pipelined call would silently pair the wrong input with its labels."""
        if self._prefetch_pending:
            raise _lib.DccnError("%s while a prefetched batch is pending on this engine: finish the pipeline with "
                                 "train_step_pipelined(last=True) or discard it with drop_prefetch()" % what)

    def pin_tuning(self):
        """Capture the library's tuning table NOW as this engine's own (dccn_rx_buffers.tuning): later ``dccn_set_tuning`` calls
        -- another thread's experiment, another plan's preference -- no longer reach this engine's steps."""
        n = int(self.lib.dccn_tuning_count())
        self._tune = (C.c_int * n)()
        check(self.lib.dccn_tuning_snapshot(self._tune, n) - n, "dccn_tuning_snapshot")
        self.buffers.tuning = C.addressof(self._tune)
        self._pipe_bufs.clear()
        self.close_graph()
        return self

    def drop_prefetch(self):
        """Forget the batch a pipelined call has normalised ahead (the next pipelined call primes itself again)."""
        self._prefetch_pending = False
        self._norm_ready = False

    def set_batch(self, x, bits):
        """Stage a batch into the engine's resident input buffers (device copy or H2D)."""
        self._no_prefetch_pending("set_batch")
        self.x.copy_(torch.as_tensor(x, dtype=torch.float32).reshape(self.x.shape), non_blocking=True)
        self.bits.copy_(torch.as_tensor(bits).to(torch.int32).reshape(self.bits.shape), non_blocking=True)

    def train_step(self, x=None, bits=None, graph: bool = False, fork: bool = False):
        if not self.train:
            raise _lib.DccnError("engine built with train=False")
        self._no_prefetch_pending("train_step")
        if x is not None:
            self.set_batch(x, bits)
        self._norm_ready = False            # x_norm is about to hold this batch, not a prefetched one
        if graph:
            self._launch_graph(1 | (2 if fork else 0))
        else:
            check(self.lib.dccn_rx_train_step(C.byref(self.shape), C.byref(self.buffers), self.hp, self._stream()),
                  "dccn_rx_train_step")

    def prime(self, x=None):
        """R0 of the batch in ``eng.x`` (or ``x``) into ``x_norm``: what ``train_step_pipelined`` expects to find."""
        if x is not None:
            self.x.copy_(torch.as_tensor(x, dtype=torch.float32).reshape(self.x.shape), non_blocking=True)
        check(self.lib.dccn_rx_normalise(C.byref(self.shape), C.byref(self.buffers), self._stream()), "dccn_rx_normalise")
        self._norm_ready = True
        self._prefetch_pending = False
        self._norm_parity = 0                # self.buffers names x_norm buffer 0 / partial-sum slot 0

    def label_slot(self, slot: int) -> torch.Tensor:
        """Label buffer ``slot`` (0 = ``eng.bits``, 1 = a second buffer of the same shape, allocated on first use)."""
        if slot == 0:
            return self.bits
        if self.bits_alt is None:
            self.bits_alt = torch.zeros_like(self.bits)
        return self.bits_alt

    def _pipe_buffers(self, slot: int, last: bool, parity: int = 0, double: bool = False, pre: int = 1,
                      ready: int = 0, gen: int = 0, keep_x: bool = True) -> RxBuffers:
        key = (slot, last, parity, double, pre, ready, gen, keep_x)
        if key not in self._pipe_bufs:
            vals = {f: getattr(self.buffers, f) for f, _ in RxBuffers._fields_}
            vals["bits"] = self.label_slot(slot).data_ptr()
            vals["x_next"] = 0 if (last or (gen and not keep_x)) else self.x.data_ptr()
            vals["gen_next"] = 0 if last else gen
            vals["x_prenormalised"] = pre
            vals["x_norm"] = self._norm_bufs[parity].data_ptr()
            vals["norm_slot"] = parity
            vals["x_norm_next"] = self._norm_bufs[parity ^ 1].data_ptr() if (double and not last) else 0
            vals["x_next_ready"] = 0 if last else ready
            self._pipe_bufs[key] = RxBuffers(*[vals[f] for f, _ in RxBuffers._fields_])
        return self._pipe_bufs[key]

    def train_step_pipelined(self, next_x=None, bits=None, graph: bool = False, slot: int = 0, last: bool = False,
                             x_ready=None):
        """One training step with R0 software-pipelined across steps: the step runs on the batch whose normalisation is
        already in ``x_norm`` (written by the previous pipelined call, or by ``prime()``; the first call primes itself from
        ``eng.x``), with the labels of THAT batch (``bits``, or what ``label_slot(slot)`` already holds), and normalises
        ``next_x`` (default: whatever ``eng.x`` holds now) behind its Adam update for the following call -- one launch and
        one kernel boundary less per step.  ``eng.x`` is only read by R0, so the caller refills the same buffer between
        calls; the labels lag one batch behind the input, hence the two label slots for producers that write both at once.
        ``last=True``: nothing follows (end of an epoch), no normalisation is issued and the next call primes again.
        ``x_ready`` (a recorded ``torch.cuda.Event``): the next batch is being written into ``eng.x`` on ANOTHER stream; the
        launch that normalises it waits for that event, everything before it runs concurrently with the producer
        (datagen.SideStreamFeeder).  Results are bit-identical to ``train_step`` fed the same batches in the same order."""
        if not self.train:
            raise _lib.DccnError("engine built with train=False")
        if not self._norm_ready:
            self.prime()
        if bits is not None:
            self.label_slot(slot).copy_(torch.as_tensor(bits).to(torch.int32).reshape(self.bits.shape), non_blocking=True)
        if next_x is not None:
            self.x.copy_(torch.as_tensor(next_x, dtype=torch.float32).reshape(self.x.shape), non_blocking=True)
        double = self._norm_bufs[1] is not None
        if graph:
            if slot != 0 or last or x_ready is not None:
                raise _lib.DccnError("captured pipelined steps use label slot 0, always prefetch and take no producer event")
            # a captured step replays fixed buffers: it keeps the batch in whichever x_norm buffer holds it now and
            # normalises the next one into the same buffer on its optimizer launch (the single-buffer form)
            self._launch_graph(1 | 4 | (8 if self._norm_parity else 0))
        else:
            ready = 0
            if x_ready is not None and not last:
                ready = int(x_ready.cuda_event)
                if not ready:
                    raise _lib.DccnError("x_ready must be an event that has been recorded")
            bufs = self._pipe_buffers(slot, last, self._norm_parity, double, 1, ready)
            check(self.lib.dccn_rx_train_step(C.byref(self.shape), C.byref(bufs), self.hp, self._stream()), "dccn_rx_train_step")
            if double and not last:
                self._norm_parity ^= 1           # the prefetched batch sits in the other buffer
        self._prefetch_pending = not last
        if last:
            self._gen_side_primed = False
            self._norm_ready = False

    def train_step_generated(self, fgen, slot: int = 0, last: bool = False, keep_x: bool = False, side=None):
        """``train_step_pipelined`` fed by the fused device generator (datagen.FusedStaticGen), ONE C call: the step trains on
        the batch normalised ahead (labels in ``label_slot(slot)``), issues the generator launch of the NEXT batch as its first
        launch (labels to the other slot) and normalises that batch -- read as (y, noise, power partials), never written as x
        unless ``keep_x`` -- on its optimizer launch.  The first call generates, materialises and normalises batch 0 itself.
        Bit-identical to ``train_step_pipelined`` on the materialised batches (tests/test_gpu_datagen.py)."""
        if not self.train:
            raise _lib.DccnError("engine built with train=False")
        if self._ride:
            raise _lib.DccnError("train_step_generated needs the single-buffer pipelining (dccn_rx_norm_rides_backward == 0)")
        if not self._norm_ready:
            fgen.make_batch(self.x, self.label_slot(slot), slot)
            self.prime()
            self._gen_side_primed = False       # the side stream must see these main-stream writers of y / noise / partials
        gen_ptr, ready = 0, 0
        if not last:
            d = fgen.arm(self.label_slot(slot ^ 1), slot ^ 1)
            gen_ptr = C.addressof(d)
            if side is not None:
                # ``side`` = (stream, ready event, step event): the generator launch goes to that stream -- after the previous
                # step (whose last launch read the single-buffered y / noise) -- and overlaps this step's first three launches
                st, ev_ready, ev_step = side
                main = torch.cuda.current_stream(self.device)
                if getattr(self, "_gen_side_primed", False):
                    st.wait_event(ev_step)
                else:
                    st.wait_stream(main)
                    self._gen_side_primed = True
                check(self.lib.dccn_gen_static_frames(C.byref(d), C.c_void_p(st.cuda_stream)), "dccn_gen_static_frames")
                ev_ready.record(st)
                ready = int(ev_ready.cuda_event)
        bufs = self._pipe_buffers(slot, last, 0, False, 1, ready, gen_ptr, keep_x)
        check(self.lib.dccn_rx_train_step(C.byref(self.shape), C.byref(bufs), self.hp, self._stream()), "dccn_rx_train_step")
        if side is not None and not last:
            side[2].record(torch.cuda.current_stream(self.device))
        self._prefetch_pending = not last
        if last:
            self._norm_ready = False

    def eval_step(self, x=None, bits=None, graph: bool = False):
        self._no_prefetch_pending("eval_step")
        if x is not None:
            self.set_batch(x, bits)
        self._norm_ready = False
        if graph:
            self._launch_graph(0)
        else:
            check(self.lib.dccn_rx_eval_step(C.byref(self.shape), C.byref(self.buffers), self._stream()),
                  "dccn_rx_eval_step")

    def _launch_graph(self, mode: int):
        if self._graph is None or self._graph_mode != mode:
            self.close_graph()
            g = C.c_void_p(0)
            torch.cuda.synchronize(self.device)
            bufs = self._pipe_buffers(0, False, 1 if mode & 8 else 0, False) if mode & 4 else self.buffers
            check(self.lib.dccn_rx_graph_create(C.byref(self.shape), C.byref(bufs), mode & 3, self.hp,
                                                self._stream(), C.byref(g)), "dccn_rx_graph_create")
            self._graph, self._graph_mode = g, mode
        check(self.lib.dccn_rx_graph_launch(self._graph, self._stream()), "dccn_rx_graph_launch")

    def close_graph(self):
        if self._graph is not None:
            self.lib.dccn_rx_graph_destroy(self._graph)
            self._graph = None

    def __del__(self):
        try:
            self.close_graph()
        except Exception:
            pass

    # ---- fetches (synchronise) -----------------------------------------------------------
    def metrics(self) -> dict:
        """ce_mean / conf_matrix / linear_ber / log_ber (+ cost, tx_power) of the last step."""
        m = read_metrics(self.metrics_buf)
        if self.tx_power is not None:
            m["tx_power"] = float(self.tx_power.item())
        return m

    def adam(self) -> dict:
        s = self.adam_state.cpu().numpy()
        return dict(global_step=float(s[0]), beta1_power=float(s[1]), beta2_power=float(s[2]), alpha=float(s[3]))

    def cost(self, m: Optional[dict] = None) -> float:
        """`cost:0` = ce_mean + berlin*REG_COEFF*sum(reg) + log(berlin) (ofdmreceiver_np.py:171)."""
        m = m or self.metrics()
        reg = sum(REG_L2 * float((self.view(n) ** 2).sum().item()) for n in REGULARIZED)
        return m["ce_mean"] + m["berlin"] * REG_COEFF * reg + m["log_ber"]


# ---- measurement helpers --------------------------------------------------------------------
class HipTimer:
    """HIP events recorded on the stream the kernels are launched on (dccn_timer_*)."""

    def __init__(self):
        self.lib = _lib.load()
        self.h = C.c_void_p(0)
        check(self.lib.dccn_timer_create(C.byref(self.h)), "dccn_timer_create")

    def start(self, stream):
        check(self.lib.dccn_timer_start(self.h, stream), "dccn_timer_start")

    def stop(self, stream):
        check(self.lib.dccn_timer_stop(self.h, stream), "dccn_timer_stop")

    def elapsed_ms(self) -> float:
        ms = C.c_float(0)
        check(self.lib.dccn_timer_elapsed_ms(self.h, C.byref(ms)), "dccn_timer_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            self.lib.dccn_timer_destroy(self.h)
        except Exception:
            pass


def op_launchers(eng: RxEngine):
    """name -> (callable launching that operator once on the current stream into the engine's
    buffers, algorithmic FLOPs per launch, kernel symbol stem).  Used by bench.py to time the
    individual kernels of the step with HIP events."""
    lib, d, B = eng.lib, eng.dims, eng.batch
    rows, dK, dN, cells = B * d.S, d.S * d.F * 2, 2 * d.D, B * d.D
    P, G = eng.params, eng.grads
    lay = eng.layout

    def seg(arena, name):
        return arena.data_ptr() + 4 * lay[name][0]

    from .ops import workspace
    nws = max(lib.dccn_dense_bwd_w_workspace_size(B, dK, dN), lib.dccn_cconv_gemm_bwd_w_workspace_size(rows, d.kin, d.F),
              lib.dccn_demod_tail_workspace_size(cells, d.nbits),
              lib.dccn_batch_moment_norm_workspace_size(B, d.S * d.kin * 2))
    nws = max(nws, lib.dccn_dense_tail_workspace_size(B, dN, d.nbits))
    ws = workspace(nws, eng.device, "bench")
    s = eng._stream
    zbuf = eng.z if eng.z is not None else torch.empty(B, dN, dtype=torch.float32, device=eng.device)
    dfft = eng.dfft if eng.dfft is not None else torch.empty(B, d.S, d.F, 2, dtype=torch.float32, device=eng.device)
    fused_bwd = bool(lib.dccn_rx_bwd_fused_supported(C.byref(eng.shape)))
    if fused_bwd:
        nws = max(nws, lib.dccn_rx_backward_workspace_size(B, d.S, d.kin, d.F, d.D))
        ws = workspace(nws, eng.device, "bench")
    tail_flops = 3.0 * cells * (4.0 * d.m + 4.0 * (d.m + 2) * d.nbits)
    ops = {
        "batch_moment_norm": (lambda: lib.dccn_batch_moment_norm_fwd(
            eng.x.data_ptr(), eng.x_norm.data_ptr(), None, None, B, d.S * d.kin * 2, 1e-9, ws.data_ptr(), nws, s()),
            0.0, "moments/normalise"),
        "cconv_fwd": (lambda: lib.dccn_cconv_gemm_fwd(
            eng.x_norm.data_ptr(), seg(P, "fft_like/conv3d/kernel"), seg(P, "fft_like/conv3d/bias"),
            eng.fft_out.data_ptr(), rows, d.kin, d.F, s()), 8.0 * rows * d.kin * d.F, "gemm<cconv_fwd>"),
        "dense_fwd": (lambda: lib.dccn_dense_fwd(
            eng.fft_out.data_ptr(), seg(P, "demodulation/dense/kernel"), seg(P, "demodulation/dense/bias"),
            zbuf.data_ptr(), B, dK, dN, s()), 2.0 * B * dK * dN, "gemm<dense_fwd>"),
        "tail_fwd_bwd": (lambda: lib.dccn_demod_tail_loss_fwd_bwd(
            zbuf.data_ptr(), eng.bits.data_ptr(), seg(P, "demodulation/conv2d/kernel"),
            None if eng.prob is None else eng.prob.data_ptr(), eng.metrics_buf.data_ptr(), eng.dz.data_ptr(),
            seg(G, "demodulation/conv2d/kernel"), cells, d.nbits, ws.data_ptr(), nws, s()),
            tail_flops, "demod_tail"),
        "dense_bwd_x": (lambda: lib.dccn_dense_bwd_x(
            eng.dz.data_ptr(), seg(P, "demodulation/dense/kernel"), dfft.data_ptr(), B, dK, dN, s()),
            2.0 * B * dK * dN, "gemm<dense_bwd_x>"),
        "dense_bwd_w": (lambda: lib.dccn_dense_bwd_w(
            eng.fft_out.data_ptr(), eng.dz.data_ptr(), seg(G, "demodulation/dense/kernel"),
            seg(G, "demodulation/dense/bias"), B, dK, dN, ws.data_ptr(), nws, s()),
            2.0 * B * dK * dN, "gemm<dense_bwd_w>+reduce"),
        "dense_bwd": (lambda: lib.dccn_dense_bwd(
            eng.fft_out.data_ptr(), eng.dz.data_ptr(), seg(P, "demodulation/dense/kernel"), dfft.data_ptr(),
            seg(G, "demodulation/dense/kernel"), seg(G, "demodulation/dense/bias"), B, dK, dN, ws.data_ptr(), nws, s()),
            4.0 * B * dK * dN, "dense_bwd_grouped(dX+dW)+reduce"),
        "dense_bwd_slabs": (lambda: lib.dccn_dense_bwd_slabs(
            eng.fft_out.data_ptr(), eng.dz.data_ptr(), seg(P, "demodulation/dense/kernel"), dfft.data_ptr(),
            seg(G, "demodulation/dense/kernel"), seg(G, "demodulation/dense/bias"), B, dK, dN, ws.data_ptr(), nws, None,
            s()), 4.0 * B * dK * dN, "dense_bwd_grouped_km_kernel (dX 32x32x2 tiles + dW k-major 16x16x4 tiles)"),
        "cconv_bwd_w": (lambda: lib.dccn_cconv_gemm_bwd_w(
            eng.x_norm.data_ptr(), dfft.data_ptr(), seg(G, "fft_like/conv3d/kernel"),
            seg(G, "fft_like/conv3d/bias"), rows, d.kin, d.F, ws.data_ptr(), nws, s()),
            8.0 * rows * d.kin * d.F, "gemm_kmajor<cconv_bwd_w>+fold"),
    }
    if fused_bwd:
        # the backward half of the training step as it runs inside dccn_rx_train_step: dX tiles with the C-Conv weight
        # gradient in their epilogue (dfft not materialised) + dense dW items, slabs / partials left for the optimizer
        ops["rx_backward"] = (lambda: lib.dccn_rx_backward(
            eng.x_norm.data_ptr(), eng.fft_out.data_ptr(), eng.dz.data_ptr(), seg(P, "demodulation/dense/kernel"), None,
            seg(G, "demodulation/dense/kernel"), seg(G, "demodulation/dense/bias"), seg(G, "fft_like/conv3d/kernel"),
            seg(G, "fft_like/conv3d/bias"), B, d.S, d.kin, d.F, d.D, 0, ws.data_ptr(), nws, s()),
            4.0 * B * dK * dN + 8.0 * rows * d.kin * d.F,
            "rx_bwd_fused_kernel (dense dX 32x32x2 tiles + C-Conv dWeff partials in their epilogue + dense dW k-major tiles)")
    if bool(lib.dccn_dense_tail_supported(B, dK, dN, d.nbits)):
        ops["dense_tail_fwd_bwd"] = (lambda: lib.dccn_dense_tail_fwd_bwd(
            eng.fft_out.data_ptr(), seg(P, "demodulation/dense/kernel"), seg(P, "demodulation/dense/bias"), None,
            eng.bits.data_ptr(), seg(P, "demodulation/conv2d/kernel"),
            None if eng.prob is None else eng.prob.data_ptr(), eng.metrics_buf.data_ptr(), eng.dz.data_ptr(),
            seg(G, "demodulation/conv2d/kernel"), B, dK, dN, d.nbits, ws.data_ptr(), nws, s()),
            2.0 * B * dK * dN + tail_flops, "gemm16<dense_fwd+tail>+finalize")
    return ops


def time_ops(eng: RxEngine, iters: int = 200, warmup: int = 20) -> Dict[str, dict]:
    """Average per-launch duration of each operator (HIP events on the launch stream)."""
    out = {}
    timer = HipTimer()
    for name, (fn, flops, sym) in op_launchers(eng).items():
        for _ in range(warmup):
            check(fn(), name)
        timer.start(eng._stream())
        for _ in range(iters):
            fn()
        timer.stop(eng._stream())
        ms = timer.elapsed_ms() / iters
        out[name] = dict(ms=ms, flops=flops, tflops=(flops / (ms * 1e-3) / 1e12) if ms > 0 else 0.0, kernel=sym)
    return out
