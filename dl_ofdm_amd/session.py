"""Graph-name API of the reference's loaders.

Every script of the reference reaches a trained receiver the same way: ``tf.train.import_meta_graph`` +
``saver.restore`` + ``graph.get_tensor_by_name('<name>:0')`` (dev/py/model.py:51-72 ``load_model_np``,
dev/py/ofdmreceiver_np_mp.py:264-285), then ``session.run([tensors...], {placeholders...})``
(dev/py/ofdmreceiver_np.py:80,256).  There is no TensorFlow graph here; the same contract is served by
:class:`Session` over the fused engine:

    sess = Session()
    y, x, iq_receiver, outputs, total_loss, ber, berlin, conf_matrix, power_tx, noise_pwr, iq_rx, iq_tx, ce_mean, SNR = \\
        load_model_np(path, sess)
    confmax, berl, loss = sess.run([conf_matrix, berlin, ce_mean], {x: xs, y: ys, SNR: snr})

``run`` stages the fed batch into the engine's resident buffers, launches ONE evaluation step (R0 -> R1 -> R2 ->
R3-R6) and hands back the named tensors from the engine's buffers / metrics record; the monitor tensors of the graph's
unused in-graph AWGN branch (``iq_rx:0``, ``noise_power:0``; ``iq_tx:0`` = fp16 of ``tx_signal:0``) come from
``dccn_ingraph_awgn``.  Fetches are NumPy arrays / scalars, like ``session.run`` returns.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import check
from .engine import PARAM_NAMES, RxDims, RxEngine

# name -> what provides it (dev/py/ofdmreceiver_np.py:123-183, dev/py/model.py:58-71)
PLACEHOLDERS = ("bits_in:0", "tx_ofdm:0", "SNR:0")
TENSOR_NAMES = PLACEHOLDERS + ("input:0", "output:0", "cost:0", "ce_mean:0", "log_ber:0", "linear_ber:0", "conf_matrix:0",
                               "tx_signal:0", "tx_power:0", "noise_power:0", "iq_rx:0", "iq_tx:0",
                               "receiver/fft_like/fft_out:0")
SCOPES = ("transmitter", "channel", "receiver", "Equalizer", "optimizer")


class Tensor:
    """Handle returned by ``get_tensor_by_name`` (hashable: usable as a ``feed_dict`` key)."""

    def __init__(self, name: str, session: "Session"):
        self.name, self.session = name, session

    def __repr__(self):
        return "<dccn tensor %r>" % self.name

    def __hash__(self):
        return hash(self.name)

    def __eq__(self, other):
        return isinstance(other, Tensor) and other.name == self.name


def monitor_tensors(eng: RxEngine, snr, seed: int = 1, call: int = 0) -> Dict[str, torch.Tensor]:
    """The graph's in-graph AWGN branch on the batch ``eng`` normalised last (never consumed by the receiver,
    ofdmreceiver_np.py:136-138): ``tx_signal`` (complex_clip), ``iq_tx`` / ``iq_rx`` (fp16 [-1,2]), ``noise_power``."""
    lib = _lib.load()
    d, B, device = eng.dims, eng.batch, eng.device
    pairs = d.S * d.kin
    nws = lib.dccn_ingraph_awgn_workspace_size(B, pairs)
    ws = torch.empty(nws, dtype=torch.uint8, device=device)
    snr_t = torch.as_tensor(np.asarray(snr, np.float32).reshape(B), dtype=torch.float32, device=device)
    txs = torch.empty(B, d.S, d.kin, 2, dtype=torch.float32, device=device)
    iq_tx = torch.empty(B * pairs, 2, dtype=torch.float16, device=device)
    iq_rx = torch.empty(B * pairs, 2, dtype=torch.float16, device=device)
    npw = torch.zeros(1, dtype=torch.float32, device=device)
    check(lib.dccn_ingraph_awgn(eng.x_norm.data_ptr(), snr_t.data_ptr(), txs.data_ptr(), iq_tx.data_ptr(),
                                iq_rx.data_ptr(), npw.data_ptr(), B, pairs, 8.0, int(seed), int(call) & 0xFFFFFFFF,
                                ws.data_ptr(), nws, eng._stream()), "dccn_ingraph_awgn")
    return dict(tx_signal=txs, iq_tx=iq_tx, iq_rx=iq_rx, noise_power=npw)


def dims_from_params(params: Dict[str, np.ndarray], nsymbol: Optional[int] = None) -> RxDims:
    """Recover the receiver shapes from checkpoint variables (live layouts of engine.param_layout)."""
    kin, F2 = params["fft_like/conv3d/kernel"].shape[-2:]
    F = F2 // 2
    rows, cols = params["demodulation/dense/kernel"].shape
    S = rows // F2
    if nsymbol is not None and nsymbol != S:
        raise ValueError("checkpoint holds %d symbols per frame, flags say %d" % (S, nsymbol))
    m = params["demodulation/conv2d/kernel"].shape[-1]
    nbits = int(round(np.log2(m)))
    return RxDims(S=S, kin=int(kin), F=int(F), D=cols // 2, nbits=nbits)


class Session:
    """``tf.Session`` stand-in bound to one restored receiver."""

    def __init__(self, device="cuda", seed: int = 1):
        self.device = torch.device(device)
        self.lib = _lib.load()
        self.params: Optional[Dict[str, np.ndarray]] = None
        self.dims: Optional[RxDims] = None
        self.crop = None                     # (CP, K): cut the cyclic prefix off the fed frames (cp=False models)
        self._engines: Dict[int, RxEngine] = {}
        self.seed, self._calls = int(seed), 0

    # ---- restore --------------------------------------------------------------------------------------
    def restore(self, params: Dict[str, np.ndarray], crop=None, nsymbol: Optional[int] = None):
        self.params = {n: np.asarray(params[n], dtype=np.float32) for n in PARAM_NAMES}
        self.dims = dims_from_params(self.params, nsymbol)
        self.crop = crop
        self.close()                         # engines of the previous model (and their captured graphs) go first

    def get_tensor_by_name(self, name: str) -> Tensor:
        if name not in TENSOR_NAMES:
            raise KeyError("The name %r refers to a Tensor which does not exist" % name)
        return Tensor(name, self)

    def close(self):
        for e in self._engines.values():
            e.close_graph()
        self._engines.clear()

    # ---- run --------------------------------------------------------------------------------------------
    def _engine(self, batch: int, want_prob: bool = True) -> RxEngine:
        key = (batch, bool(want_prob))
        if key not in self._engines:
            if len(self._engines) >= 4:                        # keep HBM bounded when batch sizes vary
                self._engines.pop(next(iter(self._engines))).close_graph()
            self._engines[key] = RxEngine(self.dims, batch, device=self.device, train=False, params=self.params,
                                          want_prob=want_prob, want_tx_power=True)
        return self._engines[key]

    def run(self, fetches, feed_dict: Dict):
        if self.params is None:
            raise _lib.DccnError("Session.run before a model was restored")
        single = isinstance(fetches, Tensor)
        names = [f.name for f in ([fetches] if single else fetches)]
        feed = {(k.name if isinstance(k, Tensor) else str(k)): v for k, v in feed_dict.items()}
        for k in feed:
            if k not in PLACEHOLDERS:
                raise KeyError("%r is not a placeholder of the receiver graph" % k)
        if "tx_ofdm:0" not in feed or "bits_in:0" not in feed:
            raise ValueError("feed_dict must hold tx_ofdm:0 and bits_in:0")
        xs = np.asarray(feed["tx_ofdm:0"], dtype=np.float32)
        if self.crop is not None and xs.shape[2] != self.dims.kin:
            cp, k = self.crop
            xs = np.ascontiguousarray(xs[:, :, cp:cp + k, :])
        ys = np.asarray(feed["bits_in:0"]).astype(np.int32)
        batch = xs.shape[0]
        eng = self._engine(batch)
        eng.eval_step(xs, ys)
        snr = np.asarray(feed.get("SNR:0", np.zeros((batch, 1))), dtype=np.float32).reshape(batch)
        out, mon, m = [], None, None
        for n in names:
            if n in ("iq_tx:0", "iq_rx:0", "noise_power:0", "tx_signal:0") and mon is None:
                mon = self._monitor(eng, snr)
            if n in ("cost:0", "ce_mean:0", "log_ber:0", "linear_ber:0", "conf_matrix:0", "tx_power:0") and m is None:
                m = eng.metrics()
            out.append(self._fetch(n, eng, m, mon, xs, ys, snr))
        return out[0] if single else out

    def _monitor(self, eng: RxEngine, snr: np.ndarray):
        self._calls += 1
        return monitor_tensors(eng, snr, self.seed, self._calls)

    def engine_for(self, batch: int, want_prob: bool = True) -> RxEngine:
        """The evaluation engine ``run`` would use for this batch size (device-side generators write their batches
        straight into its resident ``x`` / ``bits`` buffers, then call :meth:`fetch` or read ``eng.metrics()``).
        ``want_prob=False``: an engine that does not materialise ``output:0`` (sweeps only need the metrics record)."""
        if self.params is None:
            raise _lib.DccnError("Session used before a model was restored")
        return self._engine(batch, want_prob)

    def fetch(self, fetches: Sequence[Tensor], eng: RxEngine, snr=None) -> List:
        """Named tensors of the step ``eng`` ran last (``run`` = stage the feed + ``eng.eval_step()`` + this)."""
        snr = np.zeros(eng.batch, np.float32) if snr is None else np.asarray(snr, np.float32).reshape(eng.batch)
        names = [f.name if isinstance(f, Tensor) else str(f) for f in fetches]
        mon = self._monitor(eng, snr) if any(n in ("iq_tx:0", "iq_rx:0", "noise_power:0", "tx_signal:0") for n in names) else None
        m = eng.metrics()
        return [self._fetch(n, eng, m, mon, None, None, snr) for n in names]

    def _fetch(self, n, eng, m, mon, xs, ys, snr):
        if n == "bits_in:0":
            return ys
        if n == "tx_ofdm:0":
            return xs
        if n == "SNR:0":
            return snr.reshape(-1, 1)
        if n == "input:0":
            return eng.x_norm.cpu().numpy()
        if n == "output:0":
            if eng.prob is None:
                raise _lib.DccnError("output:0 fetched from an engine built with want_prob=False")
            return eng.prob.cpu().numpy()
        if n == "receiver/fft_like/fft_out:0":
            return eng.fft_out.cpu().numpy()
        if n == "cost:0":
            return np.float32(eng.cost(m))
        if n == "ce_mean:0":
            return np.float32(m["ce_mean"])
        if n == "log_ber:0":
            return np.float64(m["log_ber"])
        if n == "linear_ber:0":
            return np.float32(m["berlin"])
        if n == "conf_matrix:0":
            return np.asarray(m["conf"], dtype=np.int32)
        if n == "tx_power:0":
            return np.float32(m["tx_power"])
        if n == "noise_power:0":
            return np.float32(mon["noise_power"].item())
        if n in ("iq_tx:0", "iq_rx:0", "tx_signal:0"):
            return mon[n[:-2]].cpu().numpy()
        raise KeyError(n)


def load_model_np(path: str, session: Session, FLAGS=None, ofdmobj=None):
    """dev/py/model.py:51-72: restore the checkpoint at ``path`` into ``session`` and return the reference's tuple
    ``(y, x, iq_receiver, outputs, total_loss, ber, berlin, conf_matrix, power_tx, noise_pwr, iq_rx, iq_tx, ce_mean,
    SNR)`` of named tensors.  ``path`` is ``<save_dir>/<token>`` (.npz of this implementation or a TensorFlow bundle
    ``.index``/``.data-*`` of the reference).  FLAGS/ofdmobj (optional) let a ``cp=False`` model accept full frames the
    way the reference's graph does (it slices the prefix off inside ``ofdm_dense_rx``, model.py:1236-1240)."""
    from .receiver import read_checkpoint_file
    z = read_checkpoint_file(path)
    crop = None
    if FLAGS is not None and ofdmobj is not None and not FLAGS.cp:
        crop = (ofdmobj.CP, ofdmobj.K)
    session.restore({n: z[n] for n in PARAM_NAMES}, crop=crop, nsymbol=getattr(FLAGS, "nsymbol", None))
    print("Load Model: %s" % path)
    g = session.get_tensor_by_name
    return (g("bits_in:0"), g("tx_ofdm:0"), g("input:0"), g("output:0"), g("cost:0"), g("log_ber:0"), g("linear_ber:0"),
            g("conf_matrix:0"), g("tx_power:0"), g("noise_power:0"), g("iq_rx:0"), g("iq_tx:0"), g("ce_mean:0"),
            g("SNR:0"))
