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
from ._lib import AdamHParams, check
from .complex import VariableStore
from .model import OfdmDenseRx, equalizer_ofdm, ofdm_dense_rx

EQ_REG_COEFF = 1e-3          # ofdmreceiver_np_mp.py:321
REG_L2 = 0.01                # tf.keras.regularizers.l2(l=0.01) on every dense kernel/bias (model.py:371-462)
SUPPORTED_OPTS = (0, 9, 10)  # the variants that build ``equalizer_ofdm`` (ofdmreceiver_np_mp.py:285,301-304)


class EqualizerTrainer:
    """``session.run([train_op, power_tx, ce_mean, berlin, chan_rms], feed)`` of the equaliser harness.

    rx_params: the trained basic receiver (name -> array, engine layout); FLAGS/ofdmobj as in the
    reference.  ``train_step(x, bits[, chan_gt])`` / ``eval_step(x, bits)`` take the raw `tx_ofdm` batch
    [frames, n_sym, n_sc, 2] (host or device) and return a metrics dict."""

    def __init__(self, FLAGS, ofdmobj, rx_params: Dict[str, np.ndarray], device="cuda", seed: int = 1,
                 lr0: Optional[float] = None):
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
        self.adam_state = torch.tensor([0.0, 0.9, 0.999, 0.0], dtype=torch.float32, device=self.device)
        self.last: dict = {}

    # ---- arena ---------------------------------------------------------------------------------
    def _flatten(self):
        f32 = dict(dtype=torch.float32, device=self.device)
        sizes = [self.store.tensor(n).numel() for n in self.names]
        total = int(sum(sizes))
        self.n_params = total
        self.params, self.grads = torch.empty(total, **f32), torch.zeros(total, **f32)
        self.adam_m, self.adam_v = torch.zeros(total, **f32), torch.zeros(total, **f32)
        self.reg_coef = torch.zeros(total, **f32)
        self.layout, o = {}, 0
        for n, sz in zip(self.names, sizes):
            p = self.store.tensor(n)
            self.params[o:o + sz].copy_(p.detach().reshape(-1))
            p.data = self.params[o:o + sz].view(p.shape)              # parameter now aliases the arena
            p.grad = self.grads[o:o + sz].view(p.shape)               # autograd accumulates in place
            if "/dense" in n:
                self.reg_coef[o:o + sz] = EQ_REG_COEFF * 2.0 * REG_L2
            self.layout[n] = (o, tuple(p.shape))
            o += sz

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
        """ofdmreceiver_np_mp.py:309-317 monitor: MSE between the layer-normalised true channel and the
        layer-normalised estimate (no gradient: it is fetched, never part of total_loss)."""
        with torch.no_grad():
            g = torch.view_as_real(torch.as_tensor(chan_gt).to(self.device).to(torch.complex64)).contiguous()
            a = ops.layer_norm(g)
            b = ops.layer_norm(torch.view_as_real(chest).contiguous())
            return ((a - b) ** 2).mean()

    def _adam_step(self):
        check(self.lib.dccn_adam_tf_step(self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(),
                                         self.adam_v.data_ptr(), self.reg_coef.data_ptr(), None,
                                         self.adam_state.data_ptr(), self.hp, self.n_params,
                                         C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
              "dccn_adam_tf_step")

    def train_step(self, x, bits, chan_gt=None) -> dict:
        self.grads.zero_()
        ce, mbuf, tx_power, snr_db, chest, _, _ = self._forward(x, bits)
        ce.backward()
        self._adam_step()
        rms = self.chan_rms(chest, chan_gt) if chan_gt is not None else None
        return self._metrics(mbuf, tx_power, rms)

    @torch.no_grad()
    def eval_step(self, x, bits) -> dict:
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
