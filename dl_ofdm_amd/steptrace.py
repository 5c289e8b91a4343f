"""In-situ timeline of the receiver step (include/dccn.h ``dccn_step_trace_*``).

Every instrumented launch of ``dccn_rx_train_step`` / ``dccn_rx_eval_step`` leaves, per workgroup, the 100 MHz
``s_memrealtime`` and the shader-cycle ``s_memtime`` counters at entry and exit in a device ring buffer.  From those this
module derives what no same-kernel timing loop can show: how long each launch takes *inside* the step, how long the chip
idles between one launch's last workgroup and the next launch's first, and which clock the CUs really ran at.

The reference has no counterpart on its hot path (dev/py/ofdmreceiver_np.py:234 runs ``train_op`` without TF1 step stats);
this is measurement infrastructure for ``bench.py`` (``step.boundaries``) and ``tools/``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import check

SLOT_NAMES = {1: "cconv_fwd", 2: "dense_fwd_tail", 3: "tail", 4: "backward", 5: "cconv_bwd_w", 6: "optimizer"}
WALL_HZ = 100e6        # s_memrealtime (hipDeviceAttributeWallClockRate = 100 000 kHz on gfx950)


class StepTrace:
    def __init__(self, device="cuda", ring: int = 16):
        self.lib = _lib.load()
        self.device = torch.device(device)
        L, B, W = C.c_int(0), C.c_int(0), C.c_int(0)
        self.lib.dccn_step_trace_geometry(C.byref(L), C.byref(B), C.byref(W))
        self.launches, self.blocks, self.words = L.value, B.value, W.value
        self.ring = int(ring)
        nbytes = self.lib.dccn_step_trace_bytes(self.ring)
        assert nbytes == self.ring * self.launches * self.blocks * self.words * 8
        self.buf = torch.zeros(nbytes // 8, dtype=torch.int64, device=self.device)
        self.enabled = False

    def enable(self):
        """Clear the ring and start recording (the step counter restarts at 0)."""
        self.buf.zero_()
        torch.cuda.synchronize(self.device)
        check(self.lib.dccn_step_trace_enable(C.c_void_p(self.buf.data_ptr()), self.buf.numel() * 8, self.ring),
              "dccn_step_trace_enable")
        self.enabled = True

    def disable(self):
        check(self.lib.dccn_step_trace_enable(C.c_void_p(0), 0, 0), "dccn_step_trace_enable")
        self.enabled = False

    def __del__(self):
        try:
            if self.enabled:
                self.disable()
        except Exception:
            pass

    def collect(self) -> List[Dict[int, dict]]:
        """Synchronise and return the recorded steps in issue order (at most `ring`, oldest first): per step a dict
        slot -> {start, end (100 MHz ticks: first workgroup in, last workgroup out), blocks, sclk_mhz}."""
        torch.cuda.synchronize(self.device)
        n = int(self.lib.dccn_step_trace_steps())
        t = self.buf.cpu().numpy().view(np.uint64).reshape(self.ring, self.launches, self.blocks, self.words)
        steps = []
        for st in range(max(0, n - self.ring), n):
            e = t[st % self.ring]
            rec = {}
            for slot in range(self.launches):
                w0 = e[slot, :, 0]
                m = w0 != 0
                if not m.any():
                    continue
                w0, c0, w1, c1 = (e[slot, m, k].astype(np.int64) for k in range(4))
                done = w1 != 0
                dur = (w1 - w0)[done]
                cyc = (c1 - c0)[done]
                long_ = dur >= 300                         # >= 3 us of wall time: +-1 tick is < 0.7 % of the ratio
                sclk = float(np.median(cyc[long_] / dur[long_]) * WALL_HZ / 1e6) if long_.any() else None
                rec[slot] = dict(start=int(w0.min()), end=int(w1[done].max()) if done.any() else int(w0.max()),
                                 blocks=int(m.sum()), sclk_mhz=sclk,
                                 block_us_median=float(np.median(dur)) * 1e6 / WALL_HZ if done.any() else None)
            steps.append(rec)
        return steps


def summarise(bursts: List[List[Dict[int, dict]]]) -> dict:
    """Per launch slot: median / p90 in-situ duration, median / p90 idle gap in front of it (previous launch's last
    workgroup out -> this launch's first workgroup in, previous launch = whatever ran before it on the stream, across step
    borders too), the shader clock its long workgroups saw; plus the step period (same slot, consecutive steps)."""
    dur: Dict[int, list] = {}
    gap: Dict[int, list] = {}
    clk: Dict[int, list] = {}
    period: List[float] = []
    tick_us = 1e6 / WALL_HZ
    for steps in bursts:
        prev_end = None
        prev_first = None
        for rec in steps:
            order = sorted(rec, key=lambda s: rec[s]["start"])
            if not order:
                continue
            first = rec[order[0]]["start"]
            if prev_first is not None:
                period.append((first - prev_first) * tick_us)
            prev_first = first
            for s in order:
                r = rec[s]
                dur.setdefault(s, []).append((r["end"] - r["start"]) * tick_us)
                if prev_end is not None:
                    gap.setdefault(s, []).append((r["start"] - prev_end) * tick_us)
                if r["sclk_mhz"]:
                    clk.setdefault(s, []).append(r["sclk_mhz"])
                prev_end = r["end"]
    out = {"launches": [], "samples": len(period)}
    tot_d = tot_g = 0.0
    for s in sorted(dur):
        d, g = np.array(dur[s]), np.array(gap.get(s, [np.nan]))
        row = {"slot": s, "name": SLOT_NAMES.get(s, "slot%d" % s), "us": round(float(np.median(d)), 2),
               "us_p90": round(float(np.percentile(d, 90)), 2), "gap_before_us": round(float(np.nanmedian(g)), 2),
               "gap_before_us_p90": round(float(np.nanpercentile(g, 90)), 2),
               "sclk_mhz": round(float(np.median(clk[s])), 0) if s in clk else None}
        tot_d += row["us"]
        tot_g += row["gap_before_us"]
        out["launches"].append(row)
    out["sum_launch_us"] = round(tot_d, 2)
    out["sum_gap_us"] = round(tot_g, 2)
    if period:
        p = np.array(period)
        out["period_us"] = round(float(np.median(p)), 2)
        out["period_us_p90"] = round(float(np.percentile(p, 90)), 2)
    return out


def trace_steps(step_fn, device="cuda", ring: int = 16, bursts: int = 14, lead: int = 500,
                trace: Optional[StepTrace] = None) -> dict:
    """Run `bursts` bursts of lead + ring calls of `step_fn` with the timeline on; every burst contributes its last `ring`
    steps, so bursts * (ring - 1) step periods are sampled.  `lead` steps run first in every burst: reading a burst's stamps
    back leaves the GPU idle for milliseconds, after which it restarts ~10 % below its sustained clock and needs tens of
    milliseconds of work to recover (profiles/r04_gap.md) -- 500 C2 steps are 40 ms."""
    tr = trace or StepTrace(device, ring)
    out = []
    try:
        for _ in range(bursts):
            tr.enable()
            for _ in range(lead + tr.ring):
                step_fn()
            out.append(tr.collect())
    finally:
        tr.disable()
    res = summarise(out)
    res["steps_traced"] = sum(len(b) for b in out)
    return res
