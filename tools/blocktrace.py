#!/usr/bin/env python
"""Per-block timeline of the fused backward launch (rx_bwd.h) from the instrumented build:

    make -C dl_ofdm_amd/csrc trace
    DCCN_LIB_PATH=abl/libdccn_trace.so python tools/blocktrace.py [--config c2] [--out gpurun_out/trace.txt]

Thread 0 of every block stamps s_memrealtime (100 MHz) at entry, after the dX k-loop, and at exit, plus its hardware
id (XCC, SE, CU).  Prints: launch span, per-role start/finish histograms, per-CU busy time and the tail (time between
the median and the last block finish)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bench
    from dl_ofdm_amd import _lib
    from dl_ofdm_amd.engine import RxDims, RxEngine, op_launchers
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--out", default="")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    lib = _lib.load()
    raw = C.CDLL(_lib.LIB_PATH)
    if not hasattr(raw, "dccn_debug_set_trace"):
        raise SystemExit("not a trace build: set DCCN_LIB_PATH=abl/libdccn_trace.so (make -C dl_ofdm_amd/csrc trace)")
    raw.dccn_debug_set_trace.argtypes = [C.c_void_p]
    c = bench.CONFIGS[args.config]
    dims = RxDims(S=7, kin=c["nfft"] + c["cp"], F=c["F"], D=c["D"], nbits=c["nbits"])
    eng = RxEngine(dims, c["frames"], train=True, want_z=False, want_dfft=False)
    eng.x.normal_()
    eng.bits.random_(0, 2)
    for _ in range(3):
        eng.train_step()
    torch.cuda.synchronize()
    fn = op_launchers(eng)["rx_backward"][0]
    nblk = 4096
    buf = torch.zeros(nblk * 4, dtype=torch.int64, device="cuda")
    lines = []
    for rep in range(args.reps):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        buf.zero_()
        assert raw.dccn_debug_set_trace(C.c_void_p(buf.data_ptr())) == 0
        fn()
        torch.cuda.synchronize()
        raw.dccn_debug_set_trace(C.c_void_p(0))
        t = buf.cpu().numpy().reshape(nblk, 4)
        n = int((t[:, 0] != 0).sum())
        t = t[:n]
        t0 = t[:, 0].min()
        start, mid, end = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01, (t[:, 2] - t0) * 0.01      # us
        role = (t[:, 3] >> 48) & 0xf
        xcc = (t[:, 3] >> 32) & 0xf
        hw = t[:, 3] & 0xffffffff
        cu, se = (hw >> 8) & 0xf, (hw >> 13) & 0x7
        cuid = xcc * 64 + se * 16 + cu
        span = end.max()
        lines.append("rep %d: %d blocks, span %.2f us (first start -> last end)" % (rep, n, span))
        for r, name in ((2, "dX+dWeff"), (3, "dW item")):
            m = role == r
            if not m.any():
                continue
            dur = end[m] - start[m]
            lines.append("  %-9s n=%4d  start: median %.2f  max %.2f | duration: min %.2f  median %.2f  max %.2f | end: median %.2f  p90 %.2f  max %.2f"
                         % (name, m.sum(), np.median(start[m]), start[m].max(), dur.min(), np.median(dur), dur.max(),
                            np.median(end[m]), np.percentile(end[m], 90), end[m].max()))
            if r == 2:
                kl = mid[m] - start[m]
                ep = end[m] - mid[m]
                lines.append("            k-loop: median %.2f  max %.2f | epilogue: min %.2f  median %.2f  max %.2f"
                             % (np.median(kl), kl.max(), ep.min(), np.median(ep), ep.max()))
        # per CU: number of dX blocks, last finish, busy = union of block intervals
        ids = np.unique(cuid)
        last = np.array([end[cuid == i].max() for i in ids])
        ndx = np.array([int(((cuid == i) & (role == 2)).sum()) for i in ids])
        nall = np.array([int((cuid == i).sum()) for i in ids])
        lines.append("  CUs seen: %d | blocks per CU: min %d  median %d  max %d | dX blocks per CU: %s"
                     % (len(ids), nall.min(), int(np.median(nall)), nall.max(),
                        ", ".join("%d CUs x %d" % (int((ndx == k).sum()), k) for k in sorted(set(ndx)))))
        lines.append("  last finish per CU: min %.2f  median %.2f  p90 %.2f  max %.2f  -> tail (max - median) %.2f us = %.0f %% of the span"
                     % (last.min(), np.median(last), np.percentile(last, 90), last.max(), last.max() - np.median(last),
                        100 * (last.max() - np.median(last)) / span))
        for k in sorted(set(ndx)):
            lines.append("    CUs with %d dX blocks: last finish median %.2f  max %.2f" % (k, np.median(last[ndx == k]), last[ndx == k].max()))
        hist, edges = np.histogram(end, bins=12, range=(0, span))
        lines.append("  finish-time histogram (12 bins over the span): " + " ".join(str(int(h)) for h in hist))
    txt = "\n".join(lines)
    print(txt)
    if args.out:
        open(args.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
