#!/usr/bin/env python
"""Cache-state sensitivity of one operator launch: time it back to back (operands warm in L2), after a 64 MB sweep
(L2 of every XCD cleared, Infinity Cache still holds the operands) and after a 768 MB sweep (everything from HBM).

    python tools/coldbench.py dense_tail_fwd_bwd dense_bwd_slabs cconv_fwd cconv_bwd_w [--iters 40]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from dl_ofdm_amd.engine import HipTimer, RxDims, RxEngine, op_launchers
    ap = argparse.ArgumentParser()
    ap.add_argument("ops", nargs="+")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--config", default="c2")
    args = ap.parse_args()
    c = bench.CONFIGS[args.config]
    dims = RxDims(S=7, kin=c["nfft"] + c["cp"], F=c["F"], D=c["D"], nbits=c["nbits"])
    eng = RxEngine(dims, c["frames"], train=True, want_prob=True, want_tx_power=True, want_z=False)
    eng.x.normal_()
    eng.bits.random_(0, 2)
    eng.train_step()
    ops = op_launchers(eng)
    small = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    big = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")
    t = HipTimer()
    for name in args.ops:
        fn = ops[name][0]
        res = {}
        for mode, sweep in (("warm", None), ("L2 cleared", small), ("HBM", big)):
            tot = 0.0
            for i in range(args.iters + 3):
                if sweep is not None:
                    sweep.fill_(i & 255)
                torch.cuda.synchronize()
                t.start(eng._stream())
                fn()
                t.stop(eng._stream())
                ms = t.elapsed_ms()
                if i >= 3:
                    tot += ms
            res[mode] = tot * 1e3 / args.iters
        print("%-20s " % name + "  ".join("%s %.2f us" % (k, v) for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
