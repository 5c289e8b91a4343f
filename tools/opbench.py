#!/usr/bin/env python
"""Launch one operator of the receiver step repeatedly (for rocprofv3 --pmc / --kernel-trace runs).

    python tools/opbench.py dense_fwd --iters 50 [--config c2]
Prints the HIP-event average per launch."""
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
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--tune", default="", help="dccn_set_tuning pairs, e.g. 0=2,1=1,4=3")
    ap.add_argument("--bench-plan", action="store_true",
                    help="the engine bench.py times: output:0 / z / dfft / the summed dense gradient not materialised")
    args = ap.parse_args()
    from dl_ofdm_amd import _lib
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        assert _lib.load().dccn_set_tuning(int(k), int(v)) == 0
    c = bench.CONFIGS[args.config]
    dims = RxDims(S=7, kin=c["nfft"] + c["cp"], F=c["F"], D=c["D"], nbits=c["nbits"])
    if args.bench_plan:
        eng = RxEngine(dims, c["frames"], train=True, want_prob=False, want_tx_power=True, want_z=False, want_dfft=False,
                       want_grads=False)
    else:
        eng = RxEngine(dims, c["frames"], train=True)
    eng.x.normal_()
    eng.bits.random_(0, 2)
    eng.train_step()
    torch.cuda.synchronize()
    ops = op_launchers(eng)
    ops["step"] = (lambda: eng.train_step(), 0.0, "whole training step (eager launch sequence)")
    ops["step_pipe"] = (lambda: eng.train_step_pipelined(), 0.0, "pipelined training step (what bench.py times)")
    t = HipTimer()
    for name in args.ops:
        eng.drop_prefetch()
        fn = ops[name][0]
        for _ in range(5):
            fn()
        t.start(eng._stream())
        for _ in range(args.iters):
            fn()
        t.stop(eng._stream())
        print("%s: %.2f us/launch" % (name, t.elapsed_ms() * 1e3 / args.iters))


if __name__ == "__main__":
    main()
