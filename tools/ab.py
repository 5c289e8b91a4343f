#!/usr/bin/env python
"""A/B timing of tuning settings inside ONE process: the settings are measured in alternating rounds (A B A B ...),
so clock drift and box-to-box differences cancel; prints the median and minimum per setting.

    python tools/ab.py --tunes "0=2;0=9" [--what step,dense_tail] [--rounds 7] [--iters 300]
    DCCN_LIB_PATH=abl/libdccn_x.so python tools/ab.py ...       # another build of the library
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from dl_ofdm_amd import _lib
    from dl_ofdm_amd.engine import HipTimer, RxDims, RxEngine, op_launchers
    ap = argparse.ArgumentParser()
    ap.add_argument("--tunes", default="")
    ap.add_argument("--what", default="step")
    ap.add_argument("--config", default="c2")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--graph", action="store_true")
    args = ap.parse_args()
    lib = _lib.load()
    defaults = [lib.dccn_get_tuning(k) for k in range(lib.dccn_tuning_count())]
    specs = args.tunes.split(";") if args.tunes else [""]

    def tune(spec):
        for k, v in enumerate(defaults):
            lib.dccn_set_tuning(k, v)
        for kv in filter(None, spec.split(",")):
            k, v = kv.split("=")
            assert lib.dccn_set_tuning(int(k), int(v)) == 0, kv

    c = bench.CONFIGS[args.config]
    dims = RxDims(S=7, kin=c["nfft"] + c["cp"], F=c["F"], D=c["D"], nbits=c["nbits"])
    engines = {}
    for spec in specs:                      # one engine per setting (workspace sizes may depend on the tuning)
        tune(spec)
        eng = RxEngine(dims, c["frames"], train=True, want_prob=False, want_tx_power=True, want_z=False, want_dfft=False,
                       want_grads=False)       # the plan bench.py times
        eng.x.normal_()
        eng.bits.random_(0, 2)
        engines[spec] = eng
    t = HipTimer()
    res = {(w, s): [] for w in args.what.split(",") for s in specs}
    for r in range(args.rounds + 1):
        for spec in specs:
            tune(spec)
            eng = engines[spec]
            ops = op_launchers(eng)
            for w in args.what.split(","):
                eng.drop_prefetch()         # (a pipelined measurement may have left a normalised batch behind)
                if w == "step":
                    fn = lambda: eng.train_step(graph=args.graph)
                elif w == "step_pipe":
                    fn = lambda: eng.train_step_pipelined(graph=args.graph)
                else:
                    fn = ops[w][0]
                for _ in range(20):
                    fn()
                t.start(eng._stream())
                for _ in range(args.iters):
                    fn()
                t.stop(eng._stream())
                if r > 0:                   # round 0 = warm-up (clocks, code objects)
                    res[(w, spec)].append(t.elapsed_ms() * 1e3 / args.iters)
    for (w, spec), v in res.items():
        print("%-18s %-24s median %.2f us  min %.2f  max %.2f" % (w, spec or "(default)", statistics.median(v), min(v), max(v)),
              flush=True)


if __name__ == "__main__":
    main()
