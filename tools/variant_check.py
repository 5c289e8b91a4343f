#!/usr/bin/env python
"""Bitwise comparison of two tuning settings of the fused step (same inputs, same weights, N steps).

    python tools/variant_check.py --a 0=2 --b 0=9 [--frames 1170,53] [--steps 3]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from dl_ofdm_amd import _lib
    from dl_ofdm_amd.engine import RxDims, RxEngine
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", default="0=2")
    ap.add_argument("--b", default="0=9")
    ap.add_argument("--config", default="c2")
    ap.add_argument("--frames", default="1170,53")
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    lib = _lib.load()
    defaults = [lib.dccn_get_tuning(k) for k in range(lib.dccn_tuning_count())]

    def tune(spec):
        for k, v in enumerate(defaults):
            lib.dccn_set_tuning(k, v)
        for kv in filter(None, spec.split(",")):
            k, v = kv.split("=")
            assert lib.dccn_set_tuning(int(k), int(v)) == 0

    c = bench.CONFIGS[args.config]
    dims = RxDims(S=7, kin=c["nfft"] + c["cp"], F=c["F"], D=c["D"], nbits=c["nbits"])
    ok = True
    for frames in [int(f) for f in args.frames.split(",")]:
        outs = []
        for spec in (args.a, args.b):
            tune(spec)
            eng = RxEngine(dims, frames, train=True, seed=3, want_prob=True)
            g = torch.Generator(device="cuda")
            g.manual_seed(7)
            eng.x.copy_(torch.randn(eng.x.shape, generator=g, device="cuda"))
            eng.bits.copy_(torch.randint(0, 2, eng.bits.shape, generator=g, device="cuda", dtype=torch.int32))
            for _ in range(args.steps):
                eng.train_step()
            torch.cuda.synchronize()
            outs.append((eng.params.clone(), eng.dz.clone(), eng.prob.clone(), eng.grads.clone(), eng.metrics()))
        a, b = outs
        same = all(torch.equal(x, y) for x, y in zip(a[:4], b[:4]))
        dmax = max(float((x - y).abs().max()) for x, y in zip(a[:4], b[:4]))
        print("frames %d: %s vs %s bitwise %s (max |diff| %.3g) loss %.7f / %.7f" % (
            frames, args.a, args.b, same, dmax, a[4]["loss"] if "loss" in a[4] else a[4].get("ce_mean", 0),
            b[4]["loss"] if "loss" in b[4] else b[4].get("ce_mean", 0)), flush=True)
        ok = ok and same
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
