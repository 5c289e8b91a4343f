#!/usr/bin/env python
"""In-situ timeline (dl_ofdm_amd/steptrace.py) of the pipelined training step of one BASELINE config.

    python tools/steptl.py --config c3 [--tunes 13=3]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from dl_ofdm_amd import _lib, steptrace
    from dl_ofdm_amd.engine import RxDims, RxEngine
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--tunes", default="")
    ap.add_argument("--bursts", type=int, default=8)
    ap.add_argument("--lead", type=int, default=400)
    ap.add_argument("--prob", type=int, default=0, help="1: the step also writes output:0 (bench.py --store-prob)")
    a = ap.parse_args()
    lib = _lib.load()
    for kv in filter(None, a.tunes.split(",")):
        k, v = kv.split("=")
        assert lib.dccn_set_tuning(int(k), int(v)) == 0, kv
    c = bench.CONFIGS[a.config]
    dims = RxDims(S=7, kin=c["nfft"] + c["cp"], F=c["F"], D=c["D"], nbits=c["nbits"])
    eng = RxEngine(dims, c["frames"], train=True, want_prob=bool(a.prob), want_tx_power=True, want_z=False, want_dfft=False,
                   want_grads=False)       # bench.py's plan
    eng.x.normal_()
    eng.bits.random_(0, 2)
    for _ in range(50):
        eng.train_step_pipelined()
    res = steptrace.trace_steps(eng.train_step_pipelined, eng.device, ring=16, bursts=a.bursts, lead=a.lead)
    res["config"], res["tunes"], res["prob"] = a.config, a.tunes, a.prob
    print(json.dumps(res))


if __name__ == "__main__":
    main()
