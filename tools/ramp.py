#!/usr/bin/env python
"""How long does the GPU need to get back to speed after the stream has been idle?

For each idle time (ms) the C2 training step is issued N times right after `time.sleep(idle)`; the in-situ timeline
(dl_ofdm_amd/steptrace.py) gives, per step since wake-up, the step period and the shader clock its launches saw.
Prints one JSON line per idle time: period_us[i], sclk_mhz[i] for the first steps, and the average of steps 0-19 (what a
20-step timed region issued right after such a pause would report).

    python tools/ramp.py [--idles 0,1,5,20,50,200,1000] [--steps 60] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--idles", default="0,1,5,20,50,200,1000")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from dl_ofdm_amd.engine import RxDims, RxEngine
    from dl_ofdm_amd.steptrace import StepTrace
    dev = torch.device("cuda", 0)
    c = bench.CONFIGS["c2"]
    dims = RxDims(S=7, kin=c["nfft"] + c["cp"], F=c["F"], D=c["D"], nbits=c["nbits"])
    eng = RxEngine(dims, c["frames"], device=dev, train=True, seed=1, want_prob=True, want_tx_power=True, want_z=False,
                   want_dfft=False, want_grads=False)
    eng.x.normal_()
    eng.bits.random_(0, 2)
    step = lambda: eng.train_step_pipelined()      # noqa: E731
    tr = StepTrace(dev, ring=args.steps)
    out = open(args.out, "a") if args.out else None
    for idle in [float(v) for v in args.idles.split(",")]:
        per, clk = [], []
        for _ in range(args.reps):
            t0 = time.perf_counter()                 # busy first: every repetition starts from the same (ramped) state
            while time.perf_counter() - t0 < 0.3:
                for _ in range(50):
                    step()
                torch.cuda.synchronize(dev)
            tr.enable()
            time.sleep(idle * 1e-3)
            for _ in range(args.steps):
                step()
            steps = tr.collect()
            firsts = [min(r[s]["start"] for s in r) for r in steps if r]
            per.append([(b - a) * 0.01 for a, b in zip(firsts[:-1], firsts[1:])])
            clk.append([float(np.mean([r[s]["sclk_mhz"] for s in r if r[s]["sclk_mhz"]])) for r in steps if r])
        tr.disable()
        p = np.median(np.array(per), axis=0)
        k = np.median(np.array(clk), axis=0)
        rec = {"idle_ms": idle, "first20_avg_us": round(float(p[:20].mean()), 2), "steady_us": round(float(np.median(p[-20:])), 2),
               "period_us": [round(float(v), 1) for v in p[:40]], "sclk_mhz": [round(float(v)) for v in k[:40]]}
        line = json.dumps(rec)
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()


if __name__ == "__main__":
    main()
