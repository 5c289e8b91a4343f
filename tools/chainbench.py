#!/usr/bin/env python
"""Independent equaliser training chains next to each other on ONE GPU from ONE process: G host threads, each with its own
HIP stream, trainer, generator and epoch loop (dl_ofdm_amd/receiver_mp.py DeviceEpochLoop) -- the chains the reference driver
starts as OS processes (dev/py/run_local_ofdm.py:61-118).  Prints one JSON line per (variant, G): wall ms per chain step and the
aggregate OFDM symbols/s, next to the G = 1 figure of the same process.

    python tools/chainbench.py [--chains 1 2 4] [--steps 400] [--what loop step]
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bench_groups(args):
    import torch
    from dl_ofdm_amd import ofdm, receiver as R, receiver_mp as M
    from dl_ofdm_amd.engine import glorot_init
    from dl_ofdm_amd.equalizer_group import EqualizerChainGroup
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    base = None
    for spec in ["1x1"] + [g for g in args.groups.replace("+", ",").split(",") if g != "1x1"]:
        T, G = (int(v) for v in spec.split("x"))
        groups, streams = [], []
        for t in range(T):
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                fl = [M.Flags(nbits=args.mods[(t * G + i) % len(args.mods)], nfilter=64, channel=args.channel, device_data=True,
                              seed=10 + t * G + i, token="cb%d_%d" % (t, i), save_dir="/tmp/dccn_chainbench/") for i in range(G)]
                rx = [glorot_init(R.rx_dims(F, ofdm.ofdm_tx(F)), 1 + i) for i, F in enumerate(fl)]
                grp = EqualizerChainGroup(fl, rx)
                for c in grp.chains:
                    c.begin_epoch()
            st.synchronize()
            groups.append(grp); streams.append(st)
        steps = groups[0].chains[0].steps
        n = steps * max(1, args.steps // steps)
        bar = threading.Barrier(T + 1)
        host = [0.0] * T

        def work(t):
            grp, st = groups[t], streams[t]
            act = grp.chains
            fn = (lambda i: act[0].loop.step()) if G == 1 else (lambda i: grp.step(act, i % steps))
            with torch.cuda.stream(st):
                for i in range(steps):
                    fn(i)
                st.synchronize()
                bar.wait()
                t0 = time.perf_counter()
                for i in range(n):
                    fn(i)
                host[t] = time.perf_counter() - t0
                st.synchronize()
            bar.wait()

        ths = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        for th in ths:
            th.start()
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        wall = time.perf_counter() - t0
        for th in ths:
            th.join()
        sym = T * G * n * 73 * 7 / wall
        if base is None:
            base = sym
        rec = {"what": "epoch-loop", "threads_x_group": spec, "chains": T * G, "steps": n, "ms_per_round": round(wall / n * 1e3, 4),
               "symbols_per_s": round(sym), "vs_one_chain": round(sym / base, 3),
               "host_issue_ms_per_round_max": round(max(host) / n * 1e3, 4)}
        print(json.dumps(rec), flush=True)
        if args.out:
            open(args.out, "a").write(json.dumps(rec) + "\n")
        del groups
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--what", nargs="+", default=["loop", "step"])
    ap.add_argument("--channel", default="mixRayleigh")
    ap.add_argument("--mods", type=int, nargs="+", default=[1, 2, 3, 4], help="nbits of chain i = mods[i %% len(mods)]")
    ap.add_argument("--out", default="")
    ap.add_argument("--graph", type=int, default=0, help="1: the loop replays the step as a hipGraph (one host call instead of 21 launches)")
    ap.add_argument("--prio", type=int, default=0, help="this many of the chains (the first ones) run on HIGH-priority streams")
    ap.add_argument("--groups", default="", help="e.g. 2x2,4x1,2x4: T host threads (a HIP stream each) x G chains per launch sequence "
                                                 "(dl_ofdm_amd.equalizer_group): aggregate symbols/s of T*G chains")
    args = ap.parse_args()
    if args.groups:
        return bench_groups(args)
    import numpy as np
    import torch
    from dl_ofdm_amd import ofdm, receiver as R, receiver_mp as M
    from dl_ofdm_amd.datagen import DeviceDataGen
    from dl_ofdm_amd.engine import glorot_init
    from dl_ofdm_amd.equalizer import EqualizerTrainer
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    gmax = max(args.chains)
    chains = []
    for g in range(gmax):
        nb = args.mods[g % len(args.mods)]
        F = M.Flags(nbits=nb, channel=args.channel, nfilter=64, device_data=True, seed=10 + g, step_graph=bool(args.graph))
        o = ofdm.ofdm_tx(F)
        st = torch.cuda.Stream(device=dev, priority=-1 if g < args.prio else 0)
        with torch.cuda.stream(st):
            tr = EqualizerTrainer(F, o, glorot_init(R.rx_dims(F, o), 1 + g), device="cuda", seed=1 + g)
            gen = DeviceDataGen(F, o, device=tr.device, seed=1 + g, mobile=False, mix=False)
            B = F.batch_size // F.nsymbol
            pl = tr.resident(B)
            rs = np.random.RandomState(g)
            loop = M.DeviceEpochLoop(F, o, tr, gen, pl, 197)
            loop.begin_epoch(rs.choice(M.TRAIN_SNR_GRID, [197, B], p=M.TRAIN_SNR_PROB))
            x = torch.randn(B, 7, 80, 2, device="cuda")
            bits = torch.randint(0, 2, (B, o.frame_size, nb), dtype=torch.int32, device="cuda")
            pl2 = tr._plan(B)
            pl2.set_batch(x, bits)
        st.synchronize()
        chains.append(dict(stream=st, loop=loop.step, step=lambda pl2=pl2: pl2.run(True, False), B=B))
    torch.cuda.synchronize()
    out = open(args.out, "a") if args.out else None
    base = {}
    for what in args.what:
        for G in args.chains:
            bar = threading.Barrier(G + 1)
            host = [0.0] * G
            done = [0.0] * G

            def work(g):
                c = chains[g]
                fn = c[what]
                with torch.cuda.stream(c["stream"]):
                    for _ in range(40):
                        fn()
                    c["stream"].synchronize()
                    bar.wait()
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        fn()
                    host[g] = time.perf_counter() - t0
                    c["stream"].synchronize()
                    done[g] = time.perf_counter() - t0
                bar.wait()

            ths = [threading.Thread(target=work, args=(g,)) for g in range(G)]
            for t in ths:
                t.start()
            bar.wait()
            t0 = time.perf_counter()
            bar.wait()
            wall = time.perf_counter() - t0
            for t in ths:
                t.join()
            B = chains[0]["B"]
            rec = {"what": what, "chains": G, "frames": B, "steps_per_chain": args.steps,
                   "ms_per_group_step": round(wall / args.steps * 1e3, 4),
                   "symbols_per_s": round(G * args.steps * B * 7 / wall),
                   "host_issue_ms_per_step_max": round(max(host) / args.steps * 1e3, 4),
                   "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default"), "high_priority_chains": args.prio,
                   "ms_per_step_by_chain": [round(d / args.steps * 1e3, 4) for d in done],
                   "stream_priority_range": list(torch.cuda.Stream.priority_range()) if hasattr(torch.cuda.Stream, "priority_range") else None}
            if G == 1:
                base[what] = rec["symbols_per_s"]
            if what in base:
                rec["vs_one_chain"] = round(rec["symbols_per_s"] / base[what], 3)
            line = json.dumps(rec)
            print(line, flush=True)
            if out:
                out.write(line + "\n")


if __name__ == "__main__":
    main()
