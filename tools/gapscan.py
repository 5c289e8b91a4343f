#!/usr/bin/env python
"""Where does a C2 training step spend its time between and inside its launches, and what changes that?

Runs the headline step of bench.py under a list of conditions inside ONE process and prints one JSON line per condition:
ms per step as the median of R regions of K steps (the driver's region), of one long run, and the in-situ timeline
(dl_ofdm_amd/steptrace.py: per-launch duration, gap in front of it, shader clock).

    python tools/gapscan.py [--modes eager,graph,smi_poll,sysfs_poll,host_busy,ride2,sleepy] [--out file.jsonl]

Conditions:
  eager       stream launches (the shipped mode)
  graph       hipGraph replay of the same sequence
  ride2       tuning keys 18 + 15: three launches per step (C-Conv forward of the next batch on the optimizer launch)
  smi_poll    eager, while a child process runs `rocm-smi --showuse --showclocks --showpower --json` in a loop
  sysfs_poll  eager, while a thread reads gpu_busy_percent / pp_dpm_sclk / gpu_metrics 200 times per second
  host_busy   eager, while every other CPU of the box spins (the driver's harness shares the host)
  sleepy      eager, 20-step regions issued after 50 ms of GPU idleness each (clock ramp / power-state exit)
Process-level knobs (HSA_*, GPU_MAX_HW_QUEUES, HIP_FORCE_DEV_KERNARG ...) are environment variables: run the tool once per
setting, `--tag` labels the lines.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="eager,graph,ride2,smi_poll,sysfs_poll,host_busy,sleepy,eager")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--regions", type=int, default=9)
    ap.add_argument("--long", type=int, default=500)
    ap.add_argument("--tag", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--no-trace", action="store_true")
    args = ap.parse_args()

    import torch
    import bench
    from dl_ofdm_amd import _lib
    from dl_ofdm_amd.engine import HipTimer, RxDims, RxEngine
    from dl_ofdm_amd.steptrace import trace_steps
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    c = bench.CONFIGS["c2"]
    dims = RxDims(S=7, kin=c["nfft"] + c["cp"], F=c["F"], D=c["D"], nbits=c["nbits"])
    defaults = [lib.dccn_get_tuning(k) for k in range(lib.dccn_tuning_count())]

    def make_engine():
        eng = RxEngine(dims, c["frames"], device=dev, train=True, seed=1, want_prob=True, want_tx_power=True, want_z=False,
                       want_dfft=False, want_grads=False)
        g = torch.Generator(device=dev)
        g.manual_seed(1234)
        eng.x.copy_(torch.randn(eng.x.shape, generator=g, device=dev))
        eng.bits.copy_(torch.randint(0, 2, eng.bits.shape, generator=g, device=dev, dtype=torch.int32))
        return eng

    def measure(eng, graph, idle_s=0.0):
        step = lambda: eng.train_step_pipelined(graph=graph)      # noqa: E731
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.3:
            for _ in range(50):
                step()
            torch.cuda.synchronize(dev)
        regs, evs = [], []
        timer = HipTimer()
        for _ in range(args.regions):
            if idle_s:
                time.sleep(idle_s)
            torch.cuda.synchronize(dev)
            timer.start(eng._stream())
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            timer.stop(eng._stream())
            torch.cuda.synchronize(dev)
            regs.append((time.perf_counter() - t0) / args.steps * 1e3)
            evs.append(timer.elapsed_ms() / args.steps)
        timer.start(eng._stream())
        for _ in range(args.long):
            step()
        timer.stop(eng._stream())
        long_ms = timer.elapsed_ms() / args.long
        res = {"region_ms_median": round(statistics.median(regs), 5), "region_ms": [round(r, 5) for r in regs],
               "region_event_ms_median": round(statistics.median(evs), 5), "long_ms": round(long_ms, 5)}
        if not graph and not args.no_trace:
            res["timeline"] = trace_steps(step, dev, ring=16, bursts=8, lead=300)
        return res

    out = open(args.out, "a") if args.out else None

    def emit(mode, res):
        res = dict(mode=mode, tag=args.tag, **res)
        line = json.dumps(res)
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()

    for mode in args.modes.split(","):
        for k, v in enumerate(defaults):
            lib.dccn_set_tuning(k, v)
        stop = threading.Event()
        helpers, child = [], None
        if mode == "ride2":
            lib.dccn_set_tuning(18, 1)
            lib.dccn_set_tuning(15, 1)
        eng = make_engine()
        if mode == "smi_poll":
            child = subprocess.Popen(["bash", "-c", "while true; do rocm-smi --showuse --showclocks --showpower --json >/dev/null 2>&1; done"],
                                     start_new_session=True)
            time.sleep(1.0)
        elif mode == "sysfs_poll":
            import glob
            files = []
            for d in glob.glob("/sys/class/drm/card[0-9]*/device"):
                for f in ("gpu_busy_percent", "pp_dpm_sclk", "gpu_metrics", "mem_busy_percent"):
                    if os.path.exists(os.path.join(d, f)):
                        files.append(os.path.join(d, f))

            def poll():
                while not stop.is_set():
                    for f in files:
                        try:
                            open(f, "rb").read()
                        except Exception:
                            pass
                    time.sleep(0.005)
            helpers.append(threading.Thread(target=poll, daemon=True))
        elif mode == "host_busy":
            n = max(1, bench.effective_cpus() - 1)
            child = subprocess.Popen([sys.executable, "-c",
                                      "import multiprocessing as m,time\n"
                                      "def f():\n    x=0\n    while True: x+=1\n"
                                      "ps=[m.Process(target=f,daemon=True) for _ in range(%d)]\n"
                                      "[p.start() for p in ps]\ntime.sleep(600)" % n], start_new_session=True)
            time.sleep(0.5)
        for h in helpers:
            h.start()
        try:
            res = measure(eng, graph=(mode == "graph"), idle_s=0.05 if mode == "sleepy" else 0.0)
        finally:
            stop.set()
            if child is not None:
                import signal
                try:
                    os.killpg(child.pid, signal.SIGKILL)          # the child leads its own session: exactly our helpers
                    child.wait(timeout=5)
                except Exception:
                    pass
        res["plan"] = bench.pipeline_plan(eng)
        emit(mode, res)
        eng.close_graph()
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
