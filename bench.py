#!/usr/bin/env python
"""bench.py -- OFDM symbols/s of one DCCN basic-receiver training step (fwd + bwd + Adam) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): OFDM symbols/sec fwd+bwd, DCCN QPSK N=64.  Workload (config[1]): QPSK, N=64, CP=16,
batch = 8192 OFDM symbols = 1170 frames of 7 symbols (dev/py/ofdmreceiver_np.py:196 `batch_size//nsymbol`),
F=nfilter=64, D=320 data cells per frame.  A "step" = R0 normalise -> R1 C-Conv -> R2 dense -> R3-R6 tail/
loss/BER -> full backward -> R7 TF-Adam on synthetic N(0,1) IQ and random bits already resident in HBM
(throughput is data independent).  Nothing inside the timed region is skipped or cached.

Multi-GPU: the path shards by independent units (SNR-sweep points / batches: SURVEY.md section 8e) -- every
rank runs its own full step on its own batch, no data-path collective; the only communication is the final
RCCL all-reduce of the BER/loss table, done once inside the timed region.  scaling = "weak".

One JSON line is printed by rank 0; see the task contract for the fields.  Extra objects:
  roofline      dominant kernel of the step vs the fp32 MFMA peak (157.3 TFLOP/s)
  cpu_baseline  the reference-equivalent CPU graph (oracle/torch_ref.py, literal TF-style conv3d) on the host
  kernels       per-operator average launch time (HIP events on the launch stream)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0

CONFIGS = {
    # name: (frames, nbits, nfft, cp, F, D)
    "c1": dict(frames=36, nbits=1, nfft=64, cp=16, F=64, D=320,
               workload="BPSK, N=64/CP=16, batch=256 OFDM symbols (36 frames x 7), fwd+bwd+Adam"),
    "c2": dict(frames=1170, nbits=2, nfft=64, cp=16, F=64, D=320,
               workload="QPSK, N=64/CP=16, batch=8192 OFDM symbols (1170 frames x 7), fwd+bwd+Adam"),
    "c3": dict(frames=1170, nbits=4, nfft=64, cp=16, F=64, D=320,
               workload="16-QAM, N=64/CP=16, batch=8192 OFDM symbols (1170 frames x 7), fwd+bwd+Adam"),
    "c8": dict(frames=1170, nbits=3, nfft=64, cp=16, F=64, D=320,
               workload="8-QAM, N=64/CP=16, batch=8192 OFDM symbols (1170 frames x 7), fwd+bwd+Adam"),
    "c4": dict(frames=585, nbits=2, nfft=1024, cp=72, F=1024, D=4000,
               workload="QPSK, N=1024/CP=72, batch=4096 OFDM symbols (585 frames x 7), fwd+bwd+Adam"),
}


def pipeline_plan(eng):
    """what a pipelined step launches under the library's current plan (dccn_rx_norm_rides_backward: 0 / 1)"""
    ride = int(getattr(eng, "_ride", 0))
    if ride == 1:
        return "4 launches per step: C-Conv fwd | dense fwd + tail | fused backward + R0 of the next batch | optimizer"
    return "4 launches per step: C-Conv fwd | dense fwd + tail | fused backward | optimizer + R0 of the next batch"


def step_flops(c):
    """Algorithmic FLOPs of one training step (SURVEY.md section 8d / BASELINE.md section 2)."""
    S, kin, F, D, b = 7, c["nfft"] + c["cp"], c["F"], c["D"], c["nbits"]
    m = 2 ** b
    L1, L2, L3, L4 = 8 * S * kin * F, 8 * S * F * D, 4 * D * m, 4 * D * (m + 2) * b
    return c["frames"] * (2 * L1 + 3 * (L2 + L3 + L4))


def effective_cpus():
    """CPUs this process may really use: affinity mask and cgroup v2 quota (a container can expose 128 cores while
    granting 16 CPUs of run time; running 128 threads on those only thrashes)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def hip_device_pci(dev):
    """PCI address 'dddd:bb:dd.f' of a torch device (the box's sysfs lists every GPU of the host, not just ours)."""
    try:
        import torch
        p = torch.cuda.get_device_properties(dev)
        return "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def _cards(pci=None):
    import glob
    devs = [d for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")) if os.path.exists(os.path.join(d, "pp_dpm_sclk"))]
    if pci:
        mine = [d for d in devs if os.path.basename(os.path.realpath(d)).lower() == pci.lower()]
        if mine:
            return mine
    return devs


def copy_bandwidth_tbs(dev):
    import torch
    try:
        a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize(dev)
        return round(10 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12, 3)
    except Exception:                                    # (diagnostic only)
        return None


def gpu_clock_state(pci=None):
    """sclk / mclk / fclk / socclk DPM state, performance level, power cap and draw of every amdgpu card the box exposes
    (sysfs, read-only; the container sees the host's /sys/class/drm).  `step.boundaries[*].sclk_mhz` is the clock the CUs
    really ran at (s_memtime / s_memrealtime inside the kernels); this is what the driver says it set."""
    import glob
    out = []
    for dev in _cards(pci):
        rec = {"card": os.path.basename(os.path.dirname(dev)), "pci": os.path.basename(os.path.realpath(dev))}
        for key in ("sclk", "mclk", "fclk", "socclk"):
            try:
                lines = [ln.strip() for ln in open(os.path.join(dev, "pp_dpm_" + key)).read().splitlines() if ln.strip()]
                act = [ln for ln in lines if ln.endswith("*")]
                rec[key] = (act[0].split(":")[1].strip(" *") if act else None)
                rec[key + "_levels"] = len(lines)
            except Exception:
                pass
        try:
            rec["perf_level"] = open(os.path.join(dev, "power_dpm_force_performance_level")).read().strip()
        except Exception:
            pass
        for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
            for key, fn, scale in (("power_cap_w", "power1_cap", 1e-6), ("power_w", "power1_average", 1e-6),
                                   ("power_in_w", "power1_input", 1e-6), ("temp_c", "temp1_input", 1e-3)):
                try:
                    rec[key] = round(int(open(os.path.join(hw, fn)).read().strip()) * scale, 1)
                except Exception:
                    pass
        out.append(rec)
    return out


def box_fingerprint(pci=None):
    """What distinguishes one box of the pool from another below the (identical) container image: kernel and amdgpu driver,
    firmware versions, VBIOS, partition modes, visible-VRAM (large BAR) size, PCIe link, and the runtime knobs in the
    environment.  Read-only sysfs; missing entries are simply absent."""
    import glob
    import platform
    rec = {"kernel": platform.release()}
    try:
        rec["amdgpu_version"] = open("/sys/module/amdgpu/version").read().strip()
    except Exception:
        pass
    rec["env"] = {k: v for k, v in os.environ.items() if k.split("_")[0] in ("HSA", "HIP", "GPU", "AMD", "ROCR", "ROCM", "DCCN", "NCCL", "RCCL")}
    cards = []
    all_cards = _cards(None)
    rec["gpus_on_host"] = len(all_cards)
    for dev in _cards(pci):
        c = {"card": os.path.basename(os.path.dirname(dev)), "pci": os.path.basename(os.path.realpath(dev))}
        for key, fn in (("vbios", "vbios_version"), ("compute_partition", "current_compute_partition"),
                        ("memory_partition", "current_memory_partition"), ("vram_total", "mem_info_vram_total"),
                        ("vis_vram_total", "mem_info_vis_vram_total"), ("link_speed", "current_link_speed"),
                        ("link_width", "current_link_width"), ("numa_node", "numa_node"), ("pci_id", "device"),
                        ("revision", "revision")):
            try:
                c[key] = open(os.path.join(dev, fn)).read().strip()
            except Exception:
                pass
        fw = {}
        for f in sorted(glob.glob(os.path.join(dev, "fw_version", "*_fw_version"))):
            try:
                fw[os.path.basename(f)[:-len("_fw_version")]] = open(f).read().strip()
            except Exception:
                pass
        if fw:
            c["fw"] = {k: fw[k] for k in fw if k in ("mec", "mec2", "smc", "sdma", "rlc", "pfp", "me", "imu", "mes", "mes_kiq", "vcn", "sos", "asd")}
        cards.append(c)
    rec["cards"] = cards
    return rec


def cpu_baseline(c, budget_s=20.0):
    """Time the literal torch-CPU restatement of the reference graph (oracle; checker only) on the host."""
    import numpy as np
    import torch
    torch.set_num_threads(effective_cpus())
    from oracle import dccn_oracle as O
    from oracle.torch_ref import LiteralRx
    S, kin = 7, c["nfft"] + c["cp"]
    cfg = O.RxConfig(S=S, kin=kin, F=c["F"], D=c["D"], nbits=c["nbits"])
    frames = c["frames"]
    rng = np.random.RandomState(0)
    x = rng.randn(frames, S, kin, 2).astype(np.float32)
    bits = rng.randint(0, 2, (frames, c["D"], c["nbits"]))
    out = {}
    for form, literal in (("conv3d", True), ("gemm", False)):
        model = LiteralRx(O.init_params(cfg, 1), cfg, dtype=torch.float32, literal_conv=literal)
        model.train_step(x, bits)                      # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            model.train_step(x, bits)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s / 2 or n >= 50:
                break
        out[form] = dict(steps=n, seconds=el, sym_per_s=n * frames * S / el)
    return dict(value=out["conv3d"]["sym_per_s"], unit="OFDM symbols/s", cores=torch.get_num_threads(),
                kind="port",
                sample="%d training steps of the same %d-frame batch, torch-CPU fp32 restatement of the TF1 graph "
                       "(zero-padded conv3d form, as TensorFlow evaluates it); TF1 itself is not installable"
                       % (out["conv3d"]["steps"], frames),
                gemm_form_value=out["gemm"]["sym_per_s"],
                gemm_form_sample="%d steps, same graph with the C-Conv as a centre-tap GEMM" % out["gemm"]["steps"])


def prewarm(step, dev, seconds=0.5):
    """Untimed continuous load before a timed region.  After >= 20 ms without work the GPU restarts ~10 % below the clock it
    sustains under continuous load and needs tens of milliseconds of work to get back (tools/ramp.py, profiles/r04_gap.md:
    2140 vs 2380 MHz measured inside the kernels); a 20-step region (1.6 ms) issued right after an idle gap -- engine
    construction, a first-use code-object load -- measures that restart clock, not the step."""
    import torch
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < seconds:
        for _ in range(50):
            step()
        torch.cuda.synchronize(dev)


def measure_config(name, dev, steps=20, warmup=5, graph=False, pipeline=True, want_prob=False):
    """ms per training step of another BASELINE configuration (same engine and launch mode, HIP-event timing)."""
    import torch
    from dl_ofdm_amd.engine import HipTimer, RxDims, RxEngine
    c = CONFIGS[name]
    dims = RxDims(S=7, kin=c["nfft"] + c["cp"], F=c["F"], D=c["D"], nbits=c["nbits"])
    eng = RxEngine(dims, c["frames"], device=dev, train=True, seed=1, want_prob=want_prob, want_tx_power=True, want_z=False,
                   want_dfft=False, want_grads=False)
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    eng.x.copy_(torch.randn(eng.x.shape, generator=g, device=dev))
    eng.bits.copy_(torch.randint(0, 2, eng.bits.shape, generator=g, device=dev, dtype=torch.int32))
    step = (lambda: eng.train_step_pipelined(graph=graph)) if pipeline else (lambda: eng.train_step(graph=graph))
    prewarm(step, dev, 0.3)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    t = HipTimer()
    regs = []
    for _ in range(5):                              # five back-to-back regions, the median one is reported
        t.start(eng._stream())
        for _ in range(steps):
            step()
        t.stop(eng._stream())
        regs.append(t.elapsed_ms() / steps)
    ms = sorted(regs)[len(regs) // 2]
    fl = step_flops(c)
    m = eng.metrics()
    eng.close_graph()
    return {"workload": c["workload"], "steps": steps, "ms_per_step": ms, "symbols_per_s": c["frames"] * 7 / (ms * 1e-3),
            "algorithmic_gflop": fl / 1e9, "achieved_tflops": fl / (ms * 1e-3) / 1e12,
            "mfma_frac": fl / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, "bits_counted": int(m["count"]),
            "regions_ms": [round(r, 5) for r in regs]}


SWEEP = dict(nbits=4, channel="EVA", frames=20000, snr_lo=-10, snr_hi=29,
             workload="16-QAM, Rayleigh EVA, N=64/CP=16, SNR sweep -10:29 dB (40 points x 20 000 frames, "
                      "dev/py/ofdmreceiver_np.py:59-91), frames generated on the device, evaluation only")


def measure_sweep(dev, rank, world, reps=2, frames=None):
    """BASELINE.json configs[2]: the SNR sweep of a 16-QAM receiver on EVA, sharded over the ranks exactly like
    dl_ofdm_amd.sweep shards every sweep of the harness (point i -> rank i % world; a point's 20 000 frames are one batch:
    R0's batch statistics couple them) and reduced by ONE all-reduce of the [points, 6] table.  Per point: five generator
    launches (bits -> grid -> IFFT+CP -> EVA taps -> FIR -> AWGN) + the evaluation step + one table row update, all
    stream-ordered, no host round trip.  Strong scaling: the 40 points are fixed, time = slowest rank."""
    import torch
    import torch.distributed as dist
    from dl_ofdm_amd import _lib, ofdm, receiver as R, sweep
    from dl_ofdm_amd.datagen import DeviceDataGen
    from dl_ofdm_amd.engine import RxEngine
    lib = _lib.load()
    frames = int(frames or SWEEP["frames"])
    F = R.Flags(nbits=SWEEP["nbits"], nfilter=64, channel=SWEEP["channel"], device_data=True, seed=1, test_frames=frames)
    o = ofdm.ofdm_tx(F)
    eng = RxEngine(R.rx_dims(F, o), frames, device=dev, train=False, seed=1, want_prob=False, want_tx_power=True, want_z=True)
    gen = DeviceDataGen(F, o, device=dev, seed=1)
    gen.want_noise_power = False
    pts = sweep.make_points([F.nbits], [F.channel], range(SWEEP["snr_lo"], SWEEP["snr_hi"] + 1), base_seed=1)
    table = torch.zeros(len(pts), 6, dtype=torch.float64, device=dev)

    def evaluate_into(p, row):
        gen.seed, gen.offset = p.seed, 0
        gen.make_batch(frames, p.snr_db, out_x=eng.x, out_bits=eng.bits)
        eng.eval_step()
        _lib.check(lib.dccn_metrics_table_add(eng.metrics_buf.data_ptr(), row.data_ptr(), eng._stream()), "table_add")

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    best = None
    for rep in range(reps + 1):               # rep 0 = warm-up (code objects, clocks, RCCL channels)
        barrier()
        t0 = time.perf_counter()
        sweep.run_sweep_device(pts, evaluate_into, rank, world, device=dev, table=table)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        tmax = torch.tensor([el], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
        if rep > 0 and (best is None or el < best):
            best = el
    t = table.cpu().numpy()
    ber, _ = sweep.ber_loss(t)
    nsym = len(pts) * frames * F.nsymbol
    return {"workload": SWEEP["workload"], "points": len(pts), "frames_per_point": frames, "n_gpus": world,
            "points_per_rank": [len(sweep.shard(pts, r, world)) for r in range(world)], "scaling": "strong",
            "seconds": best, "points_per_s": len(pts) / best, "symbols_per_s": nsym / best,
            "bits_counted": int(t[:, 5].sum()), "ber_first_last": [float(ber[0]), float(ber[-1])],
            "receiver": "seed-1 initialisation, NOT trained: this object times the sharded evaluation path; its BER (~0.5) says "
                        "nothing about the receiver (trained curves: profiles/*_config5/config5_ber.csv)",
            "collective": "one all-reduce of the [40, 6] float64 table" if world > 1 else "none (1 rank)"}


def measure_e2e(dev, c, steps=200, warmup=30, channel="EPA", snr_db=10.0):
    """The training loop as dl_ofdm_amd.receiver.train runs it with --device_data: every step draws its own batch on the
    GPU (label bits -> constellation grid -> IFFT + CP as one GEMM -> per-frame Rayleigh taps -> 'same' FIR -> power
    normalisation + AWGN at `snr_db`; BASELINE configs[1]: QPSK, Rayleigh EPA, SNR = 5*nbits dB, run_local_ofdm.py:66,69)
    straight into the engine's resident buffers, then runs the pipelined training step on it: batch i+1 is generated before
    step i is issued and normalised behind step i's optimizer launch.  Nothing crosses PCIe."""
    import torch
    from dl_ofdm_amd import ofdm, receiver as R
    from dl_ofdm_amd.datagen import DeviceDataGen
    from dl_ofdm_amd.engine import RxEngine
    F = R.Flags(nbits=c["nbits"], nfilter=c["F"], channel=channel, SNR=snr_db, seed=1)
    o = ofdm.ofdm_tx(F)
    frames = c["frames"]
    eng = RxEngine(R.rx_dims(F, o), frames, device=dev, train=True, seed=1, want_prob=False, want_z=False, want_dfft=False)
    gen = DeviceDataGen(F, o, device=dev, seed=1)
    gen.want_noise_power = False

    from dl_ofdm_amd.datagen import FusedStaticGen, SideStreamFeeder
    fused = FusedStaticGen.supported(gen, eng)
    count = [0]
    if fused:
        # round 5: ONE C call per batch -- the fused generator launch of the next batch + the four launches of the step, whose
        # pipelined normalisation reads (y, noise, power partials) as its virtual input (include/dccn.h dccn_gen_static)
        fg = FusedStaticGen(gen, frames, snr_db)

        def run(n, first):
            for _ in range(n):
                eng.train_step_generated(fg, slot=count[0] & 1)
                count[0] += 1
        plan = ("5 launches, one C call: fused generator of the next batch (bits -> grid -> IFFT+CP -> taps -> FIR -> y, scaled "
                "noise, power partials) | C-Conv fwd | dense fwd + tail | fused backward | optimizer + R0 of the next batch "
                "reading x = y / sqrt(mean |y|^2) + noise as its virtual input")
    else:
        feed = SideStreamFeeder(eng, lambda slot: gen.make_batch(frames, snr_db, out_x=eng.x, out_bits=eng.label_slot(slot)))

        def run(n, first):
            # the generator on its own stream: batch i+1 is produced while the forward and backward launches of step i run
            if first:
                feed.first(0)
            for i in range(n):
                feed.next((i + 1) & 1)
                eng.train_step_pipelined(slot=i & 1, x_ready=feed.ready)
                feed.step_issued()
        plan = ("4 generator (grid, IFFT+CP GEMM, FIR drawing its own taps, AWGN) on a side stream, overlapping the first three of "
                "the 4 training-step launches (the optimizer launch waits for them)")
    run(warmup, True)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.3:        # continuous load first (see prewarm())
        run(50, False)
        torch.cuda.synchronize(dev)
    regs, issue = [], []
    for _ in range(3):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run(steps, False)
        issue.append((time.perf_counter() - t0) / steps)       # host cost of issuing a batch (no sync yet)
        torch.cuda.synchronize(dev)
        regs.append((time.perf_counter() - t0) / steps)
    dt = sorted(regs)[1]
    m = eng.metrics()
    eng.drop_prefetch()
    return {"workload": "%s + device-side generator: Rayleigh %s at %.0f dB, a fresh %d-frame batch per step" %
                        (c["workload"], channel, snr_db, frames),
            "steps": steps, "ms_per_step": dt * 1e3, "symbols_per_s": frames * 7 / dt,
            "regions_ms": [round(r * 1e3, 5) for r in regs], "host_issue_ms_per_step": round(sorted(issue)[1] * 1e3, 5),
            "launches_per_step": plan,
            "ce_mean_last": m["ce_mean"], "ber_last": m["berlin"]}


def traffic_for(path: str, config: str, op: str, build_id: str):
    """(HBM bytes per launch of `op` from the PMC passes kept in `path`, stale?, the file's build id): the counters are
    collected by separate rocprofv3 --pmc runs (tools/gpu_run.sh pmc -> tools/pmc_summary.py) and stamped with the id of the
    library they were collected on; they are reported only for THAT library -- for any other build the line says null + stale"""
    try:
        rec = json.load(open(path))
    except Exception:
        return None, False, None
    file_id = rec.get("build_id")
    if file_id != build_id:
        return None, True, file_id
    return rec.get(config, {}).get(op), False, file_id


def launch_ranks(n: int) -> int:
    """Re-run this command line as `n` ranks under torch.distributed.run (rendezvous on 127.0.0.1, a free port); the
    children see WORLD_SIZE and take the normal path.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class Watchdog:
    """`with Watchdog(seconds, what):` -- if the block has not finished in time, print what hung and end the process
    (rc 3): a collective that never completes cannot be interrupted from Python any other way."""

    def __init__(self, seconds, what):
        self.seconds, self.what, self.timer = seconds, what, None

    def _fire(self):
        sys.stderr.write("bench.py: %s did not complete within %.0f s -- giving up\n" % (self.what, self.seconds))
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        import threading
        self.timer = threading.Timer(self.seconds, self._fire)
        self.timer.daemon = True
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--graph", action="store_true",
                    help="hipGraph replay of the captured step instead of stream launches (measured slower on ROCm 7.2: "
                         "every graph kernel node pays ~0.9 us more than a same-stream launch, DESIGN.md section 5)")
    ap.add_argument("--no-graph", action="store_true", help="(default since round 2) eager launch sequence")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="every step normalises its own batch first (6 launches) instead of normalising the next batch "
                         "behind its Adam update (5 launches; same results, DESIGN.md section 3.5)")
    ap.add_argument("--fork", action="store_true", help="two-stream graph (dense dW on a forked stream) instead of the grouped dX+dW launch")
    ap.add_argument("--regions", type=int, default=7,
                    help="how many times the timed K-step region is run back to back (each bracketed by barrier + synchronize); "
                         "`value` is the MEDIAN region, all of them are listed in step.regions_ms")
    ap.add_argument("--store-prob", action="store_true",
                    help="also write `output:0` (the per-bit probabilities) to HBM in every training step; the reference's step does not fetch it")
    ap.add_argument("--no-boundaries", action="store_true", help="skip the in-situ step timeline (step.boundaries)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-times", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short C3 / C4 measurements of the `configs` object")
    ap.add_argument("--no-e2e", action="store_true", help="skip the `e2e` object (training loop with the device-side generator)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the `sweep` object (configs[2]: 40-point SNR sweep sharded over the ranks)")
    ap.add_argument("--sweep-frames", type=int, default=0, help="frames per sweep point (default 20 000, the reference's)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` (no launcher): start the N ranks ourselves, one process per GPU, exactly as the
        # driver's torch.distributed.run command line would (dev/py/locals.py:28-38: one process per job)
        raise SystemExit(launch_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    from dl_ofdm_amd import _lib
    from dl_ofdm_amd.engine import HipTimer, RxDims, RxEngine, time_ops

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # DCCN_BENCH_BACKEND=gloo lets the N>1 control flow be exercised on a box with fewer GPUs than ranks
    # (ranks then share devices); the real runs use nccl (= RCCL over xGMI), one rank per GPU
    backend = os.environ.get("DCCN_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local >= ndev:
        raise SystemExit("LOCAL_RANK %d but only %d GPUs visible" % (local, ndev))
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a rendezvous or first collective that never completes (a rank that died, a wedged xGMI link) ends the job with a
        # message and rc 3 instead of hanging until the caller's limit
        with Watchdog(float(os.environ.get("DCCN_BENCH_INIT_TIMEOUT", "300")),
                      "rank %d: init_process_group(%s) / first barrier" % (rank, backend)):
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
            dist.barrier()

    c = CONFIGS[args.config]
    S, kin = 7, c["nfft"] + c["cp"]
    dims = RxDims(S=S, kin=kin, F=c["F"], D=c["D"], nbits=c["nbits"])
    # The step is the reference's `session.run([train_op, power_tx, ce_mean, berlin], ...)` (dev/py/ofdmreceiver_np.py:234): it
    # fetches the update, the TX power, the loss and the BER -- not `output:0`.  The probabilities are therefore formed and
    # consumed in registers (loss, decisions, backward) and not written to HBM, as the harness's own loop runs it
    # (dl_ofdm_amd/receiver.py); `--store-prob` also materialises them (6 MB per C2 step, round 1-5 lines did).  z / dfft are
    # consumed inside the launches that produce them (fused dense + tail forward, fused backward) whenever the plan allows.
    eng = RxEngine(dims, c["frames"], device=dev, train=True, seed=1 + rank, want_prob=args.store_prob, want_tx_power=True,
                   want_z=False, want_dfft=False, want_grads=False)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    eng.x.copy_(torch.randn(eng.x.shape, generator=g, device=dev))
    eng.bits.copy_(torch.randint(0, 2, eng.bits.shape, generator=g, device=dev, dtype=torch.int32))
    use_graph, fork = (args.graph or args.fork) and not args.no_graph, args.fork
    pipeline = not args.no_pipeline and not fork

    def step():
        if pipeline:
            eng.train_step_pipelined(graph=use_graph)
        else:
            eng.train_step(graph=use_graph, fork=fork)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    table = torch.zeros(6, dtype=torch.float64, device=dev)      # [c00,c01,c10,c11,ce_sum,count]
    lib = _lib.load()

    def reduce_table():
        """final BER/loss reduction over xGMI (the only collective of the path): the last step's metrics record goes
        into the float64 table row by one stream-ordered launch (no host round trip), then ONE all-reduce"""
        _lib.check(lib.dccn_metrics_table_set(eng.metrics_buf.data_ptr(), table.data_ptr(), eng._stream()), "table_set")
        if world > 1:
            dist.all_reduce(table)

    step()
    reduce_table()          # first use loads torch's element-wise code objects / RCCL channels (tens of ms of GPU idleness, once)
    if world > 1:
        dist.barrier()
    pci = hip_device_pci(dev)
    # (read BEFORE the pre-warm: the sysfs DPM files are answered by the SMU and take tens of milliseconds for a node's
    # cards -- read between the pre-warm and the timed regions, as round 4 did, they left the GPU idle long enough to restart
    # it at its post-idle clock, and with the driver's K = 20 the seven 1.5-ms regions then measured the recovery:
    # 0.0767, 0.0764, 0.0763, 0.0759, 0.0754, 0.0759, 0.0753 ms in gpurun_out/r05c)
    clocks_before = gpu_clock_state(pci) if rank == 0 else None
    timer = HipTimer()
    regions, ev_regions, enq_regions = [], [], []
    # untimed pre-warm AFTER everything that leaves the GPU idle: continuous load until the clocks are where they stay
    prewarm(step, dev, 0.5)
    for _ in range(args.warmup):
        step()
    # The timed region = EXACTLY K steps between barrier + synchronize pairs, MAX over ranks.  It is run `--regions` times
    # back to back and `value` is the median region: with the driver's K = 20 one region lasts 1.6 ms, and a single sample
    # of that length cannot tell a slow box from a slow moment (0.101 vs 0.079 ms on another box of the pool).
    for _ in range(max(1, args.regions)):
        barrier()
        timer.start(eng._stream())
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        enq_regions.append(time.perf_counter() - t0)            # host cost of issuing the K steps (no sync yet)
        reduce_table()
        timer.stop(eng._stream())
        # a rank's time = common start (barrier above) -> its own K steps and the table reduction have completed; the job's
        # time = MAX over ranks.  (The closing barrier is taken after the clock is read: with the driver's K = 20 a step
        # sequence lasts 1.5 ms, and a second RCCL barrier inside it would be timed instead of the path.)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        barrier()
        ev_regions.append(timer.elapsed_ms())
        tmax = torch.tensor([el], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        regions.append(float(tmax.item()))
    clocks_after = gpu_clock_state(pci) if rank == 0 else None
    order = sorted(range(len(regions)), key=lambda i: regions[i])
    mid = order[len(order) // 2]                                  # the median region (odd count: an actual sample)
    elapsed, ev_ms, t_enqueue = regions[mid], ev_regions[mid], enq_regions[mid]

    sym_per_step = c["frames"] * S
    value = world * args.steps * sym_per_step / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    result = {
        "metric": "OFDM symbols/sec fwd+bwd, DCCN QPSK N=64" if args.config == "c2" else "OFDM symbols/sec fwd+bwd, DCCN",
        "value": value, "unit": "OFDM symbols/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": c["workload"], "frames_per_step": c["frames"], "symbols_per_step": sym_per_step,
                   "nfft": c["nfft"], "cp": c["cp"], "nfilter": c["F"], "nbits": c["nbits"],
                   "launch": (("hipGraph replay" + (" (forked dW branch)" if fork else " (grouped dense dX+dW launch)")) if use_graph
                              else "stream launches") + (", software-pipelined across steps (" + pipeline_plan(eng) + ")"
                                                          if pipeline else ", 6 launches per step"),
                   "parallelism": "independent batch per GPU, final all-reduce of the BER/loss table"},
    }
    # what the N > 1 line rests on: backend, library version, the one collective of the timed region
    try:
        rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
    except Exception:
        rccl = None
    result["distributed"] = {
        "world": world, "backend": backend if world > 1 else "none (1 rank)", "rccl_version": rccl,
        "devices_visible": ndev, "rank_device": "cuda:%d" % local,
        "collective": ("one all-reduce of the [6] float64 BER/loss table per timed region (+ one MAX all-reduce of the clock, "
                       "outside it)") if world > 1 else "none (1 rank)",
        "units_per_rank": {"steps": args.steps, "frames_per_step": c["frames"]},
        "points_per_rank": None}
    if rank == 0:
        fl = step_flops(c)
        result["step"] = {"algorithmic_gflop": fl / 1e9, "achieved_tflops": fl / (ms_per_step * 1e-3) / 1e12,
                          "mfma_frac": fl / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                          "hip_event_ms_per_step": ev_ms / args.steps,
                          "host_enqueue_ms_per_step": t_enqueue / args.steps * 1e3,
                          "regions_ms": [round(r / args.steps * 1e3, 5) for r in regions],
                          "regions_hip_event_ms": [round(r / args.steps, 5) for r in ev_regions],
                          "ber_table": [float(v) for v in table.cpu()]}
        if not args.no_boundaries and not use_graph:
            # in-situ timeline of the same step (dl_ofdm_amd/steptrace.py): per launch its duration inside the step, the idle
            # gap in front of it and the shader clock its workgroups saw -- the decomposition of ms_per_step
            from dl_ofdm_amd.steptrace import trace_steps
            t1 = HipTimer()
            bd = trace_steps(step, dev, ring=16, bursts=14)
            prewarm(step, dev, 0.3)
            t1.start(eng._stream())
            for _ in range(200):
                step()
            t1.stop(eng._stream())
            bd["untraced_ms_per_step_200"] = round(t1.elapsed_ms() / 200, 5)
            bd["note"] = ("stamps cost ~1.5 us per launch (s_memrealtime / s_memtime round trips at block entry and exit): "
                          "period_us is the TRACED step, untraced_ms_per_step_200 the same loop with stamps off")
            result["step"]["boundaries"] = bd
            w = [(r["us"], r["sclk_mhz"]) for r in bd["launches"] if r.get("sclk_mhz")]
            if w:
                # the clock the CUs really ran at (duration-weighted over the launches): the 157.3 TFLOP/s peak is quoted at
                # 2400 MHz, a chip that sustains less under this load has proportionally less to give
                sclk = sum(u * c for u, c in w) / sum(u for u, _ in w)
                result["step"]["sclk_mhz_in_step"] = round(sclk, 0)
                result["step"]["mfma_frac_at_measured_sclk"] = result["step"]["mfma_frac"] * 2400.0 / sclk
        if not args.no_kernel_times:
            kt = time_ops(eng, iters=200, warmup=20)
            result["kernels"] = {k: {"us": round(v["ms"] * 1e3, 3), "tflops": round(v["tflops"], 2), "kernel": v["kernel"]}
                                 for k, v in kt.items()}
            dfw = "dense_tail_fwd_bwd" if "dense_tail_fwd_bwd" in kt else "dense_fwd"
            if fork:
                in_step = ("cconv_fwd", dfw, "dense_bwd_x", "dense_bwd_w", "cconv_bwd_w")
            elif "rx_backward" in kt:          # the backward half of the step is one launch (rx_bwd.h)
                in_step = ("cconv_fwd", dfw, "rx_backward")
            else:
                in_step = ("cconv_fwd", dfw, "dense_bwd_slabs", "cconv_bwd_w")
            gemm = {k: v for k, v in kt.items() if k in in_step}
            dom = max(gemm, key=lambda k: gemm[k]["ms"])
            traffic, stale, build = traffic_for(os.environ.get("DCCN_PMC_TRAFFIC", os.path.join(ROOT, "profiles", "pmc_traffic.json")),
                                                args.config, dom, lib.dccn_build_id().decode())
            result["roofline"] = {"bound": "mfma", "kernel": kt[dom]["kernel"], "op": dom,
                                  "achieved": kt[dom]["tflops"], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                  "frac": kt[dom]["tflops"] / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                                  "traffic_stale": stale, "traffic_build_id": build, "build_id": lib.dccn_build_id().decode(),
                                  "avg_launch_us": kt[dom]["ms"] * 1e3, "flops_per_launch": kt[dom]["flops"]}
        if world == 1 and args.config == "c2" and not args.no_other_configs:
            # the other BASELINE.json training configurations, measured in the same run (short: <= 20 steps each, hipGraph
            # replay, HIP events on the launch stream): C3 = config[2] shape (16-QAM), C4 = config[3] (N=1024, MFMA-bound)
            eng = None
            torch.cuda.empty_cache()
            result["configs"] = {k: measure_config(k, dev, steps=20, warmup=5, graph=use_graph, pipeline=pipeline, want_prob=args.store_prob) for k in ("c3", "c4")}
    if rank == 0 and world == 1 and args.config == "c2" and not args.no_e2e:
        eng = None
        torch.cuda.empty_cache()
        result["e2e"] = measure_e2e(dev, c)
    sweep_res = None
    if args.config == "c2" and not args.no_sweep:
        # configs[2] on every rank count the driver launches: this is the curve north_star calls "1/2/4/8-GPU SNR-sweep
        # scaling" (the training value above is weak scaling of replicas by construction)
        eng = None
        torch.cuda.empty_cache()
        sweep_res = measure_sweep(dev, rank, world, frames=args.sweep_frames or None)
    if rank == 0:
        if sweep_res is not None:
            result["sweep"] = sweep_res
            result["distributed"]["points_per_rank"] = sweep_res.get("points_per_rank")
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(c)
        cu, wf, hbm, arch = _lib.device_info()
        result["device"] = {"arch": arch, "cus": cu, "clocks_before_timed_regions": clocks_before,
                            "clocks_after_timed_regions": clocks_after, "box": box_fingerprint(pci),
                            # the box's HBM speed as a plain device-to-device copy sees it (1 GiB, read + write): one box of the
                            # round's pool ran the same build 11 % slower per step with its HBM-bound launches 50 % longer
                            "copy_1GiB_TBps": copy_bandwidth_tbs(dev)}
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
