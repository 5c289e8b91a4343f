"""BASELINE.json config[4]: all modulations x fading profiles (EPA, EVA, ETU) x SNRs, the DCCN receiver (basic receiver
trained on AWGN + equaliser trained on mixRayleigh -- the reference driver's recipe, dev/py/run_local_ofdm.py:61-118)
next to the classical LMMSE / LS receivers of :mod:`dl_ofdm_amd.benchmark` (dev/m/OFDM_Benchmark_dev.m).

The (modulation, channel, SNR) points are independent units (own data, own batch statistics, own confusion matrix): they
are dealt round-robin to the ranks of a ``torch.distributed`` job, one process per GPU, and meet in ONE all-reduce of
the ``[points, 6]`` table (sweep.py).  The expensive stages shard the same way: the four (receiver -> equaliser) training
chains are dealt to ranks and the trained arenas broadcast (9 MB per modulation), the classical receivers' (modulation,
channel, estimator, SNR) units are dealt round-robin and meet in one more all-reduce.  Nothing is replicated.

    python tools/config5_sweep.py --out profiles/r02_config5 [--frames 20000] [--eq_epochs 600]
"""
from __future__ import annotations

import copy
import csv
import os
import time
from typing import Dict, Optional, Sequence

import numpy as np

CHANNELS = ("EPA", "EVA", "ETU")
SNRS = tuple(range(-10, 30))


def visible_gpus_without_hip() -> int:
    """GPUs this process will see, counted WITHOUT initialising the HIP runtime (its queue limits are read from the
    environment once, at initialisation): the visibility variables if set, else the DRM render nodes."""
    import glob
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            return len([t for t in v.split(",") if t.strip() != ""])
    return len(glob.glob("/dev/dri/renderD*"))


def shared_gpu_env(local_world: int, n_gpus: int, env=None) -> Dict[str, str]:
    """Environment additions for ranks that SHARE a GPU (local_world > n_gpus): one hardware queue per process.
    Measured (tools/c5share.sh, four config-5 chains as four processes on one MI355X, equaliser seconds per chain):
    GPU_MAX_HW_QUEUES=1: 4.9-5.9 (serial: 3.9 each, one after the other); the runtime's default of 4 queues per process:
    4.8-19.4; 2: 24-27; 8: 30-33 -- beyond a handful of queues the scheduler time-slices them in millisecond quanta, and a
    chain of 5-us launches stalls for most of every quantum.  A value already set by the user is kept."""
    env = os.environ if env is None else env
    if n_gpus > 0 and local_world > n_gpus and "GPU_MAX_HW_QUEUES" not in env:
        return {"GPU_MAX_HW_QUEUES": "1"}
    return {}


def init_distributed(backend: Optional[str] = None):
    """(rank, world, local device index); joins the torch.distributed job described by the environment, if any.
    backend: ``nccl`` (= RCCL over xGMI, one rank per GPU; default) or ``gloo`` (ranks may share a device: rank r works on
    GPU ``LOCAL_RANK % n_gpus``); also taken from ``DCCN_DIST_BACKEND``.

    Ranks sharing a GPU are not only a test device: config 5's training chains are latency-bound sequences of 73-frame steps
    that keep a few percent of an MI355X busy each, and independent processes own independent hardware queues, so
    ``torchrun --nproc-per-node 4 ... tools/config5_sweep.py --backend gloo`` (or ``tools/config5_sweep.py --share 4``) on ONE
    GPU trains the four modulations next to each other -- with ONE hardware queue per process (shared_gpu_env; set here,
    before the first HIP call of the process).  Every model is still trained by exactly one rank with its own seeds: the
    result does not depend on the sharing."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    os.environ.update(shared_gpu_env(local_world, visible_gpus_without_hip()))
    import torch
    backend = backend or os.environ.get("DCCN_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > 1 and local >= ndev:
        raise SystemExit("LOCAL_RANK %d but only %d GPUs visible" % (local, ndev))
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend)
    return rank, world, local


def job_owners(nbits_list: Sequence[int], world: int) -> Dict[int, int]:
    """Training jobs = one (receiver -> equaliser) chain per modulation (the reference driver runs each as an OS process
    of its own, dev/py/run_local_ofdm.py:61-118; the equaliser needs ITS receiver, so a chain is the independent unit).
    Longest chain first (epochs scale with nbits), each to the least-loaded rank so far (LPT): 2 ranks get {4, 1} and
    {3, 2}, 4 or more ranks one chain each."""
    load = [0] * max(world, 1)
    owners = {}
    for b in sorted((int(b) for b in nbits_list), reverse=True):
        r = min(range(len(load)), key=lambda i: (load[i], i))
        owners[b] = r
        load[r] += b
    return owners


def _broadcast(t, src: int, group=None):
    """One tensor from its owner to every rank (RCCL on device tensors; gloo stages through the host)."""
    import torch
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def train_models(out_dir: str, nbits_list: Sequence[int], frames: int, eq_epochs: int, rx_epoch_scale: float = 1.0,
                 rank: int = 0, device="cuda", verbose: bool = False, world: int = 1, group=None,
                 timing: Optional[dict] = None, ckpt_dir: Optional[str] = None, chain_streams: Optional[int] = None,
                 chain_priority: bool = True, chain_group: Optional[int] = None, sweep_plan: Optional[dict] = None):
    """nbits -> (equaliser flags, EqualizerTrainer with the best checkpoint loaded), on every rank.
    eq_epochs <= 0: the reference driver's cap of 4000 * nbits epochs (run_local_ofdm.py:96; early stopping ends it sooner).

    chain_streams (default: one per chain this rank owns): the rank's chains train NEXT TO each other, each in a host thread
    of this process issuing to a HIP stream of its own -- the reference driver starts them as OS processes
    (run_local_ofdm.py:61-118, locals.py:28-38).  A chain is a latency-bound sequence of 73-frame steps that keeps a few
    percent of an MI355X busy; several hardware queues of ONE process interleave on the chip without the context switches
    that separate processes pay.  Every chain draws from generators of its own (host RandomState, device Philox streams),
    so the trained arenas are the bits a serial run (chain_streams=1) produces.

    world > 1: each chain is trained by its owner only (job_owners); the owner then broadcasts the two flat arenas
    (receiver 2.3 MB, equaliser 7 MB) and the other ranks build the same trainer around them.  Training is bitwise
    reproducible from the seeds, so the trainers equal those of a serial run whatever the rank count."""
    import torch
    from . import ofdm, receiver as R, receiver_mp as H
    from .engine import PARAM_NAMES, param_layout
    from .equalizer import EqualizerTrainer
    owners = job_owners(nbits_list, world)
    t0, trainers, flags = time.time(), {}, {}
    mine = []
    for nbits in sorted(owners, reverse=True):
        save = os.path.join(ckpt_dir or out_dir, "ckpt_r%d/" % rank)      # (~90 MB of best-model archives per rank)
        rf = R.Flags(nbits=nbits, nfilter=64, channel="AWGN", SNR=5.0 * nbits,
                     max_epoch_num=max(1, int(1200 * nbits * rx_epoch_scale)), early_stop=200, token="C5_%dmod" % nbits,
                     save_dir=save, device_data=True, seed=nbits)
        hf = H.Flags(nbits=nbits, nfilter=64, channel="mixRayleigh", max_epoch_num=eq_epochs if eq_epochs > 0 else 4000 * nbits, early_stop=200,
                     token=rf.token, save_dir=save, device_data=True, seed=10 + nbits, test_frames=frames)
        flags[nbits] = (rf, hf)
        if owners[nbits] == rank:
            mine.append(nbits)

    def chain(unit):
        """a unit = one chain, or several chains whose equaliser stages train as ONE chain group (equalizer_group.py: one launch
        sequence for all of them, each bit for bit its solo self); their receivers train one after the other first"""
        grouped = not isinstance(unit, int)
        unit = list(unit) if grouped else [unit]
        # the group unit holds the chains with slack (smallest epoch budgets): it starts once the single chains' receivers are
        # trained, so the chains everybody waits for at the end have the GPU (and the interpreter) nearly to themselves for that
        # stage -- a scheduling choice only, every chain computes what it computes
        if grouped and rx_gate is not None:
            rx_gate.wait()
        rx_res, t_rx = {}, {}
        for nbits in unit:
            t1 = time.time()
            try:
                rx_res[nbits] = R.train(flags[nbits][0], device=device, verbose=False, run_test=False)
            finally:
                if not grouped and rx_gate is not None:
                    with gate_lock:
                        gate_left[0] -= 1
                        if gate_left[0] <= 0:
                            rx_gate.set()
            t_rx[nbits] = time.time() - t1
        t2 = time.time()
        if len(unit) == 1:
            outs = [H.train(flags[unit[0]][1], device=device, verbose=False, run_test=False, rx_params=rx_res[unit[0]]["params"])]
        else:
            from .equalizer_group import train_group
            outs = train_group([flags[b][1] for b in unit], [rx_res[b]["params"] for b in unit], device=device)
        for nbits, out in zip(unit, outs):
            H.load_checkpoint(out["best_path"], out["trainer"], with_optimizer=False)
            trainers[nbits] = (flags[nbits][1], out["trainer"])
            if sweep_plan is not None and world == 1:
                # this chain's own sweep points right away, on this chain's stream: the other chains are still training, so
                # the sweep leaves the end of the job (the points carry their seeds: who evaluates them, and when, changes nothing)
                ev = point_evaluator({nbits: trainers[nbits]}, sweep_plan["frames"])
                for p_ in sweep_plan["points"]:
                    if p_.nbits == nbits:
                        sweep_plan["rows"][p_.index] = ev(p_)
            if timing is not None:
                timing["train_rx_%d" % nbits] = t_rx[nbits]
                timing["train_eq_%d" % nbits] = time.time() - t2
            if verbose:
                print("rank %d nbits %d: receiver %d epochs %.0f s, equaliser %d epochs %.0f s%s"
                      % (rank, nbits, len(rx_res[nbits]["history"]), t_rx[nbits], len(out["history"]), time.time() - t2,
                         " (group of %d)" % len(unit) if len(unit) > 1 else ""), flush=True)

    n_streams = len(mine) if chain_streams is None else max(1, min(int(chain_streams), len(mine)))
    # chain_group: this many chains -- the ones with the smallest epoch budgets, the end of `mine` -- share ONE stream as a chain
    # group; the others keep a stream each.  Default: two of four or more chains.
    n_group = (2 if len(mine) >= 4 and n_streams > 1 else 0) if chain_group is None else int(chain_group)
    n_group = n_group if 2 <= n_group <= len(mine) else 0
    rx_gate = None
    if n_group:
        mine = mine[:len(mine) - n_group] + [tuple(mine[len(mine) - n_group:])]
        n_streams = min(n_streams, len(mine))
        if n_streams == len(mine) and len(mine) > 1:          # (every unit has its own thread: the gate cannot deadlock)
            import threading
            rx_gate, gate_lock, gate_left = threading.Event(), threading.Lock(), [len(mine) - 1]
    if n_streams <= 1:
        for nbits in mine:
            chain(nbits)
    else:
        # longest chain first (`mine` is sorted that way); a worker thread owns one HIP stream and takes the next chain when
        # its own has finished.  Worker exceptions are re-raised here.
        import queue
        import threading
        todo, errors = queue.Queue(), []
        for nbits in mine:
            todo.put(nbits)

        # stream priority: the chains with the larger epoch budgets (4000 * nbits epochs, run_local_ofdm.py:96) are the ones the
        # others wait for at the end -- the upper half of `mine` issues to HIGH-priority streams (their launches are dispatched
        # first whenever two chains have work ready), so the critical path runs near its solo speed and the shorter chains
        # fill the gaps.  Priorities only order dispatch: results do not depend on them.
        high = set(u for u in mine[:(len(mine) + 1) // 2] if isinstance(u, int)) if chain_priority else set()

        def worker():
            try:
                while True:
                    try:
                        nbits = todo.get_nowait()
                    except queue.Empty:
                        break
                    st = torch.cuda.Stream(device=device, priority=-1 if nbits in high else 0)
                    with torch.cuda.stream(st):
                        chain(nbits)
                        st.synchronize()
            except BaseException as e:                  # noqa: BLE001 -- handed to the caller's thread
                errors.append(e)

        torch.cuda.synchronize(device)
        ths = [threading.Thread(target=worker, name="c5-chain-%d" % i) for i in range(n_streams)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize(device)
        if errors:
            raise errors[0]
    if timing is not None:
        timing["train_total"] = time.time() - t0
        timing["chain_streams"] = n_streams
        timing["chain_group"] = n_group
    if world > 1:
        t3 = time.time()
        for nbits in sorted(owners, reverse=True):
            rf, hf = flags[nbits]
            o = ofdm.ofdm_tx(hf)
            lay, total = param_layout(R.rx_dims(rf, o))
            if owners[nbits] == rank:
                tr = trainers[nbits][1]
                rx_arena, eq_arena = tr.rx_arena, tr.params
                n_eq = torch.tensor([eq_arena.numel()], dtype=torch.int64, device=rx_arena.device)
            else:
                rx_arena = torch.zeros(total, dtype=torch.float32, device=device)
                n_eq = torch.zeros(1, dtype=torch.int64, device=device)
            _broadcast(rx_arena, owners[nbits], group)
            _broadcast(n_eq, owners[nbits], group)
            if owners[nbits] != rank:
                rxp = {n: rx_arena[lay[n][0]:lay[n][0] + int(np.prod(lay[n][1]))].view(*lay[n][1]).cpu().numpy()
                       for n in PARAM_NAMES}
                tr = EqualizerTrainer(hf, o, rxp, device=device, seed=hf.seed)
                assert tr.params.numel() == int(n_eq.item())
                eq_arena = tr.params
                trainers[nbits] = (hf, tr)
            with torch.no_grad():
                _broadcast(eq_arena, owners[nbits], group)
        if timing is not None:
            timing["broadcast"] = time.time() - t3
    return trainers


def point_evaluator(trainers: Dict, frames: int):
    """evaluate(point) -> the six table values of one (modulation, channel, SNR) point on its own device-generated batch"""
    import copy as _copy
    from . import ofdm
    from .datagen import DeviceDataGen
    gens = {}

    def evaluate(p):
        hf, tr = trainers[p.nbits]
        key = (p.nbits, p.channel)
        if key not in gens:
            fl = _copy.deepcopy(hf)
            fl.channel = p.channel
            gens[key] = DeviceDataGen(fl, ofdm.ofdm_tx(fl), device=tr.device, seed=p.seed)
            gens[key].want_noise_power = False
        g, pl = gens[key], tr.resident(frames)
        g.seed, g.offset = p.seed, 0
        g.make_batch(frames, p.snr_db, out_x=pl.x, out_bits=pl.bits)
        pl.run(False)
        m = tr._metrics(pl.metrics_buf, pl.tx_power)
        c = m["conf"]
        return [c[0][0], c[0][1], c[1][0], c[1][1], m["ce_sum"], m["count"]]

    return evaluate


def sweep_dccn(trainers: Dict, nbits_list: Sequence[int], channels: Sequence[str], snrs: Sequence[float], frames: int,
               rank: int = 0, world: int = 1, group=None, base_seed: int = 77, done_rows: Optional[dict] = None):
    """The sharded DCCN sweep: returns (points, float64 table [points, 6]) on every rank.  ``done_rows`` (world 1): rows some
    points already have (point index -> six values: a chain's own points, evaluated by its thread as soon as it was trained)."""
    import torch
    from . import sweep
    pts = sweep.make_points(list(nbits_list), list(channels), list(snrs), base_seed=base_seed)
    ev = point_evaluator(trainers, frames)
    done_rows = done_rows if (done_rows and world == 1) else {}

    def evaluate(p):
        return done_rows[p.index] if p.index in done_rows else ev(p)

    dev = next(iter(trainers.values()))[1].device
    table = sweep.run_sweep(pts, evaluate, rank, world, device=torch.device(dev), group=group)
    return pts, table


CLASSICAL_METHODS = ("LMMSE", "LS-Spline", "Perfect")


def classical_workers(nbits_list: Sequence[int], world: int) -> list:
    """the ranks that compute the classical units: those WITHOUT a training chain when there are any (8 GPUs: ranks 4-7 --
    they do it while ranks 0-3 train, so the stage leaves the critical path), every rank otherwise"""
    owners = set(job_owners(nbits_list, world).values())
    free = [r for r in range(max(world, 1)) if r not in owners]
    return free if free else list(range(max(world, 1)))


def classical_curves(nbits_list: Sequence[int], channels: Sequence[str], csnr: Sequence[int], n_frames: int,
                     rank: int = 0, world: int = 1, group=None, seed: int = 5, methods: Sequence[str] = CLASSICAL_METHODS,
                     device=None, workers: Optional[Sequence[int]] = None, reduce: bool = True, table=None):
    """The LMMSE / LS baseline curves (dev/m/OFDM_Benchmark_dev.m:339-456 via dl_ofdm_amd.benchmark), one unit per
    (modulation, channel, estimator, SNR) -- each with its own seeded draws -- dealt round-robin to the ``workers`` (default:
    every rank); ONE all-reduce of the [units, 2] table {bit errors, bits}.  ``device``: run them on that GPU (benchmark_gpu:
    device-side generator + libdccn receivers) instead of the host NumPy restatement.  Returns {(nbits, channel, method):
    BER per csnr} -- or, with reduce=False, this rank's un-reduced table (pass it back as ``table`` with reduce=True later:
    the two halves of the stage may lie on either side of the training stage)."""
    import torch
    from . import benchmark, receiver as R, sweep
    units = [(b, ch, m, i) for b in nbits_list for ch in channels for m in methods for i in range(len(csnr))]
    workers = list(workers) if workers is not None else list(range(max(world, 1)))
    have = table is not None
    if not have:
        table = torch.zeros(len(units), 2, dtype=torch.float64)
    cache = {}
    for u, (b, ch, m, i) in enumerate(units):
        if have or rank not in workers or workers[u % len(workers)] != rank:
            continue
        key = (b, ch, m)
        if key not in cache:
            if device is not None:          # frames from the device generator, receivers as libdccn launches
                from .benchmark_gpu import CurvePointsGPU
                cache[key] = CurvePointsGPU(R.Flags(nbits=b, channel=ch, nfilter=64), m, n_frames=n_frames, seed=seed, device=device)
            else:
                cache[key] = benchmark.CurvePoints(R.Flags(nbits=b, channel=ch), m, n_frames=n_frames, seed=seed)
        e, n = cache[key].point(i, float(csnr[i]))
        table[u, 0], table[u, 1] = e, n
    if not reduce:
        return table
    if world > 1:
        import torch.distributed as dist
        if dist.get_backend(group) != "gloo":
            table = table.cuda()
        sweep.reduce_table(table, world, group)
        table = table.cpu()
    t = table.numpy()
    out = {}
    for u, (b, ch, m, i) in enumerate(units):
        out.setdefault((b, ch, m), np.zeros(len(csnr)))[i] = t[u, 0] / max(t[u, 1], 1.0)
    return out


def run(out_dir: str, frames: int = 20000, eq_epochs: int = 600, classical_frames: int = 1500, rx_epoch_scale: float = 1.0,
        nbits_list: Sequence[int] = (1, 2, 3, 4), channels: Sequence[str] = CHANNELS, snrs: Sequence[int] = SNRS,
        classical_every: int = 3, rank: int = 0, world: int = 1, device="cuda", verbose: bool = True,
        ckpt_dir: Optional[str] = None, chain_streams: Optional[int] = None, chain_group: Optional[int] = None):
    """Train (chains dealt to ranks), sweep (points dealt to ranks), classical curves (units dealt to ranks); rank 0
    writes ``<out_dir>/config5_ber.csv`` and ``config5_timing.json`` (wall time per stage and rank).  Returns
    (points, BER per point)."""
    import json
    from . import benchmark, sweep
    os.makedirs(out_dir, exist_ok=True)
    t0 = time.time()
    timing = {}
    # The classical units need no trained model: ranks that own no training chain (job_owners: at most one chain per
    # modulation, so ranks 4.. of an 8-GPU job) compute ALL of them first, while the owners train; the table meets in its
    # all-reduce after the sweep.  The units carry their own seeds, so who computes them does not change a digit.
    csnr = list(snrs)[::classical_every]
    workers = classical_workers(nbits_list, world)
    early = len(workers) < max(world, 1)
    ctab = None
    side = None
    if early:
        tc = time.time()
        ctab = classical_curves(nbits_list, channels, csnr, classical_frames, rank, world, device=device, workers=workers,
                                reduce=False)
        timing["classical_early"] = time.time() - tc
    elif world == 1 and device is not None and (chain_streams is None or chain_streams > 1):
        # one process: the classical units need no trained model either -- a host thread with a stream of its own computes them
        # while the chains train (they carry their own seeds: who computes them, and when, does not change a digit)
        import threading
        import torch
        box = {}

        def classical_side():
            tc = time.time()
            try:
                with torch.cuda.stream(torch.cuda.Stream(device=device)):
                    box["tab"] = classical_curves(nbits_list, channels, csnr, classical_frames, rank, world, device=device,
                                                  workers=workers, reduce=False)
                    torch.cuda.current_stream(device).synchronize()
            except BaseException as e:                  # noqa: BLE001 -- re-raised by the caller's thread
                box["err"] = e
            box["t"] = time.time() - tc

        side = threading.Thread(target=classical_side, name="c5-classical")
        side.start()
    splan = None
    if world == 1 and (chain_streams is None or chain_streams > 1):
        from . import sweep as _sw
        splan = dict(frames=frames, rows={}, points=_sw.make_points(list(nbits_list), list(channels), list(snrs), base_seed=77))
    trainers = train_models(out_dir, nbits_list, frames, eq_epochs, rx_epoch_scale, rank, device, verbose, world=world,
                            timing=timing, ckpt_dir=ckpt_dir, chain_streams=chain_streams, chain_group=chain_group, sweep_plan=splan)
    t1 = time.time()
    pts, table = sweep_dccn(trainers, nbits_list, channels, snrs, frames, rank, world, done_rows=splan["rows"] if splan else None)
    ber, _ = sweep.ber_loss(table)
    timing["sweep"] = time.time() - t1
    if verbose and rank == 0:
        print("DCCN sweep done: %d points, %.0f s since start" % (len(pts), time.time() - t0), flush=True)
    t2 = time.time()
    if side is not None:
        side.join()
        if "err" in box:
            raise box["err"]
        ctab = box["tab"]
        timing["classical_beside_training"] = box["t"]
    classical = classical_curves(nbits_list, channels, csnr, classical_frames, rank, world, device=device, workers=workers,
                                 table=ctab)
    timing["classical"] = time.time() - t2
    timing["total"] = time.time() - t0
    # seconds this rank spent neither training, sweeping nor on classical units: waiting for the longest chain (the
    # `broadcast` entry is that wait plus 9 MB per modulation over xGMI) -- the config's scaling cap, stated per rank
    # (chains training next to each other on streams of their own: their wall time is train_total, not the sum of the chains')
    busy = (timing["train_total"] if timing.get("chain_streams", 1) > 1 else
            sum(v for k, v in timing.items() if k.startswith(("train_rx_", "train_eq_")))) + timing["sweep"] + \
        timing["classical"] + timing.get("classical_early", 0.0)
    timing["idle"] = max(0.0, timing["total"] - busy)
    all_t = [timing]
    if world > 1:
        import torch.distributed as dist
        all_t = [None] * world
        dist.all_gather_object(all_t, timing)
    if rank == 0:
        with open(os.path.join(out_dir, "config5_ber.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["modulation", "channel", "SNR", "DCCN+Equalizer"] + list(CLASSICAL_METHODS))
            for p in pts:
                row = [benchmark.MOD_NAMES[p.nbits - 1], p.channel, int(p.snr_db), "%.6g" % ber[p.index]]
                if int(p.snr_db) in csnr:
                    j = csnr.index(int(p.snr_db))
                    row += ["%.6g" % classical[(p.nbits, p.channel, m)][j] for m in CLASSICAL_METHODS]
                else:
                    row += ["", "", ""]
                w.writerow(row)
        # what every rank spent where: nothing is replicated (training chains, sweep points and classical units are all
        # dealt to ranks); `broadcast` + the table reductions are the only joint steps
        with open(os.path.join(out_dir, "config5_timing.json"), "w") as f:
            import torch.distributed as dist
            json.dump({"world": world, "backend": dist.get_backend() if world > 1 else "none (1 rank)",
                       "hw_queues_per_process": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default"),
                       "classical_workers": workers, "frames": frames, "eq_epochs": eq_epochs, "rx_epoch_scale": rx_epoch_scale,
                       "classical_frames": classical_frames, "points": len(pts), "job_owners": job_owners(nbits_list, world),
                       "per_rank_seconds": all_t}, f, indent=1)
        if verbose:
            print("wrote %s, total %.0f s" % (os.path.join(out_dir, "config5_ber.csv"), time.time() - t0), flush=True)
    return pts, ber
