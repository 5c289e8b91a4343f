"""BASELINE.json config[4]: all modulations x fading profiles (EPA, EVA, ETU) x SNRs, the DCCN receiver (basic receiver
trained on AWGN + equaliser trained on mixRayleigh -- the reference driver's recipe, dev/py/run_local_ofdm.py:61-118)
next to the classical LMMSE / LS receivers of :mod:`dl_ofdm_amd.benchmark` (dev/m/OFDM_Benchmark_dev.m).

The (modulation, channel, SNR) points are independent units (own data, own batch statistics, own confusion matrix): they
are dealt round-robin to the ranks of a ``torch.distributed`` job, one process per GPU, and meet in ONE all-reduce of
the ``[points, 6]`` table (sweep.py).  Every rank trains the (small) models itself -- seconds each, bitwise
reproducible from the seeds -- so no parameter broadcast is needed.

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


def init_distributed(backend: Optional[str] = None):
    """(rank, world, local device index); joins the torch.distributed job described by the environment, if any.
    backend: ``nccl`` (= RCCL over xGMI, one rank per GPU; default) or ``gloo`` (control-flow tests with ranks sharing a
    device); also taken from ``DCCN_DIST_BACKEND``."""
    import torch
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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


def train_models(out_dir: str, nbits_list: Sequence[int], frames: int, eq_epochs: int, rx_epoch_scale: float = 1.0,
                 rank: int = 0, device="cuda", verbose: bool = False):
    """nbits -> (equaliser flags, EqualizerTrainer with the best checkpoint loaded)."""
    from . import receiver as R, receiver_mp as H
    t0, trainers = time.time(), {}
    for nbits in nbits_list:
        save = os.path.join(out_dir, "ckpt_r%d/" % rank)
        rf = R.Flags(nbits=nbits, nfilter=64, channel="AWGN", SNR=5.0 * nbits,
                     max_epoch_num=max(1, int(1200 * nbits * rx_epoch_scale)), early_stop=200, token="C5_%dmod" % nbits,
                     save_dir=save, device_data=True, seed=nbits)
        res = R.train(rf, device=device, verbose=False, run_test=False)
        hf = H.Flags(nbits=nbits, nfilter=64, channel="mixRayleigh", max_epoch_num=eq_epochs, early_stop=200,
                     token=rf.token, save_dir=save, device_data=True, seed=10 + nbits, test_frames=frames)
        out = H.train(hf, device=device, verbose=False, run_test=False, rx_params=res["params"])
        H.load_checkpoint(out["best_path"], out["trainer"], with_optimizer=False)
        trainers[nbits] = (hf, out["trainer"])
        if verbose:
            print("nbits %d: receiver %d epochs, equaliser %d epochs, %.0f s"
                  % (nbits, len(res["history"]), len(out["history"]), time.time() - t0))
    return trainers


def sweep_dccn(trainers: Dict, nbits_list: Sequence[int], channels: Sequence[str], snrs: Sequence[float], frames: int,
               rank: int = 0, world: int = 1, group=None, base_seed: int = 77):
    """The sharded DCCN sweep: returns (points, float64 table [points, 6]) on every rank."""
    import torch
    from . import ofdm, sweep
    from .datagen import DeviceDataGen
    pts = sweep.make_points(list(nbits_list), list(channels), list(snrs), base_seed=base_seed)
    gens = {}

    def evaluate(p):
        hf, tr = trainers[p.nbits]
        key = (p.nbits, p.channel)
        if key not in gens:
            fl = copy.deepcopy(hf)
            fl.channel = p.channel
            gens[key] = DeviceDataGen(fl, ofdm.ofdm_tx(fl), device=tr.device, seed=p.seed)
        g, pl = gens[key], tr.resident(frames)
        g.seed, g.offset = p.seed, 0
        g.make_batch(frames, p.snr_db, out_x=pl.x, out_bits=pl.bits)
        pl.run(False)
        m = tr._metrics(pl.metrics_buf, pl.tx_power)
        c = m["conf"]
        return [c[0][0], c[0][1], c[1][0], c[1][1], m["ce_sum"], m["count"]]

    dev = next(iter(trainers.values()))[1].device
    table = sweep.run_sweep(pts, evaluate, rank, world, device=torch.device(dev), group=group)
    return pts, table


def run(out_dir: str, frames: int = 20000, eq_epochs: int = 600, classical_frames: int = 1500, rx_epoch_scale: float = 1.0,
        nbits_list: Sequence[int] = (1, 2, 3, 4), channels: Sequence[str] = CHANNELS, snrs: Sequence[int] = SNRS,
        classical_every: int = 3, rank: int = 0, world: int = 1, device="cuda", verbose: bool = True):
    """Train, sweep, and (rank 0) write ``<out_dir>/config5_ber.csv``; returns (points, BER per point)."""
    from . import benchmark, receiver as R, sweep
    os.makedirs(out_dir, exist_ok=True)
    t0 = time.time()
    trainers = train_models(out_dir, nbits_list, frames, eq_epochs, rx_epoch_scale, rank, device, verbose and rank == 0)
    pts, table = sweep_dccn(trainers, nbits_list, channels, snrs, frames, rank, world)
    ber, _ = sweep.ber_loss(table)
    if rank == 0:
        if verbose:
            print("DCCN sweep done: %d points, %.0f s" % (len(pts), time.time() - t0))
        csnr = list(snrs)[::classical_every]
        classical = {}
        for nbits in nbits_list:
            for ch in channels:
                fl = R.Flags(nbits=nbits, channel=ch)
                for m in ("LMMSE", "LS-Spline", "Perfect"):
                    classical[(nbits, ch, m)] = benchmark.ber_curve(fl, m, csnr, n_frames=classical_frames, seed=5)
        with open(os.path.join(out_dir, "config5_ber.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["modulation", "channel", "SNR", "DCCN+Equalizer", "LMMSE", "LS-Spline", "Perfect"])
            for p in pts:
                row = [benchmark.MOD_NAMES[p.nbits - 1], p.channel, int(p.snr_db), "%.6g" % ber[p.index]]
                if int(p.snr_db) in csnr:
                    j = csnr.index(int(p.snr_db))
                    row += ["%.6g" % classical[(p.nbits, p.channel, m)][j] for m in ("LMMSE", "LS-Spline", "Perfect")]
                else:
                    row += ["", "", ""]
                w.writerow(row)
        if verbose:
            print("wrote %s, total %.0f s" % (os.path.join(out_dir, "config5_ber.csv"), time.time() - t0))
    return pts, ber
