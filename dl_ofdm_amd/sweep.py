"""SNR x modulation x channel BER sweep, sharded over the GPUs of one node.

The reference evaluates sweep points one after the other in a single process
(dev/py/ofdmreceiver_np.py:59-91: 41 SNRs x 20 000 frames; dev/py/run_local_ofdm.py: one OS process
per model).  Each point is an independent unit -- its own data, its own batch statistics (R0 couples
only the frames of ONE point), its own 2x2 confusion matrix -- so the sweep shards with no data-path
collective: rank r takes points r, r+W, r+2W, ... and the only communication is ONE all-reduce
(RCCL over xGMI on GPUs, gloo on CPU) of the zero-initialised ``[points, 6]`` table
``{c00, c01, c10, c11, ce_sum, count}`` at the end.  BER = (c01+c10)/sum is computed after the
reduction in float64.
"""
from __future__ import annotations

import csv
from dataclasses import dataclass
from typing import Callable, Iterable, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

TABLE_COLS = ("c00", "c01", "c10", "c11", "ce_sum", "count")


@dataclass(frozen=True)
class SweepPoint:
    index: int
    nbits: int
    channel: str
    snr_db: float
    seed: int


def make_points(nbits_list: Sequence[int], channels: Sequence[str], snrs: Iterable[float], base_seed: int = 1) -> List[SweepPoint]:
    """Cartesian sweep in the reference's loop order (modulation, channel, SNR); seeds are fixed per point
    (the reference reseeds from the wall clock, ofdmreceiver_np.py:73 -- runs here are reproducible)."""
    pts = []
    for b in nbits_list:
        for ch in channels:
            for s in snrs:
                pts.append(SweepPoint(len(pts), int(b), str(ch), float(s), base_seed + 7919 * len(pts)))
    return pts


def shard(points: Sequence[SweepPoint], rank: int, world: int) -> List[SweepPoint]:
    """Round-robin ownership: balances the SNR-dependent cost and keeps every rank busy to the end."""
    return [p for p in points if p.index % world == rank]


def reduce_table(table: torch.Tensor, world: int = 1, group=None) -> torch.Tensor:
    """Sum the per-rank tables (rows a rank does not own are zero) over the ranks that SHARED this sweep.

    ``world`` is the caller's sharding degree, not the size of whatever process group happens to exist: a driver
    may hold a group open while every rank evaluates a different, unsharded configuration (world == 1) -- those
    tables must not be summed, and ranks with unequal numbers of sweeps would dead-lock in the collective."""
    if world > 1:
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("sweep sharded over %d ranks but no process group is initialised" % world)
        dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)
    return table


def run_sweep(points: Sequence[SweepPoint], evaluate: Callable[[SweepPoint], Sequence[float]],
              rank: int = 0, world: int = 1, device: Optional[torch.device] = None, group=None) -> np.ndarray:
    """Evaluate this rank's shard and reduce.  ``evaluate(point)`` returns the six TABLE_COLS values of
    one point (confusion counts, summed cross entropy, bit count).  Returns the full float64 table on
    every rank."""
    table = torch.zeros(len(points), len(TABLE_COLS), dtype=torch.float64, device=device or torch.device("cpu"))
    for p in shard(points, rank, world):
        row = evaluate(p)
        table[p.index] = torch.as_tensor(np.asarray(row, dtype=np.float64), device=table.device)
    reduce_table(table, world, group)
    return table.cpu().numpy()


def run_sweep_device(points: Sequence[SweepPoint], evaluate_into: Callable[[SweepPoint, torch.Tensor], None],
                     rank: int = 0, world: int = 1, device: Optional[torch.device] = None, group=None,
                     table: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The same sharding with the table kept on the device: ``evaluate_into(point, row)`` ENQUEUES the point's work on
    the current stream and accumulates its six values into ``row`` (a float64 view of the table, e.g. through
    dccn_metrics_table_add) -- no host round trip per point, so a rank's whole shard and the final all-reduce (RCCL) are
    one stream-ordered sequence.  Returns the reduced device table (the caller synchronises when it reads it)."""
    if table is None:
        table = torch.zeros(len(points), len(TABLE_COLS), dtype=torch.float64, device=device)
    else:
        table.zero_()
    for p in shard(points, rank, world):
        evaluate_into(p, table[p.index])
    reduce_table(table, world, group)
    return table


def ber_loss(table: np.ndarray):
    """(BER, mean cross entropy) per point from the reduced table, float64."""
    conf = table[:, :4]
    tot = conf.sum(axis=1)
    ber = (conf[:, 1] + conf[:, 2]) / np.maximum(tot, 1.0)
    loss = table[:, 4] / np.maximum(table[:, 5], 1.0)
    return ber, loss


def write_csv(path: str, snrs: Sequence[float], ber: Sequence[float], loss: Sequence[float]):
    """The reference's result file: index column SNR, then BER, Loss (ofdmreceiver_np.py:70,85-89)."""
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["SNR", "BER", "Loss"])
        for s, b, l in zip(snrs, ber, loss):
            w.writerow([float(s), float(b), float(l)])
