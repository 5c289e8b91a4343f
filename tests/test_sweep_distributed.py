"""The multi-GPU path of the BER sweep on CPU: world_size-2 gloo, evaluator = the oracle (tests may
use it as the checker/evaluator; the GPU evaluator is dl_ofdm_amd.receiver.test_model)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dl_ofdm_amd import sweep
from oracle import dccn_oracle as O

CFG = O.RxConfig(S=7, kin=20, F=8, D=24, nbits=2)


def _evaluate(point):
    """Deterministic per-point evaluation: data and parameters depend only on the point."""
    cfg = O.RxConfig(S=7, kin=20, F=8, D=24, nbits=point.nbits)
    rng = np.random.RandomState(point.seed)
    p = O.init_params(cfg, seed=11)
    noise = 10.0 ** (-point.snr_db / 20.0)
    x = (rng.randn(40, 7, 20, 2) * (1.0 + noise)).astype(np.float32)
    bits = rng.randint(0, 2, (40, 24, point.nbits))
    lb = O.rx_eval(p, x, bits, cfg)
    c = lb["conf"]
    return [c[0, 0], c[0, 1], c[1, 0], c[1, 1], float(lb["ce"].sum()), float(lb["ce"].size)]


def _points():
    return sweep.make_points([1, 2], ["AWGN", "EPA"], range(-2, 3), base_seed=5)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pts = _points()
        mine = sweep.shard(pts, rank, world)
        table = sweep.run_sweep(pts, _evaluate, rank, world)
        q.put((rank, [p.index for p in mine], table))
    finally:
        dist.destroy_process_group()


def test_shard_is_a_partition():
    pts = _points()
    assert len(pts) == 20 and [p.index for p in pts] == list(range(20))
    for world in (1, 2, 3, 8):
        owned = sorted(p.index for r in range(world) for p in sweep.shard(pts, r, world))
        assert owned == list(range(20))
        sizes = [len(sweep.shard(pts, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
    assert len({p.seed for p in pts}) == 20


def test_world2_gloo_matches_serial():
    serial = sweep.run_sweep(_points(), _evaluate)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = {}
    for rank, idx, table in results:
        owned[rank] = idx
        assert np.array_equal(table[:, :4], serial[:, :4]) and np.array_equal(table[:, 5], serial[:, 5])   # integer counts: exact
        np.testing.assert_allclose(table[:, 4], serial[:, 4], rtol=1e-12)
    assert sorted(owned[0] + owned[1]) == list(range(20)) and not set(owned[0]) & set(owned[1])
    ber, loss = sweep.ber_loss(serial)
    assert ber.shape == (20,) and np.all((ber >= 0) & (ber <= 1)) and np.all(loss > 0)


def _worker_unshared(rank, world, port, q):
    """Each rank runs its OWN, different, unsharded sweeps while a process group exists (the experiment driver's stage 2:
    one configuration per rank) -- rank 1 even runs one sweep more than rank 0."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tables = []
        for k in range(1 + rank):
            pts = sweep.make_points([1 + rank], ["AWGN"], range(-1 - k, 2), base_seed=50 + 7 * rank + k)
            tables.append((len(pts), sweep.run_sweep(pts, _evaluate)))          # world defaults to 1: no collective
        dist.barrier()
        q.put((rank, tables))
    finally:
        dist.destroy_process_group()


def test_unsharded_sweeps_do_not_reduce_across_an_existing_process_group():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_unshared, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0                                             # unequal sweep counts must not dead-lock
    for rank in (0, 1):
        assert len(results[rank]) == 1 + rank
        for k, (n, table) in enumerate(results[rank]):
            pts = sweep.make_points([1 + rank], ["AWGN"], range(-1 - k, 2), base_seed=50 + 7 * rank + k)
            serial = sweep.run_sweep(pts, _evaluate)
            assert n == len(pts) and np.array_equal(table, serial)         # exactly this rank's own numbers
    import pytest
    with pytest.raises(RuntimeError):
        sweep.reduce_table(torch.zeros(2, 6, dtype=torch.float64), world=2)   # sharded but no group: loud


def test_csv_and_harness_helpers(tmp_path):
    from dl_ofdm_amd import receiver
    path = tmp_path / "Test_DCCN_tok_AWGN.csv"
    sweep.write_csv(str(path), [-10, -9], [0.25, 0.125], [0.7, 0.6])
    assert path.read_text().splitlines() == ["SNR,BER,Loss", "-10.0,0.25,0.7", "-9.0,0.125,0.6"]
    F = receiver.Flags()
    assert (F.nbits, F.msg_length, F.batch_size, F.nfilter, F.SNR, F.early_stop, F.channel, F.cp, F.longcp, F.token) == \
        (1, 100800, 512, 80, 3.0, 100, "EPA", True, True, "OFDM")                 # ofdmreceiver_np.py:30-53
    F2 = receiver.parse_flags(["--nbits=2", "--cp=False", "--channel=AWGN", "--SNR=10"])
    assert F2.nbits == 2 and F2.cp is False and F2.channel == "AWGN" and F2.SNR == 10.0
    # :242  int(min(200/max(ber,1e-6), 9e5)/(55*nbits))//8
    assert receiver.ideal_batch_size(0.5, 1) == 0 and receiver.ideal_batch_size(1e-3, 2) == 227
    assert receiver.ideal_batch_size(0.0, 1) == 2045
    from dl_ofdm_amd import run_local_ofdm
    cfgs = run_local_ofdm.configurations()
    assert len(cfgs) == 16
    f0 = cfgs[0][0]
    assert (f0.nbits, f0.SNR, f0.max_epoch_num, f0.cp, f0.longcp, f0.token) == (4, 20.0, 4800, False, False,
                                                                                    "OFDM_Dense3_4mod_snr20_cpFalse")


def test_config5_unit_ownership_over_ranks():
    """who does what in BASELINE config[4] as the rank count grows (dl_ofdm_amd/config5.py): one training chain per
    modulation, longest first to the least-loaded rank; the classical units go to the ranks WITHOUT a chain when there are
    any (they run while the chains train), to every rank otherwise.  Every unit has exactly one owner."""
    from dl_ofdm_amd import config5
    nb = (1, 2, 3, 4)
    assert config5.job_owners(nb, 1) == {4: 0, 3: 0, 2: 0, 1: 0}
    assert config5.job_owners(nb, 2) == {4: 0, 3: 1, 2: 1, 1: 0}
    assert config5.job_owners(nb, 4) == {4: 0, 3: 1, 2: 2, 1: 3}
    assert config5.job_owners(nb, 8) == {4: 0, 3: 1, 2: 2, 1: 3}
    assert config5.classical_workers(nb, 1) == [0]
    assert config5.classical_workers(nb, 2) == [0, 1]
    assert config5.classical_workers(nb, 4) == [0, 1, 2, 3]
    assert config5.classical_workers(nb, 8) == [4, 5, 6, 7]
    assert config5.classical_workers((2,), 2) == [1]
    for world in (1, 2, 4, 8):
        w = config5.classical_workers(nb, world)
        owners = [w[u % len(w)] for u in range(4 * 3 * 3 * 14)]
        assert set(owners) == set(w) and all(0 <= r < world for r in owners)


def test_ranks_sharing_a_gpu_get_one_hardware_queue_each(monkeypatch):
    """config5.shared_gpu_env: only when more local ranks than GPUs, never over a value the user set; the GPU count comes
    from the visibility variables / render nodes, not from a HIP call (the runtime reads its queue limit at initialisation)"""
    from dl_ofdm_amd import config5
    assert config5.shared_gpu_env(4, 1, env={}) == {"GPU_MAX_HW_QUEUES": "1"}
    assert config5.shared_gpu_env(8, 8, env={}) == {}
    assert config5.shared_gpu_env(1, 1, env={}) == {}
    assert config5.shared_gpu_env(4, 0, env={}) == {}                       # no GPU visible: nothing to tune
    assert config5.shared_gpu_env(4, 1, env={"GPU_MAX_HW_QUEUES": "2"}) == {}
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,3")
    assert config5.visible_gpus_without_hip() == 2
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert config5.visible_gpus_without_hip() == 1
