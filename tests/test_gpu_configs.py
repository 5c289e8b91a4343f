"""BASELINE.json configurations as configurations (``-m gpu``): C4 at its full size, C5 scaled down but through the
same sharded code path, and the N > 1 launch of bench.py (ranks sharing the one GPU of the test box over gloo)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_c4_full_size_qpsk_n1024_585_frames():
    """BASELINE config[3]: QPSK, N=1024/CP=72, batch 4096 OFDM symbols = 585 frames, F=1024, D=4000 (475 GFLOP per
    step).  One training step at the full size against the float64 oracle -- R0 on the whole batch, every later stage on
    a 16-frame row subset (forward) / a 64-column subset (the two weight gradients), each stage evaluated on the GPU's
    own inputs to it -- then the size-independent properties: confusion counts add up, eager == graph replay bit for
    bit over two steps."""
    from dl_ofdm_amd.engine import RxEngine
    from oracle import dccn_oracle as O
    from test_gpu_engine import make_case, relerr
    batch, S, kin, F, D, nb = 585, 7, 1096, 1024, 4000, 2
    dims, cfg, x, bits, p = make_case(batch, nb, kin=kin, F=F, D=D, seed=8)
    eng = RxEngine(dims, batch, params=p, train=True, want_prob=True)
    eng.train_step(x, bits)
    torch.cuda.synchronize()
    m, g = eng.metrics(), eng.get_grads()
    tol = 1e-5                                                     # north_star's figure, also at K = 14336
    rng = np.random.RandomState(1)
    fr = np.sort(rng.choice(batch, 16, replace=False))
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    xn, _, _ = O.batch_moment_norm(x.reshape(batch, -1).astype(np.float64))
    xn = xn.reshape(x.shape)
    assert relerr(eng.x_norm.cpu().numpy(), xn) <= 1e-5                                          # R0, whole batch
    fft = O.cconv_gemm_fwd(xn[fr].reshape(16 * S, kin, 2), p64["fft_like/conv3d/kernel"], p64["fft_like/conv3d/bias"])
    fft_g = eng.fft_out.cpu().numpy()
    assert relerr(fft_g[fr].reshape(16 * S, F, 2), fft) <= tol                                   # R1
    a_g = fft_g.astype(np.float64).reshape(batch, -1)
    z = a_g[fr] @ p64["demodulation/dense/kernel"] + p64["demodulation/dense/bias"]              # R2 on the GPU's fft_out
    assert relerr(eng.z.cpu().numpy()[fr], z) <= tol
    zg = eng.z.cpu().numpy().astype(np.float64)
    tl = O.tail_forward_backward(zg[fr].reshape(16 * D, 2), bits[fr].reshape(16 * D, nb), p64["demodulation/conv2d/kernel"],
                                 p64["demodulation/conv2d/bias"], p64["demodulation/dense_1/kernel"],
                                 p64["demodulation/dense_1/bias"], nb)                            # R3-R6 on the GPU's z
    prob_g = eng.prob.cpu().numpy()
    assert relerr(prob_g[fr].reshape(-1, 2), tl["prob"].reshape(-1, 2)) <= 1e-5
    kink = (np.abs(tl["pre1"]).min(1) < 2e-6) | (np.abs(tl["pre2"]).min(1) < 2e-6)
    dz_g = eng.dz.cpu().numpy().astype(np.float64)
    dz_sub = tl["dz"] * (16.0 / batch)                             # the oracle averaged over its 16 frames only
    assert relerr(dz_g[fr].reshape(-1, 2)[~kink], dz_sub[~kink]) <= tol
    # loss / decisions over ALL cells from the GPU's probabilities (float64 restatement of ofdmreceiver_np.py:154-169)
    lb = O.loss_ber(prob_g.astype(np.float64), bits)
    assert abs(m["ce_mean"] - float(lb["ce_mean"])) <= 1e-5 * float(lb["ce_mean"])
    assert np.array_equal(np.array(m["conf"]), lb["conf"]) and int(np.sum(m["conf"])) == batch * D * nb == m["count"]
    # backward GEMMs on the GPU's dz / fft_out / x_norm / dfft
    dfft_g = eng.dfft.cpu().numpy().astype(np.float64)
    assert relerr(dfft_g[fr].reshape(16, -1), dz_g[fr] @ p64["demodulation/dense/kernel"].T) <= tol
    cols = np.sort(rng.choice(2 * D, 64, replace=False))
    assert relerr(g["demodulation/dense/kernel"][:, cols], a_g.T @ dz_g[:, cols]) <= tol
    assert relerr(g["demodulation/dense/bias"], dz_g.sum(0)) <= tol
    fsel = np.sort(rng.choice(F, 32, replace=False))
    xr = eng.x_norm.cpu().numpy().astype(np.float64).reshape(batch * S, kin, 2)
    d3 = dfft_g.reshape(batch * S, F, 2)[:, fsel, :]
    xi, xq, dre, dim = xr[..., 0], xr[..., 1], d3[..., 0], d3[..., 1]
    gk = g["fft_like/conv3d/kernel"]
    assert relerr(gk[:, fsel], xi.T @ dre - xq.T @ dim) <= tol                                   # dWa (Appendix A.2)
    assert relerr(gk[:, F + fsel], xi.T @ dim - xq.T @ dre) <= tol                               # dWb
    assert relerr(g["fft_like/conv3d/bias"][:F], (dfft_g.reshape(-1, F, 2)[..., 0] - dfft_g.reshape(-1, F, 2)[..., 1]).sum(0)) <= tol
    del eng, g, a_g, zg, dz_g, dfft_g, xr
    outs = []
    for graph in (False, True):
        e = RxEngine(dims, batch, params=p, train=True, want_prob=True)
        for _ in range(2):
            e.train_step(x, bits, graph=graph)
        torch.cuda.synchronize()
        outs.append((e.params.clone(), e.prob.clone(), e.metrics()["conf"]))
        del e
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]
    # the engine that never keeps the summed dense gradient (bench.py's: want_grads=False) trains to the same bits, and the
    # knobs removed in round 6 are refused
    from dl_ofdm_amd import _lib
    lib = _lib.load()
    ref_params = outs[0][0]
    del outs
    torch.cuda.empty_cache()
    e = RxEngine(dims, batch, params=p, train=True, want_prob=True, want_grads=False, want_z=False, want_dfft=False)
    for _ in range(2):
        e.train_step(x, bits)
    torch.cuda.synchronize()
    assert torch.equal(e.params, ref_params)
    del e
    torch.cuda.empty_cache()
    for key in (15, 16, 22, 23, 26):
        assert lib.dccn_set_tuning(key, 1) == -1


def test_bench_two_ranks_over_gloo_one_json_line():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` (the driver's N > 1 launch) with
    DCCN_BENCH_BACKEND=gloo so that both ranks may share this box's GPU: one JSON line from rank 0, n_gpus == 2, the
    reduced BER table holds the bits of both ranks' batches."""
    env = dict(os.environ, DCCN_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--no-cpu-baseline", "--no-kernel-times", "--sweep-frames", "1000"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["scaling"] == "weak" and out["unit"] == "OFDM symbols/s"
    # the sweep object: 40 points dealt 20 / 20, every point's bits arrive through the one all-reduce
    w = out["sweep"]
    assert w["n_gpus"] == 2 and w["points_per_rank"] == [20, 20] and w["bits_counted"] == 40 * 1000 * 320 * 4
    t = out["step"]["ber_table"]
    bits_per_rank = 1170 * 320 * 2
    assert t[5] == 2 * bits_per_rank and sum(t[:4]) == 2 * bits_per_rank
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                          "--no-kernel-times", "--sweep-frames", "1000"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    o1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    assert o1["n_gpus"] == 1 and o1["step"]["ber_table"][5] == bits_per_rank
    # same points, same seeds: the sharded sweep counts exactly the serial one's errors
    assert o1["sweep"]["points_per_rank"] == [40] and o1["sweep"]["bits_counted"] == w["bits_counted"]
    assert o1["sweep"]["ber_first_last"] == w["ber_first_last"]
    # whole-job value: both ranks' symbols over the slower rank's time
    assert out["value"] > 0 and abs(out["value"] - 2 * 6 * 8190 / (out["ms_per_step"] * 6e-3)) <= 1e-6 * out["value"]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (the shape of the driver's N = 1 command): the bench re-runs itself
    under torch.distributed.run, two ranks (gloo here, so that both may share this box's GPU), one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    env.update(DCCN_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline", "--no-kernel-times", "--no-other-configs", "--no-e2e", "--sweep-frames", "1000"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5
    assert out["distributed"]["world"] == 2 and out["sweep"]["points_per_rank"] == [20, 20]
    assert "NOT trained" in out["sweep"]["receiver"]


def test_bench_watchdog_turns_a_hang_into_an_exit_code():
    """a rendezvous that never completes (rank 1 of 2 never starts) ends with rc 3 and a message, not a hang"""
    env = {k: v for k, v in os.environ.items() if k not in ("LOCAL_WORLD_SIZE",)}
    env.update(DCCN_BENCH_BACKEND="gloo", DCCN_BENCH_INIT_TIMEOUT="8", RANK="0", WORLD_SIZE="2", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert "did not complete within" in r.stderr


def _two_gpus():
    return torch.cuda.is_available() and torch.cuda.device_count() >= 2


@pytest.mark.skipif(not _two_gpus(), reason="needs >= 2 GPUs: the RCCL (nccl backend) branch, one rank per GPU")
def test_bench_two_ranks_over_rccl_one_json_line():
    """the driver's N = 2 launch exactly as it issues it (backend nccl = RCCL over xGMI, one rank per GPU): one JSON line,
    whole-job value, the `distributed` object names backend / RCCL version / collective / points per rank, and the reduced
    tables hold both ranks' bits.  Skipped on 1-GPU boxes (the gloo twin above covers the control flow there)."""
    env = {k: v for k, v in os.environ.items() if k not in ("DCCN_BENCH_BACKEND", "DCCN_DIST_BACKEND")}
    env.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--no-cpu-baseline", "--no-kernel-times", "--sweep-frames", "1000"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    d = out["distributed"]
    assert out["n_gpus"] == 2 and d["world"] == 2 and d["backend"] == "nccl" and d["rccl_version"]
    assert d["points_per_rank"] == [20, 20] and "all-reduce" in d["collective"]
    bits_per_rank = 1170 * 320 * 2
    t = out["step"]["ber_table"]
    assert t[5] == 2 * bits_per_rank and sum(t[:4]) == 2 * bits_per_rank
    assert out["sweep"]["bits_counted"] == 40 * 1000 * 320 * 4 and "all-reduce" in out["sweep"]["collective"]


@pytest.mark.skipif(not _two_gpus(), reason="needs >= 2 GPUs: the RCCL (nccl backend) branch, one rank per GPU")
def test_config5_sweep_tool_over_rccl(tmp_path):
    """tools/config5_sweep.py --backend nccl on 2 ranks (scaled down): chains dealt to ranks, arenas broadcast over RCCL,
    one all-reduce of the table; the CSV equals the 1-rank run byte for byte."""
    env = {k: v for k, v in os.environ.items() if k not in ("DCCN_BENCH_BACKEND", "DCCN_DIST_BACKEND")}
    env.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--frames", "2000", "--eq_epochs", "4", "--rx_epoch_scale", "0.01", "--classical_frames", "40",
              "--snrs=-5,5,15,29", "--classical_every", "2"]
    tool = os.path.join(ROOT, "tools", "config5_sweep.py")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(_free_port()), tool, "--out", str(tmp_path / "two"), "--backend", "nccl"]
                         + common, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert two.returncode == 0, two.stderr[-3000:]
    one = subprocess.run([sys.executable, tool, "--out", str(tmp_path / "one")] + common, cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=1500)
    assert one.returncode == 0, one.stderr[-3000:]
    a = open(str(tmp_path / "one" / "config5_ber.csv")).read()
    b_ = open(str(tmp_path / "two" / "config5_ber.csv")).read()
    assert a == b_ and a.count("\n") >= 2
    tj = json.load(open(str(tmp_path / "two" / "config5_timing.json")))
    assert tj["world"] == 2 and tj.get("backend") == "nccl"


def test_config5_sweep_tool_share_ranks_on_one_gpu(tmp_path):
    """tools/config5_sweep.py --share 3 (scaled down): the tool re-launches itself as three gloo ranks that share the visible
    GPU, one hardware queue per process (config5.shared_gpu_env); the CSV equals the 1-rank run byte for byte."""
    env = {k: v for k, v in os.environ.items() if k not in ("DCCN_BENCH_BACKEND", "DCCN_DIST_BACKEND", "GPU_MAX_HW_QUEUES", "RANK",
                                                            "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0",
               HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", "0"))
    common = ["--frames", "1000", "--eq_epochs", "3", "--rx_epoch_scale", "0.01", "--classical_frames", "30",
              "--snrs=-5,10,29", "--classical_every", "2"]
    tool = os.path.join(ROOT, "tools", "config5_sweep.py")
    shared = subprocess.run([sys.executable, tool, "--share", "3", "--out", str(tmp_path / "shared")] + common, cwd=ROOT, env=env,
                            capture_output=True, text=True, timeout=1500)
    assert shared.returncode == 0, shared.stderr[-3000:]
    one = subprocess.run([sys.executable, tool, "--out", str(tmp_path / "one")] + common, cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=1500)
    assert one.returncode == 0, one.stderr[-3000:]
    assert open(str(tmp_path / "one" / "config5_ber.csv")).read() == open(str(tmp_path / "shared" / "config5_ber.csv")).read()
    tj = json.load(open(str(tmp_path / "shared" / "config5_timing.json")))
    assert tj["world"] == 3 and tj["backend"] == "gloo" and tj["hw_queues_per_process"] == "1"
    assert json.load(open(str(tmp_path / "one" / "config5_timing.json")))["hw_queues_per_process"] == "runtime default"


def test_config5_chains_on_streams_of_one_process_equal_the_serial_run(tmp_path):
    """config 5's training chains next to each other in ONE process (a host thread + HIP stream per chain: the default) against
    the same chains one after the other (--chain_streams 1), scaled down: every chain draws from generators of its own, so the
    CSV -- 4 trained receiver + equaliser pairs evaluated on every point -- is the same bytes."""
    env = {k: v for k, v in os.environ.items() if k not in ("DCCN_BENCH_BACKEND", "DCCN_DIST_BACKEND", "RANK", "WORLD_SIZE",
                                                            "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    common = ["--frames", "1000", "--eq_epochs", "5", "--rx_epoch_scale", "0.01", "--classical_frames", "30",
              "--snrs=-5,10,29", "--classical_every", "2"]
    tool = os.path.join(ROOT, "tools", "config5_sweep.py")
    runs = {}
    for name, extra in (("streams", []), ("nogroup", ["--chain_group", "0"]), ("serial", ["--chain_streams", "1"])):
        r = subprocess.run([sys.executable, tool, "--out", str(tmp_path / name)] + common + extra, cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stderr[-3000:]
        runs[name] = open(str(tmp_path / name / "config5_ber.csv")).read()
        tj = json.load(open(str(tmp_path / name / "config5_timing.json")))
        # default: the two chains with the largest epoch budgets on a stream each, the other two as one chain group on a third
        assert tj["per_rank_seconds"][0]["chain_streams"] == {"streams": 3, "nogroup": 4, "serial": 1}[name]
        assert tj["per_rank_seconds"][0]["chain_group"] == (2 if name == "streams" else 0)
    assert runs["streams"] == runs["serial"] == runs["nogroup"] and runs["serial"].count("\n") == 1 + 4 * 3 * 3


def _c5_worker(rank, world, port, out_dir, q, kw, classical_every=2):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from dl_ofdm_amd import config5
    r, w, local = config5.init_distributed("gloo")
    try:
        trainers = config5.train_models(out_dir, kw["nbits"], kw["frames"], kw["eq_epochs"], kw["scale"], rank=r,
                                        device="cuda:%d" % local, world=w)
        pts, table = config5.sweep_dccn(trainers, kw["nbits"], kw["channels"], kw["snrs"], kw["frames"], r, w)
        eq = {b: tr.params.detach().cpu().numpy().copy() for b, (_, tr) in trainers.items()}
        del trainers
        torch.cuda.empty_cache()
        # ... and the whole pipeline (training chains, sweep points, classical units all dealt to ranks) through run()
        _, ber = config5.run(out_dir + "_run", kw["frames"], kw["eq_epochs"], kw["classical_frames"], kw["scale"], kw["nbits"],
                             kw["channels"], kw["snrs"], classical_every=classical_every, rank=r, world=w,
                             device="cuda:%d" % local, verbose=False)
        q.put((r, table, eq, ber))
    finally:
        torch.distributed.destroy_process_group()


def test_c5_scaled_sharded_equals_serial_and_ber_falls_with_snr(tmp_path):
    """BASELINE config[4] (4 modulations x {EPA, EVA, ETU} x SNRs, DCCN receiver + equaliser next to the classical
    receivers) through dl_ofdm_amd.config5 at reduced training length and 4 SNRs, serial and on 2 ranks (gloo, sharing
    this box's GPU).  On 2 ranks every stage is dealt out -- training chains {16-QAM, BPSK} / {8-QAM, QPSK} with the
    trained arenas broadcast, sweep points and classical units round-robin -- and everything equals the serial run:
    equaliser parameters bitwise, the confusion table exactly, the CSV byte for byte.  BER is monotone in SNR."""
    import torch.multiprocessing as mp
    from dl_ofdm_amd import config5, sweep
    kw = dict(nbits=(1, 2, 3, 4), channels=config5.CHANNELS, snrs=(-5, 5, 15, 29), frames=2000, eq_epochs=6, scale=0.01,
              classical_frames=40)
    assert config5.job_owners(kw["nbits"], 2) == {4: 0, 3: 1, 2: 1, 1: 0} and set(config5.job_owners(kw["nbits"], 8).values()) == {0, 1, 2, 3}
    trainers = config5.train_models(str(tmp_path / "serial"), kw["nbits"], kw["frames"], kw["eq_epochs"], kw["scale"])
    pts, serial = config5.sweep_dccn(trainers, kw["nbits"], kw["channels"], kw["snrs"], kw["frames"])
    eq_serial = {b: tr.params.detach().cpu().numpy().copy() for b, (_, tr) in trainers.items()}
    assert len(pts) == 4 * 3 * 4 and serial.shape == (48, 6)
    del trainers
    torch.cuda.empty_cache()
    _, ber_run = config5.run(str(tmp_path / "serial_run"), kw["frames"], kw["eq_epochs"], kw["classical_frames"], kw["scale"],
                             kw["nbits"], kw["channels"], kw["snrs"], classical_every=2, verbose=False)
    torch.cuda.empty_cache()
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_c5_worker, args=(r, 2, port, str(tmp_path / "sharded"), q, kw)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in procs:
        r, table, eq, ber2 = q.get(timeout=1500)
        results[r] = (table, eq, ber2)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ber, _ = sweep.ber_loss(serial)
    for r in (0, 1):
        table, eq, ber2 = results[r]
        assert np.array_equal(table[:, :4], serial[:, :4]) and np.array_equal(table[:, 5], serial[:, 5])
        np.testing.assert_allclose(table[:, 4], serial[:, 4], rtol=1e-9)
        for b in kw["nbits"]:
            assert np.array_equal(eq[b], eq_serial[b]), (r, b)          # trained here or received by broadcast: same bits
        assert np.array_equal(ber2, ber_run) and np.array_equal(ber2, ber)
    a = open(str(tmp_path / "serial_run" / "config5_ber.csv")).read()
    b_ = open(str(tmp_path / "sharded_run" / "config5_ber.csv")).read()
    assert a == b_ and a.count("\n") == 49 and "LMMSE" in a.splitlines()[0]
    tj = json.load(open(str(tmp_path / "sharded_run" / "config5_timing.json")))
    assert tj["world"] == 2 and len(tj["per_rank_seconds"]) == 2
    assert "train_rx_4" in tj["per_rank_seconds"][0] and "train_rx_3" in tj["per_rank_seconds"][1]      # chains really were dealt out
    assert "train_rx_3" not in tj["per_rank_seconds"][0] and all("classical" in t and "sweep" in t for t in tj["per_rank_seconds"])
    assert np.all(serial[:, 5] == np.array([kw["frames"] * 320 * p.nbits for p in pts]))
    for b in range(4):
        for c in range(3):
            cur = ber[(b * 3 + c) * 4:(b * 3 + c) * 4 + 4]
            assert cur[0] > cur[-1] and np.all(np.diff(cur) <= 0.02), (b + 1, config5.CHANNELS[c], cur)
    assert ber[0] < 0.45 and ber[3] < ber[0]


def test_c5_full_size_sweep_sharded_equals_serial(tmp_path):
    """BASELINE.json configs[4] at FULL size under `-m gpu`: 4 modulations x {EPA, EVA, ETU} x 40 SNRs = 480 points of
    20 000 frames each (dev/py/run_local_ofdm.py:61-118, ofdmreceiver_np.py:59-91) plus the classical LMMSE / LS-Spline /
    perfect-CSI columns -- only the TRAINING is shortened (a handful of epochs per model; the full schedules are the
    builder-run profiles/r04_config5).  Serial and on 2 ranks (gloo, sharing this box's GPU): every point counts exactly
    frames x 320 x nbits bits, the 2-rank table equals the serial one exactly, the CSV byte for byte; BER curves do not
    rise with SNR; the perfect-CSI receiver is at least as good as every estimator on every curve."""
    import time
    import torch.multiprocessing as mp
    from dl_ofdm_amd import config5
    kw = dict(nbits=(1, 2, 3, 4), channels=config5.CHANNELS, snrs=config5.SNRS, frames=20000, eq_epochs=3, scale=0.004,
              classical_frames=300)
    t0 = time.time()
    pts, ber_run = config5.run(str(tmp_path / "serial_run"), kw["frames"], kw["eq_epochs"], kw["classical_frames"], kw["scale"],
                               kw["nbits"], kw["channels"], kw["snrs"], classical_every=3, verbose=False)
    t_serial = time.time() - t0
    torch.cuda.empty_cache()
    assert len(pts) == 480 and ber_run.shape == (480,)
    tj = json.load(open(str(tmp_path / "serial_run" / "config5_timing.json")))
    assert tj["points"] == 480 and tj["frames"] == 20000 and tj["per_rank_seconds"][0]["sweep"] < 60
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_c5_worker, args=(r, 2, port, str(tmp_path / "sharded"), q, kw, 3)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in procs:
        r, table, eq, ber2 = q.get(timeout=1500)
        results[r] = (table, ber2)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in (0, 1):
        table, ber2 = results[r]
        assert table.shape == (480, 6)
        assert np.all(table[:, 5] == np.array([kw["frames"] * 320 * p.nbits for p in pts]))     # bits_counted, every point
        assert np.array_equal(ber2, ber_run)
    a = open(str(tmp_path / "serial_run" / "config5_ber.csv")).read()
    b_ = open(str(tmp_path / "sharded_run" / "config5_ber.csv")).read()
    assert a == b_ and a.count("\n") == 481
    rows = [l.split(",") for l in a.splitlines()[1:]]
    for b in range(4):
        for c in range(3):
            cur = ber_run[(b * 3 + c) * 40:(b * 3 + c) * 40 + 40]
            assert cur[0] >= cur[-1] - 0.01 and np.all(np.diff(cur) <= 0.03), (b + 1, config5.CHANNELS[c], cur)
            # classical columns (every third SNR): perfect CSI <= LMMSE and <= LS-Spline (within sampling error), falling
            cl = np.array([[float(v) for v in r_[4:7]] for r_ in rows[(b * 3 + c) * 40:(b * 3 + c) * 40 + 40] if r_[4] != ""])
            assert cl.shape == (14, 3)
            assert np.all(cl[:, 2] <= cl[:, 0] + 0.01) and np.all(cl[:, 2] <= cl[:, 1] + 0.01), (b + 1, c, cl)
            assert cl[0, 2] > cl[-1, 2]
    print("full-size config 5 (short training): serial %.1f s, stages %s" % (t_serial, tj["per_rank_seconds"][0]))


def test_classical_receivers_on_device_generated_frames():
    """SURVEY.md 8(f-4) row on the GPU box: the LS / LMMSE / CP-enhanced baseline receivers of dl_ofdm_amd.benchmark fed
    with frames from the device-side generator (the data the DCCN sweeps use) reproduce, within sampling error, their BER
    on the host substrate's frames -- and perfect CSI (H from the generator's own channel kernel) beats the estimators."""
    from dl_ofdm_amd import benchmark as B, ofdm, radio, receiver as R
    from dl_ofdm_amd.datagen import DeviceDataGen
    for nbits, ch, snr in ((2, "EPA", 15.0), (1, "ETU", 20.0)):
        F = R.Flags(nbits=nbits, channel=ch, nfilter=64)
        o = ofdm.ofdm_tx(F)
        rxr = B.ClassicalReceiver(F, o)
        fading = radio.rayleigh_chan_lte(F, o.Fs)
        adv = rxr.advance_of(fading)
        gen = DeviceDataGen(F, o, device="cuda", seed=11)
        n = 1200
        x, bits, _, H = gen.make_batch(n, snr, want_H=True)
        x, bits = x.cpu().numpy(), bits.cpu().numpy()
        H = H.cpu().numpy()                                             # complex [n, K] (static channel)
        Hs = np.repeat(H[:, None, :], o.nSymbol, axis=1) if H.ndim == 2 else H
        dev_ber = {m: float(np.mean(rxr.receive(x, m, snr, H_true=Hs, R_long=rxr.long_term_correlation(fading),
                                                 advance=adv) != bits))
                   for m in ("Perfect", "LS-Spline", "ALMMSE", "LS-CP", "LMMSE-Fast")}
        host_ber = {m: float(B.ber_curve(F, m, [snr], n_frames=n, seed=3)[0]) for m in dev_ber}
        for m in dev_ber:
            tol = 5.0 * np.sqrt(max(host_ber[m], 1e-4) / (n * 320 * nbits) * 40.0) + 0.15 * host_ber[m]   # frames are correlated cells
            assert abs(dev_ber[m] - host_ber[m]) <= tol, (ch, m, dev_ber[m], host_ber[m])
        assert dev_ber["Perfect"] <= min(dev_ber["LS-Spline"], dev_ber["ALMMSE"], dev_ber["LS-CP"]) + 1e-4, dev_ber
