import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from dl_ofdm_amd.engine import RxDims, RxEngine
c = bench.CONFIGS["c2"]
eng = RxEngine(RxDims(7, 80, 64, 320, 2), 1170, train=True)
eng.x.normal_(); eng.bits.random_(0, 2)
for mode in ("graph", "eager"):
    g = mode == "graph"
    for _ in range(50): eng.train_step(graph=g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): eng.train_step(graph=g)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(mode, "host issue %.1f us/step, total %.1f us/step, OMP=%s, affinity=%d cpus" % ((t1 - t0) / 500 * 1e6, (t2 - t0) / 500 * 1e6, os.environ.get("OMP_NUM_THREADS"), len(os.sched_getaffinity(0))))
