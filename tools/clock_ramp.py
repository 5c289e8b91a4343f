"""Diagnostic: us/step of the captured training step over time (500 replays per sample).
   python tools/clock_ramp.py [samples] [prob(0/1)] [devrand(0/1)]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_ofdm_amd.engine import RxEngine, RxDims, HipTimer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
prob = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
devrand = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
d = RxDims(S=7, kin=80, F=64, D=320, nbits=2)
eng = RxEngine(d, 1170, train=True, want_prob=prob, seed=1)
if devrand:
    g = torch.Generator(device="cuda"); g.manual_seed(1234)
    eng.x.copy_(torch.randn(eng.x.shape, generator=g, device="cuda"))
    eng.bits.copy_(torch.randint(0, 2, eng.bits.shape, generator=g, device="cuda", dtype=torch.int32))
else:
    eng.set_batch(torch.randn(1170, 7, 80, 2), torch.randint(0, 2, (1170, 320, 2)))
eng.train_step(graph=True); torch.cuda.synchronize()
t = HipTimer(); st = torch.cuda.current_stream().cuda_stream
out = []
for i in range(n):
    t.start(st)
    for _ in range(500):
        eng.train_step(graph=True)
    t.stop(st)
    out.append(round(t.elapsed_ms() / 500 * 1e3, 1))
print("prob", prob, "devrand", devrand, "us/step:", out, flush=True)
