"""launch time of the fused static-channel generator (HIP events over a same-kernel loop); DCCN_GEN_ABL=<bits> for ablations"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_ofdm_amd import ofdm, receiver as R      # noqa: E402
from dl_ofdm_amd.datagen import DeviceDataGen, FusedStaticGen      # noqa: E402
import ctypes as C      # noqa: E402

F = R.Flags(nbits=2, nfilter=64, channel="EPA", SNR=10.0)
o = ofdm.ofdm_tx(F)
gen = DeviceDataGen(F, o, seed=1)
fg = FusedStaticGen(gen, 1170, 10.0)
bits = torch.empty(1170, o.frame_size, 2, dtype=torch.int32, device="cuda")
st = gen._stream()
for _ in range(50):
    gen.lib.dccn_gen_static_frames(C.byref(fg.arm(bits)), st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(300):
    gen.lib.dccn_gen_static_frames(C.byref(fg.arm(bits)), st)
e1.record()
torch.cuda.synchronize()
print("DCCN_GEN_ABL=%s: %.2f us per launch" % (os.environ.get("DCCN_GEN_ABL", "0"), e0.elapsed_time(e1) * 1e3 / 300))
