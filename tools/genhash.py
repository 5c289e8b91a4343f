"""checksums of the fused generator's outputs at fixed (seed, offset) -- to compare two builds of the library bit for bit
(DCCN_LIB_PATH=<other build> python tools/genhash.py)"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_ofdm_amd import ofdm, receiver as R      # noqa: E402
from dl_ofdm_amd.datagen import DeviceDataGen, FusedStaticGen      # noqa: E402


def h(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]


for chan, nbits, n in (("EPA", 2, 1170), ("ETU", 4, 73), ("AWGN", 1, 7), ("EVA", 3, 300)):
    F = R.Flags(nbits=nbits, nfilter=64, channel=chan, SNR=10.0)
    o = ofdm.ofdm_tx(F)
    gen = DeviceDataGen(F, o, seed=5)
    gen.offset = 9
    fg = FusedStaticGen(gen, n, torch.linspace(0, 25, n).numpy(), want_noise_power=True)
    x = torch.empty(n, gen.S, gen.K + gen.CP, 2, device="cuda")
    bits = torch.empty(n, o.frame_size, nbits, dtype=torch.int32, device="cuda")
    tx = torch.empty(n, gen.S, gen.K + gen.CP, 2, device="cuda")
    _, _, npow = fg.make_batch(x, bits, slot=0, tx_out=tx)
    torch.cuda.synchronize()
    print(chan, nbits, n, "x", h(x), "bits", h(bits), "tx", h(tx), "y", h(fg.y), "noise", h(fg.noise), "npow", h(npow))
