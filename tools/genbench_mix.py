"""launch time of the fused generator (+ apply) against the launch-per-stage chain, by channel, at the equaliser's batch of 73 frames"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_ofdm_amd import ofdm, receiver as R      # noqa: E402
from dl_ofdm_amd.datagen import DeviceDataGen, FusedStaticGen      # noqa: E402


def timeit(fn, n=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t0) / n * 1e6
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n, host


for chan in ("EPA", "mixRayleigh"):
    for n in (73,):
        F = R.Flags(nbits=2, nfilter=64, channel=chan, SNR=10.0)
        o = ofdm.ofdm_tx(F)
        gen = DeviceDataGen(F, o, seed=1)
        fg = FusedStaticGen(gen, n, 10.0, want_noise_power=True)
        x = torch.empty(n, gen.S, gen.K + gen.CP, 2, device="cuda")
        bits = torch.empty(n, o.frame_size, 2, dtype=torch.int32, device="cuda")
        hshape = (n, gen.S, gen.K, 2) if gen.mixed else (n, gen.K, 2)
        H = torch.empty(hshape, device="cuda")
        snr = torch.full((n,), 10.0, device="cuda")
        print(chan, n, "L", [p["L"] for p in gen.profiles] if gen.mixed else gen.L)
        print("  fused + apply, with H : %.2f us (host %.1f)" % timeit(lambda: fg.make_batch(x, bits, 0, out_H=H, snr=snr)))
        print("  fused + apply, no H   : %.2f us (host %.1f)" % timeit(lambda: fg.make_batch(x, bits, 0, snr=snr)))

        def chain(want_H):
            tx, _ = gen.transmit(n, out_bits=bits)
            gen.channel(tx, snr, out_x=x, out_H=H if want_H else None)
            gen.offset += 1
        print("  launch per stage, H   : %.2f us (host %.1f)" % timeit(lambda: chain(True)))
        print("  launch per stage, no H: %.2f us (host %.1f)" % timeit(lambda: chain(False)))
