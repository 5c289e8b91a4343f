"""End-to-end training throughput of the basic-receiver harness: data generation + fused step, host-generated
(NumPy substrate, the reference's way) vs device-generated (dl_ofdm_amd.datagen).

    python tools/e2ebench.py [--frames 1170] [--steps 200]

Prints one JSON line per mode: OFDM symbols/s including the generation of every batch.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_ofdm_amd import ofdm, radio, receiver as R      # noqa: E402
from dl_ofdm_amd.datagen import DeviceDataGen           # noqa: E402
from dl_ofdm_amd.engine import RxEngine                 # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1170)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--host-steps", type=int, default=5)
    ap.add_argument("--channel", default="EPA")
    a = ap.parse_args()
    F = R.Flags(nbits=2, nfilter=64, channel=a.channel, SNR=10.0)
    o = ofdm.ofdm_tx(F)
    eng = RxEngine(R.rx_dims(F, o), a.frames, train=True, want_prob=False)
    gen = DeviceDataGen(F, o, seed=1)
    gen.want_noise_power = False
    # the training loop of dl_ofdm_amd.receiver.train (device_data): batch i+1 is generated into eng.x / the other label
    # slot before step i is issued, and normalised behind step i's Adam update
    def run(n, first):
        if first:
            gen.make_batch(a.frames, F.SNR, out_x=eng.x, out_bits=eng.label_slot(0))
            eng.prime()
        for i in range(n):
            gen.make_batch(a.frames, F.SNR, out_x=eng.x, out_bits=eng.label_slot((i + 1) & 1))
            eng.train_step_pipelined(slot=i & 1)
    from dl_ofdm_amd.datagen import SideStreamFeeder
    feed = SideStreamFeeder(eng, lambda slot: gen.make_batch(a.frames, F.SNR, out_x=eng.x, out_bits=eng.label_slot(slot)))

    def run_overlapped(n, first):
        if first:
            feed.first(0)
        for i in range(n):
            feed.next((i + 1) & 1)
            eng.train_step_pipelined(slot=i & 1, x_ready=feed.ready)
            feed.step_issued()

    from dl_ofdm_amd.datagen import FusedStaticGen
    modes = [("device-generated, one stream", run), ("device-generated, generator on a side stream", run_overlapped)]
    if FusedStaticGen.supported(gen, eng):
        fg = FusedStaticGen(gen, a.frames, F.SNR)
        cnt = [0]

        def run_fused(n, first):
            if first:
                cnt[0] = 0
            for _ in range(n):
                eng.train_step_generated(fg, slot=cnt[0] & 1)
                cnt[0] += 1
        modes.append(("device-generated, fused generator launch + virtual input of R0, one C call per batch", run_fused))
        side = (torch.cuda.Stream(), torch.cuda.Event(), torch.cuda.Event())

        def run_fused_side(n, first):
            if first:
                cnt[0] = 0
                eng._gen_side_primed = False
            for _ in range(n):
                eng.train_step_generated(fg, slot=cnt[0] & 1, side=side)
                cnt[0] += 1
        modes.append(("device-generated, fused generator launch on a side stream + virtual input of R0", run_fused_side))
    for mode, fn in modes:
        eng.drop_prefetch()
        gen.offset = 0
        fn(20, True)
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.3:
            fn(50, False)
            torch.cuda.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(a.steps, False)
        t_issue = (time.perf_counter() - t0) / a.steps
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(json.dumps(dict(mode=mode, channel=a.channel, frames=a.frames, ms_per_step=round(dt * 1e3, 4),
                              host_issue_ms_per_step=round(t_issue * 1e3, 4),
                              symbols_per_s=round(a.frames * 7 / dt), final_ce=round(eng.metrics()["ce_mean"], 4))))
    if a.host_steps <= 0:
        return
    eng.drop_prefetch()                    # the pipelined loop above left a normalised batch behind
    fading = radio.rayleigh_chan_lte(F, o.Fs)
    np.random.seed(1)
    t0 = time.perf_counter()
    for _ in range(a.host_steps):
        xs, ys, _ = R.make_batch(F, o, fading, a.frames, F.SNR)
        eng.train_step(xs, ys)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.host_steps
    print(json.dumps(dict(mode="host-generated (NumPy substrate)", channel=a.channel, frames=a.frames,
                          ms_per_step=round(dt * 1e3, 3), symbols_per_s=round(a.frames * 7 / dt))))


if __name__ == "__main__":
    main()
