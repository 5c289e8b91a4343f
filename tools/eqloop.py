#!/usr/bin/env python
"""Wall time per step of the equaliser's on-device training loop (dl_ofdm_amd/receiver_mp.py::_train_on_device) and where
it goes: the loop as the harness runs it, the same loop without the torch-side monitors, the generator alone, the fused step
alone.  One JSON line per variant.

    python tools/eqloop.py [--nbits 2] [--channel EPA] [--steps 300]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nbits", type=int, default=2)
    ap.add_argument("--channel", default="EPA")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--out", default="")
    ap.add_argument("--fused", type=int, default=1, help="0: the launch-per-stage generator chain in the loop variants")
    ap.add_argument("--graph", type=int, default=0, help="1: the loop variants replay the step's hipGraph instead of issuing it eagerly")
    ap.add_argument("--virtual", type=int, default=1, help="0: the pipelined loop materialises every batch (no x_next_virtual)")
    args = ap.parse_args()
    import numpy as np
    import torch
    from dl_ofdm_amd import ofdm, receiver as R, receiver_mp as M
    from dl_ofdm_amd.datagen import DeviceDataGen
    from dl_ofdm_amd.engine import glorot_init
    from dl_ofdm_amd.equalizer import EqualizerTrainer
    F = M.Flags(nbits=args.nbits, channel=args.channel, nfilter=64, device_data=True, seed=1) if hasattr(M, "Flags") else None
    if F is None:
        F = M.parse_flags(["--nbits=%d" % args.nbits, "--channel=%s" % args.channel, "--device_data=True"])
    F.fused_generator = bool(args.fused)
    F.virtual_next = bool(args.virtual)
    F.step_graph = bool(args.graph)
    o = ofdm.ofdm_tx(F)
    rx_params = glorot_init(R.rx_dims(F, o), 1)
    tr = EqualizerTrainer(F, o, rx_params, device="cuda", seed=1)
    gen = DeviceDataGen(F, o, device=tr.device, seed=1, mobile=False, mix=False)
    B = F.batch_size // F.nsymbol
    pl = tr.resident(B)
    mview = pl.metrics_buf.view(torch.float32)
    acc = torch.zeros(5, dtype=torch.float32, device=tr.device)
    out = open(args.out, "a") if args.out else None

    def full():
        snr = np.random.choice(M.TRAIN_SNR_GRID, [B], p=M.TRAIN_SNR_PROB)
        tx, _ = gen.transmit(B, out_bits=pl.bits)
        _, npow, H = gen.channel(tx, snr, out_x=pl.x, want_H=True)
        gen.offset += 1
        pl.run(True)
        chan_gt = H if H.dim() == 3 else H[:, None, :].expand(-1, F.nsymbol, -1)
        rms = tr.chan_rms(torch.view_as_complex(pl.chest), chan_gt)
        acc[0:2].add_(mview[12:14]); acc[2:3].add_(pl.tx_power); acc[3:4].add_(npow); acc[4:5].add_(rms)

    def no_monitor():
        snr = np.random.choice(M.TRAIN_SNR_GRID, [B], p=M.TRAIN_SNR_PROB)
        tx, _ = gen.transmit(B, out_bits=pl.bits)
        gen.channel(tx, snr, out_x=pl.x, want_H=True)
        gen.offset += 1
        pl.run(True)

    def gen_only():
        tx, _ = gen.transmit(B, out_bits=pl.bits)
        gen.channel(tx, 10.0, out_x=pl.x, want_H=False)
        gen.offset += 1

    def step_graph():
        pl.run(True)

    def step_eager():
        pl.run(True, graph=False)

    variants = [("harness_loop", full), ("no_monitor", no_monitor), ("generator_only", gen_only), ("step_graph", step_graph),
                ("step_eager", step_eager)]
    if hasattr(M, "device_epoch_runner"):
        variants.insert(0, ("r04_loop_generator_on_side_stream", M.device_epoch_runner(F, o, tr, gen, pl, overlap=True)))
        try:
            variants.insert(0, ("r04_loop_next_batch_normalised_on_the_optimizer_launch",
                                M.device_epoch_runner(F, o, tr, gen, pl, pipeline=True)))
            variants.insert(0, ("r04_loop", M.device_epoch_runner(F, o, tr, gen, pl, pipeline=False)))
        except TypeError:                                             # (an older package)
            variants.insert(0, ("r04_loop", M.device_epoch_runner(F, o, tr, gen, pl)))
    for name, fn in variants:
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        rec = {"variant": name, "frames": B, "ms_per_step_wall": round(t_all / args.steps * 1e3, 4),
               "ms_per_step_host_issue": round(t_host / args.steps * 1e3, 4)}
        line = json.dumps(rec)
        print(line, flush=True)
        if out:
            out.write(line + "\n")


if __name__ == "__main__":
    main()
