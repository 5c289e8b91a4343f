"""BASELINE.json config[4]: all 4 modulations x 3 fading profiles (EPA, EVA, ETU) x 40 SNRs (-10..29 dB), DCCN
(receiver trained on AWGN + equaliser trained on mixRayleigh, the reference driver's recipe) next to the classical
LMMSE / LS receivers of dl_ofdm_amd/benchmark.py.  The (modulation, channel, SNR) points are dealt round-robin to the
ranks of a torch.distributed job (one process per GPU); single process without one.

    python tools/config5_sweep.py --out profiles/r01_config5 [--frames 20000] [--eq_epochs 600]
"""
import argparse
import copy
import csv
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_ofdm_amd import benchmark, ofdm, receiver as R, receiver_mp as H, sweep     # noqa: E402
from dl_ofdm_amd.datagen import DeviceDataGen                                        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="config5_out")
    ap.add_argument("--frames", type=int, default=20000)
    ap.add_argument("--eq_epochs", type=int, default=600)
    ap.add_argument("--classical_frames", type=int, default=1500)
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl")
    os.makedirs(a.out, exist_ok=True)
    channels, snrs = ("EPA", "EVA", "ETU"), list(range(-10, 30))
    t0 = time.time()
    trainers = {}
    for nbits in (1, 2, 3, 4):                      # every rank trains the (small) models itself: seconds each
        save = os.path.join(a.out, "ckpt_r%d/" % rank)
        rf = R.Flags(nbits=nbits, nfilter=64, channel="AWGN", SNR=5.0 * nbits, max_epoch_num=1200 * nbits, early_stop=200,
                     token="C5_%dmod" % nbits, save_dir=save, device_data=True, seed=nbits)
        res = R.train(rf, verbose=False, run_test=False)
        hf = H.Flags(nbits=nbits, nfilter=64, channel="mixRayleigh", max_epoch_num=a.eq_epochs, early_stop=200,
                     token=rf.token, save_dir=save, device_data=True, seed=10 + nbits, test_frames=a.frames)
        out = H.train(hf, verbose=False, run_test=False, rx_params=res["params"])
        H.load_checkpoint(out["best_path"], out["trainer"], with_optimizer=False)
        trainers[nbits] = (hf, out["trainer"])
        if rank == 0:
            print("nbits %d: receiver %d epochs, equaliser %d epochs, %.0f s" % (nbits, len(res["history"]),
                                                                              len(out["history"]), time.time() - t0))
    pts = sweep.make_points([1, 2, 3, 4], channels, snrs, base_seed=77)
    gens = {}

    def evaluate(p):
        hf, tr = trainers[p.nbits]
        key = (p.nbits, p.channel)
        if key not in gens:
            fl = copy.deepcopy(hf)
            fl.channel = p.channel
            gens[key] = DeviceDataGen(fl, ofdm.ofdm_tx(fl), device=tr.device, seed=p.seed)
        g, pl = gens[key], tr.resident(a.frames)
        g.seed, g.offset = p.seed, 0
        g.make_batch(a.frames, p.snr_db, out_x=pl.x, out_bits=pl.bits)
        pl.run(False)
        m = tr._metrics(pl.metrics_buf, pl.tx_power)
        c = m["conf"]
        return [c[0][0], c[0][1], c[1][0], c[1][1], m["ce_sum"], m["count"]]

    table = sweep.run_sweep(pts, evaluate, rank, world, device=torch.device("cuda"))
    ber, _ = sweep.ber_loss(table)
    if rank == 0:
        print("DCCN sweep done: %d points, %.0f s" % (len(pts), time.time() - t0))
        classical = {}
        for nbits in (1, 2, 3, 4):
            for ch in channels:
                fl = R.Flags(nbits=nbits, channel=ch)
                for m in ("LMMSE", "LS-Spline", "Perfect"):
                    classical[(nbits, ch, m)] = benchmark.ber_curve(fl, m, snrs[::3], n_frames=a.classical_frames, seed=5)
        with open(os.path.join(a.out, "config5_ber.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["modulation", "channel", "SNR", "DCCN+Equalizer", "LMMSE", "LS-Spline", "Perfect"])
            for p in pts:
                row = [benchmark.MOD_NAMES[p.nbits - 1], p.channel, int(p.snr_db), "%.6g" % ber[p.index]]
                if int(p.snr_db) in snrs[::3]:
                    j = snrs[::3].index(int(p.snr_db))
                    row += ["%.6g" % classical[(p.nbits, p.channel, m)][j] for m in ("LMMSE", "LS-Spline", "Perfect")]
                else:
                    row += ["", "", ""]
                w.writerow(row)
        print("wrote %s, total %.0f s" % (os.path.join(a.out, "config5_ber.csv"), time.time() - t0))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
