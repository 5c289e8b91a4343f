#!/usr/bin/env python
"""Does the high-SNR BER floor of DCCN + equaliser on EVA / ETU go away when its two suspected causes are removed?
(DESIGN.md section 11 argued, this tests.)

One modulation, the basic receiver trained once with the reference driver's recipe (AWGN at 5 dB per bit), then the
equaliser trained three ways -- always with the reference's 10-point training-SNR distribution
(dev/py/ofdmreceiver_np_mp.py:386,405) and its schedule (4000 * nbits epochs cap, early stop 200):
    mix          on mixRayleigh (the reference driver's stage 2: dev/py/run_local_ofdm.py:96-118)
    chan         on the test channel itself
    chan+align   on the test channel with every frame delayed by the channel's centre-tap advance (datagen.align_window:
                 what a receiver with timing synchronisation sees -- the classical columns are given the same alignment)
and swept on the test channel at --snrs with 20 000 frames per point, next to LMMSE / perfect CSI (aligned window).

    python tools/floor_test.py --nbits 2 --channel EVA --out gpurun_out/floor
"""
import argparse
import copy
import csv
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nbits", type=int, default=2)
    ap.add_argument("--channel", default="EVA")
    ap.add_argument("--snrs", default="5,11,20,29,40")
    ap.add_argument("--frames", type=int, default=20000)
    ap.add_argument("--eq_epochs", type=int, default=0)
    ap.add_argument("--variants", default="mix,chan,chan+align")
    ap.add_argument("--out", default="gpurun_out/floor")
    a = ap.parse_args()
    import torch
    from dl_ofdm_amd import config5, ofdm, receiver as R, receiver_mp as H, sweep
    from dl_ofdm_amd.datagen import DeviceDataGen
    os.makedirs(a.out, exist_ok=True)
    nb, snrs = a.nbits, [float(s) for s in a.snrs.split(",")]
    import tempfile
    save = tempfile.mkdtemp(prefix="dccn_floor_") + "/"        # (checkpoints stay out of the result directory)
    t0 = time.time()
    rf = R.Flags(nbits=nb, nfilter=64, channel="AWGN", SNR=5.0 * nb, max_epoch_num=1200 * nb, early_stop=200,
                 token="floor_%dmod" % nb, save_dir=save, device_data=True, seed=nb)
    res = R.train(rf, verbose=False, run_test=False)
    log = {"receiver": {"epochs": len(res["history"]), "seconds": time.time() - t0,
                        "best_train_loss": min(h["train_loss"] for h in res["history"])}}
    rows = {}
    for var in a.variants.split(","):
        t1 = time.time()
        chan = "mixRayleigh" if var == "mix" else a.channel
        hf = H.Flags(nbits=nb, nfilter=64, channel=chan, max_epoch_num=a.eq_epochs if a.eq_epochs > 0 else 4000 * nb,
                     early_stop=200, token="floor_%s" % var.replace("+", "_"), save_dir=save, device_data=True, seed=10 + nb,
                     test_frames=a.frames, align_window=(var == "chan+align"))
        out = H.train(hf, verbose=False, run_test=False, rx_params=res["params"])
        H.load_checkpoint(out["best_path"], out["trainer"], with_optimizer=False)
        tr = out["trainer"]
        hist = out["history"]
        best = min(hist, key=lambda h: h["train_loss"])
        log[var] = {"epochs": len(hist), "seconds": time.time() - t1, "best_train_loss": best["train_loss"],
                    "best_epoch": best["epoch"], "test_ber_at_best": best["test_ber"]}
        fl = copy.deepcopy(hf)
        fl.channel = a.channel
        gen = DeviceDataGen(fl, ofdm.ofdm_tx(fl), device=tr.device, seed=77)
        gen.want_noise_power = False
        pl = tr.resident(a.frames)
        ber = []
        for i, snr in enumerate(snrs):
            gen.seed, gen.offset = 77 + 13 * i, 0
            gen.make_batch(a.frames, snr, out_x=pl.x, out_bits=pl.bits)
            pl.run(False)
            ber.append(tr._metrics(pl.metrics_buf, pl.tx_power)["berlin"])
        rows["DCCN+Equalizer[%s]" % var] = ber
        print(var, log[var], ["%.3g" % b for b in ber], flush=True)
        del tr, out, pl
        torch.cuda.empty_cache()
    cl = config5.classical_curves([nb], [a.channel], snrs, 1500, device="cuda", methods=("LMMSE", "Perfect"))
    for m in ("LMMSE", "Perfect"):
        rows["%s (aligned window)" % m] = list(cl[(nb, a.channel, m)])
    with open(os.path.join(a.out, "floor_%dmod_%s.csv" % (nb, a.channel)), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["receiver"] + ["%g dB" % s for s in snrs])
        for k, v in rows.items():
            w.writerow([k] + ["%.6g" % x for x in v])
    log["total_seconds"] = time.time() - t0
    json.dump(log, open(os.path.join(a.out, "floor_%dmod_%s.json" % (nb, a.channel)), "w"), indent=1)
    print(json.dumps(log))


if __name__ == "__main__":
    main()
