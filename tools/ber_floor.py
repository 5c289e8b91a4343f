#!/usr/bin/env python
"""High-SNR BER floor of the DCCN receiver + equaliser on multipath channels.

Trains one modulation with the reference driver's own recipe (dev/py/run_local_ofdm.py:66-118: receiver on AWGN at
SNR = 5 dB per bit, up to 1200*nbits epochs, early stop 200; equaliser on mixRayleigh with opt 0, up to 4000*nbits epochs,
early stop 200, mobile as given), every batch drawn on the GPU, then measures on static EPA / EVA / ETU up to 60 dB:
    DCCN            the basic receiver alone (trained on AWGN: not expected to cope with multipath)
    DCCN+Equalizer  the full receiver
next to the classical receivers of dl_ofdm_amd.benchmark on the host substrate, each with the FFT window as radio.py
delivers the frame ("raw") and aligned to the causal channel response ("aligned").

    python tools/ber_floor.py --nbits 1 --out gpurun_out/ber_floor [--eq_epochs 4000] [--mobile true]
"""
import argparse
import copy
import csv
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nbits", type=int, default=1)
    ap.add_argument("--out", default="gpurun_out/ber_floor")
    ap.add_argument("--eq_epochs", type=int, default=-1, help="default 4000*nbits (the reference driver)")
    ap.add_argument("--early_stop", type=int, default=200)
    ap.add_argument("--mobile", default="true")
    ap.add_argument("--frames", type=int, default=20000)
    ap.add_argument("--classical_frames", type=int, default=1500)
    a = ap.parse_args()
    import torch
    from dl_ofdm_amd import benchmark as B, ofdm, receiver as R, receiver_mp as H
    from dl_ofdm_amd.datagen import DeviceDataGen
    from dl_ofdm_amd.engine import RxEngine
    nb = a.nbits
    mobile = a.mobile.lower() in ("1", "true", "yes")
    os.makedirs(a.out, exist_ok=True)
    save = os.path.join(a.out, "ckpt/")
    t0 = time.time()
    rf = R.Flags(nbits=nb, nfilter=64, channel="AWGN", SNR=5.0 * nb, max_epoch_num=1200 * nb, early_stop=200,
                 token="floor_%dmod" % nb, save_dir=save, device_data=True, seed=nb)
    res = R.train(rf, verbose=False, run_test=False)
    print("receiver: %d epochs, best train loss %.5f, %.0f s" % (len(res["history"]), min(h["train_loss"] for h in res["history"]),
                                                                  time.time() - t0), flush=True)
    hf = H.Flags(nbits=nb, nfilter=64, channel="mixRayleigh", max_epoch_num=a.eq_epochs if a.eq_epochs > 0 else 4000 * nb,
                 early_stop=a.early_stop, token=rf.token, save_dir=save, device_data=True, seed=10 + nb, mobile=mobile,
                 SNR=5.0 * nb)
    out = H.train(hf, verbose=False, run_test=False, rx_params=res["params"])
    H.load_checkpoint(out["best_path"], out["trainer"], with_optimizer=False)
    tr = out["trainer"]
    hist = out["history"]
    print("equaliser: %d epochs, best train loss %.5f (epoch %d), %.0f s" %
          (len(hist), min(h["train_loss"] for h in hist), int(np.argmin([h["train_loss"] for h in hist])), time.time() - t0),
          flush=True)
    with open(os.path.join(a.out, "equalizer_history_%dmod.csv" % nb), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["epoch", "train_loss", "test_loss", "test_ber"])
        for h in hist[:: max(1, len(hist) // 200)]:
            w.writerow([h["epoch"], "%.6f" % h["train_loss"], "%.6f" % h.get("test_loss", float("nan")),
                        "%.6g" % h.get("test_ber", float("nan"))])
    snrs = [0, 5, 10, 15, 20, 25, 29, 30, 40, 60]
    chans = ("EPA", "EVA", "ETU")
    rows = []
    o = ofdm.ofdm_tx(rf)
    eng = RxEngine(R.rx_dims(rf, o), a.frames, train=False, params=res["params"], want_prob=False)
    for ch in chans:
        fl = copy.deepcopy(hf)
        fl.channel = ch
        gen = DeviceDataGen(fl, o, device=tr.device, seed=123)
        cls = {}
        F = R.Flags(nbits=nb, channel=ch)
        for m in ("Perfect", "LMMSE", "LS-Spline", "ALMMSE-CP"):
            for aligned in (False, True):
                if m == "ALMMSE-CP" and not aligned:
                    continue
                cls[(m, aligned)] = B.ber_curve(F, m, snrs, n_frames=a.classical_frames, seed=5, aligned=aligned)
        for i, snr in enumerate(snrs):
            pl = tr.resident(a.frames)
            gen.seed, gen.offset = 1000 + i, 0
            gen.make_batch(a.frames, float(snr), out_x=pl.x, out_bits=pl.bits)
            pl.run(False)
            m_eq = tr._metrics(pl.metrics_buf, pl.tx_power)
            eng.x.copy_(pl.x)
            eng.bits.copy_(pl.bits)
            eng.eval_step()
            m_rx = eng.metrics()
            row = dict(modulation=B.MOD_NAMES[nb - 1], channel=ch, SNR=snr, DCCN=m_rx["berlin"], DCCN_Equalizer=m_eq["berlin"])
            for (m, aligned), v in cls.items():
                row["%s_%s" % (m, "aligned" if aligned else "raw")] = v[i]
            rows.append(row)
            print(" ".join("%s=%s" % (k, ("%.3g" % v) if isinstance(v, float) else v) for k, v in row.items()), flush=True)
    keys = list(rows[0].keys())
    with open(os.path.join(a.out, "ber_floor_%dmod.csv" % nb), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=keys)
        w.writeheader()
        for r in rows:
            w.writerow({k: ("%.6g" % v if isinstance(v, float) else v) for k, v in r.items()})
    print("total %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
