#!/usr/bin/env python
"""Equaliser transfer learning: the GPU trainer (fp32 HIP kernels, fused hipGraph step) against an independent CPU model
(oracle/torch_ref.LiteralEqualizer: literal conv3d graph, float64 autograd, the oracle's TF-Adam) on IDENTICAL batches
from identical initial weights -- separates "what the reference's algorithm does" from "what this implementation does"
(VERDICT r01 item 7).  Data: the reference's recipe, BPSK receiver trained on AWGN, equaliser batches of 73 frames from
mixRayleigh with the per-frame SNR distribution of dev/py/ofdmreceiver_np_mp.py:387,407 (host substrate, seeded).

    python tools/eq_trajectory.py --steps 300 --out gpurun_out/eq_traj.csv
"""
import argparse
import csv
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--frames", type=int, default=73)
    ap.add_argument("--rx_epochs", type=int, default=300)
    ap.add_argument("--out", default="gpurun_out/eq_traj.csv")
    a = ap.parse_args()
    import torch
    from dl_ofdm_amd import ofdm, receiver as R, receiver_mp as H
    from dl_ofdm_amd.equalizer import EqualizerTrainer
    from oracle import dccn_oracle as O
    from oracle import equalizer_oracle as E
    from oracle.torch_ref import LiteralEqualizer, LiteralRx
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    nb = 1
    rf = R.Flags(nbits=nb, nfilter=64, channel="AWGN", SNR=5.0, max_epoch_num=a.rx_epochs, early_stop=200, token="traj",
                 save_dir="/tmp/eq_traj/", device_data=True, seed=1)
    res = R.train(rf, verbose=False, run_test=False)
    pr = res["params"]
    hf = H.Flags(nbits=nb, nfilter=64, channel="mixRayleigh", token="traj", save_dir="/tmp/eq_traj/", seed=11, mobile=True)
    tx = ofdm.ofdm_tx(hf)
    tr = EqualizerTrainer(hf, tx, pr, seed=3)
    ecfg = E.EqConfig(S=7, K=tx.K, CP=tx.CP, cp=True, pilot_size=tx.pilot_size,
                      pilot_carriers=tuple(int(v) for v in tx.pilotCarriers))
    rcfg = O.RxConfig(S=7, kin=80, F=64, D=tx.frame_size, nbits=nb)
    p0 = tr.get_params()
    shapes = E.param_shapes(ecfg)
    p_cpu = {k: p0[k].astype(np.float64).reshape(shapes[k]) for k in tr.names}
    st = O.adam_init({k: v.astype(np.float32) for k, v in p_cpu.items()})
    st.m = {k: np.zeros_like(v) for k, v in p_cpu.items()}
    st.v = {k: np.zeros_like(v) for k, v in p_cpu.items()}
    lit_rx = LiteralRx({k: v.astype(np.float64) for k, v in pr.items()}, rcfg, dtype=torch.float64, literal_conv=False)
    # third runner: the same CPU model in float32 -- how far does rounding alone carry two correct implementations apart?
    p_c32 = {k: v.astype(np.float32) for k, v in p_cpu.items()}
    st32 = O.adam_init(p_c32)
    lit_rx32 = LiteralRx({k: v.astype(np.float32) for k, v in pr.items()}, rcfg, dtype=torch.float32, literal_conv=False)
    fading = H.RayleighChanParallel(hf, tx.Fs, mobile=True, mix=True)
    rows, t0 = [], time.time()
    for step in range(a.steps):
        np.random.seed(1000 + step)
        snr = np.random.choice(H.TRAIN_SNR_GRID, [a.frames, 1], p=H.TRAIN_SNR_PROB)
        xs, ys, _, _ = H.make_batch(hf, tx, fading, a.frames, snr)
        m = tr.train_step(xs, ys, fused=True, graph=True)
        lit = LiteralEqualizer(p_cpu, lit_rx, ecfg)
        g, info = lit.forward_backward(xs.astype(np.float64), ys)
        O.adam_tf_step(p_cpu, {k: v.reshape(p_cpu[k].shape) for k, v in g.items()}, st)
        lit32 = LiteralEqualizer(p_c32, lit_rx32, ecfg, dtype=torch.float32)
        g32, info32 = lit32.forward_backward(xs.astype(np.float32), ys)
        O.adam_tf_step(p_c32, {k: v.reshape(p_c32[k].shape).astype(np.float32) for k, v in g32.items()}, st32)
        rows.append((step, m["ce_mean"], info["ce_mean"], m["berlin"], float(info["berlin"]), info32["ce_mean"]))
        if step % 25 == 0 or step == a.steps - 1:
            pg = tr.get_params()
            dmax = max(float(np.abs(pg[k].astype(np.float64).ravel() - p_cpu[k].ravel()).max()) for k in tr.names)
            d32 = max(float(np.abs(p_c32[k].astype(np.float64).ravel() - p_cpu[k].ravel()).max()) for k in tr.names)
            print("step %4d  ce_mean gpu %.6f cpu64 %.6f cpu32 %.6f | BER gpu %.5f cpu64 %.5f | max|dparam| gpu-cpu64 %.2e "
                  "cpu32-cpu64 %.2e | %.0f s" % (step, m["ce_mean"], info["ce_mean"], info32["ce_mean"], m["berlin"],
                                                 float(info["berlin"]), dmax, d32, time.time() - t0), flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["step", "ce_mean_gpu_fp32", "ce_mean_cpu_fp64", "ber_gpu", "ber_cpu_fp64", "ce_mean_cpu_fp32"])
        for r in rows:
            w.writerow([r[0], "%.7f" % r[1], "%.7f" % r[2], "%.6f" % r[3], "%.6f" % r[4], "%.7f" % r[5]])
    d = np.array([abs(r[1] - r[2]) for r in rows])
    d2 = np.array([abs(r[5] - r[2]) for r in rows])
    print("max |ce_mean gpu - cpu64| over %d steps: %.3e (last 50: %.3e); cpu32 - cpu64: %.3e (last 50: %.3e)"
          % (len(rows), d.max(), d[-50:].max(), d2.max(), d2[-50:].max()))


if __name__ == "__main__":
    main()
