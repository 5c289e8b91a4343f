"""Time the equaliser transfer-learning step (dl_ofdm_amd.equalizer.EqualizerTrainer) on one GPU.

    python tools/eqbench.py [--frames 73 1170] [--steps 50]

Prints one JSON line per batch size: frames, ms/step (HIP events around `steps` steps, inputs resident),
OFDM symbols/s, model GFLOP/step (dense + C-Conv GEMMs, forward 1x + backward: 2x for trainable
layers, 1x for the frozen receiver's backward-to-input).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_ofdm_amd import ofdm, receiver_mp as H          # noqa: E402
from dl_ofdm_amd.engine import HipTimer, glorot_init    # noqa: E402
from dl_ofdm_amd.equalizer import EqualizerTrainer      # noqa: E402
from dl_ofdm_amd.receiver import rx_dims                 # noqa: E402


def step_flops(B, S=7, K=64, n_sc=80, F=64, D=320):
    SK2 = S * K * 2
    eq = (B * S * (2 * n_sc) * (2 * K)                      # dense
          + 3 * B * S * (2 * K) * (2 * K)                   # three (1,K) C-Convs
          + 2 * B * SK2 * 32                                # pilot bottleneck in/out
          + 3 * B * SK2 * SK2                               # dense_3, dense_4, Toeplitz smoothing conv
          + B * S * (4 * K) * (2 * n_sc))                   # dense_5
    rx = B * S * (2 * n_sc) * (2 * F) + B * (2 * S * F) * (2 * D)
    return 2.0 * (3 * eq + 2 * rx)


def bench_chains(a):
    """wall time per lock-step training step of G chains (three C calls: fused generator, fused step, monitor) against the same
    loop of ONE chain (receiver_mp.DeviceEpochLoop.step), same process; 73-frame batches, mixRayleigh"""
    import time
    from dl_ofdm_amd.equalizer_group import EqualizerChainGroup
    B = 73
    base = None
    for G in ([] if a.only_group else [1]) + [g for g in a.chains if g > 1 or a.only_group]:
        fl = [H.Flags(nbits=a.mods[i % len(a.mods)], nfilter=64, channel="mixRayleigh", device_data=True, seed=10 + i,
                      token="eqb%d" % i, save_dir="/tmp/dccn_eqbench/") for i in range(G)]
        rx = [glorot_init(rx_dims(F, ofdm.ofdm_tx(F)), 1 + i) for i, F in enumerate(fl)]
        grp = EqualizerChainGroup(fl, rx)
        act = grp.chains
        steps = act[0].steps
        for c in act:
            c.begin_epoch()
        if G == 1:
            fn = lambda i: act[0].loop.step()        # noqa: E731  (the solo loop of receiver_mp._train_on_device)
        else:
            fn = lambda i: grp.step(act, i % steps)  # noqa: E731
        for i in range(steps):                       # one whole epoch as warm-up (the loop's first step materialises its batch)
            fn(i)
        torch.cuda.synchronize()
        n = steps * max(1, a.steps // steps)          # whole epochs
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        sym = G * n * B * 7 / wall
        if G == 1:
            base = sym
        print(json.dumps(dict(path="epoch-loop", chains=G, mods=[F.nbits for F in fl], frames=B, steps=n,
                              ms_per_group_step=round(wall / n * 1e3, 4), host_issue_ms_per_group_step=round(host / n * 1e3, 4),
                              symbols_per_s=round(sym), vs_one_chain=round(sym / base, 3) if base else None,
                              tflops=round(G * step_flops(B) * n / wall / 1e12, 2))), flush=True)
        del grp, act
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, nargs="+", default=[73, 1170])
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--paths", nargs="+", default=["fused-graph", "fused-eager", "composed-autograd"])
    ap.add_argument("--ab", default="", help="KEY=V1,V2,..: time the eager fused step under each value of a dccn_set_tuning "
                                             "key, alternating inside this process (boxes differ; only this decides)")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--chains", type=int, nargs="*", default=[],
                    help="also time G chains per launch sequence (dccn_eq_train_step_grouped; modulations cycle through --mods): "
                         "the training loop of dl_ofdm_amd.equalizer_group (generator + step + monitor), per group step")
    ap.add_argument("--mods", type=int, nargs="+", default=[1, 2, 3, 4])
    ap.add_argument("--only-group", action="store_true", help="--chains: skip the one-chain reference loop (profiles)")
    a = ap.parse_args()
    if a.chains:
        return bench_chains(a)
    F = H.Flags(nbits=2, nfilter=64, channel="EPA")
    tx = ofdm.ofdm_tx(F)
    tr = EqualizerTrainer(F, tx, glorot_init(rx_dims(F, tx), 1), seed=1)
    for B in a.frames:
        x = torch.randn(B, 7, 80, 2, device="cuda")
        bits = torch.randint(0, 2, (B, tx.frame_size, 2), dtype=torch.int32, device="cuda")
        fl = step_flops(B)
        pl = tr._plan(B)
        pl.set_batch(x, bits)

        def composed():
            tr.grads.zero_()
            ce, mbuf, *_ = tr._forward(x, bits)
            ce.backward()
            tr._adam_step()

        if a.ab:
            from dl_ofdm_amd import _lib
            lib = _lib.load()
            key, vals = a.ab.split("=")
            key, vals = int(key), [int(v) for v in vals.split(",")]
            default = lib.dccn_get_tuning(key)
            times = {v: [] for v in vals}
            st = torch.cuda.current_stream().cuda_stream
            for r in range(a.rounds):
                for v in vals:
                    lib.dccn_set_tuning(key, v)
                    for _ in range(a.warmup):
                        pl.run(True, False)
                    torch.cuda.synchronize()
                    t = HipTimer()
                    t.start(st)
                    for _ in range(a.steps):
                        pl.run(True, False)
                    t.stop(st)
                    times[v].append(t.elapsed_ms() / a.steps)
            lib.dccn_set_tuning(key, default)
            print(json.dumps(dict(frames=B, key=key, median_ms={v: round(sorted(t)[len(t) // 2], 4) for v, t in times.items()},
                                  all_ms={v: [round(x, 4) for x in t] for v, t in times.items()})))
            continue
        for name, fn in (("fused-graph", lambda: pl.run(True, True)), ("fused-eager", lambda: pl.run(True, False)),
                         ("composed-autograd", composed)):
            if name not in a.paths:
                continue
            for _ in range(a.warmup):
                fn()
            torch.cuda.synchronize()
            t = HipTimer()
            st = torch.cuda.current_stream().cuda_stream
            t.start(st)
            for _ in range(a.steps):
                fn()
            t.stop(st)
            ms = t.elapsed_ms() / a.steps
            print(json.dumps(dict(path=name, frames=B, ms_per_step=round(ms, 4), symbols_per_s=round(B * 7 / ms * 1e3),
                                  gflop_per_step=round(fl / 1e9, 3), tflops=round(fl / ms / 1e9, 2))))


if __name__ == "__main__":
    main()
