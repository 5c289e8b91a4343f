#!/usr/bin/env python
"""Fixed cost vs per-k-tile cost of the fused dense+tail launch: time it at K = 64, 128, ... on the C2 buffers.

    python tools/ksweep.py [--iters 200] [--tune 0=2]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from dl_ofdm_amd import _lib
    from dl_ofdm_amd.engine import HipTimer, RxDims, RxEngine
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--tune", default="")
    ap.add_argument("--ks", default="64,128,256,384,512,640,768,896")
    args = ap.parse_args()
    lib = _lib.load()
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        assert lib.dccn_set_tuning(int(k), int(v)) == 0
    c = bench.CONFIGS[args.config]
    dims = RxDims(S=7, kin=c["nfft"] + c["cp"], F=c["F"], D=c["D"], nbits=c["nbits"])
    eng = RxEngine(dims, c["frames"], train=True)
    eng.x.normal_()
    eng.bits.random_(0, 2)
    eng.train_step()
    torch.cuda.synchronize()
    P, G = eng.params, eng.grads
    off = eng.layout

    def seg(buf, name):
        return buf.data_ptr() + 4 * off[name][0]
    B, dN = c["frames"], 2 * c["D"]
    nws = lib.dccn_dense_tail_workspace_size(B, dN, dims.nbits)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    t = HipTimer()
    s = eng._stream
    for K in [int(k) for k in args.ks.split(",")]:
        def fn():
            return lib.dccn_dense_tail_fwd_bwd(
                eng.fft_out.data_ptr(), seg(P, "demodulation/dense/kernel"), seg(P, "demodulation/dense/bias"), None,
                eng.bits.data_ptr(), seg(P, "demodulation/conv2d/kernel"), None, eng.metrics_buf.data_ptr(),
                eng.dz.data_ptr(), seg(G, "demodulation/conv2d/kernel"), B, K, dN, dims.nbits, ws.data_ptr(), nws, s())
        for _ in range(10):
            assert fn() == 0
        t.start(s())
        for _ in range(args.iters):
            fn()
        t.stop(s())
        print("dense_tail K=%4d: %.2f us/launch (2 kernels: gemm16+finalize)" % (K, t.elapsed_ms() * 1e3 / args.iters), flush=True)


if __name__ == "__main__":
    main()
