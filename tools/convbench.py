"""General-k complex convolution (dev/py/complex.py:51-92 layers_conv1d_complex): implicit GEMM (dccn_cconv_patch_fwd, the
operand loader gathers the taps) against the im2col + GEMM route (dccn_cconv_im2col writes the k-inflated patch tensor,
dccn_cconv_gemm_fwd reads it back), same process, alternating rounds: the forward, and the backward (weight + input
gradient) as implicit GEMMs (dccn_cconv_patch_bwd_w / _bwd_x) against im2col + GEMM + GEMM + col2im.

    python tools/convbench.py [--k 5] [--stride 1] [--iters 100] [--rounds 5]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_ofdm_amd import complex as CX          # noqa: E402
from dl_ofdm_amd import ops                    # noqa: E402
from dl_ofdm_amd.engine import HipTimer        # noqa: E402

SHAPES = [  # (B, L, C, F)
    (1170, 560, 2, 64),       # one frame of time-domain samples per row, 64 filters
    (256, 1120, 16, 32),
    (64, 4096, 64, 64),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--shapes", default="0,1,2", help="indices into SHAPES")
    a = ap.parse_args()
    st_ = torch.cuda.current_stream().cuda_stream
    for B, L, C, F in [SHAPES[int(i)] for i in a.shapes.split(",")]:
        x = torch.randn(B, L, C, 2, device="cuda")
        store = CX.VariableStore(seed=1)
        with torch.no_grad():
            CX.layers_conv1d_complex(x, F, a.k, strides=a.stride, padding="same", scope=store)
        times = {True: [], False: []}
        for _ in range(a.rounds):
            for implicit in (True, False):
                CX.IMPLICIT_GEMM = implicit
                with torch.no_grad():
                    for _ in range(5):
                        store.begin()
                        CX.layers_conv1d_complex(x, F, a.k, strides=a.stride, padding="same", scope=store)
                    torch.cuda.synchronize()
                    t = HipTimer()
                    t.start(st_)
                    for _ in range(a.iters):
                        store.begin()
                        CX.layers_conv1d_complex(x, F, a.k, strides=a.stride, padding="same", scope=store)
                    t.stop(st_)
                times[implicit].append(t.elapsed_ms() / a.iters)
        CX.IMPLICIT_GEMM = True
        med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
        flop = 2.0 * B * ((L + a.stride - 1) // a.stride) * (2 * a.k * C) * (2 * F)
        # backward of the same layer: one forward graph, its backward replayed (torch.autograd.grad, graph retained)
        xg = x.clone().requires_grad_()
        store.begin()
        y = CX.layers_conv1d_complex(xg, F, a.k, strides=a.stride, padding="same", scope=store)
        g = torch.randn_like(y)
        leaves = (xg, store.tensor("conv2d/kernel"), store.tensor("conv2d/bias"))
        # the parts alone: graphs in which only the input / only the variables require a gradient
        kv, bv = store.tensor("conv2d/kernel"), store.tensor("conv2d/bias")
        kv.requires_grad_(False); bv.requires_grad_(False)
        store.begin()
        y_x = CX.layers_conv1d_complex(xg, F, a.k, strides=a.stride, padding="same", scope=store)
        kv.requires_grad_(True); bv.requires_grad_(True)
        store.begin()
        y_w = CX.layers_conv1d_complex(x, F, a.k, strides=a.stride, padding="same", scope=store)
        parts = {"": (y, leaves), "_dx": (y_x, leaves[:1]), "_dw": (y_w, leaves[1:])}
        bmed = {}
        for tag, (yy, lv) in parts.items():
            btimes = {True: [], False: []}
            for _ in range(a.rounds):
                for implicit in (True, False):
                    ops._PATCH_BWD_IM2COL = not implicit
                    ops._PATCH_BWD_DX_ALWAYS = implicit
                    for _ in range(5):
                        torch.autograd.grad(yy, lv, g, retain_graph=True)
                    torch.cuda.synchronize()
                    t = HipTimer()
                    t.start(st_)
                    for _ in range(a.iters):
                        torch.autograd.grad(yy, lv, g, retain_graph=True)
                    t.stop(st_)
                    btimes[implicit].append(t.elapsed_ms() / a.iters)
            ops._PATCH_BWD_IM2COL = ops._PATCH_BWD_DX_ALWAYS = False
            for k, v in btimes.items():
                bmed[("implicit" if k else "im2col") + tag] = sorted(v)[len(v) // 2]
        # what the library picks by itself (the input gradient goes implicit only where dccn_cconv_patch_bwd_supported expects a gain)
        for _ in range(5):
            torch.autograd.grad(y, leaves, g, retain_graph=True)
        t = HipTimer()
        t.start(st_)
        for _ in range(a.iters):
            torch.autograd.grad(y, leaves, g, retain_graph=True)
        t.stop(st_)
        bmed["default"] = t.elapsed_ms() / a.iters
        print(json.dumps(dict(shape=dict(B=B, L=L, C=C, F=F, k=a.k, stride=a.stride), implicit_ms=round(med[True], 4), im2col_ms=round(med[False], 4),
                              speedup=round(med[False] / med[True], 2), implicit_tflops=round(flop / med[True] / 1e9, 2),
                              patch_tensor_mb=round(B * L * a.k * C * 2 * 4 / 1e6, 1),
                              bwd_ms={k: round(v, 4) for k, v in bmed.items()},
                              bwd_speedup=round(bmed["im2col"] / bmed["default"], 2),
                              bwd_tflops=round(2 * flop / bmed["default"] / 1e9, 2))))


if __name__ == "__main__":
    main()
