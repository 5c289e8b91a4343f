#!/usr/bin/env python
"""Time dccn_dense_fwd / _bwd_x / _bwd_w over a K sweep: time = fixed + per-k-tile * ntiles."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C

import torch

from dl_ofdm_amd import _lib
from dl_ofdm_amd.engine import HipTimer

lib = _lib.load()
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
t = HipTimer()
M, N = int(os.environ.get("M", 1170)), int(os.environ.get("N", 640))
for K in (64, 128, 256, 512, 896, 1792, 3584):
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(K, N, device="cuda")
    b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    dx = torch.empty(M, K, device="cuda")
    f = lambda: lib.dccn_dense_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, K, N, s())
    g = lambda: lib.dccn_dense_bwd_x(y.data_ptr(), w.data_ptr(), dx.data_ptr(), M, K, N, s())
    out = []
    for fn in (f, g):
        for _ in range(5):
            fn()
        t.start(s())
        for _ in range(50):
            fn()
        t.stop(s())
        out.append(t.elapsed_ms() * 1e3 / 50)
    fl = 2.0 * M * N * K
    print("K=%5d  dense_fwd %7.2f us (%5.1f TF)   dense_bwd_x(K'=%d) %7.2f us (%5.1f TF)"
          % (K, out[0], fl / out[0] / 1e6, N, out[1], fl / out[1] / 1e6))
