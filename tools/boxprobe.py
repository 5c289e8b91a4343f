"""What kind of box is this?  Device-to-device copy bandwidth (HBM-bound), a long fp32 GEMM (clock / power bound) and the in-situ
launch times of the C2 step, one JSON line -- to tell a slow box of the pool from a slow build (0.101 vs 0.079 ms)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ev_time(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")          # 1 GiB
    b = torch.empty_like(a)
    t = ev_time(lambda: b.copy_(a), 20)
    copy_tbs = 2 * a.numel() * 4 / t / 1e12
    s = torch.empty(1 << 22, dtype=torch.float32, device="cuda")          # 16 MiB: lives in the Infinity Cache
    s2 = torch.empty_like(s)
    t = ev_time(lambda: s2.copy_(s), 200)
    small_tbs = 2 * s.numel() * 4 / t / 1e12
    x = torch.randn(8192, 8192, device="cuda")
    y = torch.randn(8192, 8192, device="cuda")
    t = ev_time(lambda: torch.mm(x, y), 10)
    gemm_tflops = 2 * 8192 ** 3 / t / 1e12
    out = dict(copy_1GiB_TBps=round(copy_tbs, 3), copy_16MiB_TBps=round(small_tbs, 3), torch_mm_fp32_tflops=round(gemm_tflops, 1),
               pci=torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else None)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
