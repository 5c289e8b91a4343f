#!/usr/bin/env python
"""Correctness + timing scan over the gemm16 kernel configurations (dccn_set_tuning keys 0-6).

    python tools/tune_scan.py [--quick] > gpurun_out/tune_scan.jsonl

Every configuration is first checked against an fp64 torch reference (or the 32x32x2 path) on the C2 shapes and on
ragged shapes, then timed: per operator with HIP events, and as the whole captured training step."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import bench
from dl_ofdm_amd import _lib
from dl_ofdm_amd.engine import HipTimer, RxDims, RxEngine, op_launchers

lib = _lib.load()
dev = torch.device("cuda", 0)
KEYS = dict(dense_fwd=0, dense_bwd=1, cconv_fwd=2, cconv_bwd_w=3, dense_bwd_splits=4, cconv_bwd_splits=5, smem_kb=6, whole_k=7,
            skinny=8)


def tune(**kw):
    for k, v in kw.items():
        assert lib.dccn_set_tuning(KEYS[k], int(v)) == 0, (k, v)


def stream():
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def out(**kw):
    print(json.dumps(kw), flush=True)


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def time_fn(fn, iters=100, warm=10):
    t = HipTimer()
    for _ in range(warm):
        fn()
    t.start(stream())
    for _ in range(iters):
        fn()
    t.stop(stream())
    return t.elapsed_ms() * 1e3 / iters


def make_engine(cfg="c2", frames=None, **kw):
    c = bench.CONFIGS[cfg]
    dims = RxDims(S=7, kin=c["nfft"] + c["cp"], F=c["F"], D=c["D"], nbits=c["nbits"])
    eng = RxEngine(dims, frames or c["frames"], device=dev, train=True, seed=1, **kw)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    eng.x.copy_(torch.randn(eng.x.shape, generator=g, device=dev))
    eng.bits.copy_(torch.randint(0, 2, eng.bits.shape, generator=g, device=dev, dtype=torch.int32))
    return eng


def snapshot_step(eng):
    """one eager training step from the initial state; returns everything the step produced"""
    eng.train_step()
    torch.cuda.synchronize()
    d = dict(prob=eng.prob.clone(), dz=eng.dz.clone(), dfft=eng.dfft.clone(), grads=eng.grads.clone(),
             params=eng.params.clone(), fft=eng.fft_out.clone(), m=eng.metrics())
    return d


def check_step(tag, frames, cfg="c2", **tk):
    """training step under a tuning vs the 32x32x2 path (all keys 0) on identical inputs"""
    tune(dense_fwd=0, dense_bwd=0, cconv_fwd=0, cconv_bwd_w=0, dense_bwd_splits=0, cconv_bwd_splits=0, smem_kb=0)
    ref = snapshot_step(make_engine(cfg, frames))
    tune(**tk)
    got = snapshot_step(make_engine(cfg, frames))
    errs = {k: relerr(got[k], ref[k]) for k in ("prob", "dz", "dfft", "fft", "grads")}
    # a pre-activation within rounding of a leaky-ReLU kink flips its derivative (1 <-> 0.2): a handful of cells may differ
    # by a finite amount in dz between two correct fp32 evaluations; count them instead of taking the max
    def outliers(k, tol=2e-5):
        d = (got[k].double() - ref[k].double()).abs()
        return int((d > tol * ref[k].double().abs().max()).sum())
    nbad = {"bad_" + k: outliers(k) for k in ("dz", "dfft")}
    dp = float((got["params"] - ref["params"]).abs().max())
    conf_ok = list(got["m"]["conf"]) == list(ref["m"]["conf"])
    ce = abs(got["m"]["ce_mean"] - ref["m"]["ce_mean"]) / abs(ref["m"]["ce_mean"])
    ok = all(errs[k] < 2e-5 for k in ("prob", "fft")) and errs["grads"] < 1e-3 and nbad["bad_dz"] <= 8 and ce < 1e-6
    errs.update(nbad)
    out(kind="check_step", tag=tag, frames=frames, cfg=cfg, tuning=tk, ok=bool(ok), conf_equal=bool(conf_ok), ce_rel=ce,
        dparam_max=dp, **errs)
    return ok


def check_gemm_refs():
    """operator entry points vs fp64 torch on C2 and ragged shapes"""
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    for (M, K, N) in ((1170, 896, 640), (37, 100, 36), (585, 64, 128), (130, 896, 640)):
        x, w, dy = rn(M, K), rn(K, N) * 0.05, rn(M, N)
        nws = lib.dccn_dense_bwd_w_workspace_size(M, K, N)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        dx_ref = (dy.double() @ w.double().T)
        dw_ref = (x.double().T @ dy.double())
        db_ref = dy.double().sum(0)
        for v in range(0, 7):
            for sp in ((0,) if v == 0 else (0, 1, 2, 3, 5)):
                tune(dense_bwd=v, dense_bwd_splits=sp)
                dx, dw, db = torch.zeros(M, K, device=dev), torch.zeros(K, N, device=dev), torch.zeros(N, device=dev)
                st = lib.dccn_dense_bwd(x.data_ptr(), dy.data_ptr(), w.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                        db.data_ptr(), M, K, N, ws.data_ptr(), nws, stream())
                torch.cuda.synchronize()
                e = (relerr(dx, dx_ref), relerr(dw, dw_ref), relerr(db, db_ref))
                out(kind="check_dense_bwd", shape=[M, K, N], variant=v, splits=sp, status=st, ok=bool(st == 0 and max(e) < 2e-5),
                    err_dx=e[0], err_dw=e[1], err_db=e[2])
    tune(dense_bwd_splits=0)
    for (rows, kin, F) in ((8190, 80, 64), (77, 64, 64), (8190, 64, 64), (259, 18, 6)):
        x, w, b = rn(rows, 2 * kin), rn(kin, 2 * F) * 0.1, rn(2 * F)
        wa, wb = w[:, :F].double(), w[:, F:].double()
        xi, xq = x[:, 0::2].double(), x[:, 1::2].double()
        re = xi @ wa - xq @ wb + (b[:F] - b[F:]).double()
        im = xi @ wb - xq @ wa + (b[F:] - b[:F]).double()
        ref = torch.stack([re, im], -1).reshape(rows, 2 * F)
        for v in range(0, 7):
            tune(cconv_fwd=v)
            o = torch.zeros(rows, 2 * F, device=dev)
            st = lib.dccn_cconv_gemm_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), rows, kin, F, stream())
            torch.cuda.synchronize()
            e = relerr(o, ref)
            out(kind="check_cconv_fwd", shape=[rows, kin, F], variant=v, status=st, ok=bool(st == 0 and e < 2e-5), err=e)


def time_ops_variants(eng, iters):
    res = {}
    for v in range(0, 9):
        tune(dense_fwd=v)
        ops = op_launchers(eng)
        if v == 0:
            res["dense_fwd+tail v0"] = time_fn(ops["dense_fwd"][0], iters) + time_fn(ops["tail_fwd_bwd"][0], iters)
        else:
            for kb in (0, 96):
                tune(smem_kb=kb)
                res["dense_tail v%d smem%d" % (v, kb)] = time_fn(ops["dense_tail_fwd_bwd"][0], iters)
            tune(smem_kb=0)
    for v in range(0, 7):
        for sp in ((0,) if v == 0 else (0, 2, 3, 4, 5, 6)):
            tune(dense_bwd=v, dense_bwd_splits=sp)
            ops = op_launchers(eng)
            res["dense_bwd v%d s%d" % (v, sp)] = time_fn(ops["dense_bwd_slabs"][0], iters)
    tune(dense_bwd_splits=0)
    for v in range(0, 7):
        tune(cconv_fwd=v)
        ops = op_launchers(eng)
        res["cconv_fwd v%d" % v] = time_fn(ops["cconv_fwd"][0], iters)
    for k, t in res.items():
        out(kind="time_op", op=k, us=round(t, 2))
    return res


def time_step(eng, tag, steps=300, **tk):
    tune(**tk)
    eng.close_graph()
    for _ in range(30):
        eng.train_step(graph=True)
    torch.cuda.synchronize()
    us = time_fn(lambda: eng.train_step(graph=True), steps, 20)
    out(kind="time_step", tag=tag, tuning=tk, us=round(us, 2))
    return us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--iters", type=int, default=100)
    args = ap.parse_args()
    torch.cuda.set_device(dev)
    out(kind="device", info=list(_lib.device_info()))
    # ---- correctness ----
    check_gemm_refs()
    allok = True
    for v in range(1, 9):
        for frames in (1170, 37):
            allok &= check_step("dense_fwd v%d" % v, frames, dense_fwd=v, dense_bwd=0, cconv_fwd=0, cconv_bwd_w=0)
    allok &= check_step("dense_fwd v3 bpsk", 100, cfg="c1" if "c1" in bench.CONFIGS else "c2", dense_fwd=3, dense_bwd=0,
                        cconv_fwd=0, cconv_bwd_w=0)
    for v in range(1, 5):
        for sp in (0, 16, 100):
            allok &= check_step("cconv_bwd_w v%d s%d" % (v, sp), 1170, dense_fwd=0, dense_bwd=0, cconv_fwd=0, cconv_bwd_w=v,
                                cconv_bwd_splits=sp)
    tune(cconv_bwd_splits=0)
    allok &= check_step("all16", 1170, dense_fwd=3, dense_bwd=1, cconv_fwd=1, cconv_bwd_w=1)
    allok &= check_step("all16 ragged", 53, dense_fwd=3, dense_bwd=1, cconv_fwd=1, cconv_bwd_w=1)
    out(kind="summary", all_step_checks_ok=bool(allok))
    # ---- timing ----
    # warm the clocks
    eng = make_engine("c2", want_z=False)
    tune(dense_fwd=3, dense_bwd=1, cconv_fwd=1, cconv_bwd_w=1)
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        for _ in range(50):
            eng.train_step(graph=True)
        torch.cuda.synchronize()
    time_ops_variants(eng, args.iters)
    base = dict(dense_fwd=0, dense_bwd=0, cconv_fwd=0, cconv_bwd_w=0, dense_bwd_splits=0, cconv_bwd_splits=0, smem_kb=0)
    eng0 = make_engine("c2", want_z=True)
    time_step(eng0, "legacy", **base)
    for v in range(1, 9):
        time_step(eng0, "dense_fwd v%d" % v, **dict(base, dense_fwd=v))
    for v in range(1, 7):
        for sp in (0, 3, 4, 5):
            time_step(eng0, "dense_bwd v%d s%d" % (v, sp), **dict(base, dense_bwd=v, dense_bwd_splits=sp))
    for v in range(1, 7):
        time_step(eng0, "cconv_fwd v%d" % v, **dict(base, cconv_fwd=v))
    for v in range(1, 5):
        for sp in (0, 32, 64, 128):
            time_step(eng0, "cconv_bwd_w v%d s%d" % (v, sp), **dict(base, cconv_bwd_w=v, cconv_bwd_splits=sp))
    for combo in (dict(dense_fwd=3, dense_bwd=1, cconv_fwd=1, cconv_bwd_w=1), dict(dense_fwd=2, dense_bwd=1, cconv_fwd=1, cconv_bwd_w=1),
                  dict(dense_fwd=3, dense_bwd=2, cconv_fwd=2, cconv_bwd_w=2), dict(dense_fwd=1, dense_bwd=3, cconv_fwd=3, cconv_bwd_w=4)):
        time_step(eng0, "combo", **dict(base, **combo))


if __name__ == "__main__":
    main()
