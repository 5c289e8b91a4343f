#!/usr/bin/env python
"""the few numbers of a bench.py line one looks at first (tools/gpu_run.sh prints them into the gpurun tail)"""
import json
import sys


def main():
    ln = [l for l in open(sys.argv[1]) if l.startswith("{")]
    if not ln:
        print("no JSON line in", sys.argv[1])
        return
    r = json.loads(ln[-1])
    st = r.get("step", {})
    print("ms_per_step %.5f  mfma_frac %.4f  value %.4g  regions %s" % (r["ms_per_step"], st.get("mfma_frac", 0), r["value"],
                                                                      st.get("regions_ms")))
    bd = st.get("boundaries")
    if bd:
        print("  launches:", [(l.get("name", l.get("slot")), l.get("us"), l.get("gap_before_us")) for l in bd.get("launches", [])],
              "untraced", bd.get("untraced_ms_per_step_200"))
    rf = r.get("roofline")
    if rf:
        print("  roofline: %s %.1f us frac %.3f traffic %s" % (rf.get("op"), rf.get("avg_launch_us", 0), rf.get("frac", 0), rf.get("traffic")))
    for k, c in (r.get("configs") or {}).items():
        print("  %s: ms %.4f mfma %.4f" % (k, c.get("ms_per_step", 0), c.get("mfma_frac", 0)))
    if "e2e" in r:
        print("  e2e:", {k: r["e2e"].get(k) for k in ("ms_per_step", "host_issue_ms_per_step", "symbols_per_s")})
    if "sweep" in r:
        print("  sweep: %.3f s, %s points/rank" % (r["sweep"]["seconds"], r["sweep"]["points_per_rank"]))
    cb = r.get("cpu_baseline")
    if cb:
        print("  cpu_baseline:", {k: cb.get(k) for k in ("value", "unit", "cores", "kind")})


if __name__ == "__main__":
    main()
