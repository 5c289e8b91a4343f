#!/usr/bin/env python
"""Summarise rocprofv3 --pmc csv output (one directory per counter pass) into the text table kept under
profiles/ and the per-kernel HBM traffic json read by bench.py.

    python tools/pmc_summary.py gpurun_out/r01 profiles/r01_pmc_counters.txt profiles/pmc_traffic.json
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

OP_OF_KERNEL = [  # kernel-name substring -> bench.py operator key (C2 shapes)
    ("rx_bwd_fused_kernel", "rx_backward"),
    ("adam_rx_kernel", "adam_rx"),
    ("dense_bwd_grouped_km_kernel", "dense_bwd_slabs"),
    ("dense_bwd_grouped_kernel", "dense_bwd_slabs"),
    ("gemm16_kernel<1, 0, 1, 4, 3, 1, 64", "dense_tail_fwd_bwd"),
    ("cconv_bwd_w_km_finalize_kernel", "cconv_bwd_w"),
    ("gemm_f32_mfma_kernel<1, 0,", "dense_fwd"),
    ("gemm_f32_mfma_kernel<1, 2,", "cconv_fwd"),        # any tile configuration of the C-Conv forward
    ("cconv_fwd_staged_kernel", "cconv_fwd"),           # round 5: the staged whole-k tile (cconv_fwd.h)
    ("gemm_f32_mfma_kernel<0, 0,", "cconv_bwd_w"),
    ("cconv_bwd_w_finalize_kernel", "cconv_bwd_w"),
]


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("void ", "").replace("dccn::", "")
    return name if len(name) <= 62 else name[:59] + "..."


def main(root, out_txt, out_json):
    lines, traffic = [], defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = defaultdict(lambda: defaultdict(list))
        dur = defaultdict(dict)
        order = []
        for row in csv.DictReader(open(files[0])):
            k = short(row["Kernel_Name"])
            if k not in order:
                order.append(k)
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur[k][row["Dispatch_Id"]] = (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3
        names = sorted({c for k in acc for c in acc[k]})
        lines.append("== rocprofv3 --pmc pass: %s  (tools/opbench.py step_pipe --bench-plan: the launches bench.py times, C2 shapes, average per dispatch)" % " ".join(names))
        for k in order:
            if len(dur[k]) < 5 or k.startswith("at::") or "rocclr" in k:
                continue
            avg = {c: sum(v) / len(v) for c, v in acc[k].items()}
            lines.append("  %-62s %s  duration_us=%.4g" % (k, "  ".join("%s=%.5g" % (c, avg[c]) for c in names if c in avg),
                                                          sum(dur[k].values()) / len(dur[k])))
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                if c in avg:
                    traffic[k][c] = avg[c]
    open(out_txt, "w").write("\n".join(lines) + "\n")
    res = {"c2": {}, "detail": {}, "plan": "tools/opbench.py step_pipe --bench-plan (the launches bench.py times)", "note": "bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB units); "
                                           "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)"}
    for k, v in traffic.items():
        for sub, op in OP_OF_KERNEL:
            if sub in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                fb, wb = v["FETCH_SIZE"] * 1024.0 * 2.0, v["WRITE_SIZE"] * 1024.0
                res["c2"][op] = fb + wb
                res["detail"][op] = dict(kernel=k, fetch_bytes=fb, write_bytes=wb, hbm_bytes=fb + wb)
    # which library these counters describe (include/dccn.h dccn_build_id): bench.py reports roofline.traffic only for that build
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from dl_ofdm_amd import _lib
        res["build_id"] = _lib.load().dccn_build_id().decode()
    except Exception as e:                                   # noqa: BLE001
        res["build_id"] = None
        res["build_id_error"] = repr(e)
    json.dump(res, open(out_json, "w"), indent=1)
    print("\n".join(lines[:6]))
    print(json.dumps(res["c2"]))


if __name__ == "__main__":
    main(*sys.argv[1:4])
