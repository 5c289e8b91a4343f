#!/usr/bin/env python
"""Summarise a rocprofv3 sqlite result (kernel-trace) into a short text table:
   python tools/profile_summary.py gpurun_out/r02/kt/kt_results.db [min_calls] > profiles/r02_kernel_stats.txt"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("dccn::", "")
    if len(name) > 90:
        name = name[:87] + "..."
    return name


def main(path, min_calls=50):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("%-92s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in rows:
        if calls < min_calls:
            continue
        print("%-92s %8d %12.1f %10.3f %6.2f%%" % (short(name), calls, tot, avg, pct))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 50)
