"""Instruction histogram of one kernel in a device-only assembly dump.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S dl_ofdm_amd/csrc/dccn_abi.hip -o /tmp/dccn.s      (or dccn_abi_eq / _gen / _conv.hip: one unit at a time)
    python tools/isa_hist.py /tmp/dccn.s <symbol prefix> [--loops]
"""
import collections
import sys


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    src = open(path).read().split("\n")
    start = next(i for i, l in enumerate(src) if l.startswith(prefix) and ":" in l.split(";")[0])
    end = next(i for i in range(start, len(src)) if src[i].strip().startswith("s_endpgm"))
    c = collections.Counter()
    per_block = []
    label, bc = "entry", collections.Counter()
    for l in src[start + 1:end + 1]:
        s = l.split(";")[0].strip()
        if not s or s.startswith("."):
            if s.startswith(".LBB") and s.endswith(":"):
                per_block.append((label, bc))
                label, bc = s[:-1], collections.Counter()
            continue
        if s.endswith(":"):
            per_block.append((label, bc))
            label, bc = s[:-1], collections.Counter()
            continue
        op = s.split()[0]
        c[op] += 1
        bc[op] += 1
    per_block.append((label, bc))
    print("total", sum(c.values()))
    for k, v in c.most_common(40):
        print("  %-28s %d" % (k, v))
    if "--loops" in sys.argv:
        for lab, b in per_block:
            n = sum(b.values())
            if n >= 40:
                mf = sum(v for k, v in b.items() if "mfma" in k)
                va = sum(v for k, v in b.items() if k.startswith("v_") and "mfma" not in k)
                pk = sum(v for k, v in b.items() if k.startswith("v_pk"))
                print("%-12s n=%5d mfma=%4d valu=%5d (pk %d) ds=%3d vmem=%3d salu=%4d" % (
                    lab, n, mf, va, pk, sum(v for k, v in b.items() if k.startswith("ds_")),
                    sum(v for k, v in b.items() if k.startswith(("global_", "buffer_"))),
                    sum(v for k, v in b.items() if k.startswith("s_"))))
    for i in range(end, min(end + 400, len(src))):
        if any(t in src[i] for t in ("; NumVgprs", "; NumAgprs", "; Occupancy", "; ScratchSize", "; LDSByteSize")):
            print(src[i].strip())
        if ".end_amdhsa_kernel" in src[i]:
            break


main()
