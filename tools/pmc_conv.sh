# PMC passes over tools/convbench.py --shapes 0 (never combined with sys / hip / hsa traces): counters of the few-channel input-gradient kernel
O=gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$i -o p -- python tools/convbench.py --shapes 0 --rounds 1 --iters 3 > $O/pmc_$i.log 2>&1
  f=$(find $O/pmc_$i -name "*counter_collection.csv" | head -1)
  echo "== $set" >> $O/pmc_conv.txt
  python - "$f" >> $O/pmc_conv.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k = r.get("Kernel_Name", "")
        if "narrow" in k or "kmajor_kernel<1, 5, true>" in k:
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("parse error", e)
for k, d in acc.items():
    print(k, {n: round(sum(v) / len(v), 1) for n, v in d.items()}, "launches", max(len(v) for v in d.values()))
PY
  rm -rf $O/pmc_$i
done
cat $O/pmc_conv.txt
