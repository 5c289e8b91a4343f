O=$GRAFT_REPO_ROOT/gpurun_out/r04o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_e2e -o kt -- python $GRAFT_REPO_ROOT/tools/e2ebench.py --host-steps 0 --steps 60 > $O/e2e.log 2>&1
F=$(find /tmp/kt_e2e -name "*kernel_trace.csv" | head -1)
python - "$F" > $O/timeline.txt <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 120 dispatches: name, queue, start, end relative
tail=rows[-140:]
t0=int(tail[0]['Start_Timestamp'])
for r in tail:
    n=r['Kernel_Name'][:48]
    print("%-48s q=%s  start %8.2f  end %8.2f  dur %6.2f" % (n, r.get('Queue_Id','?'), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
tail -60 $O/timeline.txt
