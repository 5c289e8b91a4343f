#!/bin/bash
# Config 5's four training chains as four gloo ranks SHARING one GPU: equaliser seconds per chain of a short run, by the number
# of hardware queues each process may open (GPU_MAX_HW_QUEUES; "auto" = what dl_ofdm_amd/config5.py shared_gpu_env picks),
# next to the same chains one after the other in one process.      gpurun -- 'bash tools/c5share.sh [tag]'
O=gpurun_out/${1:-c5share}; mkdir -p $O
ARGS="--eq_epochs 80 --rx_epoch_scale 0.05 --frames 1000 --classical_frames 50"
run() {  # tag, env assignments...
  tag=$1; shift
  env "$@" timeout 400 python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 \
      tools/config5_sweep.py --backend gloo --out $O/$tag $ARGS > $O/$tag.log 2>&1
  python - <<PY
import json
d = json.load(open("$O/$tag/config5_timing.json"))
print("%-8s queues/process %-16s equaliser s per chain %s  total %.1f s" % ("$tag", d.get("hw_queues_per_process"),
      [round(sum(v for k, v in r.items() if k.startswith("train_eq")), 1) for r in d["per_rank_seconds"]], d["per_rank_seconds"][0]["total"]))
PY
}
run auto C5SHARE=1
for q in 1 2 4 8; do run q$q GPU_MAX_HW_QUEUES=$q; done
timeout 300 python tools/config5_sweep.py --out $O/serial $ARGS > $O/serial.log 2>&1
python - <<PY
import json
d = json.load(open("$O/serial/config5_timing.json"))
print("serial  ", {k: round(v, 1) for k, v in d["per_rank_seconds"][0].items() if k.startswith("train_eq") or k == "total"})
PY
