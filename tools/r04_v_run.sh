#!/bin/bash
# round 4, run v: config 5 at full size with 4 (and 2) gloo ranks sharing the one GPU
O=gpurun_out/r04_v; mkdir -p $O
for W in 4 2; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((29520+W)) tools/config5_sweep.py --backend gloo --out $O/config5_w$W --eq_epochs 0 > $O/config5_w$W.log 2>&1
  tail -2 $O/config5_w$W.log
done
ls $O/*
