#!/bin/bash
# round 4, run v2: config 5 at full size, 4 gloo ranks sharing the one GPU, the 8-QAM chain on a high-priority stream
O=gpurun_out/r04_v; mkdir -p $O
W=4
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 29531 tools/config5_sweep.py --backend gloo --out $O/config5_w4p --eq_epochs 0 --priority_nbits 3 > $O/config5_w4p.log 2>&1
tail -2 $O/config5_w4p.log; grep "rank" $O/config5_w4p.log | tail -4
