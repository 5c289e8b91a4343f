#!/bin/bash
# round 4, run u2: block count of the overlapped Adam launch
O=gpurun_out/r04_u; mkdir -p $O
timeout 600 python tools/ab.py --config c4 --tunes "25=1;25=2;25=3;25=4" --what step_pipe --rounds 3 --iters 20 > $O/ab_c4b.txt 2>&1; cat $O/ab_c4b.txt
