#!/bin/bash
# round 4, run r6: per-kernel times of the C4 step with the dense forward as k ranges (knob 13 = 5)
O=gpurun_out/r04_r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
DCCN_TUNE=13=5 rocprofv3 --kernel-trace --stats -d $O/c4s_kt -o kt -- python tools/opbench.py step_pipe --iters 20 --config c4 > $O/c4s_kt.log 2>&1
python tools/profile_summary.py $(find $O/c4s_kt -name "*.db" | head -1) 10 > $O/c4s_kernel_stats.txt 2>&1
cat $O/c4s_kernel_stats.txt
rm -rf $O/c4s_kt
