#!/bin/bash
# round 4, run t: equaliser experiments
O=gpurun_out/r04_t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_equalizer.py -q -m gpu -x > $O/pytest5.txt 2>&1; tail -3 $O/pytest5.txt
timeout 300 python tools/eqbench.py --frames 73 --steps 200 --ab 24=0,1,4 2>&1 | grep -v amdgpu.ids > $O/eqbench_riders3.jsonl; cat $O/eqbench_riders3.jsonl | cut -c1-120
