#!/bin/bash
O=gpurun_out/r04_t; mkdir -p $O
timeout 300 python tools/eqbench.py --frames 73 --steps 200 --ab 24=0,1,2,3 2>&1 | grep -v amdgpu.ids > $O/eqbench_riders2.jsonl; cat $O/eqbench_riders2.jsonl | cut -c1-120
