#!/bin/bash
# round 4: last check of the committed tree (full -m gpu suite + smoke)
O=gpurun_out/r04_last; mkdir -p $O
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_bench.py -q -m gpu 2>&1 | tail -2; done
