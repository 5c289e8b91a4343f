#!/bin/bash
O=gpurun_out/r04_w; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_engine.py -q -m gpu -x -k "overlap" > $O/pytest_ov.txt 2>&1; tail -5 $O/pytest_ov.txt
