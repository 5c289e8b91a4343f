#!/bin/bash
# round 4, run t: the equaliser step normalises the next batch on its optimizer launch
O=gpurun_out/r04_t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_equalizer.py -q -m gpu -x > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 300 python tools/eqloop.py 2>&1 | grep -v amdgpu.ids > $O/eqloop.jsonl; cat $O/eqloop.jsonl
