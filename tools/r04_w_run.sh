#!/bin/bash
# round 4, run w: large C-Conv weight gradient on 128x128x32 tiles over 7 k ranges (knob 26)
O=gpurun_out/r04_w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_configs.py tests/test_gpu_ops.py -q -m gpu -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python tools/variant_check.py --config c4 --a 26=0 --b 26=1 --frames 585 --steps 2 > $O/vc.txt 2>&1; tail -1 $O/vc.txt
timeout 600 python tools/ab.py --config c4 --tunes "26=0;26=1;26=0,25=0;26=1,25=0" --what step_pipe --rounds 3 --iters 20 > $O/ab_c4.txt 2>&1; cat $O/ab_c4.txt
