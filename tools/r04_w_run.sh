#!/bin/bash
# round 4, run w: tail block reduction with its conditional stores in one exec-mask region
O=gpurun_out/r04_w; mkdir -p $O; rm -f $O/ab_red.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_equalizer.py -q -m gpu -x > $O/pytest_red.txt 2>&1; tail -2 $O/pytest_red.txt
for c in c2 c8 c1; do
  for lib in dl_ofdm_amd/lib/libdccn.so abl/libdccn_prev.so dl_ofdm_amd/lib/libdccn.so abl/libdccn_prev.so; do
    echo "== $c $lib" >> $O/ab_red.txt
    DCCN_LIB_PATH=$lib timeout 300 python tools/ab.py --config $c --what step_pipe --rounds 4 --iters 300 2>&1 | grep -v amdgpu.ids >> $O/ab_red.txt
  done
done
cat $O/ab_red.txt
