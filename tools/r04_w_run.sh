#!/bin/bash
# round 4, run w2: few-row kernels with the wave index as a scalar (wave-uniform branches instead of exec-mask regions)
O=gpurun_out/r04_w; mkdir -p $O; rm -f $O/eq_scalar.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_equalizer.py -q -m gpu -x > $O/pytest_sc.txt 2>&1; tail -2 $O/pytest_sc.txt
for lib in dl_ofdm_amd/lib/libdccn.so abl/libdccn_prev.so dl_ofdm_amd/lib/libdccn.so abl/libdccn_prev.so; do
  echo "== $lib" >> $O/eq_scalar.txt
  DCCN_LIB_PATH=$lib timeout 300 python tools/eqbench.py --frames 73 --steps 300 --paths fused-eager 2>&1 | grep -v amdgpu.ids | cut -c1-90 >> $O/eq_scalar.txt
done
cat $O/eq_scalar.txt
