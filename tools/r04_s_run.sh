#!/bin/bash
# round 4, run s2: cross-barrier fragment prefetch only for long k ranges (>= 48 k-tiles)
O=gpurun_out/r04_s; mkdir -p $O; rm -f $O/ab2.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_configs.py -q -m gpu -x > $O/pytest2.txt 2>&1; tail -3 $O/pytest2.txt
for c in c2 c3 c4; do
  it=300; [ $c = c4 ] && it=20
  for lib in dl_ofdm_amd/lib/libdccn.so abl/libdccn_noxb.so; do
    echo "== $c $lib" >> $O/ab2.txt
    DCCN_LIB_PATH=$lib timeout 300 python tools/ab.py --config $c --what step_pipe --rounds 4 --iters $it 2>&1 | grep -v amdgpu.ids >> $O/ab2.txt
  done
done
cat $O/ab2.txt
