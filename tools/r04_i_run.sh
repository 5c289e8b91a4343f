set -x
O=gpurun_out/r04j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_benchmark.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -8 > $O/pytest_bm.txt
timeout 600 python tools/floor_test.py --nbits 2 --channel EVA --variants mix,chan,chan+align --out $O/floor 2>&1 | tail -3 > $O/floor.txt
cat $O/pytest_bm.txt $O/floor.txt
