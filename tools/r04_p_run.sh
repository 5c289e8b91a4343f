set -x
O=gpurun_out/r04p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_equalizer.py tests/test_gpu_harness.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -6 > $O/pytest.txt
timeout 300 python tools/eqloop.py --out $O/eqloop.jsonl > $O/eqloop.txt 2>&1
cat $O/pytest.txt; grep -v amdgpu $O/eqloop.txt
