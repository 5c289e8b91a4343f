set -x
O=gpurun_out/r04n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_datagen.py tests/test_gpu_harness.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -6 > $O/pytest.txt
timeout 300 python tools/e2ebench.py --host-steps 0 > $O/e2ebench.jsonl 2>&1
timeout 300 python tools/e2ebench.py --host-steps 0 --channel AWGN >> $O/e2ebench.jsonl 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2>$O/bench.err
cat $O/pytest.txt; grep -v amdgpu $O/e2ebench.jsonl; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['e2e'])"
