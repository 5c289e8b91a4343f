set -x
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_equalizer.py -x -q -m gpu -k "monitor or harness or chan_rms" 2>&1 | tail -8 > $O/pytest_eq.txt
timeout 600 python tools/eqloop.py --out $O/eqloop.jsonl > $O/eqloop.txt 2>&1
cat $O/pytest_eq.txt; cat $O/eqloop.txt | tail -8
