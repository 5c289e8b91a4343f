cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_now.json; cut -c1-250 gpurun_out/bench_now.json
