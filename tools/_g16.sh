cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_benchmark.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 $O/pytest.log | cut -c1-220
