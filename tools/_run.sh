cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests/test_gpu_configs.py -m gpu -q -x --durations=5 2>&1 | tail -30
