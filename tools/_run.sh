cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k classical 2>&1 | tail -12
