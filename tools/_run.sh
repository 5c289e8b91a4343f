cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_graph_golden.py -m gpu -q 2>&1 | tail -25
