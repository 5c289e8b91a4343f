cd $GRAFT_REPO_ROOT
for i in 1 2; do
DCCN_LIB_PATH=$GRAFT_REPO_ROOT/abl/libdccn_head.so timeout 120 python tools/ab.py --what step_pipe,dense_tail_fwd_bwd --rounds 5 2>&1 | grep -v amdgpu | sed 's/^/head /'
timeout 120 python tools/ab.py --what step_pipe,dense_tail_fwd_bwd --rounds 5 2>&1 | grep -v amdgpu | sed 's/^/new  /'
done
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_graph_golden.py -x -q -m gpu 2>&1 | tail -3
