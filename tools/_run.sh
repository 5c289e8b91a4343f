cd $GRAFT_REPO_ROOT
echo -n "default: "; python tools/opbench.py dense_tail_fwd_bwd tail_fwd_bwd step --iters 200 2>&1 | grep -v amdgpu | tr '\n' ' '; echo
echo -n "c3 default: "; python tools/opbench.py tail_fwd_bwd step --iters 100 --config c3 2>&1 | grep -v amdgpu | tr '\n' ' '; echo
echo -n "noslp: "; DCCN_LIB_PATH=$GRAFT_REPO_ROOT/dl_ofdm_amd/lib/libdccn_noslp.so python tools/opbench.py dense_tail_fwd_bwd tail_fwd_bwd step --iters 200 2>&1 | grep -v amdgpu | tr '\n' ' '; echo
echo -n "c3 noslp: "; DCCN_LIB_PATH=$GRAFT_REPO_ROOT/dl_ofdm_amd/lib/libdccn_noslp.so python tools/opbench.py tail_fwd_bwd step --iters 100 --config c3 2>&1 | grep -v amdgpu | tr '\n' ' '; echo
