cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_equalizer.py tests/test_gpu_ops.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -6 $O/pytest.log | cut -c1-200
for t in "19=1" "19=0" "19=1" "19=0"; do DCCN_TUNE="$t" timeout 200 python tools/eqbench.py --steps 200 --frames 73 --paths fused-eager 2>&1 | grep -v amdgpu | sed "s/^/$t /"; done
