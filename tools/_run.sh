cd $GRAFT_REPO_ROOT
for T in "7=0" "7=1"; do
echo -n "tune $T: "; python tools/opbench.py cconv_fwd cconv_bwd_w step --iters 200 --tune $T 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_equalizer.py tests/test_layer_api.py -m gpu -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline | cut -c1-300
