O=gpurun_out/g19; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_equalizer.py tests/test_gpu_datagen.py tests/test_gpu_harness.py tests/test_gpu_configs.py -x -q 2>&1 | tail -8 > $O/tests.log
python tools/eqbench.py --steps 100 --paths fused-eager > $O/eqbench.jsonl 2>&1
python tools/e2ebench.py --host-steps 0 > $O/e2e.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/eq73_kt -o kt -- python tools/eqbench.py --frames 73 --steps 100 --paths fused-eager > $O/eq73_kt.log 2>&1
python tools/profile_summary.py $(find $O/eq73_kt -name "*.db" | head -1) > $O/eq73_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/e2e_kt -o kt -- python tools/e2ebench.py --host-steps 0 > $O/e2e_kt.log 2>&1
python tools/profile_summary.py $(find $O/e2e_kt -name "*.db" | head -1) > $O/e2e_kernel_stats.txt 2>&1
rm -rf $O/eq73_kt $O/e2e_kt
cat $O/tests.log $O/eqbench.jsonl $O/e2e.log
