O=gpurun_out/g20; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_equalizer.py -x -q 2>&1 | tail -15 > $O/tests.log
cat $O/tests.log
python tools/eqbench.py --steps 100 --paths fused-eager > $O/eqbench.jsonl 2>&1
cat $O/eqbench.jsonl
rocprofv3 --kernel-trace --stats -d $O/eq73_kt -o kt -- python tools/eqbench.py --frames 73 --steps 100 --paths fused-eager > $O/eq73_kt.log 2>&1
python tools/profile_summary.py $(find $O/eq73_kt -name "*.db" | head -1) > $O/eq73_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/eq_kt -o kt -- python tools/eqbench.py --frames 1170 --steps 50 --paths fused-eager > $O/eq_kt.log 2>&1
python tools/profile_summary.py $(find $O/eq_kt -name "*.db" | head -1) > $O/eq_kernel_stats.txt 2>&1
rm -rf $O/eq73_kt $O/eq_kt
