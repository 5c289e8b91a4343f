set -x
O=gpurun_out/r04m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_equalizer.py tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_layer_api.py tests/test_gpu_harness.py tests/test_gpu_session.py tests/test_gpu_graph_golden.py -x -q -m gpu 2>&1 | tail -8 > $O/pytest_eq.txt
timeout 300 python tools/eqbench.py --frames 73 --ab 21=0,1 --rounds 5 > $O/eqbench_ab.txt 2>&1
timeout 300 python tools/eqbench.py --frames 73 1170 --paths fused-graph fused-eager > $O/eqbench.jsonl 2>$O/eqbench.err
timeout 300 python tools/eqloop.py --out $O/eqloop.jsonl > $O/eqloop.txt 2>&1
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_eq -- python $GRAFT_REPO_ROOT/tools/eqbench.py --frames 73 --paths fused-eager --steps 100 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_eq -name "*_results.db" | head -1); python tools/profile_summary.py $DB > $O/eq73_kernel_stats.txt 2>&1
cat $O/pytest_eq.txt $O/eqbench_ab.txt; tail -3 $O/eqbench.jsonl; tail -7 $O/eqloop.txt; head -30 $O/eq73_kernel_stats.txt
