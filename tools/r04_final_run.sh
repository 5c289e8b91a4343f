#!/bin/bash
# round 4, final evidence: bench lines, kernel-trace summaries (C2 bench run, C3, C4, equaliser), loops, full -m gpu suite, smoke;
# "config5" as first argument: config 5 at full size as well
R=r04_final5
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$R; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $ROOT
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --no-cpu-baseline --no-kernel-times --no-other-configs --no-sweep --no-e2e > $O/kt.log 2>&1
python tools/profile_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
rm -rf $O/kt
for c in c3 c4; do
  rocprofv3 --kernel-trace --stats -d $O/${c}_kt -o kt -- python tools/opbench.py step_pipe --iters 20 --config $c > $O/${c}_kt.log 2>&1
  python tools/profile_summary.py $(find $O/${c}_kt -name "*.db" | head -1) 10 > $O/${c}_kernel_stats.txt 2>&1
  rm -rf $O/${c}_kt
done
python tools/eqbench.py --steps 100 2>&1 | grep -v amdgpu.ids > $O/eqbench.jsonl
python tools/eqloop.py 2>&1 | grep -v amdgpu.ids > $O/eqloop.jsonl
timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
if [ "$1" = "config5" ]; then python tools/config5_sweep.py --out $O/config5 --eq_epochs 0 > $O/config5_run.log 2>&1; tail -3 $O/config5_run.log; fi
tail -3 $O/pytest_all.txt; tail -2 $O/smoke.txt; head -8 $O/kernel_stats.txt | cut -c1-150
