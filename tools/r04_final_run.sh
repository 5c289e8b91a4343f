#!/bin/bash
# round 4, final evidence: bench lines, the kernel-trace summaries of what changed in the second half of the round (C3, C4, equaliser),
# loops, full -m gpu suite, smoke; "config5" as first argument: config 5 at full size as well
R=r04_final4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$R; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $ROOT
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
for c in c3 c4; do
  rocprofv3 --kernel-trace --stats -d $O/${c}_kt -o kt -- python tools/opbench.py step_pipe --iters 20 --config $c > $O/${c}_kt.log 2>&1
  python tools/profile_summary.py $(find $O/${c}_kt -name "*.db" | head -1) 10 > $O/${c}_kernel_stats.txt 2>&1
  rm -rf $O/${c}_kt
done
python tools/ab.py --config c4 --tunes "25=0;25=1;25=2" --what step_pipe --rounds 3 --iters 20 2>&1 | grep -v amdgpu.ids > $O/c4_overlap_ab.txt
python tools/eqbench.py --steps 100 2>&1 | grep -v amdgpu.ids > $O/eqbench.jsonl
rocprofv3 --kernel-trace --stats -d $O/eq73_kt -o kt -- python tools/eqbench.py --frames 73 --steps 100 --paths fused-eager > $O/eq73_kt.log 2>&1
python tools/profile_summary.py $(find $O/eq73_kt -name "*.db" | head -1) > $O/eq73_kernel_stats.txt 2>&1
rm -rf $O/eq73_kt
python tools/eqloop.py 2>&1 | grep -v amdgpu.ids > $O/eqloop.jsonl
timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
if [ "$1" = "config5" ]; then python tools/config5_sweep.py --out $O/config5 --eq_epochs 0 > $O/config5_run.log 2>&1; tail -3 $O/config5_run.log; fi
tail -3 $O/pytest_all.txt; tail -2 $O/smoke.txt; cat $O/c4_overlap_ab.txt $O/c4_kernel_stats.txt | cut -c1-150
