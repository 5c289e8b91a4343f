set -x
O=gpurun_out/r04_final2; mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python tools/eqloop.py 2>&1 | grep -v amdgpu.ids > $O/eqloop.jsonl
python tools/e2ebench.py --host-steps 0 2>&1 | grep -v amdgpu.ids > $O/e2ebench.jsonl
timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
if [ "$1" = "config5" ]; then python tools/config5_sweep.py --out $O/config5 --eq_epochs 0 > $O/config5_run.log 2>&1; tail -3 $O/config5_run.log; fi
tail -3 $O/pytest_all.txt; tail -2 $O/smoke.txt; cat $O/eqloop.jsonl $O/e2ebench.jsonl
