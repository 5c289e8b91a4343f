#!/bin/bash
# round 4: last check of the committed tree (full -m gpu suite + smoke + the default bench line)
O=gpurun_out/r04_last; mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/pytest_all.txt; cat $O/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time python bench.py ) > $O/bench_n1.json 2> $O/bench_n1.err; tail -4 $O/bench_n1.err; python -c "
import json; j=json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['step']['mfma_frac'], j['configs']['c3']['mfma_frac'], j['configs']['c4']['mfma_frac'], j['e2e']['ms_per_step'])"
