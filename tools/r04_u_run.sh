#!/bin/bash
# round 4, run u6: the optimizer stream at the lowest priority (its own hardware queue)
O=gpurun_out/r04_u; mkdir -p $O
timeout 600 python tools/ab.py --config c4 --tunes "25=0;25=1;25=2" --what step_pipe --rounds 3 --iters 20 > $O/ab_c4f.txt 2>&1; cat $O/ab_c4f.txt
timeout 600 python tools/ab.py --config c4 --tunes "25=1;25=0" --what step_pipe --rounds 3 --iters 20 > $O/ab_c4g.txt 2>&1; cat $O/ab_c4g.txt
python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-sweep --no-e2e 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench c4', j['ms_per_step'], j['step'].get('mfma_frac'))"
