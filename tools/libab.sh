# in-situ timeline of a BASELINE config under two builds of the library, alternating:  bash tools/libab.sh <tag> <config> <libA> <libB>
mkdir -p gpurun_out/$1
for r in 1 2; do
for lib in $3 $4; do
  DCCN_LIB_PATH=$lib timeout 200 python tools/steptl.py --config $2 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    r=json.loads(ln); print('$lib', 'period_us', r.get('period_us'), [(l.get('name'), l['us'], l.get('gap_before_us')) for l in r['launches']])" | tee -a gpurun_out/$1/libab_$2.txt
done
done
