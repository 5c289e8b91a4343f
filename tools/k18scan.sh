mkdir -p gpurun_out/r06aa
for r in 1 2; do
for cfg in "5:" "5:18=1" "13:18=1" "13:18=2" "13:"; do
  m=${cfg%%:*}; t=${cfg#*:}
  DCCN_LIB_PATH=abl/libdccn_wt$m.so timeout 200 python tools/steptl.py --config c2 --tunes "$t" 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    r=json.loads(ln); print('wt$m', '$t', 'period_us', r.get('period_us'), [(l['us'], l.get('gap_before_us')) for l in r['launches']])" | tee -a gpurun_out/r06aa/k18.txt
done
done
