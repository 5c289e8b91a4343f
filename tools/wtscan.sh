# A/B of builds with different DCCN_WT_STORES masks (abl/libdccn_wt<mask>.so): in-situ timeline of the C2 step, alternating
mkdir -p gpurun_out/$1
for r in 1 2; do
for m in $2; do
  DCCN_LIB_PATH=abl/libdccn_wt$m.so timeout 200 python tools/steptl.py --config c2 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    r=json.loads(ln); print('wt$m', 'period_us', r.get('period_us'), [(l['us'], l.get('gap_before_us')) for l in r['launches']])" | tee -a gpurun_out/$1/wtscan.txt
done
done
