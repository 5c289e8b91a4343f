#!/bin/bash
# round 4, run q: 16-QAM tail with mask selects + operand prefetch
O=gpurun_out/r04_q; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -q -m gpu -x -k "tail or nbits or c3 or 16 or qam or step" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python tools/ab.py --config c3 --tunes "13=1;13=3" --what step_pipe,tail_fwd_bwd --rounds 5 > $O/ab_c3.txt 2>&1; cat $O/ab_c3.txt
timeout 200 python tools/steptl.py --config c3 > $O/tl_c3.json 2>$O/tl_c3.err; cat $O/tl_c3.json
timeout 200 python tools/steptl.py --config c3 --tunes 13=3 > $O/tl_c3_f.json 2>$O/tl_c3_f.err; cat $O/tl_c3_f.json
