#!/bin/bash
# Round profile collection on the GPU box (gpurun): bench line, kernel-trace stats, separate --pmc passes (never combined
# with sys/hip/hsa traces), stage timings.  Results land under gpurun_out/$1; the summaries to keep go to profiles/.
R=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $ROOT
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_args.json 2> $O/bench_driver_args.err
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --no-cpu-baseline --no-kernel-times --no-other-configs --no-sweep --no-e2e > $O/kt.log 2>&1
python tools/profile_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$i -o p -- python tools/opbench.py step --iters 10 > $O/pmc_$i.log 2>&1
done
python tools/pmc_summary.py $O $O/pmc_counters.txt $O/pmc_traffic.json > /dev/null 2>&1
for c in c3 c4; do
  rocprofv3 --kernel-trace --stats -d $O/${c}_kt -o kt -- python tools/opbench.py step_pipe --iters 20 --config $c > $O/${c}_kt.log 2>&1
  python tools/profile_summary.py $(find $O/${c}_kt -name "*.db" | head -1) 10 > $O/${c}_kernel_stats.txt 2>&1
done
python tools/eqbench.py --steps 100 > $O/eqbench.jsonl 2>&1
rocprofv3 --kernel-trace --stats -d $O/eq73_kt -o kt -- python tools/eqbench.py --frames 73 --steps 100 --paths fused-eager > $O/eq73_kt.log 2>&1
python tools/profile_summary.py $(find $O/eq73_kt -name "*.db" | head -1) > $O/eq73_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/eq_kt -o kt -- python tools/eqbench.py --frames 1170 --steps 50 --paths fused-eager > $O/eq_kt.log 2>&1
python tools/profile_summary.py $(find $O/eq_kt -name "*.db" | head -1) > $O/eq_kernel_stats.txt 2>&1
python tools/eqbench.py --steps 100 --ab 20=0,1 2>&1 | grep -v amdgpu.ids > $O/eqbench_replan_ab.jsonl
python tools/eqbench.py --frames 73 --steps 100 --ab 21=0,1 2>&1 | grep -v amdgpu.ids > $O/eqbench_fewrow_ab.jsonl
python tools/eqloop.py 2>&1 | grep -v amdgpu.ids > $O/eqloop.jsonl
python tools/gapscan.py --modes eager,graph,ride2,sleepy,eager 2>/dev/null > $O/gapscan.jsonl
python tools/ramp.py --idles 0,5,50,1000 --steps 60 --reps 2 2>/dev/null > $O/ramp.jsonl
python tools/convbench.py 2>&1 | grep -v amdgpu.ids > $O/convbench.jsonl
rocprofv3 --kernel-trace --stats -d $O/e2e_kt -o kt -- python tools/e2ebench.py --host-steps 0 > $O/e2e_kt.log 2>&1
python tools/profile_summary.py $(find $O/e2e_kt -name "*.db" | head -1) > $O/e2e_kernel_stats.txt 2>&1
DCCN_LIB_PATH=abl/libdccn_trace.so python tools/blocktrace.py --reps 3 --out $O/blocktrace.txt > /dev/null 2>&1
ls $O | head -60
