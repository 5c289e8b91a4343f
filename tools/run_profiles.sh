#!/bin/bash
# Round profile collection on the GPU box (gpurun): bench line, kernel-trace stats, four separate --pmc passes
# (never combined with sys/hip/hsa traces), stage timings.  Results land under gpurun_out/$1.
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $ROOT
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python bench.py --graph --no-cpu-baseline --no-other-configs --no-kernel-times > $O/bench_graph.json 2> $O/bench_graph.err
python bench.py --no-pipeline --no-cpu-baseline --no-other-configs --no-kernel-times > $O/bench_no_pipeline.json 2> $O/bench_no_pipeline.err
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --no-cpu-baseline --no-kernel-times > $O/kt.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$i -o p -- python tools/opbench.py step --iters 10 > $O/pmc_$i.log 2>&1
done
rocprofv3 --kernel-trace --stats -d $O/c4_kt -o kt -- python tools/opbench.py step --iters 20 --config c4 > $O/c4_kt.log 2>&1
python tools/eqbench.py --steps 100 > $O/eqbench.json 2>&1
rocprofv3 --kernel-trace --stats -d $O/eq73_kt -o kt -- python tools/eqbench.py --frames 73 --steps 100 --paths fused-eager > $O/eq73_kt.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/eq_kt -o kt -- python tools/eqbench.py --frames 1170 --steps 100 --paths fused-graph > $O/eq_kt.log 2>&1
python tools/e2ebench.py > $O/e2e.json 2>&1
python tools/e2ebench.py --channel AWGN >> $O/e2e.json 2>&1
rocprofv3 --kernel-trace --stats -d $O/e2e_kt -o kt -- python tools/e2ebench.py --host-steps 0 > $O/e2e_kt.log 2>&1
ls -R $O | head -50
