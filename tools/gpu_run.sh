#!/bin/bash
# One script for everything a round runs on the GPU box (replaces the per-run tools/rNN_x_run.sh files):
#     gpurun --timeout 900 -- 'bash tools/gpu_run.sh <tag> <section> [<section> ...]'
# Results land under gpurun_out/<tag>/ (merged back by gpurun); copy what is to be judged into profiles/.
# Sections (each bounded by its own timeout; a section's arguments follow it after ':' separated by ','):
#   tests[:pytest -k expression]   pytest -m gpu (whole suite, or a -k selection)    -> pytest_<n>.txt
#   files:<f1>,<f2>                pytest -m gpu on the given test files               -> pytest_files.txt
#   check:<a>:<b>                  tools/variant_check.py --a <a> --b <b> (bitwise A/B of two tuning settings)
#   ab:<what>:<tunes>              tools/ab.py --what <what> --tunes <tunes> (';' between settings), C2
#   abc:<config>:<what>:<tunes>    the same on another BASELINE config (c3, c4)
#   ablib:<what>:<lib>             this build against another build of the library (abl/*.so), alternating processes
#   tl[:tunes]                     in-situ step timeline (tools/steptl.py --config c2)
#   bench | benchd                 bench.py (defaults) | bench.py --steps 20 --warmup 5 (the driver's arguments)
#   kt                             rocprofv3 --kernel-trace --stats of the C2 bench loop  -> kernel_stats.txt
#   pmc                            separate --pmc passes (never combined with sys/hip/hsa traces) -> pmc_counters.txt, pmc_traffic.json
#   ktc                            kernel-trace summaries of C3 and C4
#   eq | eqkt | eqab:<k=v0,v1>     equaliser step bench (73 / 1170 frames) | its kernel-trace summaries | A/B of a tuning key
#   eqloop | e2e | conv            equaliser epoch loop | generate-and-train loop | general-k C-Conv bench
#   convs[:stride] | convkt        the same bench strided | kernel-trace summary of its first shape
#   config5[:args]                 tools/config5_sweep.py at full size -> config5/
#   chains[:args]                  tools/chainbench.py (G equaliser chains on G streams of one process)
#   sh:<command>                   any shell command (output -> sh_<n>.txt)
#   smoke                          __graft_entry__.smoke()
TAG=${1:-r05}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $ROOT
n=0
for sec in "$@"; do
  n=$((n+1))
  name=${sec%%:*}; arg=""; [ "$sec" != "$name" ] && arg=${sec#*:}
  echo "=== [$n] $sec" | tee -a $O/log.txt
  case $name in
    tests)
      if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -q -m gpu -k "$arg" 2>&1 | tail -25 > $O/pytest_$n.txt
      else timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -25 > $O/pytest_$n.txt; fi
      tail -6 $O/pytest_$n.txt ;;
    files)
      timeout 1500 python -m pytest $(echo $arg | tr ',' ' ') -q -m gpu -x -s 2>&1 | tail -150 > $O/pytest_files_$n.txt; tail -40 $O/pytest_files_$n.txt ;;
    check)
      a=${arg%%:*}; b=${arg#*:}
      timeout 300 python tools/variant_check.py --a "$a" --b "$b" 2>&1 | grep -v amdgpu.ids | tee -a $O/variant_check.txt ;;
    ab)
      what=${arg%%:*}; tunes=${arg#*:}
      timeout 600 python tools/ab.py --what "$what" --tunes "$tunes" --rounds 5 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt ;;
    abc)
      cfg=${arg%%:*}; rest=${arg#*:}; what=${rest%%:*}; tunes=${rest#*:}
      timeout 900 python tools/ab.py --config $cfg --what "$what" --tunes "$tunes" --rounds 4 --iters 60 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_$cfg.txt ;;
    ablib)
      what=${arg%%:*}; other=${arg#*:}
      for lib in dl_ofdm_amd/lib/libdccn.so $other dl_ofdm_amd/lib/libdccn.so $other; do
        echo "== $lib" | tee -a $O/ablib.txt
        DCCN_LIB_PATH=$lib timeout 300 python tools/ab.py --what "$what" --rounds 4 2>&1 | grep -v amdgpu.ids | tee -a $O/ablib.txt
      done ;;
    tl)
      timeout 300 python tools/steptl.py --config c2 --tunes "$arg" 2>/dev/null | tee -a $O/timeline.jsonl | python -c "
import json,sys
for ln in sys.stdin:
    r=json.loads(ln); print(r.get('tunes'), 'period_us', r.get('period_us'), [(l.get('name'), l['us'], l.get('gap_before_us')) for l in r['launches']])" ;;
    bench)  timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; python tools/bench_brief.py $O/bench_n1.json ;;
    benchd) timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; python tools/bench_brief.py $O/bench_driver_args.json ;;
    kt)
      timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --no-cpu-baseline --no-kernel-times --no-other-configs --no-sweep --no-e2e > $O/kt.log 2>&1
      python tools/profile_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1; rm -rf $O/kt; head -12 $O/kernel_stats.txt | cut -c1-160 ;;
    pmc)
      i=0
      for set in "FETCH_SIZE" "WRITE_SIZE" \
                 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
                 "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
        i=$((i+1))
        timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$i -o p -- python tools/opbench.py step_pipe --bench-plan --iters 10 > $O/pmc_$i.log 2>&1
      done
      python tools/pmc_summary.py $O $O/pmc_counters.txt $O/pmc_traffic.json > /dev/null 2>&1; rm -rf $O/pmc_[0-9]; cat $O/pmc_counters.txt | cut -c1-170 | head -30 ;;
    ktc)
      for c in c3 c4; do
        timeout 600 rocprofv3 --kernel-trace --stats -d $O/${c}_kt -o kt -- python tools/opbench.py step_pipe --iters 20 --config $c > $O/${c}_kt.log 2>&1
        python tools/profile_summary.py $(find $O/${c}_kt -name "*.db" | head -1) 10 > $O/${c}_kernel_stats.txt 2>&1; rm -rf $O/${c}_kt
      done ;;
    eq)     timeout 600 python tools/eqbench.py --steps 100 2>&1 | grep -v amdgpu.ids | tee $O/eqbench.jsonl | cut -c1-220 ;;
    eqab)   timeout 600 python tools/eqbench.py --frames 73 --steps 200 --ab "$arg" 2>&1 | grep -v amdgpu.ids | tee -a $O/eqbench_ab.jsonl | cut -c1-220 ;;
    eqkt)
      timeout 600 rocprofv3 --kernel-trace --stats -d $O/eq73_kt -o kt -- python tools/eqbench.py --frames 73 --steps 100 --paths fused-eager > $O/eq73_kt.log 2>&1
      python tools/profile_summary.py $(find $O/eq73_kt -name "*.db" | head -1) > $O/eq73_kernel_stats.txt 2>&1; rm -rf $O/eq73_kt
      timeout 600 rocprofv3 --kernel-trace --stats -d $O/eq_kt -o kt -- python tools/eqbench.py --frames 1170 --steps 50 --paths fused-eager > $O/eq_kt.log 2>&1
      python tools/profile_summary.py $(find $O/eq_kt -name "*.db" | head -1) > $O/eq_kernel_stats.txt 2>&1; rm -rf $O/eq_kt
      head -30 $O/eq73_kernel_stats.txt | cut -c1-150 ;;
    eqloop) timeout 600 python tools/eqloop.py 2>&1 | grep -v amdgpu.ids | tee $O/eqloop.jsonl | cut -c1-220 ;;
    e2e)    timeout 600 python tools/e2ebench.py 2>&1 | grep -v amdgpu.ids | tee $O/e2ebench.jsonl | cut -c1-300 ;;
    conv)   timeout 600 python tools/convbench.py 2>&1 | grep -v amdgpu.ids | tee $O/convbench.jsonl | cut -c1-220 ;;
    convs)  timeout 600 python tools/convbench.py --stride ${arg:-2} 2>&1 | grep -v amdgpu.ids | tee $O/convbench_stride${arg:-2}.jsonl | cut -c1-220 ;;
    convkt)
      timeout 600 rocprofv3 --kernel-trace --stats -d $O/conv_kt -o kt -- python tools/convbench.py --shapes 0 --rounds 1 --iters 30 > $O/conv_kt.log 2>&1
      python tools/profile_summary.py $(find $O/conv_kt -name "*.db" | head -1) 14 > $O/conv_kernel_stats.txt 2>&1; rm -rf $O/conv_kt; cut -c1-160 $O/conv_kernel_stats.txt ;;
    config5)
      timeout 2400 python tools/config5_sweep.py --out $O/config5 --eq_epochs 0 $(echo $arg | tr ',' ' ') > $O/config5_run.log 2>&1; tail -3 $O/config5_run.log; cat $O/config5/config5_timing.json | head -40 ;;
    eqgkt)
      # kernel-trace summaries of the equaliser epoch loop: one chain, and G chains per launch sequence (arg: G, default 4)
      G=${arg:-4}
      timeout 600 rocprofv3 --kernel-trace --stats -d $O/eqg1_kt -o kt -- python tools/eqbench.py --chains 1 --steps 400 > $O/eqg1_kt.log 2>&1
      python tools/profile_summary.py $(find $O/eqg1_kt -name "*.db" | head -1) 30 > $O/eqloop1_kernel_stats.txt 2>&1; rm -rf $O/eqg1_kt
      timeout 600 rocprofv3 --kernel-trace --stats -d $O/eqg_kt -o kt -- python tools/eqbench.py --chains $G --only-group --steps 400 > $O/eqg_kt.log 2>&1
      python tools/profile_summary.py $(find $O/eqg_kt -name "*.db" | head -1) 30 > $O/eqloop${G}_kernel_stats.txt 2>&1; rm -rf $O/eqg_kt
      head -34 $O/eqloop1_kernel_stats.txt | cut -c1-150; head -34 $O/eqloop${G}_kernel_stats.txt | cut -c1-150 ;;
    chains) timeout 900 python tools/chainbench.py $(echo $arg | tr ',' ' ') 2>&1 | grep -v amdgpu.ids | tee -a $O/chainbench.jsonl | cut -c1-260 ;;
    sh)     timeout 900 bash -c "$arg" 2>&1 | grep -v amdgpu.ids | tee -a $O/sh_$n.txt | tail -12 ;;
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt ;;
    *) echo "unknown section $name" ;;
  esac
done
