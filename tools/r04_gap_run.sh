set -x
O=gpurun_out/r04a; mkdir -p $O
ls /sys/class/drm/ > $O/sysfs_ls.txt 2>&1; ls /sys/class/drm/card*/device/ >> $O/sysfs_ls.txt 2>&1
python -c "import bench,json; print(json.dumps(bench.box_fingerprint())); print(json.dumps(bench.gpu_clock_state()))" > $O/box.json 2>&1
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_bench.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest_engine.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
timeout 600 python tools/gapscan.py --out $O/gapscan.jsonl > /dev/null 2> $O/gapscan.err
DCCN_LIB_ALLOW_MISSING=1 DCCN_LIB_PATH=abl/libdccn_r03.so timeout 300 python tools/gapscan.py --modes eager --no-trace --tag r03lib --out $O/gapscan.jsonl > /dev/null 2>> $O/gapscan.err
GPU_MAX_HW_QUEUES=1 timeout 300 python tools/gapscan.py --modes eager --tag hwq1 --out $O/gapscan.jsonl > /dev/null 2>> $O/gapscan.err
HSA_ENABLE_INTERRUPT=0 timeout 300 python tools/gapscan.py --modes eager --tag noint --out $O/gapscan.jsonl > /dev/null 2>> $O/gapscan.err
HIP_FORCE_DEV_KERNARG=0 timeout 300 python tools/gapscan.py --modes eager --tag hostkernarg --out $O/gapscan.jsonl > /dev/null 2>> $O/gapscan.err
AMD_DIRECT_DISPATCH=0 timeout 300 python tools/gapscan.py --modes eager --tag nodirect --out $O/gapscan.jsonl > /dev/null 2>> $O/gapscan.err
tail -3 $O/pytest_engine.txt; head -c 600 $O/bench_driver_args.json; tail -5 $O/gapscan.err
