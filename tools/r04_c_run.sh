set -x
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest_engine.txt
timeout 600 python tools/ab.py --tunes ";18=1;18=2" --what step_pipe --rounds 7 --iters 400 > $O/ab_ride.txt 2>&1
DCCN_LIB_ALLOW_MISSING=1 DCCN_LIB_PATH=abl/libdccn_r03.so timeout 300 python tools/ab.py --tunes "" --what step_pipe --rounds 5 --iters 400 > $O/ab_r03lib.txt 2>&1
timeout 300 python tools/gapscan.py --modes eager --out $O/gapscan.jsonl > /dev/null 2>$O/gapscan.err
DCCN_TUNE="18=2" timeout 300 python tools/gapscan.py --modes eager --tag ride_trail --out $O/gapscan.jsonl > /dev/null 2>>$O/gapscan.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_args.json 2> $O/bench.err
cat $O/pytest_engine.txt $O/ab_ride.txt $O/ab_r03lib.txt
