set -x
O=gpurun_out/r04k; mkdir -p $O
timeout 600 python tools/variant_check.py --config c4 --frames 585 --steps 2 --a "" --b "23=14" > $O/vc14.txt 2>&1
timeout 600 python tools/variant_check.py --config c4 --frames 585 --steps 2 --a "" --b "23=16" > $O/vc16.txt 2>&1
timeout 900 python tools/ab.py --config c4 --tunes ";23=14;23=15;23=16;22=0" --what step_pipe --rounds 3 --iters 30 > $O/ab_c4.txt 2>&1
cat $O/vc14.txt $O/vc16.txt $O/ab_c4.txt | grep -v amdgpu.ids
