cd $GRAFT_REPO_ROOT
timeout 1200 python tools/eq_trajectory.py --steps 200 --out gpurun_out/eq_traj.csv 2>&1 | grep -v amdgpu | tail -14
