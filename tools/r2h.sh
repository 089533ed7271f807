R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 400 python tools/launch_modes.py > gpurun_out/r2h_modes.log 2>&1; cat gpurun_out/r2h_modes.log
