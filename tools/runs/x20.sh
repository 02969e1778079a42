cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 200 python tools/probe_rowstep.py > gpurun_out/x20_rowstep.log 2>&1
