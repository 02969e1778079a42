cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/probe_cs_tlog.py 4096 > gpurun_out/x7_tlog4096.log 2>&1
