cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/probe_cs_tlog.py 8192 > gpurun_out/x16_tlog8192.log 2>&1
python tools/probe_cs_tlog.py 4096 > gpurun_out/x16_tlog4096.log 2>&1
