cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/probe_lauum_nt.py > gpurun_out/x15_lauum.log 2>&1
