cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 5 60 tools/micro/blockfac > gpurun_out/x5_blockfac.log 2>&1
