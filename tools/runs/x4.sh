cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 5 120 tools/micro/issue_rate > gpurun_out/x4_issue_rate.log 2>&1
