cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/x26_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/x26_tests.log
