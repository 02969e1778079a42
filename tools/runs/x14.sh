cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/probe_cs_tlog.py 4096 > gpurun_out/x14_tlog4096.log 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/x14_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/x14_tests.log
python bench.py --no-cpu-baseline > gpurun_out/x14_bench.json 2> gpurun_out/x14_bench.err
