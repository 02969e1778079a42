cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py > gpurun_out/x35_bench.json 2> gpurun_out/x35_bench.err; echo "rc=$?" >> gpurun_out/x35_bench.err
