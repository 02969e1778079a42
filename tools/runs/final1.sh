cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/f1_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/f1_tests.log
python bench.py > gpurun_out/f1_bench.json 2> gpurun_out/f1_bench.err; echo "bench rc=$?" >> gpurun_out/f1_tests.log
bash tools/prof_round.sh > gpurun_out/f1_prof.log 2>&1
