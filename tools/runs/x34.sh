cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
PG_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 5 --warmup 1 --no-legs --no-cpu-baseline > gpurun_out/x34_bench_n2.json 2> gpurun_out/x34_bench_n2.err; echo "rc=$?" >> gpurun_out/x34_bench_n2.err
