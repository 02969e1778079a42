cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof2; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench.log 2>&1; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss -- python3 $R/tools/probe_eval_single_stream.py > $O/ss.log 2>&1; echo "ss rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/tools/probe_eval_once.py > $O/pmc_f.log 2>&1; echo "pmc_f rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/tools/probe_eval_once.py > $O/pmc_w.log 2>&1; echo "pmc_w rc=$?"
python3 $R/tools/pmc_summary.py $O/pmc_f $O/pmc_w > $O/r02_pmc_eval_traffic.json; echo "summary rc=$?"
PG_NO_PY_ATEXIT=1 rocprofv3 --kernel-trace --output-format csv -d $O/teardown -- python3 $R/tools/probe_potrf.py 8192 potrf_only > $O/teardown.log 2>&1; echo "teardown rc=$?"
f=$(find $O/bench -name "*kernel_stats.csv" | head -1); cp $f $O/r02_kernel_stats_bench.csv; python3 $R/tools/stats_summary.py $f > $O/r02_kernel_stats_bench.txt
f=$(find $O/ss -name "*kernel_stats.csv" | head -1); cp $f $O/r02_kernel_stats_ss.csv; python3 $R/tools/stats_summary.py $f > $O/r02_kernel_stats_ss.txt
tail -3 $O/teardown.log
rm -rf $O/bench $O/ss $O/pmc_f $O/pmc_w $O/teardown
