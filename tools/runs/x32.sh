cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/x32; rm -rf $O; mkdir -p $O
PG_GEMM_LOG=$O/gl.txt rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/tools/probe_potrf_trace.py plain 8192 > $O/run.log 2>&1; echo "rc=$?"
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_util.py $f $O/gl.txt 0.25 > $O/util_n8192.txt 2>&1; echo "util rc=$?"
python3 $R/tools/trace_chain.py $f 3 > $O/chain_n8192.txt 2>&1; echo "chain rc=$?"
cp $f $O/kernel_trace_n8192.csv
rm -rf $O/tr
