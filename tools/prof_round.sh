cd /tmp && export TMPDIR=/tmp
RD=${PG_ROUND:-r05}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$RD; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench.log 2>&1; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss -- python3 $R/tools/probe_eval_single_stream.py > $O/ss.log 2>&1; echo "ss rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/tools/probe_eval_once.py > $O/pmc_f.log 2>&1; echo "pmc_f rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/tools/probe_eval_once.py > $O/pmc_w.log 2>&1; echo "pmc_w rc=$?"
python3 $R/tools/pmc_summary.py $O/pmc_f $O/pmc_w > $O/${RD}_pmc_eval_traffic.json; echo "summary rc=$?"
# MFMA counters (north_star: MFMA utilisation from rocprof): the factorisation and L^-T L^-1 at N = 16384, each its own pass
for w in potrf lauum; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $O/mfma_$w -- python3 $R/tools/probe_mfma_pass.py $w > $O/mfma_$w.log 2>&1; echo "mfma $w rc=$?"
  python3 $R/tools/pmc_mfma_summary.py $O/mfma_$w $w > $O/${RD}_pmc_mfma_$w.json; echo "mfma $w summary rc=$?"
done
rm -rf $O/mfma_potrf $O/mfma_lauum
PG_NO_PY_ATEXIT=1 rocprofv3 --kernel-trace --output-format csv -d $O/teardown -- python3 $R/tools/probe_potrf.py 8192 potrf_only > $O/teardown.log 2>&1; echo "teardown rc=$?"
f=$(find $O/bench -name "*kernel_stats.csv" | head -1); cp $f $O/${RD}_kernel_stats_bench.csv; python3 $R/tools/stats_summary.py $f > $O/${RD}_kernel_stats_bench.txt
f=$(find $O/ss -name "*kernel_stats.csv" | head -1); cp $f $O/${RD}_kernel_stats_ss.csv; python3 $R/tools/stats_summary.py $f > $O/${RD}_kernel_stats_ss.txt
tail -3 $O/teardown.log
# the covariance build on its own: kernel trace + FETCH / WRITE passes of tools/probe_kbuild_once.py (lower-only and mirrored, N = 16384, D = 8)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kb -- python3 $R/tools/probe_kbuild_once.py > $O/kb.log 2>&1; echo "kb rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/kb_f -- python3 $R/tools/probe_kbuild_once.py > $O/kb_f.log 2>&1; echo "kb_f rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/kb_w -- python3 $R/tools/probe_kbuild_once.py > $O/kb_w.log 2>&1; echo "kb_w rc=$?"
python3 $R/tools/kbuild_summary.py $O/kb $O/kb_f $O/kb_w > $O/${RD}_kernel_build_hbm.json; echo "kb summary rc=$?"
rm -rf $O/bench $O/ss $O/pmc_f $O/pmc_w $O/teardown $O/kb $O/kb_f $O/kb_w
# the chain from the inside (in-kernel stamps): per-step timeline, leaf phases, rows kernel vs leaf
python3 $R/tools/probe_cs_tlog.py 4096 > $O/${RD}_potrf_chain_timeline_n4096.txt 2>&1; echo "tlog4096 rc=$?"
python3 $R/tools/probe_cs_tlog.py 8192 > $O/${RD}_potrf_chain_timeline_n8192.txt 2>&1; echo "tlog8192 rc=$?"
# the leaf's 16 x 16 factor, old form against blocked form, in isolation; the rows kernel alone
$R/tools/micro/blockfac > $O/${RD}_leaf_factor_micro.txt 2>&1; echo "blockfac rc=$?"
python3 $R/tools/probe_rowstep.py > $O/${RD}_rows_kernel_alone.txt 2>&1; echo "rowstep rc=$?"
# the evaluation's wall time by phase (headline schedule) and the N = 8192 factorisation's queues
rocprofv3 --kernel-trace --output-format csv -d $O/ev -- python3 $R/tools/probe_eval_loop.py 4 > $O/ev.log 2>&1; echo "eval loop rc=$?"
f=$(find $O/ev -name "*kernel_trace.csv" | head -1); python3 $R/tools/trace_eval_gaps.py $f > $O/${RD}_eval_phases.txt; rm -rf $O/ev
PG_REC_MIN=0 rocprofv3 --kernel-trace --output-format csv -d $O/p8 -- python3 $R/tools/probe_potrf_trace.py plain 8192 > $O/p8.log 2>&1; echo "potrf trace rc=$?"
f=$(find $O/p8 -name "*kernel_trace.csv" | head -1); python3 $R/tools/trace_chain.py $f 3 > $O/${RD}_potrf_kernel_trace_n8192.txt; rm -rf $O/p8
python3 $R/tools/probe_tri_rounds.py 8192 31 32 45 55 63 64 66 > $O/${RD}_tri_rounds_mixed.txt 2>&1; echo "rounds rc=$?"
PG_GEMM_MIXED=0 python3 $R/tools/probe_tri_rounds.py 8192 31 32 45 55 63 64 66 > $O/${RD}_tri_rounds_plain.txt 2>&1; echo "rounds plain rc=$?"
python3 $R/tools/probe_big_products.py 0 > $O/${RD}_big_products.txt 2>&1; echo "big products rc=$?"
# round 5: the batched gradient path at config 4's dimension (D = 16: the matrix-pipe contraction), the batched diagonal prediction at the
# reference's own test size (launches per prediction: the K = 4 run minus the fit-only run, over 5 predictions), the tile bodies alone
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mle -- python3 $R/tools/probe_mle_batched.py 8 4096 16 4 > $O/mle.log 2>&1; echo "mle rc=$?"
f=$(find $O/mle -name "*kernel_stats.csv" | head -1); python3 $R/tools/stats_summary.py $f > $O/${RD}_mle_nc8_n4096_kernel_stats.txt; tail -1 $O/mle.log >> $O/${RD}_mle_nc8_n4096_kernel_stats.txt; rm -rf $O/mle
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pr -- python3 $R/tools/probe_predict_batched.py 10 100 3 100 4 > $O/pr.log 2>&1; echo "predict rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pr0 -- python3 $R/tools/probe_predict_batched.py 10 100 3 100 0 > $O/pr0.log 2>&1; echo "predict fit-only rc=$?"
PG_PREDICT_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prs -- python3 $R/tools/probe_predict_batched.py 10 100 3 100 4 > $O/prs.log 2>&1; echo "predict serial rc=$?"
{ echo "# batched Exact_GP.predict(var=diag), 10 experts x 100 points, m = 100: 5 predictions (1 warm-up + 4) after one fit"; f=$(find $O/pr -name "*kernel_stats.csv" | head -1); python3 $R/tools/stats_summary.py $f; tail -1 $O/pr.log;
  echo "# the fit alone (subtract)"; f=$(find $O/pr0 -name "*kernel_stats.csv" | head -1); python3 $R/tools/stats_summary.py $f;
  echo "# PG_PREDICT_SERIAL=1: the experts one by one (rounds 1-4), same 5 predictions after one fit"; f=$(find $O/prs -name "*kernel_stats.csv" | head -1); python3 $R/tools/stats_summary.py $f; tail -1 $O/prs.log; } > $O/${RD}_predict_nc10_n100_kernel_stats.txt
rm -rf $O/pr $O/pr0 $O/prs
python3 $R/tools/probe_tile_bodies.py > $O/${RD}_tile_bodies.txt 2>&1; echo "tile bodies rc=$?"
