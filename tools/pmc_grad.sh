# SQ counters of the matrix-pipe gradient contraction (pg_grad_mfma_kernel, N = 16384, D = 16, fp64): how busy the matrix pipe and the VALU
# are, per launch.  Four passes (at most four counters each), the program directly after `--`.  Output: one JSON line per pass.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcg; rm -rf $O; mkdir -p $O
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$(echo $set | md5sum | cut -c1-6)
  rocprofv3 --pmc $set --output-format csv -d $O/p_$i -- python3 $R/tools/probe_grad_once.py 16 > $O/log_$i.txt 2>&1
  f=$(find $O/p_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<PY
import csv, sys, collections, json
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(float); n = collections.Counter()
for r in rows:
    if "grad_mfma" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print(json.dumps({k: v / max(1, n[k]) * (1 if k != "GRBM_GUI_ACTIVE" else 1) for k, v in agg.items()}))
PY
  rm -rf $O/p_$i
done
