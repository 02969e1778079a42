cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ldsconf; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/a -- python3 $R/tools/probe_gemm_variants.py 4096 > $O/a.log 2>&1; echo "rc=$?"
f=$(find $O/a -name "*counter_collection.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if 'pg_gemm_kernel' in r['Kernel_Name']:
        acc[r['Kernel_Name'][:70]][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in acc.items():
    print(k, dict(v), "conflict/active = %.3f" % (v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
PY
