# I-cache behaviour of the leaf: counters per kernel (serialised by the counter pass: classic chain, pg_leaf3_kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/x2; rm -rf $O; mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -i -E "icache|ifetch|SQ_WAIT_INST|SQ_INST_CYCLES|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_INSTS_VALU |SQ_INSTS_SALU|SQ_ACTIVE_INST" $O/avail.txt | head -60 > $O/avail_grep.txt
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $O/a -- python3 $R/tools/probe_leaf.py > $O/a.log 2>&1; echo "a rc=$?"
rocprofv3 --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/b -- python3 $R/tools/probe_leaf.py > $O/b.log 2>&1; echo "b rc=$?"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $O/c -- python3 $R/tools/probe_potrf_quick.py 4096 > $O/c.log 2>&1; echo "c rc=$?"
for d in a b c; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 - "$f" > $O/$d.summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(k, r['Counter_Name'])] += 1
for k, v in acc.items():
    print(k, {c: (round(x / cnt[(k, c)], 1), cnt[(k, c)]) for c, x in v.items()})
PY
done
rm -rf $O/a $O/b $O/c
