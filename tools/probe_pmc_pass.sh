cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcprobe; rm -rf $O; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/tools/probe_eval_once.py > $O/pmc_f.log 2>&1; echo "pmc_f rc=$?"
grep -n "loss\|Error\|coupled" $O/pmc_f.log | head
