cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_hip_kernels.py -m gpu -x -q > gpurun_out/x6_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/x6_tests.log
for rep in 1 2; do
PG_TAG=blk1 python tools/probe_potrf_quick.py 4096 8192 >> gpurun_out/x6_potrf.log 2>&1
PG_TAG=blk1_alone0 PG_LEAF3_ALONE=0 python tools/probe_potrf_quick.py 4096 8192 >> gpurun_out/x6_potrf.log 2>&1
PG_TAG=blk0 PG_LEAF3_BLK=0 python tools/probe_potrf_quick.py 4096 8192 >> gpurun_out/x6_potrf.log 2>&1
done
PG_LEAF3_ALONE=0 python tools/probe_cs_tlog.py 4096 > gpurun_out/x6_tlog4096.log 2>&1
