cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/x8_potrf.log
python -m pytest tests/test_hip_kernels.py -m gpu -x -q > gpurun_out/x8_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/x8_tests.log
for rep in 1 2; do
PG_TAG=blk1 python tools/probe_potrf_quick.py 4096 8192 16384 >> gpurun_out/x8_potrf.log 2>&1
PG_TAG=blk0 PG_LEAF3_BLK=0 python tools/probe_potrf_quick.py 4096 8192 16384 >> gpurun_out/x8_potrf.log 2>&1
done
python tools/probe_cs_tlog.py 4096 > gpurun_out/x8_tlog4096.log 2>&1
python tools/probe_batch.py 8 4096 > gpurun_out/x8_batch.log 2>&1
