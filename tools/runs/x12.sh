cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/x12_potrf.log
for rep in 1 2; do
PG_TAG=base python tools/probe_potrf_quick.py 4096 8192 >> gpurun_out/x12_potrf.log 2>&1
PG_TAG=rows16_all PG_CS_ROWS16=100000 python tools/probe_potrf_quick.py 4096 8192 >> gpurun_out/x12_potrf.log 2>&1
PG_TAG=rows16_4096 PG_CS_ROWS16=4096 python tools/probe_potrf_quick.py 4096 8192 >> gpurun_out/x12_potrf.log 2>&1
PG_TAG=rows16_2048 PG_CS_ROWS16=2048 python tools/probe_potrf_quick.py 4096 8192 >> gpurun_out/x12_potrf.log 2>&1
done
PG_CS_ROWS16=4096 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "coupled or potrf" > gpurun_out/x12_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/x12_tests.log
