cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/x13_potrf.log
for rep in 1 2; do
PG_TAG=rows16 python tools/probe_potrf_quick.py 4096 8192 >> gpurun_out/x13_potrf.log 2>&1
PG_TAG=rows16_direct PG_CS_ROWS16_DIRECT=1 python tools/probe_potrf_quick.py 4096 8192 >> gpurun_out/x13_potrf.log 2>&1
PG_TAG=rows16_6144 PG_CS_ROWS16=6144 python tools/probe_potrf_quick.py 8192 >> gpurun_out/x13_potrf.log 2>&1
done
PG_CS_ROWS16_DIRECT=1 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "coupled or potrf" > gpurun_out/x13_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/x13_tests.log
