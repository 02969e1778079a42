cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/x22_potrf.log
timeout -k 10 400 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "coupled or potrf or fused or leaf" > gpurun_out/x22_tests.log 2>&1; rc=$?; echo "tests rc=$rc" >> gpurun_out/x22_tests.log
for rep in 1 2; do
PG_TAG=nth1024 timeout -k 10 200 python tools/probe_potrf_quick.py 4096 8192 >> gpurun_out/x22_potrf.log 2>&1
done
timeout -k 10 200 python tools/probe_cs_tlog.py 4096 > gpurun_out/x22_tlog4096.log 2>&1
