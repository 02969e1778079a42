cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/x30_potrf.log
timeout -k 10 400 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "potrf or leaf or coupled or fused" > gpurun_out/x30_tests.log 2>&1; rc=$?; echo "tests rc=$rc" >> gpurun_out/x30_tests.log
for rep in 1 2 3; do
PG_TAG=new timeout -k 10 200 python tools/probe_potrf_quick.py 4096 8192 >> gpurun_out/x30_potrf.log 2>&1
done
timeout -k 10 200 python tools/probe_cs_tlog.py 4096 > gpurun_out/x30_tlog4096.log 2>&1
