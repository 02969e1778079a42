cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/x33_potrf.log
PG_CS_SA_STREAM=1 PG_CS_SA_ROWS=2560 timeout -k 10 400 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "coupled or potrf or fused" > gpurun_out/x33_tests.log 2>&1; rc=$?; echo "tests rc=$rc" >> gpurun_out/x33_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
for rep in 1 2; do
PG_TAG=off timeout -k 10 200 python tools/probe_potrf_quick.py 4096 8192 16384 >> gpurun_out/x33_potrf.log 2>&1
for r in 2048 2560 3584 4608; do
PG_TAG=lean_rs$r PG_CS_SA_STREAM=1 PG_CS_SA_ROWS=$r timeout -k 10 200 python tools/probe_potrf_quick.py 4096 8192 16384 >> gpurun_out/x33_potrf.log 2>&1
done
PG_TAG=lean_us2560 PG_CS_SA_STREAM=0 PG_CS_SA_ROWS=2560 timeout -k 10 200 python tools/probe_potrf_quick.py 4096 8192 16384 >> gpurun_out/x33_potrf.log 2>&1
PG_TAG=sa2560_nolean PG_CS_LEAN=0 PG_CS_SA_STREAM=1 PG_CS_SA_ROWS=2560 timeout -k 10 200 python tools/probe_potrf_quick.py 8192 >> gpurun_out/x33_potrf.log 2>&1
done
