cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(verbose=False); g.smoke(); print('smoke ok')" > gpurun_out/x28_smoke.log 2>&1; echo "rc=$?" >> gpurun_out/x28_smoke.log
PG_LEAF3_BLK=0 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "potrf or leaf or coupled" > gpurun_out/x28_blk0_tests.log 2>&1; echo "rc=$?" >> gpurun_out/x28_blk0_tests.log
PG_CS_ROWS16=0 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "potrf or coupled" > gpurun_out/x28_rows32_tests.log 2>&1; echo "rc=$?" >> gpurun_out/x28_rows32_tests.log
