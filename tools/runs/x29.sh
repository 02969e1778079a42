cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/x29_potrf.log
for rep in 1 2; do
for pad in 0 20000 45000 70000; do
PG_TAG=pad$pad PG_CS_LDS_PAD=$pad timeout -k 10 200 python tools/probe_potrf_quick.py 4096 8192 >> gpurun_out/x29_potrf.log 2>&1
done
done
