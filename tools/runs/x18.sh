cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/x18_potrf.log
for rep in 1 2; do
for r in 32 24 16 48; do
PG_TAG=res$r PG_RESERVED_CUS=$r timeout -k 10 200 python tools/probe_potrf_quick.py 4096 8192 16384 >> gpurun_out/x18_potrf.log 2>&1
done
done
