cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/x19_potrf.log
for rep in 1 2; do
PG_TAG=base timeout -k 10 200 python tools/probe_potrf_quick.py 8192 12288 >> gpurun_out/x19_potrf.log 2>&1
for sr in 6144 4096 3072; do for nbo in 512 1024; do
PG_TAG=sync${sr}_nbo$nbo PG_SYNC_ROWS=$sr PG_NBO=$nbo timeout -k 10 200 python tools/probe_potrf_quick.py 8192 12288 >> gpurun_out/x19_potrf.log 2>&1
done; done
done
