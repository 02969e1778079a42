cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/x27_batch.log
for nbo in 0 256 1024 2048; do
for cfg in "8 4096" "8 2048" "8 9216"; do
if [ $nbo -eq 0 ]; then timeout -k 10 300 python tools/probe_batch.py $cfg 2>&1 | grep "eager_inverse=True" | sed "s/^/[default] /" >> gpurun_out/x27_batch.log
else PG_NBO=$nbo timeout -k 10 300 python tools/probe_batch.py $cfg 2>&1 | grep "eager_inverse=True" | sed "s/^/[nbo$nbo] /" >> gpurun_out/x27_batch.log; fi
done; done
