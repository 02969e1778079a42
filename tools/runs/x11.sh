cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/x11_potrf.log
for rep in 1 2; do
for w in 384 256 128 512; do
PG_TAG=panel$w PG_CS_PANEL=$w python tools/probe_potrf_quick.py 4096 8192 >> gpurun_out/x11_potrf.log 2>&1
done
done
PG_CS_PANEL=256 python tools/probe_cs_tlog.py 4096 > gpurun_out/x11_tlog4096_p256.log 2>&1
