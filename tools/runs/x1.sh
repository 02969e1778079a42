set -e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/probe_syrk_smallk.py > gpurun_out/x1_syrk.log 2>&1
PG_TAG=base python tools/probe_potrf_quick.py 8192 16384 > gpurun_out/x1_potrf.log 2>&1
PG_TAG=wide768 PG_CS_PANEL_WIDE=768 python tools/probe_potrf_quick.py 8192 16384 >> gpurun_out/x1_potrf.log 2>&1
PG_TAG=wide512 PG_CS_PANEL_WIDE=512 python tools/probe_potrf_quick.py 8192 16384 >> gpurun_out/x1_potrf.log 2>&1
PG_TAG=wide768r4096 PG_CS_PANEL_WIDE=768 PG_CS_WIDE_ROWS=4096 python tools/probe_potrf_quick.py 8192 16384 >> gpurun_out/x1_potrf.log 2>&1
PG_TAG=sb1024 PG_SB_TILE_THRESH=1024 python tools/probe_potrf_quick.py 8192 >> gpurun_out/x1_potrf.log 2>&1
PG_TAG=wide768sb1024 PG_CS_PANEL_WIDE=768 PG_SB_TILE_THRESH=1024 python tools/probe_potrf_quick.py 8192 >> gpurun_out/x1_potrf.log 2>&1
