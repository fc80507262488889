#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_query.py tests/test_normal_fusion.py tests/test_raster.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_producers.py -m gpu -x -q -s > gpurun_out/r3e_tests.log 2>&1; echo "tests rc $?"
grep "slack\|grid vs point\|folded subset" gpurun_out/r3e_tests.log | head -30; tail -3 gpurun_out/r3e_tests.log
timeout 300 python tools/recon_perf.py 2>&1 | grep "recon" | tee gpurun_out/r3e_recon.log
AVCAP_LIB=$PWD/avatarcap_amd/csrc/_abl/lib_F_-DAVC_DBG_TIMING_2.so timeout 300 python tools/timing_probe.py --recon gpurun_out/r3e_recon_time_split.md > gpurun_out/r3e_recon_timing.log 2>&1
grep -A22 "^| # | chunk" gpurun_out/r3e_recon_timing.log | head -40; grep "tile\*\*\|Tile start" gpurun_out/r3e_recon_timing.log
