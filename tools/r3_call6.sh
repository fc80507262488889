#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_query.py tests/test_normal_fusion.py tests/test_raster.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -m gpu -x -q -s > gpurun_out/r3f_tests.log 2>&1; echo "tests rc $?"
grep "slack" gpurun_out/r3f_tests.log | head -12; tail -2 gpurun_out/r3f_tests.log
timeout 300 python tools/recon_perf.py 2>&1 | grep "recon" | tee gpurun_out/r3f_recon.log
AVCAP_LIB=$PWD/avatarcap_amd/csrc/_abl/lib_F_-DAVC_DBG_TIMING_2.so timeout 300 python tools/timing_probe.py --recon gpurun_out/r3f_recon_time_split.md > gpurun_out/r3f_recon_timing.log 2>&1
timeout 400 python tools/power_wall.py 2>&1 | grep "^|" | tee gpurun_out/r3f_power_wall.md
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3f_ff -o ff -- python tools/full_frame_prof.py 3 merge > gpurun_out/r3f_ff.log 2>&1
python tools/summarize_prof.py $(ls gpurun_out/prof_r3f_ff/*/ff_kernel_stats.csv gpurun_out/prof_r3f_ff/ff_kernel_stats.csv 2>/dev/null | head -1) gpurun_out/r3f_full_frame_split.md "BASELINE configs[2] chained frame (steps 1-3, band-masked 256^3, fusion=merge): python tools/full_frame_prof.py 3 merge, 4 frames incl. warm-up"
tail -2 gpurun_out/r3f_ff.log
bash tools/run_pmc.sh r3f 256 grid > gpurun_out/r3f_pmc.log 2>&1; tail -30 gpurun_out/pmc_r3f/summary.txt
bash tools/run_pmc.sh r3f_recon 256 recon > gpurun_out/r3f_pmc_recon.log 2>&1; tail -30 gpurun_out/pmc_r3f_recon/summary.txt
