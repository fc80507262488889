#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_query.py tests/test_gpu_pipeline.py tests/test_gpu_main.py -x -q > gpurun_out/r3c_tests.log 2>&1; echo "tests rc $?"
tail -15 gpurun_out/r3c_tests.log
timeout 300 python tools/recon_perf.py 2>&1 | grep "recon" | tee gpurun_out/r3c_recon.log
timeout 200 python tools/quick_perf.py grid 2>&1 | grep "res 256" | tee gpurun_out/r3c_quick.log
