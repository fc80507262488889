#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 ./tools/ubench/mfma_order | tee gpurun_out/r3g_mfma_order.log
timeout 1500 python -m pytest tests/test_normal_fusion.py tests/test_raster.py -m gpu -x -q -s > gpurun_out/r3g_tests.log 2>&1; echo "tests rc $?"
grep "slack" gpurun_out/r3g_tests.log | cut -c1-400 | head -12; tail -2 gpurun_out/r3g_tests.log
