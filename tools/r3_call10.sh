#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_query.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -m gpu -x -q -s > gpurun_out/r3j_tests.log 2>&1; echo "tests rc $?"
grep "points per column\|folded subset" gpurun_out/r3j_tests.log | head -12; tail -3 gpurun_out/r3j_tests.log
timeout 300 python tools/band_perf.py 2>&1 | grep "band" | tee gpurun_out/r3j_band.log
AVCAP_LIB=$PWD/avatarcap_amd/csrc/_abl/lib_R2.so timeout 300 python tools/band_perf.py 2>&1 | grep "band query" | sed 's/^/R2 kernel: /' | tee -a gpurun_out/r3j_band.log
timeout 200 python tools/quick_perf.py grid 2>&1 | grep "res 256"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
