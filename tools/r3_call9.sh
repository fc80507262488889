#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_producers.py tests/test_gpu_pipeline.py tests/test_gpu_main.py -m gpu -x -q -s > gpurun_out/r3i_tests.log 2>&1; echo "tests rc $?"
grep -i "warn.*graph\|hipGraph\|Error" gpurun_out/r3i_tests.log | head; tail -3 gpurun_out/r3i_tests.log
timeout 600 python tools/full_frame_prof.py 6 merge 2>&1 | tail -1
timeout 600 python - <<'PY' 2>&1 | tail -2
import sys; sys.argv = ['x', '6', 'merge']
from avatarcap_amd import config
config.hg_graph = False
exec(open('tools/full_frame_prof.py').read())
PY
