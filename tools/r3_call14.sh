#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_main.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r3n_tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r3n_tests.log
for ov in 1 0; do
  AVC_OVERLAP=$ov timeout 600 python - <<'PY' 2>&1 | tail -1
import os, json, subprocess, sys
from avatarcap_amd import config
config.overlap_frames = os.environ['AVC_OVERLAP'] == '1'
sys.argv = ['bench.py', '--steps', '8', '--warmup', '2', '--no-cpu-baseline', '--no-masked', '--no-configs']
import runpy
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path('bench.py', run_name='__main__')
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print('overlap', config.overlap_frames, 'fps', round(d['value'], 3), 'ms/step', round(d['ms_per_step'], 2), 'launch', round(d['roofline']['avg_launch_ms'], 2), 'MHz', round(d['roofline']['clock_mhz']))
PY
done
