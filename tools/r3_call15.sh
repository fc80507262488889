#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfgv in "0 8" "1 12" "1 16" "1 24" "0 8"; do
  set -- $cfgv
  AVC_OVERLAP=$1 AVC_SPARE=$2 timeout 600 python - <<'PY' 2>&1 | tail -1
import os, json, sys, io, contextlib, runpy
from avatarcap_amd import config
config.overlap_frames = os.environ['AVC_OVERLAP'] == '1'
config.overlap_spare_cus = int(os.environ['AVC_SPARE'])
sys.argv = ['bench.py', '--steps', '8', '--warmup', '2', '--no-cpu-baseline', '--no-masked', '--no-configs']
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path('bench.py', run_name='__main__')
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print('overlap', config.overlap_frames, 'spare', config.overlap_spare_cus, 'fps', round(d['value'], 3), 'ms/step', round(d['ms_per_step'], 2), 'launch', round(d['roofline']['avg_launch_ms'], 2), 'MHz', round(d['roofline']['clock_mhz']), 'verts', d['config']['vertices_last_frame'])
PY
done
