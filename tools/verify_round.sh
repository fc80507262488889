#!/bin/bash
# usage (on the GPU box, through gpurun): tools/verify_round.sh <tag> -- end-of-round verification on one box: the whole GPU suite, smoke(), the profile round
TAG=${1:-verify}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/${TAG}_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh $TAG
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_line.json')); r=d['roofline']
print('fps', d['value'], 'ms/step', d['ms_per_step'], 'launch', r['avg_launch_ms'], 'MHz', r['clock_mhz'], 'frac', r['frac'], 'vs sustained', r['mfma_issued_vs_sustained'], 'masked', d['masked']['value'])
print('configs[2]', d['configs']['configs[2]']['ms_per_frame'], d['configs']['configs[2]']['kernel_ms'], 'configs[3]', d['configs']['configs[3]']['avatar_frame_ms'], d['configs']['configs[3]']['colour_ms'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
