#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3d_tests.log 2>&1; echo "tests rc $?"
tail -4 gpurun_out/r3d_tests.log
timeout 300 python tools/recon_perf.py 2>&1 | grep "recon" | tee gpurun_out/r3d_recon.log
AVCAP_LIB=$PWD/avatarcap_amd/csrc/_abl/lib_F_-DAVC_DBG_TIMING_2.so timeout 300 python tools/timing_probe.py --recon gpurun_out/r3d_recon_time_split.md > gpurun_out/r3d_recon_timing.log 2>&1
tail -32 gpurun_out/r3d_recon_timing.log
timeout 200 python tools/quick_perf.py grid 2>&1 | grep "res 256" | tee gpurun_out/r3d_quick.log
timeout 900 python bench.py > gpurun_out/r3d_bench_line.json 2> gpurun_out/r3d_bench.err; echo "bench rc $?"; tail -3 gpurun_out/r3d_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r3d_bench_line.json'))
print({k: d[k] for k in ('value','ms_per_step','ms_per_step_per_rank')}); print(d['roofline']); print(json.dumps(d.get('configs'), indent=1)); print(d.get('masked'))"
