#!/bin/bash
# round 3, GPU call 2: parity of the restructured query (wide chunks, balanced split), same-box A/B against round 2's kernel, the time split
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_query.py tests/test_gpu_pipeline.py -x -q > gpurun_out/r3b_tests.log 2>&1; echo "tests rc $?"
tail -5 gpurun_out/r3b_tests.log
for round in 1 2; do
  for L in avatarcap_amd/csrc/_abl/lib_R2.so avatarcap_amd/libavcap_hip.so; do
    echo -n "== $round $(basename $L): "; AVCAP_LIB=$PWD/$L timeout 200 python tools/quick_perf.py grid 2>&1 | grep "res 256"
  done
done | tee gpurun_out/r3b_ab.log
AVCAP_LIB=$PWD/avatarcap_amd/csrc/_abl/lib_F_-DAVC_DBG_TIMING_2.so timeout 300 python tools/timing_probe.py gpurun_out/r3b_time_split.md > gpurun_out/r3b_timing.log 2>&1
tail -25 gpurun_out/r3b_timing.log
for L in avatarcap_amd/csrc/_abl/lib_R2.so avatarcap_amd/libavcap_hip.so; do
  echo -n "== recon $(basename $L): "; AVCAP_LIB=$PWD/$L timeout 200 python tools/recon_perf.py 2>&1 | grep "recon decode"
done | tee gpurun_out/r3b_recon.log
