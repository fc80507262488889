#!/bin/bash
# round 3, first GPU call: parity of the rebuilt query, same-box A/B against round 2's kernel, the time split
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_query.py -x -q > gpurun_out/r3a_query_tests.log 2>&1; echo "query tests rc $?"
tail -3 gpurun_out/r3a_query_tests.log
for round in 1 2; do
  for L in avatarcap_amd/csrc/_abl/lib_R2.so avatarcap_amd/libavcap_hip.so avatarcap_amd/csrc/_abl/lib_NO_BARRIER.so; do
    echo -n "== $round $(basename $L): "; AVCAP_LIB=$PWD/$L timeout 200 python tools/quick_perf.py grid 2>&1 | grep "res 256"
  done
done | tee gpurun_out/r3a_ab.log
AVCAP_LIB=$PWD/avatarcap_amd/csrc/_abl/lib_F_-DAVC_DBG_TIMING_2.so timeout 300 python tools/timing_probe.py gpurun_out/r3a_time_split.md > gpurun_out/r3a_timing.log 2>&1
tail -40 gpurun_out/r3a_timing.log
