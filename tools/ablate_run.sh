#!/bin/bash
# A/B of libavcap_hip.so variants inside ONE gpurun call (box-to-box variance is larger than most effects):
# every variant in avatarcap_amd/csrc/_abl/lib_*.so, two rounds, dense 256^3 query time (and the recon query with ABL_RECON=1).
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for L in avatarcap_amd/csrc/_abl/lib_*.so; do
    echo -n "== $round $(basename $L): "; AVCAP_LIB=$PWD/$L timeout 120 python tools/quick_perf.py 2>&1 | grep "res 256"
    [ -n "$ABL_RECON" ] && AVCAP_LIB=$PWD/$L timeout 120 python tools/recon_perf.py 2>&1 | grep "recon decode"
  done
done
