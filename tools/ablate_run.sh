#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== baseline"; timeout 120 python tools/quick_perf.py 2>&1 | grep "res 256"
for L in avatarcap_amd/csrc/_abl/lib_*.so; do echo "== $L"; AVCAP_LIB=$PWD/$L timeout 120 python tools/quick_perf.py 2>&1 | grep "res 256"; done
