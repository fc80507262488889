#!/bin/bash
# Builds ablation variants of libavcap_hip.so (WRONG results by design; timing attribution only).
set -e
cd "$(dirname "$0")/.."
OBJ=avatarcap_amd/csrc/_obj; OUT=avatarcap_amd/csrc/_abl; mkdir -p $OUT
for V in ${ABL_VARIANTS:-NO_PREFETCH NO_BARRIER PF_SAME}; do
  TAG=$(echo "$V" | tr ' ' '_')
  FL=$(for x in $V; do echo -n "-DAVC_DBG_$x=1 "; done)
  [ -f $OUT/lib_$TAG.so ] && [ $OUT/lib_$TAG.so -nt avatarcap_amd/csrc/fused_mlp.hip ] && continue
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $FL -c avatarcap_amd/csrc/fused_mlp.hip -o $OUT/fused_$TAG.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/fused_$TAG.o $OBJ/misc.hip.o $OBJ/mesh.hip.o $OBJ/raster.hip.o $OBJ/fusion.hip.o $OBJ/knn_lbs.hip.o $OBJ/pack.cpp.o $OBJ/capi.cpp.o -o $OUT/lib_$TAG.so && echo built $TAG ) &
done
wait
ls -la $OUT/*.so
