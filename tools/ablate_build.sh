#!/bin/bash
# Builds ablation variants of libavcap_hip.so (WRONG results by design unless noted; timing attribution only).
# ABL_VARIANTS: space-separated list; each variant is a '+'-joined list of AVC_DBG_<X> knobs, or BASE, or a raw flag set
# introduced by 'F:' (e.g. "F:-mllvm,-amdgpu-mfma-vgpr-form").
set -e
cd "$(dirname "$0")/.."
OBJ=avatarcap_amd/csrc/_obj; OUT=avatarcap_amd/csrc/_abl; mkdir -p $OUT
for V in ${ABL_VARIANTS:-BASE NO_PREFETCH NO_BARRIER NO_EPI NO_LDSREAD NO_PREFETCH+NO_BARRIER+NO_EPI+NO_LDSREAD}; do
  TAG=$(echo "$V" | tr '+:,=' '____')
  FL=""
  case "$V" in
    BASE) ;;
    F:*) FL=$(echo "${V#F:}" | tr ',' ' ') ;;
    *) FL=$(for x in $(echo "$V" | tr '+' ' '); do echo -n "-DAVC_DBG_$x=1 "; done) ;;
  esac
  [ -f $OUT/lib_$TAG.so ] && [ $OUT/lib_$TAG.so -nt avatarcap_amd/csrc/fused_mlp.hip ] && continue
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form $FL -c avatarcap_amd/csrc/fused_mlp.hip -o $OUT/fused_$TAG.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/fused_$TAG.o $OBJ/fused_mlp_checked.o $OBJ/misc.hip.o $OBJ/mesh.hip.o $OBJ/raster.hip.o $OBJ/fusion.hip.o $OBJ/knn_lbs.hip.o $OBJ/pack.cpp.o $OBJ/capi.cpp.o -o $OUT/lib_$TAG.so && echo built $TAG ) &
done
wait
ls -la $OUT/*.so
