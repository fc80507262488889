#!/bin/bash
# same-box A/B of builds of fused_mlp.hip (avatarcap_amd/libavcap_ab_<tag>.so beside the product library): shader cycles of the dense launch, and whether the bits moved
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
q() { python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-masked --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  dense: fps %.3f launch %.2f ms cycles %.4e clock %.0f' % (d['value'], r['avg_launch_ms'], r['shader_cycles_per_launch'], r['clock_mhz']))"; }
for rep in 1 2; do for lib in hip $(ls avatarcap_amd/libavcap_ab_*.so | sed 's/.*libavcap_ab_//; s/.so//'); do
  L=avatarcap_amd/libavcap_ab_$lib.so; [ $lib = hip ] && L=avatarcap_amd/libavcap_hip.so
  echo "== $lib"; AVCAP_LIB=$PWD/$L q; [ $rep = 1 ] && AVCAP_LIB=$PWD/$L python tools/query_hash.py 2>&1 | grep "^avatar" | cut -c1-60
done; done
