cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
q() { python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-masked --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  dense: fps %.3f launch %.2f ms cycles %.4e clock %.0f reproduced %s' % (d['value'], r['avg_launch_ms'], r['shader_cycles_per_launch'], r['clock_mhz'], d.get('frame_reproduced')))"; }
for lib in libavcap_hip.so libavcap_ab_mlp.so libavcap_hip.so libavcap_ab_mlp.so; do echo "== $lib"; AVCAP_LIB=$PWD/avatarcap_amd/$lib q; done
for lib in libavcap_hip.so libavcap_ab_enc.so libavcap_hip.so libavcap_ab_enc.so; do echo "== $lib"; AVCAP_LIB=$PWD/avatarcap_amd/$lib python tools/enc_perf.py --iters 200 2>&1 | grep -i "ms" | head -4; done
AVCAP_LIB=$PWD/avatarcap_amd/libavcap_ab_enc.so python tools/producer_hash.py 2>&1 | grep "^unet\|^hgfilter"
