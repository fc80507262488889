#!/bin/bash
# product library against avatarcap_amd/libavcap_ab_prev.so on one box: the dense line, the band launches (masked, configs[2] kernels), and the query hashes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for L in avatarcap_amd/libavcap_ab_prev.so avatarcap_amd/libavcap_hip.so; do
  echo "== $L"; AVCAP_LIB=$PWD/$L python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['configs']; print('  dense fps %.3f launch %.2f ms cycles %.4e clock %.0f | masked %.2f fps | configs[2] %.2f ms kernels %s | example %.2f ms %s | 512^3 query %.1f ms' % (d['value'], r['avg_launch_ms'], r['shader_cycles_per_launch'], r['clock_mhz'], d['masked']['value'], c['configs[2]']['ms_per_frame'], [round(v,3) for v in c['configs[2]']['kernel_ms'].values()], c['example.yaml']['ms_per_frame'], [round(v,3) for v in c['example.yaml']['kernel_ms'].values()], c['configs[3]']['query_kernel_ms']))"
  [ $rep = 1 ] && AVCAP_LIB=$PWD/$L python tools/query_hash.py 2>&1 | grep "^avatar\|^recon" | cut -c1-150
done; done
