#!/bin/bash
# usage: tools/disasm.sh [extra hipcc flags]  -> /tmp/dis/fm.s (device assembly of csrc/fused_mlp.hip) and a per-kernel instruction histogram
# of avatar_kernel<true,false,1> (the dense launch bench.py times) in /tmp/dis/k1.hist
set -e
mkdir -p /tmp/dis
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form "$@" --cuda-device-only -S \
    "$(dirname "$0")/../avatarcap_amd/csrc/fused_mlp.hip" -o /tmp/dis/fm.s 2>/dev/null
python3 - <<'PY'
import re, collections
txt = open('/tmp/dis/fm.s').read()
for name, tag in (('_ZN3avc5plain13avatar_kernelILb1ELb0ELi1EEEvNS0_11QueryParamsE', 'k1'), ('_ZN3avc5plain12recon_kernelENS0_11QueryParamsE', 'recon')):
    m = re.search(r'^' + name + r':[^\n]*\n(.*?)^\s*s_endpgm', txt, re.S | re.M)
    if not m:
        continue
    body = m.group(1)
    open(f'/tmp/dis/{tag}.s', 'w').write(body)
    ops = [l.split()[0] for l in body.splitlines() if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    h = collections.Counter(ops)
    n = h['v_mfma_f32_32x32x16_f16']
    acc = sum(v for k, v in h.items() if k.startswith('v_accvgpr'))
    meta = re.search(r'\.amdhsa_next_free_vgpr (\d+)', txt[m.end():])
    regs = re.search(name + r'.*?\.vgpr_count:\s+(\d+)', txt, re.S)
    agpr = re.search(name + r'.*?\.agpr_count:\s+(\d+)', txt, re.S)
    print(f'{tag}: {len(ops)} instructions, {n} MFMA, {len(ops)/max(n,1):.2f} per MFMA; accvgpr {acc} ({acc/max(n,1):.2f}/MFMA), s_nop {h["s_nop"]}, s_waitcnt {h["s_waitcnt"]}, '
          f'ds_read {h["ds_read_b128"]}, buffer_load {h["buffer_load_dwordx4"]}, global_load {h["global_load_dwordx4"]}, scratch {sum(v for k, v in h.items() if k.startswith("scratch"))}')
    with open(f'/tmp/dis/{tag}.hist', 'w') as f:
        for k, v in h.most_common():
            f.write(f'{v:6d} {k}\n')
for k in ('k1', 'recon'):
    pass
import subprocess
print(subprocess.run("grep -A30 'avatar_kernelILb1ELb0ELi1' /tmp/dis/fm.s | grep -m3 'vgpr_count\\|agpr_count\\|sgpr_count' ; grep 'ILb1ELb0ELi1.*\\.num_vgpr\\|ILb1ELb0ELi1.*\\.num_agpr' /tmp/dis/fm.s | head -4", shell=True, capture_output=True, text=True).stdout)
PY
