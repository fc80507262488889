"""roofline.traffic of the bench line, measured rather than typed in.

    python tools/pmc_traffic.py collect [outdir]    on the GPU box: two rocprofv3 --pmc passes of their own (counters + --kernel-trace only) over the launch bench.py
                                                    times (tools/pmc_probe.py 256 1 grid), then writes profiles/pmc_traffic.json
    python tools/pmc_traffic.py sha                 the hash of the sources the dominant kernel is built from

profiles/pmc_traffic.json carries the counters, the derived bytes per launch ((2 x FETCH_SIZE + WRITE_SIZE) x 1024: MI355X_MICROARCH.md's gfx950 correction) and the
hash of the kernel's sources at collection time.  bench.py puts the bytes on the line only while that hash equals the tree's; otherwise `traffic` is null and
`traffic_stale` says why -- the figure cannot silently outlive the code it was measured on."""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ['avatarcap_amd/csrc/fused_mlp.hip', 'avatarcap_amd/csrc/mlp_layout.h', 'avatarcap_amd/csrc/pack.cpp']
JSON = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')


def source_sha(root=ROOT):
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(root, f), 'rb').read())
    return h.hexdigest()[:16]


def derive(counters):
    """{'avatar_kernel': {'FETCH_SIZE': KB, 'WRITE_SIZE': KB}, 'column_terms_kernel': {...}} -> bytes per launch (query + its column pass)"""
    return int(sum((2.0 * k['FETCH_SIZE'] + k['WRITE_SIZE']) * 1024.0 for k in counters.values()))


def load(root=ROOT):
    """-> (bytes or None, reference dict): None when the file is missing or was measured on other kernel sources"""
    try:
        d = json.load(open(os.path.join(root, 'profiles', 'pmc_traffic.json')))
    except (OSError, ValueError):
        return None, {'file': 'profiles/pmc_traffic.json', 'state': 'missing'}
    ref = {'file': 'profiles/pmc_traffic.json', 'bytes': d['bytes'], 'source_sha': d['source_sha'], 'command': d['command'], 'round': d.get('round')}
    if d['source_sha'] != source_sha(root):
        ref['state'] = 'stale: the kernel sources changed since the counters were collected (tools/pmc_traffic.py collect)'
        return None, ref
    ref['state'] = 'current'
    return int(d['bytes']), ref


def collect(out):
    os.makedirs(out, exist_ok=True)
    passes = ['FETCH_SIZE TCC_REQ', 'WRITE_SIZE TCC_HIT TCC_MISS']
    cmd = 'python tools/pmc_probe.py 256 1 grid'
    for i, c in enumerate(passes):
        subprocess.run(f'timeout 300 rocprofv3 --pmc {c} --kernel-trace --output-format csv --kernel-include-regex "avatar_kernel|column_terms_kernel" '
                       f'-d {out}/p{i} -o p{i} -- {cmd} > {out}/p{i}.log 2>&1', shell=True, cwd=ROOT, check=False)
    agg = {}
    for f in sorted(glob.glob(f'{out}/p*/**/*counter_collection.csv', recursive=True)):
        for r in csv.DictReader(open(f)):
            k = 'column_terms_kernel' if 'column_terms' in r['Kernel_Name'] else 'avatar_kernel'
            agg.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])       # the last dispatch wins (the first is the warm-up)
    if not all(c in agg.get(k, {}) for k in ('avatar_kernel', 'column_terms_kernel') for c in ('FETCH_SIZE', 'WRITE_SIZE')):
        print('pmc_traffic: counters missing:', agg, file=sys.stderr)
        return 1
    d = {'round': 6, 'command': f'rocprofv3 --pmc <{" | ".join(passes)}> --kernel-trace -- {cmd}', 'counters_kb_and_counts': agg, 'bytes': derive(agg),
         'formula': '(2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over avatar_kernel<true,false,1> and its column_terms_kernel, last dispatch',
         'source_sha': source_sha(), 'sources': KERNEL_SOURCES}
    os.makedirs(os.path.dirname(JSON), exist_ok=True)
    json.dump(d, open(JSON, 'w'), indent=1)
    json.dump(d, open(os.path.join(out, 'pmc_traffic.json'), 'w'), indent=1)
    print(json.dumps(d, indent=1))
    return 0


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'collect':
        sys.exit(collect(sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/pmc_traffic'))
    print(source_sha())
