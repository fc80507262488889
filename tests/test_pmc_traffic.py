"""roofline.traffic is a measurement with a provenance, not a literal: bench.py takes it from profiles/pmc_traffic.json, which
tools/pmc_traffic.py writes from rocprofv3 --pmc counters together with the hash of the dominant kernel's sources -- and drops it when those sources change."""
import json
import os
import shutil
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import pmc_traffic  # noqa: E402


def test_committed_traffic_equals_its_counters():
    d = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
    c = d['counters_kb_and_counts']
    assert d['bytes'] == pmc_traffic.derive(c) == int(sum((2.0 * c[k]['FETCH_SIZE'] + c[k]['WRITE_SIZE']) * 1024.0 for k in ('avatar_kernel', 'column_terms_kernel')))
    assert 0.3e9 < d['bytes'] < 1.0e9                              # the dense 256^3 launch moves about half a gigabyte (0.087 GB algorithmic)
    assert c['avatar_kernel']['WRITE_SIZE'] * 1024 >= 4 * 256 ** 3     # at least the occupancy volume itself is written


def test_traffic_goes_null_when_the_kernel_sources_change(tmp_path):
    for f in pmc_traffic.KERNEL_SOURCES + ['profiles/pmc_traffic.json']:
        os.makedirs(tmp_path / os.path.dirname(f), exist_ok=True)
        shutil.copy(os.path.join(ROOT, f), tmp_path / f)
    d = json.load(open(tmp_path / 'profiles' / 'pmc_traffic.json'))
    d['source_sha'] = pmc_traffic.source_sha(str(tmp_path))
    json.dump(d, open(tmp_path / 'profiles' / 'pmc_traffic.json', 'w'))
    b, ref = pmc_traffic.load(str(tmp_path))
    assert b == d['bytes'] and ref['state'] == 'current'
    with open(tmp_path / pmc_traffic.KERNEL_SOURCES[0], 'a') as fh:
        fh.write('\n// edited\n')
    b, ref = pmc_traffic.load(str(tmp_path))
    assert b is None and ref['state'].startswith('stale') and ref['bytes'] == d['bytes']
    os.remove(tmp_path / 'profiles' / 'pmc_traffic.json')
    b, ref = pmc_traffic.load(str(tmp_path))
    assert b is None and ref['state'] == 'missing'
