"""In-tree build of libavcap_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

`python -m avatarcap_amd.build` or `__graft_entry__.build()`.  The .so stays in the package
directory so it travels with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'csrc', '_obj')
LIB = os.path.join(HERE, 'libavcap_hip.so')
SOURCES = ['fused_mlp.hip', 'misc.hip', 'mesh.hip', 'raster.hip', 'fusion.hip', 'knn_lbs.hip', 'pack.cpp', 'capi.cpp']
HEADERS = ['avcap_internal.h', 'mlp_layout.h', 'mc_tables.h', os.path.join('..', '..', 'include', 'avcap.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-Wno-unused-result']
# mesh / KNN kernels promise bit-exact agreement with the C oracle (built with -ffp-contract=off):
# HIP's __fmul_rn/__fadd_rn are plain operators, so contraction has to be disabled per file.
EXTRA = {'mesh.hip': ['-ffp-contract=off'], 'knn_lbs.hip': ['-ffp-contract=off'], 'raster.hip': ['-ffp-contract=off']}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src + '.o')
        if force or _stale(op, [sp] + hdrs):
            cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + (['-x', 'hip'] if src.endswith('.cpp') else []) + ['-c', sp, '-o', op]
            jobs.append((src, cmd))

    def run(job):
        src, cmd = job
        if verbose:
            print('[avatarcap_amd.build]', ' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s + '.o') for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
        if verbose:
            print('[avatarcap_amd.build]', ' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
