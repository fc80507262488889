"""Fuzz of the HIP marching cubes (csrc/mesh.hip, through recon_util.recon_mesh) against the C oracle (oracle/mc_oracle.c, itself pinned on the real
scikit-image call by tests/tools/mc_fuzz.py and tests/golden/mc_golden.npz): random grid shapes -- thin, long, odd, non-multiples of every tile edge --,
random iso values, white noise, smooth fields, and volumes of small integers / half-integers, where every face and interior test of Lewiner's case
analysis ties and many samples EQUAL the iso value.  Vertices (float32 bits), faces, numbering and order must be identical.
    python tests/tools/mc_fuzz_gpu.py [cases] [seed]          (needs an MI355X; a checker script, not part of the product path)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from avatarcap_amd import config, synthetic as syn                     # noqa: E402
from avatarcap_amd.utils import recon_util                              # noqa: E402
from oracle import avatarcap_oracle as orc                              # noqa: E402

config.cfg = config.default_cfg(); config.device = torch.device('cuda')


def volume(rs, res, kind):
    if kind == 'noise':
        return rs.randn(*res).astype(np.float32)
    if kind == 'ints':
        return rs.randint(-2, 3, res).astype(np.float32)
    if kind == 'halves':
        return (rs.randint(-3, 4, res) * 0.5).astype(np.float32)
    if kind == 'sparse':                                                # mostly empty: a few crossed cells per tile, many tiles without any
        v = np.full(res, -1.0, np.float32)
        m = rs.rand(*res) < 0.01
        v[m] = rs.rand(int(m.sum())).astype(np.float32) + 0.1
        return v
    g = [np.linspace(-1, 1, r, dtype=np.float32) for r in res]
    x, y, z = np.meshgrid(*g, indexing='ij')
    v = np.zeros(res, np.float32)
    for _ in range(rs.randint(1, 6)):
        c = rs.uniform(-0.8, 0.8, 3); r = rs.uniform(0.1, 0.7)
        v = np.maximum(v, (r - np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2)).astype(np.float32))
    return v - 0.05


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 20260)
    bad, empty, nv, t0 = 0, 0, 0, time.time()
    for k in range(cases):
        shape_kind = rs.randint(0, 4)
        if shape_kind == 0:
            res = tuple(int(a) for a in rs.randint(2, 40, 3))
        elif shape_kind == 1:                                           # one long axis (the walking classify pass; rows longer than a tile)
            res = [int(a) for a in rs.randint(2, 12, 3)]; res[rs.randint(0, 3)] = int(rs.randint(100, 1200)); res = tuple(res)
        elif shape_kind == 2:                                           # around the tile edges
            res = tuple(int(rs.choice([7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65])) for _ in range(3))
        else:
            res = tuple(int(a) for a in rs.randint(2, 6, 3))
        kind = ['noise', 'ints', 'halves', 'sparse', 'blobs'][rs.randint(0, 5)]
        vol = volume(rs, res, kind)
        iso = float(rs.choice([0.0, 0.5, -0.5, 0.1, 1.0])) if kind in ('ints', 'halves') else float(rs.uniform(-0.3, 0.3))
        want = got = None
        try:
            want = orc.recon_mesh(vol, list(res), syn.CANO_BOUNDS, iso)
        except Exception as e:       # noqa: BLE001 -- the library's own errors (iso outside the range, no surface): the HIP path must raise the same type
            want = e
        try:
            got = recon_util.recon_mesh(torch.from_numpy(vol).cuda(), list(res), syn.CANO_BOUNDS, iso_value=iso)
        except Exception as e:       # noqa: BLE001
            got = e
        if isinstance(want, Exception) or isinstance(got, Exception):
            if type(want) is not type(got):
                bad += 1
                print(f'case {k} res {res} {kind} iso {iso}: oracle {type(want).__name__}, HIP {type(got).__name__}: {got if isinstance(got, Exception) else want}')
            else:
                empty += 1
            continue
        ok = got[0].shape == want[0].shape and got[1].shape == want[1].shape and np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)) and np.array_equal(got[1], want[1])
        nv += got[0].shape[0]
        if not ok:
            bad += 1
            print(f'case {k} res {res} {kind} iso {iso}: MISMATCH verts {got[0].shape} vs {want[0].shape}, faces {got[1].shape} vs {want[1].shape}')
    print(f'{cases} cases, {empty} raised the same error on both sides, {nv} vertices compared, {bad} mismatches, {time.time() - t0:.0f} s')
    return 1 if bad else 0


if __name__ == '__main__':
    raise SystemExit(main())
