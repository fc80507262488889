"""Marching-cubes oracle (oracle/mc_oracle.c): parity with scikit-image is UNPINNED (the library is not
vendored and not installable), so the restated algorithm is validated through invariants, and the
generated GPU case tables are checked against it cell by cell.  CPU only."""
import importlib.util
import os
from collections import Counter

import numpy as np
import pytest

from oracle import mc

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _grid(res):
    g = [np.linspace(-1, 1, r, dtype=np.float32) for r in res]
    return np.meshgrid(*g, indexing='ij')


def _edges(f):
    e = Counter()
    for a, b, c in f:
        for p, q in ((a, b), (b, c), (c, a)):
            e[(int(p), int(q))] += 1
    return e


def _is_closed_oriented_manifold(f):
    e = _edges(f)
    return all(v == 1 and e.get((k[1], k[0]), 0) == 1 for k, v in e.items())


def test_sphere_invariants():
    res = (40, 40, 40)
    x, y, z = _grid(res)
    R = 0.63
    vol = (R - np.sqrt(x * x + y * y + z * z)).astype(np.float32)          # positive inside, like the reference's 'sdf'
    h = 2.0 / (res[0] - 1)
    v, f = mc.marching_cubes(vol, 0.0, [h, h, h])
    assert _is_closed_oriented_manifold(f)
    assert len(v) - len(_edges(f)) // 2 + len(f) == 2                         # Euler characteristic of a sphere
    p = v - 1.0                                                               # index*h -> [-1,1]
    assert np.abs(np.linalg.norm(p, axis=1) - R).max() < 0.6 * h * h / R + 1e-4   # linear root of a smooth field
    tri = p[f]
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    area = 0.5 * np.linalg.norm(n, axis=1).sum()
    assert abs(area / (4 * np.pi * R * R) - 1) < 0.01
    vol6 = np.einsum('ij,ij->i', tri[:, 0], np.cross(tri[:, 1], tri[:, 2])).sum() / 6.0
    # right-hand normals point towards HIGHER values (inside here) => negative signed volume
    assert vol6 < 0 and abs(-vol6 / (4 / 3 * np.pi * R ** 3) - 1) < 0.01
    assert np.all((n * (-tri.mean(1))).sum(1) > 0)


def test_torus_genus_and_anisotropic_spacing():
    res = (48, 40, 24)
    x, y, z = _grid(res)
    vol = (0.22 - np.sqrt((np.sqrt(x * x + y * y) - 0.55) ** 2 + z * z)).astype(np.float32)
    sp = np.array([0.01, 0.02, 0.03], np.float32)
    v, f = mc.marching_cubes(vol, 0.0, sp)
    assert _is_closed_oriented_manifold(f)
    assert len(v) - len(_edges(f)) // 2 + len(f) == 0                         # genus 1
    idx = v / sp
    frac = np.abs(idx - np.round(idx))
    assert np.all((frac > 1e-4).sum(1) <= 1), 'every vertex lies on a grid edge'


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_noise_is_watertight(seed):
    """White noise hits every ambiguous configuration; the asymptotic-decider face rule must keep
    neighbouring cells consistent (no cracks) inside the volume."""
    vol = np.random.RandomState(seed).randn(14, 15, 16).astype(np.float32)
    vol = np.pad(vol, 1, constant_values=-5.0)                                # close the surface at the border
    v, f = mc.marching_cubes(vol, 0.0, [1, 1, 1])
    e = _edges(f)
    assert all(e.get((k[1], k[0]), 0) == val for k, val in e.items())         # every edge matched by its opposite
    v2, f2 = mc.marching_cubes(vol, 0.0, [1, 1, 1])
    assert np.array_equal(v, v2) and np.array_equal(f, f2)                    # deterministic


def test_edges_are_manifold():
    """Loops are triangulated so that chords inside a cube face are avoided (a fan from a fixed vertex puts ~0.7 % of the
    edges of a white-noise field on four triangles): a smooth field must give none, white noise at most a few per million
    (the loops of 8, 9 and 12 vertices that cannot avoid such a chord, meeting a neighbour that draws the same one)."""
    from scipy.ndimage import gaussian_filter
    rs = np.random.RandomState(5)
    noise = rs.randn(56, 56, 56).astype(np.float32)
    for vol, allowed in ((noise, 2e-5), (gaussian_filter(noise, 0.7).astype(np.float32), 0.0)):
        v, f = mc.marching_cubes(vol, 0.0, [1, 1, 1])
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        key = np.minimum(e[:, 0], e[:, 1]).astype(np.int64) * v.shape[0] + np.maximum(e[:, 0], e[:, 1])
        cnt = np.unique(key, return_counts=True)[1]
        assert cnt.max() <= 4 and (cnt > 2).sum() <= allowed * cnt.size, np.unique(cnt, return_counts=True)


def test_vertex_is_linear_root_and_canonical_order():
    vol = np.random.RandomState(3).randn(6, 5, 7).astype(np.float32)
    iso = 0.2
    v, f = mc.marching_cubes(vol, iso, [1, 1, 1])
    k = 0
    for li in range(vol.size):
        x, y, z = np.unravel_index(li, vol.shape)
        for ax in range(3):
            q = [x, y, z]; q[ax] += 1
            if q[ax] >= vol.shape[ax]:
                continue
            a, b = np.float32(vol[x, y, z] - np.float32(iso)), np.float32(vol[tuple(q)] - np.float32(iso))
            if (a > 0) != (b > 0):
                t = np.float32(np.float32(0) - a) / np.float32(b - a)
                exp = np.array([x, y, z], np.float32); exp[ax] += t
                assert np.allclose(v[k], exp, atol=1e-6), (k, v[k], exp)
                k += 1
    assert k == len(v)
    assert f.min() >= 0 and f.max() < len(v)


def test_empty_and_degenerate():
    v, f = mc.marching_cubes(np.zeros((4, 4, 4), np.float32), 0.0, [1, 1, 1])   # == iso everywhere: nothing is "above"
    assert len(v) == 0 and len(f) == 0
    vol = np.zeros((3, 3, 3), np.float32); vol[1, 1, 1] = 1.0                   # vertices collapse onto corners? no: t = 1 -> on the neighbour
    v, f = mc.marching_cubes(vol, 0.0, [1, 1, 1])
    assert len(v) == 6 and len(f) == 8 and _is_closed_oriented_manifold(f)


def test_generated_tables_match_oracle_cells():
    spec = importlib.util.spec_from_file_location('g', os.path.join(ROOT, 'tools', 'gen_mc_tables.py'))
    g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
    info, rows = g.build()
    header = open(os.path.join(ROOT, 'avatarcap_amd', 'csrc', 'mc_tables.h')).read()
    assert f'N_ROWS = {len(rows)}' in header, 'mc_tables.h is stale: run tools/gen_mc_tables.py'
    assert all(('0x%08xu' % w) in header for w in rows[-1])
    rs = np.random.RandomState(0)
    for cfg in range(256):
        amb = g.ambiguous_faces(cfg)
        for _ in range(12):
            val = np.where([(cfg >> c) & 1 for c in range(8)], rs.uniform(0.1, 1, 8), -rs.uniform(0.1, 1, 8)).astype(np.float32)
            variant = 0
            for i, fc in enumerate(amb):
                a, b, c, d = (val[k] for k in g.FACE_CORNERS[fc])
                p, q = np.float32(a * c), np.float32(b * d)
                variant |= int((p > q) if a > 0 else (q > p)) << i
            row = rows[(info[cfg][0] & 0xffff) + variant]
            by = [(row[w >> 2] >> (8 * (w & 3))) & 0xff for w in range(16)]
            nib = [n for x in by[1:] for n in (x & 15, x >> 4)]
            tri = np.array(nib[:3 * by[0]]).reshape(-1, 3)
            assert np.array_equal(tri, mc.cell_triangles(val)), (cfg, variant)
