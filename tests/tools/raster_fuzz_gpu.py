"""Fuzz of the HIP rasterisers (csrc/raster.hip: the orthographic canonical front / back views and the general MVP view) against the C oracle
(oracle/raster_oracle.c, itself pinned on a real OpenGL implementation, tests/golden/gl_golden.npz): random triangle soups -- blobs, slivers, zero-area and
repeated triangles, coincident depths (the same triangle twice, shared vertices), triangles partly or wholly off screen, behind the camera, a few pixels or the
whole image wide --, random image sizes, random pinhole cameras.  Every pixel bit for bit.
    python tests/tools/raster_fuzz_gpu.py [cases] [seed]          (needs an MI355X; a checker script, not part of the product path)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from avatarcap_amd import config                                        # noqa: E402
config.cfg = config.default_cfg(); config.device = torch.device('cuda')
from avatarcap_amd.utils.renderer import render_mesh_device             # noqa: E402
from avatarcap_amd.utils.visualize_util import render_cano_mesh_device  # noqa: E402
from oracle import raster                                               # noqa: E402


def soup(rs, kind):
    nt = int(rs.choice([1, 2, 7, 50, 400, 3000]))
    if kind == 'blob':                                                   # small triangles around random centres
        c = rs.uniform(-0.8, 0.8, (nt, 1, 3))
        tri = c + rs.normal(0, 10.0 ** rs.uniform(-2.5, -0.5), (nt, 3, 3))
    elif kind == 'big':
        tri = rs.uniform(-1.5, 1.5, (nt, 3, 3))
    elif kind == 'sliver':
        a = rs.uniform(-1, 1, (nt, 1, 3)); d = rs.normal(0, 0.5, (nt, 1, 3))
        tri = a + d * rs.uniform(0, 1, (nt, 3, 1)) + rs.normal(0, 1e-4, (nt, 3, 3))
    elif kind == 'lattice':                                              # vertices on a coarse lattice: edges through pixel centres, coincident depths
        tri = rs.randint(-8, 9, (nt, 3, 3)) / 8.0
    else:
        raise ValueError(kind)
    v = tri.reshape(-1, 3).astype(np.float32)
    f = np.arange(3 * nt, dtype=np.int32).reshape(nt, 3)
    if rs.rand() < 0.3 and nt > 2:                                       # repeated and zero-area triangles
        f[rs.randint(0, nt)] = f[rs.randint(0, nt)]
        f[rs.randint(0, nt), 2] = f[rs.randint(0, nt), 1]
    if rs.rand() < 0.3:                                                  # shared vertices
        f = rs.randint(0, v.shape[0], f.shape).astype(np.int32)
    a = rs.randn(v.shape[0], 3).astype(np.float32)
    return v, f, a


def pinhole(W, H, fo, tz):
    from avatarcap_amd.utils.renderer import gl_perspective_projection_matrix
    mv = np.eye(4, dtype=np.float32); mv[2, 3] = tz                      # camera space: x right, y down, z forward; the soup sits tz in front of the camera
    return (gl_perspective_projection_matrix(fo, fo, W / 2.0, H / 2.0, W, H) @ mv).astype(np.float32)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 515)
    bad, t0, px, cov = 0, time.time(), 0, [0, 0]
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()      # noqa: E731
    for k in range(cases):
        kind = ['blob', 'big', 'sliver', 'lattice'][rs.randint(0, 4)]
        v, f, a = soup(rs, kind)
        why = []
        size = int(rs.choice([16, 33, 64, 100, 256]))
        c = rs.uniform(-0.2, 0.2, 3).astype(np.float32)
        ofr, obk = raster.render_cano_mesh(v, a, f, c, size)
        fr, bk = render_cano_mesh_device(t(v), t(a), t(f), c, size)
        cov[0] += int((ofr != 0).any(-1).sum()) + int((obk != 0).any(-1).sum())
        if not (np.array_equal(fr.cpu().numpy(), ofr) and np.array_equal(bk.cpu().numpy(), obk)):
            why.append(f'canonical views {size}: {int((fr.cpu().numpy() != ofr).any(-1).sum())} + {int((bk.cpu().numpy() != obk).any(-1).sum())} pixels')
        W, H = int(rs.choice([17, 64, 200])), int(rs.choice([16, 48, 150]))
        mvp = pinhole(W, H, float(rs.uniform(10, 300)), float(rs.uniform(-0.5, 3.0)))      # (tz <= 0: the camera inside / behind the soup)
        for attrs in (None, a):
            o = raster.render_mesh(v, attrs, f, mvp, W, H)
            d = render_mesh_device(t(v), None if attrs is None else t(attrs), t(f), mvp, W, H).cpu().numpy()
            cov[1] += int((o != 0).any(-1).sum())
            if not np.array_equal(d, o):
                why.append(f'MVP view {W}x{H} attrs {attrs is not None}: {int((d != o).any(-1).sum())} pixels')
        px += 2 * size * size + 2 * W * H
        if why:
            bad += 1
            print(f'case {k}: {kind} {f.shape[0]} triangles: {why}')
    print(f'{cases} cases, {px} pixels ({cov[0]} covered in the canonical views, {cov[1]} in the MVP views), {bad} mismatches, {time.time() - t0:.0f} s')
    return 1 if bad else 0


if __name__ == '__main__':
    raise SystemExit(main())
