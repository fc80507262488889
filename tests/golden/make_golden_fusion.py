#!/usr/bin/env python3
"""Golden vectors for the canonical normal fusion, produced by RUNNING THE REFERENCE'S OWN normal_fusion.py
(/root/reference/normal_fusion/normal_fusion.py) -- build container only.

The module imports OpenCV, pytorch3d and an OpenGL renderer, none of which exists offline.  Stand-ins, each a few lines
and each restating a published, fixed algorithm (they are NOT the code under test -- the reference's torch code is):
  cv2.getStructuringElement / erode / distanceTransform / cvtColor / flip / Rodrigues : definitions of the OpenCV ops
  pytorch3d.transforms.axis_angle_to_matrix : the library's published quaternion route, written with torch ops
  glfw / OpenGL : empty modules (utils/renderer.py only needs to import); the two Renderer objects the function receives
                  are replaced by a class with the same four methods that replays the Renderer's GL calls on Mesa llvmpipe
                  (tests/golden/make_golden_gl.py: MesaRenderer, tools/gl/mesa_raster.c) -- a real OpenGL implementation
What the vectors therefore pin: merge_normal_images' optimisation loop as the reference's autograd + torch.optim.Adam run
it (resize_img, get_neighbor_images, the data / smoothness terms, the two-phase schedule, the distance-transform blend,
the face rectangle's slice semantics), and canonicalize_normal_map's per-vertex chain plus the way render_cano_mesh
composes its front / back matrices.

    python tests/golden/make_golden_fusion.py        ->  tests/golden/fusion_golden.npz
"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, HERE)

import make_golden as mg                                     # noqa: E402
from oracle import normal_fusion_oracle as nfo, raster      # noqa: E402


def install_fusion_stubs():
    config, _ = mg.install_stubs()
    cv = sys.modules['cv2']
    cv.MORPH_RECT, cv.DIST_L1, cv.COLOR_RGBA2RGB = 0, 1, 2
    cv.getStructuringElement = lambda shape, ksize: np.ones(ksize, np.uint8)
    cv.erode = lambda img, kernel, iterations=1: nfo.erode3x3(img, iterations)
    cv.distanceTransform = lambda img, dist, mask: nfo.distance_transform_l1(img)
    cv.cvtColor = lambda img, code: np.ascontiguousarray(img[..., :3])
    cv.flip = lambda img, code: np.ascontiguousarray(img[:, ::-1]) if code == 1 else np.ascontiguousarray(img[::-1])

    def rodrigues(r):
        r = np.asarray(r, np.float64).reshape(3)
        th = np.linalg.norm(r)
        if th < 1e-12:
            return np.eye(3), None
        k = r / th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K, None
    cv.Rodrigues = rodrigues
    import test_normal_fusion as tnf
    tr = types.ModuleType('pytorch3d.transforms')
    tr.axis_angle_to_matrix = tnf._aa2mat_torch
    sys.modules['pytorch3d'].transforms = tr
    sys.modules['pytorch3d.transforms'] = tr
    gl, glgl, glfw = types.ModuleType('OpenGL'), types.ModuleType('OpenGL.GL'), types.ModuleType('glfw')
    glgl.shaders = types.ModuleType('OpenGL.GL.shaders')
    gl.GL = glgl
    sys.modules.update({'OpenGL': gl, 'OpenGL.GL': glgl, 'OpenGL.GL.shaders': glgl.shaders, 'glfw': glfw})
    return config


class OracleRenderer:
    """The four methods canonicalize_normal_map / render_cano_mesh call on a Renderer; rasterisation by oracle/raster.py.
    (Kept for machines without Mesa; the committed goldens are made with make_golden_gl.MesaRenderer = real OpenGL, see main().)"""

    def __init__(self, w, h, shader):
        self.img_w, self.img_h, self.shader = w, h, shader

    def set_model(self, vertices, attrs=None, attrs2=None):
        self.v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        self.a = None if attrs is None or self.shader == 'position' else np.ascontiguousarray(attrs, np.float32).reshape(-1, 3)
        self.f = np.arange(self.v.shape[0], dtype=np.int32).reshape(-1, 3)

    def set_mvp_mat(self, mvp):
        self.mvp = np.asarray(mvp, np.float32)

    def set_mv_mat(self, mv):
        pass

    def render(self):
        return raster.render_mesh(self.v, self.a, self.f, self.mvp, self.img_w, self.img_h)


def main():
    config = install_fusion_stubs()
    torch.manual_seed(0)
    torch.set_grad_enabled(True)
    from normal_fusion.normal_fusion import merge_normal_images, merge_normal_images_cover, canonicalize_normal_map
    import test_normal_fusion as tnf
    out = {}
    # ---- G15: merge_normal_images, the reference's full 100 iterations on 512 x 512 maps ------------------------------------
    _, src, tar = tnf._case(5, H=512)
    src, tar = src.astype(np.float32), tar.astype(np.float32)
    for tag, iters, neck in (('a', 100, (-256, 150)), ('b', 10, (300, 200))):
        m = merge_normal_images(src.copy(), tar.copy(), iter_num=iters, neck_xy=neck)
        out[f'G15_{tag}_lattice'] = m[::3, ::3].astype(np.float32)            # every third pixel (the inputs are regenerated by the test)
        out[f'G15_{tag}_checksum'] = np.float64([m.astype(np.float64).sum(), np.abs(m.astype(np.float64)).sum()])
        print('G15', tag, 'done', flush=True)
    out['G15_cover_lattice'] = merge_normal_images_cover(src.copy(), tar.copy())[::3, ::3]
    # ---- G16: canonicalize_normal_map on the synthetic scene of the tests ---------------------------------------------------
    s = tnf._scene()
    # the two Renderer objects of canonicalize_normal_map: REAL OpenGL (Mesa llvmpipe through tools/gl/mesa_raster.c) where the build
    # container has it -- the golden then carries the reference's function on a real GL implementation --, the oracle rasteriser otherwise
    try:
        import make_golden_gl as mgl
        mgl.install_bin() if hasattr(mgl, 'install_bin') else None
        pos_r, att_r = mgl.MesaRenderer(s['W'], s['H'], 'position'), mgl.MesaRenderer(512, 512, 'vertex_attribute')
        pos_r.set_model(np.zeros((3, 3), np.float32)); pos_r.set_mvp_mat(np.eye(4, dtype=np.float32)); pos_r.render()       # probe
        out['G16_renderer'] = np.array(pos_r.gl_info)
    except Exception as e:        # noqa: BLE001
        print('no Mesa harness (%r): the oracle rasteriser stands in for the two Renderers' % (e,))
        pos_r, att_r = OracleRenderer(s['W'], s['H'], 'position'), OracleRenderer(512, 512, 'vertex_attribute')
        out['G16_renderer'] = np.array('oracle/raster.py')
    c = np.float32([0.01, -0.02, 0.0])
    with torch.no_grad():
        fr, bk = canonicalize_normal_map(pos_r, att_r, s['v'], s['live'], s['f'], s['obs'], torch.from_numpy(s['M']), s['mv'], s['fx'], s['fy'],
                                         s['cx'], s['cy'], c)
    out['G16_front_lattice'], out['G16_back_lattice'] = fr[::3, ::3].astype(np.float32), bk[::3, ::3].astype(np.float32)
    out['G16_cover'] = np.float64([(np.linalg.norm(fr, axis=-1) > 0).mean(), (np.linalg.norm(bk, axis=-1) > 0).mean()])
    path = os.path.join(HERE, 'fusion_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024), {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
