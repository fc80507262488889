#!/usr/bin/env python3
"""Golden images from a REAL OpenGL implementation for the canonical normal maps and the position render:
the reference's own `visualize_util.render_cano_mesh` (utils/visualize_util.py:11-52) and `Renderer` call sequence
(utils/renderer.py:326-451) run against Mesa's llvmpipe, headless, through tools/gl/mesa_raster.c -- build container only
(libgl1-mesa-dri + mesa-common-dev are installed there; glfw / PyOpenGL are not, so the `Renderer` class is a stand-in that replays
the same GL calls from C: same GLSL sources, same state, same draw).  cv2's cvtColor(RGBA2RGB) / flip / Rodrigues are restated.

    python tests/golden/make_golden_gl.py        ->  tests/golden/gl_golden.npz
Stored per scene: the full-resolution coverage masks (bit-packed), every 4th pixel of the float maps, and the input meshes' seeds
(the test rebuilds them).  tests/test_raster.py holds oracle/raster_oracle.c -- and through it the HIP kernels -- to these.
"""
import math
import os
import struct
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
REF = '/root/reference'
BIN = os.path.join(ROOT, 'tools', 'gl', 'mesa_raster')


class MesaRenderer:
    """The methods render_cano_mesh / canonicalize_normal_map call on a Renderer (utils/renderer.py:389-451)."""

    def __init__(self, img_w, img_h, shader_name='vertex_attribute'):
        # the GLSL sources are the reference's own, read from its module at run time (utils/renderer.py:10-60, selected as :337-343 does)
        from utils import renderer as ref_renderer
        self.img_w, self.img_h = img_w, img_h
        self.vs = getattr(ref_renderer, 'vs_' + shader_name).strip().encode()
        self.fs = getattr(ref_renderer, 'fs_' + shader_name).strip().encode()

    def set_model(self, vertices, vertex_attributes=None, vertex_attributes_2=None):
        self.v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        self.a = np.zeros_like(self.v) if vertex_attributes is None else np.ascontiguousarray(vertex_attributes, np.float32).reshape(-1, 3)

    def set_mvp_mat(self, mvp):
        self.mvp = np.ascontiguousarray(mvp, np.float32)

    def set_mv_mat(self, mv):
        pass

    def render(self):
        td = tempfile.mkdtemp()
        with open(td + '/in.bin', 'wb') as f:
            f.write(struct.pack('<5i', self.img_w, self.img_h, self.v.shape[0], len(self.vs), len(self.fs)) + self.mvp.tobytes() + self.v.tobytes() + self.a.tobytes()
                    + self.vs + self.fs)
        r = subprocess.run([BIN, td + '/in.bin', td + '/out.bin'], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        self.gl_info = r.stderr.strip()
        data = np.fromfile(td + '/out.bin', np.float32).reshape(self.img_h, self.img_w, 4)
        return data[::-1, :]                      # renderer.py:449


def install_bin():
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(BIN + '.c'):
        subprocess.check_call(['gcc', '-O1', BIN + '.c', '-o', BIN, '-ldl'])


def install():
    install_bin()
    cv = types.ModuleType('cv2')
    cv.COLOR_RGBA2RGB = 2
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
    gl, glgl, glfw = types.ModuleType('OpenGL'), types.ModuleType('OpenGL.GL'), types.ModuleType('glfw')
    glgl.shaders = types.ModuleType('OpenGL.GL.shaders')
    gl.GL = glgl
    sys.modules.update({'cv2': cv, 'OpenGL': gl, 'OpenGL.GL': glgl, 'OpenGL.GL.shaders': glgl.shaders, 'glfw': glfw})
    sys.path.insert(0, REF)
    import torch
    import config
    config.device = torch.device('cpu')


def main():
    install()
    import gl_scenes as sc
    from utils import visualize_util
    from utils.renderer import gl_perspective_projection_matrix
    out = {}
    info = ''
    for name in sc.CANO_SCENES:
        v, f, n, center, size = sc.cano_scene(name)
        r = MesaRenderer(size, size, 'vertex_attribute')
        front, back = visualize_util.render_cano_mesh(r, v, n, f, center)
        info = r.gl_info
        for tag, img in (('front', front), ('back', back)):
            out[f'{name}_{tag}_mask'] = np.packbits(np.linalg.norm(img, axis=-1) > 0)
            out[f'{name}_{tag}_lattice'] = np.ascontiguousarray(img[::4, ::4]).astype(np.float32)
        print(name, 'coverage', float((np.linalg.norm(front, axis=-1) > 0).mean()), float((np.linalg.norm(back, axis=-1) > 0).mean()))
    # the 'position' render of canonicalize_normal_map (normal_fusion.py:14-20): perspective camera, non-square image
    v, f, mv, fx, fy, cx, cy, W, H = sc.position_scene()
    r = MesaRenderer(W, H, 'position')
    r.set_model(v[f.reshape(-1)].astype(np.float32))
    r.set_mvp_mat(np.dot(gl_perspective_projection_matrix(fx, fy, cx, cy, W, H), mv))
    pos = r.render()
    out['position_mask'] = np.packbits(pos[..., 3] > 0)
    out['position_lattice'] = np.ascontiguousarray(pos[::3, ::3]).astype(np.float32)
    print('position coverage', float((pos[..., 3] > 0).mean()))
    out['gl_info'] = np.array(info)
    path = os.path.join(HERE, 'gl_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes;', info)


if __name__ == '__main__':
    main()
