"""Host-side mirror of the parts of the reference's `utils/renderer.py` that feed the networks: the projection
matrices (renderer.py:297-323) and `Renderer` with the 'position' and 'vertex_attribute' shaders (:10-51, :326-451).
The reference drives OpenGL through glfw and reads the frame buffer back; here `render()` is the HIP rasteriser
(csrc/raster.hip, avc_render_mesh), needs no GL context, and can leave the image on the device.  The Phong shaders
only produce the JPEG previews and are out of scope."""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib


def gl_perspective_projection_matrix(fx, fy, cx, cy, img_w, img_h, far=100.0, near=0.1, gl_space=False):
    """Pinhole intrinsics -> clip matrix for a model in the usual camera space (x right, y down, z forward),
    renderer.py:297-312: column = fx x/z + cx, row = fy y/z + cy, w = z."""
    p = np.zeros((4, 4), np.float32)
    p[0, 0], p[0, 2] = 2 * fx / img_w, (2 * cx - img_w) / img_w
    p[1, 1], p[1, 2] = -2 * fy / img_h, (img_h - 2 * cy) / img_h
    p[2, 2], p[2, 3] = (far + near) / (far - near), 2 * near * far / (near - far)
    p[3, 2] = 1.0
    if gl_space:                                     # model given in the OpenGL camera space (y up, looking down -z)
        p = p @ np.diag(np.float32([1, -1, -1, 1]))
    return p


def gl_orthographic_projection_matrix(far=-100.0, near=-0.1):
    """renderer.py:316-323 (model in the OpenGL camera space)."""
    p = np.zeros((4, 4), np.float32)
    p[0, 0] = p[1, 1] = p[3, 3] = 1.0
    p[2, 2], p[2, 3] = 2 / (far - near), -(far + near) / (far - near)
    return p


def render_mesh_device(vertices: torch.Tensor, attrs, faces: torch.Tensor, mvp, width: int, height: int) -> torch.Tensor:
    """(height, width, 4) f32 RGBA on the device: (perspective-correct attribute, 1), background 0.  attrs=None -> positions."""
    v = vertices.contiguous(); f = faces.to(torch.int32).contiguous()
    a = None if attrs is None else attrs.contiguous()
    out = torch.empty((height, width, 4), dtype=torch.float32, device=v.device)
    m = np.ascontiguousarray(mvp, np.float32).reshape(16)
    _lib.check(_lib.lib().avc_render_mesh(_lib.ctx(v.device), _lib.dev_ptr(v, name='vertices'), None if a is None else _lib.dev_ptr(a, name='attributes'),
                                          v.shape[0], _lib.dev_ptr(f, torch.int32, 'faces'), f.shape[0], _lib.f3(m), int(width), int(height),
                                          out.data_ptr(), _lib.stream_ptr(v.device)))
    return out


class Renderer:
    """Reference call surface (renderer.py:326-451): set_model takes the *unindexed* triangle soup the reference
    uploads (`vertices[faces.reshape(-1)]`), render() returns the (H, W, 4) float32 image, row 0 on top."""

    def __init__(self, img_w: int, img_h: int, mvp: np.ndarray = None, shader_name='vertex_attribute', bg_color=(0, 0, 0), window_name=''):
        if shader_name not in ('vertex_attribute', 'position'):
            raise ValueError('Invalid shader name!' if shader_name not in ('phong_geometry', 'phong_color') else
                             'the Phong preview shaders are out of scope of this build (DESIGN.md section 7)')
        self.img_w, self.img_h, self.shader_name = int(img_w), int(img_h), shader_name
        self.mvp = np.identity(4, np.float32) if mvp is None or np.ndim(mvp) != 2 else np.asarray(mvp, np.float32)
        self._v = self._a = self._f = None
        self.vnum = 0

    def set_mvp_mat(self, mvp):
        self.mvp = np.asarray(mvp, np.float32)

    def set_mv_mat(self, mv):          # only the Phong shaders read it
        pass

    def set_model(self, vertices, vertex_attributes=None, vertex_attributes_2=None):
        from .. import config
        dev = config.device
        to = lambda x: x.to(dev, torch.float32) if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
        self._v = to(vertices).reshape(-1, 3)
        self._a = None if vertex_attributes is None or self.shader_name == 'position' else to(vertex_attributes).reshape(-1, 3)
        self.vnum = self._v.shape[0]
        self._f = torch.arange(self.vnum - self.vnum % 3, dtype=torch.int32, device=dev).reshape(-1, 3)

    def render_device(self) -> torch.Tensor:
        if self._v is None:
            raise ValueError('Renderer.render: set_model has not been called')
        return render_mesh_device(self._v, self._a, self._f, self.mvp, self.img_w, self.img_h)

    def render(self) -> np.ndarray:
        return self.render_device().cpu().numpy()
