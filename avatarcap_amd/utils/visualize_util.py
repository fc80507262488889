"""Host-side mirror of the one function of the reference's `utils/visualize_util.py` that feeds the
networks: `render_cano_mesh` (visualize_util.py:11-52), the orthographic front/back raster of the
canonical mesh with its normals as vertex attribute.  The reference drives OpenGL through glfw; here it
is a HIP rasteriser (csrc/raster.hip) and needs no GL context.  Phong / perspective renders for the JPEG
outputs are out of scope."""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib


def render_cano_mesh_device(vertices: torch.Tensor, normals: torch.Tensor, faces: torch.Tensor, mesh_center, size: int = 512):
    """Device tensors in, device tensors out: front, back (size,size,3) f32."""
    v = vertices.contiguous(); n = normals.contiguous(); f = faces.to(torch.int32).contiguous()
    front = torch.empty((size, size, 3), dtype=torch.float32, device=v.device)
    back = torch.empty_like(front)
    _lib.check(_lib.lib().avc_render_cano_maps(_lib.ctx(v.device), _lib.dev_ptr(v, name='vertices'), _lib.dev_ptr(n, name='normals'), v.shape[0],
                                               _lib.dev_ptr(f, torch.int32, 'faces'), f.shape[0], _lib.f3(mesh_center), int(size),
                                               front.data_ptr(), back.data_ptr(), _lib.stream_ptr(v.device)))
    return front, back


def render_cano_mesh(renderer, vertices, normals, faces, mesh_center=np.zeros(3), colors=None):
    """Reference signature (visualize_util.py:11): numpy in, two (H,W,3) float images out.  `renderer` is
    accepted for call compatibility; only its image size is used when it has one."""
    from .. import config
    size = getattr(renderer, 'img_w', 512) if renderer is not None else 512
    dev = config.device
    attr = normals if colors is None else colors
    fr, bk = render_cano_mesh_device(torch.from_numpy(np.ascontiguousarray(vertices, np.float32)).to(dev),
                                     torch.from_numpy(np.ascontiguousarray(attr, np.float32)).to(dev),
                                     torch.from_numpy(np.ascontiguousarray(faces, np.int32)).to(dev), mesh_center, size)
    return fr.cpu().numpy(), bk.cpu().numpy()
