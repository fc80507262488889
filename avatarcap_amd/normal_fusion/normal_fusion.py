"""Host-side mirror of the reference's `normal_fusion/normal_fusion.py` (step 2 of main.py's frame loop, :405-429):
the image-observed normal map is carried to the canonical pose through the posed avatar mesh and fused with the avatar's
own canonical normal map.  The reference needs an OpenGL context, OpenCV, pytorch3d and 100 autograd iterations driven
from Python; here each function is one or a few calls into libavcap_hip.so (csrc/fusion.hip, csrc/raster.hip) and the
images can stay on the device (`*_device` variants).  Same names, argument meaning and return values as the reference."""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from ..utils.renderer import gl_perspective_projection_matrix, render_mesh_device
from ..utils.visualize_util import render_cano_mesh_device


def _dev(x, dtype=torch.float32):
    from .. import config
    if isinstance(x, torch.Tensor):
        return x.to(config.device, dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(x)).to(config.device, dtype).contiguous()


def canonicalize_normal_map_device(cano_vertices, live_vertices, faces, normal_map, vert_mats, mv, fx, fy, cx, cy, cano_smpl_center, size=512):
    """Device tensors in / out.  normal_map (H,W,3) observed in the camera image; returns the front and back canonical
    maps (size,size,3) of the observed normals (zero where the body is occluded or unobserved)."""
    H, W = int(normal_map.shape[0]), int(normal_map.shape[1])
    mv = np.asarray(mv, np.float32)
    mvp = gl_perspective_projection_matrix(fx, fy, cx, cy, W, H, gl_space=False) @ mv          # normal_fusion.py:17-18
    position_map = render_mesh_device(live_vertices, None, faces, mvp, W, H)                    # :14-20 ('position' shader)
    nv = live_vertices.shape[0]
    proj_n = torch.empty((nv, 3), dtype=torch.float32, device=live_vertices.device)
    _lib.check(_lib.lib().avc_canonicalize_normals(_lib.ctx(proj_n.device), _lib.dev_ptr(live_vertices.contiguous(), name='live_vertices'),
                                                   _lib.dev_ptr(vert_mats.contiguous(), name='vert_mats'), nv, position_map.data_ptr(),
                                                   _lib.dev_ptr(normal_map.contiguous(), name='normal_map'), H, W, _lib.f3(mv.reshape(16)),
                                                   float(fx), float(fy), float(cx), float(cy), proj_n.data_ptr(), _lib.stream_ptr(proj_n.device)))   # :27-62
    return render_cano_mesh_device(cano_vertices, proj_n, faces, cano_smpl_center, size)        # :63


def canonicalize_normal_map(pos_renderer, attri_renderer, cano_vertices, live_vertices, faces, normal_map, vert_mats, mv, fx, fy, cx, cy,
                            cano_smpl_center):
    """Reference signature (normal_fusion.py:12): numpy (vert_mats: tensor) in, two (512,512,3) numpy images out.  The two
    renderer arguments are accepted for call compatibility; only the attribute renderer's image size is read."""
    size = getattr(attri_renderer, 'img_w', 512) if attri_renderer is not None else 512
    fr, bk = canonicalize_normal_map_device(_dev(cano_vertices), _dev(live_vertices), _dev(faces, torch.int32), _dev(normal_map), _dev(vert_mats),
                                            mv, fx, fy, cx, cy, cano_smpl_center, size)
    return fr.cpu().numpy(), bk.cpu().numpy()


def merge_normal_images_device(src_img: torch.Tensor, tar_img: torch.Tensor, iter_num: int, neck_xy) -> torch.Tensor:
    H, W = int(src_img.shape[0]), int(src_img.shape[1])
    if tuple(tar_img.shape) != (H, W, 3) or src_img.shape[2] != 3:
        raise ValueError('merge_normal_images: src_img and tar_img must both be (H, W, 3)')
    src, tar = src_img.contiguous(), tar_img.contiguous()
    out = torch.empty_like(src)
    _lib.check(_lib.lib().avc_merge_normal_images(_lib.ctx(src.device), _lib.dev_ptr(src, name='src_img'), _lib.dev_ptr(tar, name='tar_img'), H, W,
                                                  int(iter_num), int(neck_xy[0]), int(neck_xy[1]), out.data_ptr(), _lib.stream_ptr(src.device)))
    return out


def merge_normal_images(src_img, tar_img, iter_num, neck_xy):
    """Canonical normal fusion using 2D rotation grids (normal_fusion.py:89-155): numpy (H,W,3) in, numpy out."""
    return merge_normal_images_device(_dev(src_img), _dev(tar_img), iter_num, neck_xy).cpu().numpy()


def merge_normal_images_cover_device(src_img: torch.Tensor, tar_img: torch.Tensor) -> torch.Tensor:
    src, tar = src_img.contiguous(), tar_img.contiguous()
    out = torch.empty_like(src)
    _lib.check(_lib.lib().avc_merge_normal_images_cover(_lib.ctx(src.device), _lib.dev_ptr(src, name='src_img'), _lib.dev_ptr(tar, name='tar_img'),
                                                        src.numel() // 3, out.data_ptr(), _lib.stream_ptr(src.device)))
    return out


def merge_normal_images_cover(src_img, tar_img):
    """Directly cover the avatar normal with the image-observed one (normal_fusion.py:158-167)."""
    return merge_normal_images_cover_device(_dev(src_img), _dev(tar_img)).cpu().numpy()
