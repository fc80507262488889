"""Host-side mirror of the reference's `utils/recon_util.py` on libavcap_hip.so.

`recon_mesh` keeps the reference signature, host-numpy return and error behaviour (recon_util.py:51-70: the exceptions of
`skimage.measure.marching_cubes` propagate); `recon_mesh_device` is the same computation returning device tensors (no host
round trip), which is what the frame loop uses.  The marching cubes is scikit-image's Lewiner algorithm restated on the device:
vertices, faces and their numbering are those of the library call.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib

_cap = {}   # per-device growing output capacity


def recon_mesh_device(occ_volume: torch.Tensor, volume_res, bounds, iso_value=0.5, with_normals=True):
    """-> vertices (V,3) f32, faces (F,3) i32, normals (V,3) f32 device tensors."""
    res = [int(r) for r in volume_res]
    vol = occ_volume.reshape(res).contiguous()
    dev = vol.device
    ctx = _lib.ctx(dev)
    b = np.asarray(bounds.detach().cpu() if isinstance(bounds, torch.Tensor) else bounds, np.float32).reshape(6)
    res_c = (C.c_int32 * 3)(*res)
    b_c = (C.c_float * 6)(*b.tolist())
    counts = (C.c_int64 * 2)()
    cap_v, cap_f = _cap.get(dev.index, (1 << 18, 1 << 19))
    while True:
        verts = torch.empty((cap_v, 3), dtype=torch.float32, device=dev)
        normals = torch.empty((cap_v, 3), dtype=torch.float32, device=dev) if with_normals else None
        faces = torch.empty((cap_f, 3), dtype=torch.int32, device=dev)
        rc = _lib.lib().avc_recon_mesh(ctx, _lib.dev_ptr(vol, name='occ_volume'), res_c, b_c, float(iso_value), verts.data_ptr(),
                                       normals.data_ptr() if with_normals else None, faces.data_ptr(), cap_v, cap_f, counts,
                                       _lib.stream_ptr(dev))
        if rc == _lib.AVC_ERR_CAPACITY:
            cap_v = max(cap_v, int(counts[0] * 1.25) + 1024)
            cap_f = max(cap_f, int(counts[1] * 1.25) + 1024)
            continue
        _lib.check(rc)
        break
    _cap[dev.index] = (cap_v, cap_f)
    V, Fn = int(counts[0]), int(counts[1])
    return verts[:V], faces[:Fn], (normals[:V] if with_normals else None)


def recon_mesh(occ_volume, volume_res, bounds, iso_value=0.5):
    """Reference signature (recon_util.py:51): occ_volume torch.Tensor (device), volume_res list,
    bounds numpy (2,3) -> vertices ndarray (V,3) f32, faces ndarray (F,3) i32, normals ndarray (V,3) f32."""
    v, f, n = recon_mesh_device(occ_volume, volume_res, bounds, iso_value)
    if v.shape[0] == 0:
        # what the library raises at recon_util.py:64 (messages: tests/golden/mc_golden.npz err_*)
        lo, hi = float(occ_volume.min()), float(occ_volume.max())
        if float(iso_value) < lo or float(iso_value) > hi:
            raise ValueError('Surface level must be within volume data range.')
        raise RuntimeError('No surface found at the given iso value.')
    return v.cpu().numpy(), f.cpu().numpy(), n.cpu().numpy()
