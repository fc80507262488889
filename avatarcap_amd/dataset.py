"""Test-mode data harness.

`SyntheticTestDataset` mirrors what `AvatarCapDataset.__init__/__getitem__` prepare in test mode
(dataset/avatarcap_dataset.py:27-125, 181-308) -- the canonical grid, the valid-band flag, the
inside/outside fill of the skipped points and the per-frame item dict -- from the synthetic body of
`avatarcap_amd.synthetic` instead of the licensed SMPL model and a captured sequence.

One-time work runs on the device: the KNN-1 band test is the HIP KNN kernel
(1.16e11 pair tests at 256^3, avatarcap_dataset.py:114); the reference's trimesh/embree
`contains` (:121-125) is replaced by the sign of the analytic capsule SDF of the synthetic body.
"""
from __future__ import annotations

import numpy as np
import torch

from . import config
from . import synthetic as syn
from .grid import generate_volume_points, volume_axes


class SyntheticTestDataset:
    def __init__(self, vol_res=None, valid='band', n_frames=8, seed=syn.SEED, device=None, bounds=None):
        self.device = torch.device(device) if device is not None else config.device
        self.vol_res = [int(r) for r in (vol_res or config.cfg['testing']['vol_res'])]
        self.body = syn.synthetic_body(seed)
        self.cano_smpl_v = torch.from_numpy(self.body['cano_smpl_v'])
        self.cano_bounds = np.asarray(bounds if bounds is not None else syn.CANO_BOUNDS, np.float32)
        v = self.body['cano_smpl_v']
        self.cano_smpl_center = (0.5 * (v.max(0) + v.min(0))).astype(np.float32)        # avatarcap_dataset.py:65-66
        self.n_frames = n_frames
        self.seed = seed
        self.valid_mode = valid

        vol_pts = generate_volume_points(self.cano_bounds, self.vol_res, self.device)     # :111
        N = vol_pts.shape[0]
        self.grid_axes = volume_axes(self.cano_bounds, self.vol_res, self.device)        # per-axis tables of the same grid (avc_avatar_query_grid)
        if valid == 'dense':
            self.infer_pts_flag = torch.ones(N, dtype=torch.bool, device=self.device)
            self.infer_pts = vol_pts
            self.invalid_pts_ov = torch.empty(0, dtype=torch.float32, device=self.device)
        elif valid == 'band':
            from .utils.smpl_util import SmplUtil
            su = SmplUtil()
            d2 = torch.empty(N, dtype=torch.float32, device=self.device)
            cv = self.cano_smpl_v.to(self.device)
            step = 1 << 22
            for s in range(0, N, step):                                                    # :114
                d, _ = su.knn_points(vol_pts[None, s:s + step], cv[None], K=1)
                d2[s:s + step] = d[0, :, 0]
            self.infer_pts_flag = d2 < 0.1 ** 2                                            # :116
            self.infer_pts = vol_pts[self.infer_pts_flag].contiguous()                     # :118
            inv = vol_pts[~self.infer_pts_flag]
            sign = torch.empty(inv.shape[0], dtype=torch.float32, device=self.device)
            for s in range(0, inv.shape[0], 1 << 21):                                      # :121-125 (contains -> [-1, 1]), on the device where there is one
                sdf = syn.body_sdf_device(inv[s:s + (1 << 21)]) if self.device.type == 'cuda' else torch.from_numpy(syn.body_sdf(inv[s:s + (1 << 21)].numpy()))
                sign[s:s + (1 << 21)] = torch.where(sdf < 0, 1.0, -1.0).to(torch.float32)
            self.invalid_pts_ov = sign
        else:
            raise ValueError("valid must be 'dense' or 'band'")
        self.valid_u8 = self.infer_pts_flag.to(torch.uint8).contiguous()
        # flat grid indices of infer_pts, same order (the band is queried by index: avc_avatar_query_grid_subset); a dense frame needs none
        self.valid_idx = None if valid == 'dense' else torch.nonzero(self.infer_pts_flag, as_tuple=False)[:, 0].to(torch.int32).contiguous()

    def __len__(self):
        return self.n_frames

    def __getitem__(self, idx):
        """Item dict with the keys the hot path reads (Appendix B), tensors on the host like the
        reference's dataset; `to_cuda(item, add_batch=True)` moves them."""
        rs = np.random.RandomState(self.seed * 7919 + idx)
        return {
            'data_idx': idx,
            'cano_pts': self.infer_pts,                                                   # already on device (:118,305)
            'valid_pts_flag': self.infer_pts_flag,
            'smpl_pos_map': rs.uniform(-1, 1, (6, 256, 256)).astype(np.float32),          # :207-213
            'cano_smpl_center': self.cano_smpl_center,
            'cano_bounds': self.cano_bounds,
            'cano2live_jnt_mats': syn.random_pose_jnt_mats(self.seed * 31 + idx),         # :193-201
        }


def to_cuda(items: dict, add_batch=False):
    """dataset/avatarcap_dataset.py:329-346.  The dict also keeps, under '_host', the items as they came in: the C-ABI takes per-sequence constants
    (`cano_smpl_center`, ...) by value, and `_lib.host_f3` reads them there instead of back from the device."""
    out = {}
    for key, data in items.items():
        if key in ('_host', '_host_ids'):
            continue
        if isinstance(data, torch.Tensor):
            out[key] = data.to(config.device)
        elif isinstance(data, np.ndarray):
            out[key] = torch.from_numpy(data).to(config.device)
        else:
            out[key] = data
        if add_batch and isinstance(out[key], torch.Tensor):
            out[key] = out[key].unsqueeze(0)
    out['_host'] = items.get('_host', items)
    out['_host_ids'] = {k: id(v) for k, v in out.items() if isinstance(v, torch.Tensor)}
    return out


def synthetic_camera(img_size=512, distance=2.6, focal=None):
    """A pinhole in front of the body for the synthetic stand-in of the captured view (the reference reads `w2c_RT` and the
    intrinsics from the sequence's camera file): x right, y down, z forward, looking down the world's -z at the origin."""
    focal = float(focal if focal is not None else 1.1 * img_size)
    w2c = np.diag(np.float32([1, -1, -1, 1])); w2c[2, 3] = distance
    return w2c, {'fx': focal, 'fy': focal, 'cx': img_size / 2.0, 'cy': img_size / 2.0, 'img_w': img_size, 'img_h': img_size}


_device_consts = {}


def _const(values, device):
    """A small constant on the device, uploaded once per (value, device): an upload from pageable memory waits for the stream it is queued on."""
    a = np.ascontiguousarray(values, np.float32)
    key = (a.tobytes(), a.shape, str(device))
    if key not in _device_consts:
        if len(_device_consts) > 64:
            _device_consts.clear()
        _device_consts[key] = torch.from_numpy(a).to(device)
    return _device_consts[key]


def synthetic_observed_normals(live_v, live_vn, faces, w2c, cam, wobble=0.25, seed=0, axes=None):
    """What a normal-estimation network would hand over for the posed mesh, synthesised: the posed normals, bent by a smooth
    random rotation field, in the image convention the reference undoes (camera frame with y and z negated), (H,W,3).
    `axes`: the (3,3) field directions already on the device (main.py's prefetch thread draws them: torch.randn under manual_seed(seed))."""
    from .utils.renderer import gl_perspective_projection_matrix, render_mesh_device
    if axes is None:
        g = torch.Generator().manual_seed(seed)
        axes = torch.randn(3, 3, generator=g).to(live_v.device)
    ax = axes
    bend = wobble * torch.sin(live_v @ ax * 4.0)                                     # smooth axis-angle field over space
    n = live_vn + torch.cross(bend, live_vn, dim=-1)
    n = n / torch.linalg.norm(n, dim=-1, keepdim=True).clamp_min(1e-12)
    R = _const(np.asarray(w2c, np.float32)[:3, :3], live_v.device)
    ncam = (n @ R.T) * _const([1., -1., -1.], live_v.device)
    mvp = gl_perspective_projection_matrix(cam['fx'], cam['fy'], cam['cx'], cam['cy'], cam['img_w'], cam['img_h']) @ np.asarray(w2c, np.float32)
    return render_mesh_device(live_v, ncam.contiguous(), faces, mvp, cam['img_w'], cam['img_h'])[..., :3].contiguous()
