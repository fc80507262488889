"""Test-mode mirror of the reference's `AvatarCapDataset` (dataset/avatarcap_dataset.py:26-326) for a CAPTURED sequence:

    <data_dir>/dataConfig.yaml                      data_type ('synthetic' | 'real'), pos_map_name, pos_map_res, camera {fx, fy, cx, cy, img_width, img_height}
    <data_dir>/smpl/shape.txt, pose_%04d.txt        SMPL shape (10) and per-frame pose (75)
    <data_dir>/smpl/smpl_pos_map_%04d[_<name>].exr  front | back position map of the posed SMPL
    <data_dir>/imgs/%03d/cams.mat                   (synthetic multi-view data) camera extrinsics
    smpl_files/basicmodel_<g>_lbs_10_207_0_v1.0.0.pkl   the licensed SMPL model (NOT part of this repository)

Same attributes and item keys as the reference's dataset in test mode; missing files raise what the reference's
own open / loadtxt calls raise (FileNotFoundError / OSError).  Training mode is out of scope.  What differs, on purpose:
  * the valid-band KNN (:114) runs on the HIP KNN kernel when a device is present (exact, same rule d^2 < 0.01);
  * `trimesh ... contains` (:121-125, embree) is the column-parity test of utils/mesh_contains.py;
  * OpenCV's three calls are restated in utils/cv_compat.py / utils/exr_io.py;
  * the NeRF ray samples of __getitem__ (sample_ray_h36m, :233-235) are not produced: main.py's test loop overwrites every one of those keys
    before the renderer reads them (main.py:468-473).
Pinned by tests/golden/dataset_golden.npz: the REFERENCE's dataset class run on a synthetic sequence with a synthetic model file
(tests/golden/make_golden_dataset.py), item by item.
"""
from __future__ import annotations

import glob
import math
import os

import numpy as np
import torch
import yaml

from . import config
from .grid import generate_volume_points, volume_axes_np
from .smpl import SmplModel, load_smpl_params
from .utils.cv_compat import load_smpl_pos_map, rodrigues


def read_data_config(data_dir):
    """dataset/avatarcap_dataset.py:32 -- FileNotFoundError when the sequence has no dataConfig.yaml, as in the reference."""
    with open(data_dir + '/dataConfig.yaml', encoding='UTF-8') as f:
        return yaml.load(f, Loader=yaml.FullLoader)


class AvatarCapDataset:
    def __init__(self, data_dir, training=False, smpl_params=None, device=None):
        if training:
            raise NotImplementedError('training mode is out of scope of the MI355X hot-path build (SURVEY.md section 2)')
        self.data_dir = data_dir
        self.training = False
        self.device = torch.device(device) if device is not None else config.device
        self.data_config = read_data_config(data_dir)                                                       # :32
        self.smpl_pose_list = sorted(glob.glob(os.path.join(self.data_dir, 'smpl/pose_*.txt')))              # :34
        self.data_type = self.data_config.get('data_type', 'synthetic')                                      # :36-48
        if self.data_type == 'synthetic':
            print('# Synthetic data')
            self.color_img_list = sorted(glob.glob(os.path.join(self.data_dir, 'imgs/*/color_view_*.jpg')))
        elif self.data_type == 'real':
            print('# Real data')
            self.color_img_list = sorted(glob.glob(os.path.join(self.data_dir, 'imgs/color/color_*.jpg')))
        else:
            raise ValueError('Invalid data type!')
        if not self.smpl_pose_list:
            raise FileNotFoundError(os.path.join(self.data_dir, 'smpl/pose_*.txt'))
        self.img_num_per_pose = max(1, len(self.color_img_list) // len(self.smpl_pose_list))                  # :50-53
        print('# Each pose contains %d view images' % self.img_num_per_pose)
        self.start_data_idx = int(os.path.basename(self.smpl_pose_list[0]).replace('pose_', '').replace('.txt', ''))   # :55
        print('# Start data index: %d' % self.start_data_idx)
        self.smpl_params = smpl_params if smpl_params is not None else load_smpl_params()
        self.smpl_shape = np.loadtxt(os.path.join(self.data_dir, 'smpl/shape.txt'))                          # :59

        # canonical SMPL: A-pose-ish legs (:62-72)
        self.cano_smpl_pose = np.zeros(75, dtype=np.float32)
        self.cano_smpl_pose[3 + 3 * 1 + 2] = math.radians(25)
        self.cano_smpl_pose[3 + 3 * 2 + 2] = math.radians(-25)
        self.cano_smpl = SmplModel(self.cano_smpl_pose, self.smpl_shape, self.smpl_params)
        center = 0.5 * (self.cano_smpl.posed_vertices.min(0) + self.cano_smpl.posed_vertices.max(0))
        self.cano_smpl_center = torch.from_numpy(center).to(torch.float32)
        self.cano_smpl_jnts = torch.from_numpy(self.cano_smpl.posed_joints).to(torch.float32)
        self.cano_smpl_v = torch.from_numpy(self.cano_smpl.posed_vertices).to(torch.float32)
        self.inv_cano_jnt_mats = torch.from_numpy(np.linalg.inv(self.cano_smpl.jnt_affine_mats)).to(torch.float32)

        # position-map pose (:75-90)
        self.pos_map_name = self.data_config.get('pos_map_name', 'cano')
        self.pos_map_res = self.data_config.get('pos_map_res', 256)
        J = self.smpl_params.joint_num
        if self.pos_map_name == 'cano':
            self.cano2posmap_jnt_mats = torch.eye(4, dtype=torch.float32)[None].repeat(J, 1, 1)
        elif self.pos_map_name == 'A':
            pose = np.zeros(75, np.float32)
            pose[3 + 16 * 3 + 2] = -math.radians(60)
            pose[3 + 17 * 3 + 2] = math.radians(60)
            mats = torch.from_numpy(SmplModel(pose, self.smpl_shape, self.smpl_params).jnt_affine_mats).to(torch.float32)
            self.cano2posmap_jnt_mats = torch.matmul(mats, self.inv_cano_jnt_mats)
        elif self.pos_map_name == 'uv':
            pass                                                                                             # "not implemented" in the reference too (:87)
        else:
            raise ValueError('Invalid pos_map_name!')

        # canonical bounds (:93-100)
        v = self.cano_smpl.posed_vertices
        lo, hi = np.min(v, axis=0), np.max(v, axis=0)
        lo[:2] -= 0.05; hi[:2] += 0.05
        lo[2] -= 0.15; hi[2] += 0.15
        self.cano_bounds = np.stack([lo, hi], axis=0).astype(np.float32)
        print('# Canonical volume len: {}'.format(self.cano_bounds[1] - self.cano_bounds[0]))

        # camera intrinsics (:103-109)
        cam = self.data_config['camera']
        self.K = np.identity(3, np.float32)
        self.K[0, 0], self.K[0, 2], self.K[1, 1], self.K[1, 2] = cam['fx'], cam['cx'], cam['fy'], cam['cy']
        self.img_w, self.img_h = cam['img_width'], cam['img_height']

        # canonical grid, valid band, inside / outside fill (:111-125)
        self.vol_res = [int(r) for r in config.cfg['testing']['vol_res']]
        vol_pts = generate_volume_points(self.cano_bounds, self.vol_res, self.device)
        self.grid_axes = tuple(torch.from_numpy(a).to(self.device) for a in volume_axes_np(self.cano_bounds, self.vol_res))
        d2 = self._nearest_d2(vol_pts, self.cano_smpl_v.to(self.device))
        self.infer_pts_flag = d2 < 0.1 ** 2
        self.infer_pts = vol_pts[self.infer_pts_flag].contiguous()
        self.infer_pts_lbs = None
        from .utils.mesh_contains import grid_contains
        inside = grid_contains(self.cano_smpl.posed_vertices, self.smpl_params.faces, *volume_axes_np(self.cano_bounds, self.vol_res), device=self.device)
        ov = 2. * inside[~self.infer_pts_flag].to(torch.float32) - 1.                                        # [0, 1] -> [-1, 1]
        self.invalid_pts_ov = ov.to(self.device)
        self.valid_u8 = self.infer_pts_flag.to(torch.uint8).contiguous()
        self.valid_idx = torch.nonzero(self.infer_pts_flag, as_tuple=False)[:, 0].to(torch.int32).contiguous()     # flat grid indices of infer_pts, same order
        self.valid_mode = 'band'
        # what FramePipeline reads from its dataset
        self.body = {'cano_smpl_v': self.cano_smpl.posed_vertices.astype(np.float32), 'skin_weights': self.smpl_params.weights}

    @staticmethod
    def _nearest_d2(pts, ref):
        """knn_points(vol_pts, cano_smpl_v, K = 1) squared distances (:114)."""
        if pts.device.type == 'cuda':
            from .utils.smpl_util import SmplUtil
            su = SmplUtil()
            out = torch.empty(pts.shape[0], dtype=torch.float32, device=pts.device)
            for s in range(0, pts.shape[0], 1 << 22):
                d, _ = su.knn_points(pts[None, s:s + (1 << 22)], ref[None], K=1)
                out[s:s + (1 << 22)] = d[0, :, 0]
            return out
        out = torch.empty(pts.shape[0], dtype=torch.float32)
        for s in range(0, pts.shape[0], 4096):                                                               # host path: plain torch, small grids only
            out[s:s + 4096] = ((pts[s:s + 4096, None, :] - ref[None]) ** 2).sum(-1).min(1)[0]
        return out

    def __len__(self):
        return len(self.smpl_pose_list) * self.img_num_per_pose                                              # :179

    def __getitem__(self, index):
        pose_idx, view_idx = index // self.img_num_per_pose, index % self.img_num_per_pose                   # :182-183
        smpl_pose_path = self.smpl_pose_list[pose_idx]
        data_idx = int(os.path.splitext(os.path.basename(smpl_pose_path))[0].replace('pose_', ''))            # :188-190
        print('data idx: %d, view idx: %d' % (data_idx, view_idx))
        live_pose = np.loadtxt(smpl_pose_path).astype(np.float32)                                            # :194-196 (hands zeroed)
        live_pose[3 + 22 * 3: 6 + 22 * 3] = 0.
        live_pose[3 + 23 * 3: 6 + 23 * 3] = 0.
        live = SmplModel(live_pose, self.smpl_shape, self.smpl_params)
        cano2live = torch.matmul(torch.from_numpy(live.jnt_affine_mats).to(torch.float32), self.inv_cano_jnt_mats)     # :198
        path = self.data_dir + '/smpl/smpl_pos_map_%04d_%s.exr' % (data_idx, self.pos_map_name)               # :207-213
        if not os.path.exists(path):
            path = self.data_dir + '/smpl/smpl_pos_map_%04d.exr' % data_idx
        smpl_pos_map = load_smpl_pos_map(path, self.pos_map_res)
        cam_path = os.path.join(self.data_dir + '/imgs/%03d/cams.mat' % data_idx)                            # :224-232
        w2c_RT = np.identity(4, dtype=np.float32)
        if os.path.exists(cam_path):
            import scipy.io as sio
            cam = sio.loadmat(cam_path)
            w2c_RT[:3, :3] = rodrigues(np.float32(cam['cam_rs'][view_idx]))
            w2c_RT[:3, 3] = np.float32(cam['cam_ts'][view_idx]).reshape(-1)
        return {                                                                                             # :253-264, :279, :304-306
            'data_idx': data_idx, 'view_idx': view_idx,
            'smpl_pose': torch.from_numpy(live_pose),
            'smpl_pos_map': torch.from_numpy(smpl_pos_map),
            'cano2live_jnt_mats': cano2live,
            'cano2posmap_jnt_mats': getattr(self, 'cano2posmap_jnt_mats', None),
            'cano_bounds': torch.from_numpy(self.cano_bounds),
            'cano_smpl_center': self.cano_smpl_center,
            'cano_smpl_jnts': self.cano_smpl_jnts,
            'live_smpl_v': torch.from_numpy(live.posed_vertices.astype(np.float32)),
            'img_h': self.img_h, 'img_w': self.img_w, 'w2c_RT': w2c_RT,
            'cano_pts': self.infer_pts, 'valid_pts_flag': self.infer_pts_flag,
        }

    generate_volume_points = staticmethod(generate_volume_points)                                            # :312
