"""SMPL body model, NumPy (reference `dataset/smpl.py:1-129`): shape blend shapes -> joint regression -> Rodrigues per joint ->
kinematic chain -> linear blend skinning.  Only needed by the real-data loader (`avatarcap_amd.avatarcap_dataset`); the licensed
model file `smpl_files/basicmodel_{m,f}_lbs_10_207_0_v1.0.0.pkl` is NOT part of this repository -- `load_smpl_params` raises the
FileNotFoundError the reference's import would raise when it is absent.  Pinned by tests/golden/dataset_golden.npz: the reference's own
SmplModel run on a synthetic model file of the same layout (tests/golden/make_golden_dataset.py).
"""
from __future__ import annotations

import os
import pickle

import numpy as np

from .utils.cv_compat import rodrigues


class SmplParams:
    """dataset/smpl.py:10-44.  `data` is the unpickled dict of the model file (chumpy / scipy objects are only touched through
    np.array / the sparse product, as the reference does)."""

    def __init__(self, model_path):
        self.model_path = model_path
        with open(model_path, 'rb') as f:                                  # FileNotFoundError propagates, as in the reference (:17)
            data = pickle.load(f, encoding='latin1')
        self.mean_vertices = np.asarray(data['v_template']).astype(np.float32)
        self.vnum = self.mean_vertices.shape[0]
        self.faces = np.asarray(data['f']).astype(np.int32)
        self.fnum = self.faces.shape[0]
        self.joints = np.asarray(data['J']).astype(np.float32)
        self.kintree = np.asarray(data['kintree_table']).astype(np.int32).transpose()
        self.joint_num = self.kintree.shape[0]
        self.weights = np.asarray(data['weights']).astype(np.float32)
        self.sparse_regressor = data['J_regressor']
        self.shape_blend_shape = np.array(data['shapedirs'], dtype=np.float32).reshape(self.vnum * 3, -1)


_params: dict[str, SmplParams] = {}


def model_path(gender=None, root=None):
    from . import config
    root = root or os.environ.get('AVC_SMPL_DIR') or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'smpl_files')
    return os.path.join(root, 'basicmodel_%s_lbs_10_207_0_v1.0.0.pkl' % (gender or config.smpl_gender))       # dataset/smpl.py:48


def load_smpl_params(gender=None, root=None) -> SmplParams:
    p = model_path(gender, root)
    if p not in _params:
        _params[p] = SmplParams(p)
    return _params[p]


class SmplModel:
    """dataset/smpl.py:51-113.  pose (75,) = global translation (3) + 24 axis-angles; shape (10,)."""

    def __init__(self, pose_coeff, shape_coeff, params: SmplParams | None = None):
        self.params = params if params is not None else load_smpl_params()
        self.pose_coeff = np.asarray(pose_coeff).reshape(75, 1)
        self.shape_coeff = np.asarray(shape_coeff).reshape(10, 1)
        self.change_shape()
        self.change_pose()

    def change_shape(self):
        P = self.params
        mean = P.mean_vertices.reshape(P.vnum * 3, 1)
        self.shaped_vertices = (mean + np.dot(P.shape_blend_shape, self.shape_coeff)).reshape(-1, 3)       # :68-70
        self.joints = P.sparse_regressor * self.shaped_vertices                                              # :73 (scipy sparse product)

    def change_pose(self):
        P = self.params
        local = []
        for j in range(P.joint_num):                                                                          # :78-90
            r = rodrigues(self.pose_coeff[3 * j + 3: 3 * j + 6])
            m = np.identity(4)
            m[0:3, 0:3] = r
            m[0:3, 3] = self.pose_coeff[0:3, 0] if j == 0 else np.dot(np.identity(3) - r, np.asarray(self.joints[j]).reshape(-1).transpose())
            local.append(m)
        mats = [local[0]]
        for j in range(1, P.joint_num):                                                                       # :93-98
            mats.append(np.dot(mats[P.kintree[j, 0]], local[j]))
        self.jnt_affine_mats = np.array(mats)
        self.posed_joints = np.zeros_like(P.joints)
        for j in range(P.joint_num):                                                                          # :101-105
            self.posed_joints[j] = np.dot(mats[j][:3, :3], np.asarray(self.joints[j]).reshape(-1)) + mats[j][:3, 3]
        self.vertex_affine_mats = np.einsum('vj,jab->vab', P.weights, self.jnt_affine_mats)                   # :108-110
        self.posed_vertices = np.einsum('vab,vb->va', self.vertex_affine_mats[:, :3, :3], self.shaped_vertices) + self.vertex_affine_mats[:, :3, 3]
