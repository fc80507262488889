"""Synthetic stand-ins for everything the reference loads from licensed / absent files.

The reference needs an SMPL pickle, a captured dataset and trained checkpoints
(`dataset/smpl.py:45-46`, `dataset/avatarcap_dataset.py:27-125`, `main.py:302-320`);
none of them exist offline.  This module produces deterministic replacements with
the same shapes / dtypes / dict keys, from plain NumPy
seeds, so fixtures only have to store seeds + inputs + expected outputs.

Nothing here is on the timed hot path; it is the harness of SURVEY.md §8(a) row H.
"""
from __future__ import annotations

import math
import numpy as np

SEED = 31359  # main.py:508-509

# SMPL kinematic tree (public: parent index of each of the 24 joints)
SMPL_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21], np.int32)

# Synthetic canonical bounds, SMPL-like bbox + margins of avatarcap_dataset.py:93-96
CANO_BOUNDS = np.array([[-0.95, -1.00, -0.30], [0.95, 0.85, 0.30]], np.float32)
N_SMPL_VERTS = 6890
N_JOINTS = 24


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------
def synth_state_dict(shapes: dict, seed: int = SEED, gain: float = 1.6, spectral_decay: bool = True) -> dict:
    """Deterministic NumPy recipe for a state_dict with the given {key: shape}.

    Rules are keyed on the *name* so the same recipe fills the reference modules (in
    tests/golden/make_golden.py) and this package's modules:
      * conv / linear weights (ndim >= 2):   U(-b, b), b = gain * sqrt(3 / fan_in)
      * weight_v:                            same as a weight
      * weight_g:                            U(0.5, 1.5) * ||v||-free scale (per out channel)
      * biases:                              N(0, 0.05^2)
      * norm weight:                         U(0.5, 1.5);  norm bias: N(0, 0.1^2)
      * running_mean:                        N(0, 0.1^2);  running_var: U(0.5, 1.5)
      * num_batches_tracked:                 0
    Keys are visited in sorted order, each with its own RandomState(seed, crc(key)), so
    adding/removing keys never perturbs the others.
    The reference initialises two heads to +-1e-5 (arch_avatar.py:17-23,60,105); with this
    recipe they get ordinary weights, which keeps the parity checks non-vacuous
    (SURVEY.md section 8(c), G5).
    """
    import zlib
    out = {}
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        rs = np.random.RandomState((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 31 - 1))
        leaf = key.split('.')[-1]
        if leaf == 'num_batches_tracked':
            out[key] = np.zeros(shape, np.int64)
        elif leaf == 'running_mean':
            out[key] = (0.1 * rs.randn(*shape)).astype(np.float32)
        elif leaf == 'running_var':
            out[key] = rs.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf == 'weight_g':
            out[key] = rs.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf in ('weight', 'weight_v') and len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            if '.up.weight' in key and len(shape) == 4:  # ConvTranspose2d: (in, out, kh, kw)
                fan_in = shape[0] * shape[2] * shape[3] // 4
            b = gain * math.sqrt(3.0 / fan_in)
            w = rs.uniform(-b, b, shape).astype(np.float32)
            if spectral_decay and (key.endswith('shared_mlp.fc_list.0.0.weight') or key.endswith('shared_mlp.fc_list.4.0.weight')):
                # Trained implicit nets carry little energy in the top octaves of the positional
                # encoding.  Without this the synthetic field has d(occ)/d(point) ~ 1e3, so fp32
                # rounding of (p + offset) alone moves the occupancy by > 1e-2 and neither the
                # reference's own fp32 path nor any fp32 implementation could meet a 1e-4 bar; it
                # also keeps the field smooth at voxel scale so marching cubes sees a surface.
                # Embedding columns = [xyz, sin f0, cos f0, sin f1, ...] (net_util.py:37); layer 4
                # sees them after its 256 hidden inputs (mlp.py:61).
                # (the embedding is 3 + 6 L wide for cano_template.pos_encoding = L: 63 for the example's 10)
                pe = shape[1] if key.endswith('fc_list.0.0.weight') else shape[1] - 256
                c0 = shape[1] - pe
                for f in range((pe - 3) // 6):
                    w[:, c0 + 3 + 6 * f: c0 + 9 + 6 * f] *= 2.0 ** (-1.35 * f)
            if spectral_decay and key.endswith('out_layer_coord_affine.weight'):
                w *= 0.05          # non-rigid offsets of a few centimetres, like a trained warping field
            out[key] = w
        elif leaf == 'weight':   # norm affine
            out[key] = rs.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf == 'bias':
            parent = key.rsplit('.', 1)[0]
            is_norm = (parent + '.running_mean') in shapes or ('bn' in parent.split('.')[-1] and len(shape) == 1
                                                              and (parent + '.weight') in shapes
                                                              and len(shapes[parent + '.weight']) == 1)
            out[key] = ((0.1 if is_norm else 0.05) * rs.randn(*shape)).astype(np.float32)
        else:
            out[key] = (0.05 * rs.randn(*shape)).astype(np.float32)
    return out


def module_shapes(module) -> dict:
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def load_synth(module, seed: int = SEED, **kw):
    """Fill a torch module (reference's or ours) with the seeded recipe."""
    import torch
    sd = synth_state_dict(module_shapes(module), seed, **kw)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return sd


# --------------------------------------------------------------------------------------
# body: capsule skeleton, "SMPL" vertices, skin weights, poses
# --------------------------------------------------------------------------------------
def _joint_rest_positions() -> np.ndarray:
    """24 joints of a T-posed body that fits CANO_BOUNDS minus its margins (y up)."""
    J = np.zeros((24, 3), np.float64)
    J[0] = (0.00, -0.05, 0.0)       # pelvis
    J[1] = (0.09, -0.14, 0.0)       # l hip
    J[2] = (-0.09, -0.14, 0.0)      # r hip
    J[3] = (0.00, 0.06, 0.0)        # spine1
    J[4] = (0.13, -0.52, 0.0)       # l knee  (legs spread like the reference's cano pose, smpl_util.py:17-18)
    J[5] = (-0.13, -0.52, 0.0)
    J[6] = (0.00, 0.19, 0.0)        # spine2
    J[7] = (0.17, -0.90, 0.0)       # l ankle
    J[8] = (-0.17, -0.90, 0.0)
    J[9] = (0.00, 0.26, 0.0)        # spine3
    J[10] = (0.18, -0.94, 0.10)     # l foot
    J[11] = (-0.18, -0.94, 0.10)
    J[12] = (0.00, 0.46, 0.0)       # neck
    J[13] = (0.08, 0.38, 0.0)       # l collar
    J[14] = (-0.08, 0.38, 0.0)
    J[15] = (0.00, 0.60, 0.0)       # head
    J[16] = (0.19, 0.40, 0.0)       # l shoulder
    J[17] = (-0.19, 0.40, 0.0)
    J[18] = (0.45, 0.40, 0.0)       # l elbow
    J[19] = (-0.45, 0.40, 0.0)
    J[20] = (0.70, 0.40, 0.0)       # l wrist
    J[21] = (-0.70, 0.40, 0.0)
    J[22] = (0.82, 0.40, 0.0)       # l hand
    J[23] = (-0.82, 0.40, 0.0)
    return J


_BONE_RADIUS = np.array([0.13, 0.085, 0.085, 0.13, 0.065, 0.065, 0.13, 0.05, 0.05, 0.13, 0.04, 0.04,
                         0.06, 0.07, 0.07, 0.10, 0.055, 0.055, 0.045, 0.045, 0.035, 0.035, 0.03, 0.03])


def _capsules():
    """(a, b, r, owner) per joint j>0: segment parent->joint, skinned to the parent joint
    (the proximal joint rotates the segment); plus a head blob owned by joint 15."""
    J = _joint_rest_positions()
    segs = []
    for j in range(1, 24):
        segs.append((J[SMPL_PARENTS[j]], J[j], _BONE_RADIUS[j], int(SMPL_PARENTS[j])))
    segs.append((J[15], J[15] + np.array([0, 0.10, 0.0]), 0.10, 15))
    return segs


def body_sdf(p: np.ndarray) -> np.ndarray:
    """Signed distance (negative inside) of the union of capsules; p (..., 3) float."""
    p = np.asarray(p, np.float64)
    d = np.full(p.shape[:-1], np.inf)
    for a, b, r, _ in _capsules():
        ab = b - a
        t = np.clip(((p - a) @ ab) / max(ab @ ab, 1e-12), 0.0, 1.0)
        c = a + t[..., None] * ab
        d = np.minimum(d, np.linalg.norm(p - c, axis=-1) - r)
    return d


def body_sdf_device(p):
    """body_sdf on a torch tensor (..., 3), in float64 on the tensor's device: the same formula, for the 14 M points of a 256^3 grid that the dataset classifies
    inside / outside (36 s of NumPy on the host, well under a second on the device)."""
    import torch
    q = p.to(torch.float64)
    d = torch.full(q.shape[:-1], float('inf'), dtype=torch.float64, device=q.device)
    for a, b, r, _ in _capsules():
        a_t, ab = torch.tensor(a, dtype=torch.float64, device=q.device), torch.tensor(b - a, dtype=torch.float64, device=q.device)
        t = (((q - a_t) @ ab) / max(float((b - a) @ (b - a)), 1e-12)).clamp_(0.0, 1.0)
        c = a_t + t[..., None] * ab
        d = torch.minimum(d, torch.linalg.norm(q - c, dim=-1) - float(r))
    return d


def synthetic_body(seed: int = SEED):
    """Returns dict(cano_smpl_v (6890,3) f32, skin_weights (6890,24) f32, joints (24,3) f32)."""
    rs = np.random.RandomState(seed + 1)
    segs = _capsules()
    lens = np.array([np.linalg.norm(b - a) + 2 * r for a, b, r, _ in segs])
    area = lens * np.array([s[2] for s in segs])
    counts = np.floor(area / area.sum() * N_SMPL_VERTS).astype(int)
    counts[0] += N_SMPL_VERTS - counts.sum()
    verts = []
    for (a, b, r, _), n in zip(segs, counts):
        ab = b - a
        L = np.linalg.norm(ab)
        u = ab / max(L, 1e-9)
        # orthonormal frame
        t = np.array([1.0, 0, 0]) if abs(u[0]) < 0.9 else np.array([0, 1.0, 0])
        e1 = np.cross(u, t); e1 /= np.linalg.norm(e1)
        e2 = np.cross(u, e1)
        s = rs.uniform(-r, L + r, n)
        phi = rs.uniform(0, 2 * math.pi, n)
        sc = np.clip(s, 0, L)
        over = s - sc                      # signed overshoot into the end caps
        rad = np.sqrt(np.maximum(r * r - over * over, 0.0))
        pts = a + sc[:, None] * u + over[:, None] * u + rad[:, None] * (np.cos(phi)[:, None] * e1 + np.sin(phi)[:, None] * e2)
        verts.append(pts)
    v = np.concatenate(verts, 0)
    # project onto the union surface (drop points buried in a neighbouring capsule)
    for _ in range(3):
        d = body_sdf(v)
        eps = 1e-4
        g = np.stack([(body_sdf(v + eps * np.eye(3)[i]) - d) / eps for i in range(3)], -1)
        g /= np.maximum(np.linalg.norm(g, axis=-1, keepdims=True), 1e-9)
        v = v - d[:, None] * g
    lo, hi = CANO_BOUNDS[0] + np.array([0.05, 0.05, 0.15]), CANO_BOUNDS[1] - np.array([0.05, 0.05, 0.15])
    v = np.clip(v, lo, hi)
    J = _joint_rest_positions()
    # skin weights: softmax(-d^2 / sigma^2) to the bones, bone j (parent->j) drives the parent joint
    W = np.zeros((N_SMPL_VERTS, 24))
    sigma2 = 0.06 ** 2
    for a, b, r, owner in segs:
        ab = b - a
        t = np.clip(((v - a) @ ab) / max(ab @ ab, 1e-12), 0.0, 1.0)
        c = a + t[:, None] * ab
        d2 = ((v - c) ** 2).sum(-1)
        W[:, owner] += np.exp(-d2 / sigma2)
    W /= W.sum(-1, keepdims=True)
    return {'cano_smpl_v': v.astype(np.float32), 'skin_weights': W.astype(np.float32), 'joints': J.astype(np.float32)}


def _rodrigues(r: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)


def random_pose_jnt_mats(seed: int, sigma: float = 0.3) -> np.ndarray:
    """cano2live joint 4x4s (24,4,4) f32 = live_jnt @ inv(cano_jnt) (avatarcap_dataset.py:193-201)
    for 24 axis-angles ~ N(0, sigma^2) through the SMPL tree (dataset/smpl.py:49-110)."""
    rs = np.random.RandomState(seed)
    J = _joint_rest_positions()
    aa = sigma * rs.randn(24, 3)
    aa[0] *= 0.3
    G = np.zeros((24, 4, 4))
    for j in range(24):
        L = np.eye(4)
        L[:3, :3] = _rodrigues(aa[j])
        L[:3, 3] = J[j] - (J[SMPL_PARENTS[j]] if j > 0 else 0)
        G[j] = L if j == 0 else G[SMPL_PARENTS[j]] @ L
    mats = np.zeros((24, 4, 4))
    for j in range(24):
        T = np.eye(4); T[:3, 3] = -J[j]
        mats[j] = G[j] @ T      # rest pose is identity rotations => inv(cano_jnt) is a translation
    return mats.astype(np.float32)


def smooth_normal_maps(seed: int, res: int = 512) -> np.ndarray:
    """(6,res,res) f32 unit-norm smooth random fields (front ++ back) for ReconNetwork input."""
    rs = np.random.RandomState(seed)
    out = []
    ys, xs = np.meshgrid(np.linspace(0, 1, res), np.linspace(0, 1, res), indexing='ij')
    for _ in range(2):
        n = np.zeros((3, res, res))
        for c in range(3):
            for _k in range(4):
                fx, fy, ph = rs.uniform(0.5, 4), rs.uniform(0.5, 4), rs.uniform(0, 6.28)
                n[c] += np.sin(2 * math.pi * (fx * xs + fy * ys) + ph)
        n[2] += 3.0
        n /= np.linalg.norm(n, axis=0, keepdims=True)
        out.append(n)
    return np.concatenate(out, 0).astype(np.float32)


def test_item(seed: int, res, valid: str = 'dense', body: dict | None = None) -> dict:
    """The test-mode item dict of avatarcap_dataset.py:253-308 (numpy, no batch dim), Appendix B keys."""
    from .grid import generate_volume_points_np
    body = body or synthetic_body()
    rs = np.random.RandomState(seed)
    pts = generate_volume_points_np(CANO_BOUNDS, res)
    v = body['cano_smpl_v']
    center = 0.5 * (v.max(0) + v.min(0))
    item = {
        'cano_bounds': CANO_BOUNDS.copy(),
        'cano_smpl_center': center.astype(np.float32),
        'smpl_pos_map': rs.uniform(-1, 1, (6, 256, 256)).astype(np.float32),
        'cano2live_jnt_mats': random_pose_jnt_mats(seed),
        'vol_pts': pts,
    }
    if valid == 'dense':
        item['valid_pts_flag'] = np.ones(pts.shape[0], bool)
    else:
        raise ValueError('band masks are built on device: see avatarcap_amd.dataset')
    item['cano_pts'] = pts[item['valid_pts_flag']]
    return item
