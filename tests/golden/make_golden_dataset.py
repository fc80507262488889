#!/usr/bin/env python3
"""Golden vectors of the test-mode data seam, produced by RUNNING THE REFERENCE'S OWN `AvatarCapDataset` and `SmplModel`
(/root/reference/dataset/avatarcap_dataset.py, dataset/smpl.py) on a synthetic sequence with a synthetic model file
(tests/synthetic_sequence.py) -- build container only.

Stand-ins for what is absent offline, each a few lines restating a published definition (they are NOT the code under test):
  cv2.Rodrigues / cv2.resize(INTER_NEAREST)      OpenCV's definitions;   cv2.imread of the .exr: returns the array the sequence builder wrote
  pytorch3d.ops.knn.knn_points                   brute force (tests/golden/make_golden.py)
  trimesh.Trimesh(...).contains                  ray casting along +X with Moeller-Trumbore, parity of the hits (the product casts along +z
                                                 per grid column with a coverage rule: two different methods must agree)
  the licensed model file                        open() of .../smpl_files/basicmodel_M_...pkl is redirected to the synthetic file
What the vectors pin: SmplModel (shape blend, joint regression, chain, LBS), the canonical pose / bounds / centre, cano2live matrices,
the 'A' position-map pose, the position-map transform (:207-213), the item dict, the valid band and the inside/outside fill.

    python tests/golden/make_golden_dataset.py      ->  tests/golden/dataset_golden.npz
"""
import builtins
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synthetic_sequence as sq             # noqa: E402


def ray_contains(verts, faces, pts):
    """+X ray, Moeller-Trumbore, odd number of hits = inside (float64)."""
    v = np.asarray(verts, np.float64); f = np.asarray(faces, np.int64)
    a, e1, e2 = v[f[:, 0]], v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]
    d = np.array([1.0, 0.0, 0.0])
    pv = np.cross(d, e2)
    det = (e1 * pv).sum(1)
    out = np.zeros(len(pts), bool)
    for s in range(0, len(pts), 512):
        p = np.asarray(pts[s:s + 512], np.float64)
        tv = p[:, None, :] - a[None]
        u = (tv * pv[None]).sum(-1) / det
        qv = np.cross(tv, e1[None])
        w = (qv * d).sum(-1) / det
        t = (qv * e2[None]).sum(-1) / det
        hit = (np.abs(det)[None] > 1e-300) & (u >= 0) & (w >= 0) & (u + w <= 1) & (t > 0)
        out[s:s + 512] = hit.sum(1) % 2 == 1
    return out


def install(td, pos_maps):
    import make_golden as mg
    # pytorch3d / skimage stubs as in make_golden, but NOT dataset.smpl (the real one must run) and a richer cv2 / trimesh
    def knn_points(p1, p2, K=1, return_nn=False, **kw):
        d = ((p1[:, :, None, :] - p2[:, None, :, :]) ** 2).sum(-1)
        dist, idx = torch.topk(d, K, dim=-1, largest=False, sorted=True)
        return dist, idx, None
    m, ops, knn = types.ModuleType('pytorch3d'), types.ModuleType('pytorch3d.ops'), types.ModuleType('pytorch3d.ops.knn')
    for mod in (ops, knn):
        mod.knn_points = knn_points
        mod.knn_gather = lambda x, idx: torch.stack([x[b][idx[b]] for b in range(idx.shape[0])], 0)
    m.ops, ops.knn = ops, knn
    sys.modules.update({'pytorch3d': m, 'pytorch3d.ops': ops, 'pytorch3d.ops.knn': knn})
    cv = types.ModuleType('cv2')
    cv.IMREAD_UNCHANGED, cv.INTER_NEAREST = -1, 0

    def Rodrigues(r):
        # cv::Rodrigues computes in double and returns a matrix of the INPUT's depth (CV_32F in -> CV_32F out): the reference's float32 pose vectors
        # (avatarcap_dataset.py:194) get R rounded to float32
        depth = np.asarray(r).dtype
        v = np.asarray(r, np.float64).reshape(3); t = np.linalg.norm(v)
        if t < 2.220446049250313e-16:
            R = np.eye(3)
        else:
            k = v / t
            K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            R = np.cos(t) * np.eye(3) + (1 - np.cos(t)) * np.outer(k, k) + np.sin(t) * K
        return (R.astype(np.float32) if depth == np.float32 else R), None

    def resize(img, dsize, interpolation=None):
        assert interpolation == cv.INTER_NEAREST
        w, h = dsize; H, W = img.shape[:2]
        ys = np.minimum((np.arange(h) * (H / h)).astype(np.int64), H - 1); xs = np.minimum((np.arange(w) * (W / w)).astype(np.int64), W - 1)
        return img[ys][:, xs]

    def imread(path, flag=None):
        return pos_maps[os.path.basename(path)].copy() if os.path.basename(path) in pos_maps else None
    cv.Rodrigues, cv.resize, cv.imread = Rodrigues, resize, imread
    sys.modules['cv2'] = cv
    tm, tmp = types.ModuleType('trimesh'), types.ModuleType('trimesh.proximity')

    class Trimesh:
        def __init__(self, vertices, faces, **kw):
            self.v, self.f = vertices, faces

        def contains(self, pts):
            return ray_contains(self.v, self.f, pts)
    tm.Trimesh, tm.proximity = Trimesh, tmp
    sys.modules.update({'trimesh': tm, 'trimesh.proximity': tmp})
    for n in ['skimage', 'skimage.measure']:
        sys.modules[n] = types.ModuleType(n)
    sys.path.insert(0, REF)
    import config
    config.device = torch.device('cpu')
    config.cfg = {'testing': {'vol_res': list(sq.VOL_RES)}, 'model': {'cano_template': {'pos_encoding': 10}, 'warping_field': {'pos_encoding': 0}},
                  'training': {}}
    pkl = os.path.join(td, 'smpl_synth.pkl')
    sq.write_smpl_file(pkl)
    real_open = builtins.open

    def patched(path, *a, **k):
        if isinstance(path, str) and path.endswith('_lbs_10_207_0_v1.0.0.pkl'):
            return real_open(pkl, *a, **k)
        return real_open(path, *a, **k)
    builtins.open = patched
    try:
        import dataset.smpl  # noqa: F401   (the reference's module-level SmplParams load)
    finally:
        builtins.open = real_open
    return config


def main():
    td = tempfile.mkdtemp()
    out = {}
    for tag, dtype_, name in (('real', 'real', 'cano'), ('syn', 'synthetic', 'A')):
        seq = os.path.join(td, tag)
        written = {}
        ids = sq.build_sequence(seq, lambda p, img: (written.__setitem__(os.path.basename(p), img), open(p, 'wb').close()), data_type=dtype_, pos_map_name=name)
        if tag == 'real':
            config = install(td, written)
            import dataset.avatarcap_dataset as ref_ds
            from dataset.avatarcap_dataset import AvatarCapDataset
            from dataset.smpl import SmplModel
            # the NeRF ray sampler of __getitem__ (needs cv2.fillPoly; main.py's test loop overwrites all of its outputs, :468-473)
            z = np.zeros((4, 3), np.float32)
            ref_ds.sample_ray_h36m = lambda *a, **k: (z, np.zeros(4, np.uint8), z, z, np.zeros(4, np.float32), np.ones(4, np.float32),
                                                     np.zeros((4, 2), np.int64), np.ones(4, bool))
        else:
            sys.modules['cv2'].imread = (lambda tbl: (lambda path, flag=None: tbl[os.path.basename(path)].copy()))(written)
        np.random.seed(5); torch.manual_seed(5)
        ds = AvatarCapDataset(seq, training=False)
        for k, idx in enumerate(ids):
            it = ds[k * ds.img_num_per_pose + (1 if tag == 'syn' else 0) * 0]
            p = f'{tag}{k}_'
            out[p + 'data_idx'] = np.int64(it['data_idx'])
            out[p + 'smpl_pos_map'] = it['smpl_pos_map'].numpy()
            out[p + 'cano2live_jnt_mats'] = it['cano2live_jnt_mats'].numpy()
            out[p + 'live_smpl_v_sample'] = it['live_smpl_v'].numpy()[::97]
            out[p + 'w2c_RT'] = np.asarray(it['w2c_RT'], np.float32)
            out[p + 'cano2posmap_jnt_mats'] = it['cano2posmap_jnt_mats'].numpy()
        out[tag + '_cano_bounds'] = ds.cano_bounds
        out[tag + '_cano_smpl_center'] = ds.cano_smpl_center.numpy()
        out[tag + '_cano_smpl_jnts'] = ds.cano_smpl_jnts.numpy()
        out[tag + '_cano_smpl_v_sample'] = ds.cano_smpl_v.numpy()[::97]
        out[tag + '_infer_pts_flag'] = np.packbits(ds.infer_pts_flag.numpy())
        out[tag + '_invalid_pts_ov'] = ds.invalid_pts_ov.numpy().astype(np.int8)
        out[tag + '_start_data_idx'] = np.int64(ds.start_data_idx)
        out[tag + '_len'] = np.int64(len(ds))
        out[tag + '_K'] = ds.K
        print(tag, 'valid', int(ds.infer_pts_flag.sum()), 'of', ds.infer_pts_flag.numel(), 'inside among invalid', int((ds.invalid_pts_ov > 0).sum()))
    # SmplModel on a random pose, directly
    rs = np.random.RandomState(3)
    sm = SmplModel(np.concatenate([0.1 * rs.randn(3), 0.4 * rs.randn(72)]).astype(np.float32), 0.7 * rs.randn(10))
    out['smpl_posed_vertices_sample'] = sm.posed_vertices[::53]
    out['smpl_jnt_affine_mats'] = sm.jnt_affine_mats
    out['smpl_posed_joints'] = sm.posed_joints
    path = os.path.join(HERE, 'dataset_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes,', len(out), 'arrays')


if __name__ == '__main__':
    main()
