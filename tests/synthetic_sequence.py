"""A synthetic captured sequence in the reference's on-disk layout (DATA.md / dataset/avatarcap_dataset.py:26-60) and a synthetic SMPL model
file of the licensed file's layout (dataset/smpl.py:14-43), regenerated from seeds by the golden generator (build container) and by the tests.
Nothing here is the licensed model: a deformed lat-long sphere with SMPL's vertex / face counts, random blend shapes, a sparse joint regressor
and the published 24-joint kinematic tree."""
import os
import pickle

import numpy as np

SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
VOL_RES = [22, 26, 12]
POS_MAP_SRC = (48, 96)          # (H, 2H) of the EXR on disk; pos_map_res 32 -> nearest resize to (32, 64)
POS_MAP_RES = 32


def synthetic_smpl_dict(seed=77):
    """6890 vertices / 13776 faces (a closed lat-long sphere with 84 rings of 82 has exactly SMPL's counts), body-ish ellipsoid."""
    from scipy import sparse
    rs = np.random.RandomState(seed)
    nl, nn = 84, 82
    th = np.linspace(0, np.pi, nl + 2)[1:-1]
    ph = np.linspace(0, 2 * np.pi, nn, endpoint=False)
    unit = [[0, 1, 0]] + [[np.sin(t) * np.cos(p), np.cos(t), np.sin(t) * np.sin(p)] for t in th for p in ph] + [[0, -1, 0]]
    v = np.array(unit) * np.array([0.28, 0.85, 0.16]) + np.array([0.0, -0.1, 0.0])
    f = []
    for j in range(nn):
        f.append([0, 1 + (j + 1) % nn, 1 + j])
    for i in range(nl - 1):
        for j in range(nn):
            a, b = 1 + i * nn + j, 1 + i * nn + (j + 1) % nn
            f += [[a, b, a + nn], [b, b + nn, a + nn]]
    last = len(v) - 1
    for j in range(nn):
        f.append([last, 1 + (nl - 1) * nn + j, 1 + (nl - 1) * nn + (j + 1) % nn])
    v, f = v.astype(np.float64), np.array(f, np.uint32)
    assert v.shape == (6890, 3) and f.shape == (13776, 3)
    # 24 joints spread along the body; regressor rows = normalised random weights on the 40 nearest vertices
    jpos = np.stack([rs.uniform(-0.2, 0.2, 24), np.linspace(0.6, -0.85, 24), rs.uniform(-0.05, 0.05, 24)], 1)
    rows, cols, vals = [], [], []
    for j in range(24):
        idx = np.argsort(((v - jpos[j]) ** 2).sum(1))[:40]
        w = rs.uniform(0.1, 1, 40); w /= w.sum()
        rows += [j] * 40; cols += idx.tolist(); vals += w.tolist()
    J_reg = sparse.csc_matrix((vals, (rows, cols)), shape=(24, 6890))
    d2 = ((v[:, None, :] - jpos[None]) ** 2).sum(-1)
    w = np.exp(-d2 / 0.02); w /= w.sum(1, keepdims=True)
    return {'v_template': v, 'f': f, 'J': np.asarray(J_reg * v), 'kintree_table': np.array([[(p if p >= 0 else 2 ** 32 - 1) for p in SMPL_PARENTS], list(range(24))], np.int64),
            'weights': w, 'J_regressor': J_reg, 'shapedirs': 0.01 * rs.randn(6890, 3, 10)}


def write_smpl_file(path, seed=77):
    with open(path, 'wb') as fh:
        pickle.dump(synthetic_smpl_dict(seed), fh, protocol=2)


def pos_map_array(data_idx, src=None):
    """(H, 2H, 3) float32 'position map' as it sits in the EXR: smooth + noise, front | back halves different."""
    rs = np.random.RandomState(500 + data_idx)
    H, W = src or POS_MAP_SRC
    ys, xs = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing='ij')
    m = np.stack([np.sin(3 * xs) + 0.1 * rs.randn(H, W), ys * xs + 0.1 * rs.randn(H, W), np.cos(2 * ys) + 0.1 * rs.randn(H, W)], -1)
    return m.astype(np.float32)


def build_sequence(root, write_exr, n_frames=2, start=3, data_type='real', pos_map_name='cano', pos_map_res=POS_MAP_RES, pos_map_src=None):
    """Writes <root>/{dataConfig.yaml, smpl/shape.txt, smpl/pose_%04d.txt, smpl/smpl_pos_map_%04d_<name>.exr[, imgs/%03d/cams.mat]} and returns
    the list of data indices.  `write_exr(path, img_bgr_float32)` is the caller's EXR writer."""
    import yaml
    os.makedirs(os.path.join(root, 'smpl'), exist_ok=True)
    with open(os.path.join(root, 'dataConfig.yaml'), 'w', encoding='UTF-8') as fh:
        yaml.safe_dump({'data_type': data_type, 'pos_map_name': pos_map_name, 'pos_map_res': pos_map_res,
                        'camera': {'fx': 550.0, 'fy': 552.0, 'cx': 255.5, 'cy': 254.0, 'img_width': 512, 'img_height': 512}}, fh)
    rs = np.random.RandomState(91)
    np.savetxt(os.path.join(root, 'smpl', 'shape.txt'), 0.5 * rs.randn(10))
    ids = []
    for k in range(n_frames):
        idx = start + k
        pose = np.concatenate([0.05 * rs.randn(3), 0.25 * rs.randn(72)])
        np.savetxt(os.path.join(root, 'smpl', 'pose_%04d.txt' % idx), pose)
        write_exr(os.path.join(root, 'smpl', 'smpl_pos_map_%04d_%s.exr' % (idx, pos_map_name)), pos_map_array(idx, pos_map_src))
        if data_type == 'synthetic':
            import scipy.io as sio
            os.makedirs(os.path.join(root, 'imgs', '%03d' % idx), exist_ok=True)
            sio.savemat(os.path.join(root, 'imgs', '%03d' % idx, 'cams.mat'), {'cam_rs': 0.3 * rs.randn(2, 3), 'cam_ts': rs.randn(2, 3)})
        ids.append(idx)
    return ids
