"""Seeded inputs shared by tests/golden/make_golden.py (which feeds them to the imported reference)
and the tests (which feed them to the oracle / the HIP path).  Pure NumPy; nothing is read from
/root/reference."""
import numpy as np

from avatarcap_amd import synthetic as syn

SEED_MLP = 4242
SEED_NET = syn.SEED
SEED_POSE = 77

MLP_CONFIGS = {
    # the four MLP instantiations on the path (arch_avatar.py:37-58, arch_recon.py:19-39)
    'shared': dict(kwargs=dict(in_channels=63, out_channels=256, inter_channels=[256] * 6, res_layers=[4],
                               nlactv='relu', last_op=None, norm=None), n_layers=7),
    'geo': dict(kwargs=dict(in_channels=256, out_channels=2, inter_channels=[128], res_layers=[],
                            nlactv='leaky_relu', last_op=None, norm=None), n_layers=2),
    'clr': dict(kwargs=dict(in_channels=256, out_channels=3, inter_channels=[256, 128], res_layers=[],
                            nlactv='relu', last_op=None, norm=None), n_layers=3),
    'recon': dict(kwargs=dict(in_channels=33, out_channels=1, inter_channels=[512, 256, 128], res_layers=[1, 2],
                              nlactv='leaky_relu', last_op='sigmoid', norm='weight'), n_layers=4),
}

PIX = np.random.RandomState(9).randint(0, 4096, (24, 2))


def points(seed, n):
    return np.random.RandomState(seed).uniform(-1, 1, (n, 3)).astype(np.float32)


def features(seed, n, c):
    return np.random.RandomState(seed).randn(n, c).astype(np.float32)


def center():
    return np.array([0.0, -0.075, 0.0125], np.float32)


def query_points(seed, n):
    """Points inside the synthetic canonical bounds, a few outside the feature map's footprint
    (exercises the 'border' clamp of grid_sample)."""
    rs = np.random.RandomState(seed)
    b = syn.CANO_BOUNDS
    p = rs.uniform(b[0], b[1], (n, 3))
    p[: n // 16] *= 1.3
    return p.astype(np.float32)


def grid_subset(total, n):
    return np.sort(np.random.RandomState(31).choice(total, n, replace=False))


def pose_feat_map(seed=201):
    """(64,256,256) stand-in for UnetNoCond7DS(smpl_pos_map): smooth + noise, O(1)."""
    rs = np.random.RandomState(seed)
    ys, xs = np.meshgrid(np.linspace(0, 1, 256), np.linspace(0, 1, 256), indexing='ij')
    m = np.empty((64, 256, 256), np.float32)
    for c in range(64):
        fx, fy, ph = rs.uniform(0.5, 6, 3)
        m[c] = np.sin(2 * np.pi * (fx * xs + fy * ys) + ph) + 0.25 * rs.randn(256, 256)
    return m


def img_feat_map(seed=202):
    rs = np.random.RandomState(seed)
    ys, xs = np.meshgrid(np.linspace(0, 1, 256), np.linspace(0, 1, 256), indexing='ij')
    m = np.empty((32, 256, 256), np.float32)
    for c in range(32):
        fx, fy, ph = rs.uniform(0.5, 6, 3)
        m[c] = np.cos(2 * np.pi * (fx * xs + fy * ys) + ph) + 0.25 * rs.randn(256, 256)
    return m


def pos_map(res, seed=203):
    return np.random.RandomState(seed).uniform(-1, 1, (6, res, res)).astype(np.float32)


def normal_maps(res, seed=204):
    return syn.smooth_normal_maps(seed, res)


def blend_weight_volume(seed=205):
    w = np.random.RandomState(seed).rand(12, 10, 6, 24).astype(np.float32)
    return w / w.sum(-1, keepdims=True)


def surface_points(seed, n, body):
    rs = np.random.RandomState(seed)
    v = body['cano_smpl_v'][rs.choice(body['cano_smpl_v'].shape[0], n, replace=False)]
    return (v + rs.uniform(-0.02, 0.02, v.shape)).astype(np.float32)


def unit_vectors(seed, n):
    v = np.random.RandomState(seed).randn(n, 3)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def sdf_volume(res):
    """Positive-inside ellipsoid 'sdf' on a res^3 grid + its voxel size (anisotropic on purpose)."""
    g = [np.linspace(-1, 1, res, dtype=np.float32)] * 3
    x, y, z = np.meshgrid(*g, indexing='ij')
    vol = (0.55 - np.sqrt((x / 1.0) ** 2 + (y / 0.8) ** 2 + (z / 0.6) ** 2) * 0.7).astype(np.float32)
    voxel = np.array([0.02, 0.03, 0.05], np.float32)
    return vol, voxel


def grid_points_m11(seed, n):
    p = np.random.RandomState(seed).uniform(-1.05, 1.05, (n, 3))
    return p.astype(np.float32)


def points01(seed, n):
    return np.random.RandomState(seed).uniform(-0.05, 1.05, (n, 3)).astype(np.float32)


def raw_and_z(seed, rays, samples):
    rs = np.random.RandomState(seed)
    raw = rs.rand(rays, samples, 4).astype(np.float32)
    raw[..., 3] *= 0.3
    z = np.sort(rs.uniform(0.9, 1.1, (rays, samples)), -1).astype(np.float32)
    return raw, z


def live_smpl_vertices(body, jnt_mats):
    """Posed synthetic-SMPL vertices: plain LBS of the canonical vertices with their own skin weights."""
    M = np.einsum('nj,jxy->nxy', body['skin_weights'].astype(np.float64), jnt_mats.astype(np.float64))
    v = body['cano_smpl_v'].astype(np.float64)
    return (np.einsum('nxy,ny->nx', M[:, :3, :3], v) + M[:, :3, 3]).astype(np.float32)


def live_query_points(seed, n, live_v):
    rs = np.random.RandomState(seed)
    v = live_v[rs.choice(live_v.shape[0], n, replace=False)]
    return (v + rs.uniform(-0.03, 0.03, v.shape)).astype(np.float32)


def ply_mesh():
    rs = np.random.RandomState(14)
    v = rs.randn(7, 3).astype(np.float32); n = unit_vectors(15, 7)
    f = rs.randint(0, 7, (5, 3)).astype(np.int32); c = rs.rand(7, 3).astype(np.float32) * 0.99
    return v, f, n, c


# (cano_template.pos_encoding, warping_field.pos_encoding) pairs of tests/golden/make_golden_posenc.py -- none of them configs/example.yaml's (10, 0)
POSENC_VARIANTS = [(6, 4), (0, 0), (10, 10), (3, 1)]
