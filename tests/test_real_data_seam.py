"""The real-data seam of `main.py -m test`: dataConfig.yaml, the SMPL model, the pose / shape /
position-map files, the item dict -- `avatarcap_amd.avatarcap_dataset.AvatarCapDataset` against goldens produced by the REFERENCE's own dataset
class on the same synthetic sequence (tests/golden/make_golden_dataset.py), and the EXR reader against a file written by the OpenEXR library.
CPU only."""
import os

import numpy as np
import pytest
import torch

import synthetic_sequence as sq
from avatarcap_amd import config
from common import maxabs
from test_host import _write_exr

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, 'golden', 'dataset_golden.npz'))


def _exr(path, img):
    _write_exr(path, img[..., ::-1].copy(), 'RGB', 2, 3)            # the array is in cv.imread's B,G,R order; FLOAT, ZIP


@pytest.fixture(scope='module')
def smpl_params(tmp_path_factory):
    from avatarcap_amd.smpl import SmplParams
    p = str(tmp_path_factory.mktemp('smpl') / 'basicmodel_M_lbs_10_207_0_v1.0.0.pkl')
    sq.write_smpl_file(p)
    return SmplParams(p)


def test_smpl_model_matches_reference(smpl_params):
    from avatarcap_amd.smpl import SmplModel
    rs = np.random.RandomState(3)
    sm = SmplModel(np.concatenate([0.1 * rs.randn(3), 0.4 * rs.randn(72)]).astype(np.float32), 0.7 * rs.randn(10), smpl_params)
    assert maxabs(sm.posed_vertices[::53], G['smpl_posed_vertices_sample']) < 1e-6
    assert maxabs(sm.jnt_affine_mats, G['smpl_jnt_affine_mats']) < 1e-6
    assert maxabs(sm.posed_joints, G['smpl_posed_joints']) < 1e-6


@pytest.mark.parametrize('tag,data_type,name', [('real', 'real', 'cano'), ('syn', 'synthetic', 'A')])
def test_dataset_matches_reference(tmp_path, smpl_params, tag, data_type, name):
    from avatarcap_amd.avatarcap_dataset import AvatarCapDataset
    config.cfg = config.default_cfg()
    config.cfg['testing']['vol_res'] = list(sq.VOL_RES)
    ids = sq.build_sequence(str(tmp_path), _exr, data_type=data_type, pos_map_name=name)
    ds = AvatarCapDataset(str(tmp_path), training=False, smpl_params=smpl_params, device='cpu')
    assert ds.start_data_idx == int(G[tag + '_start_data_idx']) and len(ds) == int(G[tag + '_len'])
    assert ds.data_config['camera']['fx'] == 550.0 and ds.pos_map_res == sq.POS_MAP_RES and ds.data_type == data_type
    assert np.array_equal(ds.K, G[tag + '_K'])
    assert maxabs(ds.cano_bounds, G[tag + '_cano_bounds']) < 1e-6 and maxabs(ds.cano_smpl_center.numpy(), G[tag + '_cano_smpl_center']) < 1e-6
    assert maxabs(ds.cano_smpl_jnts.numpy(), G[tag + '_cano_smpl_jnts']) < 1e-6
    assert maxabs(ds.cano_smpl_v.numpy()[::97], G[tag + '_cano_smpl_v_sample']) < 1e-6
    flag = np.unpackbits(G[tag + '_infer_pts_flag'])[:ds.infer_pts_flag.numel()].astype(bool)
    got = ds.infer_pts_flag.numpy()
    assert (got != flag).sum() <= 2                                   # d^2 < 0.01 at float32 rounding distance from the threshold
    if (got != flag).sum() == 0:
        # inside / outside fill of the skipped points: the reference's trimesh.contains was stood in for by a +X Moeller-Trumbore ray cast,
        # the product uses +z column parity with a coverage rule -- two different methods, same answer
        assert np.array_equal(ds.invalid_pts_ov.numpy().astype(np.int8), G[tag + '_invalid_pts_ov'])
    for k, idx in enumerate(ids):
        it = ds[k]
        p = f'{tag}{k}_'
        assert it['data_idx'] == int(G[p + 'data_idx']) == idx
        assert it['smpl_pos_map'].shape == (6, sq.POS_MAP_RES, sq.POS_MAP_RES)
        assert np.array_equal(it['smpl_pos_map'].numpy(), G[p + 'smpl_pos_map'])                   # EXR -> nearest resize -> half split, exact
        assert maxabs(it['cano2live_jnt_mats'].numpy(), G[p + 'cano2live_jnt_mats']) < 2e-6
        assert maxabs(it['live_smpl_v'].numpy()[::97], G[p + 'live_smpl_v_sample']) < 1e-6
        assert maxabs(it['w2c_RT'], G[p + 'w2c_RT']) < 1e-6
        assert maxabs(it['cano2posmap_jnt_mats'].numpy(), G[p + 'cano2posmap_jnt_mats']) < 2e-6
        for key in ('cano_pts', 'valid_pts_flag', 'cano_bounds', 'cano_smpl_center', 'live_smpl_v', 'smpl_pose'):
            assert key in it


def test_missing_files_raise_like_the_reference(tmp_path, smpl_params):
    from avatarcap_amd.avatarcap_dataset import AvatarCapDataset
    from avatarcap_amd import smpl
    config.cfg = config.default_cfg()
    config.cfg['testing']['vol_res'] = list(sq.VOL_RES)
    with pytest.raises(FileNotFoundError):                            # no dataConfig.yaml (avatarcap_dataset.py:32)
        AvatarCapDataset(str(tmp_path), smpl_params=smpl_params, device='cpu')
    with pytest.raises(FileNotFoundError):                            # the licensed model file is not shipped (dataset/smpl.py:17,48)
        smpl.load_smpl_params(root=str(tmp_path / 'nowhere'))
    sq.build_sequence(str(tmp_path), _exr)
    os.remove(str(tmp_path / 'smpl' / 'shape.txt'))
    with pytest.raises((FileNotFoundError, OSError)):
        AvatarCapDataset(str(tmp_path), smpl_params=smpl_params, device='cpu')
    with pytest.raises(NotImplementedError):
        AvatarCapDataset(str(tmp_path), training=True, smpl_params=smpl_params)


def test_exr_reader_against_a_file_written_by_openexr():
    """tests/golden/openexr_sample.exr is CPython's Lib/test/imghdrdata/python.exr (16x16 RGBA, HALF, written by the OpenEXR library, PSF
    licence); openexr_sample.ppm is the same image from the same directory as 8-bit RGB.  cv.imread order = B, G, R, A."""
    from avatarcap_amd.utils.exr_io import read_exr
    img = read_exr(os.path.join(HERE, 'golden', 'openexr_sample.exr'))
    raw = open(os.path.join(HERE, 'golden', 'openexr_sample.ppm'), 'rb').read().split(b'\n', 3)
    assert raw[0] == b'P6' and raw[1] == b'16 16' and raw[2] == b'255'
    px = np.frombuffer(raw[3], np.uint8)[:16 * 16 * 3].reshape(16, 16, 3).astype(np.float64) / 255.0
    assert img.shape == (16, 16, 4) and img.dtype == np.float32
    assert maxabs(img[..., [2, 1, 0]], px) < 5e-4                      # half precision of values in [0, 1]
    assert set(np.unique(img[..., 3])) <= {0.0, 1.0} or img[..., 3].max() <= 1.0


def test_cv_compat_definitions():
    from avatarcap_amd.utils.cv_compat import resize_nearest, rodrigues
    a = np.arange(5 * 7 * 2, dtype=np.float32).reshape(5, 7, 2)
    r = resize_nearest(a, (3, 2))                                      # (w, h) like cv.resize
    assert r.shape == (2, 3, 2) and np.array_equal(r[1, 2], a[2, 4]) and np.array_equal(r[0, 0], a[0, 0])    # floor(1*5/2)=2, floor(2*7/3)=4
    up = resize_nearest(a, (14, 10))
    assert np.array_equal(up[::2, ::2], a) and np.array_equal(up[1::2, 1::2], a)
    R = rodrigues(np.array([0.0, 0.0, np.pi / 2]))
    assert np.allclose(R, [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-12) and np.allclose(rodrigues(np.zeros(3)), np.eye(3))
    w = np.array([0.3, -0.2, 0.5]); R = rodrigues(w)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12 and np.allclose(R @ w, w)
