"""BASELINE configs[1] at its full size -- one dense 256^3 frame -- checked through properties that do not
need a full-size CPU run: sampled oracle parity, independence of the per-point result from its position in the
batch, vertex count = number of sign-changing grid edges, a closed surface away from the volume border,
unit normals, exact KNN (grid search vs exhaustive scan) and partition-of-unity skin weights."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from avatarcap_amd import _lib, config
from common import geotex_sd, maxabs

pytestmark = pytest.mark.gpu
RES = [256, 256, 256]


@pytest.fixture(scope='module')
def frame():
    from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar
    from avatarcap_amd.pipeline import FramePipeline
    config.cfg = config.default_cfg()
    config.cfg['testing']['vol_res'] = RES
    config.device = torch.device('cuda')
    ds = SyntheticTestDataset(RES, valid='dense', n_frames=1)
    net = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume()).to('cuda').eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in geotex_sd().items()})
    pipe = FramePipeline(net, ds)
    items = to_cuda(ds[0], add_batch=True)
    return pipe, ds, items, pipe.avatar_frame(items)


def test_sampled_occupancy_matches_oracle_and_is_batch_independent(frame):
    from oracle import avatarcap_oracle as orc
    pipe, ds, items, out = frame
    vol = out['occ_volume']
    assert vol.numel() == 256 ** 3 and bool(torch.isfinite(vol).all())
    rs = np.random.RandomState(11)
    sel = np.sort(rs.choice(vol.numel(), 1500, replace=False))
    pts = ds.infer_pts[torch.from_numpy(sel).cuda()].cpu().numpy()
    fmap = pipe.network.warping_field.pose_feat_map[0].cpu().numpy()
    ref = orc.occupancy_query(pts, fmap, ds.cano_smpl_center, geotex_sd())['cano_pts_ov'][:, 0]
    assert maxabs(vol.cpu().numpy()[sel], ref) < 1e-4                               # BASELINE.json: occupancy within 1e-4
    # the value of a point does not depend on which other points share its launch, tile or wave
    perm = torch.from_numpy(rs.permutation(vol.numel())[:200_003]).cuda()
    sub = dict(items); sub['cano_pts'] = ds.infer_pts[perm][None].contiguous()
    again = pipe.occ_net.query(sub)['cano_pts_ov'][0, :, 0]
    # (the dense launch of the frame is column-folded -- fp32 column terms for the 64 feature columns of conv1 / conv5 -- and agrees with the
    # point-by-point kernel to rounding; within one kernel the independence is bitwise)
    assert float((again - vol[perm]).abs().max()) < 2e-5                               # (each is within ~1e-5 of the fp64 oracle)
    back = torch.flip(perm[:70_001], [0])
    sub['cano_pts'] = ds.infer_pts[back][None].contiguous()
    assert torch.equal(pipe.occ_net.query(sub)['cano_pts_ov'][0, :, 0], torch.flip(again[:70_001], [0]))


def test_mesh_topology_at_full_size(frame):
    pipe, ds, items, out = frame
    v, f, n, vol = out['cano_v'], out['f'].long(), out['cano_vn'], out['occ_volume'].reshape(RES)
    inside = vol > config.iso_value
    crossings = sum(int((inside.narrow(a, 0, 255) != inside.narrow(a, 1, 255)).sum()) for a in range(3))
    # one vertex per sign-changing grid edge + the centre vertices of the few cells whose Lewiner tiling needs one
    assert crossings <= v.shape[0] <= crossings + max(64, crossings // 1000), (v.shape[0], crossings)
    assert int(f.min()) == 0 and int(f.max()) == v.shape[0] - 1
    assert bool((f[:, 0] != f[:, 1]).all() and (f[:, 1] != f[:, 2]).all() and (f[:, 0] != f[:, 2]).all())
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key = torch.minimum(e[:, 0], e[:, 1]) * v.shape[0] + torch.maximum(e[:, 0], e[:, 1])
    uniq, cnt = torch.unique(key, return_counts=True)
    assert int(cnt.max()) == 2                                                        # manifold: no edge shared by 3+ faces
    b0, b1 = torch.from_numpy(ds.cano_bounds[0]).cuda(), torch.from_numpy(ds.cano_bounds[1]).cuda()
    voxel = (b1 - b0) / 256
    border = ((v - b0 < 1.01 * voxel) | (b1 - v < 1.01 * voxel)).any(1)
    open_e = uniq[cnt == 1]
    assert bool(border[open_e // v.shape[0]].all() and border[open_e % v.shape[0]].all())   # open edges only on the volume border
    # opposite orientation across every shared edge: directed edges are unique
    dkey = e[:, 0] * v.shape[0] + e[:, 1]
    assert torch.unique(dkey).numel() == dkey.numel()
    nn = torch.linalg.norm(n, dim=1)
    assert float((nn - 1).abs().max()) < 1e-5


def test_lbs_at_full_vertex_count(frame, monkeypatch):
    from avatarcap_amd.utils.smpl_util import smpl_util
    pipe, ds, items, out = frame
    v = out['cano_v']
    lbs = smpl_util.calculate_lbs(v[None])
    _lib.set_option('knn_search', 3)
    lbs_b = smpl_util.calculate_lbs(v[None])
    _lib.set_option('knn_search', 0)
    assert torch.equal(lbs, lbs_b)                                                    # grid search == exhaustive scan, bit for bit
    s = lbs[0].sum(1)
    d2, _ = smpl_util.knn_points(v[None], ds.cano_smpl_v.cuda()[None], K=1)
    near = d2[0, :, 0] < 0.1                              # exp(-0.1 / 0.005) = 2e-9 >> the 1e-16 of smpl_util.py:36
    assert int(near.sum()) > 1000
    assert float((s[near] - 1).abs().max()) < 1e-5      # partition of unity where the Gaussian weights have not underflowed
    assert float(s.max()) < 1 + 1e-5                      # ... and sum / (sum + 1e-16) in [0, 1] everywhere else
    assert bool((lbs >= 0).all())
    eye = torch.eye(4, device='cuda').expand(1, 24, 4, 4).contiguous()
    same = smpl_util.skinning(v[None], lbs, eye)[0]
    assert float((same[near] - v[near]).abs().max()) < 2e-6
    assert torch.equal(out['live_v'], smpl_util.skinning(v[None], lbs, items['cano2live_jnt_mats'])[0])


@pytest.mark.parametrize('res', [[256, 256, 256], [384, 384, 128]], ids=['256x256x256', 'example.yaml 384x384x128'])
def test_full_avatarcap_frame_on_the_band(res):
    """BASELINE configs[2] at its full size, and the reference's OWN configuration (configs/example.yaml:14-17: vol_res 384 x 384 x 128, only the
    points within 0.1 m of the canonical SMPL evaluated, dataset/avatarcap_dataset.py:111-125): steps 1-3 chained on the band-masked grid.
    Sampled oracle parity of BOTH volumes -- the avatar's occupancy on the pose feature map of this frame and the reconstruction decoder's on the
    fused maps' image features --, untouched fill values outside the band, a manifold mesh that is open only where the body leaves the volume."""
    from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda, synthetic_camera, synthetic_observed_normals
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar
    from avatarcap_amd.network.arch_recon import ReconNetwork
    from avatarcap_amd.pipeline import FramePipeline
    from common import recon_sd
    from oracle import avatarcap_oracle as orc
    config.cfg = config.default_cfg()
    config.cfg['testing']['vol_res'] = res
    config.device = torch.device('cuda')
    ds = SyntheticTestDataset(res, valid='band', n_frames=1)
    net = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume()).to('cuda').eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in geotex_sd().items()})
    rn = ReconNetwork().to('cuda').eval()
    rn.load_state_dict({k: torch.from_numpy(v) for k, v in recon_sd().items()})
    pipe = FramePipeline(net, ds, rn)
    items = to_cuda(ds[0], add_batch=True)
    a = pipe.avatar_frame(items)
    flag = ds.infer_pts_flag
    assert 0.1 < float(flag.float().mean()) < 0.3
    rs = np.random.RandomState(2)
    sel = np.sort(rs.choice(ds.infer_pts.shape[0], 1500, replace=False))
    selc = torch.from_numpy(sel).cuda()
    pts = ds.infer_pts[selc].cpu().numpy()
    # 1. the avatar's band volume (main.py:360-364): avatar_kernel<warp, geometry only, column-folded band> on this frame's pose feature map
    assert torch.equal(a['occ_volume'][~flag], ds.invalid_pts_ov)                     # main.py:363
    fmap = net.warping_field.pose_feat_map[0].cpu().numpy()
    ref_a = orc.occupancy_query(pts, fmap, ds.cano_smpl_center, geotex_sd())['cano_pts_ov'][:, 0]
    err_a = maxabs(a['occ_volume'][flag][selc].cpu().numpy(), ref_a)
    w2c, cam = synthetic_camera()
    obs = synthetic_observed_normals(a['live_v'], a['live_vn'], a['f'], w2c, cam, seed=3)
    items['front_normal'], items['back_normal'], _ = pipe.fuse_normals(a, obs, w2c, cam, 'merge', iter_num=100)
    r = pipe.recon_frame(items)
    vol = r['occ_volume']
    assert torch.equal(vol[~flag], ds.invalid_pts_ov)                                 # main.py:443
    # 2. the reconstruction volume (main.py:440-443)
    with torch.no_grad():
        imap = rn.get_feat_maps(torch.cat([items['front_normal'], items['back_normal']], 1))[-1][0].cpu().numpy()
    ref = orc.recon_infer(pts, imap, ds.cano_smpl_center, recon_sd())
    err_r = maxabs(vol[flag][selc].cpu().numpy(), ref)
    print(f'vol_res {res}: {int(flag.sum())} band points, avatar occupancy vs oracle {err_a:.2e}, reconstruction occupancy vs oracle {err_r:.2e}, '
          f'{a["cano_v"].shape[0]} / {r["cano_v"].shape[0]} vertices')
    assert err_a < 1e-4 and err_r < 1e-4
    b0, b1 = torch.from_numpy(ds.cano_bounds[0]).cuda(), torch.from_numpy(ds.cano_bounds[1]).cuda()
    voxel = (b1 - b0) / torch.tensor(res, dtype=torch.float32, device='cuda')
    for mesh in (a, r):
        v, f = mesh['cano_v'], mesh['f'].long()
        assert v.shape[0] > 10000
        e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        key = torch.minimum(e[:, 0], e[:, 1]) * v.shape[0] + torch.maximum(e[:, 0], e[:, 1])
        uniq, cnt = torch.unique(key, return_counts=True)
        hist = {int(c): int((cnt == c).sum()) for c in torch.unique(cnt)}
        assert int(cnt.max()) == 2, hist                                              # manifold: no edge shared by 3+ faces
        border = ((v - b0 < 1.01 * voxel) | (b1 - v < 1.01 * voxel)).any(1)
        open_e = uniq[cnt == 1]
        assert bool(border[open_e // v.shape[0]].all() and border[open_e % v.shape[0]].all()), hist   # open only where the body leaves the volume
        assert mesh['live_v'].shape == v.shape and bool(torch.isfinite(mesh['live_v']).all())
