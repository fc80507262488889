"""GPU parity of the fused MLP queries (csrc/fused_mlp.hip) through the C-ABI, against
(a) the golden vectors produced by the reference itself and (b) the CPU oracle on fresh inputs.
Tolerance: 1e-4 absolute on occupancy / offsets (BASELINE.json north_star), fp32 in/out."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from avatarcap_amd import _lib, config, synthetic as syn
from common import geotex_sd, recon_sd, maxabs

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _t(x, dev='cuda'):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


@pytest.fixture(scope='module')
def net():
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar
    config.cfg = config.default_cfg()
    n = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume()).to('cuda').eval()
    n.load_state_dict({k: torch.from_numpy(v) for k, v in geotex_sd().items()})
    return n


def _batch(pts):
    return {'cano_pts': _t(pts[None]), 'cano_smpl_center': _t(gi.center()[None])}


def test_avatar_query_matches_reference_golden(net, golden):
    from avatarcap_amd.network.arch_avatar import OccupancyNet
    net.warping_field.pose_feat_map = _t(gi.pose_feat_map()[None])
    net.warping_field._map_on_device = None
    pts = gi.query_points(104, 2048)
    for if_type in ('sdf', 'occupancy'):
        config.if_type = if_type
        out = OccupancyNet(net).query(_batch(pts))
        assert out['cano_pts_ov'].shape == (1, 2048, 1) and out['nonrigid_offset'].shape == (1, 2048, 3)
        assert maxabs(out['cano_pts_ov'][0].cpu().numpy(), golden[f'G5_occ_{if_type}']) < TOL
        assert maxabs(out['nonrigid_offset'][0].cpu().numpy(), golden['G5_offset']) < TOL
    config.if_type = 'sdf'
    assert maxabs(net.warping_field.query(_t(pts[None]), _batch(pts))[0].cpu().numpy(), golden['G4_offset']) < TOL
    rgb, alpha, occ = net.cano_template.forward(_t(pts[None]))
    assert maxabs(rgb[0].cpu().numpy(), golden['G5_tmpl_rgb']) < TOL
    assert maxabs(alpha[0].cpu().numpy(), golden['G5_tmpl_alpha']) < TOL
    assert maxabs(occ[0].cpu().numpy(), golden['G5_tmpl_occ']) < TOL


@pytest.mark.parametrize('n', [1, 31, 128, 129, 5000, 70001])
def test_avatar_query_vs_oracle_ragged(net, n):
    """ragged sizes around the 32-point wave / 128-point tile boundaries"""
    from avatarcap_amd.network.arch_avatar import OccupancyNet
    from oracle import avatarcap_oracle as orc
    config.if_type = 'sdf'
    fmap = gi.pose_feat_map(seed=300 + n % 7)
    net.warping_field.pose_feat_map = _t(fmap[None])
    net.warping_field._map_on_device = None
    pts = gi.query_points(500 + n, n)
    out = OccupancyNet(net).query(_batch(pts))
    sel = np.arange(n) if n <= 6000 else np.sort(np.random.RandomState(n).choice(n, 6000, replace=False))
    ref = orc.occupancy_query(pts[sel], fmap, gi.center(), geotex_sd())
    assert maxabs(out['cano_pts_ov'][0].cpu().numpy()[sel], ref['cano_pts_ov']) < TOL
    assert maxabs(out['nonrigid_offset'][0].cpu().numpy()[sel], ref['nonrigid_offset']) < TOL


def test_avatar_query_empty_and_errors(net):
    from avatarcap_amd.network.arch_avatar import OccupancyNet
    from avatarcap_amd import _lib
    net.warping_field.pose_feat_map = _t(gi.pose_feat_map()[None])
    net.warping_field._map_on_device = None
    out = OccupancyNet(net).query(_batch(np.zeros((0, 3), np.float32)))
    assert out['cano_pts_ov'].shape == (1, 0, 1)
    net.warping_field.pose_feat_map = None
    with pytest.raises(AttributeError):
        OccupancyNet(net).query(_batch(gi.query_points(1, 8)))
    with pytest.raises((TypeError, RuntimeError)):
        net.warping_field.pose_feat_map = _t(gi.pose_feat_map()[None])
        OccupancyNet(net).query({'cano_pts': torch.zeros(1, 8, 3, dtype=torch.float64, device='cuda'),
                                 'cano_smpl_center': _t(gi.center()[None])})


def test_recon_decoder_matches_reference_golden(golden):
    from avatarcap_amd.network.arch_recon import ReconNetwork
    rn = ReconNetwork().to('cuda').eval()
    rn.load_state_dict({k: torch.from_numpy(v) for k, v in recon_sd().items()})
    pts = gi.query_points(104, 2048)
    y = rn.decode(_t(pts[None]), _t(gi.img_feat_map()[None]), _t(gi.center()[None]))
    assert y.shape == (1, 2048)                                     # the reference's return shape (arch_recon.py:73-76), used as output[0] (main.py:442)
    assert maxabs(y[0].cpu().numpy(), golden['G6_decoder']) < TOL
    # whole infer(): HGFilter on the HIP encoder + fused decoder
    nm = gi.normal_maps(64)
    items = {'cano_pts': _t(pts[None]), 'cano_smpl_center': _t(gi.center()[None]),
             'front_normal': _t(nm[None, :3]), 'back_normal': _t(nm[None, 3:])}
    y2 = rn.infer(items)
    assert y2.shape == golden['G6_recon'].shape
    err = maxabs(y2.cpu().numpy(), golden['G6_recon'])
    print(f'infer() vs G6_recon: {err:.3e}')
    assert err < 1e-4   # measured 8e-7 (the split-fp16 MFMA encoder vs the reference on the CPU): north_star's bar holds end to end


@pytest.mark.parametrize('n', [1, 33, 4097])
def test_recon_decoder_vs_oracle(n):
    from avatarcap_amd.network.arch_recon import ReconNetwork
    from oracle import avatarcap_oracle as orc
    rn = ReconNetwork().to('cuda').eval()
    rn.load_state_dict({k: torch.from_numpy(v) for k, v in recon_sd().items()})
    pts = gi.query_points(700 + n, n)
    imap = gi.img_feat_map(seed=210)
    y = rn.decode(_t(pts[None]), _t(imap[None]), _t(gi.center()[None]))
    assert maxabs(y.cpu().numpy().reshape(-1), orc.recon_infer(pts, imap, gi.center(), recon_sd())) < TOL


def test_avatar_query_large_preactivations(net):
    """Softplus pre-activations far beyond torch's threshold (20) and beyond 64*ln2: the log2-domain
    evaluation must keep returning x there (a clamp once silently saturated them)."""
    from avatarcap_amd.network.arch_avatar import OccupancyNet
    from oracle import avatarcap_oracle as orc
    config.if_type = 'sdf'
    fmap = (6.0 * gi.pose_feat_map(seed=321)).astype(np.float32)
    net.warping_field.pose_feat_map = _t(fmap[None])
    net.warping_field._map_on_device = None
    pts = gi.query_points(4321, 3000)
    out = OccupancyNet(net).query(_batch(pts))
    ref = orc.occupancy_query(pts, fmap, gi.center(), geotex_sd())
    scale = max(1.0, float(np.abs(ref['nonrigid_offset']).max()))
    assert maxabs(out['nonrigid_offset'][0].cpu().numpy(), ref['nonrigid_offset']) < 1e-4 * scale
    # the template is ill-conditioned in its input (2^9 positional frequency): compare it on the offsets the GPU produced
    q = pts.astype(np.float64) + out['nonrigid_offset'][0].cpu().numpy().astype(np.float64)
    _, _, occ = orc.double_tnet(q.astype(np.float32), geotex_sd(), with_colour=False)
    assert maxabs(out['cano_pts_ov'][0].cpu().numpy(), occ) < 1e-4


@pytest.mark.parametrize('per_col', [5, 6, 11, 32, 45])
def test_grid_subset_query_runs_per_wave(net, per_col):
    """The band's column terms ride the xyz k-step, six RUNS of equal adjacent columns per pass (fused_mlp.hip: ColSegs): `per_col` points of every column
    in grid order put 32 / per_col runs into a wave -- 7 (one more than a pass holds), 6, 3 - 4, 1 - 2 (a band), 1 --, a column met again later is a second
    run, and a ragged tail.  Folded vs point-by-point vs the oracle."""
    from avatarcap_amd.network.arch_avatar import OccupancyNet
    from avatarcap_amd.grid import generate_volume_points_np, volume_axes
    from oracle import avatarcap_oracle as orc
    config.if_type = 'sdf'
    res = (6, 7, 48)
    fmap = gi.pose_feat_map()
    net.warping_field.pose_feat_map = _t(fmap[None])
    allp = generate_volume_points_np(syn.CANO_BOUNDS, res)
    base = (np.arange(res[0] * res[1])[:, None] * res[2] + np.arange(per_col)[None, :] + 1).reshape(-1)
    idx = np.concatenate([base, base[:77] + 1 if per_col < 45 else base[:77]]).astype(np.int32)[:-3]      # some columns come back as later runs; ragged count
    pts = allp[idx]
    index = torch.from_numpy(idx).cuda()
    ax = volume_axes(syn.CANO_BOUNDS, res, 'cuda')
    a = OccupancyNet(net).query(_batch(pts))
    g = OccupancyNet(net).query_grid(_batch(pts), ax, res, want_offset=True, index=index)
    d_occ, d_off = maxabs(g['cano_pts_ov'].cpu().numpy(), a['cano_pts_ov'].cpu().numpy()), maxabs(g['nonrigid_offset'].cpu().numpy(), a['nonrigid_offset'].cpu().numpy())
    print(f'{per_col} points per column: folded subset vs point-by-point: occupancy {d_occ:.2e}, offsets {d_off:.2e}')
    assert 0 < d_occ < 2e-5 and d_off < 2e-5
    ref = orc.occupancy_query(pts, fmap, gi.center(), geotex_sd())
    assert maxabs(g['cano_pts_ov'][0].cpu().numpy(), ref['cano_pts_ov']) < TOL and maxabs(g['nonrigid_offset'][0].cpu().numpy(), ref['nonrigid_offset']) < TOL
    assert torch.equal(OccupancyNet(net).query_grid(_batch(pts), ax, res, index=index)['cano_pts_ov'], g['cano_pts_ov'])       # deterministic


@pytest.mark.parametrize('res', [(4, 6, 128), (3, 5, 256), (5, 4, 50)])
def test_recon_grid_query(res):
    """avc_recon_query_grid: the decoder on the dense grid without the point array.  A last axis of a multiple of 128 points is column-folded (the 32
    image-feature columns of fc0 / fc1 / fc2 as one fp32 vector per (x, y) column): ~1e-6 from the point-by-point decode of the materialised points and
    within 1e-4 of the oracle; any other last axis (and "column_fold" 0) runs the point-by-point kernel on generated points: bit-identical."""
    from avatarcap_amd.network.arch_recon import ReconNetwork
    from avatarcap_amd.grid import generate_volume_points_np, volume_axes
    from oracle import avatarcap_oracle as orc
    rn = ReconNetwork().to('cuda').eval()
    rn.load_state_dict({k: torch.from_numpy(v) for k, v in recon_sd().items()})
    imap = gi.img_feat_map(seed=211)
    pts = generate_volume_points_np(syn.CANO_BOUNDS, res)
    ax = volume_axes(syn.CANO_BOUNDS, res, 'cuda')
    a = rn.decode(_t(pts[None]), _t(imap[None]), _t(gi.center()[None]))
    g = rn.decode_grid(ax, res, _t(imap[None]), _t(gi.center()[None]))
    assert g.shape == a.shape == (1, pts.shape[0])
    d = maxabs(g.cpu().numpy(), a.cpu().numpy())
    print(f'res {res}: grid vs point-by-point decode {d:.2e}')
    if res[2] % 128 == 0:
        assert 0 < d < 2e-5                                                       # (0 < : the folded kernel really ran)
    else:
        assert torch.equal(g, a)
    assert maxabs(g.cpu().numpy().reshape(-1), orc.recon_infer(pts, imap, gi.center(), recon_sd())) < TOL
    assert torch.equal(rn.decode_grid(ax, res, _t(imap[None]), _t(gi.center()[None])), g)          # deterministic
    _lib.set_option('column_fold', 0)
    try:
        u = rn.decode_grid(ax, res, _t(imap[None]), _t(gi.center()[None]))
    finally:
        _lib.set_option('column_fold', 1)
    assert torch.equal(u, a)


def _band_like_indices(res, n, seed):
    """Flat grid indices shaped like a valid band: the first half walks the (x, y) columns in order and takes one interval of z from each (long runs of
    one column: one or two runs per 32-point wavefront), the second half is a shuffle of other grid points (a new column per point)."""
    rs = np.random.RandomState(seed)
    X, Y, Z = res
    runs = []
    while sum(len(r) for r in runs) < n // 2:
        c = len(runs)                                      # column after column, like the flat order of a band
        z0 = int(rs.randint(0, max(1, Z // 3)))
        z1 = int(rs.randint(min(Z - 1, z0 + min(20, Z - 1)), Z)) + 1
        runs.append(c * Z + np.arange(z0, z1))
        if len(runs) >= X * Y:
            break
    band = np.concatenate(runs)[: n // 2] if runs else np.zeros(0, np.int64)
    rest = np.setdiff1d(np.arange(X * Y * Z), band)
    tail = rs.choice(rest, n - band.size, replace=False)
    return np.concatenate([band, tail]).astype(np.int32)


@pytest.mark.parametrize('res,n', [((7, 9, 50), 2001), ((5, 4, 128), 1), ((3, 3, 40), 300), ((6, 5, 256), 5000)])
def test_recon_grid_subset_query(res, n):
    """avc_recon_query_grid_subset: the valid band by flat grid indices (any order, ragged counts), column-folded by runs of columns since round 5:
    recon_fold_kernel<2> on the tiles whose wavefronts hold at most two runs of columns (the band-like half of the indices), the point-by-point kernel on
    the tiles band_prepass_kernel leaves out (the shuffled half).  ~1e-6 from the point-by-point decode of the materialised points, within 1e-4 of the
    oracle, reproducible; a point's value depends on its place in the launch only through which of the two kernels evaluates its tile."""
    from avatarcap_amd.network.arch_recon import ReconNetwork
    from avatarcap_amd.grid import generate_volume_points_np, volume_axes
    from oracle import avatarcap_oracle as orc
    rn = ReconNetwork().to('cuda').eval()
    rn.load_state_dict({k: torch.from_numpy(v) for k, v in recon_sd().items()})
    imap = gi.img_feat_map(seed=212)
    allp = generate_volume_points_np(syn.CANO_BOUNDS, res)
    idx = _band_like_indices(res, n, n)
    assert idx.size == n and np.unique(idx).size == n
    pts = allp[idx]
    index = torch.from_numpy(idx).cuda()
    ax = volume_axes(syn.CANO_BOUNDS, res, 'cuda')
    a = rn.decode(_t(pts[None]), _t(imap[None]), _t(gi.center()[None]))
    g = rn.decode_grid(ax, res, _t(imap[None]), _t(gi.center()[None]), index=index)
    assert g.shape == (1, n)
    diff = (g - a).abs()[0].cpu().numpy()
    d = float(diff.max())
    e = maxabs(g.cpu().numpy().reshape(-1), orc.recon_infer(pts, imap, gi.center(), recon_sd()))
    back = rn.decode_grid(ax, res, _t(imap[None]), _t(gi.center()[None]), index=torch.flip(index, [0]).contiguous())
    f = maxabs(torch.flip(back, [1]).cpu().numpy(), g.cpu().numpy())
    print(f'res {res} n {n}: folded subset vs point-by-point decode {d:.2e} ({int((diff > 0).sum())} of {n} values differ), vs oracle {e:.2e}, '
          f'same points in reverse order {f:.2e}')
    assert d < 2e-5 and e < TOL and f < 2e-5
    if n >= 2000:
        assert (diff[: n // 4] > 0).any()                                         # the folded kernel really ran on the band-like part ...
        assert not (diff[n // 2 + 128:] > 0).any()                                # ... and the shuffled part went to the point-by-point kernel: identical bits
    assert torch.equal(rn.decode_grid(ax, res, _t(imap[None]), _t(gi.center()[None]), index=index), g)          # deterministic
    _lib.set_option('column_fold', 0)
    try:
        u = rn.decode_grid(ax, res, _t(imap[None]), _t(gi.center()[None]), index=index)
    finally:
        _lib.set_option('column_fold', 1)
    assert torch.equal(u, a)
    with pytest.raises(TypeError):
        rn.decode_grid(ax, res, _t(imap[None]), _t(gi.center()[None]), index=index.to(torch.int64))


@pytest.mark.parametrize('shape,G', [((1, 64, 128, 128), 32), ((2, 96, 17, 23), 32), ((1, 256, 16, 16), 32), ((3, 8, 5, 7), 4)])
def test_group_norm_relu_matches_torch(shape, G):
    """avc_group_norm against torch.nn.functional.group_norm (+ relu); odd sizes take the unaligned path."""
    from avatarcap_amd import _lib

    def norm_relu(m, x):                               # the stand-alone op of the C-ABI (the encoder itself fuses its GroupNorms: csrc/conv_enc.hip)
        y = torch.empty_like(x)
        N, Cc = x.shape[0], x.shape[1]
        _lib.check(_lib.lib().avc_group_norm(_lib.ctx(x.device), x.data_ptr(), N, Cc, x.numel() // (N * Cc), m.num_groups, m.weight.data_ptr(),
                                             m.bias.data_ptr(), float(m.eps), 1, y.data_ptr(), _lib.stream_ptr(x.device)))
        return y
    g = torch.Generator().manual_seed(sum(shape))
    m = torch.nn.GroupNorm(G, shape[1]).cuda()
    with torch.no_grad():
        m.weight.copy_(torch.randn(shape[1], generator=g)); m.bias.copy_(torch.randn(shape[1], generator=g))
    x = (torch.randn(shape, generator=g) * 3 + 5).cuda()                         # mean well away from 0
    with torch.no_grad():
        y = norm_relu(m, x)
        ref = torch.relu(torch.nn.functional.group_norm(x.double(), G, m.weight.double(), m.bias.double(), m.eps)).float()
    assert y.shape == x.shape and float((y - ref).abs().max()) < 5e-6
    with torch.no_grad():
        assert torch.equal(norm_relu(m, x), y)                                     # deterministic


@pytest.mark.parametrize('seed,gain', [(7, 1.6), (123, 1.0), (2024, 2.2)])
def test_avatar_query_other_weights(seed, gain):
    """Nothing in the packing (BN / log2 folds, fp16 splits, scale checks) may depend on one lucky set of weights: fresh
    networks with other seeds and gains (small to fairly ill-conditioned), occupancy + offsets + colour vs the oracle."""
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
    from common import geotex_shapes
    from oracle import avatarcap_oracle as orc
    sd = syn.synth_state_dict(geotex_shapes(), seed, gain=gain)
    net = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume()).to('cuda').eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    fmap = gi.pose_feat_map()
    net.warping_field.pose_feat_map = _t(fmap[None])
    pts = gi.query_points(300 + seed, 1500)
    out = OccupancyNet(net).query(_batch(pts))
    ref64 = orc.occupancy_query(pts, fmap, gi.center(), sd)
    ref32 = orc.occupancy_query(pts, fmap, gi.center(), sd, dt=np.float32)
    # the bar is the reference's own fp32 arithmetic: measure what fp32 itself loses on this network (fp32 vs fp64 oracle)
    slack = maxabs(ref32['cano_pts_ov'], ref64['cano_pts_ov'])
    scale = max(1.0, float(np.abs(ref64['cano_pts_ov']).max()))                  # gain 2.2 drives the SDF head to |values| ~ 100
    err = maxabs(out['cano_pts_ov'][0].cpu().numpy(), ref64['cano_pts_ov'])
    print(f'seed {seed} gain {gain}: scale {scale:.1f}, err {err:.3e}, fp32-oracle slack {slack:.3e}')
    assert err < TOL + 2 * slack      # 1e-4, plus what the reference's own fp32 arithmetic loses on an ill-conditioned network (2.6e-3 at gain 2.2)
    assert maxabs(out['nonrigid_offset'][0].cpu().numpy(), ref64['nonrigid_offset']) < TOL


def test_error_paths_of_the_side_entries():
    """Bad arguments come back as AVC_ERR_ARG with a message (the reference would raise ValueError / shape errors)."""
    from avatarcap_amd import _lib
    from avatarcap_amd.utils.renderer import render_mesh_device
    from avatarcap_amd.normal_fusion.normal_fusion import merge_normal_images_device
    config.device = torch.device('cuda')
    x = torch.randn(1, 48, 4, 4, device='cuda'); y = torch.empty_like(x)
    h = _lib.ctx(x.device)
    with pytest.raises(_lib.AvcapError, match='divisible'):
        _lib.check(_lib.lib().avc_group_norm(h, x.data_ptr(), 1, 48, 16, 32, None, None, 1e-5, 1, y.data_ptr(), None))
    v = torch.zeros(3, 3, device='cuda'); f = torch.zeros((1, 3), dtype=torch.int32, device='cuda')
    with pytest.raises(_lib.AvcapError, match='image size'):
        render_mesh_device(v, None, f, np.eye(4, dtype=np.float32), 20000, 4)
    assert float(render_mesh_device(v, None, f[:0], np.eye(4, dtype=np.float32), 8, 8).abs().max()) == 0.0      # no faces: background
    with pytest.raises(ValueError):
        merge_normal_images_device(torch.zeros(8, 8, 3, device='cuda'), torch.zeros(8, 9, 3, device='cuda'), 2, (0, 0))
    with pytest.raises(_lib.AvcapError, match='iter_num'):
        merge_normal_images_device(torch.zeros(8, 8, 3, device='cuda'), torch.zeros(8, 8, 3, device='cuda'), -1, (0, 0))
    out = torch.empty(0, 3, device='cuda')
    _lib.check(_lib.lib().avc_canonicalize_normals(h, None, None, 0, None, None, 4, 4, _lib.f3(np.eye(4, dtype=np.float32).reshape(16)), 1.0, 1.0, 0.0, 0.0,
                                                   None, None))                                                   # nv = 0 is a no-op
    sing = np.zeros(16, np.float32)
    with pytest.raises(_lib.AvcapError, match='singular'):
        _lib.check(_lib.lib().avc_canonicalize_normals(h, v.data_ptr(), torch.zeros(3, 4, 4, device='cuda').data_ptr(), 3, torch.zeros(4, 4, 4, device='cuda').data_ptr(),
                                                       torch.zeros(4, 4, 3, device='cuda').data_ptr(), 4, 4, _lib.f3(sing), 1.0, 1.0, 0.0, 0.0, torch.zeros(3, 3, device='cuda').data_ptr(), None))


def test_grid_query_equals_point_query(net):
    """avc_avatar_query_grid generates the points from the grid index (three per-axis tables built with the reference's own float32
    arithmetic): occupancy and offsets must equal the query on the materialised points BIT FOR BIT, ragged last tile included."""
    from avatarcap_amd.network.arch_avatar import OccupancyNet
    from avatarcap_amd.grid import generate_volume_points, generate_volume_points_np, volume_axes, volume_axes_np
    config.if_type = 'sdf'
    net.warping_field.pose_feat_map = _t(gi.pose_feat_map()[None])
    res = (19, 23, 30)                                     # 13,110 points: not a multiple of the 128-point tile
    ax = volume_axes_np(syn.CANO_BOUNDS, res)
    pts_np = generate_volume_points_np(syn.CANO_BOUNDS, res)
    ix, iy, iz = np.unravel_index(np.arange(pts_np.shape[0]), res)
    assert np.array_equal(pts_np, np.stack([ax[0][ix], ax[1][iy], ax[2][iz]], -1))        # the tables ARE the reference grid
    batch = _batch(pts_np)
    a = OccupancyNet(net).query(batch)
    g = OccupancyNet(net).query_grid(batch, volume_axes(syn.CANO_BOUNDS, res, 'cuda'), res, want_offset=True)
    assert torch.equal(a['cano_pts_ov'], g['cano_pts_ov']) and torch.equal(a['nonrigid_offset'], g['nonrigid_offset'])
    g2 = OccupancyNet(net).query_grid(batch, volume_axes(syn.CANO_BOUNDS, res, 'cuda'), res)
    assert 'nonrigid_offset' not in g2 and torch.equal(g2['cano_pts_ov'], a['cano_pts_ov'])
    with pytest.raises(ValueError):
        OccupancyNet(net).query_grid(batch, volume_axes(syn.CANO_BOUNDS, (19, 23, 31), 'cuda'), res)


@pytest.mark.parametrize('res', [(3, 5, 128), (2, 3, 256), (1, 1, 384)])
def test_column_folded_grid_query(net, res, monkeypatch):
    """A dense grid whose last axis holds a multiple of 128 points is evaluated column-folded: the 64 pose-feature columns of conv1 / conv5 enter
    as one fp32 vector per (x, y) column instead of through the split-fp16 products (fused_mlp.hip, column_terms_kernel).  Same algebra, other
    rounding: a few 1e-6 from the point-by-point query (each ~1e-5 from the fp64 oracle), bit-identical to it when the folding is switched off."""
    from avatarcap_amd.network.arch_avatar import OccupancyNet
    from avatarcap_amd.grid import generate_volume_points_np, volume_axes
    from oracle import avatarcap_oracle as orc
    config.if_type = 'sdf'
    fmap = gi.pose_feat_map()
    net.warping_field.pose_feat_map = _t(fmap[None])
    pts_np = generate_volume_points_np(syn.CANO_BOUNDS, res)
    batch = _batch(pts_np)
    a = OccupancyNet(net).query(batch)
    g = OccupancyNet(net).query_grid(batch, volume_axes(syn.CANO_BOUNDS, res, 'cuda'), res, want_offset=True)
    d_occ, d_off = maxabs(g['cano_pts_ov'].cpu().numpy(), a['cano_pts_ov'].cpu().numpy()), maxabs(g['nonrigid_offset'].cpu().numpy(), a['nonrigid_offset'].cpu().numpy())
    print(f'res {res}: folded vs point-by-point: occupancy {d_occ:.2e}, offsets {d_off:.2e}')
    assert 0 < d_occ < 2e-5 and d_off < 2e-5                                   # (0 < : the folded path really ran; each path is ~1e-5 from fp64)
    ref = orc.occupancy_query(pts_np, fmap, gi.center(), geotex_sd())
    assert maxabs(g['cano_pts_ov'][0].cpu().numpy(), ref['cano_pts_ov']) < TOL and maxabs(g['nonrigid_offset'][0].cpu().numpy(), ref['nonrigid_offset']) < TOL
    g2 = OccupancyNet(net).query_grid(batch, volume_axes(syn.CANO_BOUNDS, res, 'cuda'), res)
    assert torch.equal(g2['cano_pts_ov'], g['cano_pts_ov'])                    # deterministic, with or without the offsets written
    _lib.set_option('column_fold', 0)
    try:
        u = OccupancyNet(net).query_grid(batch, volume_axes(syn.CANO_BOUNDS, res, 'cuda'), res, want_offset=True)
    finally:
        _lib.set_option('column_fold', 1)
    assert torch.equal(u['cano_pts_ov'], a['cano_pts_ov']) and torch.equal(u['nonrigid_offset'], a['nonrigid_offset'])


@pytest.mark.parametrize('res,n', [((7, 9, 50), 2001), ((5, 4, 128), 1), ((3, 3, 40), 360)])
def test_grid_subset_query(net, res, n, monkeypatch):
    """avc_avatar_query_grid_subset: the valid band by its flat grid indices (any order, ragged counts).  Column-folded with per-lane column blocks:
    a few 1e-6 from the point-by-point query of the same points and within 1e-4 of the oracle; with the folding off, bit for bit the point query."""
    from avatarcap_amd.network.arch_avatar import OccupancyNet
    from avatarcap_amd.grid import generate_volume_points_np, volume_axes
    from oracle import avatarcap_oracle as orc
    config.if_type = 'sdf'
    fmap = gi.pose_feat_map()
    net.warping_field.pose_feat_map = _t(fmap[None])
    allp = generate_volume_points_np(syn.CANO_BOUNDS, res)
    rs = np.random.RandomState(n)
    idx = rs.choice(allp.shape[0], n, replace=False).astype(np.int32)
    if n > 100:
        idx[: n // 2] = np.sort(idx[: n // 2])                                  # half in grid order (runs inside columns, like a band), half scattered
    pts = allp[idx]
    index = torch.from_numpy(idx).cuda()
    ax = volume_axes(syn.CANO_BOUNDS, res, 'cuda')
    a = OccupancyNet(net).query(_batch(pts))
    g = OccupancyNet(net).query_grid(_batch(pts), ax, res, want_offset=True, index=index)
    assert g['cano_pts_ov'].shape == (1, n, 1) and g['nonrigid_offset'].shape == (1, n, 3)
    d_occ, d_off = maxabs(g['cano_pts_ov'].cpu().numpy(), a['cano_pts_ov'].cpu().numpy()), maxabs(g['nonrigid_offset'].cpu().numpy(), a['nonrigid_offset'].cpu().numpy())
    print(f'res {res} n {n}: folded subset vs point-by-point: occupancy {d_occ:.2e}, offsets {d_off:.2e}')
    assert d_occ < 2e-5 and d_off < 2e-5
    ref = orc.occupancy_query(pts, fmap, gi.center(), geotex_sd())
    assert maxabs(g['cano_pts_ov'][0].cpu().numpy(), ref['cano_pts_ov']) < TOL and maxabs(g['nonrigid_offset'][0].cpu().numpy(), ref['nonrigid_offset']) < TOL
    # a point's value does not depend on which points share its launch: the same indices reversed
    back = OccupancyNet(net).query_grid(_batch(pts), ax, res, index=torch.flip(index, [0]).contiguous())
    assert torch.equal(torch.flip(back['cano_pts_ov'], [1]), g['cano_pts_ov'])
    _lib.set_option('column_fold', 0)
    try:
        u = OccupancyNet(net).query_grid(_batch(pts), ax, res, want_offset=True, index=index)
    finally:
        _lib.set_option('column_fold', 1)
    assert torch.equal(u['cano_pts_ov'], a['cano_pts_ov']) and torch.equal(u['nonrigid_offset'], a['nonrigid_offset'])
    with pytest.raises(TypeError):
        OccupancyNet(net).query_grid(_batch(pts), ax, res, index=index.to(torch.int64))


def test_range_check_trips_on_fp16_overflow():
    """config.check_range -> avc_set_range_check: the same network with its warp MLP scaled until a post-activation value leaves
    the fp16 range must raise AVC_ERR_RANGE; the unscaled network must pass the check with bit-identical outputs."""
    from avatarcap_amd import _lib
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
    from common import geotex_shapes
    config.if_type = 'sdf'
    pts = gi.query_points(77, 3000)
    fmap = _t(gi.pose_feat_map()[None])

    def run(sd, check):
        n = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume()).to('cuda').eval()
        n.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        n.warping_field.pose_feat_map = fmap
        config.check_range = check
        try:
            return OccupancyNet(n).query(_batch(pts))
        finally:
            config.check_range = False

    sd = geotex_sd()
    plain, checked = run(sd, False), run(sd, True)
    assert torch.equal(plain['cano_pts_ov'], checked['cano_pts_ov']) and torch.equal(plain['nonrigid_offset'], checked['nonrigid_offset'])
    big = dict(sd)
    for k in ('warping_field.mlp.conv1.weight', 'warping_field.mlp.conv1.bias'):
        big[k] = sd[k] * np.float32(3000.0)                  # Softplus is ~linear for large inputs: conv2's outputs pass 65504
    for k in ('warping_field.mlp.conv2.weight',):
        big[k] = sd[k] * np.float32(30.0)
    with pytest.raises(_lib.AvcapError) as ei:
        run(big, True)
    assert ei.value.status == _lib.AVC_ERR_RANGE and '65504' in str(ei.value)
    out = run(big, False)                                    # unchecked: no error is raised -- which is why the check exists
    assert out['cano_pts_ov'].shape == (1, 3000, 1)


def test_two_networks_on_one_device_keep_their_own_weights():
    """A context holds one packed network; modules cache 'already packed'.  Alternating two GeoTexAvatar / ReconNetwork instances
    with different weights must re-pack instead of evaluating with the other's weights (pose map binding likewise)."""
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
    from avatarcap_amd.network.arch_recon import ReconNetwork
    from common import geotex_shapes
    config.if_type = 'sdf'
    pts = gi.query_points(5, 700)
    nets, maps = [], [_t(gi.pose_feat_map()[None]), _t(gi.pose_feat_map(seed=99)[None])]
    for seed in (gi.SEED_NET, 4242):
        n = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume()).to('cuda').eval()
        n.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(geotex_shapes(), seed).items()})
        nets.append(n)
    for n, m in zip(nets, maps):
        n.warping_field.pose_feat_map = m
    first = [OccupancyNet(n).query(_batch(pts))['cano_pts_ov'].clone() for n in nets]
    assert not torch.equal(first[0], first[1])
    for _ in range(2):
        for n, ref in zip(nets, first):
            assert torch.equal(OccupancyNet(n).query(_batch(pts))['cano_pts_ov'], ref)
    rns = []
    for seed in (gi.SEED_NET, 777):
        rn = ReconNetwork().to('cuda').eval()
        rn.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(syn.module_shapes(rn), seed).items()})
        rns.append(rn)
    imap = _t(gi.img_feat_map()[None])
    ys = [rn.decode(_t(pts[None]), imap, _t(gi.center()[None])).clone() for rn in rns]
    assert not torch.equal(ys[0], ys[1])
    for rn, ref in zip(rns, ys):
        assert torch.equal(rn.decode(_t(pts[None]), imap, _t(gi.center()[None])), ref)


def test_context_options_and_launch_clock(net):
    """avc_set_option: unknown names and out-of-range values are AVC_ERR_ARG (nothing is read from the environment after avc_ctx_create);
    avc_timing_read_cycles: the shader cycles of the timed launches divided by their device time is a plausible gfx950 shader clock."""
    import ctypes as C
    from avatarcap_amd.network.arch_avatar import OccupancyNet
    from avatarcap_amd.grid import volume_axes
    ctx = _lib.ctx(torch.device('cuda', 0))
    for name, value in (('no_such_option', 1), ('column_fold', 2), ('knn_search', 4), ('mlp_blocks', -1), ('fusion_graph', 7)):
        with pytest.raises(_lib.AvcapError) as e:
            _lib.set_option(name, value)
        assert e.value.status == -1
    for name, value in (('column_fold', 1), ('knn_search', 0), ('mlp_blocks', 0), ('fusion_graph', 1)):
        _lib.set_option(name, value)
    res = (8, 8, 128)
    net.warping_field.pose_feat_map = _t(gi.pose_feat_map()[None])
    batch = {'cano_smpl_center': _t(gi.center()[None])}
    ax = volume_axes(syn.CANO_BOUNDS, res, 'cuda')
    q = OccupancyNet(net)
    q.query_grid(batch, ax, res); torch.cuda.synchronize()
    _lib.check(_lib.lib().avc_timing_enable(ctx, 1))
    for _ in range(3):
        full = q.query_grid(batch, ax, res)
    _lib.set_option('mlp_blocks', 16)                       # fewer persistent workgroups: same bits
    few = q.query_grid(batch, ax, res)
    _lib.set_option('mlp_blocks', 0)
    torch.cuda.synchronize()
    ms, nl, cyc, nc = C.c_double(), C.c_int64(), C.c_double(), C.c_int64()
    _lib.check(_lib.lib().avc_timing_read(ctx, 0, C.byref(ms), C.byref(nl), 1))
    _lib.check(_lib.lib().avc_timing_read_cycles(ctx, 0, C.byref(cyc), C.byref(nc)))
    _lib.check(_lib.lib().avc_timing_enable(ctx, 0))
    assert nl.value == 4 and nc.value == 4 and ms.value > 0
    mhz = cyc.value / (ms.value * 1e3)
    print(f'launch clock: {cyc.value:.3e} cycles / {ms.value:.3f} ms = {mhz:.0f} MHz')
    assert 500 < mhz < 3000                                 # (the column pass is inside the event pair, not inside the cycle count: a lower bound of the clock)
    assert torch.equal(full['cano_pts_ov'], few['cano_pts_ov'])


@pytest.mark.parametrize('blocks', [248, 5])
def test_queries_do_not_depend_on_the_workgroup_count(net, blocks):
    """Multi-rank runs give the persistent query kernels `CUs - 8` workgroups (parallel.leave_cus_for_the_exchange; avc_set_option "mlp_blocks"), which
    does not divide the tile count: every launch form -- dense and band, avatar and recon, folded -- must return the bits of the default launch."""
    from avatarcap_amd.network.arch_avatar import OccupancyNet
    from avatarcap_amd.network.arch_recon import ReconNetwork
    from avatarcap_amd.grid import volume_axes
    config.if_type = 'sdf'
    net.warping_field.pose_feat_map = _t(gi.pose_feat_map(seed=77)[None])
    net.warping_field._map_on_device = None
    rn = ReconNetwork().to('cuda').eval()
    rn.load_state_dict({k: torch.from_numpy(v) for k, v in recon_sd().items()})
    imap = _t(gi.img_feat_map(seed=78)[None])
    res = (9, 7, 256)                                                              # 126 tiles
    ax = volume_axes(syn.CANO_BOUNDS, res, 'cuda')
    idx = torch.from_numpy(_band_like_indices(res, 6000, 5)).cuda()
    items = {'cano_smpl_center': _t(gi.center()[None])}

    def run():
        q = OccupancyNet(net)
        return (q.query_grid(items, ax, list(res))['cano_pts_ov'], q.query_grid(items, ax, list(res), index=idx)['cano_pts_ov'],
                rn.decode_grid(ax, res, imap, items['cano_smpl_center']), rn.decode_grid(ax, res, imap, items['cano_smpl_center'], index=idx))
    want = run()
    _lib.set_option('mlp_blocks', blocks)
    try:
        got = run()
    finally:
        _lib.set_option('mlp_blocks', 0)
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_band_fuzz_folded_launches_against_the_point_kernels(monkeypatch):
    """tests/tools/band_fuzz_gpu.py, 120 cases: random grid shapes and index lists -- bands, one to five points per column (more runs in a wavefront than a pass holds),
    scattered, the whole grid; reversed, shuffled, with columns met again and duplicated indices, ragged counts; one workgroup per CU or 1 / 3 / 7 persistent ones walking many
    tiles each -- through both folded subset queries and the dense launches against the point-by-point kernels (2e-5 / 5e-6), the folding switched off bit for bit, the same
    indices reversed bit for bit (avatar) / 2e-6 (recon).  (The round's campaign: 1,800 cases, no failure.)"""
    import importlib.util
    import os
    import sys
    spec = importlib.util.spec_from_file_location('band_fuzz_gpu', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'band_fuzz_gpu.py'))
    mod = importlib.util.module_from_spec(spec)
    dev, cfg, ift = config.device, config.cfg, config.if_type
    try:
        spec.loader.exec_module(mod)
        monkeypatch.setattr(sys, 'argv', ['band_fuzz_gpu.py', '120', '5'])
        assert mod.main() == 0
    finally:
        config.device, config.cfg, config.if_type = dev, cfg, ift
        _lib.set_option('mlp_blocks', 0); _lib.set_option('column_fold', 1)
