"""`model.cano_template.pos_encoding` / `model.warping_field.pos_encoding` other than configs/example.yaml's 10 / 0 (VERDICT round 4, row b+): the reference sizes
the first layer of DoubleTNet / WarpingField (and the res-concat layers shared.4 / conv5) from the two keys (network/arch_avatar.py:33-36, 97-100, 122;
utils/net_util.py:40-55).  The HIP path packs any value 0 .. 10 (csrc/pack.cpp, mlp_layout.h) and runs a warping field with an encoding in front through
avatar_kernel<.., WPE> (csrc/fused_mlp.hip).  Goldens: the imported reference built with those keys (tests/golden/make_golden_posenc.py)."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi
from avatarcap_amd import _lib, config, synthetic as syn
from common import geotex_sd_posenc, maxabs

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _t(x, dev='cuda'):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def _net(lt, lw):
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar
    config.cfg = config.default_cfg()
    config.cfg['model']['cano_template']['pos_encoding'], config.cfg['model']['warping_field']['pos_encoding'] = lt, lw
    try:
        n = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume()).to('cuda').eval()
    finally:
        config.cfg = config.default_cfg()
    n.load_state_dict({k: torch.from_numpy(v) for k, v in geotex_sd_posenc(lt, lw).items()})
    return n


@pytest.mark.parametrize('variant', range(len(gi.POSENC_VARIANTS)))
def test_avatar_query_with_other_positional_encodings(variant):
    from avatarcap_amd.network.arch_avatar import OccupancyNet
    from avatarcap_amd.grid import generate_volume_points_np, volume_axes
    from oracle import avatarcap_oracle as orc
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'posenc_golden.npz'))
    lt, lw = gi.POSENC_VARIANTS[variant]
    tag = f'T{lt}_W{lw}'
    net = _net(lt, lw)
    fmap = gi.pose_feat_map()
    net.warping_field.pose_feat_map = _t(fmap[None])
    net.warping_field._map_on_device = None
    pts = gi.query_points(130 + variant, 1024)
    batch = {'cano_pts': _t(pts[None]), 'cano_smpl_center': _t(gi.center()[None])}
    config.if_type = 'sdf'
    out = OccupancyNet(net).query(batch)
    e_occ = maxabs(out['cano_pts_ov'][0].cpu().numpy(), g[tag + '_occ'])
    e_off = maxabs(net.warping_field.query(_t(pts[None]), batch)[0].cpu().numpy(), g[tag + '_offset'])
    rgb, alpha, occ = net.cano_template.forward(_t(pts[None]))
    e_t = max(maxabs(rgb[0].cpu().numpy(), g[tag + '_tmpl_rgb']), maxabs(alpha[0].cpu().numpy(), g[tag + '_tmpl_alpha']), maxabs(occ[0].cpu().numpy(), g[tag + '_tmpl_occ']))
    print(f'pos_encoding template {lt} / warp {lw}: occupancy {e_occ:.2e}, offsets {e_off:.2e}, template heads {e_t:.2e} from the reference')
    assert e_occ < TOL and e_off < TOL and e_t < TOL
    # fresh, ragged input against the oracle
    p2 = gi.query_points(900 + variant, 4097)
    o2 = OccupancyNet(net).query({'cano_pts': _t(p2[None]), 'cano_smpl_center': _t(gi.center()[None])})
    sd = geotex_sd_posenc(lt, lw)
    ref = orc.occupancy_query(p2, fmap, gi.center(), sd, tmpl_pos_encoding=lt, warp_pos_encoding=lw)
    assert maxabs(o2['cano_pts_ov'][0].cpu().numpy(), ref['cano_pts_ov']) < TOL and maxabs(o2['nonrigid_offset'][0].cpu().numpy(), ref['nonrigid_offset']) < TOL
    # the warped query WITH the colour head (avatar_kernel<true, true, 0, WPE>: what the colour path launches): offsets against the oracle's, the template's three heads
    # against the oracle evaluated at the points the GPU warped to (the template is steep in its input; cf. test_avatar_query_large_preactivations)
    occ3, off3, rgba3 = net._avatar_query(_t(p2[None]), {'cano_smpl_center': _t(gi.center()[None])}, want_offset=True, want_rgba=True)
    # (same warp arithmetic as the geometry kernel: same offsets; its occupancy goes through shared.6 unfolded, cf. pack.cpp: other rounding)
    assert torch.equal(off3, o2['nonrigid_offset']) and float((occ3 - o2['cano_pts_ov']).abs().max()) < 2e-5
    q3 = (p2.astype(np.float64) + off3[0].cpu().numpy().astype(np.float64)).astype(np.float32)
    rgb_o, alpha_o, occ_o = orc.double_tnet(q3, sd, pos_encoding=lt)
    e_c = max(maxabs(rgba3[0, :, :3].cpu().numpy(), rgb_o), maxabs(rgba3[0, :, 3:].cpu().numpy(), alpha_o), maxabs(occ3[0].cpu().numpy(), occ_o))
    print(f'pos_encoding template {lt} / warp {lw}: warped colour query (rgb, sigma, occ) {e_c:.2e} from the oracle')
    assert e_c < TOL
    # the grid entry points: a warping field with an encoding stays point by point (bit-identical to the point query on the materialised points);
    # without one (lw == 0) the launch is column-folded whatever the template's encoding (~1e-6)
    res = (6, 5, 128)
    gp = generate_volume_points_np(syn.CANO_BOUNDS, res)
    ax = volume_axes(syn.CANO_BOUNDS, res, 'cuda')
    items = {'cano_pts': _t(gp[None]), 'cano_smpl_center': _t(gi.center()[None])}
    a = OccupancyNet(net).query(items)['cano_pts_ov']
    d = OccupancyNet(net).query_grid(items, ax, list(res))['cano_pts_ov']
    idx = torch.from_numpy(np.sort(np.random.RandomState(variant).choice(gp.shape[0], 1500, replace=False)).astype(np.int32)).cuda()
    b = OccupancyNet(net).query_grid(items, ax, list(res), index=idx)['cano_pts_ov']
    if lw > 0:
        assert torch.equal(d, a) and torch.equal(b[0], a[0][idx.long()])
    else:
        assert 0 < float((d - a).abs().max()) < 2e-5 and float((b[0] - a[0][idx.long()]).abs().max()) < 2e-5
    assert maxabs(d[0].cpu().numpy(), orc.occupancy_query(gp, fmap, gi.center(), sd, tmpl_pos_encoding=lt, warp_pos_encoding=lw)['cano_pts_ov']) < TOL


def test_pos_encoding_beyond_ten_octaves_is_refused():
    """The kernels evaluate ten octaves (2^0 .. 2^9, the example's 10): a larger value fails loudly at pack time -- no silent truncation."""
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
    config.cfg = config.default_cfg()
    config.cfg['model']['cano_template']['pos_encoding'] = 11
    try:
        n = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume()).to('cuda').eval()
    finally:
        config.cfg = config.default_cfg()
    n.warping_field.pose_feat_map = _t(gi.pose_feat_map()[None])
    with pytest.raises(_lib.AvcapError, match='0 .. 10'):
        OccupancyNet(n).query({'cano_pts': _t(gi.query_points(1, 64)[None]), 'cano_smpl_center': _t(gi.center()[None])})
