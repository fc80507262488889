"""Weights beyond the fp16 range of the split kernels (round 5 refused any layer with |w| > 3e4).  csrc/pack.cpp now
  * rebalances out-of-range rows of the ReLU / LeakyReLU / linear layers (cano_template, the recon decoder) against the matching input columns of their
    consumers -- exact powers of two, invisible to the kernels: a network whose huge row is compensated by a tiny column gives what the network without
    either gives (to the 1e-8 of an fp16-subnormal `lo` half);
  * gives the warping field's seven Conv1d + BatchNorm1d + Softplus layers one power-of-two scale and runs them on the `scaled` build of the kernels
    (fused_mlp.hip AVC_LAYER_SCALE), which undoes it in front of the Softplus: a BatchNorm row folded to |w| ~ 1e5 is evaluated to the 1e-4 bar;
  * still refuses what has no remedy: an out-of-range head layer."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from avatarcap_amd import _lib, config
from common import geotex_sd, recon_sd, maxabs

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to('cuda')


def _net(sd):
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar
    config.cfg = config.default_cfg()
    config.if_type = 'sdf'
    n = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume()).to('cuda').eval()
    n.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    n.warping_field.pose_feat_map = _t(gi.pose_feat_map()[None])
    n.warping_field._map_on_device = None
    return n


def _query(net, pts):
    from avatarcap_amd.network.arch_avatar import OccupancyNet
    out = OccupancyNet(net).query({'cano_pts': _t(pts[None]), 'cano_smpl_center': _t(gi.center()[None])})
    return out['cano_pts_ov'][0].clone(), out['nonrigid_offset'][0].clone()


def test_template_rows_out_of_range_are_rebalanced_exactly():
    pts = gi.query_points(911, 3000)
    base = dict(geotex_sd())
    occ0, off0 = _query(_net(base), pts)
    sd = dict(base)
    k2, k3 = 'cano_template.shared_mlp.fc_list.2.0', 'cano_template.shared_mlp.fc_list.3.0'
    g0, g5 = 'cano_template.geo_mlp.fc_list.0.0', 'cano_template.shared_mlp.fc_list.5.0'
    w2, b2, w3 = sd[k2 + '.weight'].copy(), sd[k2 + '.bias'].copy(), sd[k3 + '.weight'].copy()
    for row, e in ((17, 21), (200, 24)):                       # channel 17 carries 2^21, channel 200 2^24 too much; the next layer's columns undo it
        w2[row] *= np.float32(2.0 ** e); b2[row] *= np.float32(2.0 ** e); w3[:, row] *= np.float32(2.0 ** -e)
    sd[k2 + '.weight'], sd[k2 + '.bias'], sd[k3 + '.weight'] = w2, b2, w3
    # ... and across the linear shared.6 that the geometry stream folds into geo.0: shared.5 row -> shared.6 column
    k6 = 'cano_template.shared_mlp.fc_list.6'
    w5, b5, w6 = sd[g5 + '.weight'].copy(), sd[g5 + '.bias'].copy(), sd[k6 + '.weight'].copy()
    w5[3] *= np.float32(2.0 ** 22); b5[3] *= np.float32(2.0 ** 22); w6[:, 3] *= np.float32(2.0 ** -22)
    sd[g5 + '.weight'], sd[g5 + '.bias'], sd[k6 + '.weight'] = w5, b5, w6
    assert float(np.abs(w2).max()) > 1e5 and float(np.abs(w5).max()) > 1e5          # round 5 refused this checkpoint
    occ1, off1 = _query(_net(sd), pts)
    # the same function; not always the same bits -- the row comes back to within a factor 2 of where it was, and a weight's `lo` half below 2^-14 is an fp16
    # subnormal with a fixed 2^-24 spacing, so halving a weight does not halve its split exactly (the differences are of the order 1e-8)
    assert float((occ1 - occ0).abs().max()) < 2e-6 and float((off1 - off0).abs().max()) < 2e-6
    from oracle import avatarcap_oracle as orc
    ref = orc.occupancy_query(pts[:800], gi.pose_feat_map(), gi.center(), sd)
    assert maxabs(occ1.cpu().numpy()[:800], ref['cano_pts_ov']) < TOL


def test_recon_decoder_weight_norm_gain_out_of_range():
    from avatarcap_amd.network.arch_recon import ReconNetwork
    pts = gi.query_points(912, 2000)
    imap = gi.img_feat_map(seed=210)

    def run(sd):
        rn = ReconNetwork().to('cuda').eval()
        rn.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
        return rn.decode(_t(pts[None]), _t(imap[None]), _t(gi.center()[None]))[0].clone()
    base = dict(recon_sd())
    y0 = run(base)
    sd = dict(base)
    g, b = sd['image_decoder.fc_list.1.0.weight_g'].copy(), sd['image_decoder.fc_list.1.0.bias'].copy()
    v2 = sd['image_decoder.fc_list.2.0.weight_v'].copy()
    g[40] *= np.float32(2.0 ** 20); b[40] *= np.float32(2.0 ** 20)                  # W = g v / |v|: a gain of a million on one output channel of fc1 ...
    v2[:, 40] *= np.float32(2.0 ** -20)                                             # ... that fc2 all but ignores (its weight_norm renormalises the rows: not the base network)
    sd['image_decoder.fc_list.1.0.weight_g'], sd['image_decoder.fc_list.1.0.bias'], sd['image_decoder.fc_list.2.0.weight_v'] = g, b, v2
    y1 = run(sd)
    from oracle import avatarcap_oracle as orc
    ref = orc.recon_infer(pts, imap, gi.center(), sd)
    assert np.isfinite(y1.cpu().numpy()).all() and 0.02 < float(y1.mean()) < 0.98 and float(y1.std()) > 0.01      # not a saturated sigmoid
    assert maxabs(y1.cpu().numpy(), ref) < TOL
    assert float((y1 - y0).abs().max()) < 0.2


def test_warping_field_batchnorm_fold_out_of_range_runs_on_the_scaled_kernels():
    """A BatchNorm1d channel with a tiny running variance and a large gamma folds to |w| ~ 1e5 (eps = 1e-5 caps 1 / sqrt(var + eps) at 316, so this takes
    gamma ~ 4e3): its own Softplus output is far in the linear region or dead, the rest of the layer must come out as before.  The dead form keeps every
    activation inside the fp16 range, so the comparison with the oracle is meaningful: <= 1e-4."""
    pts = gi.query_points(913, 3000)
    sd = dict(geotex_sd())
    for layer, ch in ((3, 77), (6, 5)):
        gam, bet, var = sd[f'warping_field.mlp.bn{layer}.weight'].copy(), sd[f'warping_field.mlp.bn{layer}.bias'].copy(), sd[f'warping_field.mlp.bn{layer}.running_var'].copy()
        gam[ch] = np.float32(6000.0); var[ch] = np.float32(1e-7); bet[ch] = np.float32(-1.0e8)      # folded row: w x 6000 x 315 ~ 1e5; pre-activation -1e8 +- 4e6: Softplus -> 0
        sd[f'warping_field.mlp.bn{layer}.weight'], sd[f'warping_field.mlp.bn{layer}.bias'], sd[f'warping_field.mlp.bn{layer}.running_var'] = gam, bet, var
    net = _net(sd)
    occ, off = _query(net, pts)
    from oracle import avatarcap_oracle as orc
    ref = orc.occupancy_query(pts[:1000], gi.pose_feat_map(), gi.center(), sd)
    e_occ, e_off = maxabs(occ.cpu().numpy()[:1000], ref['cano_pts_ov']), maxabs(off.cpu().numpy()[:1000], ref['nonrigid_offset'])
    print(f'scaled kernels vs oracle: occupancy {e_occ:.2e}, offsets {e_off:.2e}')
    assert e_occ < TOL and e_off < TOL
    # the grid entry points (column-folded streams) and the range-checking build take the scale too
    from avatarcap_amd.grid import volume_axes
    from avatarcap_amd.network.arch_avatar import OccupancyNet
    res = (8, 8, 128)
    axes = volume_axes(np.float32([[-0.5, -0.6, -0.2], [0.5, 0.6, 0.2]]), res, 'cuda')
    g = OccupancyNet(net).query_grid({'cano_smpl_center': _t(gi.center()[None])}, axes, res)['cano_pts_ov'][0, :, 0]
    gp = torch.stack(torch.meshgrid(*axes, indexing='ij'), -1).reshape(-1, 3)
    pq = OccupancyNet(net).query({'cano_pts': gp[None].contiguous(), 'cano_smpl_center': _t(gi.center()[None])})['cano_pts_ov'][0, :, 0]
    assert float((g - pq).abs().max()) < 2e-5
    config.check_range = True
    try:
        occ_c, _ = _query(net, pts[:512])
    finally:
        config.check_range = False
    assert float((occ_c - occ[:512]).abs().max()) < 1e-6
    # the unscaled checkpoint still runs the default kernels: same bits as ever (the fixture of test_gpu_query.py covers the goldens)
    occ_b, _ = _query(_net(dict(geotex_sd())), pts[:512])
    assert not torch.equal(occ_b, occ[:512])


def test_an_out_of_range_head_is_still_refused():
    sd = dict(geotex_sd())
    w = sd['cano_template.geo_mlp.fc_list.1.weight'].copy()
    w[0] *= np.float32(1e7)
    sd['cano_template.geo_mlp.fc_list.1.weight'] = w
    with pytest.raises(_lib.AvcapError, match='output layer'):
        _query(_net(sd), gi.query_points(914, 64))
