"""Pins the CPU oracle (oracle/avatarcap_oracle.py) to outputs of the reference itself
(tests/golden/reference_golden.npz, produced by tests/golden/make_golden.py from /root/reference).
CPU only.  Tolerances: the reference runs fp32, the oracle fp64 => a few 1e-6 relative."""
import numpy as np
import pytest

import golden_inputs as gi
from avatarcap_amd import synthetic as syn
from avatarcap_amd.grid import generate_volume_points_np, linspace01_f32
from oracle import avatarcap_oracle as orc
from common import geotex_sd, recon_sd, mlp_sd, offset_decoder_sd, maxabs


def rel(a, b):
    return maxabs(a, b) / max(1e-12, float(np.max(np.abs(b))))


def test_G0_grid(golden):
    assert np.array_equal(linspace01_f32(17), golden['G0_lin17'])
    assert np.array_equal(linspace01_f32(256), golden['G0_lin256'])
    for name, res in (('toy', (4, 3, 2)), ('odd', (5, 7, 6))):
        assert np.array_equal(generate_volume_points_np(syn.CANO_BOUNDS, res), golden[f'G0_{name}_pts'])


def test_G1_embedder(golden):
    x = gi.points(101, 256)
    assert maxabs(orc.embed(x.astype(np.float64), 10), golden['G1_embed10']) < 5e-5   # sin(512 x) in fp32 vs fp64
    assert np.array_equal(orc.embed(x, 0), golden['G1_embed0'])
    # same arithmetic type as the reference => tight
    assert maxabs(orc.embed(x, 10), golden['G1_embed10']) < 5e-7


@pytest.mark.parametrize('name', list(gi.MLP_CONFIGS))
def test_G2_mlp(golden, name):
    c = gi.MLP_CONFIGS[name]
    k = c['kwargs']
    x = gi.features(102, 300, k['in_channels'])
    y = orc.mlp_forward(x, mlp_sd(name), '', c['n_layers'], tuple(k['res_layers']), k['nlactv'], k['last_op'])
    assert y.shape == golden[f'G2_{name}'].shape
    assert rel(y, golden[f'G2_{name}']) < 2e-5


def test_G3_offset_decoder(golden):
    y = orc.offset_decoder(gi.features(103, 300, 67), offset_decoder_sd(), '')
    assert rel(y, golden['G3_offset_decoder']) < 2e-5


def test_G4_G5_avatar_query(golden):
    sd, fmap, pts, c = geotex_sd(), gi.pose_feat_map(), gi.query_points(104, 2048), gi.center()
    off = orc.warping_query(pts, fmap, c, sd)
    assert maxabs(off, golden['G4_offset']) < 2e-5 * max(1.0, float(np.abs(golden['G4_offset']).max()))
    for if_type in ('sdf', 'occupancy'):
        o = orc.occupancy_query(pts, fmap, c, sd, if_type)
        g = golden[f'G5_occ_{if_type}']
        assert np.abs(g).max() > 0.05, 'vacuous fixture'
        assert maxabs(o['cano_pts_ov'], g) < 1e-4
        assert maxabs(o['nonrigid_offset'], golden['G5_offset']) < 1e-4
    rgb, alpha, occ = orc.double_tnet(pts, sd)
    assert maxabs(rgb, golden['G5_tmpl_rgb']) < 1e-4
    assert maxabs(alpha, golden['G5_tmpl_alpha']) < 1e-4
    assert maxabs(occ, golden['G5_tmpl_occ']) < 1e-4
    gpts = generate_volume_points_np(syn.CANO_BOUNDS, (64, 64, 64))[gi.grid_subset(64 ** 3, 1500)]
    assert maxabs(orc.occupancy_query(gpts, fmap, c, sd)['cano_pts_ov'], golden['G5_grid64_sel_occ']) < 1e-4


@pytest.mark.parametrize('variant', range(len(gi.POSENC_VARIANTS)))
def test_posenc_variants_of_the_avatar_query(variant):
    """model.cano_template.pos_encoding / model.warping_field.pos_encoding other than the example's (10, 0): the oracle against the imported reference built
    with those keys (tests/golden/make_golden_posenc.py; arch_avatar.py:33-36, 97-100, 122)."""
    import os
    from common import geotex_sd_posenc
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'posenc_golden.npz'))
    lt, lw = gi.POSENC_VARIANTS[variant]
    assert g['variants'].tolist()[variant] == [lt, lw]
    tag = f'T{lt}_W{lw}'
    sd, fmap, pts, c = geotex_sd_posenc(lt, lw), gi.pose_feat_map(), gi.query_points(130 + variant, 1024), gi.center()
    assert sd['cano_template.shared_mlp.fc_list.0.0.weight'].shape[1] == 3 + 6 * lt and sd['warping_field.mlp.conv1.weight'].shape[1] == 3 + 6 * lw + 64
    off = orc.warping_query(pts, fmap, c, sd, pos_encoding=lw)
    assert maxabs(off, g[tag + '_offset']) < 2e-5 * max(1.0, float(np.abs(g[tag + '_offset']).max()))
    o = orc.occupancy_query(pts, fmap, c, sd, 'sdf', tmpl_pos_encoding=lt, warp_pos_encoding=lw)
    assert np.abs(g[tag + '_occ']).max() > 0.05, 'vacuous fixture'
    assert maxabs(o['cano_pts_ov'], g[tag + '_occ']) < 1e-4
    rgb, alpha, occ = orc.double_tnet(pts, sd, pos_encoding=lt)
    assert maxabs(rgb, g[tag + '_tmpl_rgb']) < 1e-4 and maxabs(alpha, g[tag + '_tmpl_alpha']) < 1e-4 and maxabs(occ, g[tag + '_tmpl_occ']) < 1e-4


def test_G6_recon_decoder(golden):
    y = orc.recon_infer(gi.query_points(104, 2048), gi.img_feat_map(), gi.center(), recon_sd())
    assert maxabs(y, golden['G6_decoder']) < 2e-5
    # full infer with the reference's own HGFilter output as the feature map
    y2 = orc.recon_infer(gi.query_points(104, 2048), golden['G6_img_feat'], gi.center(), recon_sd())
    assert maxabs(y2, golden['G6_recon'][0] if golden['G6_recon'].ndim == 2 else golden['G6_recon']) < 2e-5


def test_G8_lbs(golden, body):
    vp = gi.surface_points(105, 700, body)
    lbs = orc.calculate_lbs(vp, body['cano_smpl_v'], body['skin_weights'])
    assert maxabs(lbs, golden['G8_lbs']) < 2e-6
    jm = syn.random_pose_jnt_mats(gi.SEED_POSE)
    live, mats = orc.skinning(vp, golden['G8_lbs'], jm)
    assert maxabs(live, golden['G8_live']) < 2e-6
    assert maxabs(mats, golden['G8_mats']) < 2e-6
    nrm = gi.unit_vectors(106, 700)
    assert maxabs(orc.skinning_normal(nrm, golden['G8_lbs'], jm), golden['G8_live_normals']) < 2e-6


def test_G9_normals(golden):
    vol, voxel = gi.sdf_volume(32)
    nv = orc.extract_normal_volume(vol, voxel)
    assert rel(nv[::5, ::5, ::5], golden['G9_normal_volume_slice']) < 1e-5
    n = orc.extract_normal_from_volume(vol, voxel, gi.grid_points_m11(107, 400))
    assert maxabs(n, golden['G9_normals']) < 2e-5


def test_G10_blend_weights(golden):
    w = orc.cano_blend_weight_volume(gi.blend_weight_volume(), gi.points01(108, 300))
    assert maxabs(w, golden['G10_blend_w']) < 2e-6


def test_G11_raw2outputs(golden):
    raw, zv = gi.raw_and_z(109, 50, 64)
    r = orc.raw2outputs(raw, zv)
    for k, v in zip(('rgb_map', 'disp_map', 'acc_map', 'weights', 'depth_map'), r):
        assert rel(v, golden[f'G11_{k}']) < 1e-5, k


def test_G12_geotex_forward_cano(golden, body):
    wp = gi.surface_points(110, 600, body) + 0.01 * gi.unit_vectors(111, 600)
    dists = np.full((600, 1), 0.0016, np.float32)
    raw, occ, off = orc.geotex_forward_cano(wp, dists, gi.pose_feat_map(), gi.center(), syn.CANO_BOUNDS,
                                            body['cano_smpl_v'], geotex_sd())
    assert maxabs(raw, golden['G12_raw']) < 1e-4
    assert maxabs(occ, golden['G12_occ']) < 1e-4
    assert maxabs(off, golden['G12_off']) < 1e-4


def test_G13_geotex_forward_posed(golden, body):
    jm = syn.random_pose_jnt_mats(gi.SEED_POSE + 1, sigma=0.15)
    live_v = gi.live_smpl_vertices(body, jm)
    wl = gi.live_query_points(112, 500, live_v)
    raw, occ, off, _ = orc.geotex_forward_posed(wl, np.full((500, 1), 0.0016, np.float32), gi.pose_feat_map(), gi.center(), syn.CANO_BOUNDS,
                                                live_v, body['skin_weights'], gi.blend_weight_volume(), jm, geotex_sd())
    assert maxabs(raw, golden['G13_raw']) < 2e-4
    assert maxabs(occ, golden['G13_occ']) < 2e-4
    assert maxabs(off, golden['G13_off']) < 1e-4


def test_torch_cpu_restatement_matches_numpy_oracle():
    """oracle/torch_cpu.py (what bench.py's cpu_baseline times) against the NumPy oracle on the golden inputs."""
    import torch
    from oracle import avatarcap_oracle as orc, torch_cpu
    from common import geotex_sd
    sd = geotex_sd()
    tsd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items() if v.dtype == np.float32}
    pts, fmap, c = gi.query_points(104, 1500), gi.pose_feat_map(), gi.center()
    ref = orc.occupancy_query(pts, fmap, c, sd)
    for if_type in ('sdf', 'occupancy'):
        occ, off = torch_cpu.occupancy_query(torch.from_numpy(pts), torch.from_numpy(fmap), torch.from_numpy(c), tsd, if_type, chunk=700)
        want = ref['cano_pts_ov'] if if_type == 'sdf' else 1 / (1 + np.exp(-ref['cano_pts_ov']))
        assert np.abs(occ.numpy() - want).max() < 2e-5 and np.abs(off.numpy() - ref['nonrigid_offset']).max() < 1e-5
    body = syn.synthetic_body()
    vp = gi.surface_points(105, 500, body)
    lbs = torch_cpu.calculate_lbs(torch.from_numpy(vp), torch.from_numpy(body['cano_smpl_v']), torch.from_numpy(body['skin_weights']))
    assert np.abs(lbs.numpy() - orc.calculate_lbs(vp, body['cano_smpl_v'], body['skin_weights'])).max() < 1e-6


def test_torch_cpu_unet_and_normals_match_the_goldens(golden):
    """oracle/torch_cpu.py's U-Net and vertex normals (what bench.py's cpu_baseline times since round 5) against the reference's own outputs:
    G7 (UnetNoCond7DS at 128^2, sampled pixels) and G9 (extract_normal_from_volume on an analytic SDF volume)."""
    import torch
    from oracle import torch_cpu
    from avatarcap_amd.network.unets import UnetNoCond7DS
    shapes = syn.module_shapes(UnetNoCond7DS(input_nc=6, output_nc=64, nf=32))                 # the stand-alone module of the golden: its own key names
    sd = {'u.' + k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in syn.synth_state_dict(shapes, gi.SEED_NET).items()}
    y = torch_cpu.unet7ds(sd, torch.from_numpy(gi.pos_map(128)[None]), prefix='u').numpy()[0]
    assert y.shape == (64, 128, 128)
    assert maxabs(y[:, gi.PIX[:, 0] % 128, gi.PIX[:, 1] % 128], golden['G7_unet_samples']) < 2e-5
    vol, voxel = gi.sdf_volume(32)
    gp = gi.grid_points_m11(107, 400)
    n = torch_cpu.vertex_normals(torch.from_numpy(vol), voxel, torch.from_numpy(gp)).numpy()
    assert maxabs(n, golden['G9_normals']) < 2e-5


def test_host_raw2outputs_is_the_references(golden):
    """avatarcap_amd/utils/nerf_util.py::raw2outputs (the torch compositor of the posed / temp colour branches, and the yardstick of the device compositor):
    the reference's own outputs G11 (nerf_util.py:185-212), bit for bit -- the exclusive running product is written as a shifted inclusive one, same values."""
    import torch
    from avatarcap_amd.utils.nerf_util import raw2outputs
    raw, zv = gi.raw_and_z(109, 50, 64)
    out = raw2outputs(torch.from_numpy(raw), torch.from_numpy(zv))
    for k, v in zip(('rgb_map', 'disp_map', 'acc_map', 'weights', 'depth_map'), out):
        assert np.array_equal(v.numpy(), golden['G11_' + k]), k
    white = raw2outputs(torch.from_numpy(raw), torch.from_numpy(zv), white_bkgd=True)[0].numpy()
    assert np.array_equal(white, golden['G11_rgb_map'] + (1.0 - golden['G11_acc_map'])[:, None])
