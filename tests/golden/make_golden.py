#!/usr/bin/env python3
"""Generates the golden fixtures tests/golden/*.npz by IMPORTING THE REFERENCE (/root/reference).

Runs only in the build container (the GPU box has no /root/reference).  Only seeds, small inputs
and expected outputs are stored: weights are re-created on both sides from
avatarcap_amd.synthetic.synth_state_dict(seed) and big inputs from tests/golden_inputs.py.

The reference is imported unmodified, with stand-ins for what is absent offline
(SURVEY.md section 8(c)): pytorch3d (brute-force KNN), cv2 / skimage / trimesh (empty modules),
dataset.smpl (the licensed SMPL pickle), config.device = cpu.

    python tests/golden/make_golden.py
"""
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

from avatarcap_amd import synthetic as syn   # noqa: E402
import golden_inputs as gi                   # noqa: E402


def install_stubs():
    def knn_points(p1, p2, K=1, return_nn=False, **kw):
        d = ((p1[:, :, None, :] - p2[:, None, :, :]) ** 2).sum(-1)
        dist, idx = torch.topk(d, K, dim=-1, largest=False, sorted=True)
        return dist, idx, None

    def knn_gather(x, idx):
        return torch.stack([x[b][idx[b]] for b in range(idx.shape[0])], 0)

    m, ops, knn = types.ModuleType('pytorch3d'), types.ModuleType('pytorch3d.ops'), types.ModuleType('pytorch3d.ops.knn')
    for mod in (ops, knn):
        mod.knn_points, mod.knn_gather = knn_points, knn_gather
    m.ops, ops.knn = ops, knn
    sys.modules.update({'pytorch3d': m, 'pytorch3d.ops': ops, 'pytorch3d.ops.knn': knn})
    for n in ['cv2', 'skimage', 'skimage.measure', 'trimesh', 'trimesh.proximity']:
        sys.modules[n] = types.ModuleType(n)
    sys.modules['skimage'].measure = sys.modules['skimage.measure']
    body = syn.synthetic_body()
    ds = types.ModuleType('dataset.smpl')
    ds.smpl_params = types.SimpleNamespace(weights=body['skin_weights'])
    ds.SmplModel = object   # only the static generate_volume_points of the dataset module is used
    sys.path.insert(0, REF)
    import config
    config.device = torch.device('cpu')
    import dataset  # noqa: F401  (namespace package of the reference)
    sys.modules['dataset.smpl'] = ds
    td = tempfile.mkdtemp()
    np.save(td + '/cano_base_blend_weight_volume.npy', gi.blend_weight_volume())
    config.cfg = {'model': {'cano_template': {'pos_encoding': 10}, 'warping_field': {'pos_encoding': 0}},
                  'training': {'training_data_dir': td}}
    return config, body


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def main():
    config, body = install_stubs()
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    from network.mlp import MLP, OffsetDecoder
    from network.arch_avatar import GeoTexAvatar, OccupancyNet, CanoBlendWeightVolume
    from network.arch_recon import ReconNetwork
    from network.unets import UnetNoCond7DS
    from network.HGFilters import HGFilter
    from utils.net_util import get_embedder
    from utils.smpl_util import smpl_util
    from utils import recon_util
    from utils.nerf_util import raw2outputs
    from dataset.avatarcap_dataset import AvatarCapDataset

    out = {}

    # ---- G0: grid points / flat order ------------------------------------------------------
    for name, res in (('toy', (4, 3, 2)), ('odd', (5, 7, 6))):
        out[f'G0_{name}_pts'] = AvatarCapDataset.generate_volume_points(syn.CANO_BOUNDS, res).numpy()
    out['G0_lin17'] = torch.linspace(0, 1, steps=17, dtype=torch.float32).numpy()
    out['G0_lin256'] = torch.linspace(0, 1, steps=256, dtype=torch.float32).numpy()

    # ---- G1: Embedder ----------------------------------------------------------------------
    x = gi.points(101, 256)
    for mr in (10, 0):
        emb, dim = get_embedder(mr, input_dims=3)
        out[f'G1_embed{mr}'] = emb(t(x)).numpy()

    # ---- G2: MLP x4 configs ----------------------------------------------------------------
    cfgs = gi.MLP_CONFIGS
    for name, c in cfgs.items():
        m = MLP(**c['kwargs']).eval()
        syn.load_synth(m, gi.SEED_MLP)
        xin = gi.features(102, 300, c['kwargs']['in_channels'])
        out[f'G2_{name}'] = m(t(xin.T[None])).numpy()[0].T

    # ---- G3: OffsetDecoder (eval BN with randomised running stats) ---------------------------
    od = OffsetDecoder(67).eval()
    syn.load_synth(od, gi.SEED_MLP)
    xin = gi.features(103, 300, 67)
    out['G3_offset_decoder'] = od(t(xin.T[None])).numpy()[0].T

    # ---- G4/G5: WarpingField.query / OccupancyNet.query -------------------------------------
    net = GeoTexAvatar().eval()
    syn.load_synth(net, gi.SEED_NET)
    fmap = gi.pose_feat_map()
    net.warping_field.pose_feat_map = t(fmap[None])
    pts = gi.query_points(104, 2048)
    center = gi.center()
    batch = {'cano_smpl_center': t(center[None]), 'cano_pts': t(pts[None])}
    out['G4_offset'] = net.warping_field.query(t(pts[None]), batch).numpy()[0]
    for if_type in ('sdf', 'occupancy'):
        config.if_type = if_type
        o = OccupancyNet(net).query(batch)
        out[f'G5_occ_{if_type}'] = o['cano_pts_ov'].numpy()[0]
        out['G5_offset'] = o['nonrigid_offset'].numpy()[0]
    config.if_type = 'sdf'
    rgb, alpha, occ = net.cano_template.forward(t(pts[None]))
    out['G5_tmpl_rgb'], out['G5_tmpl_alpha'], out['G5_tmpl_occ'] = rgb.numpy()[0], alpha.numpy()[0], occ.numpy()[0]
    # config-1 style subset of a 64^3 grid
    gpts = AvatarCapDataset.generate_volume_points(syn.CANO_BOUNDS, (64, 64, 64)).numpy()
    sel = gi.grid_subset(64 ** 3, 1500)
    batch_g = {'cano_smpl_center': t(center[None]), 'cano_pts': t(gpts[sel][None])}
    out['G5_grid64_sel_occ'] = OccupancyNet(net).query(batch_g)['cano_pts_ov'].numpy()[0]

    # ---- G6: ReconNetwork.infer --------------------------------------------------------------
    rn = ReconNetwork().eval()
    syn.load_synth(rn, gi.SEED_NET)
    nm = gi.normal_maps(64)     # small maps keep HGFilter cheap: (6,64,64) -> feature (32,32,32)
    items = {'cano_pts': t(pts[None]), 'cano_smpl_center': t(center[None]),
             'front_normal': t(nm[None, :3]), 'back_normal': t(nm[None, 3:])}
    out['G6_recon'] = rn.infer(items).numpy()
    out['G6_img_feat'] = rn.get_feat_maps(t(nm[None]))[-1].numpy()[0]
    # decoder alone on a given random feature map (what the HIP kernel replaces)
    imap = gi.img_feat_map()
    import torch.nn.functional as F
    cp = t(pts[None]) - t(center[None])[:, None]
    grid = torch.cat([cp[..., 0].view(1, -1, 1, 1), -cp[..., 1].view(1, -1, 1, 1)], -1)
    feat = F.grid_sample(t(imap[None]), grid, 'bilinear', 'border', True).squeeze(-1)
    out['G6_decoder'] = rn.image_decoder(torch.cat([feat, cp[..., 2].view(1, 1, -1)], 1)).numpy()[0, 0]

    # ---- G7: producers -----------------------------------------------------------------------
    un = UnetNoCond7DS(input_nc=6, output_nc=64, nf=32, up_mode='upconv', use_dropout=False).eval()
    syn.load_synth(un, gi.SEED_NET)
    y = un(t(gi.pos_map(128)[None])).numpy()[0]
    out['G7_unet_samples'] = y[:, gi.PIX[:, 0] % 128, gi.PIX[:, 1] % 128]
    hgf = HGFilter(1, 4, 6, 32, 'group', 'no_down', False).eval()
    syn.load_synth(hgf, gi.SEED_NET)
    y = hgf(t(nm[None]))[0][-1].numpy()[0]
    out['G7_hg_samples'] = y[:, gi.PIX[:, 0] % 32, gi.PIX[:, 1] % 32]

    # ---- G8: LBS -----------------------------------------------------------------------------
    smpl_util.set_cano_smpl_vertices(t(body['cano_smpl_v']))
    vp = gi.surface_points(105, 700, body)
    lbs = smpl_util.calculate_lbs(t(vp[None]))
    jm = syn.random_pose_jnt_mats(gi.SEED_POSE)
    live, mats = smpl_util.skinning(t(vp[None]), lbs, t(jm[None]), True)
    nrm = gi.unit_vectors(106, 700)
    out['G8_lbs'], out['G8_live'], out['G8_mats'] = lbs.numpy()[0], live.numpy()[0], mats.numpy()[0]
    out['G8_live_normals'] = smpl_util.skinning_normal(t(nrm[None]), lbs, t(jm[None])).numpy()[0]

    # ---- G9: normals from an analytic SDF volume ----------------------------------------------
    vol, voxel = gi.sdf_volume(32)
    gp = gi.grid_points_m11(107, 400)
    out['G9_normal_volume_slice'] = recon_util.extract_normal_volume(t(vol), voxel).numpy()[::5, ::5, ::5]
    out['G9_normals'] = recon_util.extract_normal_from_volume(t(vol), voxel, t(gp)).numpy()

    # ---- G10: CanoBlendWeightVolume -----------------------------------------------------------
    cw = CanoBlendWeightVolume(config.cfg['training']['training_data_dir'] + '/cano_base_blend_weight_volume.npy')
    p01 = gi.points01(108, 300)
    out['G10_blend_w'] = cw.forward(t(p01[None])).numpy()[0]

    # ---- G11: raw2outputs ---------------------------------------------------------------------
    raw, zv = gi.raw_and_z(109, 50, 64)
    r = raw2outputs(t(raw), t(zv))
    for k, v in zip(('rgb_map', 'disp_map', 'acc_map', 'weights', 'depth_map'), r):
        out[f'G11_{k}'] = v.numpy()

    # ---- G12: GeoTexAvatar.forward(pts_space='cano') (colour path) -----------------------------
    wp = gi.surface_points(110, 600, body) + 0.01 * gi.unit_vectors(111, 600)
    dists = np.full((1, 600, 1), 0.0016, np.float32)
    b2 = {'cano_smpl_center': t(center[None]), 'cano_bounds': t(syn.CANO_BOUNDS[None])}
    o = net.forward(t(wp[None].copy()), None, t(dists), b2, pts_space='cano')
    out['G12_raw'], out['G12_occ'], out['G12_off'] = o['raw'].numpy()[0], o['occ'].numpy()[0], o['nonrigid_offset'].numpy()[0]

    # ---- G13: GeoTexAvatar.forward(pts_space='posed') (inverse skinning + blend-weight volume) -------
    jm13 = syn.random_pose_jnt_mats(gi.SEED_POSE + 1, sigma=0.15)
    live_v = gi.live_smpl_vertices(body, jm13)
    wl = gi.live_query_points(112, 500, live_v)
    b3 = {'cano_smpl_center': t(center[None]), 'cano_bounds': t(syn.CANO_BOUNDS[None]), 'live_smpl_v': t(live_v[None]),
          'cano2live_jnt_mats': t(jm13[None])}
    o = net.forward(t(wl[None].copy()), None, t(np.full((1, 500, 1), 0.0016, np.float32)), b3, pts_space='posed')
    out['G13_raw'], out['G13_occ'], out['G13_off'] = o['raw'].numpy()[0], o['occ'].numpy()[0], o['nonrigid_offset'].numpy()[0]

    # ---- G14: binary PLY layout (utils/obj_io.py:223-269), bytes of three tiny meshes ---------------
    from utils import obj_io
    pv, pf, pn, pc = gi.ply_mesh()
    for tag, kw in (('v', {}), ('vn', {'normals': pn}), ('vnc', {'normals': pn, 'colors': pc.copy()})):
        fn = os.path.join(tempfile.mkdtemp(), 'm.ply')
        obj_io.save_mesh_as_ply(fn, pv, pf, **kw)
        out['G14_ply_' + tag] = np.frombuffer(open(fn, 'rb').read(), np.uint8)

    path = os.path.join(HERE, 'reference_golden.npz')
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024), 'keys', len(out))


if __name__ == '__main__':
    main()
