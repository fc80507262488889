"""sha256 of the fused queries' outputs on fixed seeded inputs (points, dense grid, band subset, colour head, recon): a before / after check for builds that must not
move a bit.  AVCAP_LIB selects the library."""
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import config, synthetic as syn
config.cfg = config.default_cfg(); config.device = torch.device('cuda'); config.if_type = 'sdf'
import golden_inputs as gi
from common import geotex_sd, recon_sd
from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
from avatarcap_amd.network.arch_recon import ReconNetwork
from avatarcap_amd.grid import generate_volume_points_np, volume_axes
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
h = lambda x: hashlib.sha256(x.contiguous().cpu().numpy().tobytes()).hexdigest()[:12]
net = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume()).to('cuda').eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in geotex_sd().items()})
net.warping_field.pose_feat_map = t(gi.pose_feat_map()[None])
occ = OccupancyNet(net)
res = (16, 12, 128)
allp = generate_volume_points_np(syn.CANO_BOUNDS, res)
ax = volume_axes(syn.CANO_BOUNDS, res, 'cuda')
center = t(gi.center()[None])
rs = np.random.RandomState(3)
idx = np.sort(rs.choice(allp.shape[0], 7001, replace=False)).astype(np.int32)
b_all = {'cano_pts': t(allp[None]), 'cano_smpl_center': center}
a = occ.query(b_all)
g = occ.query_grid(b_all, ax, res, want_offset=True)
s = occ.query_grid({'cano_pts': t(allp[idx][None]), 'cano_smpl_center': center}, ax, res, want_offset=True, index=torch.from_numpy(idx).cuda())
print('avatar points', h(a['cano_pts_ov']), h(a['nonrigid_offset']), ' dense grid', h(g['cano_pts_ov']), h(g['nonrigid_offset']), ' band', h(s['cano_pts_ov']), h(s['nonrigid_offset']))
rn = ReconNetwork().to('cuda').eval(); rn.load_state_dict({k: torch.from_numpy(v) for k, v in recon_sd().items()})
imap = t(gi.img_feat_map(seed=212)[None])
print('recon points', h(rn.decode(t(allp[None]), imap, center)), ' dense grid', h(rn.decode_grid(ax, res, imap, center)), ' band', h(rn.decode_grid(ax, res, imap, center, index=torch.from_numpy(idx).cuda())))
out = net.cano_template(t(allp[None, :5000])) if hasattr(net, 'cano_template') else None
try:
    o = net({'cano_pts': t(allp[None, :5000]), 'cano_smpl_center': center}, 'cano') if False else None
except Exception:
    o = None
