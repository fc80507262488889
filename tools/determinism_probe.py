import sys, numpy as np, torch
sys.path.insert(0, '.')
from avatarcap_amd import config, synthetic as syn
from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda
from avatarcap_amd.network.arch_avatar import GeoTexAvatar
from avatarcap_amd.pipeline import FramePipeline
dev = torch.device('cuda'); config.device = dev; config.cfg = config.default_cfg()
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to(dev).eval(); syn.load_synth(net, syn.SEED)
config.cfg['testing']['vol_res'] = [128] * 3
ds = SyntheticTestDataset([128] * 3, valid='band', n_frames=1)
pipe = FramePipeline(net, ds)
items = to_cuda(ds[0], add_batch=True)
outs, maps = [], []
for _ in range(3):
    o = pipe.avatar_frame(items)
    outs.append(o); maps.append(net.warping_field.pose_feat_map.clone())
print('pose_feat_map equal:', [bool(torch.equal(maps[0], m)) for m in maps[1:]], 'max diff', [float((maps[0] - m).abs().max()) for m in maps[1:]])
print('occ equal:', [bool(torch.equal(outs[0]['occ_volume'], o['occ_volume'])) for o in outs[1:]], 'verts', [o['cano_v'].shape[0] for o in outs])
# same map, query twice
q1 = pipe.occ_net.query(items)['cano_pts_ov'].clone(); q2 = pipe.occ_net.query(items)['cano_pts_ov']
print('query deterministic given the map:', bool(torch.equal(q1, q2)))
