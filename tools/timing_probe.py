import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import config, synthetic as syn
config.cfg = config.default_cfg()
from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
from avatarcap_amd.grid import generate_volume_points
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to('cuda').eval()
sd = syn.synth_state_dict(syn.module_shapes(net), syn.SEED)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.warping_field.pose_feat_map = torch.randn(1, 64, 256, 256, device='cuda')
pts = generate_volume_points(syn.CANO_BOUNDS, (256, 256, 256), 'cuda')[None]
batch = {'cano_pts': pts, 'cano_smpl_center': torch.zeros(1, 3, device='cuda')}
for _ in range(2):
    o = OccupancyNet(net).query(batch)
torch.cuda.synchronize()
d = o['nonrigid_offset'].view(-1).view(torch.int64)[:2 * 1024].cpu().numpy().reshape(1024, 2)
print('per-wave total cycles: mean %.3e min %.3e max %.3e' % (d[:, 0].mean(), d[:, 0].min(), d[:, 0].max()))
print('per-wave barrier-wait cycles: mean %.3e (%.1f%% of total)  per chunk %.0f' % (d[:, 1].mean(), 100 * d[:, 1].mean() / d[:, 0].mean(), d[:, 1].mean() / (512 * 59)))
