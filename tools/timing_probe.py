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
d = o['nonrigid_offset'].view(-1).view(torch.int64)[:4 * 1024].cpu().numpy().reshape(1024, 4)
tot = d[:, 0].mean()
print('per-wave total cycles: mean %.4e min %.4e max %.4e  -> %.2f cycles per MFMA' % (tot, d[:, 0].min(), d[:, 0].max(), tot / (4920 * 512)))
print('chunk entry (LDS-DMA drain + barrier): %.1f %% of total, %.0f cycles per chunk; of which drain %.1f %% (%.0f per chunk)' %
      (100 * d[:, 1].mean() / tot, d[:, 1].mean() / (512 * 59), 100 * d[:, 2].mean() / tot, d[:, 2].mean() / (512 * 59)))
print('tile prologues (point, gathers, positional encoding): %.1f %% of total, %.0f cycles per tile' % (100 * d[:, 3].mean() / tot, d[:, 3].mean() / 512))
