"""Minimal driver for rocprofv3 --pmc runs: one warm-up and N dense avatar queries at RES^3."""
import sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import config, synthetic as syn
config.cfg = config.default_cfg()
from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
from avatarcap_amd.grid import generate_volume_points
res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to('cuda').eval()
sd = syn.synth_state_dict(syn.module_shapes(net), syn.SEED)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.warping_field.pose_feat_map = torch.randn(1, 64, 256, 256, device='cuda')
from avatarcap_amd.grid import volume_axes
batch = {'cano_smpl_center': torch.zeros(1, 3, device='cuda')}
axes = volume_axes(syn.CANO_BOUNDS, (res, res, res), 'cuda')
mode = sys.argv[3] if len(sys.argv) > 3 else 'grid'       # 'grid' = what bench.py launches (points from the index, no offsets); 'pts' = the (N,3) array
if mode == 'pts':
    batch['cano_pts'] = generate_volume_points(syn.CANO_BOUNDS, (res, res, res), 'cuda')[None]
if mode == 'recon':                        # the folded recon query on the dense grid (avc_recon_query_grid)
    sys.path.insert(0, 'tests')
    import golden_inputs as gi
    from common import recon_sd
    from avatarcap_amd.network.arch_recon import ReconNetwork
    rn = ReconNetwork().to('cuda').eval()
    rn.load_state_dict({k: torch.from_numpy(v) for k, v in recon_sd().items()})
    imap = torch.from_numpy(gi.img_feat_map()[None]).cuda()
    center = torch.from_numpy(gi.center()[None]).cuda()
    for _ in range(reps + 1):
        rn.decode_grid(axes, (res, res, res), imap, center)
    torch.cuda.synchronize()
    print('done')
    sys.exit(0)
for _ in range(reps + 1):
    if mode == 'pts':
        OccupancyNet(net).query(batch)
    else:
        OccupancyNet(net).query_grid(batch, axes, (res, res, res))
torch.cuda.synchronize()
print('done')
