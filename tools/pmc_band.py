"""Minimal driver for rocprofv3 --pmc runs of the BAND launches (round 5): one warm-up and N launches each of the avatar query and the reconstruction query on the
valid band of the synthetic body at 256^3 (avc_avatar_query_grid_subset / avc_recon_query_grid_subset)."""
import sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import config, synthetic as syn
config.cfg = config.default_cfg(); config.device = torch.device('cuda')
from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda
from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
from avatarcap_amd.network.arch_recon import ReconNetwork
import golden_inputs as gi
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
res = [256] * 3
config.cfg['testing']['vol_res'] = res
ds = SyntheticTestDataset(res, valid='band', n_frames=1)
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).cuda().eval(); syn.load_synth(net, syn.SEED)
rn = ReconNetwork().cuda().eval(); syn.load_synth(rn, syn.SEED)
items = to_cuda(ds[0], add_batch=True)
net.warping_field.pose_feat_map = torch.randn(1, 64, 256, 256, device='cuda')
imap = torch.from_numpy(gi.img_feat_map()[None]).cuda()
center = items['cano_smpl_center']
q = OccupancyNet(net)
for _ in range(reps + 1):
    q.query_grid(items, ds.grid_axes, res, index=ds.valid_idx)
    rn.decode_grid(ds.grid_axes, res, imap, center, index=ds.valid_idx)
torch.cuda.synchronize()
print('band points', ds.valid_idx.numel())
