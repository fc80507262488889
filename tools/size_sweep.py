"""Fused avatar query time per point vs launch size (contiguous grid points) and for the band-masked point set."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import config, synthetic as syn
config.cfg = config.default_cfg(); config.device = torch.device('cuda')
from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
from avatarcap_amd.dataset import SyntheticTestDataset
from avatarcap_amd.grid import generate_volume_points
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to('cuda').eval(); syn.load_synth(net, syn.SEED)
net.warping_field.pose_feat_map = torch.randn(1, 64, 256, 256, device='cuda')
pts = generate_volume_points(syn.CANO_BOUNDS, (256, 256, 256), 'cuda')
ds = SyntheticTestDataset([256] * 3, valid='band', n_frames=1)
q = OccupancyNet(net)
c = torch.zeros(1, 3, device='cuda')


def t(p, reps):
    b = {'cano_pts': p[None].contiguous(), 'cano_smpl_center': c}
    q.query(b); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): q.query(b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in (1 << 20, 2800408, 1 << 23, 1 << 24):
    ms = t(pts[:n], max(2, (1 << 25) // n))
    print(f'contiguous n={n:9d}: {ms:7.2f} ms  {ms * 1e6 / n:.3f} ns/pt', flush=True)
ms = t(ds.infer_pts, 10)
print(f'band       n={ds.infer_pts.shape[0]:9d}: {ms:7.2f} ms  {ms * 1e6 / ds.infer_pts.shape[0]:.3f} ns/pt')
mid = pts[(1 << 23):(1 << 23) + 2800408]
ms = t(mid, 10)
print(f'mid-volume n={mid.shape[0]:9d}: {ms:7.2f} ms  {ms * 1e6 / mid.shape[0]:.3f} ns/pt')
