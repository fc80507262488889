"""BASELINE configs[3]'s colour leg in isolation (main.py:464-477 on 200k vertices of a 256^3 mesh) for a rocprofv3 kernel split."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from avatarcap_amd import config, synthetic as syn
from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda
from avatarcap_amd.network.arch_avatar import GeoTexAvatar
from avatarcap_amd.pipeline import FramePipeline
dev = torch.device('cuda'); config.device = dev; config.cfg = config.default_cfg(); config.cfg['testing']['vol_res'] = [256] * 3
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to(dev).eval(); syn.load_synth(net, syn.SEED)
ds = SyntheticTestDataset([256] * 3, valid='band', n_frames=1)
pipe = FramePipeline(net, ds)
items = to_cuda(ds[0], add_batch=True)
a = pipe.avatar_frame(items)
nv = min(200_000, a['cano_v'].shape[0])
v, n = a['cano_v'][:nv].contiguous(), a['cano_vn'][:nv].contiguous()
for _ in range(3):
    rgb = pipe.colour_vertices(items, v, n)
torch.cuda.synchronize()
print('colour leg done', nv, tuple(rgb.shape))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    rgb = pipe.colour_vertices(items, v, n)
e1.record()
torch.cuda.synchronize()
print(f'colour_vertices: {e0.elapsed_time(e1) / 3:.2f} ms per {nv} vertices')
