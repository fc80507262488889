"""Timing of step 2 (canonical normal fusion) on a band-masked 256^3 avatar frame: position render + canonicalisation +
canonical renders + 100-iteration merge, all on the device."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from avatarcap_amd import config, synthetic as syn
from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda, synthetic_camera, synthetic_observed_normals
from avatarcap_amd.network.arch_avatar import GeoTexAvatar
from avatarcap_amd.pipeline import FramePipeline
dev = torch.device('cuda'); config.device = dev; config.cfg = config.default_cfg()
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to(dev).eval(); syn.load_synth(net, syn.SEED)
config.cfg['testing']['vol_res'] = [256] * 3
ds = SyntheticTestDataset([256] * 3, valid='band', n_frames=1)
pipe = FramePipeline(net, ds)
a = pipe.avatar_frame(to_cuda(ds[0], add_batch=True))
w2c, cam = synthetic_camera()
obs = synthetic_observed_normals(a['live_v'], a['live_vn'], a['f'], w2c, cam, seed=0)
for manner in ('merge', 'cover'):
    pipe.fuse_normals(a, obs, w2c, cam, manner); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): pipe.fuse_normals(a, obs, w2c, cam, manner)
    torch.cuda.synchronize()
    print('fuse_normals(%s): %.2f ms (%d vertices, %d faces, 512x512 maps, 100 iterations)' % (manner, (time.perf_counter() - t) / 5 * 1e3, a['cano_v'].shape[0], a['f'].shape[0]))
