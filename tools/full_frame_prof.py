"""BASELINE configs[2] as one chained frame, steps 1-3 of main.py on the device (avatar -> canonical normal fusion with a
synthesised observed normal map -> HGFilter + recon decoder -> mesh -> LBS), 256^3 band-masked like the reference; run
under rocprofv3 --kernel-trace --stats for the per-kernel split.  argv: [frames] [merge|cover|none]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from avatarcap_amd import config, synthetic as syn
from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda, synthetic_camera, synthetic_observed_normals
from avatarcap_amd.network.arch_avatar import GeoTexAvatar
from avatarcap_amd.network.arch_recon import ReconNetwork
from avatarcap_amd.pipeline import FramePipeline
import os
if os.environ.get('CUDNN_BENCH'): torch.backends.cudnn.benchmark = True      # MIOpen exhaustive search instead of the immediate-mode pick
dev = torch.device('cuda'); config.device = dev; config.cfg = config.default_cfg()
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to(dev).eval(); syn.load_synth(net, syn.SEED)
rn = ReconNetwork().to(dev).eval(); syn.load_synth(rn, syn.SEED)
config.cfg['testing']['vol_res'] = [256] * 3
ds = SyntheticTestDataset([256] * 3, valid='band', n_frames=4)
pipe = FramePipeline(net, ds, rn)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
manner = sys.argv[2] if len(sys.argv) > 2 else 'merge'
w2c, cam = synthetic_camera()
for i in range(n + 1):
    if i == 1: torch.cuda.synchronize(); t = time.perf_counter()
    items = to_cuda(ds[i % 4], add_batch=True)
    if manner == 'none':
        a, r = pipe.full_frame(items)
    else:
        a = pipe.avatar_frame(items)                                                             # step 1
        obs = synthetic_observed_normals(a['live_v'], a['live_vn'], a['f'], w2c, cam, seed=i)   # stands in for the inferred image normals
        items['front_normal'], items['back_normal'], _ = pipe.fuse_normals(a, obs, w2c, cam, manner)   # step 2
        r = pipe.recon_frame(items)                                                              # step 3
torch.cuda.synchronize()
print('full frame (configs[2], %d valid points, fusion=%s): %.2f ms/frame, avatar %d verts, recon %d verts' %
      (ds.infer_pts.shape[0], manner, (time.perf_counter() - t) / n * 1e3, a['cano_v'].shape[0], r['cano_v'].shape[0]))
