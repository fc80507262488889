"""Where the time to the first mesh goes (example.yaml's grid, synthetic weights): imports, modules + weights, dataset (grid, band, fill), pipeline (LBS lists),
first full frame (weight packing, encoder plans, graph capture, pinned buffers, first-use kernel loads), second frame."""
import sys, time
t0 = time.perf_counter()
import numpy as np, torch
sys.path.insert(0, '.')
from avatarcap_amd import config, synthetic as syn, _lib
dev = torch.device('cuda'); config.device = dev; config.cfg = config.default_cfg()
res = [384, 384, 128]; config.cfg['testing']['vol_res'] = res
torch.zeros(1, device=dev); torch.cuda.synchronize()
t1 = time.perf_counter()
from avatarcap_amd.network.arch_avatar import GeoTexAvatar
from avatarcap_amd.network.arch_recon import ReconNetwork
from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda, synthetic_camera, synthetic_observed_normals
from avatarcap_amd.pipeline import FramePipeline
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to(dev).eval(); syn.load_synth(net, syn.SEED)
rn = ReconNetwork().to(dev).eval(); syn.load_synth(rn, syn.SEED)
torch.cuda.synchronize(); t2 = time.perf_counter()
ds = SyntheticTestDataset(res, valid='band', n_frames=3)
torch.cuda.synchronize(); t3 = time.perf_counter()
pipe = FramePipeline(net, ds, rn)
torch.cuda.synchronize(); t4 = time.perf_counter()
w2c, cam = synthetic_camera()
ts = []
for i in range(3):
    it = to_cuda(ds[i], add_batch=True)
    torch.cuda.synchronize(); ta = time.perf_counter()
    a = pipe.avatar_frame(it)
    torch.cuda.synchronize(); tb = time.perf_counter()
    obs = synthetic_observed_normals(a['live_v'], a['live_vn'], a['f'], w2c, cam, seed=i)
    it = dict(it); it['front_normal'], it['back_normal'], _ = pipe.fuse_normals(a, obs, w2c, cam, 'merge')
    torch.cuda.synchronize(); tc = time.perf_counter()
    r = pipe.recon_frame(it)
    torch.cuda.synchronize(); td = time.perf_counter()
    ts.append((tb - ta, tc - tb, td - tc))
print(f'import torch + first device touch {t1 - t0:.2f} s; modules + synthetic weights {t2 - t1:.2f} s; dataset {t3 - t2:.2f} s; pipeline (LBS bind) {t4 - t3:.3f} s')
for i, (a_, f_, r_) in enumerate(ts):
    print(f'frame {i}: avatar_frame {a_ * 1e3:7.1f} ms, fusion {f_ * 1e3:7.1f} ms, recon_frame {r_ * 1e3:7.1f} ms')
