import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from avatarcap_amd import config, synthetic as syn, _lib
dev = torch.device('cuda'); config.device = dev; config.cfg = config.default_cfg()
L = _lib.lib()
names = ['avc_hgfilter_pack', 'avc_unet_pack', 'avc_pack_template_weights', 'avc_pack_warp_weights', 'avc_pack_recon_weights', 'avc_unet_forward', 'avc_hgfilter_forward', 'avc_lbs_prepare', 'avc_recon_mesh']
acc = {}
class Wrap:
    def __init__(s, f, n): s.f, s.n = f, n; s.restype = f.restype; s.argtypes = f.argtypes
    def __call__(s, *a):
        torch.cuda.synchronize(); t = time.perf_counter(); r = s.f(*a); torch.cuda.synchronize(); acc.setdefault(s.n, []).append(time.perf_counter() - t); return r
for n in names:
    setattr(L, n, Wrap(getattr(L, n), n))
from avatarcap_amd.network.arch_avatar import GeoTexAvatar
from avatarcap_amd.network.arch_recon import ReconNetwork
from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda, synthetic_camera, synthetic_observed_normals
from avatarcap_amd.pipeline import FramePipeline
res = [128, 128, 64]; config.cfg['testing']['vol_res'] = res
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to(dev).eval(); syn.load_synth(net, syn.SEED)
rn = ReconNetwork().to(dev).eval(); syn.load_synth(rn, syn.SEED)
ds = SyntheticTestDataset(res, valid='band', n_frames=2)
pipe = FramePipeline(net, ds, rn)
w2c, cam = synthetic_camera()
for i in range(2):
    t = time.perf_counter()
    it = to_cuda(ds[i], add_batch=True)
    a = pipe.avatar_frame(it)
    obs = synthetic_observed_normals(a['live_v'], a['live_vn'], a['f'], w2c, cam, seed=i)
    it = dict(it); it['front_normal'], it['back_normal'], _ = pipe.fuse_normals(a, obs, w2c, cam, 'merge')
    r = pipe.recon_frame(it); torch.cuda.synchronize()
    print(f'frame {i}: {1e3 * (time.perf_counter() - t):.1f} ms')
for n, v in acc.items():
    print(f'{n:28s} calls {len(v)} first {1e3 * v[0]:8.1f} ms  later {1e3 * (sum(v[1:]) / max(1, len(v) - 1)):8.2f} ms')
