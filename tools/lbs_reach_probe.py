"""How far from the canonical SMPL vertices the per-cell candidate lists of avc_lbs_prepare should reach: LBS time of a band frame's avatar mesh (synthetic
body, 256^3) and the size of the lists, per reach."""
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from avatarcap_amd import _lib, config, synthetic as syn
config.cfg = config.default_cfg(); config.device = torch.device('cuda')
from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda
from avatarcap_amd.network.arch_avatar import GeoTexAvatar
from avatarcap_amd.pipeline import FramePipeline
from avatarcap_amd.utils.smpl_util import smpl_util
config.cfg['testing']['vol_res'] = [256] * 3
ds = SyntheticTestDataset([256] * 3, valid='band', n_frames=1)
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).cuda().eval(); syn.load_synth(net, syn.SEED)
pipe = FramePipeline(net, ds)
out = pipe.avatar_frame(to_cuda(ds[0], add_batch=True))
v = out['cano_v'][None].contiguous()
d2, _ = smpl_util.knn_points(v, smpl_util.cano_smpl_vertices[None], K=4)
d4 = d2[0, :, 3].sqrt()
print(f'{v.shape[1]} vertices; distance to the 4th nearest SMPL vertex: median {float(d4.median()):.3f}, 90 % {float(d4.quantile(0.9)):.3f}, 99 % {float(d4.quantile(0.99)):.3f}, max {float(d4.max()):.3f} m')
ctx = _lib.ctx(torch.device('cuda', 0))
ref = None
for reach in (0, 80, 120, 140, 160, 200):
    _lib.set_option('lbs_reach_mm', reach)
    _lib.set_owner(ctx, 'lbs_bound', None)
    t = time.perf_counter(); smpl_util.set_cano_smpl_vertices(smpl_util.cano_smpl_vertices); torch.cuda.synchronize(); tp = time.perf_counter() - t
    st = (C.c_int64 * 4)(); _lib.check(_lib.lib().avc_lbs_bound_stats(ctx, st))
    lbs = smpl_util.calculate_lbs(v); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): lbs = smpl_util.calculate_lbs(v)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    ref = lbs if ref is None else ref
    print(f'reach {reach:4d} mm: prepare {tp*1e3:6.1f} ms, {st[1]} cells, {st[2]} list entries ({st[2]*16/1e6:.0f} MB); calculate_lbs {dt*1e6:7.1f} us; identical to reach 0: {bool(torch.equal(lbs, ref))}', flush=True)
_lib.set_option('lbs_reach_mm', 120)
