"""One-off runs of the BASELINE.json parity-test configurations that are not the bench line
(configs[2] full AvatarCap at 256^3, configs[3] 512^3 + colour + marching cubes); prints a JSON summary."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from avatarcap_amd import config, synthetic as syn
from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda
from avatarcap_amd.network.arch_avatar import GeoTexAvatar
from avatarcap_amd.network.arch_recon import ReconNetwork
from avatarcap_amd.pipeline import FramePipeline

def timed(fn, n=2):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n, out

dev = torch.device('cuda'); config.device = dev; config.cfg = config.default_cfg()
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to(dev).eval(); syn.load_synth(net, syn.SEED)
rn = ReconNetwork().to(dev).eval(); syn.load_synth(rn, syn.SEED)
res_out = {}
# configs[2]: full AvatarCap, 256^3 band-masked like the reference: avatar pass + recon pass (normal maps synthetic)
config.cfg['testing']['vol_res'] = [256] * 3
ds = SyntheticTestDataset([256] * 3, valid='band', n_frames=2)
pipe = FramePipeline(net, ds, rn)
items = to_cuda(ds[0], add_batch=True)
nm = torch.from_numpy(syn.smooth_normal_maps(7, 512)).to(dev)
items['front_normal'], items['back_normal'] = nm[None, :3], nm[None, 3:]
ta, a = timed(lambda: pipe.avatar_frame(items))
tr, r = timed(lambda: pipe.recon_frame(items))
res_out['configs[2] full AvatarCap 256^3 (band)'] = {'valid_points': int(ds.infer_pts.shape[0]), 'avatar_frame_ms': ta * 1e3, 'recon_frame_ms': tr * 1e3,
                                                   'avatar_verts': int(a['cano_v'].shape[0]), 'recon_verts': int(r['cano_v'].shape[0])}
del ds, pipe; torch.cuda.empty_cache()
# configs[3]: 512^3 dense + marching cubes + colour head on the vertices
config.cfg['testing']['vol_res'] = [512] * 3
ds = SyntheticTestDataset([512] * 3, valid='dense', n_frames=1)
pipe = FramePipeline(net, ds, rn)
items = to_cuda(ds[0], add_batch=True)
t5, a5 = timed(lambda: pipe.avatar_frame(items), n=1)
nv = min(200_000, a5['cano_v'].shape[0])
tc, rgb = timed(lambda: pipe.colour_vertices(items, a5['cano_v'][:nv].contiguous(), a5['cano_vn'][:nv].contiguous()), n=1)
res_out['configs[3] 512^3 dense + MC + colour'] = {'points': 512 ** 3, 'avatar_frame_ms': t5 * 1e3, 'verts': int(a5['cano_v'].shape[0]), 'faces': int(a5['f'].shape[0]),
                                                 'colour_vertices': nv, 'colour_ms': tc * 1e3, 'rgb_finite': bool(torch.isfinite(rgb).all())}
print(json.dumps(res_out, indent=1))
