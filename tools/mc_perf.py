"""Scratch timing of recon_mesh_device (marching cubes + normals) on a smooth-noise 256^3 volume, with and without normals."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from avatarcap_amd import config, synthetic as syn
config.cfg = config.default_cfg(); config.device = torch.device('cuda')
from avatarcap_amd.utils import recon_util
res = [int(a) for a in (sys.argv[1:4] or (256, 256, 256))]
cells = int(sys.argv[4]) if len(sys.argv) > 4 else 64
rs = np.random.RandomState(0)
noise = torch.from_numpy(rs.randn(cells, cells, cells).astype(np.float32)).cuda()
vol = torch.nn.functional.interpolate(noise[None, None], size=res, mode='trilinear')[0, 0].contiguous().reshape(-1)
for wn in (True, False):
    v, f, n = recon_util.recon_mesh_device(vol, res, syn.CANO_BOUNDS, iso_value=0.0, with_normals=wn); torch.cuda.synchronize()
    t = time.time()
    for _ in range(10): v, f, n = recon_util.recon_mesh_device(vol, res, syn.CANO_BOUNDS, iso_value=0.0, with_normals=wn)
    torch.cuda.synchronize()
    print(f'res {res} normals={wn}: {(time.time()-t)/10*1e3:.3f} ms  {v.shape[0]} vertices {f.shape[0]} faces', flush=True)
# the frame's case: a body-sized closed surface with ripples (made on the device: no CPU work on the GPU box), ~0.6 M vertices at 256^3 like the avatar's
ax = [torch.linspace(-1, 1, r, device='cuda') for r in res]
X, Y, Z = torch.meshgrid(*ax, indexing='ij')
lvl = (torch.sqrt((X / 0.45) ** 2 + (Y / 0.8) ** 2 + (Z / 0.3) ** 2) - 1.0 + 0.03 * torch.sin(37 * X) * torch.sin(41 * Y) * torch.sin(43 * Z)).contiguous().reshape(-1)
for wn in (True,):
    v, f, n = recon_util.recon_mesh_device(lvl, res, syn.CANO_BOUNDS, iso_value=0.0, with_normals=wn); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): v, f, n = recon_util.recon_mesh_device(lvl, res, syn.CANO_BOUNDS, iso_value=0.0, with_normals=wn)
    e1.record(); torch.cuda.synchronize()
    print(f'rippled ellipsoid, res {res} normals={wn}: {e0.elapsed_time(e1) / 20:.3f} ms  {v.shape[0]} vertices {f.shape[0]} faces', flush=True)
# reference points for the classify pass (one streaming read of the same 67 MB): what stock reductions take on this box
for name, fn in (('lvl.sum()', lambda: lvl.sum()), ('(lvl > 0).sum()', lambda: (lvl > 0).sum()), ('lvl.max()', lambda: lvl.max())):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  ({lvl.numel() * 4 / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12:.2f} TB/s)', flush=True)
