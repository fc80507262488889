"""calculate_lbs timing: scattered queries (worst case for the wave-cooperative grid search) and queries in
marching-cubes vertex order (the frame's case): the grid search as shipped (per-wave choice), each of its two searches forced
(avc_set_option knn_search 1 | 2), and the exhaustive scan (knn_search 3)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import _lib, config, synthetic as syn
config.cfg = config.default_cfg(); config.device = torch.device('cuda')
from avatarcap_amd.utils.smpl_util import SmplUtil
from avatarcap_amd.utils import recon_util
from avatarcap_amd.grid import generate_volume_points_np
body = syn.synthetic_body()
su = SmplUtil(body['skin_weights']); su.set_cano_smpl_vertices(torch.from_numpy(body['cano_smpl_v']).cuda())


def timeit(name, pts):
    for mode in ('grid', 'lane', 'wave', 'brute'):
        _lib.set_option('knn_search', {'grid': 0, 'lane': 1, 'wave': 2, 'brute': 3}[mode])
        su.calculate_lbs(pts); torch.cuda.synchronize()
        t = time.time()
        for _ in range(5): su.calculate_lbs(pts)
        torch.cuda.synchronize(); print(f'{name:28s} n={pts.shape[1]:8d} {mode:5s}: {(time.time()-t)/5*1e3:7.3f} ms', flush=True)
    _lib.set_option('knn_search', 0)


rs = np.random.RandomState(0)
timeit('scattered', torch.from_numpy(rs.uniform(syn.CANO_BOUNDS[0], syn.CANO_BOUNDS[1], (1, 1_900_000, 3)).astype(np.float32)).cuda())
res = [256, 256, 256]
pts = generate_volume_points_np(syn.CANO_BOUNDS, res)
sdf = torch.from_numpy(np.concatenate([syn.body_sdf(c) for c in np.array_split(pts, 64)]).astype(np.float32)).cuda()
v, f, n = recon_util.recon_mesh_device(sdf, res, syn.CANO_BOUNDS, iso_value=0.0)
timeit('body surface (MC order)', v[None].contiguous())
noise = torch.from_numpy(rs.randn(64, 64, 64).astype(np.float32)).cuda()
vol = torch.nn.functional.interpolate(noise[None, None], size=res, mode='trilinear')[0, 0].contiguous()
v, f, n = recon_util.recon_mesh_device(vol.reshape(-1), res, syn.CANO_BOUNDS, iso_value=0.0)
timeit('noise surface (MC order)', v[None].contiguous())
