import sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import config, synthetic as syn
config.cfg = config.default_cfg(); config.device = torch.device('cuda')
from avatarcap_amd.utils.smpl_util import SmplUtil
body = syn.synthetic_body()
su = SmplUtil(body['skin_weights']); su.set_cano_smpl_vertices(torch.from_numpy(body['cano_smpl_v']).cuda())
for n in (600_000, 1_900_000):
    pts = torch.from_numpy(np.random.RandomState(0).uniform(syn.CANO_BOUNDS[0], syn.CANO_BOUNDS[1], (1, n, 3)).astype(np.float32)).cuda()
    su.calculate_lbs(pts); torch.cuda.synchronize()
    t = time.time()
    for _ in range(5): l = su.calculate_lbs(pts)
    torch.cuda.synchronize(); print(f'calculate_lbs n={n}: {(time.time()-t)/5*1e3:.3f} ms')
