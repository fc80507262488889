"""Scratch timing of the fused reconstruction query (ReconNetwork.decode) on n random points."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import config
config.cfg = config.default_cfg()
import golden_inputs as gi
from common import recon_sd
from avatarcap_amd.network.arch_recon import ReconNetwork
rn = ReconNetwork().to('cuda').eval()
rn.load_state_dict({k: torch.from_numpy(v) for k, v in recon_sd().items()})
imap = torch.from_numpy(gi.img_feat_map()[None]).cuda()
center = torch.from_numpy(gi.center()[None]).cuda()
MAC = 33 * 512 + 545 * 256 + 289 * 128 + 161
for n in (2_800_000, 16_777_216):
    pts = (torch.rand(1, n, 3, device='cuda') - 0.5) * torch.tensor([1.0, 1.8, 0.5], device='cuda')
    y = rn.decode(pts, imap, center); torch.cuda.synchronize()
    t0 = time.time(); reps = 5
    for _ in range(reps): y = rn.decode(pts, imap, center)
    torch.cuda.synchronize(); dt = (time.time() - t0) / reps
    print(f'recon decode n={n}: {dt*1e3:.2f} ms  {dt/n*1e9:.3f} ns/pt  {n*MAC*2/dt/1e12:.1f} TFLOP/s algorithmic', flush=True)
