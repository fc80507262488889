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

# the same decoder on the dense 256^3 grid (column-folded: avc_recon_query_grid) and on a band of it by indices, with the shader clock of the launches
import ctypes as C
from avatarcap_amd import _lib, synthetic as syn
from avatarcap_amd.grid import volume_axes
res = (256, 256, 256)
ax = volume_axes(syn.CANO_BOUNDS, res, 'cuda')
ctx = _lib.ctx(torch.device('cuda', 0))
idx = torch.nonzero(torch.rand(256 ** 3, device='cuda') < 0.17)[:, 0].to(torch.int32).contiguous()
for name, index in (('dense grid', None), (f'grid subset n={idx.numel()}', idx)):
    for fold in (1, 0):
        _lib.set_option('column_fold', fold)
        y = rn.decode_grid(ax, res, imap, center, index=index); torch.cuda.synchronize()
        _lib.check(_lib.lib().avc_timing_enable(ctx, 1))
        for _ in range(5): y = rn.decode_grid(ax, res, imap, center, index=index)
        torch.cuda.synchronize()
        ms, nl, cyc = C.c_double(), C.c_int64(), C.c_double()
        _lib.check(_lib.lib().avc_timing_read(ctx, 1, C.byref(ms), C.byref(nl), 1))
        _lib.check(_lib.lib().avc_timing_read_cycles(ctx, 1, C.byref(cyc), C.byref(nl)))
        _lib.check(_lib.lib().avc_timing_enable(ctx, 0))
        n = y.numel()
        print(f'recon {name} {"folded" if fold else "point-by-point"}: {ms.value:.2f} ms  {ms.value/n*1e6:.3f} ns/pt  {n*MAC*2/ms.value/1e9:.1f} TFLOP/s algorithmic  '
              f'{cyc.value / max(ms.value, 1e-9) / 1e3:.0f} MHz', flush=True)
_lib.set_option('column_fold', 1)
