"""Scratch timing of the reconstruction query on the valid band (avc_recon_query_grid_subset), synthetic body: column-folded by runs (recon_fold_kernel<2>,
round 5) against the point-by-point kernel on generated coordinates, with the clock of the launches; the dense folded launch beside it."""
import ctypes as C
import sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import _lib, config, synthetic as syn
config.cfg = config.default_cfg()
from avatarcap_amd.dataset import SyntheticTestDataset
from avatarcap_amd.network.arch_recon import ReconNetwork
import golden_inputs as gi
dev = torch.device('cuda'); config.device = dev
rn = ReconNetwork().to(dev).eval(); syn.load_synth(rn, syn.SEED)
imap = torch.from_numpy(gi.img_feat_map()[None]).cuda()
ctx = _lib.ctx(dev)
for res in ([256] * 3, [384, 384, 128]):
    ds = SyntheticTestDataset(res, valid='band', n_frames=1)
    center = torch.from_numpy(np.asarray(ds.cano_smpl_center, np.float32)[None]).cuda()
    n = ds.valid_idx.numel()
    cols = (ds.valid_idx.long() // res[2])
    pad = (-n) % 32
    cw = torch.cat([cols, cols[-1:].expand(pad)]).reshape(-1, 32)
    per_wave = ((cw[:, 1:] != cw[:, :-1]).sum(1) + 1)
    print(f'res {res}: band of {n} points; runs per wave: mean {per_wave.float().mean():.2f}, max {int(per_wave.max())}, waves with > 2 runs: '
          f'{100.0 * (per_wave > 2).float().mean():.2f} %, > 8: {100.0 * (per_wave > 8).float().mean():.3f} %')
    out = {}
    for name, index, N in (('band', ds.valid_idx, n), ('dense', None, int(np.prod(res)))):
        for fold in (1, 0):
            _lib.set_option('column_fold', fold)
            y = rn.decode_grid(ds.grid_axes, res, imap, center, index=index); torch.cuda.synchronize()
            _lib.check(_lib.lib().avc_timing_enable(ctx, 1))
            for _ in range(10): y = rn.decode_grid(ds.grid_axes, res, imap, center, index=index)
            torch.cuda.synchronize()
            ms, nl, cyc = C.c_double(), C.c_int64(), C.c_double()
            _lib.check(_lib.lib().avc_timing_read(ctx, 1, C.byref(ms), C.byref(nl), 1))
            _lib.check(_lib.lib().avc_timing_read_cycles(ctx, 1, C.byref(cyc), C.byref(nl)))
            _lib.check(_lib.lib().avc_timing_enable(ctx, 0))
            out[(name, fold)] = y
            print(f'  recon {name} query {"folded" if fold else "point-by-point"}: {ms.value:.3f} ms  {ms.value / N * 1e6:.3f} ns/pt  {cyc.value / ms.value / 1e3:.0f} MHz  '
                  f'{N * 387072 / ms.value / 1e9:.0f} TFLOP/s algorithmic', flush=True)
        _lib.set_option('column_fold', 1)
    print(f'  folded vs point-by-point: band {float((out[("band", 1)] - out[("band", 0)]).abs().max()):.2e}, dense {float((out[("dense", 1)] - out[("dense", 0)]).abs().max()):.2e}; '
          f'folded band vs folded dense at the band points: {float((out[("band", 1)][0] - out[("dense", 1)][0][ds.valid_idx.long()]).abs().max()):.2e}')
    del ds
