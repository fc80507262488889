"""Scratch timing of the avatar query on the valid band (avc_avatar_query_grid_subset), 256^3, synthetic body; folded vs unfolded, with the clock of the launches."""
import ctypes as C
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from avatarcap_amd import _lib, config, synthetic as syn
config.cfg = config.default_cfg()
from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda
from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
dev = torch.device('cuda'); config.device = dev
config.cfg['testing']['vol_res'] = [256] * 3
ds = SyntheticTestDataset([256] * 3, valid='band', n_frames=1)
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to(dev).eval(); syn.load_synth(net, syn.SEED)
items = to_cuda(ds[0], add_batch=True)
net.warping_field.precompute_conv(items)
q = OccupancyNet(net)
ctx = _lib.ctx(dev)
cols = (ds.valid_idx.long() // 256)
runs = (cols[1:] != cols[:-1]).sum().item() + 1
n = ds.valid_idx.numel()
pad = (-n) % 32
cw = torch.cat([cols, cols[-1:].expand(pad)]).reshape(-1, 32)
per_wave = ((cw[:, 1:] != cw[:, :-1]).sum(1) + 1)
print(f'band: {n} points, {runs} runs along z (mean length {n / runs:.1f}); runs per wave: mean {per_wave.float().mean():.2f}, max {int(per_wave.max())}, '
      f'waves with > 6 runs: {100.0 * (per_wave > 6).float().mean():.2f} %')
cus = torch.cuda.get_device_properties(0).multi_processor_count
for what, index, npts, reps in (('band', ds.valid_idx, n, 10), ('dense', None, 256 ** 3, 3)):
    for fold in (1, 0):
        if index is None and not fold:
            continue
        _lib.set_option('column_fold', fold)
        q.query_grid(items, ds.grid_axes, [256] * 3, index=index); torch.cuda.synchronize()
        _lib.check(_lib.lib().avc_timing_enable(ctx, 1))
        for _ in range(reps): q.query_grid(items, ds.grid_axes, [256] * 3, index=index)
        torch.cuda.synchronize()
        ms, nl, cyc = C.c_double(), C.c_int64(), C.c_double()
        _lib.check(_lib.lib().avc_timing_read(ctx, 0, C.byref(ms), C.byref(nl), 1))
        _lib.check(_lib.lib().avc_timing_read_cycles(ctx, 0, C.byref(cyc), C.byref(nl)))
        _lib.check(_lib.lib().avc_timing_enable(ctx, 0))
        tiles = (npts + 127) // 128
        tiles_wg0 = (tiles - 1) // cus + 1             # tiles of workgroup 0, whose s_memtime stamps `cyc` is
        print(f'{what} query {"folded" if fold else "point-by-point"}: {ms.value:.3f} ms (with its column pass)  {ms.value / npts * 1e6:.3f} ns/pt  '
              f'kernel {cyc.value:.4e} shader cycles = {cyc.value / tiles_wg0:.0f} per tile of workgroup 0 ({tiles_wg0} tiles; mean {tiles / cus:.2f} per workgroup)', flush=True)
_lib.set_option('column_fold', 1)
