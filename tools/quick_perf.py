"""Scratch timing of the fused avatar query (not the bench contract)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import config, synthetic as syn
config.cfg = config.default_cfg()
import golden_inputs as gi
from common import geotex_sd
from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
net = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume()).to('cuda').eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in geotex_sd().items()})
net.warping_field.pose_feat_map = torch.from_numpy(gi.pose_feat_map()[None]).cuda()
from avatarcap_amd.grid import generate_volume_points, volume_axes
GRID = 'grid' in sys.argv[1:]          # the dense launch of the frame loop: points generated from the grid index (column-folded at these sizes)
for res in (128, 256):
    pts = generate_volume_points(syn.CANO_BOUNDS, (res, res, res), 'cuda')[None]
    batch = {'cano_pts': pts, 'cano_smpl_center': torch.from_numpy(gi.center()[None]).cuda()}
    ax = volume_axes(syn.CANO_BOUNDS, (res, res, res), 'cuda')
    run = (lambda: OccupancyNet(net).query_grid(batch, ax, (res, res, res))) if GRID else (lambda: OccupancyNet(net).query(batch))
    o = run(); torch.cuda.synchronize()
    import ctypes as C
    from avatarcap_amd import _lib
    ctx = _lib.ctx(torch.device('cuda', 0))
    _lib.check(_lib.lib().avc_timing_enable(ctx, 1))
    t0 = time.time(); reps = 3
    for _ in range(reps): o = run()
    torch.cuda.synchronize(); dt = (time.time() - t0) / reps
    ms, nl, cyc = C.c_double(), C.c_int64(), C.c_double()
    _lib.check(_lib.lib().avc_timing_read(ctx, 0, C.byref(ms), C.byref(nl), 1))
    _lib.check(_lib.lib().avc_timing_read_cycles(ctx, 0, C.byref(cyc), C.byref(nl)))
    _lib.check(_lib.lib().avc_timing_enable(ctx, 0))
    n = res ** 3
    print(f'res {res}: {dt*1e3:.2f} ms  {n/dt/1e6:.1f} Mpts/s  {n*1773568/dt/1e12:.1f} TFLOP/s algorithmic  ({n*1773568*3/dt/1e12:.0f} issued)  '
          f'device {ms.value:.2f} ms, {cyc.value:.4e} shader cycles = {cyc.value / max(ms.value, 1e-9) / 1e3:.0f} MHz', flush=True)
