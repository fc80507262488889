"""Fuzz of the column-folded grid queries (csrc/fused_mlp.hip: avatar_kernel<.., 2> / recon_fold_kernel<2> + band_prepass_kernel + left-over tiles, and the dense
folds <.., 1>): random grid shapes, and index lists with the run structures the folding has to keep apart -- bands (runs along z), single points per column, many
short runs per wavefront (more than a pass / the recon kernel holds), columns met again, reversed and shuffled orders, DUPLICATED indices, counts around every tile and
wave edge -- against the point-by-point kernels on the same points.  A folded launch rounds differently (fp32 column terms): the bar is the tests' 2e-5 (avatar
occupancy and offsets) / 5e-6 (recon); with the folding off it is bit for bit.  The same indices in reverse order must give the same bits (a point's value does not
depend on its place in a launch; recon: 1e-6, which kernel evaluates a point depends on its tile).
    python tests/tools/band_fuzz_gpu.py [cases] [seed]          (needs an MI355X; a checker script, not part of the product path)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from avatarcap_amd import _lib, config, synthetic as syn               # noqa: E402
config.cfg = config.default_cfg(); config.device = torch.device('cuda')
import golden_inputs as gi                                             # noqa: E402
from common import geotex_sd, recon_sd, maxabs                         # noqa: E402
from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet   # noqa: E402
from avatarcap_amd.network.arch_recon import ReconNetwork              # noqa: E402
from avatarcap_amd.grid import generate_volume_points_np, volume_axes  # noqa: E402


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def indices(rs, res, kind):
    n_all = res[0] * res[1] * res[2]
    cols = res[0] * res[1]
    if kind == 'band':                                                   # a run of consecutive z per column, a random subset of the columns
        out = []
        for c in np.nonzero(rs.rand(cols) < rs.uniform(0.1, 1.0))[0]:
            a = rs.randint(0, res[2]); b = rs.randint(a, res[2]) + 1
            out.append(c * res[2] + np.arange(a, b))
        idx = np.concatenate(out) if out else np.zeros(0, np.int64)
    elif kind == 'short_runs':                                           # 1 .. 5 points per column: up to 32 runs in a wavefront
        k = rs.randint(1, 6)
        idx = (np.arange(cols)[:, None] * res[2] + rs.randint(0, max(1, res[2] - k)) + np.arange(k)[None, :]).reshape(-1)
        idx = idx[idx < n_all]
    elif kind == 'scatter':
        idx = rs.choice(n_all, rs.randint(1, min(n_all, 4000) + 1), replace=False)
    elif kind == 'dense':
        idx = np.arange(n_all)
    else:
        raise ValueError(kind)
    how = rs.randint(0, 5)
    if how == 1:
        idx = idx[::-1]
    elif how == 2:
        idx = rs.permutation(idx)
    elif how == 3 and idx.size > 4:                                      # the first part comes back later: columns met again, duplicated indices
        idx = np.concatenate([idx, idx[: rs.randint(1, idx.size)]])
    if idx.size > 3 and rs.rand() < 0.5:                                 # ragged: not a multiple of anything
        idx = idx[: idx.size - rs.randint(0, min(200, idx.size - 1))]
    return idx[:n_all].astype(np.int32)                                  # (the subset entries take at most as many indices as the grid has points)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 808)
    config.if_type = 'sdf'
    net = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume()).to('cuda').eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in geotex_sd().items()})
    fmap = gi.pose_feat_map()
    net.warping_field.pose_feat_map = _t(fmap[None])
    occ = OccupancyNet(net)
    rn = ReconNetwork().to('cuda').eval()
    rn.load_state_dict({k: torch.from_numpy(v) for k, v in recon_sd().items()})
    imap = _t(gi.img_feat_map(seed=212)[None])
    center = _t(gi.center()[None])
    bad, worst, t0, npts = 0, [0.0, 0.0, 0.0], time.time(), 0
    for k in range(cases):
        res = (int(rs.randint(1, 9)), int(rs.randint(1, 9)), int(rs.choice([1, 7, 31, 32, 33, 48, 64, 100, 128, 130, 256])))
        kind = ['band', 'short_runs', 'scatter', 'dense'][rs.randint(0, 4)]
        idx = indices(rs, res, kind)
        if idx.size == 0:
            continue
        allp = generate_volume_points_np(syn.CANO_BOUNDS, res)
        pts = allp[idx]
        npts += idx.size
        batch = {'cano_pts': _t(pts[None]), 'cano_smpl_center': center}
        index = torch.from_numpy(idx).cuda()
        ax = volume_axes(syn.CANO_BOUNDS, res, 'cuda')
        why = []
        blocks = int(rs.choice([0, 0, 1, 3, 7]))                         # persistent workgroups: few of them walk many tiles each (the index two tiles ahead, bench-size launches)
        _lib.set_option('mlp_blocks', blocks)
        a = occ.query(batch)
        g = occ.query_grid(batch, ax, res, want_offset=True, index=index)
        d_occ, d_off = maxabs(g['cano_pts_ov'].cpu().numpy(), a['cano_pts_ov'].cpu().numpy()), maxabs(g['nonrigid_offset'].cpu().numpy(), a['nonrigid_offset'].cpu().numpy())
        if not (d_occ < 2e-5 and d_off < 2e-5):
            why.append(f'avatar folded vs points: occ {d_occ:.2e} off {d_off:.2e}')
        if idx.size == allp.shape[0] and np.array_equal(idx, np.arange(idx.size)):          # the whole grid in grid order: also as the dense launches
            gd = occ.query_grid(batch, ax, res, want_offset=True)
            dd = max(maxabs(gd['cano_pts_ov'].cpu().numpy(), a['cano_pts_ov'].cpu().numpy()), maxabs(gd['nonrigid_offset'].cpu().numpy(), a['nonrigid_offset'].cpu().numpy()))
            rd = maxabs(rn.decode_grid(ax, res, imap, center).cpu().numpy().reshape(-1), rn.decode(_t(pts[None]), imap, center).cpu().numpy().reshape(-1))
            if not (dd < 2e-5 and rd < 5e-6):
                why.append(f'dense launches vs points: avatar {dd:.2e} recon {rd:.2e}')
        back = occ.query_grid(batch, ax, res, index=torch.flip(index, [0]).contiguous())
        if not torch.equal(torch.flip(back['cano_pts_ov'], [1]), g['cano_pts_ov']):
            why.append('avatar: reversed order gives other bits')
        _lib.set_option('column_fold', 0)
        try:
            u = occ.query_grid(batch, ax, res, want_offset=True, index=index)
            ru = rn.decode_grid(ax, res, imap, center, index=index)
        finally:
            _lib.set_option('column_fold', 1)
        if not (torch.equal(u['cano_pts_ov'], a['cano_pts_ov']) and torch.equal(u['nonrigid_offset'], a['nonrigid_offset'])):
            why.append('avatar: unfolded grid launch != point query')
        ra = rn.decode(_t(pts[None]), imap, center)
        rg = rn.decode_grid(ax, res, imap, center, index=index)
        d_rec = maxabs(rg.cpu().numpy().reshape(-1), ra.cpu().numpy().reshape(-1))
        if not d_rec < 5e-6:
            why.append(f'recon folded vs points: {d_rec:.2e}')
        if not torch.equal(ru.reshape(-1), ra.reshape(-1)):
            why.append('recon: unfolded grid launch != point decode')
        rb = rn.decode_grid(ax, res, imap, center, index=torch.flip(index, [0]).contiguous())
        d_rev = maxabs(torch.flip(rb.reshape(-1), [0]).cpu().numpy(), rg.cpu().numpy().reshape(-1))
        if not d_rev < 2e-6:
            why.append(f'recon: reversed order differs by {d_rev:.2e}')
        worst = [max(worst[0], d_occ), max(worst[1], d_off), max(worst[2], d_rec)]
        if why:
            bad += 1
            print(f'case {k}: res {res} {kind} n {idx.size} workgroups {blocks or "one per CU"}: {why}')
    _lib.set_option('mlp_blocks', 0)
    print(f'{cases} cases, {npts} points, worst folded-vs-points avatar occ {worst[0]:.2e} off {worst[1]:.2e} recon {worst[2]:.2e}, {bad} failures, {time.time() - t0:.0f} s')
    return 1 if bad else 0


if __name__ == '__main__':
    raise SystemExit(main())
