"""Parity of the fused queries over MANY synthetic networks (the GPU tests hold four seeds / gains): for every (seed, gain) a fresh
GeoTexAvatar and ReconNetwork from the seeded recipe (avatarcap_amd.synthetic), a fresh pose / image feature map and fresh query points; the HIP point queries
against the fp64 oracle, beside what the reference's own fp32 arithmetic loses on the same network (fp32 oracle vs fp64 oracle).  Prints one markdown table row per
network and the worst case; exit status 1 when a network misses the bar of the GPU tests (1e-4 x max(1, |occ|max) + 2 x the fp32 slack; offsets and recon 1e-4).
    python tests/tools/parity_sweep.py [networks] [points]           (needs an MI355X; a checker script, not part of the product path)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from avatarcap_amd import config, synthetic as syn                     # noqa: E402
config.cfg = config.default_cfg(); config.device = torch.device('cuda')
import golden_inputs as gi                                             # noqa: E402
from common import geotex_shapes, maxabs                              # noqa: E402
from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet   # noqa: E402
from avatarcap_amd.network.arch_recon import ReconNetwork              # noqa: E402
from oracle import avatarcap_oracle as orc                             # noqa: E402

TOL = 1e-4


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def main():
    n_nets = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    n_pts = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    rs = np.random.RandomState(606)
    gains = [0.6, 1.0, 1.4, 1.8, 2.2]
    rn = ReconNetwork().to('cuda').eval()
    rshapes = syn.module_shapes(rn)
    print('| seed | gain | occ scale | occ err / scale | fp32 slack / scale | offsets err | recon err | verdict |')
    print('|---:|---:|---:|---:|---:|---:|---:|---|')
    worst, bad = [0.0, 0.0, 0.0], 0
    for k in range(n_nets):
        seed, gain = int(rs.randint(1, 1 << 30)), gains[k % len(gains)]
        sd = syn.synth_state_dict(geotex_shapes(), seed, gain=gain)
        net = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume()).to('cuda').eval()
        net.load_state_dict({a: torch.from_numpy(b) for a, b in sd.items()})
        fmap = gi.pose_feat_map(seed=seed % 1000)
        net.warping_field.pose_feat_map = _t(fmap[None])
        pts = gi.query_points(seed % 100000, n_pts)
        batch = {'cano_pts': _t(pts[None]), 'cano_smpl_center': _t(gi.center()[None])}
        out = OccupancyNet(net).query(batch)
        r64 = orc.occupancy_query(pts, fmap, gi.center(), sd)
        r32 = orc.occupancy_query(pts, fmap, gi.center(), sd, dt=np.float32)
        scale = max(1.0, float(np.abs(r64['cano_pts_ov']).max()))
        slack = maxabs(r32['cano_pts_ov'], r64['cano_pts_ov'])
        e_occ = maxabs(out['cano_pts_ov'][0].cpu().numpy(), r64['cano_pts_ov'])
        e_off = maxabs(out['nonrigid_offset'][0].cpu().numpy(), r64['nonrigid_offset'])
        rsd = syn.synth_state_dict(rshapes, seed, gain=gain)
        rn.load_state_dict({a: torch.from_numpy(b) for a, b in rsd.items()})
        imap = gi.img_feat_map(seed=seed % 997)
        y = rn.decode(_t(pts[None]), _t(imap[None]), _t(gi.center()[None]))
        e_rec = maxabs(y.cpu().numpy().reshape(-1), orc.recon_infer(pts, imap, gi.center(), rsd))
        ok = e_occ < TOL * scale + 2 * slack and e_off < TOL and e_rec < TOL
        bad += not ok
        worst = [max(worst[0], e_occ / scale), max(worst[1], e_off), max(worst[2], e_rec)]
        print(f'| {seed} | {gain} | {scale:.1f} | {e_occ / scale:.2e} | {slack / scale:.2e} | {e_off:.2e} | {e_rec:.2e} | {"ok" if ok else "MISS"} |', flush=True)
    print(f'\n{n_nets} networks x {n_pts} points: worst occupancy error / scale {worst[0]:.2e}, worst offsets error {worst[1]:.2e}, worst recon error {worst[2]:.2e}; {bad} misses')
    return 1 if bad else 0


if __name__ == '__main__':
    raise SystemExit(main())
