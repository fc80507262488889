"""Per-layer pass budget of the fused query.

The kernel forms every product as three fp16 MFMA passes, w_hi*x_hi + w_hi*x_lo + w_lo*x_hi.  Dropping the w_lo*x_hi pass of ONE layer is the
same as rounding that layer's weights to fp16 (11 significant bits); dropping w_hi*x_lo rounds its input activations.  This tool perturbs one
affine layer at a time in the fp64 oracle and reports what the occupancy (and the warping offset) move by, on the golden network and on the
three other seeds / gains tests/test_gpu_query.py uses.  A layer could run on two passes if its worst case stayed under a 5e-5 budget.

    python tests/tools/pass_budget.py            (CPU, ~2 min)
"""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import synthetic as syn
import golden_inputs as gi
from common import geotex_sd, geotex_shapes
from oracle import avatarcap_oracle as orc

BUDGET = 5e-5
N = 2048


def f16(a):
    return a.astype(np.float16).astype(np.float64)


def run(pts, fmap, sd, hook):
    orc._matmul = hook
    try:
        r = orc.occupancy_query(pts, fmap, gi.center(), sd)
    finally:
        orc._matmul = plain
    return r['cano_pts_ov'], r['nonrigid_offset']


def plain(inp, W, tag):
    return inp @ W.T


def main():
    fmap = gi.pose_feat_map()
    nets = [('golden', geotex_sd(), gi.query_points(11, N))]
    for seed, gain in ((7, 1.6), (123, 1.0), (2024, 2.2)):
        nets.append((f'seed {seed} gain {gain}', syn.synth_state_dict(geotex_shapes(), seed, gain=gain), gi.query_points(300 + seed, N)))
    tags = []
    def collect(inp, W, tag):
        tags.append((tag, W.shape))
        return inp @ W.T
    run(nets[0][2], fmap, nets[0][1], collect)
    seen, layers = set(), []
    for t, shp in tags:
        if t not in seen:
            seen.add(t); layers.append((t, shp))
    rows = []
    for tag, shp in layers:
        row = {'layer': tag, 'shape': shp}
        for mode in ('w', 'x'):
            worst_occ = worst_off = worst_rel = 0.0
            gold = None
            for name, sd, pts in nets:
                occ0, off0 = run(pts, fmap, sd, plain)
                def hook(inp, W, t, tag=tag, mode=mode):
                    if t != tag:
                        return inp @ W.T
                    return (inp @ f16(W).T) if mode == 'w' else (f16(inp) @ W.T)
                occ, off = run(pts, fmap, sd, hook)
                scale = max(1.0, float(np.abs(occ0).max()))
                worst_occ = max(worst_occ, float(np.abs(occ - occ0).max()))
                worst_rel = max(worst_rel, float(np.abs(occ - occ0).max()) / scale)
                worst_off = max(worst_off, float(np.abs(off - off0).max()))
                if gold is None:
                    gold = (float(np.abs(occ - occ0).max()), float(np.abs(off - off0).max()))
            row[mode] = (worst_occ, worst_rel, worst_off)
            row[mode + '_gold'] = gold
        rows.append(row)
        print('%-42s %-8s | drop w_lo*x_hi: golden net occ %.1e off %.1e, worst of 4 nets occ/scale %.1e off %.1e | drop w_hi*x_lo: golden occ %.1e off %.1e, worst %.1e off %.1e' %
              (row['layer'], 'x'.join(map(str, row['shape'])), *row['w_gold'], row['w'][1], row['w'][2], *row['x_gold'], row['x'][1], row['x'][2]), flush=True)
    ok = [(r['layer'], m) for r in rows for m in ('w', 'x') if max(r[m][1], r[m][2]) < BUDGET]
    macs = sum(r['shape'][0] * r['shape'][1] for r in rows)
    saved = sum(r['shape'][0] * r['shape'][1] for r in rows if any(max(r[m][1], r[m][2]) < BUDGET for m in ('w', 'x')))
    print('\nlayers with a pass to spare under %.0e (occupancy relative to max(1, |occ|max), offsets absolute): %s' % (BUDGET, ok or 'none'))
    print('MACs in such layers: %d of %d (%.1f %%) -> MFMA passes saved: %.1f %%' % (saved, macs, 100 * saved / macs, 100 * saved / macs / 3))


if __name__ == '__main__':
    main()
