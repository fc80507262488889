"""Time split of the fused avatar query from in-kernel s_memtime stamps (an AVC_DBG_TIMING=2 build of libavcap_hip.so; WRONG offsets by design).

    ABL_VARIANTS="F:-DAVC_DBG_TIMING=2" bash tools/ablate_build.sh
    AVCAP_LIB=$PWD/avatarcap_amd/csrc/_abl/lib_F_-DAVC_DBG_TIMING_2.so python tools/timing_probe.py [out.md]

The dense 256^3 launch of the frame loop (column-folded, points from the grid index).  Every wave stamps (entry, own LDS-DMA pieces drained,
barrier released) at each of the 64 chunk steps of its workgroup's LAST tile, plus the tile's start and end; the launch total, the time at
chunk entries and the prologue time are accumulated over all tiles.  Cycles are shader cycles (s_memtime), means over the 1024 waves.
"""
import sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import config, synthetic as syn
config.cfg = config.default_cfg()
from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
from avatarcap_amd.grid import volume_axes

RES = 256
# chunk steps of one tile of avatar_kernel<true,false,1>, in launch order: (label, class, MFMAs)
SEQ = []
def layer(name, cls, pairs, ks0, ks1=0):
    ka = (ks0 + ks1 + 1) // 2 if ks1 else ks0         # a pair's k-steps are walked as two chunks of about equal size (fused_mlp.hip dense())
    for p in range(pairs):
        SEQ.append((f'{name} pair {p}' + (' (first half)' if ks1 else ''), cls, 6 * ka))
        if ks1:
            SEQ.append((f'{name} pair {p} (second half)', cls, 6 * (ks0 + ks1 - ka)))
SP, RL = 'Softplus layer', 'ReLU layer'
SEQ.append(('conv1 (xyz k-step, 8 tiles)', 'conv1: wide chunk (+ pair 0 epilogue exposed)', 24))
for i in (2, 3, 4):
    layer(f'conv{i}', SP, 4, 16)
layer('conv5', SP, 4, 16, 1)
for i in (6, 7):
    layer(f'conv{i}', SP, 4, 16)
SEQ.append(('offset head (+ positional encoding behind it)', 'one-tile heads', 48))
SEQ.append(('shared.0 (positional encoding, 8 tiles)', 'shared.0: wide chunk (+ pair 0 epilogue exposed)', 96))
for i in (1, 2, 3):
    layer(f'shared.{i}', RL, 4, 16)
layer('shared.4', RL, 4, 16, 4)
layer('shared.5', RL, 4, 16)
layer('geo.0 o shared.6', RL, 2, 16)
SEQ.append(('geo.1 head (+ output store)', 'one-tile heads', 24))


RSEQ = [('fc0 rows 0..255 (z k-step, 8 tiles) + flush pair 0', 'fc0 wide chunks', 24)]
RSEQ += [(f'fc1 over x[{64 * c}..{64 * c + 63}]', 'fc1 chunks (4 k-steps x 8 tiles)', 96) for c in range(4)]
RSEQ += [('[fc0 rows 256..511 | fc1 z] + flush pair 0', 'fc0 wide chunks', 48)]
RSEQ += [(f'fc1 over x[{256 + 64 * c}..{256 + 64 * c + 63}]', 'fc1 chunks (4 k-steps x 8 tiles)', 96) for c in range(4)]
RSEQ += [('fc2 pair 0 first half (+ fc1 pairs 1..3 epilogue)', 'fc2', 54), ('fc2 pair 0 second half', 'fc2', 48), ('fc2 pair 1 first half', 'fc2', 54),
         ('fc2 pair 1 second half', 'fc2', 48), ('fc3 head (+ sigmoid, store, next tile\'s column loads)', 'fc3 head', 24)]


def recon_main():
    global SEQ
    SEQ = RSEQ
    sys.path.insert(0, 'tests')
    import golden_inputs as gi
    from common import recon_sd
    from avatarcap_amd.network.arch_recon import ReconNetwork
    rn = ReconNetwork().to('cuda').eval()
    rn.load_state_dict({k: torch.from_numpy(v) for k, v in recon_sd().items()})
    imap = torch.from_numpy(gi.img_feat_map()[None]).cuda()
    center = torch.from_numpy(gi.center()[None]).cuda()
    axes = volume_axes(syn.CANO_BOUNDS, (RES,) * 3, 'cuda')
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(2):
        ev[0].record()
        y = rn.decode_grid(axes, (RES,) * 3, imap, center)
        ev[1].record()
    torch.cuda.synchronize()
    report(y.view(-1).view(torch.int64), ev[0].elapsed_time(ev[1]), 'recon_fold_kernel<1>` (dense 256^3 grid, column-folded; with its column pass')


def main():
    net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to('cuda').eval()
    sd = syn.synth_state_dict(syn.module_shapes(net), syn.SEED)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.warping_field.pose_feat_map = torch.randn(1, 64, 256, 256, device='cuda')
    axes = volume_axes(syn.CANO_BOUNDS, (RES,) * 3, 'cuda')
    batch = {'cano_smpl_center': torch.zeros(1, 3, device='cuda')}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(2):
        ev[0].record()
        o = OccupancyNet(net).query_grid(batch, axes, (RES,) * 3, want_offset=True)
        ev[1].record()
    torch.cuda.synchronize()
    report(o['nonrigid_offset'].view(-1).view(torch.int64), ev[0].elapsed_time(ev[1]), 'avatar_kernel<true,false,1>` (dense 256^3, column-folded')


def report(raw, ms, what):
    nw = 1024
    summ = raw[:4 * nw].cpu().numpy().reshape(nw, 4).astype(np.float64)
    st = raw[8192:8192 + nw * 256].cpu().numpy().reshape(nw, 256)
    tiles_per_wave = RES ** 3 // 128 // 256
    mfma_tile = sum(m for _, _, m in SEQ)
    tot = summ[:, 0].mean()
    lines = []
    P = lines.append
    P(f'# Time split of `{what}), s_memtime stamps of an `AVC_DBG_TIMING=2` build -- round 3')
    P('')
    P(f'`tools/timing_probe.py`: launch {ms:.2f} ms (instrumented build), {tot:.4e} shader cycles per wave = {tot / ms / 1e3:.0f} MHz, '
      f'{tot / (mfma_tile * tiles_per_wave):.2f} cycles per MFMA ({mfma_tile} MFMAs x {tiles_per_wave} tiles per wave); wave-to-wave spread of the total '
      f'{summ[:, 0].min() / tot - 1:+.2%} .. {summ[:, 0].max() / tot - 1:+.2%}.')
    P('')
    P('Accumulated over ALL tiles (mean over the 1024 waves):')
    P('')
    P('| | cycles per tile | share of the launch |')
    P('|---|---:|---:|')
    P(f'| whole tile | {tot / tiles_per_wave:.0f} | 100 % |')
    P(f'| chunk entries: `s_waitcnt vmcnt(0)` on the wave\'s own LDS-DMA pieces | {summ[:, 2].mean() / tiles_per_wave:.0f} | {100 * summ[:, 2].mean() / tot:.2f} % |')
    P(f'| chunk entries: workgroup barrier | {(summ[:, 1] - summ[:, 2]).mean() / tiles_per_wave:.0f} | {100 * (summ[:, 1] - summ[:, 2]).mean() / tot:.2f} % |')
    P(f'| prologue sections (point / axis loads, xyz split; positional encoding after the offset head) | {summ[:, 3].mean() / tiles_per_wave:.0f} | {100 * summ[:, 3].mean() / tot:.2f} % |')
    P('')
    n = st[:, :].copy()
    nst = len(SEQ) + 2                       # opening stamp, one per chunk step, closing stamp
    cnt = st[:, 3 * (nst - 1) + 2]
    ok = cnt == nst - 1
    P(f'Last tile of every workgroup, per chunk step ({int(ok.sum())} of {nw} waves stamped all {nst - 1} steps; mean cycles over those waves; '
      f'`body` = barrier release -> entry of the next step, i.e. MFMAs + everything issued in their shadow):')
    P('')
    s3 = st[ok, :3 * nst].reshape(-1, nst, 3).astype(np.float64)
    entry, drained, released = s3[:, 1:-1, 0], s3[:, 1:-1, 1], s3[:, 1:-1, 2]
    nxt = np.concatenate([s3[:, 2:-1, 0], s3[:, -1:, 0]], axis=1)
    drain = (drained - entry).mean(0); bar = (released - drained).mean(0); body = (nxt - released).mean(0)
    pro = (s3[:, 1, 0] - s3[:, 0, 0]).mean()
    tile_total = (s3[:, -1, 0] - s3[:, 0, 0]).mean()
    P('| # | chunk step | MFMAs | drain | barrier | body | body cycles per MFMA |')
    P('|---:|---|---:|---:|---:|---:|---:|')
    for i, (name, cls, m) in enumerate(SEQ):
        P(f'| {i} | {name} | {m} | {drain[i]:.0f} | {bar[i]:.0f} | {body[i]:.0f} | {body[i] / m:.1f} |')
    P('')
    P(f'Tile start -> first chunk entry (prologue: axis loads, xyz split, bias queue): {pro:.0f} cycles.  Whole last tile: {tile_total:.0f} cycles '
      f'(all tiles: {tot / tiles_per_wave:.0f}).')
    P('')
    P('By class (sum over the tile):')
    P('')
    P('| class | steps | MFMAs | drain + barrier | body | body per MFMA | share of the tile | cycles above 32 per MFMA |')
    P('|---|---:|---:|---:|---:|---:|---:|---:|')
    classes = []
    for _, cls, _ in SEQ:
        if cls not in classes:
            classes.append(cls)
    for cls in classes:
        idx = [i for i, (_, c, _) in enumerate(SEQ) if c == cls]
        m = sum(SEQ[i][2] for i in idx)
        db = sum(drain[i] + bar[i] for i in idx); b = sum(body[i] for i in idx)
        P(f'| {cls} | {len(idx)} | {m} | {db:.0f} | {b:.0f} | {b / m:.1f} | {100 * (db + b) / tile_total:.1f} % | {db + b - 32 * m:.0f} |')
    P(f'| prologue before the first chunk | | 0 | | {pro:.0f} | | {100 * pro / tile_total:.1f} % | {pro:.0f} |')
    P(f'| **tile** | {len(SEQ)} | {mfma_tile} | {drain.sum() + bar.sum():.0f} | {body.sum():.0f} | {body.sum() / mfma_tile:.1f} | 100 % | {tile_total - 32 * mfma_tile:.0f} |')
    text = '\n'.join(lines) + '\n'
    print(text)
    out = [a for a in sys.argv[1:] if a != '--recon']
    if out:
        open(out[0], 'w').write(text)


if __name__ == '__main__':
    recon_main() if '--recon' in sys.argv else main()
