"""Who corrupts whom?  Victims on the main stream -- marching cubes + normals of a fixed volume, then LBS + skinning of its mesh, three times per iteration --
beside an aggressor on a side stream (the look-ahead U-Net as FramePipeline.avatar_frame runs it, the HGFilter, plain element-wise launches, or nothing); every
victim output compared bit for bit with the quiet run.  Round 6 found 94 corrupted outputs in 600 beside the U-Net: one x component in the last 16 lanes of a wave,
a packed-f32 VALU result read as store data an instruction later (csrc/store_settle.h, profiles/r06_store_hazard.md).  Exit status 1 when anything differs.
    python tools/race_probe.py [unet|hgfilter|torch|none] [iterations] [option=value ...]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import config, synthetic as syn, _lib
dev = torch.device('cuda'); config.device = dev; config.cfg = config.default_cfg()
from avatarcap_amd.network.arch_avatar import GeoTexAvatar
from avatarcap_amd.network.arch_recon import ReconNetwork
from avatarcap_amd.utils.smpl_util import SmplUtil
which = sys.argv[1] if len(sys.argv) > 1 else 'unet'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for kv in sys.argv[3:]:
    k, v = kv.split('='); _lib.set_option(k, int(v))
body = syn.make_body(syn.SEED) if hasattr(syn, 'make_body') else None
rs = np.random.RandomState(1)
nv = 20000
pts = torch.from_numpy(rs.uniform(-0.5, 0.5, (1, nv, 3)).astype(np.float32)).cuda()
nrm = torch.from_numpy(rs.randn(1, nv, 3).astype(np.float32)).cuda()
lbs = torch.softmax(torch.from_numpy(rs.randn(1, nv, 24).astype(np.float32)).cuda() * 3, -1).contiguous()
jm = torch.from_numpy(rs.randn(1, 24, 4, 4).astype(np.float32)).cuda()
su = SmplUtil(np.abs(rs.randn(6890, 24)).astype(np.float32))
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to(dev).eval(); syn.load_synth(net, syn.SEED)
rn = ReconNetwork().to(dev).eval(); syn.load_synth(rn, syn.SEED)
pos_map = torch.from_numpy(rs.randn(1, 6, 256, 256).astype(np.float32)).cuda()
img = torch.from_numpy(rs.randn(1, 6, 512, 512).astype(np.float32)).cuda()
big = torch.zeros(1 << 22, device=dev)


from avatarcap_amd.utils import recon_util
from avatarcap_amd.dataset import SyntheticTestDataset
ds = SyntheticTestDataset([48, 64, 32], valid='band', n_frames=1)
su.set_cano_smpl_vertices(ds.cano_smpl_v); su.set_smpl_skinning_weights(ds.body['skin_weights'])
g = [np.linspace(-1, 1, r, dtype=np.float32) for r in (96, 128, 64)]
X, Y, Z = np.meshgrid(*g, indexing='ij')
vol = torch.from_numpy((0.55 - np.sqrt(X * X + Y * Y * 0.6 + Z * Z) + 0.05 * np.sin(9 * X) * np.cos(7 * Y)).astype(np.float32)).cuda()


def victim():
    v, f, n = recon_util.recon_mesh_device(vol, [96, 128, 64], ds.cano_bounds, iso_value=0.0)
    po, no, mo, _ = su.lbs_skinning(v[None], n[None], jm, return_pt_mats=True)
    return v, n, po, no, mo


def aggressor():
    if which == 'unet':
        return net.warping_field.unet(pos_map)
    if which == 'hgfilter':
        return rn.image_filter(img) if hasattr(rn, 'image_filter') else rn.get_feat_maps(img)
    if which == 'torch':
        for _ in range(20):
            big.mul_(1.0001).add_(1e-3)
    return None


ref = victim(); aref = aggressor(); torch.cuda.synchronize()
aref = aref.clone() if isinstance(aref, torch.Tensor) else (aref[-1].clone() if isinstance(aref, (list, tuple)) and len(aref) and isinstance(aref[-1], torch.Tensor) else None)
abad = 0
side = torch.cuda.Stream(dev)
bad, t0 = 0, time.time()
for it in range(iters):
    ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(dev))
    if which != 'none':
        with torch.cuda.stream(side):
            side.wait_event(ev)
            aout = aggressor()
    outs = [victim() for _ in range(3)]
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    if which != 'none' and aref is not None:          # the aggressor is a victim too: its own output beside the main stream's kernels
        ao = aout if isinstance(aout, torch.Tensor) else aout[-1]
        if not torch.equal(ao, aref):
            abad += 1
            if abad <= 3:
                print(f'iteration {it}: the {which} output differs from its quiet run in {int((ao != aref).sum())} values')
    for o in outs:
        for name, a, b in zip(('v', 'n', 'po', 'no', 'mo'), o, ref):
            if not torch.equal(a, b):
                d = (a != b).reshape(a.shape[-2] if a.dim() > 2 and name != 'mo' else (a.shape[0] if a.dim() == 2 else a.shape[1]), -1)
                rows = torch.nonzero(d.any(1))[:, 0]
                cols = torch.nonzero(d.any(0))[:, 0]
                bad += 1
                if bad <= 4 and name in ('no', 'po', 'n') and cols.tolist() == [0]:
                    a2, b2 = a.reshape(-1, 3), b.reshape(-1, 3)
                    for r_ in rows[:16].tolist():
                        same = torch.nonzero(b2[:, 0] == a2[r_, 0])[:, 0].tolist()
                        same_any = [(int(t // 3), int(t % 3)) for t in torch.nonzero(b2.reshape(-1) == a2[r_, 0])[:, 0].tolist()][:4]
                        print(f'    row {r_}: got x {a2[r_, 0].item():+.7f} want {b2[r_, 0].item():+.7f}; rows of the reference with that x: {same[:4]}; anywhere (row, col): {same_any}')
                if bad <= 6:
                    print(f'iteration {it}: {name} differs in rows {rows[0].item()}..{rows[-1].item()} ({rows.numel()} rows; first row mod 64 = {rows[0].item() % 64}), columns {cols.tolist()}')
print(f'aggressor {which} {sys.argv[3:]}: {bad} corrupted victim outputs in {iters} iterations x 3, {abad} corrupted aggressor outputs ({time.time() - t0:.1f} s)')
raise SystemExit(1 if (bad or abad) else 0)
