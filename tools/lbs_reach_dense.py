"""tools/lbs_reach_probe.py on the DENSE stress frame (bench.py's headline frame: a surface that fills the whole volume, most of it far from the
body): LBS time and list size per reach of the per-cell candidate lists."""
import ctypes as C, sys, time
import torch
sys.path.insert(0, '.')
import bench
from avatarcap_amd import _lib
from avatarcap_amd.dataset import to_cuda
from avatarcap_amd.utils.smpl_util import smpl_util
dev = torch.device('cuda', 0)
pipe, _ = bench.build_pipeline(256, 'dense', 1, dev)
out = pipe.avatar_frame(to_cuda(pipe.ds[0], add_batch=True))
v = out['cano_v'][None].contiguous()
d2, _ = smpl_util.knn_points(v, smpl_util.cano_smpl_vertices[None], K=4)
d4 = d2[0, :, 3].sqrt()
q = [float(d4.quantile(x)) for x in (0.25, 0.5, 0.75, 0.9, 0.99)]
print(f'{v.shape[1]} vertices; distance to the 4th nearest SMPL vertex: quartiles {q[0]:.3f} {q[1]:.3f} {q[2]:.3f}, 90 % {q[3]:.3f}, 99 % {q[4]:.3f}, max {float(d4.max()):.3f} m')
ctx = _lib.ctx(dev)
ref = None
for reach in [int(a) for a in sys.argv[1:]] or (0, 140, 250, 400, 600, 1000):
    _lib.set_option('lbs_reach_mm', reach)
    _lib.set_owner(ctx, 'lbs_bound', None)
    t = time.perf_counter(); smpl_util.set_cano_smpl_vertices(smpl_util.cano_smpl_vertices); torch.cuda.synchronize(); tp = time.perf_counter() - t
    st = (C.c_int64 * 4)(); _lib.check(_lib.lib().avc_lbs_bound_stats(ctx, st))
    lbs = smpl_util.calculate_lbs(v); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): lbs = smpl_util.calculate_lbs(v)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    ref = lbs if ref is None else ref
    n1, jm = out['cano_vn'][None].contiguous(), to_cuda(pipe.ds[0], add_batch=True)['cano2live_jnt_mats']
    def timed(fn, reps=10):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e6
    t_sk = timed(lambda: smpl_util.skinning(v, lbs, jm, True)); t_sn = timed(lambda: smpl_util.skinning_normal(n1, lbs, jm))
    t_f = timed(lambda: smpl_util.lbs_skinning(v, n1, jm, return_pt_mats=True)); t_fp = timed(lambda: smpl_util.lbs_skinning(v, None, jm))
    print(f'          skinning + mats {t_sk:6.1f} us, skinning_normal {t_sn:6.1f} us; fused launch (points, normals, mats) {t_f:6.1f} us, fused points only {t_fp:6.1f} us', flush=True)
    print(f'reach {reach:4d} mm: prepare {tp*1e3:6.1f} ms, {st[1]} cells, {st[2]} list entries ({st[2]*16/1e6:.0f} MB); calculate_lbs {dt*1e6:7.1f} us; identical to reach 0: {bool(torch.equal(lbs, ref))}', flush=True)
