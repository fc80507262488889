"""Can the canonical normal fusion (a chain of ~115 small launches, the iteration kernels without LDS and with <= 64 VGPRs) run BESIDE the next frame's avatar
query on the same CUs?  The persistent query workgroups hold all of a CU's LDS but 408 of its 512 registers per SIMD lane: an LDS-free wave of <= 104 registers
can be co-resident.  Band frame at 256^3 (BASELINE configs[2]): the query alone, the fusion alone, one after the other on one stream, and the two on two streams."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from avatarcap_amd import config, synthetic as syn
from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda, synthetic_camera, synthetic_observed_normals
from avatarcap_amd.network.arch_avatar import GeoTexAvatar
from avatarcap_amd.pipeline import FramePipeline
dev = torch.device('cuda'); config.device = dev; config.cfg = config.default_cfg()
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to(dev).eval(); syn.load_synth(net, syn.SEED)
config.cfg['testing']['vol_res'] = [256] * 3
ds = SyntheticTestDataset([256] * 3, valid='band', n_frames=2)
pipe = FramePipeline(net, ds)
it0, it1 = to_cuda(ds[0], add_batch=True), to_cuda(ds[1], add_batch=True)
a = pipe.avatar_frame(it0)
w2c, cam = synthetic_camera()
obs = synthetic_observed_normals(a['live_v'], a['live_vn'], a['f'], w2c, cam, seed=0)
net.warping_field.precompute_conv(it1)
side = torch.cuda.Stream(dev)


def query():
    return pipe._avatar_query(it1)


def fusion():
    return pipe.fuse_normals(a, obs, w2c, cam, 'merge')


def both():
    ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(dev))
    q = query()                                   # main stream: ~11 ms of one persistent launch
    with torch.cuda.stream(side):
        side.wait_event(ev)
        f = fusion()
    torch.cuda.current_stream(dev).wait_stream(side)
    return q, f


def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def seq():
    query(); fusion()


tq, tf, ts, tb = timed(query), timed(fusion), timed(seq), timed(both)
print(f'avatar band query alone {tq:.3f} ms; fusion alone {tf:.3f} ms; one after the other {ts:.3f} ms; on two streams {tb:.3f} ms (saves {ts - tb:.3f} of the fusion\'s {tf:.3f})')
q1 = query(); f1 = fusion(); torch.cuda.synchronize()
q2, f2 = both(); torch.cuda.synchronize()
print('same results:', bool(torch.equal(q1['cano_pts_ov'], q2['cano_pts_ov'])), bool(torch.equal(f1[0], f2[0])))

# ---- do LDS-free waves run beside the query at all when it holds every CU?  A chain of 100 small element-wise launches (no LDS, a few VGPRs) on the side stream
x = torch.zeros(512 * 512, device=dev)


def chain(n=100):
    for _ in range(n):
        x.add_(1.0)


t_chain = timed(chain)
for spare in (0, 8):
    from avatarcap_amd import _lib
    _lib.set_option('mlp_blocks', 0 if spare == 0 else 256 - spare)
    query(); torch.cuda.synchronize()
    e0, e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream(dev))
    query()
    with torch.cuda.stream(side):
        side.wait_event(e0)
        chain()
        e1.record(side)
    e2.record(torch.cuda.current_stream(dev))
    torch.cuda.synchronize()
    print(f'spare CUs {spare}: chain of 100 small launches alone {t_chain:.3f} ms; beside the query it ends {e0.elapsed_time(e1):.3f} ms after the query starts, the query after {e0.elapsed_time(e2):.3f} ms')
_lib.set_option('mlp_blocks', 0)
