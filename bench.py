#!/usr/bin/env python3
"""bench.py -- reconstructed-mesh frames/sec at 256^3 (BASELINE.json metric), 1..8 MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 8 --frames 64                     (BASELINE configs[4]: a batch of 64 frames sharded 8-way, meshes all-gathered)
    python bench.py --gpus 2 --dry-run                       (no GPU: launcher + gloo all-gather of stand-in meshes; what the CPU test runs)

A step = one frame of BASELINE.json configs[1] ("AvatarNet occupancy-only, 256^3 grid, random SMPL
pose") on every rank: UNet7DS pose-feature map (the hand-written convolution kernel, csrc/conv_enc.hip) -> fused warp+template occupancy query over
ALL 256^3 grid points (dense: nothing is masked out) -> marching cubes + normals -> KNN-4 LBS ->
skinned live mesh, all on the device (avatarcap_amd.pipeline.FramePipeline.avatar_frame, i.e.
main.py:357-367,383-389).  Frames are independent, so ranks process different frames (weak scaling,
no data-path collective); for N > 1 the batch's meshes are all-gathered once over RCCL inside the
timed region (avatarcap_amd.parallel).  Inputs (weights, grid, per-frame pose maps / joint
matrices) are resident in HBM before the timed region.

The JSON line also carries:
  roofline     -- the dominant kernel (avatar_kernel): algorithmic FLOP/launch (1,773,568 per point,
                  SURVEY.md 8(d)) / mean launch time measured with HIP events on the launch stream,
                  against the dense fp16 MFMA peak (the kernel issues 3 fp16 MFMA passes per fp32
                  product -- 4,728 MFMAs per 32 points once the 64 pose-feature columns of conv1 / conv5 are folded per
                  grid column, DESIGN.md 2.5 -- so `mfma_util` = 2.73 x frac is the matrix-pipe utilisation; the launch
                  time includes the column pass that feeds the folded kernel; `traffic` is the
                  HBM byte count of the committed rocprofv3 --pmc passes of this launch, profiles/r05_pmc_avatar.md;
                  `clock_mhz` = s_memtime cycles of the timed launches / their device time = the clock the chip held in THIS run;
                  `sustained_mfma_tflops_measured` is the rate a pure MFMA + LDS-read loop of full-entropy operands holds on this
                  part at its power cap, profiles/r03_power_wall.md -- information, not `peak`).
  cpu_baseline -- the CPU restatements of oracle/ (the query on stock PyTorch CPU ops with identical weights on all host
                  cores, C marching cubes, NumPy LBS) timed on a bounded sample and scaled to one 256^3 frame.
  configs      -- BASELINE configs[2] (AvatarCap full: avatar + canonical normal fusion, 100 iterations + HGFilter + reconstruction query, the
                  reference's valid band at 256^3) and configs[3] (512^3 dense + colour head + marching cubes), a few frames each after the headline
                  measurement, with their stage split (N = 1 only; never `value`).
  masked       -- the same frame with the reference's own valid-band masking (only points within 0.1 m
                  of the body are evaluated, dataset/avatarcap_dataset.py:114-118), for information.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_POINT = 1_773_568          # warp 428,288 + shared 425,472 + geo 33,024 MAC
MFMA_ISSUED_PER_POINT = 4728 * 32 * 32 * 16 * 2 / 32   # 4728 v_mfma_f32_32x32x16_f16 per 32 points (3 split passes, tile padding, shared.6 folded
                                                        # into geo.0, the 64 feature columns of conv1 / conv5 folded per grid column: DESIGN.md section 2)
PEAK_F16_TFLOPS = 2500.0            # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md)
SUSTAINED_F16_TFLOPS = 1505.0      # what this part sustains at its power cap on split-fp16 MFMAs of FULL-ENTROPY operands fed from LDS, nothing else in
                                    # the instruction stream (1.47 GHz): tools/ubench/mfma_order.hip, profiles/r03_power_wall.md -- information only
                                    # (round 1's 1673 was measured on low-entropy operands)
# roofline.traffic -- HBM bytes per launch -- comes from PMC counters, which only a separate `rocprofv3 --pmc` pass can collect: tools/pmc_traffic.py
# writes them, with the hash of the kernel's sources, to profiles/pmc_traffic.json; this script carries the figure only while that hash is the tree's.


class _stdout_to_stderr:
    """Route the process's fd 1 to fd 2 for a while: RCCL prints a banner from C on its first use, the reference's module
    constructors print from Python -- and this script owes the driver exactly ONE line on stdout."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:                                    # RCCL's banner sits in the C library's stdout buffer: flush it while fd 1 still is stderr
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:                       # noqa: BLE001
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def build_pipeline(res, valid, n_frames, device):
    from avatarcap_amd import config, synthetic as syn
    from avatarcap_amd.dataset import SyntheticTestDataset
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar
    from avatarcap_amd.pipeline import FramePipeline
    config.device = device
    config.cfg = config.default_cfg()
    config.cfg['testing']['vol_res'] = [res, res, res]
    ds = SyntheticTestDataset([res, res, res], valid=valid, n_frames=n_frames, device=device)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):     # the reference's constructors announce themselves on stdout; this script prints ONE line there
        net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to(device).eval()
    sd = syn.synth_state_dict(syn.module_shapes(net), syn.SEED)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return FramePipeline(net, ds), sd


def _cpu_info():
    """(model name, physical cores, sockets, logical CPUs this process may use) of the host, from /proc/cpuinfo
    ('report core count, model name')."""
    model, cores, socks = 'unknown', set(), set()
    try:
        phys = core = None
        for line in open('/proc/cpuinfo'):
            k, _, v = line.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'model name':
                model = v
            elif k == 'physical id':
                phys = v; socks.add(v)
            elif k == 'core id':
                core = v
            elif not k and phys is not None:
                cores.add((phys, core)); phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    return model, (len(cores) or avail), (len(socks) or 1), avail


def cpu_baseline(pipe, sd, frame_in, frame_out, res, budget_s=12.0):
    """The CPU restatements of oracle/ on the host cores, beside the GPU number -- a reported baseline, not a target:
      * BASELINE configs[0] (64^3 grid) IN FULL, 5 repeats, median: U-Net on the frame's 256^2 position map, the query of all 262,144 grid points, marching
        cubes, Sobel normals + trilinear fetch at the vertices, KNN-4 LBS + skinning -- stock PyTorch CPU ops with identical weights (oracle/torch_cpu.py) and
        the C marching cubes (oracle/mc_oracle.c);
      * the dense 256^3 frame of `value`: the same pieces, the query on >= 4 chunks of 262,144 points scaled to the 64 chunks of the grid (a full CPU frame is
        minutes), marching cubes and normals on the full volume the GPU frame produced (checked elsewhere to be the oracle's), LBS on a vertex sample."""
    from oracle import avatarcap_oracle as orc, mc as omc, torch_cpu
    from avatarcap_amd import synthetic as syn
    from avatarcap_amd.grid import generate_volume_points_np
    ds = pipe.ds
    N = res ** 3
    model, phys, socks, avail = _cpu_info()
    tsd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items() if v.dtype == np.float32}
    tc = torch.from_numpy(np.asarray(ds.cano_smpl_center, np.float32))
    tv, tw = torch.from_numpy(ds.body['cano_smpl_v']), torch.from_numpy(ds.body['skin_weights'])
    pos_map = frame_in['smpl_pos_map'].detach().cpu().float()
    jm = frame_in['cano2live_jnt_mats'][0].detach().cpu().numpy()
    bounds = np.asarray(ds.cano_bounds, np.float32)

    def timed(fn):
        t = time.perf_counter(); r = fn(); return time.perf_counter() - t, r

    # -- thread count: more threads is not always faster here; pick the best of a sweep that includes the physical core count
    pts_all = ds.infer_pts.cpu()
    tf = torch_cpu.unet7ds(tsd, pos_map)[0]                                          # (64,256,256): the pose feature map, on the CPU
    n0 = 65536
    best = None
    for th in sorted({min(c, avail) for c in (8, 16, 32, 64, phys, avail)}):
        torch.set_num_threads(th)
        torch_cpu.occupancy_query(pts_all[:8192], tf, tc, tsd)                       # warm the thread pool / oneDNN
        dt_, _ = timed(lambda: torch_cpu.occupancy_query(pts_all[:n0], tf, tc, tsd))
        if best is None or dt_ < best[0]:
            best = (dt_, th)
        if dt_ > 2.0:
            break
    threads = best[1]
    torch.set_num_threads(threads)

    def mesh_stage(vol, r3, iso=0.0):
        """marching cubes (C, one thread) + normals (torch) + LBS + skinning of what comes out; -> seconds by piece, vertex count"""
        voxel = ((bounds[1] - bounds[0]) / np.asarray(r3, np.float32)).astype(np.float32)
        t_mc, (v, f) = timed(lambda: omc.marching_cubes(np.ascontiguousarray(vol.reshape(r3)), iso, voxel))
        V = v.shape[0]
        if V == 0:
            return {'mc': t_mc, 'normals': 0.0, 'lbs': 0.0}, 0
        verts = (v + bounds[0] + np.float32(0.5) * voxel).astype(np.float32)
        gridp = (2 * (verts - bounds[0]) / (bounds[1] - bounds[0]) - 1.0).astype(np.float32)
        t_n, _ = timed(lambda: torch_cpu.vertex_normals(torch.from_numpy(np.ascontiguousarray(vol.reshape(r3))), voxel, torch.from_numpy(gridp)))
        nv = min(V, 65536)
        vv = verts[:: max(1, V // nv)][:nv]
        t_l, _ = timed(lambda: orc.skinning(vv, torch_cpu.calculate_lbs(torch.from_numpy(vv), tv, tw).numpy(), jm))
        return {'mc': t_mc, 'normals': t_n, 'lbs': t_l * V / len(vv)}, V

    # -- (A) 64^3 in full, 5 repeats, median
    r64 = [64, 64, 64]
    p64 = torch.from_numpy(generate_volume_points_np(bounds, r64))
    runs = []
    for _ in range(5):
        t_u, fm = timed(lambda: torch_cpu.unet7ds(tsd, pos_map)[0])
        t_q, (occ, _) = timed(lambda: torch_cpu.occupancy_query(p64, fm, tc, tsd))
        st, V64 = mesh_stage(occ[:, 0].numpy(), r64)
        runs.append({'unet': t_u, 'query': t_q, **st})
        if sum(sum(r.values()) for r in runs) > 3 * budget_s:                        # a very slow host: fewer repeats, said so below
            break
    tot64 = sorted(sum(r.values()) for r in runs)
    med64 = tot64[len(tot64) // 2]
    rmed = sorted(runs, key=lambda r: sum(r.values()))[len(runs) // 2]
    # -- (B) the dense frame of `value`
    per0 = best[0] / n0
    n1 = int(min(N, max(4 * 262144, (budget_s / max(per0, 1e-9)) // 262144 * 262144)))
    t_q, _ = timed(lambda: torch_cpu.occupancy_query(pts_all[:: N // n1][:n1], tf, tc, tsd))
    per_pt = t_q / n1
    t_u, _ = timed(lambda: torch_cpu.unet7ds(tsd, pos_map))
    st, V = mesh_stage(frame_out['occ_volume'].reshape(res, res, res).cpu().numpy(), [res] * 3)
    frame_s = t_u + per_pt * N + st['mc'] + st['normals'] + st['lbs']
    m0 = 16384
    t_np, _ = timed(lambda: orc.occupancy_query(pts_all[:m0].numpy(), tf.numpy(), ds.cano_smpl_center, sd, dt=np.float32))
    return {'value': 1.0 / frame_s, 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
            'cpu_model': model, 'physical_cores': phys, 'sockets': socks, 'logical_cpus_available': avail,
            'sample': f'stock PyTorch-CPU restatement (oracle/torch_cpu.py) on {threads} threads of {phys} physical cores ({model}): U-Net in full '
                      f'({t_u:.2f} s); query '
                      f'{n1} of {N} grid points = {n1 // 262144} chunks of 262,144 ({t_q:.1f} s, {per_pt*1e6:.2f} us/pt) scaled to {N}; C marching cubes on '
                      f'the full '
                      f'{res}^3 volume ({st["mc"]:.2f} s, 1 thread); Sobel normals (torch conv3d) + fetch at the {V} vertices ({st["normals"]:.2f} s); '
                      f'torch-CPU KNN-4 '
                      f'LBS + skinning on a 65,536-vertex sample scaled to {V} ({st["lbs"]:.2f} s)',
            'seconds_per_frame': frame_s,
            'seconds_per_frame_by_piece': {'unet7ds': t_u, 'query (scaled)': per_pt * N, 'marching cubes': st['mc'], 'normals': st['normals'],
                                           'lbs + skinning (scaled)': st['lbs']},
            'config0_64cube_full': {'workload': 'BASELINE configs[0]: 64^3 grid (262,144 points), one frame in full on the CPU: U-Net, query, marching '
                                                'cubes, normals, LBS',
                                    'repeats': len(runs), 'median_seconds_per_frame': med64, 'frames_per_s': 1.0 / med64, 'all_seconds': tot64,
                                    'median_run_by_piece': rmed, 'vertices': V64},
            'numpy_port_us_per_point': t_np / m0 * 1e6, 'torch_cpu_us_per_point': per_pt * 1e6}


def _free_port():
    from avatarcap_amd.parallel import free_port
    return free_port()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N ranks of this script, one per GPU, under
    torch.distributed.run on 127.0.0.1 and pass rank 0's JSON line through.  (Round 1 silently measured ONE GPU in this case.)"""
    from avatarcap_amd.parallel import self_launch as launch
    return launch(__file__, sys.argv[1:], args.gpus)


def dry_run(args, world, rank):
    """No GPU: the launcher, the rendezvous and the mesh all-gather on gloo with stand-in meshes.  Prints the same line shape with
    n_gpus = the ranks the process group really has (tests/test_parallel_gloo.py)."""
    import torch.distributed as dist
    from avatarcap_amd.parallel import all_gather_meshes, shard_frames
    if world > 1:
        from avatarcap_amd.parallel import init_process_group
        with _stdout_to_stderr():                        # gloo announces its peers on stdout
            init_process_group('gloo', rank, world, timeout_s=args.dist_timeout)
    K = args.frames // world if args.frames else args.steps
    n_frames = world * K
    g = torch.Generator().manual_seed(1234)
    sizes = torch.randint(5, 40, (n_frames,), generator=g).tolist()
    mesh = lambda f: {'v': torch.full((sizes[f], 3), float(f)), 'vn': torch.full((sizes[f], 3), -float(f)),    # noqa: E731
                      'f': torch.full((2 * sizes[f], 3), f, dtype=torch.int32)}
    t0 = time.perf_counter()
    with _stdout_to_stderr():
        from avatarcap_amd.parallel import MeshExchange, verify_gathered_meshes
        ex = MeshExchange(n_frames)                      # driven as the timed loop drives it: pump() inside "the next frame", submit() behind it
        for k in range(ex.steps):
            ex.pump()
            ex.submit(mesh(k * world + rank))
        got = ex.finish()
    dt = time.perf_counter() - t0
    with _stdout_to_stderr():
        complaints = verify_gathered_meshes(got, {f: mesh(f) for f in shard_frames(n_frames, rank, world)})
    for c in complaints:
        print('# bench.py --dry-run: ' + c, file=sys.stderr, flush=True)
    ok = not complaints and len(got) == n_frames and all(
        got[f]['v'].shape[0] == sizes[f] and float(got[f]['v'][0, 0]) == float(f) and int(got[f]['f'][0, 0]) == f for f in range(n_frames))
    if world > 1:
        flag = torch.tensor([0 if ok else 1])
        with _stdout_to_stderr():
            dist.all_reduce(flag)
        ok = int(flag) == 0
    ranks = dist.get_world_size() if dist.is_initialized() else 1
    if rank == 0:
        print(json.dumps({'metric': 'dry run (no GPU): launcher + gloo all-gather of stand-in meshes', 'value': n_frames / max(dt, 1e-9),
                          'unit': 'frames/s', 'n_gpus': ranks, 'rccl_ranks': 0, 'gloo_ranks': ranks, 'steps': K, 'warmup': 0,
                          'frames': n_frames, 'all_gather_ok': bool(ok), 'meshes_verified': bool(ok), 'data': 'synthetic'}), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--frames', type=int, default=0, help='total frames of the batch over all ranks (BASELINE configs[4]: 64); sets steps = frames / gpus')
    ap.add_argument('--dry-run', action='store_true', help='no GPU: launcher + gloo all-gather of stand-in meshes')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-masked', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='skip the BASELINE configs[2] / configs[3] legs of the line')
    ap.add_argument('--dist-timeout', type=float, default=180.0, help='seconds after which a rendezvous / collective that does not complete ends the job')
    ap.add_argument('--spare-cus', type=int, default=None, help='N > 1: CUs the persistent query kernels leave to the exchange (0: none; AVC_MLP_BLOCKS wins); '
                                                                'default: chosen by the warm-up A/B (8, 4 or 0)')
    ap.add_argument('--no-autotune', action='store_true', help='N > 1: no warm-up A/B of the exchange transport / spare CUs (p2p, 8 unless set otherwise)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    if args.frames:
        if args.frames % world:
            raise SystemExit(f'bench.py: --frames {args.frames} is not a multiple of --gpus {world}')
        args.steps = args.frames // world
    if args.dry_run:
        raise SystemExit(dry_run(args, world, rank))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: no HIP device visible (there is no CPU fallback for the hot path)')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    import torch.distributed as dist
    force_dist = os.environ.get('AVC_FORCE_DIST') == '1'    # exercise the RCCL all-gather with a single rank (tests/test_gpu_pipeline.py)
    from avatarcap_amd.parallel import init_process_group, barrier_or_die
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(_free_port()))
        with _stdout_to_stderr():
            init_process_group('nccl', rank, world, device, timeout_s=args.dist_timeout)       # bounded: a missing rank ends the job with a message
    rccl_ranks = dist.get_world_size() if dist.is_initialized() else 0
    if world > 1 and rccl_ranks != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the RCCL process group has {rccl_ranks} ranks')

    from avatarcap_amd import _lib
    from avatarcap_amd.dataset import to_cuda
    from avatarcap_amd.parallel import MeshExchange, all_gather_meshes, pin_to_gpu_numa
    query_wgs = 0
    if world > 1:
        pin_to_gpu_numa(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)))       # each rank on the cores of its GPU's NUMA node
        from avatarcap_amd.parallel import leave_cus_for_the_exchange
        # RCCL's copy kernels need somewhere to run beside the query
        query_wgs = leave_cus_for_the_exchange(device, 8 if args.spare_cus is None else args.spare_cus)
    K, W, res = args.steps, args.warmup, args.res
    n_frames = world * (K + W)
    pipe, sd = build_pipeline(res, 'dense', n_frames, device)
    ctx = _lib.ctx(device)
    # inputs of this rank's frames, resident in HBM before timing
    my = [to_cuda(pipe.ds[(s * world) + rank], add_batch=True) for s in range(K + W)]

    def barrier(what='timed region'):
        torch.cuda.synchronize()
        if world > 1:
            barrier_or_die(what, rank)
        torch.cuda.synchronize()

    # inside a batch the pipeline is told the next frame, so that its U-Net launches are queued behind the running query (pipeline.py);
    # not across the border of the timed region: its K frames contain exactly K U-Net passes
    out = None
    for s in range(W):
        out = pipe.avatar_frame(my[s], next_items=my[s + 1] if s + 1 < W else None)
    if world > 1 or force_dist:   # warm the exchange too (RCCL sets its point-to-point connections up on first use)
        # (--warmup 0: a token mesh)
        warm = ({'v': out['live_v'], 'vn': out['live_vn'], 'f': out['f']} if out is not None and out.get('live_v') is not None else
                {'v': torch.zeros((3, 3), device=device), 'vn': torch.zeros((3, 3), device=device), 'f': torch.zeros((1, 3), dtype=torch.int32, device=device)})
        with _stdout_to_stderr():
            all_gather_meshes([warm], world, force=force_dist)
            torch.cuda.synchronize()
    # ---- N > 1: the first run on a multi-GPU box tunes itself.  Nobody has run this path on more than one GPU (no such box was available to the build), so
    # the transport of the mesh exchange (point-to-point sends, one xGMI link per pair, vs `world` broadcasts) and the CUs the persistent query leaves to
    # RCCL's copy kernels (8, 4, 0) are not guesses but a warm-up A/B: two frames + their exchange under each setting, MAX over the ranks, the earliest
    # setting within 2 % of the fastest (parallel.choose_exchange_config).  Untimed; what was chosen and the whole table go on the line.
    exchange_mode = os.environ.get('AVC_EXCHANGE', 'p2p')
    autotune = None
    if (world > 1 or force_dist) and not args.no_autotune:
        from avatarcap_amd.parallel import choose_exchange_config, set_spare_cus
        modes = [os.environ['AVC_EXCHANGE']] if os.environ.get('AVC_EXCHANGE') else ['p2p', 'broadcast']
        spares = [args.spare_cus] if (args.spare_cus is not None or os.environ.get('AVC_MLP_BLOCKS')) else [8, 4, 0]
        cands = [(m, sp) for m in modes for sp in spares]

        def measure(c, steps=2):
            mode, spare = c
            if spare is not None and not os.environ.get('AVC_MLP_BLOCKS'):
                set_spare_cus(device, spare)
            ex_ = MeshExchange(world * (steps + 1), force=force_dist, mode=mode)
            pipe.exchange = ex_
            t_ = 0.0
            for s_ in range(steps + 1):                      # one step to settle (the transport's connections, the allocator), then `steps` timed
                if s_ == 1:
                    barrier('warm-up A/B')
                    t_ = time.perf_counter()
                o_ = pipe.avatar_frame(my[s_ % len(my)])
                ex_.submit({'v': o_['live_v'], 'vn': o_['live_vn'], 'f': o_['f']})
            pipe.exchange = None
            ex_.finish()
            torch.cuda.synchronize()
            return time.perf_counter() - t_

        if len(cands) > 1:
            with _stdout_to_stderr():
                autotune = choose_exchange_config(cands, measure, device=device)
            exchange_mode, spare = autotune['choice']
            autotune = {'candidates': [list(c) for c in cands], 'warmup_ab_ms': autotune['table_ms'], 'choice': [exchange_mode, spare],
                        'rule': 'two frames + exchange per setting after one settling step, MAX over ranks, earliest setting within 2 % of the fastest'}
            if not os.environ.get('AVC_MLP_BLOCKS'):
                query_wgs = set_spare_cus(device, spare)
        pipe._next_map = None
    barrier('start of the timed region')
    _lib.check(_lib.lib().avc_timing_enable(ctx, 1))
    t0 = time.perf_counter()
    # the batch's meshes are exchanged step by step WHILE the following frames compute (parallel.MeshExchange: exact sizes, asynchronous
    # point-to-point sends on RCCL's stream); what is left behind the last frame is that frame's own mesh
    ex = MeshExchange(world * K, force=force_dist, mode=exchange_mode) if (world > 1 or force_dist) else None
    pipe.exchange = ex                              # avatar_frame pumps it behind its query launch: step s - 1 travels while frame s computes
    for s in range(W, W + K):
        out = pipe.avatar_frame(my[s], next_items=my[s + 1] if s + 1 < W + K else None)
        if ex is not None:
            ex.submit({'v': out['live_v'], 'vn': out['live_vn'], 'f': out['f']})
    pipe.exchange = None
    torch.cuda.synchronize()
    t_frames = time.perf_counter() - t0            # this rank's K frames (the exchange of the earlier steps ran beside them)
    t_gather, rx_bytes, gathered = 0.0, 0, None
    if ex is not None:
        tg = time.perf_counter()
        gathered = ex.finish()
        torch.cuda.synchronize()
        t_gather = time.perf_counter() - tg        # the exchange's tail: the LAST step's meshes (+ waiting for the slowest rank's last frame)
        rx_bytes = ex.bytes_received
    barrier('end of the timed region')
    dt = time.perf_counter() - t0
    _lib.check(_lib.lib().avc_timing_enable(ctx, 0))
    # ---- outside the timed region: the exchange validates itself.  Every owner's integer checksums (V, F, sum of the bit patterns of [v | vn], sum of the
    # face indices, position-weighted sum) of the meshes it produced are all-gathered and compared, on every rank, with what arrived in each frame
    # slot (parallel.verify_gathered_meshes); the owner's last produced mesh is compared tensor for tensor with its slot (the packing).
    meshes_verified, complaints = None, []
    if ex is not None:
        from avatarcap_amd.parallel import verify_gathered_meshes
        with _stdout_to_stderr():
            if len(gathered) != world * K:
                complaints.append(f'rank {rank}: {len(gathered)} meshes gathered, {world * K} expected')
            else:
                complaints += verify_gathered_meshes(gathered, {s_ * world + rank: gathered[s_ * world + rank] for s_ in range(K)}, force=force_dist)
                last = gathered[(K - 1) * world + rank]
                if not (torch.equal(last['v'], out['live_v']) and torch.equal(last['vn'], out['live_vn']) and torch.equal(last['f'], out['f'])):
                    complaints.append(f'rank {rank}: the slot of its last frame does not hold the mesh it produced')
            bad = torch.tensor([len(complaints)], dtype=torch.int64, device=device)
            if world > 1:
                dist.all_reduce(bad)
        meshes_verified = int(bad.item()) == 0
        for c in complaints:
            print('# bench.py: MESH EXCHANGE CHECK FAILED -- ' + c, file=sys.stderr, flush=True)
    pumped = ex.pumped_early if ex is not None else 0
    del gathered, ex
    per_rank = [[dt, t_frames, t_gather, float(rx_bytes), float(pumped)]]
    if world > 1:
        mine_t = torch.tensor(per_rank[0], dtype=torch.float64, device=device)
        all_t = torch.empty(world * 5, dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(all_t, mine_t)
        per_rank = all_t.reshape(world, 5).cpu().tolist()
        dt = max(r[0] for r in per_rank)           # MAX over ranks

    import ctypes as C
    avg_ms, launches, avg_cyc, ncyc = C.c_double(), C.c_int64(), C.c_double(), C.c_int64()
    _lib.check(_lib.lib().avc_timing_read(ctx, 0, C.byref(avg_ms), C.byref(launches), 1))
    _lib.check(_lib.lib().avc_timing_read_cycles(ctx, 0, C.byref(avg_cyc), C.byref(ncyc)))     # s_memtime of the same launches -> the clock the chip held
    N = res ** 3
    achieved = N * FLOP_PER_POINT / (avg_ms.value * 1e-3) / 1e12 if avg_ms.value > 0 else 0.0

    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import pmc_traffic
        traffic_bytes, traffic_ref = pmc_traffic.load(ROOT)
        n_cus = torch.cuda.get_device_properties(device).multi_processor_count
        line = {
            'metric': 'reconstructed-mesh frames/sec at 256^3 grid (avatar occupancy-only, dense query + marching cubes + LBS)',
            'value': world * K / dt, 'unit': 'frames/s', 'n_gpus': world, 'rccl_ranks': rccl_ranks, 'steps': K, 'warmup': W,
            'ms_per_step': dt / K * 1e3,
            'ms_per_step_per_rank': [r[1] / K * 1e3 for r in per_rank], 'exchange_tail_ms_per_rank': [r[2] * 1e3 for r in per_rank],
            'exchange_tail_ms': max(r[2] for r in per_rank) * 1e3, 'all_gather_rx_mb_per_rank': [r[3] / 1e6 for r in per_rank],
            'exchange_steps_sent_inside_the_next_frame_per_rank': [int(r[4]) for r in per_rank], 'meshes_verified': meshes_verified,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (products as 3 split-fp16 MFMA passes, fp32 accumulate)', 'data': 'synthetic',
            'config': {'workload': f'BASELINE configs[1]: AvatarNet occupancy-only, {res}^3 grid dense ({N} points/frame), random SMPL pose, '
                                   f'UNet7DS + fused query + marching cubes + normals + KNN-4 LBS per frame',
                       'grid': [res] * 3, 'points_per_frame': N, 'vertices_last_frame': int(out['cano_v'].shape[0]),
                       'faces_last_frame': int(out['f'].shape[0]), 'parallelism': f'frame-sharded x{world}',
                       'frames_in_batch': world * K, 'meshes_all_gathered': bool(world > 1 or force_dist),
                       'query_workgroups': query_wgs or 'one per CU', 'exchange_transport': exchange_mode if (world > 1 or force_dist) else None,
                       'exchange_autotune': autotune,
                       'mesh_exchange': 'exact-size point-to-point sends per step (every pair of GPUs over its own xGMI link), issued from a side stream '
                                        'behind the NEXT frame\'s query launch (parallel.MeshExchange.pump): '
                                        'K - 1 of a rank\'s K steps travel beside compute, the last one is `exchange_tail_ms`; `meshes_verified`: per-frame '
                                        'integer checksums '
                                        'of every received mesh against its owner\'s, checked on every rank outside the timed region',
                       'semantics': '`value` is the DENSE stress variant BASELINE configs[1] names (every one of the 256^3 grid points evaluated); the '
                                    'reference itself '
                                    'evaluates only the valid band around the canonical SMPL and fills the rest (main.py:362-363): that is `masked` and the '
                                    '`configs` legs below',
                       'dense_points': 'generated from the grid index (avc_avatar_query_grid), offsets not written',
                       # third-party arithmetic whose pinned version could not be had offline: parity is pinned on what IS here
                       'unpinned': ['scikit-image 0.17.2 marching_cubes (pinned on 0.18.3)', 'pytorch3d 0.6.0 knn_points tie order',
                                    'trimesh 3.9.15 contains', 'opencv resize / Rodrigues']},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F16_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_F16_TFLOPS,
                         # HBM-side bytes per launch of the dense 256^3 query + its column pass: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 of separate
                         # rocprofv3 --pmc passes over this very launch (the guide's gfx950 correction).  Counters cannot be read from inside this
                         # process: the figure is that
                         # pass's -- null for any other grid, and null once the kernel's sources differ from the ones it was measured on (`traffic_ref.state`)
                         'traffic': traffic_bytes if res == 256 else None, 'traffic_unit': 'bytes per launch', 'traffic_ref': traffic_ref,
                         'traffic_note': 'separate rocprofv3 --pmc passes of the same launch (tools/pmc_traffic.py -> profiles/pmc_traffic.json, carried '
                                         'only while the kernel '
                                         'sources hash to what they were when the counters were read): ~0.52 GB against 0.087 GB algorithmic -- the 131 MB '
                                         'column table written '
                                         'and read back, the feature map once per XCD; 0.1 % of the HBM bandwidth (the kernel is MFMA / power bound)',
                         'kernel': 'avc::avatar_kernel<true,false,1> (+ its column_terms_kernel pass, timed together)',
                         'avg_launch_ms': avg_ms.value, 'launches': launches.value,
                         'shader_cycles_per_launch': avg_cyc.value, 'clock_mhz': (avg_cyc.value / (avg_ms.value * 1e3)) if avg_ms.value > 0 else 0.0,
                         'cycles_per_mfma': avg_cyc.value / (4728 * -(-(N // 128) // min(N // 128, query_wgs or n_cus))) if avg_cyc.value > 0 else 0.0,
                         'algorithmic_flop_per_launch': N * FLOP_PER_POINT,
                         'mfma_issued_tflops': N * MFMA_ISSUED_PER_POINT / (avg_ms.value * 1e-3) / 1e12 if avg_ms.value > 0 else 0.0,
                         'mfma_util': (N * MFMA_ISSUED_PER_POINT / (avg_ms.value * 1e-3) / 1e12 / PEAK_F16_TFLOPS) if avg_ms.value > 0 else 0.0,
                         'fp32_mfma_peak_equiv': achieved / 157.3,
                         'sustained_mfma_tflops_measured': SUSTAINED_F16_TFLOPS,
                         'mfma_issued_vs_sustained': (N * MFMA_ISSUED_PER_POINT / (avg_ms.value * 1e-3) / 1e12 / SUSTAINED_F16_TFLOPS)
                                                     if avg_ms.value > 0 else 0.0},
        }
        if world == 1:
            # the HBM-bound kernels of the same frame, timed on its own volume and mesh (algorithmic bytes of SURVEY.md 8(d) / time / 8 TB/s)
            try:
                from avatarcap_amd import config as cfg_
                from avatarcap_amd.utils import recon_util
                from avatarcap_amd.utils.smpl_util import smpl_util
                vol, V, Fc = out['occ_volume'], int(out['cano_v'].shape[0]), int(out['f'].shape[0])

                def timed(fn, reps=5):
                    fn(); torch.cuda.synchronize()
                    t = time.perf_counter()
                    for _ in range(reps):
                        fn()
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t) / reps

                t_mc = timed(lambda: recon_util.recon_mesh_device(vol, pipe.vol_res, pipe.ds.cano_bounds, iso_value=cfg_.iso_value))
                v1, n1, jm = out['cano_v'][None], out['cano_vn'][None], my[-1]['cano2live_jnt_mats']

                def lbs_all():
                    smpl_util.lbs_skinning(v1, n1, jm, return_pt_mats=True)      # what the frame runs (pipeline.avatar_frame)
                t_lbs = timed(lbs_all)
                # (the fused launch keeps a vertex's 24 blend weights in registers: points + normals in, points + normals + 4x4 out, the skin-weight table)
                mc_bytes, lbs_bytes = 4 * N + 24 * V + 12 * Fc, V * (12 + 12 + 12 + 12 + 64) + 83_000 + 24 * 4 * 6890
                def hbm(t, nbytes, **more):
                    return {'bound': 'hbm', 'ms': t * 1e3, 'algorithmic_bytes': nbytes, 'achieved': nbytes / t / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                            'frac': nbytes / t / 8e12, **more}
                line['roofline_secondary'] = {
                    'marching cubes + normals (mesh.hip, 6 launches, one host wait at the end)': hbm(t_mc, mc_bytes),
                    'KNN-4 LBS + skinning of points and normals, one launch (knn_lbs.hip: lbs_skin_grid_kernel)': hbm(
                        t_lbs, lbs_bytes, note='the search is VALU-bound (DESIGN.md section 3): the HBM fraction is reported for completeness')}
            except Exception as e:       # informational only
                line['roofline_secondary'] = {'error': repr(e)}
            # ---- outside the timed region: the frame reproduces itself.  The timed loop runs the next frame's U-Net on a side stream beside this frame's
            # marching cubes / LBS; round 6 found kernels that were bitwise deterministic alone and not beside it (profiles/r06_store_hazard.md).  One
            # frame with the look-ahead running beside its tail against the same frame alone, every mesh tensor bit for bit.
            try:
                if len(my) >= 2:
                    pipe._next_map = None
                    busy = pipe.avatar_frame(my[0], next_items=my[1])
                    torch.cuda.synchronize()
                    pipe._next_map = None
                    quiet = pipe.avatar_frame(my[0])
                    torch.cuda.synchronize()
                    line['frame_reproduced'] = all(torch.equal(busy[k_], quiet[k_]) for k_ in ('cano_v', 'cano_vn', 'f', 'live_v', 'live_vn', 'vert_mats'))
            except Exception as e:       # informational only
                line['frame_reproduced'] = repr(e)
        if world == 1 and not args.no_masked:
            try:
                line['masked'] = masked_run(res, device, K, W)
            except Exception as e:       # informational leg only
                line['masked'] = {'error': repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line['cpu_baseline'] = cpu_baseline(pipe, sd, my[-1], out, res)
            except Exception as e:       # a host without the oracle's build must not cost the run its GPU measurement
                line['cpu_baseline'] = {'error': repr(e)}
        if world == 1 and not args.no_configs:
            del pipe, my
            torch.cuda.empty_cache()
            try:
                line['configs'] = other_configs(device)
            except Exception as e:       # informational legs only
                line['configs'] = {'error': repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()
    if meshes_verified is False:
        raise SystemExit('bench.py: the gathered meshes do not match what their owners produced (see stderr): the line above is not a valid measurement')


def other_configs(device, frames=3):
    """BASELINE configs[2], the reference's own example.yaml grid and configs[3] on the driver-timed line, N = 1, a few frames each; never `value`."""
    from avatarcap_amd import config, synthetic as syn, _lib
    from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda, synthetic_camera, synthetic_observed_normals
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar
    from avatarcap_amd.network.arch_recon import ReconNetwork
    from avatarcap_amd.pipeline import FramePipeline
    import contextlib
    import ctypes as C
    out = {}
    ctx = _lib.ctx(device)

    def stage_ms(fn, reps):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3, r

    with contextlib.redirect_stdout(sys.stderr):
        net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to(device).eval(); syn.load_synth(net, syn.SEED)
        rn = ReconNetwork().to(device).eval(); syn.load_synth(rn, syn.SEED)
    # ---- configs[2]: AvatarCap full on the reference's valid band -- at 256^3 and at the reference's own vol_res (configs/example.yaml:14-17)
    def full_leg(res, what):
        config.cfg['testing']['vol_res'] = list(res)
        ds = SyntheticTestDataset(list(res), valid='band', n_frames=frames + 1, device=device)
        pipe = FramePipeline(net, ds, rn)
        items = [to_cuda(ds[i], add_batch=True) for i in range(frames + 1)]
        w2c, cam = synthetic_camera()
        a = pipe.avatar_frame(items[0])
        obs = synthetic_observed_normals(a['live_v'], a['live_vn'], a['f'], w2c, cam, seed=0)       # the image-observed normal map of step 2, synthesised once
        pipe.avatarcap_frame(items[0], obs, w2c, cam); torch.cuda.synchronize()
        _lib.check(_lib.lib().avc_timing_enable(ctx, 1))
        t = time.perf_counter()
        for i in range(1, frames + 1):
            a, r = pipe.avatarcap_frame(items[i], obs, w2c, cam, next_items=items[i + 1] if i < frames else None)
        torch.cuda.synchronize()
        full_ms = (time.perf_counter() - t) / frames * 1e3
        q = [(C.c_double(), C.c_int64()) for _ in range(2)]
        for w in range(2):
            _lib.check(_lib.lib().avc_timing_read(ctx, w, C.byref(q[w][0]), C.byref(q[w][1]), 1))
        _lib.check(_lib.lib().avc_timing_enable(ctx, 0))
        it = dict(items[1])
        t_av, a = stage_ms(lambda: pipe.avatar_frame(items[1]), 3)
        t_fu, fm = stage_ms(lambda: pipe.fuse_normals(a, obs, w2c, cam), 3)
        it['front_normal'], it['back_normal'] = fm[0], fm[1]
        t_re, r = stage_ms(lambda: pipe.recon_frame(it), 3)
        imgs = torch.cat([it['front_normal'], it['back_normal']], dim=1)
        t_hg, _ = stage_ms(lambda: rn.bind_feat_map(imgs), 3)          # the encoder as recon_frame runs it: its channel-last output bound as the decoder's map
        t_un, _ = stage_ms(lambda: net.warping_field.unet(items[1]['smpl_pos_map']), 3)
        leg = {'workload': 'AvatarCap full (main.py:357-453): avatar query + marching cubes + LBS, canonical normal fusion (100 iterations), HGFilter '
                           '(hand-written HIP encoder), reconstruction query + marching cubes + LBS; ' + what, 'vol_res': list(res), 'frames': frames,
               'valid_points': int(ds.infer_pts.shape[0]), 'valid_fraction': float(ds.infer_pts.shape[0]) / float(np.prod(res)),
               'ms_per_frame': full_ms, 'frames_per_s': 1e3 / full_ms,
               'stage_ms': {'avatar_frame (unet + band query + mc + lbs)': t_av, 'normal maps + fusion': t_fu,
                            'recon_frame (hgfilter + band query + mc + lbs)': t_re, 'of which hgfilter': t_hg, 'of which unet7ds (in avatar_frame)': t_un},
               'kernel_ms': {'avatar query (band, column-folded)': q[0][0].value, 'recon query (band, column-folded)': q[1][0].value},
               'avatar_vertices': int(a['cano_v'].shape[0]), 'recon_vertices': int(r['cano_v'].shape[0])}
        del ds, pipe, items, a, r, obs, fm, it, imgs
        torch.cuda.empty_cache()
        return leg

    out['configs[2]'] = full_leg([256] * 3, "256^3 grid, the reference's valid band")
    # secondary roofline fractions of the other kernels of the path, from the legs' own timings (algorithmic work of SURVEY.md 8(d) / time / peak)
    c2 = out['configs[2]']
    nb = c2['valid_points']
    def mfma(flop, ms):
        return {'bound': 'mfma', 'achieved_tflops': flop / (ms * 1e-3) / 1e12}
    km, sm = c2['kernel_ms'], c2['stage_ms']
    out['secondary_rooflines'] = {
        'avatar band query (avatar_kernel, column-folded band)': mfma(nb * FLOP_PER_POINT, km['avatar query (band, column-folded)']),
        'recon band query (recon_fold_kernel<2>, column-folded band)': mfma(nb * 387072, km['recon query (band, column-folded)']),
        'HGFilter encoder (conv_enc.hip, ~70 launches incl. their gaps)': mfma(232.3e9, sm['of which hgfilter']),
        'UNet7DS (conv_enc.hip, 18 launches; weight-stream bound at its deep levels)': mfma(10.35e9, sm['of which unet7ds (in avatar_frame)']),
    }
    for v in out['secondary_rooflines'].values():
        v['peak_tflops'] = PEAK_F16_TFLOPS
        v['frac'] = v['achieved_tflops'] / PEAK_F16_TFLOPS
    out['example.yaml'] = full_leg([384, 384, 128], "the reference's own configuration: vol_res 384 x 384 x 128 (configs/example.yaml:14-17), valid band "
                                                    "(dataset/avatarcap_dataset.py:111-125) -- what `main.py -c configs/example.yaml -m test` computes per "
                                                    "frame")
    # ---- configs[3]: 512^3 dense + marching cubes + colour head on the vertices (HBM-bound stress)
    config.cfg['testing']['vol_res'] = [512] * 3
    ds = SyntheticTestDataset([512] * 3, valid='dense', n_frames=2, device=device)
    pipe = FramePipeline(net, ds, rn)
    items = [to_cuda(ds[i], add_batch=True) for i in range(2)]
    _lib.check(_lib.lib().avc_timing_enable(ctx, 1))
    t5, a5 = stage_ms(lambda: pipe.avatar_frame(items[1]), 2)
    qa, ql = C.c_double(), C.c_int64()
    _lib.check(_lib.lib().avc_timing_read(ctx, 0, C.byref(qa), C.byref(ql), 1))
    _lib.check(_lib.lib().avc_timing_enable(ctx, 0))
    nv = min(200_000, int(a5['cano_v'].shape[0]))
    v, n = a5['cano_v'][:nv].contiguous(), a5['cano_vn'][:nv].contiguous()
    tc, rgb = stage_ms(lambda: pipe.colour_vertices(items[1], v, n), 2)
    out['configs[3]'] = {'workload': '512^3 dense grid (134,217,728 points) + marching cubes + normals + LBS, colour head on 200 k vertices (64 samples per '
                                     'ray)',
                         'avatar_frame_ms': t5, 'frames_per_s': 1e3 / t5, 'query_kernel_ms': qa.value, 'vertices': int(a5['cano_v'].shape[0]),
                         'faces': int(a5['f'].shape[0]), 'colour_vertices': nv, 'colour_ms': tc, 'rgb_finite': bool(torch.isfinite(rgb).all())}
    del ds, pipe, items, a5, v, n, rgb
    torch.cuda.empty_cache()
    out['main_py_e2e'] = main_py_e2e(out['example.yaml']['ms_per_frame'])
    return out


def main_py_e2e(device_ms, frames=32):
    """The entry point itself, end to end: `python main.py -c configs/example.yaml -m test --synthetic --frames 32` as a
    process of its own, files written to a scratch directory, timed by the loop itself (--timing-json: steady frames from the third on, until the
    last file is closed).
    Three output shapes: none; the reference's mesh output (live avatar + live reconstruction as PLY, main.py:491-498); PLY + every mesh tensor as .npz."""
    import shutil
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix='avc_e2e_')
    legs = {}
    try:
        for tag, flags in (('no outputs', ['--no-npz']), ('ply (the reference\'s mesh output)', ['--no-npz', '--save-ply']), ('ply + npz', ['--save-ply'])):
            tj = os.path.join(tmp, 'timing.json')
            cmd = [sys.executable, os.path.join(ROOT, 'main.py'), '-c', os.path.join(ROOT, 'configs', 'example.yaml'), '-m', 'test', '--synthetic',
                   '--frames', str(frames), '--output-dir', os.path.join(tmp, 'out'), '--timing-json', tj] + flags
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
            if r.returncode != 0 or not os.path.exists(tj):
                legs[tag] = {'error': (r.stderr or r.stdout)[-400:]}
                continue
            t = json.load(open(tj))
            legs[tag] = {'ms_per_frame': t['e2e_ms_per_frame'], 'frames_per_s': 1e3 / t['e2e_ms_per_frame'],
                         'vs_device_figure': t['e2e_ms_per_frame'] / device_ms,
                         'device_done_ms_per_frame': t['device_ms_per_frame'], 'mb_written_per_frame': t['bytes_written'] / 1e6 / t['frames'],
                         'writer_tail_ms': t['writer_tail_ms'], 'waited_for_writer_slot_ms': t['waited_for_writer_slot_ms'],
                         'h2d_copies_per_frame': t['h2d_copies'] / t['frames'], 'first_frame_ms': t['first_frame_ms']}
            shutil.rmtree(os.path.join(tmp, 'out'), ignore_errors=True)
            os.remove(tj)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {'command': f'python main.py -c configs/example.yaml -m test --synthetic --frames {frames} [--no-npz] [--save-ply]', 'frames': frames,
            'device_figure_ms': device_ms, 'timed': 'by the loop itself (main.py --timing-json): frames 3..N, until the last file is closed',
            'scratch': 'tempfile.mkdtemp() of the box', 'legs': legs}


def masked_run(res, device, K, W):
    """Same frame with the reference's valid-band masking (informational)."""
    from avatarcap_amd.dataset import to_cuda
    pipe, _ = build_pipeline(res, 'band', K + W, device)
    items = [to_cuda(pipe.ds[s], add_batch=True) for s in range(K + W)]
    for s in range(W):
        out = pipe.avatar_frame(items[s], next_items=items[s + 1] if s + 1 < W else None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(W, W + K):
        out = pipe.avatar_frame(items[s], next_items=items[s + 1] if s + 1 < W + K else None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {'value': K / dt, 'unit': 'frames/s', 'valid_points': int(pipe.ds.infer_pts.shape[0]),
            'valid_fraction': float(pipe.ds.infer_pts.shape[0]) / res ** 3, 'vertices_last_frame': int(out['cano_v'].shape[0])}


if __name__ == '__main__':
    main()
