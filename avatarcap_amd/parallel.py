"""Frame-sharded data parallelism.

The reference has no multi-GPU code; its frame loop (main.py:348) carries no state between frames, so
frames shard embarrassingly: frame f runs on rank f mod world_size, one process per GPU, weights /
grid / SMPL tables replicated.  The only exchange is an all-gather(v) of the finished meshes
(torch.distributed; backend 'nccl' = RCCL over xGMI on the GPU box, 'gloo' in the CPU tests):
  1. all_gather of the per-frame (V, F) counts,
  2. one all_gather of the [verts | normals] float buffers and one of the int32 faces, padded to the
     largest rank (xGMI moves the ~1 GB of a 64-frame batch in well under one frame time, so a single
     large collective per batch beats per-frame ones).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_frames(n_frames: int, rank: int, world_size: int) -> list[int]:
    """Frames owned by `rank`: f with f % world_size == rank, ascending."""
    return list(range(rank, n_frames, world_size))


def shard_range(n: int, rank: int, world_size: int) -> tuple[int, int]:
    """Latency mode (optional): one frame across the GPUs.  Rank r owns the contiguous slab
    [lo, hi) of the n query points (equal slabs, the last one shorter); z is the fastest grid axis, so a slab is a run of
    whole (x, y) columns up to its two ends."""
    per = (n + world_size - 1) // world_size
    return min(rank * per, n), min((rank + 1) * per, n)


def all_gather_slabs(local: torch.Tensor, n: int, group=None, force: bool = False) -> torch.Tensor:
    """The one exchange of latency mode: every rank contributes the values of its slab (shard_range order) and receives
    the whole (n,) array -- 67 MB for a 256^3 occupancy volume, after which marching cubes runs wherever it is wanted."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not force:
        return local[:n]
    per = (n + world - 1) // world
    pad = torch.zeros(per, dtype=local.dtype, device=local.device)
    pad[:local.numel()] = local.reshape(-1)
    flat = torch.empty(world * per, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(flat, pad, group=group)
    return flat[:n]


def _pack(meshes, device):
    counts = torch.tensor([[m['v'].shape[0], m['f'].shape[0]] for m in meshes], dtype=torch.int64, device=device).reshape(-1, 2)
    vn = [torch.cat([m['v'].reshape(-1, 3), m['vn'].reshape(-1, 3)], 1).reshape(-1) for m in meshes]
    fs = [m['f'].reshape(-1).to(torch.int32) for m in meshes]
    vbuf = torch.cat(vn) if vn else torch.empty(0, dtype=torch.float32, device=device)
    fbuf = torch.cat(fs) if fs else torch.empty(0, dtype=torch.int32, device=device)
    return counts, vbuf.to(torch.float32), fbuf


def _collective_device(meshes, group, device):
    """Device of the send / receive buffers: the meshes' own, else the caller's, else -- for a rank that owns no frame (fewer frames
    than ranks) -- the current HIP device under the nccl (= RCCL) backend, which cannot move host tensors, and the host under gloo."""
    if device is not None:
        return torch.device(device)
    if meshes:
        return meshes[0]['v'].device
    if dist.is_initialized() and dist.get_backend(group) == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def all_gather_meshes(meshes: list[dict], n_frames: int, group=None, force: bool = False, device=None) -> list[dict]:
    """meshes: this rank's frames -- exactly shard_frames(n_frames, rank, world), in that (ascending) order -- each
    {'v' (V,3) f32, 'vn' (V,3) f32, 'f' (F,3) i32}.  Returns, on every rank, all n_frames meshes in frame order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1 and not force:      # `force` runs the collectives even with one rank (RCCL smoke test)
        return list(meshes)
    mine = len(shard_frames(n_frames, rank, world))
    if len(meshes) != mine:
        raise ValueError(f'all_gather_meshes: rank {rank} of {world} owns {mine} of {n_frames} frames but was given {len(meshes)} meshes')
    device = _collective_device(meshes, group, device)
    kmax = (n_frames + world - 1) // world
    counts, vbuf, fbuf = _pack(meshes, device)
    cpad = torch.zeros((kmax, 2), dtype=torch.int64, device=device)
    cpad[:counts.shape[0]] = counts
    all_counts = [torch.empty_like(cpad) for _ in range(world)]
    dist.all_gather(all_counts, cpad, group=group)
    all_counts = torch.stack(all_counts).cpu()                       # (world, kmax, 2)
    vmax = int(all_counts[:, :, 0].sum(1).max()) * 6
    fmax = int(all_counts[:, :, 1].sum(1).max()) * 3
    vpad = torch.zeros(max(vmax, 1), dtype=torch.float32, device=device); vpad[:vbuf.numel()] = vbuf
    fpad = torch.zeros(max(fmax, 1), dtype=torch.int32, device=device); fpad[:fbuf.numel()] = fbuf
    # one flat receive buffer per collective (no per-rank staging copies); the per-rank pieces below are views of it
    flat_v = torch.empty(world * vpad.numel(), dtype=vpad.dtype, device=device)
    flat_f = torch.empty(world * fpad.numel(), dtype=fpad.dtype, device=device)
    dist.all_gather_into_tensor(flat_v, vpad, group=group)
    dist.all_gather_into_tensor(flat_f, fpad, group=group)
    all_v, all_f = flat_v.reshape(world, -1), flat_f.reshape(world, -1)
    out = [None] * n_frames
    for r in range(world):
        vo = fo = 0
        for k, f in enumerate(shard_frames(n_frames, r, world)):
            V, Fn = int(all_counts[r, k, 0]), int(all_counts[r, k, 1])
            vn = all_v[r][vo:vo + 6 * V].reshape(V, 6)
            out[f] = {'v': vn[:, :3], 'vn': vn[:, 3:], 'f': all_f[r][fo:fo + 3 * Fn].reshape(Fn, 3)}
            vo += 6 * V; fo += 3 * Fn
    return out


# ---- the sharded frame loop of `-m test` and the process plumbing around it -------------------------------------------------------------
def free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        return so.getsockname()[1]


def self_launch(script: str, argv: list[str], n_ranks: int) -> int:
    """`python <script> --gpus N ...` with N > 1 and no launcher around it: start N ranks of the script, one per GPU, under
    torch.distributed.run on 127.0.0.1 and return its exit status (rank 0's stdout passes through)."""
    import os
    import subprocess
    import sys
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n_ranks}', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(script)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '8')
    return subprocess.call(cmd, env=env)


def init_process_group(backend: str, rank: int, world: int, device=None, timeout_s: float = 120.0):
    """torch.distributed.init_process_group with a BOUNDED rendezvous and bounded collectives: a rank that never shows up (a GPU that failed to
    come up, a wrong WORLD_SIZE) ends the job with a message and a non-zero status after `timeout_s` instead of hanging it."""
    import datetime
    import os
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    kw = {'device_id': device} if (backend == 'nccl' and device is not None) else {}
    try:
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s), **kw)
    except Exception as e:      # noqa: BLE001 -- DistNetworkError / DistStoreError / RuntimeError depending on where the rendezvous gave up
        raise SystemExit(f'rank {rank}: rendezvous of {world} ranks at {os.environ["MASTER_ADDR"]}:{os.environ["MASTER_PORT"]} failed within '
                         f'{timeout_s:.0f} s ({type(e).__name__}: {e})')
    got = dist.get_world_size()
    if got != world:
        raise SystemExit(f'rank {rank}: the process group has {got} ranks, {world} were asked for')


def barrier_or_die(what: str, rank: int, timeout_s: float = 120.0, group=None):
    """dist.barrier() that turns a peer's absence into an error message: under nccl the watchdog aborts the process after the group's timeout;
    under gloo the call raises.  Either way the job ends non-zero instead of hanging."""
    try:
        dist.barrier(group=group)
    except Exception as e:      # noqa: BLE001
        raise SystemExit(f'rank {rank}: barrier "{what}" failed ({type(e).__name__}: {e}) -- a peer rank is gone or stuck')


def run_sharded(frames: list, process, rank: int = 0, world: int = 1, log=print) -> dict:
    """The frame loop of main.py:348, sharded: this rank runs frames[rank::world] in order.  `process(k, frame, next_frame)` does one
    frame (next_frame: the one this rank runs after it, or None -- FramePipeline's look-ahead).  A frame that raises is logged and SKIPPED: the
    loop carries no state between frames (main.py:348), so one bad frame -- a missing .exr, an empty surface -- does not take the others down.
    Returns {'done': [frames], 'failed': [(frame, 'Type: message')], 'results': {frame: what process returned}}."""
    import traceback
    mine = [frames[i] for i in shard_frames(len(frames), rank, world)]
    done, failed, results = [], [], {}
    for k, fr in enumerate(mine):
        nxt = mine[k + 1] if k + 1 < len(mine) else None
        try:
            results[fr] = process(k, fr, nxt)
            done.append(fr)
        except Exception as e:      # noqa: BLE001 -- per-frame containment is the point
            failed.append((fr, f'{type(e).__name__}: {e}'))
            log(f'# rank {rank}: frame {fr} FAILED and is skipped -- {type(e).__name__}: {e}')
            log(''.join(traceback.format_exception(type(e), e, e.__traceback__)).rstrip())
    return {'done': done, 'failed': failed, 'results': results}


def gather_summaries(summary: dict, group=None) -> list[dict]:
    """Every rank's {'done', 'failed'} lists on every rank (one all_gather_object; host data, a few hundred bytes)."""
    small = {'done': summary['done'], 'failed': summary['failed']}
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [small]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, small, group=group)
    return out
