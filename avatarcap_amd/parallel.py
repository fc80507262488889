"""Frame-sharded data parallelism.

The reference has no multi-GPU code; its frame loop (main.py:348) carries no state between frames, so
frames shard embarrassingly: frame f runs on rank f mod world_size, one process per GPU, weights /
grid / SMPL tables replicated.  The only exchange is an all-gather(v) of the finished meshes
(torch.distributed; backend 'nccl' = RCCL over xGMI on the GPU box, 'gloo' in the CPU tests), exact-size
and overlapped with the frames that follow (MeshExchange):
  step k = the frames k * world .. k * world + world - 1, one per rank;
  1. submit(mesh) packs the frame's mesh into one buffer and starts an asynchronous all-gather of the step's (V, F) counts (16 bytes per
     rank) from a side stream: pinned host buffers in and out, an event behind the copy -- the compute stream is never drained for it;
  2. pump() -- called by FramePipeline.avatar_frame right behind the NEXT frame's query launch, while the host has nothing to do -- waits
     for that event and issues the step's meshes -- EXACTLY 6 V + 3 F 32-bit words each ([verts | normals] and the faces in one buffer), no
     padding to the largest rank -- from the side stream as ONE batched group of point-to-point sends and receives: every rank sends its
     buffer to each peer over the link the two share (the node's xGMI is fully connected), so RCCL moves step k - 1 while frame k computes.  A caller that never pumps still gets the same collectives in the same order (submit(k) sends step k - 1 first);
  3. finish() sends what has not travelled -- with pump() in the frame loop: the LAST step only, i.e. each rank's last mesh --, makes the
     caller's stream wait for everything in flight and returns all meshes in frame order on every rank.
verify_gathered_meshes() is the exchange's self-check (per-frame integer checksums from the owners against what arrived, slot by slot).
all_gather_meshes(meshes, n_frames) is the same exchange for meshes that already exist.
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


class MeshExchange:
    """Exact-size, overlapped all-gather(v) of a batch's per-frame meshes (module docstring).  Every rank calls submit() once per step, in step
    order, with its frame of that step ({'v' (V,3) f32, 'vn' (V,3) f32, 'f' (F,3) i32}) or None when it owns none (the last step of a batch whose
    size is not a multiple of the world size), then finish().  pump() may be called at any time in between (FramePipeline.avatar_frame calls it right
    behind the query launch): it issues the payload broadcasts of every step submitted so far, so that step k - 1 travels WHILE frame k computes and
    only the last step is left for finish().  Whoever calls what when, every rank issues the same collectives in the same order -- AG(0), B(0) x world,
    AG(1), B(1) x world, ... -- because submit(k) pumps step k - 1 itself before it starts AG(k).
    `force` runs the collectives even with one rank (the single-GPU RCCL test)."""

    def __init__(self, n_frames: int, group=None, device=None, force: bool = False, mode: str | None = None):
        import os
        self.n_frames, self.group, self.force = int(n_frames), group, force
        # how a step's meshes travel: 'p2p' (default) -- every rank SENDS its buffer to each peer and receives each peer's, one batched group of
        # point-to-point operations: on the node's fully connected xGMI every pair has a link of its own (7 x ~77 GB/s inbound per GPU), where a ring
        # broadcast or all-gather is bound by ONE link for all eight meshes (~9x the time at N = 8); 'broadcast' -- `world` broadcasts (round 4's form,
        # kept for A/B on the 8-GPU box: AVC_EXCHANGE=broadcast)
        self.mode = mode or os.environ.get('AVC_EXCHANGE', 'p2p')
        if self.mode not in ('p2p', 'broadcast'):
            raise ValueError(f"MeshExchange: mode is 'p2p' or 'broadcast', not {self.mode!r}")
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.active = self.world > 1 or force
        self.device = torch.device(device) if device is not None else None
        self.steps = (self.n_frames + self.world - 1) // self.world
        self.out = [None] * self.n_frames
        self._k = 0                    # steps submitted
        self._sent = 0                 # steps whose payload broadcasts have been issued
        self._counts, self._mine, self._own, self._works, self._foreign = [], [], [], [], []
        self._comm = None              # HIP: the side stream the collectives are issued from (they then wait for IT, not for the compute stream)
        self.bytes_received = 0
        self.pumped_early = 0          # steps whose broadcasts were issued by pump() (inside the next frame), not by submit() / finish()

    def _src(self, r):                 # global rank of group rank r (broadcast takes global ranks)
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def submit(self, mesh):
        k = self._k
        if k >= self.steps:
            raise ValueError(f'MeshExchange: {self.steps} steps for {self.n_frames} frames on {self.world} ranks, submit() called again')
        f = k * self.world + self.rank
        if (mesh is None) != (f >= self.n_frames):
            raise ValueError(f'MeshExchange: rank {self.rank} step {k}: frame {f} of {self.n_frames} ' + ('is missing' if mesh is None else 'does not exist'))
        if not self.active:
            self._k += 1
            self.out[f] = {'v': mesh['v'], 'vn': mesh['vn'], 'f': mesh['f']}
            return
        if self.device is None:
            self.device = _collective_device([mesh] if mesh is not None else [], self.group, None)
        dev = self.device
        self._pump(early=False)        # step k - 1, unless pump() has sent it already: B(k - 1) precedes AG(k) on every rank
        if mesh is None:
            buf, V, F = torch.empty(0, dtype=torch.int32, device=dev), 0, 0
        else:
            V, F = int(mesh['v'].shape[0]), int(mesh['f'].shape[0])
            vn = torch.cat([mesh['v'].reshape(-1, 3), mesh['vn'].reshape(-1, 3)], 1).to(torch.float32).contiguous()
            buf = torch.cat([vn.reshape(-1).view(torch.int32), mesh['f'].reshape(-1).to(torch.int32)])       # 6 V + 3 F words, exactly
        if dev.type == 'cuda':
            # The counts are host integers (tensor shapes): they go out from a pinned buffer on the side stream and come back into one, with an event
            # behind the copy.  Nothing of this touches the compute stream: no `.cpu()` that would drain it, and RCCL's stream -- which waits for
            # the stream a collective is ISSUED from -- waits for the side stream only.
            if self._comm is None:
                self._comm = torch.cuda.Stream(dev)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(dev))                      # the packed buffer is complete once the compute stream gets here
            cin = torch.tensor([V, F], dtype=torch.int64).pin_memory()
            chost = torch.empty(2 * self.world, dtype=torch.int64).pin_memory()
            with torch.cuda.stream(self._comm):
                counts = cin.to(dev, non_blocking=True)
                allc = torch.empty(2 * self.world, dtype=torch.int64, device=dev)
                w = dist.all_gather_into_tensor(allc, counts, group=self.group, async_op=True)
                w.wait()                                                       # the SIDE stream waits for RCCL's; the host does not
                chost.copy_(allc, non_blocking=True)
                arrived = torch.cuda.Event()
                arrived.record(self._comm)
            self._counts.append({'host': chost, 'arrived': arrived, 'ready': ready, 'keep': (cin, counts, allc)})
        else:
            counts = torch.tensor([V, F], dtype=torch.int64, device=dev)
            allc = torch.empty(2 * self.world, dtype=torch.int64, device=dev)
            w = dist.all_gather_into_tensor(allc, counts, group=self.group, async_op=True)
            self._counts.append({'host': allc, 'work': w})
        self._mine.append(buf)
        self._own.append((V, F))
        self._k += 1

    def pump(self):
        """Issue the payload broadcasts of every submitted step that has not travelled yet (normally: the previous frame's step, called from inside
        the current frame, behind its query launch).  Waits on the host for that step's counts, i.e. for the slowest rank to have SUBMITTED the
        step -- never for this rank's own compute stream.  A no-op when there is nothing to send."""
        if self.active:
            self._pump(early=True)

    def _pump(self, early):
        for j in range(self._sent, self._k):
            self._send(j)
            self.pumped_early += int(early)

    def _send(self, j):
        """Step j's payload: `world` broadcasts of exact size (asynchronous; the receive buffers are the output meshes' storage)."""
        c = self._counts[j]
        if 'arrived' in c:
            c['arrived'].synchronize()                                         # side stream only
        else:
            c['work'].wait()
        cnt = c['host'].reshape(self.world, 2).tolist()
        if cnt[self.rank] != list(self._own[j]):
            raise RuntimeError(f'MeshExchange: rank {self.rank} step {j}: the gathered counts hold {cnt[self.rank]} in this rank\'s slot, it sent {list(self._own[j])}')
        import contextlib
        on_side = self._comm is not None
        with (torch.cuda.stream(self._comm) if on_side else contextlib.nullcontext()):
            if on_side:
                self._comm.wait_event(c['ready'])                              # this rank's packed buffer of step j
            ops = []
            for r in range(self.world):
                f = j * self.world + r
                if f >= self.n_frames:
                    continue
                V, F = int(cnt[r][0]), int(cnt[r][1])
                if r == self.rank:
                    buf = self._mine[j]
                    if buf.numel() != 6 * V + 3 * F:
                        raise RuntimeError(f'MeshExchange: rank {self.rank} step {j}: own buffer holds {buf.numel()} words, counts say {6 * V + 3 * F}')
                else:
                    buf = torch.empty(6 * V + 3 * F, dtype=torch.int32, device=self.device)       # HIP: from the side stream's pool (finish() hands it over)
                    self.bytes_received += 4 * buf.numel()
                    self._foreign.append(buf)
                if buf.numel():
                    if self.mode == 'broadcast':
                        self._works.append(dist.broadcast(buf, src=self._src(r), group=self.group, async_op=True))
                    elif r != self.rank:
                        ops.append(dist.P2POp(dist.irecv, buf, self._src(r), self.group))
                vn = buf[:6 * V].view(torch.float32).reshape(V, 6)
                self.out[f] = {'v': vn[:, :3], 'vn': vn[:, 3:], 'f': buf[6 * V:].reshape(F, 3)}
            if self.mode == 'p2p':
                mine = self._mine[j]
                if mine.numel():                                               # (an empty mesh -- a failed frame, a rank without a frame -- is neither sent nor received: both ends know from the counts)
                    ops += [dist.P2POp(dist.isend, mine, self._src(r), self.group) for r in range(self.world) if r != self.rank]
                if ops:
                    self._works += dist.batch_isend_irecv(ops)                 # one group: RCCL pairs the sends and receives whatever their order in the list
        self._sent = j + 1

    def finish(self) -> list:
        if self._k != self.steps:
            raise ValueError(f'MeshExchange.finish: {self._k} of {self.steps} steps were submitted')
        if self.active:
            self._pump(early=False)
            for w in self._works:
                w.wait()                                                       # HIP: the CURRENT stream waits for RCCL's; the meshes are its to read from here on
            self._works.clear()
            if self._comm is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_stream(self._comm)
                for buf in self._foreign:                                      # allocated on the side stream, read (and one day freed) on the caller's
                    buf.record_stream(cur)
            self._foreign.clear()
            self._counts.clear()
        return self.out


def mesh_checksum(mesh, device=None) -> torch.Tensor:
    """(V, F, sum of the 32-bit patterns of [v | vn], sum of the face indices, position-weighted sum of all words) as five int64 -- exact integer
    arithmetic (wrapping), so two copies of a mesh agree bit for bit or the checksums differ; the weighted sum catches permuted or shifted content."""
    if mesh is None:
        return torch.zeros(5, dtype=torch.int64, device=device)
    v = torch.cat([mesh['v'].reshape(-1, 3), mesh['vn'].reshape(-1, 3)], 1).to(torch.float32).contiguous().reshape(-1).view(torch.int32).to(torch.int64)
    f = mesh['f'].reshape(-1).to(torch.int64)
    words = torch.cat([v, f])
    weighted = (words * (torch.arange(words.numel(), dtype=torch.int64, device=words.device) % 65521 + 1)).sum() if words.numel() else words.sum()
    return torch.stack([torch.tensor(mesh['v'].shape[0], dtype=torch.int64, device=words.device), torch.tensor(mesh['f'].shape[0], dtype=torch.int64, device=words.device),
                        v.sum(), f.sum(), weighted])


def verify_gathered_meshes(gathered: list, mine: dict, group=None, force: bool = False) -> list[str]:
    """Self-validation of an exchange (bench.py runs it OUTSIDE the timed region): every owner's checksum of the mesh it PRODUCED (`mine`: frame -> mesh,
    this rank's frames) is all-gathered, and every rank compares the checksum of every mesh it RECEIVED, slot by slot.  Returns the list of
    complaints (empty: every mesh is in its frame slot, complete and bit-identical to its owner's)."""
    n = len(gathered)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = next((m['v'].device for m in gathered if m is not None), torch.device('cpu'))
    own = torch.zeros((n, 5), dtype=torch.int64, device=dev)
    for f, m in mine.items():
        if f % world != rank:
            return [f'rank {rank} claims frame {f}, which rank {f % world} owns']
        own[f] = mesh_checksum(m, dev)
    if (world > 1 or force) and dist.is_initialized():
        allsum = torch.empty((world, n, 5), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allsum.reshape(-1), own.reshape(-1).contiguous(), group=group)
    else:
        allsum = own[None]
    allsum = allsum.cpu()
    bad = []
    for f in range(n):
        want = allsum[f % world, f]
        if gathered[f] is None:
            bad.append(f'frame {f}: nothing received')
            continue
        got = mesh_checksum(gathered[f], dev).cpu()
        if not torch.equal(got, want):
            bad.append(f'frame {f} (owner rank {f % world}) on rank {rank}: received (V, F, sums) {got.tolist()} != produced {want.tolist()}')
    return bad


def all_gather_meshes(meshes: list[dict], n_frames: int, group=None, force: bool = False, device=None) -> list[dict]:
    """meshes: this rank's frames -- exactly shard_frames(n_frames, rank, world), in that (ascending) order -- each
    {'v' (V,3) f32, 'vn' (V,3) f32, 'f' (F,3) i32}.  Returns, on every rank, all n_frames meshes in frame order (MeshExchange, all steps at once)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1 and not force:      # `force` runs the collectives even with one rank (RCCL smoke test)
        return list(meshes)
    mine = len(shard_frames(n_frames, rank, world))
    if len(meshes) != mine:
        raise ValueError(f'all_gather_meshes: rank {rank} of {world} owns {mine} of {n_frames} frames but was given {len(meshes)} meshes')
    ex = MeshExchange(n_frames, group, _collective_device(meshes, group, device), force)
    for k in range(ex.steps):
        ex.submit(meshes[k] if k < len(meshes) else None)
    return ex.finish()


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


def init_process_group(backend: str, rank: int, world: int, device=None, timeout_s: float = 120.0, collective_timeout_s: float | None = None):
    """torch.distributed.init_process_group with a BOUNDED rendezvous: a rank that never shows up (a GPU that failed to come up, a wrong WORLD_SIZE)
    ends the job with a message and a non-zero status after `timeout_s` instead of hanging it.  The collectives that follow get their OWN bound,
    `collective_timeout_s` (default 10 x the rendezvous bound, at least 30 min): in `main.py -m test` the first collective after the rendezvous is reached
    only when a rank has finished its whole shard, and ranks legitimately drift apart by minutes on a long sequence (I/O, a slower GPU, frames that fail
    fast) -- one timeout for both would either let a dead rendezvous hang for half an hour or kill a healthy job."""
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
    # every rank is here: from now on the (longer) collective bound applies
    barrier_or_die('rendezvous', rank, timeout_s)
    if collective_timeout_s is None:
        collective_timeout_s = max(10.0 * timeout_s, 1800.0)
    try:
        from torch.distributed.distributed_c10d import _set_pg_timeout
        _set_pg_timeout(datetime.timedelta(seconds=collective_timeout_s))
    except Exception as e:      # noqa: BLE001 -- a torch without the hook: one bound governs both (say so once)
        if rank == 0:
            print(f'# parallel: could not raise the collective timeout ({type(e).__name__}: {e}); rendezvous and collectives share {timeout_s:.0f} s', flush=True)


def leave_cus_for_the_exchange(device, spare: int = 8, log=None) -> int:
    """Multi-rank runs: the persistent query kernels take `CUs - spare` workgroups instead of one per CU (avc_set_option "mlp_blocks").  The fused queries
    hold a whole CU each (160 KB of LDS, every VGPR): a launch on ALL CUs leaves RCCL's copy kernels nowhere to run until it ends -- the exchange then
    lands on the frame's tail and, worse, on the NEXT query's launch, whose workgroups that find their CU taken start late and, tiles being assigned
    statically, end the launch that much later.  With a few CUs left free the exchange runs beside the query (`MeshExchange.pump`) and is gone before the
    next launch.  Cost on one GPU: +0.6 % query time for 8 CUs (profiles/r03_overlap_experiment.md; the launch is power-bound, the clock rises as CUs
    idle).  AVC_MLP_BLOCKS, when set, wins.  Returns the number of workgroups now used (0: unchanged)."""
    import os
    from . import _lib
    if os.environ.get('AVC_MLP_BLOCKS') or spare <= 0:
        return 0
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    n = max(1, cus - spare)
    _lib.set_option('mlp_blocks', n, device)
    if log:
        log(f'# fused queries on {n} of {cus} CUs: {spare} left to RCCL\'s copy kernels')
    return n


def set_spare_cus(device, spare: int) -> int:
    """`leave_cus_for_the_exchange` that can also take the CUs back (spare 0 -> one workgroup per CU again): what the warm-up's A/B switches between."""
    from . import _lib
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    n = 0 if spare <= 0 else max(1, cus - spare)
    _lib.set_option('mlp_blocks', n, device)
    return n


def choose_exchange_config(candidates: list, measure, group=None, tolerance: float = 0.02, device=None) -> dict:
    """The first multi-GPU run tunes itself: `candidates` is an ordered list of hashable configurations -- here (transport,
    spare CUs) -- and `measure(candidate)` returns this rank's seconds for a few steps under it (the caller brackets it with barriers).  Every
    candidate's time is the MAX over the ranks (one all-reduce of the whole table, so every rank decides on identical numbers); the choice is the
    EARLIEST candidate within `tolerance` of the fastest -- the list order is the preference (the documented default first), and a later candidate
    has to win by more than noise to displace an earlier one.  Deterministic: same table, same order, same answer on every rank.
    -> {'choice': candidate, 'index': i, 'table_ms': [...], 'ranks': world}."""
    times = []
    for c in candidates:
        times.append(float(measure(c)))
    t = torch.tensor(times, dtype=torch.float64)
    world = 1
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        if dist.get_backend(group) == 'nccl':
            t = t.to(device if device is not None else torch.device('cuda', torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        t = t.cpu()
    table = t.tolist()
    best = min(table)
    index = next(i for i, v in enumerate(table) if v <= best * (1.0 + tolerance))
    return {'choice': candidates[index], 'index': index, 'table_ms': [v * 1e3 for v in table], 'ranks': world}


def pin_to_gpu_numa(device_index: int, local_world: int = 1, log=None) -> dict:
    """CPU affinity of this rank := the cores of its GPU's NUMA node (/sys/bus/pci/devices/<bdf>/numa_node), split evenly among the ranks that share
    the node, and OMP / MKL threads to match: eight Python ranks on one host otherwise migrate across sockets and oversubscribe each other (bench.py's
    --gpus 8 run is host-paced between launches).  Best effort: returns what it did, {} when the topology is not readable (containers, no sysfs)."""
    import os
    try:
        props = torch.cuda.get_device_properties(device_index)
        bdf = f'{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0'
        node = int(open(f'/sys/bus/pci/devices/{bdf}/numa_node').read())
        if node < 0:
            return {}
        cpus = _parse_cpulist(open(f'/sys/devices/system/node/node{node}/cpulist').read())
        allowed = sorted(set(cpus) & os.sched_getaffinity(0))
        if not allowed:
            return {}
        # ranks whose GPUs sit on the same node share its cores: give each an equal, disjoint slice
        same = []
        for d in range(torch.cuda.device_count()):
            p = torch.cuda.get_device_properties(d)
            b = f'{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0'
            try:
                if int(open(f'/sys/bus/pci/devices/{b}/numa_node').read()) == node:
                    same.append(d)
            except OSError:
                pass
        same = [d for d in same if d < max(local_world, 1)] or [device_index]
        share = max(1, len(allowed) // len(same))
        k = same.index(device_index) if device_index in same else 0
        mine = allowed[k * share:(k + 1) * share] or allowed
        os.sched_setaffinity(0, mine)
        threads = max(1, min(len(mine), 16))
        torch.set_num_threads(threads)
        os.environ['OMP_NUM_THREADS'] = str(threads)
        info = {'numa_node': node, 'cpus': len(mine), 'first_cpu': mine[0], 'threads': threads}
        if log:
            log(f'# rank on GPU {device_index} ({bdf}): NUMA node {node}, pinned to {len(mine)} cores from {mine[0]}, {threads} threads')
        return info
    except Exception:           # noqa: BLE001 -- never fatal
        return {}


def _parse_cpulist(text: str) -> list[int]:
    out = []
    for part in text.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        out.extend(range(int(a), int(b or a) + 1))
    return out


def barrier_or_die(what: str, rank: int, timeout_s: float = 120.0, group=None):
    """dist.barrier() that turns a peer's absence into an error message: under nccl the watchdog aborts the process after the group's timeout;
    under gloo the call raises.  Either way the job ends non-zero instead of hanging."""
    try:
        dist.barrier(group=group)
    except Exception as e:      # noqa: BLE001
        raise SystemExit(f'rank {rank}: barrier "{what}" failed ({type(e).__name__}: {e}) -- a peer rank is gone or stuck')


def device_is_gone(e: BaseException) -> bool:
    """Errors after which the DEVICE is not worth another call -- neither a frame nor a collective: out of memory, a HIP runtime fault (sticky: every
    later call fails too)."""
    if isinstance(e, (MemoryError, torch.cuda.OutOfMemoryError)):
        return True
    msg = str(e)
    return 'HIP error' in msg or 'hipError' in msg or 'AVC_ERR_HIP' in msg or '(status -3)' in msg or getattr(e, 'status', None) == -3


def _range_error(e: BaseException) -> bool:
    """AVC_ERR_RANGE: these WEIGHTS leave the range of the split-fp16 arithmetic -- every frame would; the device itself is fine."""
    return getattr(e, 'status', None) == -5 or '(status -5)' in str(e)


def _fatal(e: BaseException) -> bool:
    return device_is_gone(e) or _range_error(e)


def run_sharded(frames: list, process, rank: int = 0, world: int = 1, log=print, max_consecutive_failures: int = 3) -> dict:
    """The frame loop of main.py:348, sharded: this rank runs frames[rank::world] in order.  `process(k, frame, next_frame)` does one
    frame (next_frame: the one this rank runs after it, or None -- FramePipeline's look-ahead).  A frame that raises is logged and SKIPPED: the
    loop carries no state between frames (main.py:348), so one bad frame -- a missing .exr, an empty surface -- does not take the others down.
    What is NOT contained: an out-of-memory or HIP runtime error (the device state is gone: the remaining frames are reported as failed without being
    tried), and `max_consecutive_failures` failures in a row OF THE SAME EXCEPTION TYPE (something systematic: same treatment; 0 or None disables
    the limit -- three missing .exr files in a row are then three skipped frames and nothing more).
    Returns {'done': [frames], 'failed': [(frame, 'Type: message')], 'results': {frame: what process returned}, 'aborted': reason or None}."""
    import traceback
    mine = [frames[i] for i in shard_frames(len(frames), rank, world)]
    done, failed, results = [], [], {}
    streak, streak_type, aborted = 0, None, None
    for k, fr in enumerate(mine):
        if aborted:
            failed.append((fr, f'not attempted: {aborted}'))
            continue
        nxt = mine[k + 1] if k + 1 < len(mine) else None
        try:
            results[fr] = process(k, fr, nxt)
            done.append(fr)
            streak, streak_type = 0, None
        except Exception as e:      # noqa: BLE001 -- per-frame containment is the point
            failed.append((fr, f'{type(e).__name__}: {e}'))
            log(f'# rank {rank}: frame {fr} FAILED and is skipped -- {type(e).__name__}: {e}')
            log(''.join(traceback.format_exception(type(e), e, e.__traceback__)).rstrip())
            streak = streak + 1 if type(e) is streak_type else 1
            streak_type = type(e)
            if _fatal(e):
                aborted = (f'frame {fr}: AVC_ERR_RANGE -- the checkpoint drives a feature or activation out of the fp16 range of the fused kernels'
                           if _range_error(e) else f'frame {fr} hit a fatal device error ({type(e).__name__})')
            elif max_consecutive_failures and streak >= max_consecutive_failures:
                aborted = f'{streak} frames in a row failed ({type(e).__name__})'
            if aborted:
                log(f'# rank {rank}: {aborted} -- the remaining {len(mine) - k - 1} frame(s) of this rank are not attempted')
    return {'done': done, 'failed': failed, 'results': results, 'aborted': aborted}


def gather_summaries(summary: dict, group=None) -> list[dict]:
    """Every rank's {'done', 'failed'} lists on every rank (one all_gather_object; host data, a few hundred bytes)."""
    small = {'done': summary['done'], 'failed': summary['failed']}
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [small]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, small, group=group)
    return out
