"""Frame sharding + mesh all-gather over torch.distributed with the gloo backend, world_size 2,
on CPU (the N > 1 path of bench.py; RCCL on the GPU box).  CPU only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from avatarcap_amd.parallel import all_gather_meshes, all_gather_slabs, shard_frames, shard_range


def _mesh(frame):
    rs = np.random.RandomState(100 + frame)
    V, F = 5 + 3 * frame, 2 + frame * (frame % 3)
    return {'v': torch.from_numpy(rs.randn(V, 3).astype(np.float32)), 'vn': torch.from_numpy(rs.randn(V, 3).astype(np.float32)),
            'f': torch.from_numpy(rs.randint(0, V, (F, 3)).astype(np.int32))}


def _worker(rank, world, port, n_frames, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        mine = [_mesh(f) for f in shard_frames(n_frames, rank, world)]
        out = all_gather_meshes(mine, n_frames)
        ok = len(out) == n_frames
        for f, m in enumerate(out):
            ref = _mesh(f)
            ok = ok and all(torch.equal(m[k], ref[k]) for k in ('v', 'vn', 'f'))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def test_shard_frames():
    assert shard_frames(7, 0, 2) == [0, 2, 4, 6] and shard_frames(7, 1, 2) == [1, 3, 5]
    assert sorted(sum((shard_frames(64, r, 8) for r in range(8)), [])) == list(range(64))
    assert shard_frames(1, 3, 8) == []


@pytest.mark.parametrize('n_frames', [5, 2])
def test_all_gather_meshes_world2(n_frames):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_all_gather_single_process_is_identity():
    m = [_mesh(0), _mesh(1)]
    assert all_gather_meshes(m, 2) is not m and len(all_gather_meshes(m, 2)) == 2


def _slab_worker(rank, world, port, n, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        full = torch.arange(n, dtype=torch.float32) * 0.5 - 3
        lo, hi = shard_range(n, rank, world)
        got = all_gather_slabs(full[lo:hi].clone(), n)
        q.put((rank, bool(torch.equal(got, full))))
    finally:
        dist.destroy_process_group()


def test_shard_range():
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert shard_range(16777216, 7, 8) == (14680064, 16777216)
    assert torch.equal(all_gather_slabs(torch.arange(5.), 5), torch.arange(5.))          # no process group: identity


@pytest.mark.parametrize('n', [1001, 3])
def test_all_gather_slabs_world2(n):
    """Latency mode's one exchange: the occupancy slabs of a frame split over the ranks come back as the whole volume."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slab_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus N` without a launcher must start N ranks itself (round 1 silently measured one GPU) and report the number
    of ranks the process group really has; --frames F is BASELINE configs[4]'s batch (F frames sharded N-way, meshes all-gathered).
    --dry-run swaps the GPU work for stand-in meshes on gloo, so the launcher / rendezvous / gather logic runs here."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    env = dict(os.environ); env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    for argv, ranks, frames in ((['--gpus', '2', '--steps', '3', '--dry-run'], 2, 6), (['--gpus', '4', '--frames', '8', '--dry-run'], 4, 8),
                                (['--gpus', '1', '--dry-run'], 1, 5)):
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + argv, capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, r.stdout                                   # exactly ONE line on stdout
        line = json.loads(lines[0])
        assert line['n_gpus'] == ranks and line['gloo_ranks'] == ranks and line['frames'] == frames and line['all_gather_ok'] is True
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '3', '--frames', '8', '--dry-run'], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and 'not a multiple' in (r.stderr + r.stdout)


def test_all_gather_meshes_validates_its_shard():
    """A rank must hand over exactly its shard_frames(); a mismatch raises instead of silently mis-assigning frames."""
    import torch
    import torch.distributed as dist
    from avatarcap_amd.parallel import all_gather_meshes
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0)); port = so.getsockname()[1]
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    try:
        m = {'v': torch.zeros(4, 3), 'vn': torch.zeros(4, 3), 'f': torch.zeros((2, 3), dtype=torch.int32)}
        with pytest.raises(ValueError, match='owns 2 of 2 frames'):
            all_gather_meshes([m], 2, force=True)
        out = all_gather_meshes([m, m], 2, force=True)
        assert len(out) == 2 and out[1]['v'].shape == (4, 3)
        assert all_gather_meshes([], 0, force=True) == []                   # a rank / batch without frames
    finally:
        dist.destroy_process_group()


def test_main_shards_frames_and_contains_a_failing_frame(tmp_path):
    """`python main.py -m test --gpus N`: the product entry point shards main.py:348's frame loop itself -- frame k of the
    list on rank k mod N, every rank writing its own files --, skips a frame that raises (logged, exit status 1 at the end, the others done) and, with
    --gather-meshes, all-gathers the batch's meshes to rank 0.  --dry-run swaps FramePipeline for stand-in meshes on gloo; the loop, the launcher,
    the rendezvous and the gather are the product's."""
    import subprocess
    import sys
    import numpy as np
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    env = dict(os.environ); env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    ok = tmp_path / 'ok'
    r = subprocess.run([sys.executable, os.path.join(root, 'main.py'), '-m', 'test', '--dry-run', '--frames', '7', '--gpus', '2', '--gather-meshes',
                        '--gather-batch', '2', '--output-dir', str(ok)], capture_output=True, text=True, env=env, timeout=600)       # 4 steps in batches of 2
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert sorted(p.name for p in ok.glob('*_mesh.npz')) == ['%04d_mesh.npz' % f for f in range(7)]
    lines = r.stdout.splitlines()
    assert sum('rank 0: frame' in l for l in lines) == 4 and sum('rank 1: frame' in l for l in lines) == 3       # 0,2,4,6 | 1,3,5
    assert any('7 of 7 frames done on 2 rank(s)' in l for l in lines)
    allm = np.load(ok / 'all_avatar_meshes.npz')
    assert allm['frames'].tolist() == list(range(7))
    for f in range(7):
        assert allm['v_%04d' % f].shape == (5 + f % 7, 3) and float(allm['v_%04d' % f][0, 0]) == f + 0.5 and int(allm['f_%04d' % f][0, 0]) == f
    bad = tmp_path / 'bad'
    r = subprocess.run([sys.executable, os.path.join(root, 'main.py'), '-m', 'test', '--dry-run', '--frames', '6', '--gpus', '2', '--gather-meshes',
                        '--dry-fail', '3', '--output-dir', str(bad)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    assert sorted(p.name for p in bad.glob('*_mesh.npz')) == ['%04d_mesh.npz' % f for f in (0, 1, 2, 4, 5)]       # frame 3 skipped, 5 (same rank, after it) done
    assert 'frame 3 FAILED and is skipped' in r.stdout and '5 of 6 frames done on 2 rank(s); FAILED: 3 (RuntimeError' in r.stdout
    allm = np.load(bad / 'all_avatar_meshes.npz')
    assert allm['v_0003'].shape == (0, 3) and allm['v_0005'].shape == (10, 3)                                     # the failed frame travels as an empty mesh


def test_run_sharded_and_bounded_rendezvous():
    """run_sharded: order, look-ahead argument, containment; init_process_group: a rendezvous that cannot complete ends with a message, not a hang."""
    import subprocess
    import sys
    from avatarcap_amd.parallel import run_sharded
    seen = []

    def process(k, fr, nxt):
        seen.append((k, fr, nxt))
        if fr == 'c':
            raise ValueError('boom')
        return fr.upper()
    s = run_sharded(list('abcdefg'), process, rank=0, world=2, log=lambda m: None)
    assert seen == [(0, 'a', 'c'), (1, 'c', 'e'), (2, 'e', 'g'), (3, 'g', None)]
    assert s['done'] == ['a', 'e', 'g'] and s['failed'] == [('c', 'ValueError: boom')] and s['results'] == {'a': 'A', 'e': 'E', 'g': 'G'}
    assert run_sharded([], process, 0, 1)['done'] == []
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    code = ("import sys; sys.path.insert(0, %r); from avatarcap_amd import parallel; import os; os.environ['MASTER_PORT'] = str(parallel.free_port());"
            "parallel.init_process_group('gloo', 0, 2, timeout_s=3.0)" % root)                # rank 1 never comes
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'rendezvous of 2 ranks' in r.stderr and 'failed within 3 s' in r.stderr


def _exchange_worker(rank, world, port, n_frames, q):
    """The overlapped, exact-size exchange as bench.py / main.py drive it: one submit() per step, work between the steps, finish() at the end."""
    from avatarcap_amd.parallel import MeshExchange
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ex = MeshExchange(n_frames)
        for k in range(ex.steps):
            f = k * world + rank
            ex.submit(_mesh(f) if f < n_frames else None)
            torch.randn(64, 64) @ torch.randn(64, 64)                 # "the next frame" between two steps
        out = ex.finish()
        ok = len(out) == n_frames
        for f, m in enumerate(out):
            ref = _mesh(f)
            ok = ok and all(torch.equal(m[key], ref[key]) and m[key].dtype == ref[key].dtype for key in ('v', 'vn', 'f'))
        # exact sizes: what this rank received is the other ranks' meshes, word for word -- no padding to the largest
        want = sum(4 * (6 * _mesh(f)['v'].shape[0] + 3 * _mesh(f)['f'].shape[0]) for f in range(n_frames) if f % world != rank)
        q.put((rank, bool(ok), ex.bytes_received == want))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,n_frames', [(2, 7), (4, 10), (4, 3), (2, 1)])
def test_mesh_exchange_overlapped_exact_size(world, n_frames):
    """MeshExchange on gloo, world 2 and 4: frames that do not fill the last step (a rank submits None), fewer frames than ranks, meshes of different
    sizes (incl. F == 0): every rank ends with every mesh, bit for bit, and has received exactly the bytes of the others' meshes."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True, True) for r in range(world)]


def test_mesh_exchange_validates_its_steps():
    from avatarcap_amd.parallel import MeshExchange
    ex = MeshExchange(2)                                               # no process group: one rank, nothing travels
    with pytest.raises(ValueError, match='is missing'):
        ex.submit(None)
    ex = MeshExchange(2)
    ex.submit(_mesh(0))
    with pytest.raises(ValueError, match='1 of 2 steps'):
        ex.finish()
    ex.submit(_mesh(1))
    out = ex.finish()
    assert torch.equal(out[1]['v'], _mesh(1)['v'])
    with pytest.raises(ValueError, match='submit\\(\\) called again'):
        ex.submit(_mesh(2))


def test_run_sharded_stops_on_fatal_errors_and_failure_streaks():
    """A HIP fault or an out-of-memory error is sticky: the remaining frames are reported as failed WITHOUT being tried (each would fail the same way, with
    a traceback each); so are the frames behind three failures in a row.  A single bad frame is still contained."""
    from avatarcap_amd.parallel import run_sharded, _parse_cpulist, pin_to_gpu_numa
    tried = []

    def oom(k, fr, nxt):
        tried.append(fr)
        if fr == 2:
            raise torch.cuda.OutOfMemoryError('HIP out of memory')
        return fr
    s = run_sharded(list(range(6)), oom, log=lambda m: None)
    assert tried == [0, 1, 2] and s['done'] == [0, 1] and [f for f, _ in s['failed']] == [2, 3, 4, 5] and 'fatal device error' in s['aborted']
    assert s['failed'][1][1].startswith('not attempted')
    tried.clear()

    def flaky(k, fr, nxt):
        tried.append(fr)
        if fr in (1, 3, 4, 5):
            raise ValueError('bad frame')
        return fr
    s = run_sharded(list(range(8)), flaky, log=lambda m: None)
    assert tried == [0, 1, 2, 3, 4, 5] and s['done'] == [0, 2] and '3 frames in a row' in s['aborted'] and len(s['failed']) == 6
    tried.clear()
    s = run_sharded(list(range(4)), lambda k, fr, nxt: (_ for _ in ()).throw(RuntimeError('libavcap_hip: hipLaunch failed: HIP error (status -3)')), log=lambda m: None)
    assert len(s['done']) == 0 and s['failed'][1][1].startswith('not attempted')
    assert _parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11] and _parse_cpulist('') == []
    assert pin_to_gpu_numa(0) == {} or 'numa_node' in pin_to_gpu_numa(0)         # no GPU / no sysfs topology: a no-op, never an error


def _pump_worker(rank, world, port, n_frames, mode, q):
    """The exchange as the frame loops drive it since round 5: pump() inside "the next frame", submit() behind it.  `mode`:
    'all'   -- every rank pumps; ranks drift apart by more than a step's worth of time (rank r sleeps before some of its steps);
    'mixed' -- only even ranks pump, the others leave step k - 1 to submit(k): the collectives must still pair up (same order on every rank);
    'corrupt' -- after the exchange rank 1 damages what it received: verify_gathered_meshes must name the frames."""
    import time
    from avatarcap_amd.parallel import MeshExchange, verify_gathered_meshes, mesh_checksum
    if mode.endswith('+broadcast'):                                    # round 4's transport (one broadcast per mesh), kept for A/B on the 8-GPU box
        os.environ['AVC_EXCHANGE'] = 'broadcast'
        mode = mode[:-len('+broadcast')]
    else:
        os.environ.pop('AVC_EXCHANGE', None)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ex = MeshExchange(n_frames)
        assert ex.mode == os.environ.get('AVC_EXCHANGE', 'p2p')
        for k in range(ex.steps):
            if mode == 'all' and (k + rank) % 3 == 0:
                time.sleep(0.15 * (1 + rank))                          # this rank falls behind; its peers wait in pump() for its counts, not forever
            if mode != 'mixed' or rank % 2 == 0:
                ex.pump()                                              # "inside frame k": sends step k - 1
            f = k * world + rank
            ex.submit(_mesh(f) if f < n_frames else None)
        early = ex.pumped_early
        out = ex.finish()
        ok = len(out) == n_frames and all(torch.equal(out[f][key], _mesh(f)[key]) for f in range(n_frames) for key in ('v', 'vn', 'f'))
        mine = {f: _mesh(f) for f in shard_frames(n_frames, rank, world)}
        if mode == 'corrupt' and rank == 1:
            out[0]['v'][0, 0] += 1.0                                   # one word of a received mesh
            out[2], out[2 + world] = out[2 + world], out[2]            # two meshes of the same owner in each other's frame slots
        bad = verify_gathered_meshes(out, mine)
        want_early = (ex.steps - 1) if (mode != 'mixed' or rank % 2 == 0) else 0
        same = torch.equal(mesh_checksum(_mesh(3)), mesh_checksum(_mesh(3))) and not torch.equal(mesh_checksum(_mesh(3)), mesh_checksum(_mesh(4)))
        q.put((rank, bool(ok), early == want_early, bad, bool(same)))
    finally:
        dist.destroy_process_group()


def _run_pump(world, n_frames, mode):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pump_worker, args=(r, world, port, n_frames, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize('world,n_frames,mode', [(2, 7, 'all'), (4, 14, 'all'), (4, 3, 'all'), (2, 6, 'mixed'), (4, 9, 'mixed'), (4, 10, 'all+broadcast'), (2, 5, 'mixed+broadcast')])
def test_mesh_exchange_pump_overlaps_all_but_the_last_step(world, n_frames, mode):
    """pump() from inside the next frame sends step k - 1 before submit(k): of a rank's K steps K - 1 have gone out when finish() is reached, whatever
    the drift between the ranks, and a rank that never pumps still pairs its collectives with those of ranks that do (VERDICT round 4, next #1c).  Both
    transports: the batched point-to-point group (default: every pair of GPUs over its own xGMI link) and one broadcast per mesh."""
    for rank, ok, early_ok, bad, same in _run_pump(world, n_frames, mode):
        assert ok and early_ok and bad == [] and same, (rank, ok, early_ok, bad)


def test_gathered_mesh_checksums_catch_damage_and_misplaced_slots():
    """What bench.py's `meshes_verified` rests on: a changed word in a received mesh and two meshes delivered to each other's frame slots are
    reported, by frame, on the rank that holds them -- and only there."""
    res = _run_pump(2, 6, 'corrupt')
    assert res[0][1] and res[0][3] == []
    bad = res[1][3]
    assert [b.split(':')[0] for b in bad] == ['frame %d (owner rank 0) on rank 1' % f for f in (0, 2, 4)], bad


def _tune_worker(rank, world, port, q):
    """Every rank sees different timings; the choice must be the same on all of them: MAX over ranks per candidate, the earliest within tolerance of the best."""
    from avatarcap_amd.parallel import choose_exchange_config
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        cands = [('p2p', 8), ('p2p', 4), ('p2p', 0), ('broadcast', 8), ('broadcast', 4), ('broadcast', 0)]
        # rank r's own seconds: ('p2p', 4) is the fastest on ranks 0..2 but rank 3 is slow under it; ('broadcast', 8) is the best worst-case by far
        local = {('p2p', 8): 0.120 + 0.001 * rank, ('p2p', 4): 0.100 if rank < 3 else 0.150, ('p2p', 0): 0.130, ('broadcast', 8): 0.105,
                 ('broadcast', 4): 0.1055 - 0.0001 * rank, ('broadcast', 0): 0.140}
        calls = []
        r1 = choose_exchange_config(cands, lambda c: (calls.append(c), local[c])[1])
        # a tie inside the tolerance goes to the EARLIER candidate (the documented default first)
        near = {c: 0.1 for c in cands}; near[('broadcast', 0)] = 0.0995 if rank == 0 else 0.0990
        r2 = choose_exchange_config(cands, lambda c: near[c])
        q.put((rank, r1['choice'], r1['index'], [round(v, 3) for v in r1['table_ms']], calls == cands, r2['choice'], r1['ranks']))
    finally:
        dist.destroy_process_group()


def test_exchange_autotune_all_ranks_agree_world4():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tune_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len({r[1:] for r in [(x[0],) + tuple(map(lambda v: tuple(v) if isinstance(v, list) else v, x[1:])) for x in res]}) == 1     # identical on every rank
    _, choice, index, table, in_order, tie_choice, ranks = res[0]
    assert choice == ('broadcast', 8) and index == 3 and ranks == 4 and in_order          # 105.0 ms worst case; ('broadcast', 4) at 105.5 is within 2 % but later
    assert table == [123.0, 150.0, 130.0, 105.0, 105.5, 140.0]                             # MAX over the ranks per candidate
    assert tie_choice == ('p2p', 8)                                                        # 0.5 % faster is noise: the default stays


def test_exchange_autotune_single_process():
    from avatarcap_amd.parallel import choose_exchange_config
    r = choose_exchange_config(['a', 'b', 'c'], lambda c: {'a': 0.3, 'b': 0.2, 'c': 0.1}[c])
    assert r['choice'] == 'c' and r['index'] == 2 and r['ranks'] == 1
