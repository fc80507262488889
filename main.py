#!/usr/bin/env python3
"""Test-mode entry point with the reference's CLI and config surface (main.py:507-529):

    python main.py -c configs/example.yaml -m test                       a captured sequence (cfg testing.testing_data_dir), like the reference
    python main.py -c configs/example.yaml -m test --synthetic --frames 2    synthetic body / poses / seeded weights (what bench.py and the tests use)

`-m test` runs `run_avatarcap`'s frame loop (main.py:348-498) through avatarcap_amd.pipeline on the HIP device:
  1. canonical avatar geometry (fused occupancy query -> marching cubes -> normals) and skinning to the live pose   (:357-389)
  2. canonical normal fusion of the image-observed normal map with the avatar's own maps                            (:405-433)
  3. the image-conditioned reconstruction network on the fused maps                                                 (:438-453)
  4. vertex colours from the texture template, transferred to the reconstruction                                    (:464-485)
and writes the meshes as PLY (obj_io.save_mesh_as_ply, :491-498) and .npz.  Not done here: the OpenGL Phong previews written as .jpg
(:391-399, :500-504 -- visualisation).

The loop is asynchronous on both ends (avatarcap_amd.frame_io): a worker thread loads the item dicts two frames ahead and uploads each with ONE
non-blocking copy from a pinned staging buffer on a copy stream; the finished meshes leave through pinned slots on the same copy stream and are written by
writer threads.  The loop thread itself only waits where marching cubes reads its counts.  `--sync-io` is the reference's shape of the loop (blocking
`to_cuda`, `.cpu()`, files written inline) for comparison; `--timing-json` writes what the loop measured about itself.

Multi-GPU (the reference has none): `--gpus N` (or a torchrun / torch.distributed.run launch) shards the frame list, frame k of it
on rank k mod N, one process per GPU; every rank writes the files of its own frames (names carry the data index, so ranks never collide);
`--gather-meshes` additionally all-gathers the batch's live avatar meshes over RCCL and has rank 0 write `<output_dir>/all_avatar_meshes.npz`.
A frame that raises is logged with its traceback and skipped (the loop carries no state between frames, main.py:348); the exit status is 1 if any
frame of any rank failed.  `--dry-run` (no GPU) runs launcher, rendezvous (gloo), sharding, error containment and gather with stand-in meshes.

With a captured sequence the loader is avatarcap_amd.avatarcap_dataset.AvatarCapDataset (the reference's dataset in test mode): it needs the
sequence directory, the checkpoints named in the yaml and the licensed SMPL model file under smpl_files/ (or $AVC_SMPL_DIR) -- none of which
can be redistributed; a missing file raises the FileNotFoundError the reference raises.  `-m train` is out of scope.
"""
import os
import sys
from argparse import ArgumentParser

import numpy as np
import torch


def _observed_normals_host(ds, data_idx, view_idx):
    """The image-observed normal map of step 2 (main.py:406-411): cv.imread(<...>.exr, IMREAD_UNCHANGED) -> (H, W, 3) float32, on the host (the prefetch
    thread uploads it with the rest of the frame's item dict)."""
    from avatarcap_amd.utils.exr_io import read_exr
    if ds.data_config['data_type'] == 'synthetic':
        path = ds.data_dir + '/imgs/%03d/normal_view_%03d.exr' % (data_idx, view_idx)
    elif ds.data_config['data_type'] == 'real':
        path = ds.data_dir + '/imgs/normal/normal_%04d.exr' % data_idx
    else:
        raise ValueError('Invalid data type!')
    return np.ascontiguousarray(read_exr(path)[..., :3], np.float32)


class _StandInPipeline:
    """--dry-run: what FramePipeline hands back, made up on the host (frame f -> a mesh of 5 + f % 7 vertices filled with f), so that launcher,
    rendezvous, sharding, per-frame error containment, file output and the mesh all-gather run without a GPU (tests/test_parallel_gloo.py).
    `fail` = frame numbers whose processing raises."""

    def __init__(self, fail=()):
        self.fail = set(fail)
        self.exchange = None

    def avatar_frame(self, items, next_items=None):
        f = int(items['data_idx'])
        if self.exchange is not None:            # where FramePipeline.avatar_frame pumps: behind the query launch
            self.exchange.pump()
        if f in self.fail:
            raise RuntimeError(f'stand-in failure of frame {f}')
        nv = 5 + f % 7
        v = torch.full((nv, 3), float(f))
        return {'cano_v': v, 'cano_vn': -v, 'f': torch.full((2 * nv, 3), f, dtype=torch.int32), 'live_v': v + 0.5, 'live_vn': -v,
                'occ_volume': torch.zeros(1)}


def run_avatarcap(w_recon=True, save_avatar_mesh=False, save_final_mesh=False, w_nerf=False, frame_idx=None, view_idx=0, interval=1,
                  synthetic=False, n_frames=2, valid='band', integrate_manner='merge', rank=0, world=1, gather_meshes=False, gather_batch=8,
                  dry_run=False, dry_fail=(), max_failure_streak=3, force_dist=False, save_npz=True, sync_io=False, io_threads=3, io_slots=4,
                  timing=None):
    from avatarcap_amd import config, parallel
    cfg = config.cfg
    out_dir = cfg['testing']['output_dir']
    os.makedirs(out_dir, exist_ok=True)
    log = lambda msg: (sys.stdout.write(str(msg) + '\n'), sys.stdout.flush())   # one write per line: ranks share the launcher's stdout    # noqa: E731

    nerf_net = None
    if dry_run:
        pipe, ds, renderer = _StandInPipeline(dry_fail), None, None
        w_recon = w_nerf = False
        img_num_per_pose, start_data_idx, data_num = 1, 0, n_frames
        load_host = lambda i: {'data_idx': i}                                     # noqa: E731
    else:
        from avatarcap_amd import synthetic as syn
        from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda
        from avatarcap_amd.network.arch_avatar import GeoTexAvatar, NerfRenderer
        from avatarcap_amd.network.arch_recon import ReconNetwork
        from avatarcap_amd.pipeline import FramePipeline
        if synthetic:
            network = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to(config.device).eval()
            syn.load_synth(network, syn.SEED)
            recon_net = ReconNetwork().to(config.device).eval()
            syn.load_synth(recon_net, syn.SEED)
            ds = SyntheticTestDataset(cfg['testing']['vol_res'], valid=valid, n_frames=n_frames)
            img_num_per_pose, start_data_idx, data_num = 1, 0, n_frames
        else:
            from avatarcap_amd.avatarcap_dataset import AvatarCapDataset
            network = GeoTexAvatar().to(config.device).eval()                       # main.py:296-297 (reads training_data_dir's blend-weight volume)
            if cfg['testing']['net_ckpt'] is not None:
                print('# Loading GeoTexAvatar network from %s' % cfg['testing']['net_ckpt'])
                network.load_state_dict(torch.load(cfg['testing']['net_ckpt'] + '/net.pt')['network'])     # :302-305
            fin = cfg['testing'].get('net_ckpt_finetuned', None)                    # :307-314: the texture template may come from a finetuned copy
            if fin is not None:
                print('# Loading finetuned GeoTexAvatar network from %s' % fin)
                nerf_net = GeoTexAvatar().to(config.device).eval()
                nerf_net.load_state_dict(torch.load(fin + '/net.pt')['network'])
            recon_net = ReconNetwork().to(config.device).eval()
            if cfg['testing'].get('recon_net_ckpt') is not None:
                print('# Loading reconstruction network from %s' % cfg['testing']['recon_net_ckpt'])
                recon_net.load_state_dict(torch.load(cfg['testing']['recon_net_ckpt'] + '/recon_net.pt')['network'])   # :316-320
            ds = AvatarCapDataset(cfg['testing']['testing_data_dir'], False)          # :323
            img_num_per_pose, start_data_idx = ds.img_num_per_pose, ds.start_data_idx
            data_num = len(ds) // img_num_per_pose
            print('# Data num: %d' % data_num)
        pipe = FramePipeline(network, ds, recon_net)
        renderer = NerfRenderer(nerf_net) if nerf_net is not None else None

        def load_host(i):
            """Everything of frame i that is read or made on the host (main.py:349-351 and the image-observed normal map of :406-411): runs on the
            prefetch thread, two frames ahead of the device."""
            item = dict(ds[i * img_num_per_pose + view_idx])
            if w_recon and synthetic:       # the stand-in of the captured normal map: a seeded field bends the posed normals (dataset.py)
                item['observed_bend_axes'] = torch.randn(3, 3, generator=torch.Generator().manual_seed(i)).numpy()
            elif w_recon:
                item['observed_normal'] = _observed_normals_host(ds, int(item['data_idx']), view_idx)
            return item

    if frame_idx is None:                                                        # main.py:337-345
        frames = list(range(0, data_num, interval))
    elif isinstance(frame_idx, int):
        frames = [frame_idx - start_data_idx]
    elif isinstance(frame_idx, list):
        frames = (np.array(frame_idx, np.int32) - start_data_idx).tolist()
    else:
        raise TypeError('Invalid frame_idx!')

    # Host work off the critical path (avatarcap_amd.frame_io): the item dicts of this rank's frames are loaded and uploaded two frames ahead by a worker
    # thread (one pinned staging buffer, one non-blocking copy per frame), the finished meshes leave through pinned slots to writer threads.
    import time
    from avatarcap_amd.frame_io import FramePrefetcher, MeshWriter
    dev = None if dry_run else config.device
    mine = [frames[j] for j in parallel.shard_frames(len(frames), rank, world)]
    prefetch = FramePrefetcher(load_host, mine, dev, depth=1 if sync_io else 2)
    writer = MeshWriter(dev, slots=io_slots, threads=io_threads)
    # (Leaving a few CUs to the output copies -- blit kernels on this runtime -- as multi-rank runs do for RCCL was measured and does not help: 21.7 / 22.7 /
    # 21.4 - 22.6 / 23.1 ms per frame with 0 / 4 / 8 / 16 spare CUs; what the PLY leg lost was host time between two frames, see emit_outputs below.)
    clock = {'start': [], 'bytes_written': 0, 'files': 0}
    lock = __import__('threading').Lock()

    # Weights read from disk have never been through the kernels: the split-fp16 arithmetic of the fused queries is exact only while every feature
    # and activation stays below 65504 (include/avcap.h, "numeric range"), and an overflow is SILENT (a ReLU swallows the NaN).  The first frame
    # of every rank therefore runs with the range check on (avc_set_range_check: the checked flavour of the kernels, one synchronisation per
    # query); a trip is fatal for the whole run (parallel.run_sharded: AVC_ERR_RANGE).  Synthetic weights are generated inside the range.
    # The user's own setting (config.check_range) is kept; the forced check stays on until one frame has COMPLETED under it (a first frame that fails
    # on a missing file has not checked anything).
    check = {'user': bool(getattr(config, 'check_range', False)), 'pending': not (synthetic or dry_run)}

    def process(k, i, nxt_i):
        clock['start'].append(time.perf_counter())
        checking = check['user'] or check['pending']
        config.check_range = checking
        try:
            items = prefetch.get(i)
            if sync_io and dev is not None:
                torch.cuda.synchronize(dev)                   # the reference's blocking to_cuda
            nxt = prefetch.peek(nxt_i)      # its U-Net is queued behind this frame's query (FramePipeline.avatar_frame); None: its own turn reports it
            return frame(k, i, items, nxt, checking)
        finally:
            prefetch.drop(i)

    def frame(k, i, items, nxt, checking):
        data_idx = int(items['data_idx'])
        a = pipe.avatar_frame(items, next_items=nxt)                              # step 1
        save = {'cano_v': a['cano_v'], 'cano_vn': a['cano_vn'], 'f': a['f'], 'live_v': a.get('live_v'), 'live_vn': a.get('live_vn')}
        if w_recon:
            # step 2: canonical normal fusion (main.py:405-433)
            if a['cano_v'].shape[0] > 0:
                if synthetic:       # the captured image's normal map is synthesised (dataset.synthetic_observed_normals)
                    from avatarcap_amd.dataset import synthetic_camera, synthetic_observed_normals
                    w2c, cam = synthetic_camera()
                    observed = synthetic_observed_normals(a['live_v'], a['live_vn'], a['f'], w2c, cam, seed=i, axes=items['observed_bend_axes'][0])
                else:
                    cam = ds.data_config['camera']
                    w2c = np.asarray(items['_host']['w2c_RT'], np.float32)          # as loaded: no read-back of what was just uploaded
                    observed = items['observed_normal'][0]
                items['front_normal'], items['back_normal'], _ = pipe.fuse_normals(a, observed, w2c, cam, integrate_manner)
            else:
                items['front_normal'], items['back_normal'] = pipe.cano_normal_maps(a['cano_v'], a['cano_vn'], a['f'])
            r = pipe.recon_frame(items)                                           # step 3
            save.update({'recon_' + k_: v for k_, v in r.items() if k_ != 'occ_volume'})
        if w_nerf:                                                                # step 4 (main.py:464-477)
            save['live_vc'] = pipe.colour_vertices(items, a['cano_v'], a['cano_vn'], renderer)
            if w_recon and save['recon_cano_v'].shape[0] > 0 and a['cano_v'].shape[0] > 0:        # main.py:478-482
                save['recon_live_vc'] = pipe.transfer_colours(save['recon_cano_v'], a['cano_v'], save['live_vc'])
        # ---- outputs (main.py:491-498): the file bytes are assembled on the device (obj_io.ply_records_device), copied out behind the frame's kernels
        # and written by the writer threads; nothing here waits for the device
        # Assembling and handing over takes the loop thread 1 - 2 ms, and HERE the device's queue is empty (marching cubes has just read its counts): the
        # section is parked and runs behind the NEXT frame's query launch (FramePipeline.after_query), or after the loop for the last frame.
        def outputs(save=save, a=a, data_idx=data_idx, i=i):
            try:
                emit_outputs(save, a, data_idx, i)
            except Exception as e:      # noqa: BLE001 -- reported for ITS frame, not for the frame in whose shadow it runs
                deferred_failures.append((i, f'{type(e).__name__}: {e}'))
        if sync_io or not hasattr(pipe, 'after_query'):
            outputs()
        else:
            pipe.after_query.append(outputs)
        log('# %sframe %d (data idx %d): avatar %d verts / %d faces%s' % ('rank %d: ' % rank if world > 1 else '', i, data_idx, a['cano_v'].shape[0],
            a['f'].shape[0], (', recon %d verts' % save['recon_cano_v'].shape[0]) if w_recon else ''))
        if checking:
            check['pending'] = False              # this frame went through every kernel with the range check on
        if gather_meshes and a.get('live_v') is not None:
            return {'v': a['live_v'], 'vn': a['live_vn'], 'f': a['f']}
        return None

    deferred_failures = []

    def emit_outputs(save, a, data_idx, i):
        from avatarcap_amd.utils import obj_io
        t_out = time.perf_counter()
        out_t, plys = {}, []
        if save_npz:
            out_t.update({k_: v for k_, v in save.items() if v is not None})
        if save_avatar_mesh and a.get('live_v') is not None:                          # main.py:491-493
            hdr, rec = obj_io.ply_records_device(a['live_v'], a['f'], a['live_vn'], save['live_vc'] if w_nerf else None)
            out_t.update({'avatar.' + k_: v for k_, v in rec.items()})
            plys.append(('%s/%04d_avatar.ply' % (out_dir, data_idx), hdr, 'avatar.'))
        if w_recon and save_final_mesh and save.get('recon_live_v') is not None:      # main.py:495-498
            hdr, rec = obj_io.ply_records_device(save['recon_live_v'], save['recon_f'], save['recon_live_vn'], save.get('recon_live_vc'))
            out_t.update({'recon.' + k_: v for k_, v in rec.items()})
            plys.append(('%s/%04d_recon.ply' % (out_dir, data_idx), hdr, 'recon.'))
        npz_keys = [k_ for k_ in save if save[k_] is not None] if save_npz else []

        def write(arrays, data_idx=data_idx, plys=plys, npz_keys=npz_keys):
            n = 0
            if npz_keys:
                path = os.path.join(out_dir, '%04d_mesh.npz' % data_idx)
                np.savez(path, **{k_: arrays[k_] for k_ in npz_keys})
                n += os.path.getsize(path)
            for path, hdr, prefix in plys:
                obj_io.write_ply_records(path, hdr, arrays, prefix)
                n += os.path.getsize(path)
            with lock:
                clock['bytes_written'] += n
                clock['files'] += len(plys) + (1 if npz_keys else 0)

        if out_t:
            writer.submit(out_t, write, tag=i)
            if sync_io:
                writer.drain()                                # the reference's shape: .cpu() and the file writes inside the frame
        clock['out_host_s'] = clock.get('out_host_s', 0.0) + time.perf_counter() - t_out

    # --gather-meshes: the one collective of the throughput mode: every rank's finished avatar meshes, exchanged step by step
    # WHILE the following frames compute (parallel.MeshExchange: exact sizes, asynchronous), in batches of `gather_batch` steps so that a long
    # sequence does not pile every mesh of every rank up in HBM: after a batch rank 0 moves it to host memory and the device copies are dropped.
    # A failed or unattempted frame travels as an empty mesh, so the ranks' steps stay aligned.
    gathered = {}                                            # rank 0: frame -> {'v', 'vn', 'f'} numpy
    gather = {'ex': None, 'step': 0}
    steps = (len(frames) + world - 1) // world

    def empty_mesh():
        return {'v': torch.zeros((0, 3), device=dev), 'vn': torch.zeros((0, 3), device=dev), 'f': torch.zeros((0, 3), dtype=torch.int32, device=dev)}

    def submit_step(mesh):
        k = gather['step']
        b0 = (k // gather_batch) * gather_batch               # first step of this batch
        if gather['ex'] is None:
            lo, hi = b0 * world, min(len(frames), (b0 + gather_batch) * world)
            gather['ex'] = (parallel.MeshExchange(hi - lo, device=dev, force=force_dist), lo, hi)
            pipe.exchange = gather['ex'][0]                   # avatar_frame pumps it: step k - 1 travels while frame k computes
        ex, lo, hi = gather['ex']
        has_frame = k * world + rank < len(frames)
        ex.submit((mesh or empty_mesh()) if has_frame else None)
        gather['step'] = k + 1
        if k + 1 == min(steps, b0 + gather_batch):            # the batch is complete: collect it, keep nothing on the device
            pipe.exchange = None
            for fr, m in zip(frames[lo:hi], ex.finish()):
                if rank == 0:
                    gathered[fr] = {key: m[key].cpu().numpy() for key in ('v', 'vn', 'f')}
            gather['ex'] = None                               # drops the exchange and with it every device copy of the batch (own and received)

    def process_and_submit(k, fr, nxt):
        try:
            mesh = process(k, fr, nxt)
        except Exception as e:      # noqa: BLE001 -- re-raised below; run_sharded contains it
            if parallel.device_is_gone(e):
                # a sticky HIP fault or an out-of-memory error: no collective can be issued on this device any more.  The rank ends the JOB (below) --
                # under torch.distributed.run the launcher then stops its peers -- instead of leaving them in a collective until the watchdog's timeout.
                gather['dead'] = f'{type(e).__name__}: {e}'
            elif gather_meshes:
                submit_step(None)                             # the failed frame travels as an empty mesh: the ranks' steps stay aligned
            raise
        if gather_meshes:
            submit_step(mesh)
            # what stays in run_sharded's `results` is host data: the device tensors belong to the exchange and go with its batch
            return None if mesh is None else (int(mesh['v'].shape[0]), int(mesh['f'].shape[0]))
        return mesh

    summary = parallel.run_sharded(frames, process_and_submit, rank, world, log, max_consecutive_failures=max_failure_streak)
    if hasattr(pipe, 'run_after_query') and not gather.get('dead'):
        pipe.run_after_query()                                 # the last frame's outputs (nothing follows in whose shadow they could run)
    t_enqueued = time.perf_counter()
    if dev is not None and not gather.get('dead'):
        torch.cuda.synchronize(dev)
    t_device = time.perf_counter()
    for fr, why in deferred_failures + writer.close():         # a frame whose files could not be written has failed
        if fr in summary['done']:
            summary['done'].remove(fr)
        summary['failed'].append((fr, 'output: ' + why))
        log(f'# rank {rank}: frame {fr} FAILED while its files were written -- {why}')
    t_written = time.perf_counter()
    prefetch.close()
    if timing is not None:
        # what the loop measured about itself: the first frames pay for first-use work (launch plans, hipGraph capture, pinned allocations), so the steady
        # figure starts at frame `skip`; `e2e` ends when the last file is on its way to the disk (closed by the writer threads), `device` when the
        # last kernel has finished
        n, st = len(clock['start']), clock['start']
        skip = min(2, max(0, n - 1))
        timing.update({
            'frames': n, 'frames_failed': len(summary['failed']), 'skipped_first': skip, 'sync_io': bool(sync_io),
            'outputs': {'npz': bool(save_npz), 'ply_avatar': bool(save_avatar_mesh), 'ply_recon': bool(save_final_mesh and w_recon)},
            'e2e_ms_per_frame': (t_written - st[skip]) / (n - skip) * 1e3 if n > skip else None,
            'device_ms_per_frame': (t_device - st[skip]) / (n - skip) * 1e3 if n > skip else None,
            'host_enqueue_ms_per_frame': (t_enqueued - st[skip]) / (n - skip) * 1e3 if n > skip else None,
            'writer_tail_ms': (t_written - t_device) * 1e3, 'first_frame_ms': (st[1] - st[0]) * 1e3 if n > 1 else None,
            'bytes_written': clock['bytes_written'], 'files_written': clock['files'], 'd2h_bytes': writer.d2h_bytes,
            'h2d_copies': prefetch.h2d_copies, 'h2d_bytes': prefetch.h2d_bytes, 'waited_for_writer_slot_ms': writer.waited_for_slot_s * 1e3,
            'io_threads': io_threads, 'io_slots': io_slots,
            'output_section_host_ms_per_frame': clock.get('out_host_s', 0.0) / max(1, n) * 1e3, 'output_dir': out_dir, 'vol_res': None if dry_run else list(cfg['testing']['vol_res'])})
    if gather.get('dead') and world > 1:
        log('# rank %d: device lost (%s) -- leaving the job without touching the process group; the launcher stops the other ranks' % (rank, gather['dead']))
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(3)
    if gather_meshes and not gather.get('dead'):
        while gather['step'] < steps:                          # frames this rank never attempted (abort) or does not own (last, partial step)
            submit_step(None)
    pipe.exchange = None
    summary['results'] = {}
    everyone = parallel.gather_summaries(summary)
    n_failed = sum(len(s_['failed']) for s_ in everyone)
    if gather_meshes and rank == 0:
        np.savez(os.path.join(out_dir, 'all_avatar_meshes.npz'), frames=np.asarray(frames, np.int64),
                 **{'%s_%04d' % (key, fr): gathered[fr][key] for fr in frames for key in ('v', 'vn', 'f')})
        log('# gathered %d meshes (%d vertices) over %d rank(s)' % (len(gathered), sum(int(m['v'].shape[0]) for m in gathered.values()), world))
    if rank == 0:
        log('# %d of %d frames done on %d rank(s)%s' % (sum(len(s_['done']) for s_ in everyone), len(frames), world,
            '' if not n_failed else '; FAILED: ' + ', '.join('%s (%s)' % (fr, why) for s_ in everyone for fr, why in s_['failed'])))
    return n_failed


def main(argv=None):
    torch.manual_seed(31359)
    np.random.seed(31359)

    arg_parser = ArgumentParser()
    arg_parser.add_argument('-c', '--config_path', type=str, help='Configuration file path.')
    arg_parser.add_argument('-m', '--mode', type=str, default='test', choices=['train', 'test'], help='Train or test.')
    arg_parser.add_argument('--synthetic', action='store_true', help='synthetic body / weights instead of the licensed data')
    arg_parser.add_argument('--frames', type=int, default=2)
    arg_parser.add_argument('--valid', type=str, default='band', choices=['band', 'dense'])
    arg_parser.add_argument('--save-ply', action='store_true', help='write the live avatar / recon meshes as PLY (obj_io layout)')
    arg_parser.add_argument('--no-npz', action='store_true', help='do not write <idx>_mesh.npz (every mesh tensor of the frame)')
    arg_parser.add_argument('--sync-io', action='store_true',
                            help="the reference's loop shape: blocking upload, .cpu(), files written inside the frame (for comparison)")
    arg_parser.add_argument('--io-threads', type=int, default=3, help='writer threads behind the loop')
    arg_parser.add_argument('--io-slots', type=int, default=4, help='finished frames that may wait for the disk (pinned slots) before the loop does')
    arg_parser.add_argument('--timing-json', type=str, default=None, help='write what the frame loop measured about itself (rank 0) to this file')
    arg_parser.add_argument('--nerf', action='store_true', help='also evaluate vertex colours (w_nerf)')
    arg_parser.add_argument('--integrate', type=str, default='merge', choices=['merge', 'cover'], help='normal fusion manner (main.py:281)')
    arg_parser.add_argument('--gpus', type=int, default=0, help='shard the frames over this many GPUs of the node (one process each); 0: whatever launched us')
    arg_parser.add_argument('--gather-meshes', action='store_true', help='all-gather the live avatar meshes of the batch (RCCL) and write them on rank 0')
    arg_parser.add_argument('--gather-batch', type=int, default=8, help='--gather-meshes: steps (frames per rank) exchanged and moved to the host at a time')
    arg_parser.add_argument('--max-failure-streak', type=int, default=3,
                            help='give up on a rank\'s remaining frames after this many consecutive failures OF THE SAME KIND (0: never)')
    arg_parser.add_argument('--check-range', action='store_true',
                            help='run EVERY frame with the fp16 range check of the fused kernels (default: until one frame has passed it)')
    arg_parser.add_argument('--dist-timeout', type=float, default=180.0, help='seconds a rank may take to show up at the rendezvous')
    arg_parser.add_argument('--collective-timeout', type=float, default=None,
                            help='seconds a collective may wait for the slowest rank AFTER the rendezvous (default: 10 x --dist-timeout, at least 1800)')
    arg_parser.add_argument('--output-dir', type=str, default=None, help='overrides testing.output_dir of the yaml')
    arg_parser.add_argument('--dry-run', action='store_true', help='no GPU: stand-in meshes on gloo (launcher, sharding, error containment, gather)')
    arg_parser.add_argument('--dry-fail', type=int, nargs='*', default=[], help='--dry-run: frames whose processing raises')
    args = arg_parser.parse_args(argv)

    from avatarcap_amd import config, parallel
    if args.mode == 'train':
        raise SystemExit('-m train is out of scope for the MI355X hot-path build (SURVEY.md section 2, row 12)')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:                         # no launcher around us: start the ranks ourselves
        raise SystemExit(parallel.self_launch(__file__, sys.argv[1:] if argv is None else list(argv), args.gpus))
    world, rank, local_rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus and world != args.gpus:
        raise SystemExit(f'main.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    config.cfg = config.load_config(args.config_path) if args.config_path else config.default_cfg()
    if args.output_dir:
        config.cfg['testing']['output_dir'] = args.output_dir
    if args.check_range:
        config.check_range = True
    if args.dry_run:
        config.device = torch.device('cpu')
    else:
        if not torch.cuda.is_available():
            raise SystemExit('main.py -m test needs an MI355X: no HIP device visible (there is no CPU fallback for the hot path)')
        torch.cuda.set_device(local_rank)
        config.device = torch.device('cuda', local_rank)
    # AVC_FORCE_DIST=1 (tests): one rank, but the process group is RCCL and --gather-meshes runs the exchange's collectives -- the N > 1 code path on one GPU
    force_dist = world == 1 and not args.dry_run and os.environ.get('AVC_FORCE_DIST') == '1'
    if force_dist:
        os.environ.setdefault('MASTER_PORT', str(parallel.free_port()))
    if world > 1 or force_dist:
        parallel.init_process_group('gloo' if args.dry_run else 'nccl', rank, world, None if args.dry_run else config.device,
                                    timeout_s=args.dist_timeout, collective_timeout_s=args.collective_timeout)
        if not args.dry_run:
            parallel.pin_to_gpu_numa(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)), log=print if rank == 0 else None)
            if args.gather_meshes:
                parallel.leave_cus_for_the_exchange(config.device, log=print if rank == 0 else None)
    timing = {} if args.timing_json else None
    try:
        n_failed = run_avatarcap(w_recon=True, save_avatar_mesh=args.save_ply, save_final_mesh=args.save_ply, w_nerf=args.nerf,
                                 synthetic=args.synthetic, n_frames=args.frames, valid=args.valid, integrate_manner=args.integrate,
                                 rank=rank, world=world, gather_meshes=args.gather_meshes, gather_batch=max(1, args.gather_batch), dry_run=args.dry_run,
                                 dry_fail=args.dry_fail, max_failure_streak=args.max_failure_streak, force_dist=force_dist,
                                 save_npz=not args.no_npz, sync_io=args.sync_io, io_threads=max(1, args.io_threads), io_slots=max(1, args.io_slots),
                                 timing=timing)
        if timing is not None and rank == 0:
            import json
            timing.update({'world': world, 'argv': list(sys.argv[1:] if argv is None else argv)})
            with open(args.timing_json, 'w') as fh:
                json.dump(timing, fh)
    finally:
        if world > 1 or force_dist:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
    return 1 if n_failed else 0


if __name__ == '__main__':
    raise SystemExit(main())
