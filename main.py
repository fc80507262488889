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


def _observed_normals(ds, data_idx, view_idx, device):
    """The image-observed normal map of step 2 (main.py:406-411): cv.imread(<...>.exr, IMREAD_UNCHANGED) -> (H, W, 3) on the device."""
    from avatarcap_amd.utils.exr_io import read_exr
    if ds.data_config['data_type'] == 'synthetic':
        path = ds.data_dir + '/imgs/%03d/normal_view_%03d.exr' % (data_idx, view_idx)
    elif ds.data_config['data_type'] == 'real':
        path = ds.data_dir + '/imgs/normal/normal_%04d.exr' % data_idx
    else:
        raise ValueError('Invalid data type!')
    return torch.from_numpy(np.ascontiguousarray(read_exr(path)[..., :3], np.float32)).to(device)


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
                  dry_run=False, dry_fail=(), max_failure_streak=3, force_dist=False):
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
        load = lambda i: {'data_idx': i}                                          # noqa: E731
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
        load = lambda i: to_cuda(ds[i * img_num_per_pose + view_idx], add_batch=True)     # main.py:349-351   # noqa: E731

    if frame_idx is None:                                                        # main.py:337-345
        frames = list(range(0, data_num, interval))
    elif isinstance(frame_idx, int):
        frames = [frame_idx - start_data_idx]
    elif isinstance(frame_idx, list):
        frames = (np.array(frame_idx, np.int32) - start_data_idx).tolist()
    else:
        raise TypeError('Invalid frame_idx!')

    loaded = {}                                                                  # one frame of look-ahead: items of the frame this rank runs next

    def items_of(i):
        if i not in loaded:
            loaded.clear()
            loaded[i] = load(i)
        return loaded[i]

    # Weights read from disk have never been through the kernels: the split-fp16 arithmetic of the fused queries is exact only while every feature
    # and activation stays below 65504 (include/avcap.h, "numeric range"), and an overflow is SILENT (a ReLU swallows the NaN).  The first frame
    # of every rank therefore runs with the range check on (avc_set_range_check: the checked flavour of the kernels, one synchronisation per
    # query); a trip is fatal for the whole run (parallel.run_sharded: AVC_ERR_RANGE).  Synthetic weights are generated inside the range.
    # The user's own setting (config.check_range) is kept; the forced check stays on until one frame has COMPLETED under it (a first frame that fails
    # on a missing file has not checked anything).
    check = {'user': bool(getattr(config, 'check_range', False)), 'pending': not (synthetic or dry_run)}

    def process(k, i, nxt_i):
        checking = check['user'] or check['pending']
        config.check_range = checking
        items = items_of(i)
        nxt = None
        if nxt_i is not None:
            try:
                nxt = load(nxt_i)           # its U-Net is queued behind this frame's query (FramePipeline.avatar_frame)
            except Exception:               # noqa: BLE001 -- the next frame's own turn will report what is wrong with it
                nxt = None
        data_idx = int(items['data_idx'])
        a = pipe.avatar_frame(items, next_items=nxt)                              # step 1
        loaded.clear()
        if nxt is not None:
            loaded[nxt_i] = nxt
        save = {'cano_v': a['cano_v'], 'cano_vn': a['cano_vn'], 'f': a['f'], 'live_v': a.get('live_v'), 'live_vn': a.get('live_vn')}
        if w_recon:
            # step 2: canonical normal fusion (main.py:405-433)
            if a['cano_v'].shape[0] > 0:
                if synthetic:       # the captured image's normal map is synthesised (dataset.synthetic_observed_normals)
                    from avatarcap_amd.dataset import synthetic_camera, synthetic_observed_normals
                    w2c, cam = synthetic_camera()
                    observed = synthetic_observed_normals(a['live_v'], a['live_vn'], a['f'], w2c, cam, seed=i)
                else:
                    cam = ds.data_config['camera']
                    w2c = items['w2c_RT'][0].cpu().numpy() if isinstance(items['w2c_RT'], torch.Tensor) else np.asarray(items['w2c_RT'], np.float32)
                    observed = _observed_normals(ds, data_idx, view_idx, config.device)
                items['front_normal'], items['back_normal'], _ = pipe.fuse_normals(a, observed, w2c, cam, integrate_manner)
            else:
                items['front_normal'], items['back_normal'] = pipe.cano_normal_maps(a['cano_v'], a['cano_vn'], a['f'])
            r = pipe.recon_frame(items)                                           # step 3
            save.update({'recon_' + k_: v for k_, v in r.items() if k_ != 'occ_volume'})
        if w_nerf:                                                                # step 4 (main.py:464-477)
            save['live_vc'] = pipe.colour_vertices(items, a['cano_v'], a['cano_vn'], renderer)
            if w_recon and save['recon_cano_v'].shape[0] > 0 and a['cano_v'].shape[0] > 0:        # main.py:478-482
                save['recon_live_vc'] = pipe.transfer_colours(save['recon_cano_v'], a['cano_v'], save['live_vc'])
        if save_avatar_mesh and a.get('live_v') is not None:                          # main.py:491-493
            from avatarcap_amd.utils import obj_io
            obj_io.save_mesh_as_ply('%s/%04d_avatar.ply' % (out_dir, data_idx), a['live_v'].cpu().numpy(), a['f'].cpu().numpy(),
                                    a['live_vn'].cpu().numpy(), save['live_vc'].cpu().numpy() if w_nerf else None)
        if w_recon and save_final_mesh and 'recon_live_v' in save:                     # main.py:495-498
            from avatarcap_amd.utils import obj_io
            obj_io.save_mesh_as_ply('%s/%04d_recon.ply' % (out_dir, data_idx), save['recon_live_v'].cpu().numpy(),
                                    save['recon_f'].cpu().numpy(), save['recon_live_vn'].cpu().numpy(),
                                    save['recon_live_vc'].cpu().numpy() if 'recon_live_vc' in save else None)
        np.savez(os.path.join(out_dir, '%04d_mesh.npz' % data_idx),
                 **{k_: v.cpu().numpy() for k_, v in save.items() if v is not None})
        log('# %sframe %d (data idx %d): avatar %d verts / %d faces%s' % ('rank %d: ' % rank if world > 1 else '', i, data_idx, a['cano_v'].shape[0],
            a['f'].shape[0], (', recon %d verts' % save['recon_cano_v'].shape[0]) if w_recon else ''))
        if checking:
            check['pending'] = False              # this frame went through every kernel with the range check on
        if gather_meshes and a.get('live_v') is not None:
            return {'v': a['live_v'], 'vn': a['live_vn'], 'f': a['f']}
        return None

    # --gather-meshes: the one collective of the throughput mode: every rank's finished avatar meshes, exchanged step by step
    # WHILE the following frames compute (parallel.MeshExchange: exact sizes, asynchronous), in batches of `gather_batch` steps so that a long
    # sequence does not pile every mesh of every rank up in HBM: after a batch rank 0 moves it to host memory and the device copies are dropped.
    # A failed or unattempted frame travels as an empty mesh, so the ranks' steps stay aligned.
    dev = None if dry_run else config.device
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
    arg_parser.add_argument('--nerf', action='store_true', help='also evaluate vertex colours (w_nerf)')
    arg_parser.add_argument('--integrate', type=str, default='merge', choices=['merge', 'cover'], help='normal fusion manner (main.py:281)')
    arg_parser.add_argument('--gpus', type=int, default=0, help='shard the frames over this many GPUs of the node (one process each); 0: whatever launched us')
    arg_parser.add_argument('--gather-meshes', action='store_true', help='all-gather the live avatar meshes of the batch (RCCL) and write them on rank 0')
    arg_parser.add_argument('--gather-batch', type=int, default=8, help='--gather-meshes: steps (frames per rank) exchanged and moved to the host at a time')
    arg_parser.add_argument('--max-failure-streak', type=int, default=3,
                            help='give up on a rank\'s remaining frames after this many consecutive failures OF THE SAME KIND (0: never)')
    arg_parser.add_argument('--check-range', action='store_true', help='run EVERY frame with the fp16 range check of the fused kernels (default: until one frame has passed it)')
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
    try:
        n_failed = run_avatarcap(w_recon=True, save_avatar_mesh=args.save_ply, save_final_mesh=args.save_ply, w_nerf=args.nerf,
                                 synthetic=args.synthetic, n_frames=args.frames, valid=args.valid, integrate_manner=args.integrate,
                                 rank=rank, world=world, gather_meshes=args.gather_meshes, gather_batch=max(1, args.gather_batch), dry_run=args.dry_run,
                                 dry_fail=args.dry_fail, max_failure_streak=args.max_failure_streak, force_dist=force_dist)
    finally:
        if world > 1 or force_dist:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
    return 1 if n_failed else 0


if __name__ == '__main__':
    raise SystemExit(main())
