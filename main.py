#!/usr/bin/env python3
"""Test-mode entry point with the reference's CLI and config surface (main.py:507-529):

    python main.py -c configs/example.yaml -m test [--synthetic --frames 2]

`-m test` runs steps 1 and 3 of `run_avatarcap`'s frame loop (main.py:348-453) -- canonical avatar
geometry, skinning to the live pose, and (with a reconstruction checkpoint) the image-conditioned
reconstruction -- through avatarcap_amd.pipeline on the HIP device.  Rendering (OpenGL), image I/O
and normal fusion are outside the ported path: meshes are written as .npz.

The captured dataset, SMPL model and checkpoints of the reference are not redistributable; with
`--synthetic` the synthetic body / poses / seeded weights of avatarcap_amd.synthetic stand in
(what bench.py and the tests use).  `-m train` is out of scope.
"""
import os
from argparse import ArgumentParser

import numpy as np
import torch


def run_avatarcap(w_recon=True, save_avatar_mesh=False, save_final_mesh=False, w_nerf=False, frame_idx=None, interval=1,
                  synthetic=False, n_frames=2, valid='band', integrate_manner='merge'):
    from avatarcap_amd import config, synthetic as syn
    from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar
    from avatarcap_amd.network.arch_recon import ReconNetwork
    from avatarcap_amd.pipeline import FramePipeline
    cfg = config.cfg
    out_dir = cfg['testing']['output_dir']
    os.makedirs(out_dir, exist_ok=True)

    if synthetic:
        network = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to(config.device).eval()
        syn.load_synth(network, syn.SEED)
        recon_net = ReconNetwork().to(config.device).eval()
        syn.load_synth(recon_net, syn.SEED)
    else:
        network = GeoTexAvatar().to(config.device).eval()                       # main.py:296-297
        if cfg['testing']['net_ckpt'] is not None:
            print('# Loading GeoTexAvatar network from %s' % cfg['testing']['net_ckpt'])
            network.load_state_dict(torch.load(cfg['testing']['net_ckpt'] + '/net.pt')['network'])     # :302-305
        recon_net = ReconNetwork().to(config.device).eval()
        if cfg['testing'].get('recon_net_ckpt') is not None:
            print('# Loading reconstruction network from %s' % cfg['testing']['recon_net_ckpt'])
            recon_net.load_state_dict(torch.load(cfg['testing']['recon_net_ckpt'] + '/recon_net.pt')['network'])   # :316-320
        raise SystemExit('captured-sequence loading needs the licensed SMPL model and dataset of the reference; '
                         'run with --synthetic (see the module docstring)')

    ds = SyntheticTestDataset(cfg['testing']['vol_res'], valid=valid, n_frames=n_frames)
    pipe = FramePipeline(network, ds, recon_net)
    frames = list(range(0, len(ds), interval)) if frame_idx is None else ([frame_idx] if isinstance(frame_idx, int) else list(frame_idx))
    for i in frames:
        items = to_cuda(ds[i], add_batch=True)                                    # main.py:350-351
        a = pipe.avatar_frame(items)                                              # step 1
        save = {'cano_v': a['cano_v'], 'cano_vn': a['cano_vn'], 'f': a['f'], 'live_v': a.get('live_v'), 'live_vn': a.get('live_vn')}
        if w_recon:
            # step 2: canonical normal fusion (main.py:405-429).  The captured image's normal map is synthesised here
            # (dataset.synthetic_observed_normals); with a real sequence it is read from normal_%04d.exr
            if a['cano_v'].shape[0] > 0:
                from avatarcap_amd.dataset import synthetic_camera, synthetic_observed_normals
                w2c, cam = synthetic_camera()
                observed = synthetic_observed_normals(a['live_v'], a['live_vn'], a['f'], w2c, cam, seed=i)
                items['front_normal'], items['back_normal'], _ = pipe.fuse_normals(a, observed, w2c, cam, integrate_manner)
            else:
                items['front_normal'], items['back_normal'] = pipe.cano_normal_maps(a['cano_v'], a['cano_vn'], a['f'])
            r = pipe.recon_frame(items)                                           # step 3
            save.update({'recon_' + k: v for k, v in r.items() if k != 'occ_volume'})
        if w_nerf:                                                                # step 4 (main.py:464-477)
            save['live_vc'] = pipe.colour_vertices(items, a['cano_v'], a['cano_vn'])
            if w_recon and save['recon_cano_v'].shape[0] > 0 and a['cano_v'].shape[0] > 0:        # main.py:478-482
                save['recon_live_vc'] = pipe.transfer_colours(save['recon_cano_v'], a['cano_v'], save['live_vc'])
        from avatarcap_amd.utils import obj_io
        if save_avatar_mesh and a.get('live_v') is not None:                          # main.py:491-493
            obj_io.save_mesh_as_ply('%s/%04d_avatar.ply' % (out_dir, items['data_idx']), a['live_v'].cpu().numpy(), a['f'].cpu().numpy(),
                                    a['live_vn'].cpu().numpy(), save['live_vc'].cpu().numpy() if w_nerf else None)
        if w_recon and save_final_mesh and 'recon_live_v' in save:                     # main.py:495-498
            obj_io.save_mesh_as_ply('%s/%04d_recon.ply' % (out_dir, items['data_idx']), save['recon_live_v'].cpu().numpy(),
                                    save['recon_f'].cpu().numpy(), save['recon_live_vn'].cpu().numpy(),
                                    save['recon_live_vc'].cpu().numpy() if 'recon_live_vc' in save else None)
        np.savez(os.path.join(out_dir, '%04d_mesh.npz' % items['data_idx']),
                 **{k: v.cpu().numpy() for k, v in save.items() if v is not None})
        print('# frame %d: avatar %d verts / %d faces%s' % (i, a['cano_v'].shape[0], a['f'].shape[0],
              (', recon %d verts' % save['recon_cano_v'].shape[0]) if w_recon else ''))


if __name__ == '__main__':
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
    args = arg_parser.parse_args()

    from avatarcap_amd import config
    config.cfg = config.load_config(args.config_path) if args.config_path else config.default_cfg()
    if args.mode == 'train':
        raise SystemExit('-m train is out of scope for the MI355X hot-path build (SURVEY.md section 2, row 12)')
    run_avatarcap(w_recon=True, save_avatar_mesh=args.save_ply, save_final_mesh=args.save_ply, w_nerf=args.nerf,
                  synthetic=args.synthetic, n_frames=args.frames, valid=args.valid, integrate_manner=args.integrate)
