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

With a captured sequence the loader is avatarcap_amd.avatarcap_dataset.AvatarCapDataset (the reference's dataset in test mode): it needs the
sequence directory, the checkpoints named in the yaml and the licensed SMPL model file under smpl_files/ (or $AVC_SMPL_DIR) -- none of which
can be redistributed; a missing file raises the FileNotFoundError the reference raises.  `-m train` is out of scope.
"""
import os
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


def run_avatarcap(w_recon=True, save_avatar_mesh=False, save_final_mesh=False, w_nerf=False, frame_idx=None, view_idx=0, interval=1,
                  synthetic=False, n_frames=2, valid='band', integrate_manner='merge'):
    from avatarcap_amd import config, synthetic as syn
    from avatarcap_amd.dataset import SyntheticTestDataset, to_cuda
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar, NerfRenderer
    from avatarcap_amd.network.arch_recon import ReconNetwork
    from avatarcap_amd.pipeline import FramePipeline
    from avatarcap_amd.utils import obj_io
    cfg = config.cfg
    out_dir = cfg['testing']['output_dir']
    os.makedirs(out_dir, exist_ok=True)

    nerf_net = None
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
    if frame_idx is None:                                                        # main.py:337-345
        frames = list(range(0, data_num, interval))
    elif isinstance(frame_idx, int):
        frames = [frame_idx - start_data_idx]
    elif isinstance(frame_idx, list):
        frames = (np.array(frame_idx, np.int32) - start_data_idx).tolist()
    else:
        raise TypeError('Invalid frame_idx!')

    load = lambda i: to_cuda(ds[i * img_num_per_pose + view_idx], add_batch=True)     # main.py:349-351
    nxt = load(frames[0]) if frames else None
    for n, i in enumerate(frames):
        items, nxt = nxt, (load(frames[n + 1]) if n + 1 < len(frames) else None)  # one frame of look-ahead: its U-Net is queued behind this frame's query
        data_idx = int(items['data_idx'])
        a = pipe.avatar_frame(items, next_items=nxt)                              # step 1
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
            save.update({'recon_' + k: v for k, v in r.items() if k != 'occ_volume'})
        if w_nerf:                                                                # step 4 (main.py:464-477)
            save['live_vc'] = pipe.colour_vertices(items, a['cano_v'], a['cano_vn'], renderer)
            if w_recon and save['recon_cano_v'].shape[0] > 0 and a['cano_v'].shape[0] > 0:        # main.py:478-482
                save['recon_live_vc'] = pipe.transfer_colours(save['recon_cano_v'], a['cano_v'], save['live_vc'])
        if save_avatar_mesh and a.get('live_v') is not None:                          # main.py:491-493
            obj_io.save_mesh_as_ply('%s/%04d_avatar.ply' % (out_dir, data_idx), a['live_v'].cpu().numpy(), a['f'].cpu().numpy(),
                                    a['live_vn'].cpu().numpy(), save['live_vc'].cpu().numpy() if w_nerf else None)
        if w_recon and save_final_mesh and 'recon_live_v' in save:                     # main.py:495-498
            obj_io.save_mesh_as_ply('%s/%04d_recon.ply' % (out_dir, data_idx), save['recon_live_v'].cpu().numpy(),
                                    save['recon_f'].cpu().numpy(), save['recon_live_vn'].cpu().numpy(),
                                    save['recon_live_vc'].cpu().numpy() if 'recon_live_vc' in save else None)
        np.savez(os.path.join(out_dir, '%04d_mesh.npz' % data_idx),
                 **{k: v.cpu().numpy() for k, v in save.items() if v is not None})
        print('# frame %d (data idx %d): avatar %d verts / %d faces%s' % (i, data_idx, a['cano_v'].shape[0], a['f'].shape[0],
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
