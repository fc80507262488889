"""`python main.py -c <yaml> -m test` on a CAPTURED-SEQUENCE layout (no --synthetic): dataConfig.yaml, SMPL model file, pose / shape / position-map
files, checkpoints and image-normal EXRs are all read from disk through the reference-shaped loader (avatarcap_amd.avatarcap_dataset), steps 1-4
of the frame loop run on the device, PLYs are written (main.py:275-498).  The files are synthetic stand-ins of the right layout
(tests/synthetic_sequence.py): the licensed model / data / checkpoints cannot ship."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

import synthetic_sequence as sq
from avatarcap_amd import config, synthetic as syn
from test_host import _write_exr

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def test_main_test_mode_on_a_sequence_directory(tmp_path):
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar
    from avatarcap_amd.network.arch_recon import ReconNetwork
    data, train, out = tmp_path / 'testing', tmp_path / 'training', tmp_path / 'out'
    smpl_dir = tmp_path / 'smpl_files'
    for d in (data, train, smpl_dir):
        os.makedirs(d)
    sq.write_smpl_file(str(smpl_dir / 'basicmodel_M_lbs_10_207_0_v1.0.0.pkl'))
    ids = sq.build_sequence(str(data), lambda p, img: _write_exr(p, img[..., ::-1].copy(), 'RGB', 2, 3), n_frames=2, start=7, data_type='real',
                            pos_map_res=256, pos_map_src=(192, 384))     # the 7-level U-Net needs its real 256^2 input
    os.makedirs(data / 'imgs' / 'normal')
    for idx in ids:                                                   # the image-observed normal maps of step 2 (main.py:409)
        nm = syn.smooth_normal_maps(40 + idx, 512)[:3].transpose(1, 2, 0)
        nm[:, :140] = 0                                               # part of the image shows no body
        _write_exr(str(data / 'imgs' / 'normal' / ('normal_%04d.exr' % idx)), np.ascontiguousarray(nm[..., ::-1]), 'RGB', 1, 3)
    np.save(str(train / 'cano_base_blend_weight_volume.npy'), np.full((4, 4, 4, 24), 1 / 24, np.float32))
    config.cfg = config.default_cfg()
    config.cfg['training']['training_data_dir'] = str(train)
    net = GeoTexAvatar()
    rn = ReconNetwork()
    for name, m, fn in (('avatar', net, 'net.pt'), ('recon', rn, 'recon_net.pt')):
        os.makedirs(tmp_path / name)
        sd = syn.synth_state_dict(syn.module_shapes(m), syn.SEED)
        torch.save({'network': {k: torch.from_numpy(v) for k, v in sd.items()}}, str(tmp_path / name / fn))
    cfg = {'training': {'training_data_dir': str(train)},
           'testing': {'vol_res': [40, 96, 36], 'recon_net_ckpt': str(tmp_path / 'recon'), 'net_ckpt': str(tmp_path / 'avatar'),
                       'net_ckpt_finetuned': None, 'testing_data_dir': str(data), 'output_dir': str(out)},
           'model': {'cano_template': {'pos_encoding': 10}, 'warping_field': {'pos_encoding': 0}}}
    with open(tmp_path / 'cfg.yaml', 'w') as fh:
        yaml.safe_dump(cfg, fh)
    env = dict(os.environ, AVC_SMPL_DIR=str(smpl_dir))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'main.py'), '-c', str(tmp_path / 'cfg.yaml'), '-m', 'test', '--save-ply', '--nerf', '--integrate', 'cover'],
                       capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert '# Real data' in r.stdout and '# Start data index: 7' in r.stdout and '# Data num: 2' in r.stdout
    for idx in ids:
        m = np.load(str(out / ('%04d_mesh.npz' % idx)))
        assert m['cano_v'].shape[0] > 100 and m['f'].max() < m['cano_v'].shape[0] and m['live_v'].shape == m['cano_v'].shape
        assert m['recon_cano_v'].shape[0] > 0 and m['live_vc'].shape == m['cano_v'].shape
        ply = open(str(out / ('%04d_avatar.ply' % idx)), 'rb').read()
        assert ply.startswith(b'ply\n') and (b'element vertex %d' % m['cano_v'].shape[0]) in ply[:400]
        assert os.path.exists(str(out / ('%04d_recon.ply' % idx)))
    # a missing model file is the reference's FileNotFoundError, not a silent fallback
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'main.py'), '-c', str(tmp_path / 'cfg.yaml'), '-m', 'test'],
                        capture_output=True, text=True, env=dict(os.environ, AVC_SMPL_DIR=str(tmp_path / 'nowhere')), timeout=600, cwd=ROOT)
    assert r2.returncode != 0 and 'FileNotFoundError' in r2.stderr


def test_main_refuses_a_checkpoint_that_leaves_the_fp16_range(tmp_path):
    """Weights read from disk have never been through the kernels: `main.py -m test` runs the first frame of every rank with the range
    check on, and a net.pt whose activations overflow the split-fp16 arithmetic ends the run non-zero with AVC_ERR_RANGE instead of writing meshes
    of silently wrong values (a ReLU would have swallowed the NaN)."""
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar
    from avatarcap_amd.network.arch_recon import ReconNetwork
    data, train, out = tmp_path / 'testing', tmp_path / 'training', tmp_path / 'out'
    smpl_dir = tmp_path / 'smpl_files'
    for d in (data, train, smpl_dir):
        os.makedirs(d)
    sq.write_smpl_file(str(smpl_dir / 'basicmodel_M_lbs_10_207_0_v1.0.0.pkl'))
    ids = sq.build_sequence(str(data), lambda p, img: _write_exr(p, img[..., ::-1].copy(), 'RGB', 2, 3), n_frames=3, start=0, data_type='real',
                            pos_map_res=256, pos_map_src=(192, 384))
    os.makedirs(data / 'imgs' / 'normal')
    for idx in ids:
        nm = syn.smooth_normal_maps(40 + idx, 512)[:3].transpose(1, 2, 0)
        _write_exr(str(data / 'imgs' / 'normal' / ('normal_%04d.exr' % idx)), np.ascontiguousarray(nm[..., ::-1]), 'RGB', 1, 3)
    np.save(str(train / 'cano_base_blend_weight_volume.npy'), np.full((4, 4, 4, 24), 1 / 24, np.float32))
    config.cfg = config.default_cfg()
    config.cfg['training']['training_data_dir'] = str(train)
    for name, m, fn in (('avatar', GeoTexAvatar(), 'net.pt'), ('recon', ReconNetwork(), 'recon_net.pt')):
        os.makedirs(tmp_path / name)
        sd = syn.synth_state_dict(syn.module_shapes(m), syn.SEED)
        if name == 'avatar':                                                  # a hidden layer of the template 1e5 x too large (weights ~1.5e4: the packer takes them): its activations pass 65504
            sd['cano_template.shared_mlp.fc_list.2.0.weight'] = sd['cano_template.shared_mlp.fc_list.2.0.weight'] * 1.0e5
        torch.save({'network': {k: torch.from_numpy(v) for k, v in sd.items()}}, str(tmp_path / name / fn))
    cfg = {'training': {'training_data_dir': str(train)},
           'testing': {'vol_res': [40, 96, 36], 'recon_net_ckpt': str(tmp_path / 'recon'), 'net_ckpt': str(tmp_path / 'avatar'),
                       'net_ckpt_finetuned': None, 'testing_data_dir': str(data), 'output_dir': str(out)},
           'model': {'cano_template': {'pos_encoding': 10}, 'warping_field': {'pos_encoding': 0}}}
    with open(tmp_path / 'cfg.yaml', 'w') as fh:
        yaml.safe_dump(cfg, fh)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'main.py'), '-c', str(tmp_path / 'cfg.yaml'), '-m', 'test'],
                       capture_output=True, text=True, env=dict(os.environ, AVC_SMPL_DIR=str(smpl_dir)), timeout=900, cwd=ROOT)
    assert r.returncode != 0, r.stdout[-2000:]
    assert 'AVC_ERR_RANGE' in r.stdout and 'status -5' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert 'not attempted' in r.stdout and not list(out.glob('*_mesh.npz'))       # nothing was written; frames 1, 2 were not run on weights known to be bad


def test_main_gathers_meshes_through_rccl_on_one_gpu(tmp_path):
    """`main.py --synthetic --gather-meshes` with AVC_FORCE_DIST=1: one rank, but the process group is RCCL and the product loop drives the exchange as an
    8-GPU run does -- `FramePipeline.avatar_frame` pumps it behind its query launch, steps are submitted frame by frame, batches of `--gather-batch` steps
    are finished and moved to the host -- on the real pipeline's device tensors.  What rank 0 writes must be, frame for frame, what the frames' own files hold."""
    out = tmp_path / 'out'
    env = dict(os.environ, AVC_FORCE_DIST='1', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1')
    env.pop('MASTER_PORT', None)
    cfg = {'training': {'training_data_dir': None},
           'testing': {'vol_res': [48, 64, 32], 'recon_net_ckpt': None, 'net_ckpt': None, 'net_ckpt_finetuned': None, 'testing_data_dir': None, 'output_dir': str(out)},
           'model': {'cano_template': {'pos_encoding': 10}, 'warping_field': {'pos_encoding': 0}}}
    with open(tmp_path / 'cfg.yaml', 'w') as fh:
        yaml.safe_dump(cfg, fh)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'main.py'), '-c', str(tmp_path / 'cfg.yaml'), '-m', 'test', '--synthetic', '--frames', '5', '--gather-meshes',
                        '--gather-batch', '2'], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert '# gathered 5 meshes' in r.stdout and '5 of 5 frames done on 1 rank(s)' in r.stdout
    allm = np.load(str(out / 'all_avatar_meshes.npz'))
    assert allm['frames'].tolist() == list(range(5))
    for f in range(5):
        m = np.load(str(out / ('%04d_mesh.npz' % f)))
        assert m['live_v'].shape[0] > 50
        assert np.array_equal(allm['v_%04d' % f], m['live_v']) and np.array_equal(allm['vn_%04d' % f], m['live_vn']) and np.array_equal(allm['f_%04d' % f], m['f'])


def test_async_frame_loop_writes_what_the_blocking_loop_writes(tmp_path):
    """The frame loop with its host work off the critical path (avatarcap_amd.frame_io: prefetch thread + one pinned upload per frame, meshes out through
    pinned slots to writer threads) against `--sync-io`, the reference's shape of the loop (blocking upload, .cpu(), files written inside the frame):
    every file of every frame byte for byte -- .npz members and both PLYs, colours included -- and the loop's own account of its copies."""
    import json
    cfg = {'training': {'training_data_dir': None},
           'testing': {'vol_res': [48, 64, 32], 'recon_net_ckpt': None, 'net_ckpt': None, 'net_ckpt_finetuned': None, 'testing_data_dir': None, 'output_dir': None},
           'model': {'cano_template': {'pos_encoding': 10}, 'warping_field': {'pos_encoding': 0}}}
    with open(tmp_path / 'cfg.yaml', 'w') as fh:
        yaml.safe_dump(cfg, fh)
    outs = {}
    for tag, extra in (('async', ['--io-slots', '2', '--io-threads', '2']), ('sync', ['--sync-io'])):
        out = tmp_path / tag
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'main.py'), '-c', str(tmp_path / 'cfg.yaml'), '-m', 'test', '--synthetic', '--frames', '6', '--save-ply',
                            '--nerf', '--output-dir', str(out), '--timing-json', str(tmp_path / (tag + '.json'))] + extra,
                           capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        assert '6 of 6 frames done' in r.stdout
        outs[tag] = out
    for f in range(6):
        a, s = np.load(str(outs['async'] / ('%04d_mesh.npz' % f))), np.load(str(outs['sync'] / ('%04d_mesh.npz' % f)))
        assert sorted(a.files) == sorted(s.files) and {'cano_v', 'live_v', 'recon_live_v', 'live_vc', 'recon_live_vc'} <= set(a.files)
        for k in a.files:
            assert np.array_equal(a[k], s[k]), (f, k)
        assert a['cano_v'].shape[0] > 50 and a['recon_cano_v'].shape[0] > 50
        for name in ('%04d_avatar.ply' % f, '%04d_recon.ply' % f):
            assert open(outs['async'] / name, 'rb').read() == open(outs['sync'] / name, 'rb').read(), name
        # the PLY is the reference writer's byte layout of the same arrays
        from avatarcap_amd.utils import obj_io
        obj_io.save_mesh_as_ply(str(tmp_path / 'ref.ply'), a['live_v'], a['f'], a['live_vn'], a['live_vc'].copy())
        assert open(tmp_path / 'ref.ply', 'rb').read() == open(outs['async'] / ('%04d_avatar.ply' % f), 'rb').read()
    t = json.load(open(tmp_path / 'async.json'))
    assert t['frames'] == 6 and t['h2d_copies'] == 6 and t['files_written'] == 18 and t['frames_failed'] == 0      # ONE upload per frame, three files per frame
    assert t['bytes_written'] == sum(os.path.getsize(outs['async'] / n) for n in os.listdir(outs['async']))


def test_prefetcher_and_writer_on_the_device():
    """FramePrefetcher: one packed upload, typed views, usable on the current stream without a host wait; MeshWriter: tensors produced on the compute stream
    arrive intact in the writer thread (event-ordered copy on the side stream), also when the slot has to grow and when submissions outrun the writers."""
    from avatarcap_amd.frame_io import FramePrefetcher, MeshWriter
    dev = torch.device('cuda', 0)
    rs = np.random.RandomState(3)
    host = {i: {'data_idx': i, 'a': rs.randn(6, 64, 64).astype(np.float32), 'b': rs.randint(0, 9, (5,)).astype(np.int32), 'c': rs.rand(7) > 0.5,
                'd': torch.from_numpy(rs.randn(24, 4, 4).astype(np.float32)), 'e': np.zeros((0, 3), np.float32), 'on_dev': torch.arange(4, device=dev)} for i in range(6)}
    pf = FramePrefetcher(lambda i: host[i], list(range(6)), dev, depth=2)
    w = MeshWriter(dev, slots=2, threads=2)
    got = {}
    for i in range(6):
        it = pf.get(i)
        assert it['a'].is_cuda and it['a'].shape == (1, 6, 64, 64) and it['c'].dtype == torch.bool and it['on_dev'].shape == (1, 4)
        big = (it['a'][0] * 2.0).repeat(1 + 40 * (i % 3), 1, 1)                        # slot sizes vary: the pinned buffer grows
        w.submit({'x': big, 'b': it['b'][0] + 1, 'c': it['c'][0], 'd': it['d'][0], 'e': it['e'][0]},
                 lambda arr, i=i: got.__setitem__(i, {k: v.copy() for k, v in arr.items()}), tag=i)
        pf.drop(i)
    assert w.close() == [] and pf.h2d_copies == 6
    pf.close()
    for i in range(6):
        assert np.array_equal(got[i]['x'], np.tile(host[i]['a'] * 2.0, (1 + 40 * (i % 3), 1, 1))) and np.array_equal(got[i]['b'], host[i]['b'] + 1)
        assert np.array_equal(got[i]['c'], host[i]['c']) and np.array_equal(got[i]['d'], host[i]['d'].numpy()) and got[i]['e'].shape == (0, 3)
