"""Host-side logic and the C-ABI surface.  CPU only (no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import golden_inputs as gi
from avatarcap_amd import config, synthetic as syn
from common import geotex_shapes, maxabs

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def test_library_exports_every_declared_symbol():
    from avatarcap_amd import _lib
    lib = _lib.lib()
    hdr = open(os.path.join(ROOT, 'include', 'avcap.h')).read()
    declared = set(re.findall(r'\b(avc_[a-z_0-9]+)\s*\(', hdr))
    assert len(declared) >= 18
    for name in declared:
        assert getattr(lib, name) is not None
    assert declared == set(_lib.exported_symbols())
    assert lib.avc_version() == 100


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_product_path_fails_loudly_without_gpu():
    from avatarcap_amd import _lib
    h = ctypes.c_void_p()
    rc = _lib.lib().avc_ctx_create(0, ctypes.byref(h))
    assert rc != 0 and len(_lib.lib().avc_last_error()) > 0
    with pytest.raises(Exception):
        _lib.ctx(torch.device('cpu'))
    config.cfg = config.default_cfg()
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
    net = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume())
    net.warping_field.pose_feat_map = torch.zeros(1, 64, 256, 256)
    with pytest.raises(Exception):     # no eager fallback exists
        OccupancyNet(net).query({'cano_pts': torch.zeros(1, 8, 3), 'cano_smpl_center': torch.zeros(1, 3)})
    from avatarcap_amd.network.mlp import MLP
    with pytest.raises(RuntimeError):
        MLP(3, 1, [4])(torch.zeros(1, 3, 2))


def test_state_dict_surface_matches_reference_checkpoints():
    """125 keys for net.pt, 214 for recon_net.pt, exact shapes of the hot-path tensors."""
    sh = geotex_shapes()
    assert len(sh) == 125
    assert sh['cano_template.shared_mlp.fc_list.4.0.weight'] == (256, 319, 1)
    assert sh['cano_template.shared_mlp.fc_list.6.weight'] == (256, 256, 1)
    assert sh['cano_template.geo_mlp.fc_list.1.weight'] == (2, 128, 1)
    assert sh['cano_template.clr_mlp.fc_list.2.weight'] == (3, 128, 1)
    assert sh['warping_field.mlp.conv1.weight'] == (256, 67, 1)
    assert sh['warping_field.mlp.conv5.weight'] == (256, 323, 1)
    assert sh['warping_field.mlp.bn7.running_var'] == (256,)
    assert sh['warping_field.out_layer_coord_affine.weight'] == (3, 256, 1)
    assert 'warping_field.unet.upconv4.up.weight' in sh            # dead weights stay loadable (unets.py:188)
    assert sum(k.startswith('warping_field.unet.') for k in sh) == 50
    from avatarcap_amd.network.arch_recon import ReconNetwork
    rs = syn.module_shapes(ReconNetwork())
    assert len(rs) == 214
    assert rs['image_decoder.fc_list.1.0.weight_v'] == (256, 545, 1) and rs['image_decoder.fc_list.1.0.weight_g'] == (256, 1, 1)
    assert rs['image_decoder.fc_list.3.weight'] == (1, 128, 1)
    assert sum(k.startswith('image_encoder.') for k in rs) == 203


def test_config_surface(tmp_path):
    cfg = config.load_config(os.path.join(ROOT, 'configs', 'example.yaml'))
    assert cfg['testing']['vol_res'] == [384, 384, 128]
    assert cfg['model']['cano_template']['pos_encoding'] == 10 and cfg['model']['warping_field']['pos_encoding'] == 0
    assert config.if_type == 'sdf' and config.iso_value == 0. and config.sdf_thres == 0.1 and config.N_samples == 64
    with pytest.raises(ValueError):
        config._iso_for('bogus')


def test_producers_match_reference_golden(golden):
    """The stock-torch restatement of UNet7DS (incl. the upconv3-twice quirk; tests/torch_unet.py) vs the reference on CPU; the HGFilter weight container carries the reference's tensors, and the
    stock-torch restatement the GPU tests hold the HIP encoder to launch by launch (tests/torch_hgfilter.py) reproduces the reference's golden."""
    from avatarcap_amd.network.unets import UnetNoCond7DS
    from avatarcap_amd.network.HGFilters import HGFilter
    torch.set_grad_enabled(False)
    un = UnetNoCond7DS(input_nc=6, output_nc=64, nf=32).eval()
    syn.load_synth(un, gi.SEED_NET)
    from torch_unet import unet7ds_torch
    y = unet7ds_torch(un, torch.from_numpy(gi.pos_map(128)[None])).numpy()[0]      # the restatement the GPU tests hold the HIP U-Net to
    g = golden['G7_unet_samples']
    assert maxabs(y[:, gi.PIX[:, 0] % 128, gi.PIX[:, 1] % 128], g) < 1e-4 * max(1.0, np.abs(g).max())
    hg = HGFilter(1, 4, 6, 32, 'group', 'no_down', False).eval()
    syn.load_synth(hg, gi.SEED_NET)
    from torch_hgfilter import hgfilter_trace
    trace = hgfilter_trace(hg, torch.from_numpy(gi.normal_maps(64)[None]))
    y = trace[-1][2].numpy()[0]
    g = golden['G7_hg_samples']
    assert maxabs(y[:, gi.PIX[:, 0] % 32, gi.PIX[:, 1] % 32], g) < 1e-5 * max(1.0, np.abs(g).max())
    assert len(trace) == 2 + 4 + 3 + 4 + (13 * 3 + 4 + 4) + 3 + 2          # one entry per tensor-producing launch of the HIP plan
    with pytest.raises(RuntimeError, match='HIP device only'):             # the encoder has no CPU / PyTorch path
        hg(torch.from_numpy(gi.normal_maps(64)[None]))
    with pytest.raises(RuntimeError, match='HIP device only'):
        un(torch.from_numpy(gi.pos_map(128)[None]))
    with pytest.raises(NotImplementedError):
        HGFilter(2, 4, 6, 32, 'group', 'conv64', False)


def test_synthetic_body_and_pose():
    b = syn.synthetic_body()
    assert b['cano_smpl_v'].shape == (6890, 3) and b['skin_weights'].shape == (6890, 24)
    assert np.allclose(b['skin_weights'].sum(1), 1, atol=1e-5)
    lo, hi = syn.CANO_BOUNDS
    assert np.all(b['cano_smpl_v'] >= lo) and np.all(b['cano_smpl_v'] <= hi)
    d = np.abs(syn.body_sdf(b['cano_smpl_v']))
    assert np.percentile(d, 99) < 0.01 and d.max() < 0.06      # a few extremities are clipped into the bounds
    jm = syn.random_pose_jnt_mats(3)
    assert jm.shape == (24, 4, 4) and np.allclose(jm[:, 3], [0, 0, 0, 1])
    R = jm[:, :3, :3]
    assert np.allclose(np.einsum('jab,jcb->jac', R, R), np.eye(3), atol=1e-5)
    ident = syn.random_pose_jnt_mats(3, sigma=0.0)
    assert np.allclose(ident, np.eye(4), atol=1e-6)


def test_dense_dataset_item_keys_cpu():
    from avatarcap_amd.dataset import SyntheticTestDataset
    config.cfg = config.default_cfg()
    ds = SyntheticTestDataset([6, 5, 4], valid='dense', n_frames=2, device='cpu')
    it = ds[1]
    for k in ('cano_pts', 'valid_pts_flag', 'smpl_pos_map', 'cano_smpl_center', 'cano_bounds', 'cano2live_jnt_mats'):
        assert k in it
    assert it['cano_pts'].shape == (120, 3) and it['smpl_pos_map'].shape == (6, 256, 256)
    assert it['cano2live_jnt_mats'].shape == (24, 4, 4)
    with pytest.raises(ValueError):
        SyntheticTestDataset([4, 4, 4], valid='nope', device='cpu')


def test_ply_writer_is_byte_identical_to_reference(golden, tmp_path):
    from avatarcap_amd.utils.obj_io import save_mesh_as_ply
    v, f, n, c = gi.ply_mesh()
    for tag, kw in (('v', {}), ('vn', {'normals': n}), ('vnc', {'normals': n, 'colors': c.copy()})):
        fn = tmp_path / f'{tag}.ply'
        save_mesh_as_ply(str(fn), v, f, **kw)
        assert np.array_equal(np.frombuffer(fn.read_bytes(), np.uint8), golden['G14_ply_' + tag]), tag


def _write_exr(path, img, names, pixel_type, compression):
    """An independent writer of the OpenEXR scanline layout (header, offset table, chunks; RLE / ZIPS / ZIP with the
    byte split + delta predictor) -- test-side only."""
    import struct, zlib
    H, W, C = img.shape
    dt = {1: '<f2', 2: '<f4', 0: '<u4'}[pixel_type]
    order = sorted(range(C), key=lambda i: names[i])                      # channels are stored alphabetically
    def attr(name, typ, val): return name.encode() + b'\0' + typ.encode() + b'\0' + struct.pack('<i', len(val)) + val
    ch = b''.join(names[i].encode() + b'\0' + struct.pack('<iB3xii', pixel_type, 0, 1, 1) for i in order) + b'\0'
    box = struct.pack('<4i', 0, 0, W - 1, H - 1)
    hdr = struct.pack('<ii', 20000630, 2) + attr('channels', 'chlist', ch) + attr('compression', 'compression', bytes([compression])) + \
        attr('dataWindow', 'box2i', box) + attr('displayWindow', 'box2i', box) + attr('lineOrder', 'lineOrder', b'\0') + \
        attr('pixelAspectRatio', 'float', struct.pack('<f', 1.0)) + attr('screenWindowCenter', 'v2f', struct.pack('<2f', 0, 0)) + \
        attr('screenWindowWidth', 'float', struct.pack('<f', 1.0)) + b'\0'
    lines = {0: 1, 1: 1, 2: 1, 3: 16}[compression]
    chunks = []
    for y in range(0, H, lines):
        raw = b''.join(img[r, :, i].astype(dt).tobytes() for r in range(y, min(y + lines, H)) for i in order)
        data = raw
        if compression:
            t = np.frombuffer(raw, np.uint8)
            t = np.concatenate([t[0::2], t[1::2]]).astype(np.int64)
            d = t.copy(); d[1:] = (t[1:] - t[:-1] + 128 + 256) & 0xff
            pre = d.astype(np.uint8).tobytes()
            if compression == 1:                                           # RLE: runs of >= 3 as (count - 1, byte), literals as (-n, bytes)
                out, i = bytearray(), 0
                while i < len(pre):
                    j = i
                    while j + 1 < len(pre) and pre[j + 1] == pre[i] and j - i < 126: j += 1
                    if j - i >= 2: out += bytes([j - i, pre[i]]); i = j + 1
                    else:
                        k = i
                        while k < len(pre) and k - i < 127 and not (k + 2 < len(pre) and pre[k] == pre[k + 1] == pre[k + 2]): k += 1
                        out += bytes([256 - (k - i)]) + pre[i:k]; i = k
                comp = bytes(out)
            else:
                comp = zlib.compress(pre)
            data = comp if len(comp) < len(raw) else raw
        chunks.append(struct.pack('<ii', y, len(data)) + data)
    pos = len(hdr) + 8 * len(chunks)
    table = b''
    for c in chunks: table += struct.pack('<Q', pos); pos += len(c)
    open(path, 'wb').write(hdr + table + b''.join(chunks))


def test_exr_reader_round_trip(tmp_path):
    """utils/exr_io.read_exr (stands in for cv.imread(..., IMREAD_UNCHANGED) of the image-normal EXR files, main.py:408-410)."""
    from avatarcap_amd.utils.exr_io import read_exr
    rs = np.random.RandomState(0)
    img = rs.randn(37, 29, 3).astype(np.float32); img[5:20, 3:25] = 0          # flat regions exercise RLE runs
    for comp in (0, 1, 2, 3):
        for ptype in (2, 1):
            p = str(tmp_path / f'n_{comp}_{ptype}.exr')
            _write_exr(p, img, 'RGB', ptype, comp)
            got = read_exr(p)
            want = img if ptype == 2 else img.astype(np.float16).astype(np.float32)
            assert got.dtype == np.float32 and got.shape == (37, 29, 3)
            assert np.array_equal(got, want[..., ::-1])                       # cv.imread order: B, G, R
            assert np.array_equal(read_exr(p, order='RGB'), want)
    p = str(tmp_path / 'rgba.exr')
    rgba = rs.rand(8, 8, 4).astype(np.float32)
    _write_exr(p, rgba, 'RGBA', 2, 3)
    assert np.array_equal(read_exr(p), rgba[..., [2, 1, 0, 3]])
    bad = str(tmp_path / 'piz.exr')
    _write_exr(bad, img, 'RGB', 2, 0)
    raw = bytearray(open(bad, 'rb').read()); i = raw.index(b'compression\0compression\0') + 24 + 4; raw[i] = 4
    open(bad, 'wb').write(raw)
    with pytest.raises(NotImplementedError, match='PIZ'):
        read_exr(bad)
    open(bad, 'wb').write(b'not an exr file at all')
    with pytest.raises(ValueError):
        read_exr(bad)


def test_tensor_watch_sees_what_the_packers_must_see():
    """_lib.TensorWatch: the per-query 'have the packed weights changed?' check without walking the module tree (0.03 ms instead of 0.7): in-place edits under
    no_grad, buffers (BatchNorm running statistics), load_state_dict, a parameter re-assigned in its slot; a deep copy gets a watch of its own."""
    import copy
    import contextlib
    import io
    import torch
    from avatarcap_amd import config
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar
    old_cfg, old_dev = config.cfg, config.device
    config.cfg, config.device = config.default_cfg(), torch.device('cpu')
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32))
        s = [net._weights_version()]
        assert net._weights_version() == s[0]
        with torch.no_grad():
            net.cano_template.geo_mlp.fc_list[1].weight.mul_(2)
        s.append(net._weights_version())
        net.warping_field.mlp.bn3.running_var.add_(1)
        s.append(net._weights_version())
        net.cano_template.geo_mlp.fc_list[1].weight = torch.nn.Parameter(torch.zeros_like(net.cano_template.geo_mlp.fc_list[1].weight))
        s.append(net._weights_version())
        net.load_state_dict(net.state_dict())
        s.append(net._weights_version())
        assert len(set(s)) == len(s)
        twin = copy.deepcopy(net)
        assert twin._weights_version() != net._weights_version() and twin._weights_version() == twin._weights_version()
    finally:
        config.cfg, config.device = old_cfg, old_dev


def test_build_flags_keep_packed_f32_out_of_the_elementwise_units():
    """profiles/r06_store_hazard.md: a packed-f32 VALU result read as store data an instruction later reached memory stale beside another kernel.  hipcc forms
    those instructions by itself; the translation units whose kernels store what they have just computed are built without them (and fused_mlp.hip, where they cost
    issue time beside the MFMAs).  conv_enc.hip keeps them (its element-wise kernels settle() their stores: csrc/store_settle.h)."""
    from avatarcap_amd import build
    nopk = ' '.join(build.NOPK)
    assert '-packed-fp32-ops' in nopk
    for unit in ('mesh.hip', 'knn_lbs.hip', 'raster.hip', 'render.hip', 'misc.hip', 'fusion.hip', 'fused_mlp.hip'):
        assert nopk in ' '.join(build.EXTRA[unit]), unit
    for unit in ('mesh.hip', 'knn_lbs.hip', 'raster.hip', 'render.hip'):          # bit-exact against the C oracles: no contraction
        assert '-ffp-contract=off' in build.EXTRA[unit], unit
    assert 'store_settle.h' in build.HEADERS
    src = open(os.path.join(os.path.dirname(build.__file__), 'csrc', 'conv_enc.hip')).read()
    assert src.count('settle(') >= 3 and '#include "store_settle.h"' in src
