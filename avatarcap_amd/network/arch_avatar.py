"""Host-side mirror of the reference's `network/arch_avatar.py` for the test-mode hot path.

Same class names, constructor defaults, state_dict keys, call signatures and return shapes
(SURVEY.md section 8(b)); the per-point math runs in libavcap_hip.so (csrc/fused_mlp.hip).

  DoubleTNet        arch_avatar.py:26-83     weights + `forward(pts)` -> (rgb, alpha, occ)
  WarpingField      arch_avatar.py:86-140    U-Net on PyTorch-ROCm (once per frame) + `query`
  CanoBlendWeightVolume  :143-165            trilinear 24-ch blend weights (colour path)
  GeoTexAvatar      arch_avatar.py:168-237   container + colour-path `forward`
  OccupancyNet      arch_avatar.py:352-381   `query(batch)` -> {'cano_pts_ov', 'nonrigid_offset'}
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import config
from .. import _lib
from .mlp import MLP, OffsetDecoder
from .unets import UnetNoCond7DS


def _mlp_entries(m: MLP):
    out = []
    for l, fc in enumerate(m.fc_list):
        conv = fc[0] if isinstance(fc, nn.Sequential) else fc
        if hasattr(conv, 'weight_g'):
            out.append({'w': conv.weight_v, 'b': conv.bias, 'g': conv.weight_g})
        else:
            out.append({'w': conv.weight, 'b': conv.bias})
    return out


class DoubleTNet(nn.Module):
    """Template Geo-Tex network (occupancy + NeRF colour), reference arch_avatar.py:26-83."""

    def __init__(self):
        super().__init__()
        self.pos_encoding_freq = config.cfg['model']['cano_template'].get('pos_encoding', 10)
        print('# Canonical Template: Positional encoding %d' % self.pos_encoding_freq)
        in_channels = 3 + 6 * self.pos_encoding_freq            # get_embedder (net_util.py:40-55)
        self.shared_mlp = MLP(in_channels, 256, [256] * 6, res_layers=[4], nlactv='relu')
        self.geo_mlp = MLP(256, 2, [128], nlactv='leaky_relu')
        self.clr_mlp = MLP(256, 3, [256, 128], nlactv='relu')
        with torch.no_grad():                                     # init_out_weights (arch_avatar.py:17-23,60)
            self.geo_mlp.fc_list[-1].weight.uniform_(-1e-5, 1e-5)
            self.geo_mlp.fc_list[-1].bias.zero_()

    def pack_into(self, ctx):
        shared = _lib.DenseList(_mlp_entries(self.shared_mlp))
        geo = _lib.DenseList(_mlp_entries(self.geo_mlp))
        clr = _lib.DenseList(_mlp_entries(self.clr_mlp))
        _lib.check(_lib.lib().avc_pack_template_weights(ctx, shared.arr, geo.arr, clr.arr, int(self.pos_encoding_freq)))

    def forward(self, pts):
        """pts (B,N,3) -> rgb (B,N,3), alpha (B,N,1), occ (B,N,1)   (arch_avatar.py:65-83).
        Needs the owning GeoTexAvatar to have packed the weights (it does so lazily)."""
        owner = getattr(self, '_owner', None)
        if owner is None:
            raise RuntimeError('DoubleTNet.forward: construct it inside a GeoTexAvatar')
        return owner()._template_forward(pts)


class WarpingField(nn.Module):
    """Pose-dependent warping field, reference arch_avatar.py:86-140."""

    def __init__(self):
        super().__init__()
        self.pose_feat_dim = 64
        self.unet = UnetNoCond7DS(input_nc=6, output_nc=self.pose_feat_dim, nf=32, up_mode='upconv', use_dropout=False)
        self.pos_encoding_freq = config.cfg['model']['warping_field'].get('pos_encoding', 0)
        print('# Warping Field: positional encoding %d' % self.pos_encoding_freq)
        in_channels = 3 + 6 * self.pos_encoding_freq + self.pose_feat_dim
        self.mlp = OffsetDecoder(in_channels)
        self.out_layer_coord_affine = nn.Conv1d(256, 3, 1)
        with torch.no_grad():                                     # init_out_weights (arch_avatar.py:104-105)
            self.out_layer_coord_affine.weight.uniform_(-1e-5, 1e-5)
            self.out_layer_coord_affine.bias.zero_()
        self.pose_feat_map = None
        self._map_on_device = None      # (data_ptr, batch index) of the map the context currently holds

    def pack_into(self, ctx):
        convs = _lib.DenseList([{'w': getattr(self.mlp, f'conv{i}').weight, 'b': getattr(self.mlp, f'conv{i}').bias} for i in range(1, 8)])
        bns = _lib.BnList([{'gamma': getattr(self.mlp, f'bn{i}').weight, 'beta': getattr(self.mlp, f'bn{i}').bias,
                            'mean': getattr(self.mlp, f'bn{i}').running_mean, 'var': getattr(self.mlp, f'bn{i}').running_var,
                            'eps': getattr(self.mlp, f'bn{i}').eps} for i in range(1, 8)])
        out = _lib.DenseList([{'w': self.out_layer_coord_affine.weight, 'b': self.out_layer_coord_affine.bias}])
        _lib.check(_lib.lib().avc_pack_warp_weights(ctx, convs.arr, bns.arr, out.arr, int(self.pos_encoding_freq)))

    def precompute_conv(self, batch):
        """self.pose_feat_map = unet(smpl_pos_map)  (arch_avatar.py:109-111); MIOpen, once per frame."""
        self.pose_feat_map = self.unet(batch['smpl_pos_map']).contiguous()
        self._map_on_device = None

    def bind_map(self, ctx, b):
        if self.pose_feat_map is None:
            raise AttributeError('pose_feat_map is None: call WarpingField.precompute_conv(batch) first')
        key = (id(self), self.pose_feat_map.data_ptr(), self.pose_feat_map._version, b)
        if self._map_on_device != key or not _lib.owns(ctx, 'pose_map', key):
            m = self.pose_feat_map[b]
            _lib.check(_lib.lib().avc_set_pose_feat_map(ctx, _lib.dev_ptr(m, name='pose_feat_map'), m.shape[0], m.shape[1], m.shape[2],
                                                        _lib.stream_ptr(m.device)))
            self._map_on_device = key
            _lib.set_owner(ctx, 'pose_map', key)

    def query(self, pts, batch):
        """pts (B,N,3) -> offsets (B,N,3)  (arch_avatar.py:113-140)."""
        owner = getattr(self, '_owner', None)
        if owner is None:
            raise RuntimeError('WarpingField.query: construct it inside a GeoTexAvatar')
        return owner()._avatar_query(pts, batch, want_offset=True, want_rgba=False)[1]


class CanoBlendWeightVolume:
    """Trilinear fetch of the 24-channel canonical blend-weight volume (arch_avatar.py:143-165) -- `avc_blend_weight_sample` (csrc/render.hip) on the
    channel-last volume, the layout of the .npy the reference loads (:145).  `base_weight_volume` keeps the reference's attribute, (1, 24, X, Y, Z)."""

    def __init__(self, base_weight_volume_path=None, base_weight_volume=None):
        if base_weight_volume is None:
            base_weight_volume = np.load(base_weight_volume_path)          # (X,Y,Z,24) (arch_avatar.py:145)
        v = np.ascontiguousarray(np.asarray(base_weight_volume, np.float32))
        if v.ndim != 4 or v.shape[3] % 4:
            raise ValueError(f'CanoBlendWeightVolume: a (X, Y, Z, 24) volume is expected, got {v.shape}')
        self._xyzc = torch.from_numpy(v).to(config.device)
        self.base_weight_volume = self._xyzc.permute((3, 0, 1, 2))[None]     # a view: (1, 24, X, Y, Z)

    def forward(self, pts):
        """pts (B,N,3) scaled to [0,1] -> (B,N,24)."""
        if not pts.is_cuda:
            raise RuntimeError('CanoBlendWeightVolume runs on the HIP device only (csrc/render.hip); there is no CPU path')
        B, N, _ = pts.shape
        pts = pts.contiguous().float()
        if self._xyzc.device != pts.device:                                   # follows its first caller to the device, once
            self._xyzc = self._xyzc.to(pts.device)
            self.base_weight_volume = self._xyzc.permute((3, 0, 1, 2))[None]
        vol = self._xyzc
        X, Y, Z, Cn = vol.shape
        out = torch.empty((B, N, Cn), dtype=torch.float32, device=pts.device)
        _lib.check(_lib.lib().avc_blend_weight_sample(_lib.ctx(pts.device), _lib.dev_ptr(vol, name='base_weight_volume'), (C.c_int32 * 3)(X, Y, Z), Cn,
                                                      pts.data_ptr(), B * N, out.data_ptr(), _lib.stream_ptr(pts.device)))
        return out


class GeoTexAvatar(nn.Module):
    def __init__(self, base_weight_volume=None):
        super().__init__()
        import weakref
        self.cano_template = DoubleTNet()
        if base_weight_volume is not None:
            self.cano_weight_volume = CanoBlendWeightVolume(base_weight_volume=base_weight_volume)
        else:
            tdir = (config.cfg.get('training') or {}).get('training_data_dir')
            # the reference loads this file unconditionally, even in test mode (arch_avatar.py:174)
            self.cano_weight_volume = CanoBlendWeightVolume(str(tdir) + '/cano_base_blend_weight_volume.npy')
        self.warping_field = WarpingField()
        ref = weakref.ref(self)
        object.__setattr__(self.cano_template, '_owner', ref)
        object.__setattr__(self.warping_field, '_owner', ref)
        self._packed_version = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: setattr(module, '_packed_version', None))

    # ---- packing (BatchNorm folded from running stats => eval-mode semantics, as in main.py:297) ----
    def _weights_version(self):
        w = self.__dict__.get('_watch')
        if w is None:
            w = _lib.TensorWatch(self)
            object.__setattr__(self, '_watch', w)
        return w.signature()

    def _ctx(self, device):
        ctx = _lib.ctx(device)
        ver = (ctx, id(self), self._weights_version())
        if self._packed_version != ver or not _lib.owns(ctx, 'avatar', ver):     # another network may have packed into this context since
            self.warping_field.pack_into(ctx)
            self.cano_template.pack_into(ctx)
            self._packed_version = ver
            _lib.set_owner(ctx, 'avatar', ver)
        _lib.apply_range_check(ctx)
        return ctx

    def _avatar_query(self, pts, batch, want_offset=True, want_rgba=False):
        B, N, _ = pts.shape
        pts = pts.contiguous()
        ctx = self._ctx(pts.device)
        occ = torch.empty((B, N, 1), dtype=torch.float32, device=pts.device)
        off = torch.empty((B, N, 3), dtype=torch.float32, device=pts.device) if want_offset else None
        rgba = torch.empty((B, N, 4), dtype=torch.float32, device=pts.device) if want_rgba else None
        sig = 1 if config.if_type == 'occupancy' else 0
        if config.if_type not in ('sdf', 'occupancy'):
            raise ValueError('Invalid config.if_type!')                   # arch_avatar.py:81-82
        for b in range(B):
            self.warping_field.bind_map(ctx, b if self.warping_field.pose_feat_map.shape[0] > 1 else 0)
            _lib.check(_lib.lib().avc_avatar_query(
                ctx, _lib.dev_ptr(pts[b], name='cano_pts'), N, _lib.host_f3(batch, 'cano_smpl_center', b), sig,
                occ[b].data_ptr(), off[b].data_ptr() if want_offset else None, rgba[b].data_ptr() if want_rgba else None,
                _lib.stream_ptr(pts.device)))
        return occ, off, rgba

    def _template_forward(self, pts):
        B, N, _ = pts.shape
        pts = pts.contiguous()
        ctx = self._ctx(pts.device)
        occ = torch.empty((B, N, 1), dtype=torch.float32, device=pts.device)
        rgba = torch.empty((B, N, 4), dtype=torch.float32, device=pts.device)
        sig = 1 if config.if_type == 'occupancy' else 0
        _lib.check(_lib.lib().avc_template_query(ctx, _lib.dev_ptr(pts.view(-1, 3), name='pts'), B * N, sig,
                                                 occ.data_ptr(), rgba.data_ptr(), _lib.stream_ptr(pts.device)))
        return rgba[..., :3], rgba[..., 3:], occ

    def forward(self, wpts, viewdirs, dists, batch, pts_space='posed'):
        """Colour / NeRF path (arch_avatar.py:178-237): returns {'raw' (B,N,4), 'occ', 'nonrigid_offset'}."""
        from ..utils.smpl_util import smpl_util
        assert pts_space in ('posed', 'cano', 'temp')
        B, N = wpts.shape[:2]
        if pts_space == 'posed':                                           # inverse skinning (:189-205)
            d2, idx = smpl_util.knn_points(wpts, batch['live_smpl_v'], K=1)
            near_flag = d2[:, :, 0] < 0.08 * 0.08
            with torch.no_grad():
                w = smpl_util.smpl_skinning_weights[idx[:, :, 0]]
                live2cano = torch.linalg.inv(batch['cano2live_jnt_mats'])
                cano_ = smpl_util.skinning(wpts, w, live2cano)
                lo, hi = batch['cano_bounds'][:, 0], batch['cano_bounds'][:, 1]
                cano_ = (cano_ - lo[:, None]) / (hi - lo)[:, None]
            w = self.cano_weight_volume.forward(cano_)
            cano_pts = smpl_util.skinning(wpts, w.contiguous(), live2cano)
        else:
            cano_pts = wpts
            d2, _ = smpl_util.knn_points(wpts, smpl_util.cano_smpl_vertices[None].expand(B, -1, -1), K=1)
            near_flag = d2[:, :, 0] < 0.08 * 0.08                          # :208-209
        if pts_space in ('posed', 'cano'):
            occ, offsets, rgba = self._avatar_query(cano_pts, batch, want_offset=True, want_rgba=True)
            cano_pts = cano_pts + offsets                                  # :213
        else:
            rgb, alpha, occ = self._template_forward(cano_pts)
            rgba = torch.cat([rgb, alpha], -1)
            offsets = torch.zeros_like(cano_pts)
        rgb, alpha = rgba[..., :3], rgba[..., 3:].clone()
        inside = (cano_pts > batch['cano_bounds'][:, :1]) & (cano_pts < batch['cano_bounds'][:, 1:])   # :222-224
        alpha[inside.sum(2) != 3] = 0
        alpha[~near_flag] = 0
        alpha = 1. - torch.exp(-alpha * dists)                             # :228-230
        return {'raw': torch.cat([rgb, alpha], -1), 'occ': occ, 'nonrigid_offset': offsets}


class NerfRenderer:
    """Volume rendering along rays with the GeoTexAvatar as the field (arch_avatar.py:240-349), eval mode
    (no stratified perturbation: `config.perturb` only acts while training, :257).  Used by the test
    loop to colour the avatar's vertices: rays start one unit off the surface along the normal and
    march back through it (main.py:464-477)."""

    def __init__(self, net: GeoTexAvatar):
        self.net = net

    def get_wsampling_points(self, ray_o, ray_d, near, far):
        t_vals = torch.linspace(0., 1., steps=config.N_samples).to(near)                   # :251
        z_vals = near[..., None] * (1. - t_vals) + far[..., None] * t_vals
        pts = ray_o[:, :, None] + ray_d[:, :, None] * z_vals[..., None]
        return pts, z_vals

    def get_density_color(self, wpts, viewdir, z_vals, batch, pts_space):
        n_batch, n_pixel, n_sample = wpts.shape[:3]
        wpts = wpts.reshape(n_batch, n_pixel * n_sample, -1)
        dists = z_vals[..., 1:] - z_vals[..., :-1]                                         # :279-281
        dists = torch.cat([dists, dists[..., -1:]], dim=2).reshape(n_batch, n_pixel * n_sample, -1)
        return self.net(wpts, None, dists, batch, pts_space)

    def get_pixel_value(self, ray_o, ray_d, near, far, occ, depth, batch, pts_space='posed', near_dist=0.05, far_dist=0.05):
        valid = depth > 1e-6                                                               # :289-291
        near = torch.where(valid, depth - near_dist, near)
        far = torch.where(valid, depth + far_dist, far)
        wpts, z_vals = self.get_wsampling_points(ray_o, ray_d, near, far)
        ret = self.get_density_color(wpts, ray_d, z_vals, batch, pts_space)
        n_batch, n_pixel, n_sample = z_vals.shape
        from ..utils.nerf_util import raw2outputs
        raw = ret['raw'].reshape(-1, n_sample, 4)
        rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z_vals.reshape(-1, n_sample), white_bkgd=False)
        ret.update({'rgb_map': rgb_map.view(n_batch, n_pixel, -1), 'acc_map': acc_map.view(n_batch, n_pixel),
                    'depth_map': depth_map.view(n_batch, n_pixel), 'raw': raw.view(n_batch, -1, 4)})
        return ret

    def render(self, batch, pts_space='posed', near_dist=0.05, far_dist=0.05, chunk=1 << 16, want_raw=False):
        """batch keys: ray_o, ray_d (B,P,3), near, far, occupancy, depth (B,P).  The reference walks the rays 2048 at a time (:330); the fused kernel
        has no activation tensors to bound, so `chunk` only caps the per-sample buffers of the posed / temp branches (the device path of 'cano' sizes its
        own scratch and ignores it).
        pts_space == 'cano' (what main.py:475 runs for the vertex colours): `avc_render_rays_cano` -- sample points, the fused query with the colour
        head, the near / inside masks, alpha and raw2outputs on the device, nothing per sample in PyTorch.  It returns rgb_map, acc_map, depth_map and, with
        want_raw=True, 'raw' (B, P*S, 4); the per-sample 'occ' / 'nonrigid_offset' the reference's dict also carries (:311-318) are not materialised --
        asking the returned dict for them raises a KeyError that says so (GeoTexAvatar.forward returns them).  batch['near'] / batch['far'] are updated in
        place for the rays with depth > 1e-6, as the reference does (:287-290)."""
        if pts_space == 'cano' and batch['ray_o'].is_cuda:
            out = self._render_cano(batch, near_dist, far_dist, want_raw)
            valid = batch['depth'] > 1e-6                                                  # the reference's side effect on the caller's tensors (:288-290)
            batch['near'].copy_(torch.where(valid, batch['depth'] - near_dist, batch['near']))   # in place like the reference's masked assignment, without
            batch['far'].copy_(torch.where(valid, batch['depth'] + far_dist, batch['far']))      # the nonzero() (a stream drain) boolean indexing launches
            return out
        n_pixel = batch['ray_o'].shape[1]
        rets = []
        for i in range(0, n_pixel, chunk):
            sl = slice(i, i + chunk)
            rets.append(self.get_pixel_value(batch['ray_o'][:, sl], batch['ray_d'][:, sl], batch['near'][:, sl], batch['far'][:, sl],
                                             batch['occupancy'][:, sl], batch['depth'][:, sl], batch, pts_space, near_dist, far_dist))
        return {k: torch.cat([r[k] for r in rets], dim=1) for k in rets[0].keys()}

    def _render_cano(self, batch, near_dist, far_dist, want_raw=False):
        from ..utils.smpl_util import smpl_util
        net = self.net
        ray_o = batch['ray_o'].contiguous().float()
        B, P, _ = ray_o.shape
        dev = ray_o.device
        ctx = net._ctx(dev)
        if config.if_type not in ('sdf', 'occupancy'):
            raise ValueError('Invalid config.if_type!')
        S = int(config.N_samples)
        f = lambda k: batch[k].contiguous().float()                                       # noqa: E731
        ray_d, near, far, depth = f('ray_d'), f('near'), f('far'), f('depth')
        if smpl_util.cano_smpl_vertices is None:
            raise ValueError('Canonical smpl vertices are invalid!')                          # utils/smpl_util.py:31
        smpl_v = smpl_util.cano_smpl_vertices.contiguous().float()
        t_vals = torch.linspace(0., 1., steps=S).to(near)                                  # :251, the host's linspace as the reference takes it
        rgb = torch.empty((B, P, 3), dtype=torch.float32, device=dev)
        acc = torch.empty((B, P), dtype=torch.float32, device=dev)
        dep = torch.empty((B, P), dtype=torch.float32, device=dev)
        raw = torch.empty((B, P * S, 4), dtype=torch.float32, device=dev) if want_raw else None
        for b in range(B):
            net.warping_field.bind_map(ctx, b if net.warping_field.pose_feat_map.shape[0] > 1 else 0)
            _lib.check(_lib.lib().avc_render_rays_cano(
                ctx, ray_o[b].data_ptr(), ray_d[b].data_ptr(), near[b].data_ptr(), far[b].data_ptr(), depth[b].data_ptr(), float(near_dist), float(far_dist),
                t_vals.data_ptr(), P, S, _lib.host_f3(batch, 'cano_smpl_center', b), _lib.host_f3(batch, 'cano_bounds', b), _lib.dev_ptr(smpl_v, name='cano_smpl_vertices'),
                smpl_v.shape[0], 1 if config.if_type == 'occupancy' else 0, rgb[b].data_ptr(), acc[b].data_ptr(), dep[b].data_ptr(), None, None,
                raw[b].data_ptr() if want_raw else None, _lib.stream_ptr(dev)))
        out = _DeviceRenderDict({'rgb_map': rgb, 'acc_map': acc, 'depth_map': dep})
        if want_raw:
            out['raw'] = raw
        return out


class _DeviceRenderDict(dict):
    """What NerfRenderer.render(pts_space='cano') returns on the device: a dict whose missing reference keys explain themselves."""

    def __missing__(self, key):
        if key == 'raw':
            raise KeyError("'raw': the device path of NerfRenderer.render(pts_space='cano') materialises it only with want_raw=True")
        if key in ('occ', 'nonrigid_offset'):
            raise KeyError(f"'{key}': not materialised by the device path of NerfRenderer.render(pts_space='cano') (per-sample tensors of the fused kernel); "
                           "GeoTexAvatar.forward(..., pts_space='cano') returns it")
        raise KeyError(key)


class OccupancyNet:
    """Chunk-free grid query (arch_avatar.py:352-381)."""

    def __init__(self, net: GeoTexAvatar):
        self.net = net

    def query(self, batch):
        """batch['cano_pts'] (B,N,3), ['cano_smpl_center'] (B,3); warping_field.precompute_conv(batch)
        must have run.  -> {'cano_pts_ov': (B,N,1), 'nonrigid_offset': (B,N,3)}"""
        occ, off, _ = self.net._avatar_query(batch['cano_pts'], batch, want_offset=True, want_rgba=False)
        return {'cano_pts_ov': occ, 'nonrigid_offset': off}

    def query_grid(self, batch, axes, res, want_offset=False, index=None):
        """The same query on the dense grid without the (N,3) point tensor: `axes` = grid.volume_axes(bounds, res) (three device tensors), point order
        as generate_volume_points; `index` (int32 device tensor of flat grid indices) restricts it to those points -- the valid band, in the order of
        dataset.infer_pts.  The offsets (which main.py:360-364 never reads) are only produced on request.  Launches whose tiles share their (x, y)
        column -- a last axis of a multiple of 128 points -- and all `index` launches are column-folded (fused_mlp.hip): a few 1e-6 from query() on the
        materialised points; the others are bit-identical to it.  -> {'cano_pts_ov': (1,N,1)[, 'nonrigid_offset': (1,N,3)]}"""
        net = self.net
        res = [int(r) for r in res]
        N = res[0] * res[1] * res[2] if index is None else int(index.numel())
        dev = axes[0].device
        ctx = net._ctx(dev)
        if config.if_type not in ('sdf', 'occupancy'):
            raise ValueError('Invalid config.if_type!')
        occ = torch.empty((1, N, 1), dtype=torch.float32, device=dev)
        off = torch.empty((1, N, 3), dtype=torch.float32, device=dev) if want_offset else None
        net.warping_field.bind_map(ctx, 0)
        for a, r in zip(axes, res):
            if a.numel() != r:
                raise ValueError(f'query_grid: axis table of {a.numel()} entries for a resolution of {r}')
        ax = (_lib.dev_ptr(axes[0], name='axis_x'), _lib.dev_ptr(axes[1], name='axis_y'), _lib.dev_ptr(axes[2], name='axis_z'))
        sig = 1 if config.if_type == 'occupancy' else 0
        if index is None:
            _lib.check(_lib.lib().avc_avatar_query_grid(ctx, *ax, (C.c_int32 * 3)(*res), _lib.host_f3(batch, 'cano_smpl_center', 0), sig,
                                                        occ.data_ptr(), off.data_ptr() if want_offset else None, _lib.stream_ptr(dev)))
        else:
            if index.dtype != torch.int32 or not index.is_contiguous() or index.device != dev:
                raise TypeError('query_grid: index must be a contiguous int32 tensor on the device of the axis tables')
            _lib.check(_lib.lib().avc_avatar_query_grid_subset(ctx, *ax, (C.c_int32 * 3)(*res), index.data_ptr(), N, _lib.host_f3(batch, 'cano_smpl_center', 0), sig,
                                                               occ.data_ptr(), off.data_ptr() if want_offset else None, _lib.stream_ptr(dev)))
        out = {'cano_pts_ov': occ}
        if want_offset:
            out['nonrigid_offset'] = off
        return out
