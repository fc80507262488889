"""Host-side mirror of the reference's `network/arch_recon.py` (ReconNetwork, :9-76).

`image_encoder` (HGFilter) runs once per frame as hand-written HIP (csrc/conv_enc.hip); the per-point decoder
(bilinear 32-channel sample + z + weight-normed LeakyReLU MLP + sigmoid) runs in recon_kernel
(csrc/fused_mlp.hip).  state_dict keys match recon_net.pt.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib
from . import HGFilters as hg
from .mlp import MLP
from .arch_avatar import _mlp_entries


class ReconNetwork(nn.Module):
    def __init__(self):
        super().__init__()
        self.image_encoder = hg.HGFilter(1, 4, 6, 32, 'group', 'no_down', False)
        self.image_decoder = MLP(in_channels=33, out_channels=1, inter_channels=[512, 256, 128], res_layers=[1, 2],
                                 nlactv='leaky_relu', norm='weight', last_op='sigmoid')
        self._packed_version = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: setattr(module, '_packed_version', None))

    def get_feat_maps(self, image):
        """`self.image_encoder(image)[0]` (arch_recon.py:52-53): the hand-written HIP encoder (csrc/conv_enc.hip), replayed as one hipGraph per
        input size (`config.hg_graph`, avc_set_option "enc_graph"; plain launches of the same kernels give the same bits)."""
        from .. import config
        if image.is_cuda:
            _lib.set_option('enc_graph', 1 if getattr(config, 'hg_graph', True) else 0, image.device)
        feat, _ = self.image_encoder.encode(image, want_feat=True)
        return [feat]

    def bind_feat_map(self, image):
        """The encoder's output for this frame becomes the context's image feature map directly (channel-last as the decoder reads it: no NCHW
        tensor, no avc_set_img_feat_map).  -> the token `decode` / `decode_grid` take in place of a feature map."""
        from .. import config
        _lib.set_option('enc_graph', 1 if getattr(config, 'hg_graph', True) else 0, image.device)
        self.image_encoder.encode(image, want_feat=False, bind=True)
        ctx = _lib.ctx(image.device)
        token = ('bound', id(self), int(image.data_ptr()), tuple(image.shape))
        _lib.set_owner(ctx, 'img_feat_map', token)
        return token

    def _set_map(self, ctx, img_feat_map, dev):
        if isinstance(img_feat_map, tuple):                       # bind_feat_map's token
            if not _lib.owns(ctx, 'img_feat_map', img_feat_map):
                raise RuntimeError('decode: the bound image feature map has been replaced since bind_feat_map()')
            return
        m = img_feat_map
        _lib.check(_lib.lib().avc_set_img_feat_map(ctx, _lib.dev_ptr(m, name='img_feat_map'), m.shape[0], m.shape[1], m.shape[2], _lib.stream_ptr(dev)))
        _lib.set_owner(ctx, 'img_feat_map', None)

    def _ctx(self, device):
        ctx = _lib.ctx(device)
        w = self.__dict__.get('_watch')
        if w is None:
            w = _lib.TensorWatch(self.image_decoder)
            object.__setattr__(self, '_watch', w)
        ver = (ctx, id(self), w.signature())
        if self._packed_version != ver or not _lib.owns(ctx, 'recon', ver):      # another ReconNetwork may have packed into this context since
            fc = _lib.DenseList(_mlp_entries(self.image_decoder))
            _lib.check(_lib.lib().avc_pack_recon_weights(ctx, fc.arr))
            self._packed_version = ver
            _lib.set_owner(ctx, 'recon', ver)
        _lib.apply_range_check(ctx)
        return ctx

    def infer_grid(self, items, axes, res, index=None):
        """`infer` for items['cano_pts'] that ARE the points of the canonical grid (`axes` = grid.volume_axes(bounds, res), order of
        generate_volume_points) or the subset of it given by the flat indices `index` (int32, device; the valid band in the order of
        dataset.infer_pts): the coordinates are generated in the kernel; a dense grid whose last axis is a multiple of 128 points is column-folded
        (include/avcap.h avc_recon_query_grid: ~1e-6 from `infer` on the same points), everything else is bit-identical to `infer`.  -> (1,N)"""
        with torch.no_grad():
            imgs = torch.cat([items['front_normal'], items['back_normal']], dim=1)
            return self.decode_grid(axes, res, self.bind_feat_map(imgs), items['cano_smpl_center'], index)

    def decode_grid(self, axes, res, img_feat_map, center, index=None):
        import ctypes as C
        res = [int(r) for r in res]
        dev = axes[0].device
        bound = isinstance(img_feat_map, tuple)
        if not bound and img_feat_map.shape[0] != 1:
            raise ValueError('decode_grid: one frame at a time (B == 1)')
        N = res[0] * res[1] * res[2] if index is None else int(index.numel())
        for a, r in zip(axes, res):
            if a.numel() != r:
                raise ValueError(f'decode_grid: axis table of {a.numel()} entries for a resolution of {r}')
        ctx = self._ctx(dev)
        out = torch.empty((1, N), dtype=torch.float32, device=dev)
        self._set_map(ctx, img_feat_map if bound else img_feat_map[0], dev)
        ax = (_lib.dev_ptr(axes[0], name='axis_x'), _lib.dev_ptr(axes[1], name='axis_y'), _lib.dev_ptr(axes[2], name='axis_z'))
        if index is None:
            _lib.check(_lib.lib().avc_recon_query_grid(ctx, *ax, (C.c_int32 * 3)(*res), _lib.f3(center[0]), out.data_ptr(), _lib.stream_ptr(dev)))
        else:
            if index.dtype != torch.int32 or not index.is_contiguous() or index.device != dev:
                raise TypeError('decode_grid: index must be a contiguous int32 tensor on the device of the axis tables')
            _lib.check(_lib.lib().avc_recon_query_grid_subset(ctx, *ax, (C.c_int32 * 3)(*res), index.data_ptr(), N, _lib.f3(center[0]),
                                                              out.data_ptr(), _lib.stream_ptr(dev)))
        return out

    def infer(self, items):
        """items['cano_pts'] (1,N,3), ['cano_smpl_center'] (1,3), ['front_normal'], ['back_normal'] (1,3,512,512)
        -> (1,N) occupancy in [0,1]   (arch_recon.py:45-76)."""
        with torch.no_grad():
            pts = items['cano_pts'].contiguous()
            B, N, _ = pts.shape
            imgs = torch.cat([items['front_normal'], items['back_normal']], dim=1)
            if B == 1:
                return self.decode(pts, self.bind_feat_map(imgs), items['cano_smpl_center'])
            return self.decode(pts, self.get_feat_maps(imgs)[-1], items['cano_smpl_center'])

    def decode(self, pts, img_feat_map, center):
        """The per-point loop of infer (arch_recon.py:55-73) for a given feature map."""
        B, N, _ = pts.shape
        ctx = self._ctx(pts.device)
        out = torch.empty((B, N), dtype=torch.float32, device=pts.device)
        bound = isinstance(img_feat_map, tuple)
        if bound and B != 1:
            raise ValueError('decode: a bound feature map serves one frame (B == 1)')
        for b in range(B):
            self._set_map(ctx, img_feat_map if bound else img_feat_map[b], pts.device)
            _lib.check(_lib.lib().avc_recon_query(ctx, _lib.dev_ptr(pts[b], name='cano_pts'), N, _lib.f3(center[b]),
                                                  out[b].data_ptr(), _lib.stream_ptr(pts.device)))
        # the reference concatenates (B,1,n) chunks and squeezes dim 0 (arch_recon.py:73-76): (1,N) for B == 1, which is what
        # main.py:442 indexes with [0]; (B,1,N) otherwise
        return out[:, None, :].squeeze(0)
