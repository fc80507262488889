"""Image-feature producer: PIFu stacked hourglass as the reference configures it
(`HGFilter(1, 4, 6, 32, 'group', 'no_down', False)`, network/arch_recon.py:29; HGFilters.py:124-219).

The modules below are WEIGHT CONTAINERS, state_dict-compatible with the reference (203 keys under `image_encoder.`): the
encoder itself is hand-written HIP for gfx950 (csrc/conv_enc.hip: implicit-GEMM convolutions on split-fp16 MFMA, GroupNorm +
ReLU applied while the consumer stages its input, statistics produced by the convolution before, one hipGraph per input size),
reached through `avc_hgfilter_pack` / `avc_hgfilter_forward`.  There is no PyTorch / MIOpen path: `forward` raises without the
library or on a CPU tensor (rounds 1-3 ran these layers on MIOpen: ~200 launches and 5.3 ms of kernels per 512^2 frame).
"""
import torch
import torch.nn as nn


def _norm(kind, c):
    return nn.GroupNorm(32, c) if kind == 'group' else nn.BatchNorm2d(c)


class ConvBlock(nn.Module):
    """Three pre-activated 3x3 convs (c/2, c/4, c/4) concatenated + residual (HGFilters.py:33-75).
    `bn4` is allocated even when the 1x1 projection is absent -- it is in the checkpoints."""

    def __init__(self, cin, cout, norm='batch'):
        super().__init__()
        h, q = cout // 2, cout // 4
        self.conv1 = nn.Conv2d(cin, h, 3, 1, 1, bias=False)
        self.conv2 = nn.Conv2d(h, q, 3, 1, 1, bias=False)
        self.conv3 = nn.Conv2d(q, q, 3, 1, 1, bias=False)
        self.bn1, self.bn2, self.bn3, self.bn4 = _norm(norm, cin), _norm(norm, h), _norm(norm, q), _norm(norm, cin)
        self.downsample = None
        if cin != cout:
            self.downsample = nn.Sequential(self.bn4, nn.ReLU(True), nn.Conv2d(cin, cout, 1, 1, bias=False))

    def forward(self, x):
        raise RuntimeError('ConvBlock is evaluated inside avc_hgfilter_forward (HGFilter.forward); it has no stand-alone path')


class HourGlass(nn.Module):
    """Recursive hourglass: avg-pool down, bicubic(align_corners=True) up (HGFilters.py:77-121)."""

    def __init__(self, depth, n_features, norm='batch'):
        super().__init__()
        self.depth = depth
        for level in range(depth, 0, -1):
            self.add_module(f'b1_{level}', ConvBlock(n_features, n_features, norm))
            self.add_module(f'b2_{level}', ConvBlock(n_features, n_features, norm))
        self.add_module('b2_plus_1', ConvBlock(n_features, n_features, norm))
        for level in range(1, depth + 1):
            self.add_module(f'b3_{level}', ConvBlock(n_features, n_features, norm))

    def forward(self, x):
        raise RuntimeError('HourGlass is evaluated inside avc_hgfilter_forward (HGFilter.forward); it has no stand-alone path')


class HGFilter(nn.Module):
    def __init__(self, stack, depth, in_ch, last_ch, norm='batch', down_type='conv64', use_sigmoid=True):
        super().__init__()
        if down_type != 'no_down' or stack != 1 or norm != 'group':
            raise NotImplementedError("only stack=1, norm='group', down_type='no_down' (what ReconNetwork builds) is on the path")
        self.n_stack, self.use_sigmoid = stack, use_sigmoid
        self.conv1 = nn.Conv2d(in_ch, 64, 7, 2, 3)
        self.bn1 = _norm(norm, 64)
        self.conv2 = ConvBlock(64, 128, norm)
        self.conv3 = ConvBlock(128, 128, norm)
        self.conv4 = ConvBlock(128, 256, norm)
        self.m0 = HourGlass(depth, 256, norm)
        self.top_m_0 = ConvBlock(256, 256, norm)
        self.conv_last0 = nn.Conv2d(256, 256, 1)
        self.bn_end0 = _norm(norm, 256)
        self.l0 = nn.Conv2d(256, last_ch, 1)
        self._packed = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: setattr(module, '_packed', None))

    def __getstate__(self):                        # the pack token refers to a live context: never pickled / deep-copied
        d = dict(self.__dict__)
        d['_packed'] = None
        d.pop('_watch', None)
        return d

    def _ctx(self, device):
        """The device's context with THIS module's weights packed (again after load_state_dict / in-place edits / another module's pack)."""
        from .. import _lib
        ctx = _lib.ctx(device)
        w = self.__dict__.get('_watch')
        if w is None:
            w = _lib.TensorWatch(self)
            object.__setattr__(self, '_watch', w)
        ver = (ctx, id(self), w.signature())
        if self._packed != ver or not _lib.owns(ctx, 'hgfilter', ver):
            w = _lib.HGFilterWeights(self)
            _lib.check(_lib.lib().avc_hgfilter_pack(ctx, w.struct))
            self._packed = ver
            _lib.set_owner(ctx, 'hgfilter', ver)
        _lib.apply_range_check(ctx)
        return ctx

    def encode(self, x, want_feat=True, want_normx=False, bind=False):
        """One avc_hgfilter_forward per batch item.  x (B,6,H,W) float32 on the HIP device -> (feat (B,last_ch,H1,W1) | None, normx | None);
        bind=True also makes the (last) item's feature map the context's image feature map (no NCHW round trip, B must be 1)."""
        from .. import _lib
        if not x.is_cuda:
            raise RuntimeError('HGFilter runs on the HIP device only (csrc/conv_enc.hip); there is no CPU path')
        if torch.is_grad_enabled() and x.requires_grad:
            raise RuntimeError('HGFilter: inference only -- no gradient flows through the HIP encoder (training is out of scope, SURVEY.md section 2)')
        if bind and x.shape[0] != 1:
            raise ValueError('HGFilter.encode(bind=True): one frame at a time (B == 1)')
        x = x.contiguous().float()
        B, _, H, W = x.shape
        H1, W1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        ctx = self._ctx(x.device)
        feat = torch.empty((B, self.l0.out_channels, H1, W1), dtype=torch.float32, device=x.device) if want_feat else None
        normx = torch.empty((B, 128, H1, W1), dtype=torch.float32, device=x.device) if want_normx else None
        for b in range(B):
            _lib.check(_lib.lib().avc_hgfilter_forward(ctx, x[b].data_ptr(), H, W, feat[b].data_ptr() if want_feat else None,
                                                       normx[b].data_ptr() if want_normx else None, 1 if bind else 0, _lib.stream_ptr(x.device)))
        return feat, normx

    def forward(self, x):
        """-> ([outputs[-1]], normx)   (HGFilters.py:176-219 with stack == 1)"""
        feat, normx = self.encode(x, want_feat=True, want_normx=True)
        return [torch.tanh(feat) if self.use_sigmoid else feat], normx
