"""Pose-feature producer: the 7-level U-Net of the warping field (reference `UnetNoCond7DS`, network/unets.py:169-229), once per frame.

The modules below are WEIGHT CONTAINERS, state_dict-compatible with the reference (50 keys incl. the dead `upconv4.*`); the arithmetic is the
hand-written gfx950 encoder (csrc/conv_enc.hip, `avc_unet_pack` / `avc_unet_forward`): 18 launches of the split-fp16 MFMA convolution kernel in one
hipGraph -- stride-2 convolutions as 3x3 convolutions of space-to-depth tensors, transposed convolutions as 3x3 convolutions with parity-major outputs,
BatchNorm folded into the weights, activations applied while staging, concatenations written in place.  There is no CPU / PyTorch path: a CPU tensor
raises.  The stock-torch restatement the tests hold the kernels to is tests/torch_unet.py.
"""
import torch
import torch.nn as nn


class _Down(nn.Module):
    """LeakyReLU(0.2) -> Conv2d(k4,s2,p1,no bias) -> BatchNorm2d(affine=False); act/bn optional
    (reference Conv2DBlock, unets.py:10-27: the activation comes *before* the conv)."""

    def __init__(self, cin, cout, bn=True, act=True):
        super().__init__()
        self.act = act
        self.conv = nn.Conv2d(cin, cout, 4, 2, 1, bias=False)
        if bn:
            self.bn = nn.BatchNorm2d(cout, affine=False)


class _Up(nn.Module):
    """ReLU -> (ConvTranspose2d k4 s2 p1 | bilinear x2 + Conv2d k3) -> BN(affine=False) -> cat skip
    (reference UpConv2DBlock, unets.py:30-60)."""

    def __init__(self, cin, cout, mode='upconv', bn=True, bias=False):
        super().__init__()
        if mode == 'upconv':
            self.up = nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=bias)
        else:
            self.up = nn.Sequential(nn.Upsample(mode='bilinear', scale_factor=2, align_corners=False),
                                    nn.Conv2d(cin, cout, 3, 1, 1))
        if bn:
            self.bn = nn.BatchNorm2d(cout, affine=False)


class UnetNoCond7DS(nn.Module):
    def __init__(self, input_nc=3, output_nc=3, nf=64, up_mode='upconv', use_dropout=False):
        super().__init__()
        assert up_mode == 'upconv' and not use_dropout, 'only the configuration the reference instantiates'
        if input_nc > 8 or output_nc % 32 or nf % 32:
            raise NotImplementedError('the HIP U-Net takes <= 8 input channels and channel counts that are multiples of 32 (the reference: 6 -> 64, nf 32)')
        w = [nf, 2 * nf, 4 * nf, 8 * nf, 8 * nf, 8 * nf, 8 * nf]
        self.conv1 = _Down(input_nc, w[0], bn=False, act=False)
        for i in range(1, 7):
            setattr(self, f'conv{i + 1}', _Down(w[i - 1], w[i], bn=(i != 6)))
        self.upconv1 = _Up(8 * nf, 8 * nf)
        self.upconv2 = _Up(16 * nf, 8 * nf)
        self.upconv3 = _Up(16 * nf, 8 * nf)
        self.upconv4 = _Up(16 * nf, 4 * nf)           # present in checkpoints, never used (unets.py:188 vs :214)
        self.upconvC5 = _Up(12 * nf, 2 * nf, 'upsample')
        self.upconvC6 = _Up(4 * nf, nf, 'upsample')
        self.upconvC7 = _Up(2 * nf, output_nc, 'upsample', bn=False, bias=True)
        self.output_nc = output_nc
        self._packed = None

    def __getstate__(self):                    # the packed weights belong to this process's device context: never pickled / deep-copied
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
        if self._packed != ver or not _lib.owns(ctx, 'unet', ver):
            w = _lib.UNetWeights(self)
            _lib.check(_lib.lib().avc_unet_pack(ctx, w.struct))
            self._packed = ver
            _lib.set_owner(ctx, 'unet', ver)
        _lib.apply_range_check(ctx)
        return ctx

    def forward(self, x, bind=False):
        """x (B, 6, H, W) float32 on the HIP device, H and W multiples of 128 -> (B, output_nc, H, W)   (unets.py:201-229, eval mode: BatchNorm on
        its running statistics).  bind=True (B == 1) also makes the result the context's pose feature map without the NCHW round trip."""
        from .. import _lib
        if not x.is_cuda:
            raise RuntimeError('UnetNoCond7DS runs on the HIP device only (csrc/conv_enc.hip); there is no CPU path')
        if self.training:
            raise RuntimeError('UnetNoCond7DS: inference only (eval mode: BatchNorm2d on running statistics); call .eval()')
        if torch.is_grad_enabled() and x.requires_grad:
            raise RuntimeError('UnetNoCond7DS: inference only -- no gradient flows through the HIP encoder (training is out of scope, SURVEY.md section 2)')
        if bind and x.shape[0] != 1:
            raise ValueError('UnetNoCond7DS.forward(bind=True): one frame at a time (B == 1)')
        x = x.contiguous().float()
        B, C, H, W = x.shape
        if C != self.conv1.conv.in_channels:
            raise ValueError(f'UnetNoCond7DS: {C} input channels, the module was built for {self.conv1.conv.in_channels}')
        if C != 6:
            raise NotImplementedError('the HIP U-Net reads the 6-channel SMPL position map (front | back xyz, arch_avatar.py:95)')
        from .. import config
        _lib.set_option('enc_graph', 1 if getattr(config, 'hg_graph', True) else 0, x.device)
        ctx = self._ctx(x.device)
        y = torch.empty((B, self.output_nc, H, W), dtype=torch.float32, device=x.device)
        for b in range(B):
            _lib.check(_lib.lib().avc_unet_forward(ctx, x[b].data_ptr(), H, W, y[b].data_ptr(), 1 if bind else 0, _lib.stream_ptr(x.device)))
        return y
