"""Image-feature producer: PIFu stacked hourglass as the reference configures it
(`HGFilter(1, 4, 6, 32, 'group', 'no_down', False)`, network/arch_recon.py:29; HGFilters.py:124-219).
Runs once per frame on PyTorch-ROCm / MIOpen (232 GFLOP); its sampling is fused into the HIP
recon-query kernel.  state_dict-compatible with the reference (203 keys under `image_encoder.`).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _norm(kind, c):
    return nn.GroupNorm(32, c) if kind == 'group' else nn.BatchNorm2d(c)


def norm_relu(m: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """relu(norm(x)) -- every normalisation of the encoder is followed by one (HGFilters.py:64-66,178,204).
    On the GPU a GroupNorm goes through the fused HIP op (avc_group_norm): eager PyTorch spends more time in its
    55 statistics launches per frame than in the convolutions.  Inference only (no autograd through the HIP op)."""
    if isinstance(m, nn.GroupNorm) and x.is_cuda and x.dtype == torch.float32 and not (torch.is_grad_enabled() and x.requires_grad):
        from .. import _lib
        x = x.contiguous()
        y = torch.empty_like(x)
        N, C = x.shape[0], x.shape[1]
        _lib.check(_lib.lib().avc_group_norm(_lib.ctx(x.device), x.data_ptr(), N, C, x.numel() // (N * C), m.num_groups,
                                             m.weight.data_ptr() if m.weight is not None else None,
                                             m.bias.data_ptr() if m.bias is not None else None, float(m.eps), 1, y.data_ptr(),
                                             _lib.stream_ptr(x.device)))
        return y
    return F.relu(m(x))


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
        o1 = self.conv1(norm_relu(self.bn1, x))
        o2 = self.conv2(norm_relu(self.bn2, o1))
        o3 = self.conv3(norm_relu(self.bn3, o2))
        res = x if self.downsample is None else self.downsample[2](norm_relu(self.bn4, x))
        return torch.cat([o1, o2, o3], 1) + res


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

    def _level(self, level, x):
        up1 = self._modules[f'b1_{level}'](x)
        low = self._modules[f'b2_{level}'](F.avg_pool2d(x, 2, stride=2))
        low = self._level(level - 1, low) if level > 1 else self._modules['b2_plus_1'](low)
        low = self._modules[f'b3_{level}'](low)
        return up1 + F.interpolate(low, scale_factor=2, mode='bicubic', align_corners=True)

    def forward(self, x):
        return self._level(self.depth, x)


class HGFilter(nn.Module):
    def __init__(self, stack, depth, in_ch, last_ch, norm='batch', down_type='conv64', use_sigmoid=True):
        super().__init__()
        if down_type != 'no_down' or stack != 1:
            raise NotImplementedError("only stack=1, down_type='no_down' (what ReconNetwork builds) is on the path")
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

    def forward(self, x):
        x = norm_relu(self.bn1, self.conv1(x))
        normx = x = self.conv2(x)                       # 'no_down' branch (HGFilters.py:184-185)
        x = self.conv4(self.conv3(x))
        ll = self.top_m_0(self.m0(x))
        ll = norm_relu(self.bn_end0, self.conv_last0(ll))
        out = self.l0(ll)
        return [torch.tanh(out) if self.use_sigmoid else out], normx
