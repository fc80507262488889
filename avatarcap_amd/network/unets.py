"""Pose-feature producer: the 7-level U-Net of the warping field (reference `UnetNoCond7DS`,
network/unets.py:169-229).  Runs once per frame on PyTorch-ROCm / MIOpen (SURVEY.md section 2 row 4:
10.35 GFLOP, not the hot loop); its *sampling* is fused into the HIP query kernel.

state_dict-compatible with the reference (50 keys incl. the dead `upconv4.*`).
"""
import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F


@contextlib.contextmanager
def deterministic_convs():
    """MIOpen picks an atomics-based transposed-convolution kernel for this U-Net by default: the pose feature map then differs by ~2e-6
    from call to call, which is enough to move a marching-cubes vertex count by a few units between two runs of the same frame.  With
    PyTorch's deterministic flag MIOpen is asked for deterministic solutions only (bit-identical across calls, and 10 % faster here:
    tools/determinism_producers.py).  The flag is restored on exit."""
    prev = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        yield
    finally:
        torch.backends.cudnn.deterministic = prev


class _Down(nn.Module):
    """LeakyReLU(0.2) -> Conv2d(k4,s2,p1,no bias) -> BatchNorm2d(affine=False); act/bn optional
    (reference Conv2DBlock, unets.py:10-27: the activation comes *before* the conv)."""

    def __init__(self, cin, cout, bn=True, act=True):
        super().__init__()
        self.act = act
        self.conv = nn.Conv2d(cin, cout, 4, 2, 1, bias=False)
        if bn:
            self.bn = nn.BatchNorm2d(cout, affine=False)

    def forward(self, x):
        if self.act:
            x = F.leaky_relu(x, 0.2)
        x = self.conv(x)
        return self.bn(x) if hasattr(self, 'bn') else x


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

    def forward(self, x, skip=None):
        x = self.up(F.relu(x))
        if hasattr(self, 'bn'):
            x = self.bn(x)
        return x if skip is None else torch.cat([x, skip], 1)


class UnetNoCond7DS(nn.Module):
    def __init__(self, input_nc=3, output_nc=3, nf=64, up_mode='upconv', use_dropout=False):
        super().__init__()
        assert up_mode == 'upconv' and not use_dropout, 'only the configuration the reference instantiates'
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

    def forward(self, x):
        """On the HIP device the ~70 MIOpen launches of one pass (0.7 ms of kernels that the host needs ~1.4 ms to enqueue -- more when eight ranks share a
        host) are recorded ONCE per input shape and set of weights as a hipGraph and replayed with one launch per frame (`config.unet_graph`): the same
        kernels on the same arguments, bit for bit the eager pass (tests/test_gpu_producers.py).  The result is a fresh tensor, not the graph's buffer."""
        from .. import config
        if x.is_cuda and getattr(config, 'unet_graph', True) and not (torch.is_grad_enabled() and x.requires_grad):
            y = self._graph_forward(x)
            if y is not None:
                return y
        with deterministic_convs():
            return self._forward(x)

    def __getstate__(self):                    # a captured graph belongs to this process and these buffers: never pickled / deep-copied
        d = dict(self.__dict__)
        d.pop('_graph', None); d.pop('_graph_failed', None)
        return d

    def _graph_forward(self, x):
        key = (tuple(x.shape), x.dtype, x.device, tuple(p._version for p in self.parameters()), tuple(p.data_ptr() for p in self.parameters()),
               tuple(b._version for b in self.buffers()))
        g = self.__dict__.get('_graph')
        if g is None or g['key'] != key:
            if self.__dict__.get('_graph_failed') == key:
                return None
            try:
                with torch.no_grad():
                    static_in = x.detach().clone()
                    side = torch.cuda.Stream(device=x.device)
                    side.wait_stream(torch.cuda.current_stream(x.device))
                    with torch.cuda.stream(side), deterministic_convs():
                        for _ in range(2):          # MIOpen's solution search and workspaces settle before the capture
                            self._forward(static_in)
                    torch.cuda.current_stream(x.device).wait_stream(side)
                    graph = torch.cuda.CUDAGraph()
                    with deterministic_convs(), torch.cuda.graph(graph):
                        out = self._forward(static_in)
                g = {'key': key, 'graph': graph, 'in': static_in, 'out': out}
                self.__dict__['_graph'] = g
            except Exception as e:              # noqa: BLE001 -- a capture the runtime refuses: the eager launches of the same kernels, and say so once
                import warnings
                warnings.warn(f'UNet7DS: hipGraph capture failed ({type(e).__name__}: {e}); launching eagerly')
                self.__dict__['_graph_failed'] = key
                self.__dict__.pop('_graph', None)
                return None
        g['in'].copy_(x)
        g['graph'].replay()
        return g['out'].clone()

    def _forward(self, x):
        d = [x]
        for i in range(1, 8):
            d.append(getattr(self, f'conv{i}')(d[-1]))
        u = self.upconv1(d[7], d[6])
        u = self.upconv2(u, d[5])
        u = self.upconv3(u, d[4])
        u = self.upconv3(u, d[3])                      # reference quirk: upconv3 applied twice (unets.py:213-214)
        u = self.upconvC5(u, d[2])
        u = self.upconvC6(u, d[1])
        return self.upconvC7(u)
