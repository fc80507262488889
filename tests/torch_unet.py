"""Stock-torch restatement of UnetNoCond7DS.forward (reference network/unets.py:201-229, blocks :10-60) on the weight containers of
avatarcap_amd/network/unets.py: what the HIP U-Net (csrc/conv_enc.hip) is held to.  Test infrastructure: it is itself pinned to the reference's golden
(tests/test_host.py); the product has no PyTorch path."""
import torch
import torch.nn.functional as F


def _down(blk, x):
    if blk.act:
        x = F.leaky_relu(x, 0.2)
    x = blk.conv(x)
    return blk.bn(x) if hasattr(blk, 'bn') else x


def _up(blk, x, skip=None):
    x = blk.up(F.relu(x))
    if hasattr(blk, 'bn'):
        x = blk.bn(x)
    return x if skip is None else torch.cat([x, skip], 1)


def unet7ds_trace(m, x):
    """-> [(name, tensor)] of every tensor one launch of the HIP plan produces, in order; the last one is the module's output."""
    assert not m.training
    d, out = [x], []
    for i in range(1, 8):
        d.append(_down(getattr(m, f'conv{i}'), d[-1]))
        out.append((f'conv{i}', d[-1]))
    u = _up(m.upconv1, d[7], d[6]); out.append(('upconv1', u))
    u = _up(m.upconv2, u, d[5]); out.append(('upconv2', u))
    u = _up(m.upconv3, u, d[4]); out.append(('upconv3', u))
    u = _up(m.upconv3, u, d[3]); out.append(('upconv3 (again)', u))      # reference quirk: upconv3 applied twice (unets.py:213-214)
    u = _up(m.upconvC5, u, d[2]); out.append(('upconvC5', u))
    u = _up(m.upconvC6, u, d[1]); out.append(('upconvC6', u))
    u = _up(m.upconvC7, u); out.append(('upconvC7', u))
    return out


def unet7ds_torch(m, x):
    return unet7ds_trace(m, x)[-1][1]
