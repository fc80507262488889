"""TEST REFERENCE ONLY: the image encoder restated on stock torch ops (fp32), tensor by tensor in the order of the HIP launch plan
(csrc/conv_enc.hip build_plan), so that a GPU test can name the first launch that disagrees.  Follows the reference's
network/HGFilters.py:33-75 (ConvBlock), :77-121 (HourGlass), :176-219 (HGFilter.forward).  Not imported by the product."""
import torch
import torch.nn.functional as F


def _nr(gn, x):
    return F.relu(F.group_norm(x, gn.num_groups, gn.weight, gn.bias, gn.eps))


def _block(b, x, trace, name):
    res = x
    if b.downsample is not None:
        res = F.conv2d(_nr(b.bn4, x), b.downsample[2].weight)
        trace.append((f'{name}.downsample', 'raw', res))
    o1 = F.conv2d(_nr(b.bn1, x), b.conv1.weight, padding=1)
    trace.append((f'{name}.conv1', 'raw', o1))
    o2 = F.conv2d(_nr(b.bn2, o1), b.conv2.weight, padding=1)
    trace.append((f'{name}.conv2', 'raw', o2))
    o3 = F.conv2d(_nr(b.bn3, o2), b.conv3.weight, padding=1)
    y = torch.cat([o1, o2, o3], 1) + res
    trace.append((f'{name}.conv3', 'y', y))
    return y


def _level(hg, lvl, x, trace):
    low = F.avg_pool2d(x, 2, stride=2)
    trace.append((f'pool_{lvl}', 'raw', low))
    up1 = _block(hg._modules[f'b1_{lvl}'], x, trace, f'b1_{lvl}')
    low = _block(hg._modules[f'b2_{lvl}'], low, trace, f'b2_{lvl}')
    low = _level(hg, lvl - 1, low, trace) if lvl > 1 else _block(hg._modules['b2_plus_1'], low, trace, 'b2_plus_1')
    low = _block(hg._modules[f'b3_{lvl}'], low, trace, f'b3_{lvl}')
    out = up1 + F.interpolate(low, scale_factor=2, mode='bicubic', align_corners=True)
    trace.append((f'upadd_{lvl}', 'raw', out))
    return out


@torch.no_grad()
def hgfilter_trace(m, x):
    """-> [(name, 'raw' | 'y', tensor (1,C,H,W))] in launch order, the last entry being outputs[-1]."""
    trace = []
    t0 = F.conv2d(x, m.conv1.weight, m.conv1.bias, stride=2, padding=3)
    trace.append(('conv1', 'raw', t0))
    x = _nr(m.bn1, t0)
    trace.append(('relu(bn1)', 'raw', x))
    x = _block(m.conv2, x, trace, 'conv2')
    x = _block(m.conv3, x, trace, 'conv3')
    x = _block(m.conv4, x, trace, 'conv4')
    x = _level(m.m0, m.m0.depth, x, trace)
    x = _block(m.top_m_0, x, trace, 'top_m_0')
    cl = F.conv2d(x, m.conv_last0.weight, m.conv_last0.bias)
    trace.append(('conv_last0', 'raw', cl))
    out = F.conv2d(_nr(m.bn_end0, cl), m.l0.weight, m.l0.bias)
    trace.append(('l0', 'raw', out))
    return trace
