"""Parameter containers with the reference's state_dict surface for `network/mlp.py`.

These modules hold weights only (so `load_state_dict(torch.load(...)['network'])` works with the
reference's checkpoints).  They deliberately have NO forward(): the math runs
in the fused HIP kernels (csrc/fused_mlp.hip) after `pack` folds BatchNorm / weight_norm; there is
no eager fallback to route to by accident.
"""
import torch
import torch.nn as nn


def _conv1(cin, cout):
    return nn.Conv1d(cin, cout, 1)


class MLP(nn.Module):
    """Key surface of reference `MLP` (mlp.py:5-54): `fc_list.{l}.0.{weight,bias}` for hidden layers
    (`weight_g`/`weight_v` when norm == 'weight'), `fc_list.{L}.{weight,bias}` for the last one.
    Layers listed in `res_layers` take `all_channels[l] + in_channels` inputs ([x | x0], mlp.py:61)."""

    def __init__(self, in_channels, out_channels, inter_channels, res_layers=(), nlactv='relu',
                 last_op=None, norm=None):
        super().__init__()
        self.res_layers = list(res_layers)
        self.nlactv = nlactv            # 'relu' | 'leaky_relu' (slope 0.02) | 'soft_plus'
        self.last_op = last_op          # None | 'sigmoid' | 'tanh'
        self.norm = norm
        self.all_channels = [in_channels] + list(inter_channels) + [out_channels]
        self.fc_list = nn.ModuleList()
        for l in range(len(inter_channels)):
            cin = self.all_channels[l] + (in_channels if l in self.res_layers else 0)
            conv = _conv1(cin, self.all_channels[l + 1])
            if norm == 'weight':
                conv = nn.utils.weight_norm(conv)      # same (deprecated) API => same keys as the reference
            self.fc_list.append(nn.Sequential(conv, nn.Identity()))
        self.fc_list.append(_conv1(self.all_channels[-2], out_channels))

    def forward(self, *a, **k):
        raise RuntimeError('avatarcap_amd MLP is a weight container; queries run in the HIP library')


class OffsetDecoder(nn.Module):
    """Key surface of reference `OffsetDecoder` (mlp.py:75-112): conv1..7 + bn1..7, conv5 takes
    in_size + hsize inputs ([x0 | x4], mlp.py:106)."""

    def __init__(self, in_size, hsize=256):
        super().__init__()
        self.hsize, self.in_size = hsize, in_size
        for i in range(1, 8):
            cin = in_size if i == 1 else (hsize + in_size if i == 5 else hsize)
            setattr(self, f'conv{i}', _conv1(cin, hsize))
        for i in range(1, 8):
            setattr(self, f'bn{i}', nn.BatchNorm1d(hsize))

    def forward(self, *a, **k):
        raise RuntimeError('avatarcap_amd OffsetDecoder is a weight container; queries run in the HIP library')
