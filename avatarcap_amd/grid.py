"""Canonical query grid.

Restates `AvatarCapDataset.generate_volume_points` (dataset/avatarcap_dataset.py:312-326):
samples sit on linspace(0,1,res) *corners* (end points inclusive), the flat index is
i = x*Ry*Rz + y*Rz + z (torch.meshgrid 'ij', z fastest) and points are mapped into the
canonical bounds as pts * (b1 - b0) + b0 in float32.
"""
from __future__ import annotations

import numpy as np


def linspace01_f32(steps: int) -> np.ndarray:
    """torch.linspace(0, 1, steps, dtype=float32) on the CPU, bit-for-bit: torch fills the lower half
    as start + step*i and the upper half as end - step*(steps-1-i) with a float32 `step` and ONE
    rounding per element (fused multiply-add); float64 holds step*i exactly, so rounding the
    float64 expression once reproduces it (checked against torch in tests/test_oracle_golden.py)."""
    if steps == 1:
        return np.zeros(1, np.float32)
    step = np.float64(np.float32(1.0) / np.float32(steps - 1))
    i = np.arange(steps, dtype=np.float64)
    lo = (step * i).astype(np.float32)
    hi = (1.0 - step * (steps - 1 - i)).astype(np.float32)
    return np.where(np.arange(steps) < steps // 2, lo, hi).astype(np.float32)


def generate_volume_points_np(bounds: np.ndarray, res) -> np.ndarray:
    """(Rx*Ry*Rz, 3) float32, same values and order as the reference (see module docstring)."""
    bounds = np.asarray(bounds, np.float32)
    xs, ys, zs = (linspace01_f32(int(r)) for r in res)
    xv, yv, zv = np.meshgrid(xs, ys, zs, indexing='ij')
    pts = np.stack([xv.reshape(-1), yv.reshape(-1), zv.reshape(-1)], -1).astype(np.float32)
    return (pts * (bounds[1] - bounds[0]) + bounds[0]).astype(np.float32)


def volume_axes_np(bounds: np.ndarray, res):
    """The three per-axis coordinate tables of the grid: axis_a[k] = lin_a[k] * (b1_a - b0_a) + b0_a with the reference's float32
    roundings (one for the product, one for the sum -- avatarcap_dataset.py:323), so that point i = x*Ry*Rz + y*Rz + z of
    generate_volume_points_np equals (axis_x[x], axis_y[y], axis_z[z]) bit for bit.  Input of avc_avatar_query_grid."""
    bounds = np.asarray(bounds, np.float32)
    ln = (bounds[1] - bounds[0]).astype(np.float32)
    return tuple((linspace01_f32(int(r)) * ln[a] + bounds[0][a]).astype(np.float32) for a, r in enumerate(res))


def volume_axes(bounds, res, device=None):
    import torch
    from . import config
    device = device if device is not None else config.device
    return tuple(torch.from_numpy(a).to(device) for a in volume_axes_np(bounds, res))


def generate_volume_points(bounds, testing_res=(256, 256, 256), device=None):
    """Reference signature (avatarcap_dataset.py:312): (N,3) float32 tensor on `device`.  Built on the
    host with the bit-exact restatement above and uploaded once per sequence, so the query points are
    identical to the reference's CPU path whatever the device's own linspace kernel does."""
    import torch
    from . import config
    device = device if device is not None else config.device
    return torch.from_numpy(generate_volume_points_np(bounds, testing_res)).to(device)
