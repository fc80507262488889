"""Inside / outside of a closed triangle mesh for the points of the canonical grid -- what the reference asks trimesh + embree for
(`cano_smpl_trimesh.contains(invalid_pts)`, dataset/avatarcap_dataset.py:121-125) to fill the grid points it does not evaluate.
Ray parity along +z with a half-open coverage rule, so that a ray through a shared edge or vertex counts exactly one of the triangles
around it; evaluated per (x, y) COLUMN of the grid (z is the fastest axis, so all points of a column share their crossings).
float64 throughout.  Embree's behaviour on rays that graze an edge is its own; away from such rays the answer is the mesh's."""
from __future__ import annotations

import numpy as np
import torch


def _cover(ax, ay, bx, by, cx, cy, px, py):
    """Half-open point-in-triangle in the xy plane for either winding (top-left rule on the edge functions); returns (inside, orientation)."""
    area = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)
    sgn = torch.sign(area)

    def edge(x0, y0, x1, y1):
        e = ((x1 - x0) * (py - y0) - (y1 - y0) * (px - x0)) * sgn
        dx, dy = (x1 - x0) * sgn, (y1 - y0) * sgn
        top_left = (dy > 0) | ((dy == 0) & (dx < 0))
        return (e > 0) | ((e == 0) & top_left)
    return edge(ax, ay, bx, by) & edge(bx, by, cx, cy) & edge(cx, cy, ax, ay) & (area != 0), area


@torch.no_grad()
def grid_contains(vertices, faces, axis_x, axis_y, axis_z, device=None, chunk=2048) -> torch.Tensor:
    """vertices (V,3), faces (F,3) of a closed mesh; axis_* the per-axis coordinates of the grid (grid.volume_axes).
    -> bool tensor (Rx*Ry*Rz,) in the grid's flat order (x slowest, z fastest): True = inside."""
    dev = torch.device(device) if device is not None else torch.device('cpu')
    v = torch.as_tensor(np.asarray(vertices), dtype=torch.float64, device=dev)
    f = torch.as_tensor(np.asarray(faces).astype(np.int64), device=dev)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    gx = torch.as_tensor(np.asarray(axis_x), dtype=torch.float64, device=dev)
    gy = torch.as_tensor(np.asarray(axis_y), dtype=torch.float64, device=dev)
    gz = torch.as_tensor(np.asarray(axis_z), dtype=torch.float64, device=dev)
    Rx, Ry, Rz = gx.numel(), gy.numel(), gz.numel()
    px = gx[:, None].expand(Rx, Ry).reshape(-1)
    py = gy[None, :].expand(Rx, Ry).reshape(-1)
    out = torch.zeros((Rx * Ry, Rz), dtype=torch.bool, device=dev)
    for s in range(0, px.numel(), chunk):
        x, y = px[s:s + chunk, None], py[s:s + chunk, None]
        inside, area = _cover(a[None, :, 0], a[None, :, 1], b[None, :, 0], b[None, :, 1], c[None, :, 0], c[None, :, 1], x, y)
        col, tri = torch.nonzero(inside, as_tuple=True)
        if col.numel() == 0:
            continue
        # z of the crossing by barycentric interpolation
        A, B, C = a[tri], b[tri], c[tri]
        qx, qy = px[s + col], py[s + col]
        w0 = ((B[:, 0] - qx) * (C[:, 1] - qy) - (B[:, 1] - qy) * (C[:, 0] - qx)) / area[0, tri]
        w1 = ((C[:, 0] - qx) * (A[:, 1] - qy) - (C[:, 1] - qy) * (A[:, 0] - qx)) / area[0, tri]
        zc = w0 * A[:, 2] + w1 * B[:, 2] + (1 - w0 - w1) * C[:, 2]
        # a grid point is inside when an odd number of crossings lies above it
        above = (zc[:, None] > gz[None, :]).to(torch.int32)                      # (hits, Rz)
        cnt = torch.zeros((min(chunk, px.numel() - s), Rz), dtype=torch.int32, device=dev)
        cnt.index_add_(0, col, above)
        out[s:s + chunk] = (cnt & 1).bool()
    return out.reshape(-1)
