"""Mesh output in the reference's binary little-endian PLY layout (`utils/obj_io.py:223-269`):
header, then per vertex `3f [3f] [3B]`, then per face `int count(=3) + 3 int`.  The reference packs
one struct per element in a Python loop; this writes the same bytes with NumPy structured arrays."""
from __future__ import annotations

import numpy as np


def save_mesh_as_ply(path, vertices, faces=None, normals=None, colors=None):
    vertices = np.asarray(vertices, np.float32)
    fields = [('x', '<f4'), ('y', '<f4'), ('z', '<f4')]
    header = ['ply', 'format binary_little_endian 1.0', 'element vertex %d' % vertices.shape[0],
              'property float x', 'property float y', 'property float z']
    if normals is not None:
        normals = np.asarray(normals, np.float32)
        fields += [('nx', '<f4'), ('ny', '<f4'), ('nz', '<f4')]
        header += ['property float nx', 'property float ny', 'property float nz']
    if colors is not None:
        colors = np.array(colors, copy=True)
        if colors.max() < 1.:                      # obj_io.py:246-247 (scales floats in [0,1) to bytes)
            colors = colors * 255
        colors = colors.astype(np.uint8)
        fields += [('red', 'u1'), ('green', 'u1'), ('blue', 'u1')]
        header += ['property uchar red', 'property uchar green', 'property uchar blue']
    face_num = 0 if faces is None else int(np.asarray(faces).shape[0])
    header += ['element face %d' % face_num, 'property list int int vertex_indices', 'end_header']
    vrec = np.empty(vertices.shape[0], dtype=np.dtype(fields))
    vrec['x'], vrec['y'], vrec['z'] = vertices[:, 0], vertices[:, 1], vertices[:, 2]
    if normals is not None:
        vrec['nx'], vrec['ny'], vrec['nz'] = normals[:, 0], normals[:, 1], normals[:, 2]
    if colors is not None:
        vrec['red'], vrec['green'], vrec['blue'] = colors[:, 0], colors[:, 1], colors[:, 2]
    with open(path, 'wb') as fp:
        fp.write(('\n'.join(header) + '\n').encode('ascii'))
        fp.write(vrec.tobytes())
        if faces is not None:
            f = np.asarray(faces, np.int32)
            frec = np.empty((f.shape[0], 4), '<i4')
            frec[:, 0] = 3
            frec[:, 1:] = f
            fp.write(frec.tobytes())
