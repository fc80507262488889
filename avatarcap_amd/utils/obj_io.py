"""Mesh output in the reference's binary little-endian PLY layout (`utils/obj_io.py:223-269`):
header, then per vertex `3f [3f] [3B]`, then per face `int count(=3) + 3 int`.  The reference packs
one struct per element in a Python loop; this writes the same bytes with NumPy structured arrays."""
from __future__ import annotations

import numpy as np


def ply_header(n_vertices, n_faces, normals=True, colors=False) -> bytes:
    """The header `save_mesh_as_ply` writes (obj_io.py:225-238)."""
    header = ['ply', 'format binary_little_endian 1.0', 'element vertex %d' % n_vertices, 'property float x', 'property float y', 'property float z']
    if normals:
        header += ['property float nx', 'property float ny', 'property float nz']
    if colors:
        header += ['property uchar red', 'property uchar green', 'property uchar blue']
    header += ['element face %d' % n_faces, 'property list int int vertex_indices', 'end_header']
    return ('\n'.join(header) + '\n').encode('ascii')


def ply_records_device(vertices, faces=None, normals=None, colors=None):
    """The byte payload of `save_mesh_as_ply` assembled ON THE DEVICE: -> (header bytes, {'vrec': (V, 3|6) float32 or, with colours, (V, 15|27) uint8; 'frec': (F, 4) int32}).
    The per-vertex records `3f [3f] [3B]` and the per-face records `int 3 + 3 int` (obj_io.py:249-268) are two concatenations there; what crosses to
    the host is the file's own bytes, and the writer thread does nothing but `write` (avatarcap_amd.frame_io.MeshWriter).  The colour rule of :246-248
    (floats below 1 are scaled by 255, then truncated to bytes) is evaluated without reading the maximum back."""
    import torch
    V = int(vertices.shape[0])
    v = vertices.to(torch.float32)
    if colors is None:
        # records of whole floats: one float32 concatenation, handed over as it is (the writer takes the bytes of whatever array it gets)
        out = {'vrec': (torch.cat([v, normals.to(torch.float32)], dim=1) if normals is not None else v).contiguous()}
    else:
        # 3 bytes of colour behind 12 or 24 bytes of floats: the record is not a multiple of 4 bytes, so it is assembled byte-wise
        cols = [v.contiguous().reshape(-1).view(torch.uint8).view(V, 12)]
        if normals is not None:
            cols.append(normals.to(torch.float32).contiguous().reshape(-1).view(torch.uint8).view(V, 12))
        c = colors.to(torch.float32)
        scale = torch.where(c.max() < 1., 255., 1.).to(torch.float32) if V else 1.
        cols.append((c * scale).to(torch.uint8).contiguous().view(V, 3))
        out = {'vrec': torch.cat(cols, dim=1)}
    F = 0
    if faces is not None:
        F = int(faces.shape[0])
        f = faces.to(torch.int32)
        frec = torch.empty((F, 4), dtype=torch.int32, device=f.device)
        frec[:, 0] = 3
        frec[:, 1:] = f
        out['frec'] = frec
    return ply_header(V, F, normals is not None, colors is not None), out


def write_ply_records(path, header: bytes, arrays: dict, prefix: str = ''):
    """Writer-thread half of `ply_records_device`."""
    with open(path, 'wb') as fp:
        fp.write(header)
        for key in ('vrec', 'frec'):
            a = arrays.get(prefix + key)
            if a is not None and a.size:
                fp.write(memoryview(np.ascontiguousarray(a).reshape(-1)).cast('B'))


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
