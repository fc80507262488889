"""Meshes of the OpenGL-golden scenes (tests/golden/make_golden_gl.py generates the images in the build container, tests/test_raster.py
rebuilds the same meshes): analytic bodies through the marching-cubes oracle, analytic normals."""
import numpy as np

from oracle import mc

CANO_SCENES = ('sphere', 'torus_offcentre', 'two_blobs')


def _grid(n):
    g = np.linspace(-1, 1, n, dtype=np.float32)
    return np.meshgrid(g, g, g, indexing='ij'), 2.0 / (n - 1)


def _mesh(vol, h):
    v, f = mc.marching_cubes(vol, 0.0, [h, h, h])
    return (v - 1.0).astype(np.float32), np.ascontiguousarray(f[:, [2, 1, 0]])


def cano_scene(name):
    """-> vertices (V,3), faces (F,3) int32 (counter-clockwise seen from outside), unit normals (V,3), mesh centre (3,), image size"""
    (x, y, z), h = _grid(48)
    if name == 'sphere':
        v, f = _mesh((0.6 - np.sqrt(x * x + y * y + z * z)).astype(np.float32), h)
        n = v / np.linalg.norm(v, axis=1, keepdims=True)
        return v, f, n.astype(np.float32), np.zeros(3, np.float32), 512
    if name == 'torus_offcentre':           # a hole, self-occlusion, and a centre that is not the origin
        v, f = _mesh((0.2 - np.sqrt((np.sqrt(x * x + z * z) - 0.5) ** 2 + y * y)).astype(np.float32), h)
        q = np.sqrt(v[:, 0] ** 2 + v[:, 2] ** 2)
        ring = np.stack([v[:, 0] / q * 0.5, np.zeros_like(q), v[:, 2] / q * 0.5], 1)
        n = v - ring
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        R = np.array([[1, 0, 0], [0, np.cos(0.6), -np.sin(0.6)], [0, np.sin(0.6), np.cos(0.6)]], np.float32)      # tilt it so that the hole shows
        return (v @ R.T + np.float32([0.07, -0.03, 0.1])).astype(np.float32), f, (n @ R.T).astype(np.float32), np.float32([0.07, -0.03, 0.1]), 512
    if name == 'two_blobs':                  # one body in front of the other: the depth test decides
        a = 0.35 - np.sqrt((x + 0.15) ** 2 + y ** 2 + (z - 0.3) ** 2)
        b = 0.45 - np.sqrt((x - 0.1) ** 2 + (y - 0.05) ** 2 + (z + 0.35) ** 2)
        v, f = _mesh(np.maximum(a, b).astype(np.float32), h)
        ca, cb = np.float32([-0.15, 0, 0.3]), np.float32([0.1, 0.05, -0.35])
        da, db = np.linalg.norm(v - ca, axis=1) - 0.35, np.linalg.norm(v - cb, axis=1) - 0.45
        n = np.where((np.abs(da) < np.abs(db))[:, None], v - ca, v - cb)
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        return v, f, n.astype(np.float32), np.float32([0.02, 0.01, -0.05]), 256
    raise ValueError(name)


def position_scene():
    """A posed-mesh stand-in seen by the pinhole of canonicalize_normal_map: vertices, faces, model-view (world -> camera, y down, z forward),
    intrinsics, image size (non-square)."""
    v, f, _, _, _ = cano_scene('two_blobs')
    th = 0.2
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], np.float32) @ np.diag(np.float32([1, -1, -1]))
    mv = np.eye(4, dtype=np.float32); mv[:3, :3] = R; mv[:3, 3] = [0.03, -0.02, 2.4]
    return v, f, mv, 420.0, 415.0, 203.0, 148.0, 400, 300
