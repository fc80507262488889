"""ctypes wrapper over oracle/raster_oracle.c -- TEST INFRASTRUCTURE ONLY (pinned on Mesa llvmpipe: tests/golden/gl_golden.npz, see raster_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libraster_oracle.so')
_lib = None


def build(force=False):
    src = os.path.join(_HERE, 'raster_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-shared', '-fPIC', src, '-o', _SO, '-lm'])
    return _SO


def render_cano_mesh(vertices, attrs, faces, center, size=512):
    """-> front (size,size,3) f32, back (size,size,3) f32; see raster_oracle.c for the conventions."""
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.raster_oracle.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    v = np.ascontiguousarray(vertices, np.float32); a = np.ascontiguousarray(attrs, np.float32)
    f = np.ascontiguousarray(faces, np.int32); c = np.ascontiguousarray(center, np.float32)
    outs = []
    for view in (0, 1):
        o = np.zeros((size, size, 3), np.float32)
        _lib.raster_oracle(v.ctypes.data, a.ctypes.data, v.shape[0], f.ctypes.data, f.shape[0], c.ctypes.data, size, view, o.ctypes.data)
        outs.append(o)
    return outs[0], outs[1]


def render_mesh(vertices, attrs, faces, mvp, width, height):
    """General MVP view (raster_mvp_oracle): -> (height, width, 4) f32 RGBA = (attribute, 1), background 0.
    attrs=None renders the vertex positions (the reference's 'position' shader)."""
    global _lib
    if _lib is None:
        render_cano_mesh(np.zeros((3, 3), np.float32), np.zeros((3, 3), np.float32), np.zeros((0, 3), np.int32), np.zeros(3, np.float32), 2)
    _lib.raster_mvp_oracle.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    v = np.ascontiguousarray(vertices, np.float32); f = np.ascontiguousarray(faces, np.int32)
    a = None if attrs is None else np.ascontiguousarray(attrs, np.float32)
    m = np.ascontiguousarray(mvp, np.float32).reshape(16)
    o = np.zeros((height, width, 4), np.float32)
    _lib.raster_mvp_oracle(v.ctypes.data, None if a is None else a.ctypes.data, v.shape[0], f.ctypes.data, f.shape[0], m.ctypes.data,
                           width, height, o.ctypes.data)
    return o
