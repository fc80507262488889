"""ctypes wrapper over oracle/mc_oracle.c -- TEST INFRASTRUCTURE ONLY (parity unpinned vs scikit-image,
see the header of mc_oracle.c).  `build()` compiles with gcc into oracle/_build/."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libmc_oracle.so')
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, 'mc_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-shared', '-fPIC', src, '-o', _SO])
    return _SO


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.mc_oracle.restype = ctypes.c_int
        _lib.mc_oracle.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                   ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64),
                                   ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]
        _lib.mc_oracle_free.argtypes = [ctypes.c_void_p]
        _lib.mc_oracle_cell.restype = ctypes.c_int
        _lib.mc_oracle_cell.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def marching_cubes(vol: np.ndarray, iso: float, spacing) -> tuple[np.ndarray, np.ndarray]:
    """vol (X,Y,Z) f32 -> verts (V,3) f32 in index*spacing units, faces (F,3) i32 (pre-flip winding)."""
    lib = _load()
    vol = np.ascontiguousarray(vol, np.float32)
    sp = np.ascontiguousarray(spacing, np.float32)
    X, Y, Z = vol.shape
    pv, pf = ctypes.c_void_p(), ctypes.c_void_p()
    nv, nf = ctypes.c_int64(), ctypes.c_int64()
    rc = lib.mc_oracle(vol.ctypes.data, X, Y, Z, ctypes.c_float(iso), sp.ctypes.data,
                       ctypes.byref(pv), ctypes.byref(nv), ctypes.byref(pf), ctypes.byref(nf))
    if rc != 0:
        raise MemoryError('mc_oracle failed')
    verts = np.ctypeslib.as_array(ctypes.cast(pv, ctypes.POINTER(ctypes.c_float)), (max(nv.value, 1), 3))[:nv.value].copy()
    faces = np.ctypeslib.as_array(ctypes.cast(pf, ctypes.POINTER(ctypes.c_int32)), (max(nf.value, 1), 3))[:nf.value].copy()
    lib.mc_oracle_free(pv); lib.mc_oracle_free(pf)
    return verts, faces


def cell_triangles(val8) -> np.ndarray:
    """Triangles (as cube-edge ids) of one cell with corner values val8 - iso; (T,3) int."""
    lib = _load()
    v = np.ascontiguousarray(val8, np.float32)
    tri = np.zeros(36, np.int32)
    n = lib.mc_oracle_cell(v.ctypes.data, tri.ctypes.data)
    return tri[:3 * n].reshape(-1, 3).copy()
