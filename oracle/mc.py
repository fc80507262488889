"""ctypes wrapper over oracle/mc_oracle.c -- TEST INFRASTRUCTURE ONLY.  The C file restates scikit-image's Lewiner
marching cubes and is pinned against the real library (tests/golden/mc_golden.npz, tests/tools/mc_fuzz.py; see its header).
`build()` compiles with gcc into oracle/_build/."""
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
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, 'lewiner_luts.h')))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < newest:
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
        _lib.mc_oracle_last_tiling.restype = ctypes.c_int
    return _lib


def marching_cubes(vol: np.ndarray, iso: float, spacing) -> tuple[np.ndarray, np.ndarray]:
    """What `skimage.measure.marching_cubes(vol, iso, spacing=spacing)` returns as (vertices, faces): vol (X,Y,Z) f32 ->
    verts (V,3) f32 = index * spacing, faces (F,3) i32 (the library's 'descent' winding, before the reference's flip).
    An empty surface gives empty arrays (the library raises; oracle.recon_mesh restates that)."""
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
    verts = np.ctypeslib.as_array(ctypes.cast(pv, ctypes.POINTER(ctypes.c_float)), (nv.value, 3)).copy() if nv.value else np.zeros((0, 3), np.float32)
    faces = np.ctypeslib.as_array(ctypes.cast(pf, ctypes.POINTER(ctypes.c_int32)), (nf.value, 3)).copy() if nf.value else np.zeros((0, 3), np.int32)
    lib.mc_oracle_free(pv); lib.mc_oracle_free(pf)
    return verts, faces


def cell_triangles(val8) -> np.ndarray:
    """Triangles (as the library's cube-edge ids, 12 = centre vertex) of one cell with corner values v0..v7 minus iso; (T,3) int."""
    lib = _load()
    v = np.ascontiguousarray(val8, np.float32)
    tri = np.zeros(36, np.int32)
    n = lib.mc_oracle_cell(v.ctypes.data, tri.ctypes.data)
    return tri[:3 * n].reshape(-1, 3).copy()
