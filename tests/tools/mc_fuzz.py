#!/opt/conda/bin/python3.9
"""Differential test of oracle/mc_oracle.c against the real scikit-image (conda env of the build container).

Run:  /opt/conda/bin/python3.9 tests/tools/mc_fuzz.py [n_volumes] [seed]
Random volumes of several kinds (white noise = every case / sub-case; small integers = ties of the face /
interior tests and values equal to iso; smooth fields; flat plateaus) go through
`skimage.measure.marching_cubes` and through the oracle; vertices and faces must agree exactly."""
import ctypes
import os
import subprocess
import sys
import warnings

import numpy as np

warnings.filterwarnings('ignore')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SO = os.path.join(ROOT, 'oracle', '_build', 'libmc_oracle.so')


def load():
    src = os.path.join(ROOT, 'oracle', 'mc_oracle.c')
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-shared', '-fPIC', src, '-o', SO])
    lib = ctypes.CDLL(SO)
    lib.mc_oracle.restype = ctypes.c_int
    lib.mc_oracle.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p,
                              ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64),
                              ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]
    lib.mc_oracle_free.argtypes = [ctypes.c_void_p]
    return lib


def oracle_mc(lib, vol, iso, spacing):
    vol = np.ascontiguousarray(vol, np.float32)
    sp = np.ascontiguousarray(spacing, np.float32)
    pv, pf = ctypes.c_void_p(), ctypes.c_void_p()
    nv, nf = ctypes.c_int64(), ctypes.c_int64()
    rc = lib.mc_oracle(vol.ctypes.data, vol.shape[0], vol.shape[1], vol.shape[2], ctypes.c_float(iso), sp.ctypes.data,
                       ctypes.byref(pv), ctypes.byref(nv), ctypes.byref(pf), ctypes.byref(nf))
    assert rc == 0
    v = np.ctypeslib.as_array(ctypes.cast(pv, ctypes.POINTER(ctypes.c_float)), (nv.value, 3)).copy() if nv.value else np.zeros((0, 3), np.float32)
    f = np.ctypeslib.as_array(ctypes.cast(pf, ctypes.POINTER(ctypes.c_int32)), (nf.value, 3)).copy() if nf.value else np.zeros((0, 3), np.int32)
    lib.mc_oracle_free(pv); lib.mc_oracle_free(pf)
    return v, f


def make_volume(rng, kind):
    shp = tuple(int(s) for s in rng.integers(2, 9, 3))
    if kind == 0:
        return rng.standard_normal(shp).astype(np.float32), 0.0
    if kind == 1:
        return rng.integers(-2, 3, shp).astype(np.float32), 0.0
    if kind == 2:
        return rng.integers(0, 4, shp).astype(np.float32), float(rng.choice([0.5, 1.0, 1.5, 2.0]))
    if kind == 3:   # smooth
        g = np.stack(np.meshgrid(*[np.linspace(-1, 1, s) for s in shp], indexing='ij'), -1)
        c = rng.uniform(-0.5, 0.5, 3)
        return (np.linalg.norm(g - c, axis=-1) - rng.uniform(0.3, 0.9)).astype(np.float32), 0.0
    if kind == 4:   # two-level plateaus with a few exact iso values
        v = rng.choice(np.array([-1.0, 1.0, 0.0], np.float32), shp, p=[0.45, 0.45, 0.1])
        return v, 0.0
    v = (rng.standard_normal(shp) * rng.choice([1e-6, 1e-3, 1.0, 1e3])).astype(np.float32)
    return v, 0.0


def main():
    from skimage import measure
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lib = load()
    rng = np.random.default_rng(seed)
    bad = 0
    ncell = 0
    for it in range(n):
        kind = it % 6
        vol, iso = make_volume(rng, kind)
        sp = rng.uniform(0.5, 2.0, 3).astype(np.float32) if it % 2 else np.ones(3, np.float32)
        ncell += (vol.shape[0] - 1) * (vol.shape[1] - 1) * (vol.shape[2] - 1)
        try:
            sv, sf, _, _ = measure.marching_cubes(vol, iso, spacing=sp)
            sv = sv.astype(np.float32) if sv.dtype != np.float32 else sv
        except (RuntimeError, ValueError):
            sv, sf = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
            if not (vol.min() <= iso <= vol.max()):
                continue
        ov, of = oracle_mc(lib, vol, iso, sp)
        ok = sv.shape == ov.shape and sf.shape == of.shape and np.array_equal(sf, of) and np.array_equal(sv, ov)
        if not ok:
            bad += 1
            if bad <= 5:
                print('MISMATCH it', it, 'kind', kind, 'shape', vol.shape, 'iso', iso, 'V', sv.shape, ov.shape, 'F', sf.shape, of.shape)
                if sv.shape == ov.shape:
                    print('  max |dv|', np.abs(sv - ov).max() if sv.size else 0)
                if sf.shape == of.shape:
                    d = np.nonzero((sf != of).any(1))[0]
                    print('  first differing faces', d[:5], sf[d[:3]].tolist(), of[d[:3]].tolist())
                np.save('/tmp/mc_fuzz_bad_%d.npy' % bad, vol)
    print('volumes', n, 'cells', ncell, 'mismatching volumes', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
