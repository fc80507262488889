#!/opt/conda/bin/python3.9
"""Generates tests/golden/mc_golden.npz: inputs and outputs of the REAL `skimage.measure.marching_cubes`
(the call of the reference's utils/recon_util.py:64) for a set of small volumes.

Run in the build container:   /opt/conda/bin/python3.9 tests/golden/make_golden_mc.py
scikit-image 0.18.3 lives under /opt/conda (python3.9); the reference pins 0.17.2 -- the same
`_marching_cubes_lewiner_cy` extension and look-up tables as far as we know (0.17 introduced
`marching_cubes` with method='lewiner' as default; nothing under measure/_marching_cubes_lewiner* is
listed in the 0.18 release notes).  The GPU box never runs this script; it only reads the .npz.

Stored per case k:  vol_k (float32 volume), iso_k, spacing_k (float32 x3), verts_k (V,3) float32 and
faces_k (F,3) int32 exactly as the library returns them (gradient_direction='descent', allow_degenerate=True).
Cases: white noise (every case and sub-case of the algorithm that random data reaches: 36 of the 38 tiling tables;
12.1.2 and 13.5.2 did not occur in 4e5 directed trials nor in 1.9e7 fuzzed cells, tests/tools/mc_fuzz.py), small integers and plateaus
(ties of the face / interior tests, values equal to iso), analytic bodies (sphere, torus, two spheres),
anisotropic spacing, non-cubic shapes, the minimum 2x2x2 volume, scaled noise (1e-6 .. 1e3).
"""
import os
import sys
import warnings

import numpy as np

warnings.filterwarnings('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    rng = np.random.default_rng(20240921)
    out = []
    for shp in [(12, 12, 12), (9, 14, 7), (16, 5, 11), (7, 7, 20)]:
        out.append(('noise', rng.standard_normal(shp).astype(np.float32), 0.0, rng.uniform(0.5, 2.0, 3)))
    out.append(('noise_iso', rng.uniform(0, 1, (12, 10, 11)).astype(np.float32), 0.5, (1, 1, 1)))
    for shp in [(10, 10, 10), (6, 9, 12)]:
        out.append(('ints', rng.integers(-2, 3, shp).astype(np.float32), 0.0, (1, 1, 1)))
    out.append(('ints_iso1', rng.integers(0, 4, (10, 9, 8)).astype(np.float32), 1.0, (1, 1, 1)))
    out.append(('plateau', rng.choice(np.array([-1.0, 1.0, 0.0], np.float32), (10, 10, 10), p=[0.45, 0.45, 0.1]), 0.0, (1, 1, 1)))
    out.append(('checker', ((np.indices((8, 8, 8)).sum(0) & 1) * 2 - 1).astype(np.float32), 0.0, (1, 1, 1)))     # case 13 everywhere, all ties
    for s in [1e-6, 1e-3, 1e3]:
        out.append(('scaled', (rng.standard_normal((8, 8, 8)) * s).astype(np.float32), 0.0, (1, 1, 1)))
    g = np.stack(np.meshgrid(*[np.linspace(-1, 1, 24)] * 3, indexing='ij'), -1)
    out.append(('sphere', (np.linalg.norm(g, axis=-1) - 0.7).astype(np.float32), 0.0, (2 / 23,) * 3))
    out.append(('sphere_occ', (1 / (1 + np.exp(8 * (np.linalg.norm(g, axis=-1) - 0.6)))).astype(np.float32), 0.5, (0.1, 0.2, 0.3)))
    q = np.sqrt(g[..., 0] ** 2 + g[..., 1] ** 2) - 0.6
    out.append(('torus', (np.sqrt(q ** 2 + g[..., 2] ** 2) - 0.25).astype(np.float32), 0.0, (1, 1, 1)))
    two = np.minimum(np.linalg.norm(g - [0.0, 0.0, 0.36], axis=-1), np.linalg.norm(g + [0.0, 0.0, 0.36], axis=-1)) - 0.35
    out.append(('two_spheres', two.astype(np.float32), 0.0, (1, 1, 1)))
    out.append(('min', np.array([[[1, -1], [-1, 1]], [[-1, 1], [1, -2]]], np.float32), 0.0, (1, 1, 1)))
    out.append(('thin', rng.standard_normal((2, 2, 30)).astype(np.float32), 0.0, (1, 1, 1)))
    return out


def main():
    import skimage
    from skimage import measure
    store = {'skimage_version': np.array(skimage.__version__)}
    # which binary made these vectors: sha256 of the Lewiner extension and of its look-up tables, as installed (the reference pins 0.17.2,
    # requirements.txt:8; only 0.18.3 exists in this image and nothing offline can prove the two releases' extensions identical -- DESIGN.md section 4)
    import glob
    import hashlib
    mdir = os.path.dirname(measure.__file__)
    for key, pat in (('sha256_lewiner_cy_so', '_marching_cubes_lewiner_cy*.so'), ('sha256_lewiner_luts_py', '_marching_cubes_lewiner_luts.py'),
                     ('sha256_lewiner_py', '_marching_cubes_lewiner.py')):
        f = sorted(glob.glob(os.path.join(mdir, pat)))[0]
        store[key] = np.array(os.path.basename(f) + ' ' + hashlib.sha256(open(f, 'rb').read()).hexdigest())
        print(key, store[key])
    names = []
    for k, (name, vol, iso, sp) in enumerate(cases()):
        sp = np.asarray(sp, np.float32)
        v, f, _, _ = measure.marching_cubes(vol, iso, spacing=sp)
        assert v.dtype == np.float32 and f.dtype == np.int32, (v.dtype, f.dtype)
        store['vol_%d' % k] = vol
        store['iso_%d' % k] = np.float64(iso)
        store['spacing_%d' % k] = sp
        store['verts_%d' % k] = v
        store['faces_%d' % k] = f
        names.append(name)
        print(k, name, vol.shape, 'V', v.shape[0], 'F', f.shape[0])
    store['names'] = np.array(names)
    # error behaviour of the library the host mirror restates
    for label, vol, iso in [('no_surface', np.where(np.arange(8).reshape(2, 2, 2) < 4, 0.0, 1.0).astype(np.float32) * 0 + np.float32(1.0), 1.0),
                            ('out_of_range', np.zeros((3, 3, 3), np.float32), 0.5)]:
        try:
            measure.marching_cubes(vol, iso)
            store['err_' + label] = np.array('none')
        except Exception as e:  # noqa: BLE001
            store['err_' + label] = np.array(type(e).__name__ + ': ' + str(e))
        print(label, store['err_' + label])
    path = os.path.join(HERE, 'mc_golden.npz')
    np.savez_compressed(path, **store)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    sys.exit(main())
