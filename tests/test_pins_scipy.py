"""Pins, with what this image DOES have, of the stand-ins for libraries it does not.

OpenCV and pytorch3d are absent, so four small functions of the oracle / host mirror restate their definitions: `cv2.erode` (3x3 rectangle,
iterations), `cv2.distanceTransform(DIST_L1, 3)`, `cv2.Rodrigues`, `pytorch3d.transforms.axis_angle_to_matrix`
(normal_fusion/normal_fusion.py:104-108, :72-75; dataset/smpl.py:81).  SciPy implements the same mathematical objects independently --
`ndimage.binary_erosion`, `ndimage.distance_transform_cdt(metric='taxicab')`, `spatial.transform.Rotation.from_rotvec` -- and is held against them
here.  What this does NOT pin is a convention that only OpenCV itself could confirm (the erosion's border value, the chamfer's saturation); those
are stated where they are restated.
"""
import numpy as np
import pytest
from scipy import ndimage
from scipy.spatial.transform import Rotation

from oracle import normal_fusion_oracle as nfo


def _masks():
    rs = np.random.RandomState(5)
    out = [(rs.rand(64, 48) < 0.93).astype(np.uint8), (rs.rand(40, 40) < 0.6).astype(np.uint8)]
    yy, xx = np.mgrid[:96, :80]
    out.append(((yy - 50) ** 2 / 30 ** 2 + (xx - 38) ** 2 / 22 ** 2 < 1).astype(np.uint8))                 # a blob well inside
    out.append((np.abs(yy - 48) + np.abs(xx - 2) < 40).astype(np.uint8))                                    # a blob touching the border
    ring = ((yy - 48) ** 2 + (xx - 40) ** 2 < 35 ** 2) & ((yy - 48) ** 2 + (xx - 40) ** 2 > 12 ** 2)
    out.append(ring.astype(np.uint8))
    out.append(np.ones((20, 30), np.uint8)); out.append(np.zeros((7, 9), np.uint8))
    return out


@pytest.mark.parametrize('iterations', [1, 3])
def test_erode_equals_scipy_binary_erosion(iterations):
    """cv.erode(mask, getStructuringElement(MORPH_RECT, (3, 3)), iterations) == scipy's binary erosion with the full 3x3 structure, the outside of the
    image counting as set (OpenCV's default border value for erosion is +inf: `border_value=1`)."""
    for m in _masks():
        ref = ndimage.binary_erosion(m.astype(bool), structure=np.ones((3, 3), bool), iterations=iterations, border_value=1)
        assert np.array_equal(nfo.erode3x3(m, iterations).astype(bool), ref)


def test_l1_distance_transform_equals_scipy_taxicab_cdt():
    """cv.distanceTransform(mask, DIST_L1, 3) is the exact city-block distance to the nearest zero pixel: scipy's chamfer transform with the taxicab
    metric computes the same integers.  (A mask without any zero pixel has no defined distance: OpenCV saturates, the restatement caps at 8192 --
    that convention is not scipy's to confirm and is left out.)"""
    for m in _masks():
        if m.min() > 0:
            continue
        ref = ndimage.distance_transform_cdt(m, metric='taxicab')
        got = nfo.distance_transform_l1(m)
        assert got.dtype == np.float32 and np.array_equal(got, ref.astype(np.float32))


def _rotvecs():
    rs = np.random.RandomState(9)
    v = rs.normal(0, 1.0, (200, 3))
    v[:20] *= 1e-4; v[20:30] *= 1e-9; v[30:40] *= 3.0                                     # small, tiny and large angles
    v[40] = [np.pi, 0, 0]; v[41] = [0, 0, 0]; v[42] = [0, 1e-20, 0]
    return v


def test_rodrigues_equals_scipy_rotvec():
    from avatarcap_amd.utils.cv_compat import rodrigues
    from avatarcap_amd.synthetic import _rodrigues
    for v in _rotvecs():
        R = Rotation.from_rotvec(v).as_matrix()
        got = rodrigues(v)
        assert got.dtype == np.float64 and np.abs(got - R).max() < 1e-14
        assert np.abs(_rodrigues(v) - R).max() < 1e-6
        v32 = v.astype(np.float32)
        got32 = rodrigues(v32)                                                               # OpenCV's output depth = input depth (double arithmetic inside)
        assert got32.dtype == np.float32
        assert np.array_equal(got32, Rotation.from_rotvec(v32.astype(np.float64)).as_matrix().astype(np.float32)) or \
            np.abs(got32.astype(np.float64) - Rotation.from_rotvec(v32.astype(np.float64)).as_matrix()).max() < 6e-8


def test_axis_angle_to_matrix_equals_scipy_rotvec():
    """pytorch3d's axis_angle_to_matrix goes through a quaternion; the rotation it defines is the rotation vector's."""
    v = _rotvecs()
    R = Rotation.from_rotvec(v).as_matrix()
    got = nfo.axis_angle_to_matrix(v.astype(np.float64))
    assert np.abs(got - R).max() < 1e-13
    got32 = nfo.axis_angle_to_matrix(v.astype(np.float32))
    assert got32.dtype == np.float32 and np.abs(got32 - R).max() < 1e-6
    # and the torch twin the fusion goldens were generated with (tests/golden/make_golden_fusion.py -> tests/test_normal_fusion.py)
    torch = pytest.importorskip('torch')
    import test_normal_fusion as tnf
    gt = tnf._aa2mat_torch(torch.from_numpy(v)).numpy()
    assert np.abs(gt - R).max() < 1e-13


def test_mc_golden_records_the_library_build():
    """The marching-cubes goldens name the binary that made them (version + sha256 of the Lewiner extension and of its tables)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mc_golden.npz'))
    assert str(g['skimage_version']) == '0.18.3'
    for key in ('sha256_lewiner_cy_so', 'sha256_lewiner_luts_py', 'sha256_lewiner_py'):
        name, digest = str(g[key]).split()
        assert name.startswith('_marching_cubes_lewiner') and len(digest) == 64
