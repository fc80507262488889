"""The three OpenCV calls on the real-data seam of the test path, from their published definitions (OpenCV is not part of this build,
so these are UNPINNED against the library itself):
  cv.Rodrigues(rvec)                         dataset/smpl.py:81, dataset/avatarcap_dataset.py:229
  cv.resize(img, (w, h), INTER_NEAREST)      dataset/avatarcap_dataset.py:211
  cv.imread(path, IMREAD_UNCHANGED) on .exr  -> utils/exr_io.read_exr (pinned on a file written by the OpenEXR library)
"""
from __future__ import annotations

import numpy as np


def rodrigues(rvec) -> np.ndarray:
    """Axis-angle (3,) or (3,1) -> 3x3 rotation: R = cos(t) I + (1 - cos(t)) r r^T + sin(t) [r]_x with t = |rvec|, r = rvec / t; the identity when
    t < DBL_EPSILON (OpenCV calib3d, cv::Rodrigues).  Like cv::Rodrigues the arithmetic is double whatever the input, and the OUTPUT HAS THE INPUT'S
    DEPTH: a float32 vector -- the reference's pose vectors are (`np.loadtxt(...).astype(np.float32)`, avatarcap_dataset.py:194) -- returns R rounded
    to float32, which is what then enters `(I - R) J` and the kinematic chain (dataset/smpl.py:81-90).  Pinned against
    scipy.spatial.transform.Rotation.from_rotvec (tests/test_host.py); OpenCV itself is absent from this image."""
    a = np.asarray(rvec)
    v = a.astype(np.float64).reshape(3)
    t = float(np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]))
    if t < np.finfo(np.float64).eps:
        R = np.identity(3)
    else:
        r = v / t
        c, s = np.cos(t), np.sin(t)
        K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
        R = c * np.identity(3) + (1 - c) * np.outer(r, r) + s * K
    return R.astype(np.float32) if a.dtype == np.float32 else R


def resize_nearest(img: np.ndarray, dsize) -> np.ndarray:
    """cv.resize(img, dsize=(width, height), interpolation=cv.INTER_NEAREST): dst(y, x) = src(min(floor(y * H / h), H - 1),
    min(floor(x * W / w), W - 1)) -- OpenCV's nearest neighbour samples at the pixel's ORIGIN, not its centre (imgproc resizeNN)."""
    w, h = int(dsize[0]), int(dsize[1])
    H, W = img.shape[:2]
    ys = np.minimum(np.floor(np.arange(h) * (H / h)).astype(np.int64), H - 1)
    xs = np.minimum(np.floor(np.arange(w) * (W / w)).astype(np.int64), W - 1)
    return img[ys][:, xs]


def load_smpl_pos_map(path: str, pos_map_res: int) -> np.ndarray:
    """The test-mode position-map read of AvatarCapDataset.__getitem__ (dataset/avatarcap_dataset.py:207-213): the EXR holds the front and the
    back rendering of the posed SMPL side by side, (H, 2H, 3); nearest resize to (res, 2 res), split into halves, stack on the channel
    axis, channels first -> (6, res, res) float32."""
    from .exr_io import read_exr
    m = read_exr(path)[..., :3]
    m = resize_nearest(m, (2 * pos_map_res, pos_map_res))
    m = np.concatenate([m[:, :pos_map_res, :], m[:, pos_map_res:, :]], axis=-1)
    return np.ascontiguousarray(m.transpose((2, 0, 1)), np.float32)
