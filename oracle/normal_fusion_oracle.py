"""CPU restatement of the reference's canonical normal fusion (normal_fusion/normal_fusion.py) -- TEST INFRASTRUCTURE ONLY.

PINNED against the reference's own code: tests/golden/make_golden_fusion.py runs /root/reference/normal_fusion/
normal_fusion.py itself (its autograd + torch.optim.Adam loop, its resize / neighbour / blend / face-rectangle code, its
per-vertex canonicalisation and render_cano_mesh's matrices) on the tests' synthetic inputs and stores the results in
tests/golden/fusion_golden.npz; tests/test_normal_fusion.py holds this file to them.  OpenCV and pytorch3d do not exist
offline, so that run uses stand-ins for exactly their calls -- what stays UNPINNED is: cv2.erode /
cv2.distanceTransform (restated below from their definitions; checked against brute force) and pytorch3d's
axis_angle_to_matrix (restated as published: axis_angle_to_quaternion + quaternion_to_matrix).  The two OpenGL renderers of that run are
real OpenGL (Mesa llvmpipe, headless: tests/golden/make_golden_gl.py), which also pins oracle/raster_oracle.c.  The hand-written gradients are also checked against torch.autograd to 1e-12.
Every function cites the reference line it follows.
"""
import numpy as np


# ---- canonicalize_normal_map, per-vertex part (normal_fusion.py:27-60) ------------------------------------------
def canonicalize_vertex_normals(live_v, vert_mats, position_map, normal_map, mv, fx, fy, cx, cy, dt=np.float32):
    """live_v (n,3); vert_mats (n,4,4); position_map (H,W,4) from the 'position' render; normal_map (H,W,3) observed;
    -> canonical per-vertex observed normal (n,3), zero where the vertex is occluded / unobserved."""
    v = live_v.astype(dt); mv = mv.astype(dt)
    H, W = normal_map.shape[:2]
    cam = v @ mv[:3, :3].T + mv[:3, 3]                                                   # :28
    gx = dt(2.) * ((cam[:, 0] / cam[:, 2] * dt(fx) + dt(cx)) / dt(W)) - dt(1.)           # :29,31
    gy = dt(2.) * ((cam[:, 1] / cam[:, 2] * dt(fy) + dt(cy)) / dt(H)) - dt(1.)           # :30,32

    def nearest(g, n):                                                                    # F.grid_sample 'nearest','border',align_corners=True (:35,48)
        pix = np.clip((g + dt(1.)) * dt(0.5) * dt(n - 1), dt(0.), dt(n - 1))
        return np.rint(pix).astype(np.int64)                                              # nearbyint: ties to even
    ix, iy = nearest(gx, W), nearest(gy, H)
    ok = np.isfinite(gx) & np.isfinite(gy)
    ix = np.where(ok, ix, 0); iy = np.where(ok, iy, 0)
    proj_v = position_map.astype(dt)[iy, ix, :3]
    vis = np.linalg.norm(v - proj_v, axis=-1) < dt(0.05)                                  # :36
    n = normal_map.astype(dt)[iy, ix, :3].copy()
    valid = vis & (np.linalg.norm(n, axis=-1) > dt(1e-6)) & ok                            # :49
    n[:, 1:] *= dt(-1.)                                                                   # :59
    n = n @ np.linalg.inv(mv.astype(np.float64))[:3, :3].astype(dt).T                     # :60
    A = vert_mats.astype(np.float64)[:, :3, :3]
    det = np.linalg.det(A)
    good = np.abs(det) > 1e-12                                                            # the reference's linalg.inv raises on singular input
    Ainv = np.zeros_like(A); Ainv[good] = np.linalg.inv(A[good])
    n = np.einsum('vij,vj->vi', Ainv.astype(dt), n)                                       # :61
    n[~(valid & good)] = 0                                                                # :62
    return n


# ---- OpenCV pieces of merge_normal_images (normal_fusion.py:104-108) ----------------------------------------------
def erode3x3(mask, iterations=3):
    """cv.erode(mask, 3x3 rectangle, iterations): a pixel survives iff its whole (2*iterations+1)^2 neighbourhood inside the
    image is set (OpenCV's default border value for erosion is +inf: the outside never erodes)."""
    m = np.asarray(mask).astype(bool)
    H, W = m.shape
    r = int(iterations)
    p = np.ones((H + 2 * r, W + 2 * r), bool)
    p[r:r + H, r:r + W] = m
    out = np.ones((H, W), bool)
    for dy in range(2 * r + 1):
        for dx in range(2 * r + 1):
            out &= p[dy:dy + H, dx:dx + W]
    return out.astype(np.uint8)


DT_CAP = 8192.0     # OpenCV's 3x3 chamfer saturates near (INT_MAX >> 2) / 2^16 when the image holds no zero pixel


def distance_transform_l1(mask):
    """cv.distanceTransform(mask, cv.DIST_L1, 3): city-block distance of every non-zero pixel to the nearest zero pixel
    (exact for the 3x3 mask), float32; 0 on zero pixels."""
    m = np.asarray(mask) > 0
    H, W = m.shape
    big = np.float64(1e18)
    g = np.where(m, big, 0.0)
    for x in range(1, W): g[:, x] = np.minimum(g[:, x], g[:, x - 1] + 1)               # along rows
    for x in range(W - 2, -1, -1): g[:, x] = np.minimum(g[:, x], g[:, x + 1] + 1)
    for y in range(1, H): g[y] = np.minimum(g[y], g[y - 1] + 1)                         # along columns (L1 is separable)
    for y in range(H - 2, -1, -1): g[y] = np.minimum(g[y], g[y + 1] + 1)
    return np.minimum(g, DT_CAP).astype(np.float32)


# ---- pytorch3d.transforms.axis_angle_to_matrix, forward and hand-written backward -----------------------------------
def axis_angle_to_matrix(aa):
    """(...,3) -> (...,3,3): axis_angle_to_quaternion (half angle; sin(t/2)/t replaced by 1/2 - t^2/48 below 1e-6) then
    quaternion_to_matrix (two_s = 2 / |q|^2)."""
    dt = aa.dtype
    th = np.sqrt((aa * aa).sum(-1, keepdims=True))
    half = th * dt.type(0.5)
    small = th < dt.type(1e-6)
    k = np.where(small, dt.type(0.5) - th * th / dt.type(48.), np.sin(half) / np.where(small, dt.type(1.), th))
    q = np.concatenate([np.cos(half), aa * k], -1)
    r, i, j, kk = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    s2 = dt.type(2.) / (q * q).sum(-1)
    R = np.stack([1 - s2 * (j * j + kk * kk), s2 * (i * j - kk * r), s2 * (i * kk + j * r),
                  s2 * (i * j + kk * r), 1 - s2 * (i * i + kk * kk), s2 * (j * kk - i * r),
                  s2 * (i * kk - j * r), s2 * (j * kk + i * r), 1 - s2 * (i * i + j * j)], -1)
    return R.reshape(aa.shape[:-1] + (3, 3))


def axis_angle_to_matrix_backward(aa, G):
    """dL/daa given G = dL/dR (...,3,3); the norm's gradient at the zero vector is 0 (torch's convention)."""
    dt = aa.dtype
    th = np.sqrt((aa * aa).sum(-1))
    half = th * dt.type(0.5)
    small = th < dt.type(1e-6)
    ths = np.where(small, dt.type(1.), th)
    sh, ch = np.sin(half), np.cos(half)
    k = np.where(small, dt.type(0.5) - th * th / dt.type(48.), sh / ths)
    dk = np.where(small, -th / dt.type(24.), (ch * dt.type(0.5) * th - sh) / (ths * ths))
    r, i, j, kk = ch, aa[..., 0] * k, aa[..., 1] * k, aa[..., 2] * k
    N = r * r + i * i + j * j + kk * kk
    s2 = dt.type(2.) / N
    G00, G01, G02, G10, G11, G12, G20, G21, G22 = [G[..., a, b] for a in range(3) for b in range(3)]
    P = [-(j * j + kk * kk), i * j - kk * r, i * kk + j * r, i * j + kk * r, -(i * i + kk * kk), j * kk - i * r, i * kk - j * r, j * kk + i * r, -(i * i + j * j)]
    dN = sum(g * p for g, p in zip([G00, G01, G02, G10, G11, G12, G20, G21, G22], P)) * (-s2 / N)
    dr = s2 * (-kk * G01 + j * G02 + kk * G10 - i * G12 - j * G20 + i * G21) + dN * 2 * r
    di = s2 * (-2 * i * (G11 + G22) + j * (G01 + G10) + kk * (G02 + G20) + r * (G21 - G12)) + dN * 2 * i
    dj = s2 * (-2 * j * (G00 + G22) + i * (G01 + G10) + kk * (G12 + G21) + r * (G02 - G20)) + dN * 2 * j
    dkk = s2 * (-2 * kk * (G00 + G11) + i * (G02 + G20) + j * (G12 + G21) + r * (G10 - G01)) + dN * 2 * kk
    dq = np.stack([di, dj, dkk], -1)
    dth = dr * (-sh * dt.type(0.5)) + (dq * aa).sum(-1) * dk
    unit = np.where(small[..., None] & (th[..., None] == 0), dt.type(0.), aa / np.where(th == 0, dt.type(1.), th)[..., None])
    return dq * k[..., None] + dth[..., None] * unit


# ---- resize_img (normal_fusion.py:80-86): bilinear, align_corners=True -----------------------------------------------
def _lin_weights(n_in, n_out, dt):
    pos = np.arange(n_out, dtype=np.float64) * ((n_in - 1) / (n_out - 1) if n_out > 1 else 0.0)
    i0 = np.minimum(np.floor(pos).astype(np.int64), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    t = (pos - i0).astype(dt)
    Wm = np.zeros((n_out, n_in), dt)
    Wm[np.arange(n_out), i0] += 1 - t
    Wm[np.arange(n_out), i1] += t
    return Wm


def resize_bilinear(img, shape):
    Wy, Wx = _lin_weights(img.shape[0], shape[0], img.dtype), _lin_weights(img.shape[1], shape[1], img.dtype)
    t = np.tensordot(Wy, img, axes=(1, 0))                     # (y, X, c)
    return np.tensordot(Wx, t, axes=(1, 1)).transpose(1, 0, 2)  # (x, y, c) -> (y, x, c)


def resize_bilinear_backward(g, shape_in):
    Wy, Wx = _lin_weights(shape_in[0], g.shape[0], g.dtype), _lin_weights(shape_in[1], g.shape[1], g.dtype)
    t = np.tensordot(Wy.T, g, axes=(1, 0))                     # (Y, x, c)
    return np.tensordot(Wx.T, t, axes=(1, 1)).transpose(1, 0, 2)


# ---- smoothness term (normal_fusion.py:66-78, 127-131) ----------------------------------------------------------------
def _shift(img, di, dj):
    """get_neighbor_images' affine grid with nearest sampling and zero padding: out[y, x] = img[y + di, x + dj] or 0."""
    H, W = img.shape[:2]
    out = np.zeros_like(img)
    ys, xs = slice(max(0, -di), min(H, H - di)), slice(max(0, -dj), min(W, W - dj))
    out[ys, xs] = img[ys.start + di:ys.stop + di, xs.start + dj:xs.stop + dj]
    return out


def smooth_loss_and_grad(rot):
    M = rot.dtype.type(rot.size)
    loss, grad = rot.dtype.type(0.), np.zeros_like(rot)
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            if di == 0 and dj == 0: continue
            d = _shift(rot, di, dj) - rot
            loss = loss + (d * d).sum() / M
            grad += (-2 / M) * d                                 # through the "- rot" term
            grad += (2 / M) * _shift(d, -di, -dj)                # through the shifted copy: d[q] with q + (di,dj) = p
    return loss, grad


class _Adam:
    """torch.optim.Adam defaults (betas 0.9 / 0.999, eps 1e-8), in torch's order of operations."""

    def __init__(self, shape, lr, dt):
        self.m, self.v, self.t, self.lr, self.dt = np.zeros(shape, dt), np.zeros(shape, dt), 0, lr, dt

    def step(self, p, g):
        dt = self.dt
        self.t += 1
        self.m = self.m * dt(0.9) + g * dt(1 - 0.9)
        self.v = self.v * dt(0.999) + g * g * dt(1 - 0.999)
        bc1, bc2 = 1 - 0.9 ** self.t, 1 - 0.999 ** self.t
        denom = np.sqrt(self.v) / dt(np.sqrt(bc2)) + dt(1e-8)
        return p - dt(self.lr / bc1) * (self.m / denom)


def fusion_loss_and_grads(rot, src, tar, valid):
    """total_loss of one iteration (normal_fusion.py:120-133) and its gradients w.r.t. rot (64,64,3) and src (H,W,3)."""
    dt = rot.dtype
    up = resize_bilinear(rot, src.shape[:2])
    R = axis_angle_to_matrix(up)
    res = np.einsum('ijab,ijb->ija', R, src) - tar
    n = dt.type(3 * int(valid.sum()))
    if n > 0:
        w = valid[..., None].astype(dt)
        data = (res * res * w).sum() / n
        gres = res * w * (2 / n)
        g_src = np.einsum('ijab,ija->ijb', R, gres)
        g_rot = resize_bilinear_backward(axis_angle_to_matrix_backward(up, gres[..., :, None] * src[..., None, :]), rot.shape[:2])
    else:
        data, g_src, g_rot = dt.type(0.), np.zeros_like(src), np.zeros_like(rot)
    sl, sg = smooth_loss_and_grad(rot)
    return data + sl, g_rot + sg, g_src


def merge_normal_images(src_img, tar_img, iter_num, neck_xy, dt=np.float32, grid=64):
    """normal_fusion.py:89-155.  src_img: avatar normal map, tar_img: image-observed canonical normal map, both (H,W,3)."""
    dt = np.dtype(dt).type
    src = src_img.astype(dt); tar = tar_img.astype(dt)
    src_mask = np.sqrt((src * src).sum(-1)) > 0                                         # :100
    tar_mask = np.sqrt((tar * tar).sum(-1)) > 0                                         # :101
    er = erode3x3(tar_mask, 3)                                                          # :104-105
    dtm = distance_transform_l1(er).astype(dt)                                          # :106
    valid = src_mask & (er > 0)                                                         # :108-110
    init = src.copy()
    rot = np.zeros((grid, grid, 3), dt)                                                 # :114
    opt_rot, opt_src = _Adam(rot.shape, 1e-2, dt), _Adam(src.shape, 1e-1, dt)           # :117-118
    for it in range(iter_num):
        _, g_rot, g_src = fusion_loss_and_grads(rot, src, tar, valid)
        if it < iter_num / 2: rot = opt_rot.step(rot, g_rot)                            # :134-137
        else: src = opt_src.step(src, g_src)                                            # :138-141
    d = dtm[..., None] / dt(5.)                                                         # :144-145
    w0 = np.where(d > 1, dt(0.), dt(1.))                                                # :146-147
    src = (src * d + init * w0) / (d + w0)                                              # :148
    r0, c0, r1, c1 = neck_xy[1] - 90, neck_xy[0] - 35, neck_xy[1], neck_xy[0] + 35      # :151 (Python slice semantics, negatives wrap)
    src[r0:r1, c0:c1] = init[r0:r1, c0:c1]                                              # :152
    return src


def merge_normal_images_cover(src_img, tar_img):
    """normal_fusion.py:158-167"""
    out = src_img.copy()
    m = np.linalg.norm(tar_img, axis=-1) > 1e-6
    out[m] = tar_img[m]
    return out
