"""Orthographic normal-map rasteriser and general MVP view: the CPU oracle against images a REAL OpenGL
implementation produced for the reference's own render_cano_mesh / Renderer call sequence (Mesa llvmpipe, tests/golden/gl_golden.npz),
analytic checks of the oracle, bit-exact agreement of the HIP kernels with it and the HIP kernels against the OpenGL images directly (GPU)."""
import os
import numpy as np
import pytest

from oracle import mc, raster


def _sphere_mesh(n=40, R=0.6):
    g = np.linspace(-1, 1, n, dtype=np.float32)
    x, y, z = np.meshgrid(g, g, g, indexing='ij')
    vol = (R - np.sqrt(x * x + y * y + z * z)).astype(np.float32)
    h = 2.0 / (n - 1)
    v, f = mc.marching_cubes(vol, 0.0, [h, h, h])
    v = v - 1.0
    return v.astype(np.float32), f[:, [2, 1, 0]].copy(), (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def test_oracle_on_sphere():
    v, f, nrm = _sphere_mesh()
    fr, bk = raster.render_cano_mesh(v, nrm, f, np.zeros(3, np.float32), 256)
    m = np.linalg.norm(fr, axis=-1) > 0
    assert abs(m.sum() / (np.pi * (0.6 * 128) ** 2) - 1) < 0.01                       # silhouette area
    rr, cc = np.nonzero(m)
    xs, ys = (cc + 0.5) / 128 - 1, 1 - (rr + 0.5) / 128                             # row 0 is y = +1
    inner = np.sqrt(xs * xs + ys * ys) < 0.5
    exp = np.stack([xs, ys, np.sqrt(np.maximum(0.36 - xs * xs - ys * ys, 0))], -1) / 0.6
    assert np.abs(fr[m] - exp)[inner].max() < 0.01                                    # front map = outward normal of the near side
    mb = np.linalg.norm(bk, axis=-1) > 0
    assert np.array_equal(m, mb)                                                      # maps are pixel-aligned (back is mirrored back)
    assert np.all(fr[m][:, 2] > 0) and np.all(bk[mb][:, 2] < 0)                       # back map shows the far side, normals not rotated
    assert np.abs(bk[mb][:, :2] - fr[m][:, :2])[inner].max() < 0.02
    # translation by -center
    fr2, _ = raster.render_cano_mesh(v + np.float32([0.1, -0.2, 0.3]), nrm, f, np.float32([0.1, -0.2, 0.3]), 256)
    assert np.abs(fr2 - fr).max() < 1e-5


def test_oracle_depth_and_culling():
    # two parallel quads facing +z at z = 0 and z = 0.5: the front map must show the nearer (z = 0.5) one
    q = np.float32([[-0.5, -0.5, 0], [0.5, -0.5, 0], [0.5, 0.5, 0], [-0.5, 0.5, 0]])
    v = np.concatenate([q, q + np.float32([0, 0, 0.5])])
    f = np.int32([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7]])                       # CCW seen from +z
    a = np.concatenate([np.tile(np.float32([[1, 0, 0]]), (4, 1)), np.tile(np.float32([[0, 1, 0]]), (4, 1))])
    fr, bk = raster.render_cano_mesh(v, a, f, np.zeros(3, np.float32), 64)
    assert np.allclose(fr[32, 32], [0, 1, 0]) and np.allclose(bk[32, 32], 0)          # back view culls +z-facing faces
    fr, bk = raster.render_cano_mesh(v, a, f[:, [2, 1, 0]].copy(), np.zeros(3, np.float32), 64)
    assert np.allclose(fr[32, 32], 0) and np.allclose(bk[32, 32], [1, 0, 0])          # flipped: visible from behind, nearest there is z = 0
    assert np.count_nonzero(np.linalg.norm(bk, axis=-1)) == 32 * 32                   # exact coverage of a pixel-aligned square (top-left rule)


@pytest.mark.gpu
@pytest.mark.parametrize('size', [64, 512])
def test_hip_rasteriser_matches_oracle(size):
    import torch
    from avatarcap_amd.utils.visualize_util import render_cano_mesh_device
    v, f, nrm = _sphere_mesh(48)
    rs = np.random.RandomState(0)
    v2 = np.concatenate([v, (0.4 * v + np.float32([0.3, 0.2, 0.4]))]).astype(np.float32)          # a second blob partly in front
    f2 = np.concatenate([f, f + v.shape[0]]).astype(np.int32)
    a2 = np.concatenate([nrm, rs.randn(*nrm.shape).astype(np.float32)])
    c = np.float32([0.05, -0.03, 0.1])
    ofr, obk = raster.render_cano_mesh(v2, a2, f2, c, size)
    fr, bk = render_cano_mesh_device(torch.from_numpy(v2).cuda(), torch.from_numpy(a2).cuda(), torch.from_numpy(f2).cuda(), c, size)
    assert np.array_equal(fr.cpu().numpy(), ofr)
    assert np.array_equal(bk.cpu().numpy(), obk)
    e, _ = render_cano_mesh_device(torch.zeros(3, 3).cuda(), torch.zeros(3, 3).cuda(), torch.zeros((0, 3), dtype=torch.int32).cuda(), c, 32)
    assert float(e.abs().max()) == 0.0


def _sliver_scene(n=6000, size=64, seed=3):
    """Needles and slivers: nearly collinear corner triples whose corners sit within a sub-pixel step of pixel-centre rows / columns -- the 1/256-pixel
    snap of the coverage test and the unsnapped positions of the attribute planes then disagree about the triangle's area, down to its sign."""
    rs = np.random.RandomState(seed)
    px = 2.0 / size
    a = rs.uniform(-0.9, 0.9, (n, 2))
    d = rs.uniform(-1, 1, (n, 2)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    L = rs.uniform(0.5, 12, (n, 1)) * px
    axis_aligned = rs.rand(n) < 0.6
    d[axis_aligned] = np.where(rs.rand(axis_aligned.sum(), 1) < 0.5, [[1.0, 0.0]], [[0.0, 1.0]])
    a[axis_aligned] = (np.floor(a[axis_aligned] / px) + 0.5) * px                  # ... through pixel centres
    nrm = np.stack([-d[:, 1], d[:, 0]], 1)
    eps = rs.uniform(-1.5, 1.5, (n, 3, 1)) * px / 256                                # corner offsets across the needle: a sub-pixel step or so
    t = np.stack([np.zeros((n, 1)), 0.5 * L * rs.uniform(0.8, 1.2, (n, 1)), L], 1)  # along it
    xy = a[:, None, :] + t * d[:, None, :] + eps * nrm[:, None, :]
    z = rs.uniform(-0.3, 0.3, (n, 3, 1))
    v = np.concatenate([xy, z], 2).reshape(-1, 3).astype(np.float32)
    f = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    attr = rs.uniform(-1, 1, (3 * n, 3)).astype(np.float32)
    return v, f, attr


def test_oracle_slivers_interpolate_inside_their_corners():
    """attributes are interpolated with the barycentrics of the UNSNAPPED corners while coverage comes from the snapped ones; a sliver
    that wins a pixel with a near-zero / opposite-sign unsnapped area must not extrapolate (or emit inf / NaN): every covered pixel's value lies
    within its triangle's corner values (weights clamped to [0, 1] and renormalised for such triangles, raster_oracle.c: bary_guard)."""
    v, f, attr = _sliver_scene()
    hit = 0
    for view_img in raster.render_cano_mesh(v, attr, f, np.zeros(3, np.float32), 64):
        assert np.isfinite(view_img).all()
        hit += int((np.abs(view_img).sum(-1) > 0).sum())
    assert hit > 200                                                                 # the scene does cover pixels
    # every triangle on its own (no depth competition): the pixel values against the corner range
    worst = 0.0
    for t in range(0, f.shape[0], 7):
        fr, bk = raster.render_cano_mesh(v, attr, f[t:t + 1], np.zeros(3, np.float32), 64)
        lo, hi = attr[f[t]].min(0), attr[f[t]].max(0)
        for img in (fr, bk):
            m = np.abs(img).sum(-1) > 0
            if m.any():
                worst = max(worst, float(np.maximum(lo - img[m], img[m] - hi).max()))
    span = 2.0
    assert worst < 0.51 * span, worst            # ordinary near-edge overshoot is allowed up to the guard's threshold (weights in [-0.5, 1.5]) ...
    M, _ = _pinhole(64, 64, 80.0, 2.0)
    im = raster.render_mesh(v, attr, f, M, 64, 64)
    assert np.isfinite(im).all() and float(np.abs(im[..., :3]).max()) <= 1.0 + 0.51 * span


@pytest.mark.gpu
def test_hip_rasterisers_match_oracle_on_slivers():
    import torch
    from avatarcap_amd.utils.visualize_util import render_cano_mesh_device
    from avatarcap_amd.utils.renderer import render_mesh_device
    v, f, attr = _sliver_scene()
    c = np.zeros(3, np.float32)
    ofr, obk = raster.render_cano_mesh(v, attr, f, c, 64)
    fr, bk = render_cano_mesh_device(torch.from_numpy(v).cuda(), torch.from_numpy(attr).cuda(), torch.from_numpy(f).cuda(), c, 64)
    assert np.array_equal(fr.cpu().numpy(), ofr) and np.array_equal(bk.cpu().numpy(), obk)
    M, _ = _pinhole(64, 64, 80.0, 2.0)
    om = raster.render_mesh(v, attr, f, M, 64, 64)
    gm = render_mesh_device(torch.from_numpy(v).cuda(), torch.from_numpy(attr).cuda(), torch.from_numpy(f).cuda(), M, 64, 64)
    assert np.array_equal(gm.cpu().numpy(), om)


def _pinhole(W, H, f, tz):
    from avatarcap_amd.utils.renderer import gl_perspective_projection_matrix
    mv = np.eye(4, dtype=np.float32); mv[2, 3] = tz
    return gl_perspective_projection_matrix(f, f, W / 2, H / 2, W, H) @ mv, mv


def test_oracle_perspective_position_map():
    """'position' shader through gl_perspective_projection_matrix (normal_fusion.py:14-20): every covered pixel
    holds the surface point that projects onto it, the near side wins, inside-out meshes show their far side."""
    v, f, nrm = _sphere_mesh()
    W, H, fo = 320, 240, 300.0
    mvp, mv = _pinhole(W, H, fo, 3.0)
    img = raster.render_mesh(v, None, f, mvp, W, H)
    m = img[..., 3] > 0
    assert np.array_equal(np.unique(img[..., 3]), [0.0, 1.0]) and np.all(img[~m] == 0)
    assert abs(m.sum() / (np.pi * (0.6 / np.sqrt(9 - 0.36) * fo) ** 2) - 1) < 0.01
    rr, cc = np.nonzero(m)
    p = img[m][:, :3]; pc = p @ mv[:3, :3].T + mv[:3, 3]
    assert np.abs(pc[:, 0] / pc[:, 2] * fo + W / 2 - (cc + 0.5)).max() < 0.01      # perspective-correct interpolation
    assert np.abs(pc[:, 1] / pc[:, 2] * fo + H / 2 - (rr + 0.5)).max() < 0.01
    assert np.all(p[:, 2] < 0) and np.abs(np.linalg.norm(p, axis=1) - 0.6).max() < 3e-3
    far = raster.render_mesh(v, None, f[:, ::-1].copy(), mvp, W, H)
    assert np.mean(far[far[..., 3] > 0][:, 2] > 0) > 0.9
    behind = raster.render_mesh(v, None, f, _pinhole(W, H, fo, -3.0)[0], W, H)       # camera inside-out: w <= 0 -> nothing
    assert not behind.any()
    att = raster.render_mesh(v, nrm, f, mvp, W, H)                                    # 'vertex_attribute' shader
    assert np.abs(att[m][:, :3] - p / 0.6).max() < 0.02


def test_oracle_orthographic_mvp_equals_cano_front_view():
    """The general view with the reference's orthographic front matrix (visualize_util.py:15-22) reproduces the
    dedicated front map."""
    from avatarcap_amd.utils.renderer import gl_orthographic_projection_matrix
    v, f, nrm = _sphere_mesh()
    c = np.float32([0.05, -0.02, 0.1])
    model = np.eye(4, dtype=np.float32); model[:3, 3] = -c; model[2, 3] -= 10
    img = raster.render_mesh(v, nrm, f, gl_orthographic_projection_matrix() @ model, 256, 256)
    fr, _ = raster.render_cano_mesh(v, nrm, f, c, 256)
    assert np.array_equal(img[..., 3] > 0, np.linalg.norm(fr, axis=-1) > 0)
    assert np.abs(img[..., :3] - fr).max() < 1e-5


@pytest.mark.gpu
def test_hip_mvp_rasteriser_matches_oracle():
    import torch
    from avatarcap_amd.utils.renderer import Renderer, render_mesh_device
    from avatarcap_amd import config
    config.device = torch.device('cuda')
    v, f, nrm = _sphere_mesh(48)
    v2 = np.concatenate([v, 0.4 * v + np.float32([0.3, 0.2, -0.5])]).astype(np.float32)
    f2 = np.concatenate([f, f + v.shape[0]]).astype(np.int32)
    a2 = np.concatenate([nrm, -nrm]).astype(np.float32)
    for (W, H, fo, tz) in ((320, 240, 300.0, 3.0), (512, 512, 900.0, 2.0), (64, 48, 40.0, 0.9)):     # the last one: camera close, parts off screen
        mvp, _ = _pinhole(W, H, fo, tz)
        for attrs in (None, a2):
            o = raster.render_mesh(v2, attrs, f2, mvp, W, H)
            d = render_mesh_device(torch.from_numpy(v2).cuda(), None if attrs is None else torch.from_numpy(attrs).cuda(),
                                   torch.from_numpy(f2).cuda(), mvp, W, H)
            assert np.array_equal(d.cpu().numpy(), o)
    r = Renderer(320, 240, shader_name='position')                                   # reference call surface, triangle soup
    r.set_model(v2[f2.reshape(-1)]); r.set_mvp_mat(_pinhole(320, 240, 300.0, 3.0)[0])
    assert np.array_equal(r.render(), raster.render_mesh(v2, None, f2, _pinhole(320, 240, 300.0, 3.0)[0], 320, 240))
    with pytest.raises(ValueError):
        Renderer(8, 8, shader_name='phong_color')


# ---------------------------------------------------------------- against a real OpenGL implementation
_GL = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'gl_golden.npz'))


def _check_against_gl(img, mask_bits, lattice, step, what):
    H, W = img.shape[:2]
    gm = np.unpackbits(mask_bits)[:H * W].reshape(H, W).astype(bool)
    om = (np.linalg.norm(img[..., :3], axis=-1) > 0) if img.shape[-1] == 3 else (img[..., 3] > 0)
    assert np.array_equal(gm, om), f'{what}: {int((gm != om).sum())} pixels covered differently from OpenGL'
    d = np.abs(img[::step, ::step] - lattice)[gm[::step, ::step]]
    assert d.mean() < 1e-6 and d.max() < 1e-4, (what, float(d.mean()), float(d.max()))     # measured: mean 1e-7, worst 5e-5 (a grazing triangle)
    return float(d.max())


def test_oracle_matches_opengl():
    """Coverage identical to Mesa llvmpipe pixel for pixel (fill rule, sub-pixel snapping, culling, depth test, the back view's rotation +
    flip), interpolated normals / positions to ~1e-7: the reference's render_cano_mesh and its 'position' render on a real GL."""
    import gl_scenes as sc
    from avatarcap_amd.utils.renderer import gl_perspective_projection_matrix
    assert 'llvmpipe' in str(_GL['gl_info']) or 'Mesa' in str(_GL['gl_info'])
    for name in sc.CANO_SCENES:
        v, f, n, c, size = sc.cano_scene(name)
        fr, bk = raster.render_cano_mesh(v, n, f, c, size)
        _check_against_gl(fr, _GL[f'{name}_front_mask'], _GL[f'{name}_front_lattice'], 4, name + ' front')
        _check_against_gl(bk, _GL[f'{name}_back_mask'], _GL[f'{name}_back_lattice'], 4, name + ' back')
    v, f, mv, fx, fy, cx, cy, W, H = sc.position_scene()
    pos = raster.render_mesh(v, None, f, gl_perspective_projection_matrix(fx, fy, cx, cy, W, H) @ mv, W, H)
    _check_against_gl(pos, _GL['position_mask'], _GL['position_lattice'], 3, 'position')


@pytest.mark.gpu
def test_hip_rasterisers_match_opengl():
    import torch
    import gl_scenes as sc
    from avatarcap_amd.utils.renderer import gl_perspective_projection_matrix, render_mesh_device
    from avatarcap_amd.utils.visualize_util import render_cano_mesh_device
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()                      # noqa: E731
    for name in sc.CANO_SCENES:
        v, f, n, c, size = sc.cano_scene(name)
        fr, bk = render_cano_mesh_device(t(v), t(n), t(f), c, size)
        _check_against_gl(fr.cpu().numpy(), _GL[f'{name}_front_mask'], _GL[f'{name}_front_lattice'], 4, name + ' front (HIP)')
        _check_against_gl(bk.cpu().numpy(), _GL[f'{name}_back_mask'], _GL[f'{name}_back_lattice'], 4, name + ' back (HIP)')
    v, f, mv, fx, fy, cx, cy, W, H = sc.position_scene()
    pos = render_mesh_device(t(v), None, t(f), gl_perspective_projection_matrix(fx, fy, cx, cy, W, H) @ mv, W, H)
    _check_against_gl(pos.cpu().numpy(), _GL['position_mask'], _GL['position_lattice'], 3, 'position (HIP)')


@pytest.mark.gpu
def test_hip_rasterisers_fuzz_against_the_oracle(monkeypatch):
    """tests/tools/raster_fuzz_gpu.py, 120 cases: random triangle soups (blobs, image-wide triangles, slivers, lattice vertices with edges through pixel centres and
    coincident depths, zero-area and repeated triangles, shared vertices), random image sizes and pinhole cameras including ones inside the soup; the canonical front /
    back views and the MVP view with and without attributes, every pixel bit for bit.  (The round's campaign: 1,200 cases, 56 M pixels, no mismatch.)"""
    import importlib.util
    import sys
    import torch
    from avatarcap_amd import config
    spec = importlib.util.spec_from_file_location('raster_fuzz_gpu', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'raster_fuzz_gpu.py'))
    mod = importlib.util.module_from_spec(spec)
    dev, cfg = config.device, config.cfg
    try:
        spec.loader.exec_module(mod)
        monkeypatch.setattr(sys, 'argv', ['raster_fuzz_gpu.py', '120', '3'])
        assert mod.main() == 0
    finally:
        config.device, config.cfg = dev, cfg
