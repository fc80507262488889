"""Canonical normal fusion (reference normal_fusion/normal_fusion.py).  The oracle is held to goldens
produced by running the reference's own module (tests/golden/make_golden_fusion.py; OpenCV / pytorch3d / OpenGL calls stood
in for), its hand-written gradients to torch.autograd on a torch restatement of the same loss, its OpenCV stand-ins to
brute-force definitions -- and the HIP kernels to the oracle and to the same goldens."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import normal_fusion_oracle as nfo


def _aa2mat_torch(aa):
    """pytorch3d.transforms.axis_angle_to_matrix as published (axis_angle_to_quaternion + quaternion_to_matrix)."""
    angles = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    small = angles.abs() < 1e-6
    k = torch.empty_like(angles)
    k[~small] = torch.sin(half[~small]) / angles[~small]
    k[small] = 0.5 - (angles[small] * angles[small]) / 48
    q = torch.cat([torch.cos(half), aa * k], dim=-1)
    r, i, j, kk = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + kk * kk), two_s * (i * j - kk * r), two_s * (i * kk + j * r),
                     two_s * (i * j + kk * r), 1 - two_s * (i * i + kk * kk), two_s * (j * kk - i * r),
                     two_s * (i * kk - j * r), two_s * (j * kk + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def _loss_torch(rot, src, tar, valid):
    """One iteration's total_loss written with the reference's torch calls (normal_fusion.py:66-86, 120-133)."""
    H, W = src.shape[:2]
    theta = torch.tensor([[1, 0, 0], [0, 1, 0]], dtype=rot.dtype)
    grid = F.affine_grid(theta[None], torch.Size((1, 1, H, W)), align_corners=True)
    up = F.grid_sample(rot.permute(2, 0, 1)[None], grid, 'bilinear', 'border', True)[0].permute(1, 2, 0)
    R = _aa2mat_torch(up)
    data = torch.square(torch.einsum('ijab,ijb->ija', R, src) - tar)[valid].mean()
    gh, gw = rot.shape[:2]
    smooth = 0.
    for i in (-1, 0, 1):
        for j in (-1, 0, 1):
            if i == 0 and j == 0: continue
            th = torch.tensor([[1, 0, j / (gh / 2)], [0, 1, i / (gw / 2)]], dtype=rot.dtype)
            g = F.affine_grid(th[None], torch.Size((1, 1, gh, gw)), align_corners=True)
            nb = F.grid_sample(rot.permute(2, 0, 1)[None], g, mode='nearest', align_corners=True)[0].permute(1, 2, 0)
            smooth = smooth + torch.square(nb - rot).mean()
    return data + smooth


def _case(seed, H=96, grid=16):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:H]
    body = ((yy - H / 2) ** 2 / (0.42 * H) ** 2 + (xx - H / 2) ** 2 / (0.25 * H) ** 2) < 1
    n = rs.randn(H, H, 3); n[..., 2] += 2; n /= np.linalg.norm(n, axis=-1, keepdims=True)
    src = n * body[..., None]
    t = n + 0.3 * rs.randn(H, H, 3); t /= np.linalg.norm(t, axis=-1, keepdims=True)
    tar = t * (body & (xx > H * 0.3))[..., None]
    rot = 0.2 * rs.randn(grid, grid, 3); rot[:3] = 0                       # some exactly-zero rotations (norm's subgradient)
    return rot, src, tar


def test_gradients_match_autograd():
    rot, src, tar = _case(0)
    valid = (np.linalg.norm(src, axis=-1) > 0) & (np.linalg.norm(tar, axis=-1) > 0)
    loss, g_rot, g_src = nfo.fusion_loss_and_grads(rot.astype(np.float64), src.astype(np.float64), tar.astype(np.float64), valid)
    tr = torch.tensor(rot, dtype=torch.float64, requires_grad=True); ts = torch.tensor(src, dtype=torch.float64, requires_grad=True)
    with torch.enable_grad():                                   # (other test modules switch autograd off globally)
        lt = _loss_torch(tr, ts, torch.tensor(tar, dtype=torch.float64), torch.tensor(valid))
        lt.backward()
    lt = lt.detach()
    assert abs(float(lt) - float(loss)) < 1e-12
    assert np.abs(tr.grad.numpy() - g_rot).max() < 1e-12 * max(1, np.abs(g_rot).max()) + 1e-14
    assert np.abs(ts.grad.numpy() - g_src).max() < 1e-13
    assert np.abs(g_rot).max() > 1e-6 and np.abs(g_src).max() > 1e-6
    # rotation matrices themselves
    aa = np.concatenate([rot.reshape(-1, 3), np.zeros((2, 3)), np.float64([[1e-8, 0, 0], [3.0, -1.0, 0.5]])])
    assert np.abs(_aa2mat_torch(torch.tensor(aa)).numpy() - nfo.axis_angle_to_matrix(aa)).max() < 1e-14


def test_all_zero_rotation_start_and_no_valid_pixels():
    rot, src, tar = _case(1)
    z = np.zeros_like(rot)
    valid = (np.linalg.norm(src, axis=-1) > 0) & (np.linalg.norm(tar, axis=-1) > 0)
    _, g, _ = nfo.fusion_loss_and_grads(z, src, tar, valid)
    tr = torch.tensor(z, requires_grad=True)
    with torch.enable_grad():
        _loss_torch(tr, torch.tensor(src), torch.tensor(tar), torch.tensor(valid)).backward()
    assert np.isfinite(g).all() and np.abs(tr.grad.numpy() - g).max() < 1e-12     # the first Adam step of the reference starts here
    out = nfo.merge_normal_images(src, np.zeros_like(tar), 6, (40, 60), np.float64, grid=16)
    assert np.array_equal(out, src)                                                # nothing observed: the avatar map comes back


def test_opencv_stand_ins():
    rs = np.random.RandomState(3)
    m = rs.rand(40, 50) > 0.12
    m[10:30, 15:40] = True
    e = nfo.erode3x3(m, 3)
    pad = np.ones((46, 56), bool); pad[3:43, 3:53] = m
    brute = np.array([[pad[y:y + 7, x:x + 7].all() for x in range(50)] for y in range(40)])
    assert np.array_equal(e.astype(bool), brute) and e.dtype == np.uint8
    d = nfo.distance_transform_l1(e)
    zy, zx = np.nonzero(e == 0)
    yy, xx = np.mgrid[0:40, 0:50]
    ref = (np.abs(yy[..., None] - zy) + np.abs(xx[..., None] - zx)).min(-1)
    assert np.array_equal(d, ref.astype(np.float32)) and d.dtype == np.float32
    assert np.all(nfo.distance_transform_l1(np.ones((5, 5), np.uint8)) == nfo.DT_CAP)


def test_merge_runs_and_blends():
    rot, src, tar = _case(2)
    neck = (70, 50)                                                                # face rectangle rows [-40, 50) -> empty wrap, see below
    out = nfo.merge_normal_images(src, tar, 20, neck, np.float64, grid=16)
    out32 = nfo.merge_normal_images(src, tar, 20, neck, np.float32, grid=16)
    assert np.isfinite(out).all() and np.abs(out - out32).max() < 2e-3            # fp32 run stays near the fp64 run
    body = np.linalg.norm(src, axis=-1) > 0
    obs = nfo.erode3x3(np.linalg.norm(tar, axis=-1) > 0, 3) > 0
    assert np.array_equal(out[~obs], src[~obs])                                    # dt = 0 outside the eroded observation: avatar normal kept
    inner = (nfo.distance_transform_l1(obs.astype(np.uint8)) > 5) & body
    e_before = np.linalg.norm(src - tar, axis=-1)[inner].mean(); e_after = np.linalg.norm(out - tar, axis=-1)[inner].mean()
    assert e_after < 0.6 * e_before                                                 # the fused map moved towards the observation
    # the face rectangle [neck_y - 90, neck_y) x [neck_x - 35, neck_x + 35) follows the avatar, with Python's slice semantics:
    # (70, 50) gives rows [-40:50] = [56:50] = nothing on a 96-row image; (48, 95) gives rows [5:95], columns [13:83]
    face = nfo.merge_normal_images(src, tar, 20, (48, 95), np.float64, grid=16)
    assert np.array_equal(face[5:95, 13:83], src[5:95, 13:83]) and not np.array_equal(out[5:95, 13:83], src[5:95, 13:83])
    keep = np.ones(src.shape[:2], bool); keep[5:95, 13:83] = False
    assert np.array_equal(face[keep], out[keep])
    cov = nfo.merge_normal_images_cover(src, tar)
    m = np.linalg.norm(tar, axis=-1) > 1e-6
    assert np.array_equal(cov[m], tar[m]) and np.array_equal(cov[~m], src[~m])



def _within_measured_slack(d, slacks, what, factor=4.0, floors=(2e-6, 2e-4, 2e-3)):
    """Bounds from MEASURED slack instead of hand-set numbers (as tests/test_gpu_512.py does for colours).  `slacks` = one or more arrays
    |some fp32 evaluation - fp64 oracle| on the very same inputs: the fp32 run of the oracle itself and, where the goldens hold it, the REFERENCE's own
    fp32 run (torch autograd + torch.optim.Adam).  They say what the iterations lose in fp32: Adam's normalised step amplifies rounding wherever a
    gradient is near zero, so a handful of pixels drift by 1e-4 .. 1e-2 while the bulk agrees to 1e-7; the oracle's fp32 run shares the fp64 run's
    operation order and loses least, the reference's autograd run ~50x more.  Another fp32 evaluation with yet another summation order (the HIP
    kernels) may differ from fp64 by a small multiple of the largest measured slack, quantile by quantile: mean, 99.9 % and worst pixel."""
    slacks = slacks if isinstance(slacks, (list, tuple)) else [slacks]
    rows = []
    for (name, q), floor in zip((('mean', None), ('99.9 %', 0.999), ('max', 1.0)), floors):
        dv = float(d.mean() if q is None else np.quantile(d, q))
        sv = max(float(sl.mean() if q is None else np.quantile(sl, q)) for sl in slacks)
        rows.append((name, dv, sv, factor * sv + floor))
    print(what + ': ' + '; '.join('%s %.2e (measured fp32 slack %.2e, bound %.2e)' % r for r in rows))
    for name, dv, sv, bound in rows:
        assert dv <= bound, (what, name, dv, sv, bound)

# ---------------------------------------------------------------- the oracle against the reference's own code
@pytest.fixture(scope='module')
def fusion_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fusion_golden.npz'))


def test_oracle_merge_matches_reference_run(fusion_golden):
    """tests/golden/make_golden_fusion.py ran the reference's merge_normal_images (its autograd, its torch.optim.Adam, its
    resize / neighbour / blend / face-rectangle code; OpenCV and pytorch3d calls replaced by stand-ins) on these inputs."""
    _, src, tar = _case(5, H=512)
    src, tar = src.astype(np.float32), tar.astype(np.float32)
    for tag, iters, neck in (('b', 10, (300, 200)), ('a', 100, (-256, 150))):
        out = nfo.merge_normal_images(src, tar, iters, neck, np.float32)
        ref64 = nfo.merge_normal_images(src, tar, iters, neck, np.float64)
        # both are fp32 runs of the same Adam recursion with different summation orders (autograd vs hand-written): each is held to the fp64 oracle
        # within the measured fp32 slack, and so is their difference (twice: two fp32 runs)
        slack = np.abs(out - ref64)[::3, ::3]
        _within_measured_slack(np.abs(fusion_golden[f'G15_{tag}_lattice'] - ref64[::3, ::3]), slack, f'reference run {tag} vs fp64 oracle')
        ref_slack = np.abs(fusion_golden[f'G15_{tag}_lattice'] - ref64[::3, ::3])
        _within_measured_slack(np.abs(out[::3, ::3] - fusion_golden[f'G15_{tag}_lattice']), [slack, ref_slack], f'fp32 oracle vs reference run {tag}', factor=2.0)
        assert abs(out.astype(np.float64).sum() - fusion_golden[f'G15_{tag}_checksum'][0]) < 2e-3 * fusion_golden[f'G15_{tag}_checksum'][1] ** 0.5 + 1
    assert np.array_equal(nfo.merge_normal_images_cover(src, tar)[::3, ::3], fusion_golden['G15_cover_lattice'])


def test_oracle_canonicalize_matches_reference_run(fusion_golden):
    """The reference's canonicalize_normal_map ran with its two Renderer objects on a REAL OpenGL implementation (Mesa llvmpipe,
    tests/golden/make_golden_gl.py: MesaRenderer); here the dedicated pieces (per-vertex restatement + the orthographic front / back
    rasteriser of the oracle) must reproduce its images: same pixels, values to 1e-4 (measured: worst 3.8e-5, mean 3e-8)."""
    from oracle import raster
    s = _scene()
    pos = raster.render_mesh(s['live'], None, s['f'], s['mvp'], s['W'], s['H'])
    n = nfo.canonicalize_vertex_normals(s['live'], s['M'], pos, s['obs'], s['mv'], s['fx'], s['fy'], s['cx'], s['cy'])
    fr, bk = raster.render_cano_mesh(s['v'], n, s['f'], np.float32([0.01, -0.02, 0.0]), 512)
    for img, key in ((fr, 'G16_front_lattice'), (bk, 'G16_back_lattice')):
        g = fusion_golden[key]
        assert np.array_equal(np.linalg.norm(img[::3, ::3], axis=-1) > 0, np.linalg.norm(g, axis=-1) > 0)      # same pixels drawn
        d = np.abs(img[::3, ::3] - g)
        assert d.max() < 1e-4 and d.mean() < 1e-6
    assert abs((np.linalg.norm(fr, axis=-1) > 0).mean() - fusion_golden['G16_cover'][0]) < 1e-6


# ---------------------------------------------------------------- GPU: HIP kernels against the oracle
def _scene():
    """A sphere in the canonical pose, skinned by smoothly varying per-vertex matrices, seen by a pinhole camera; the
    'observed' normal map is rendered from the posed mesh in the convention the reference undoes (camera frame, y and z negated)."""
    from oracle import raster
    from test_raster import _sphere_mesh
    from avatarcap_amd.utils.renderer import gl_perspective_projection_matrix
    v, f, n = _sphere_mesh(44, 0.55)
    rs = np.random.RandomState(4)

    def rotm(w):
        th = np.linalg.norm(w); k = w / th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    R0, R1 = rotm(np.array([0.2, -0.3, 0.1])), rotm(np.array([-0.1, 0.4, 0.3]))
    a = (0.5 + 0.5 * v[:, 1:2] / 0.55)[..., None]                                   # blend weight varies with height
    A = (1 - a) * R0 + a * R1
    M = np.tile(np.eye(4, dtype=np.float32), (v.shape[0], 1, 1)); M[:, :3, :3] = A; M[:, :3, 3] = np.float32([0.05, -0.1, 0.02])
    live = np.einsum('vij,vj->vi', A, v) + M[:, :3, 3]
    live_n = np.einsum('vij,vj->vi', A, n)
    W, H, fo = 400, 300, 380.0
    mv = np.eye(4, dtype=np.float32); mv[:3, :3] = rotm(np.array([0.05, 0.1, -0.02])); mv[:3, 3] = [0.02, 0.03, 2.6]
    mvp = gl_perspective_projection_matrix(fo, fo, W / 2 + 3, H / 2 - 2, W, H) @ mv
    ncam = live_n @ mv[:3, :3].T
    obs = raster.render_mesh(live, ncam * np.float32([1, -1, -1]), f, mvp, W, H)[..., :3].copy()
    obs[:, :120] = 0                                                                # part of the image is unobserved
    return dict(v=v.astype(np.float32), f=f, n=n.astype(np.float32), M=M.astype(np.float32), live=live.astype(np.float32), obs=obs.astype(np.float32),
                mv=mv, fx=fo, fy=fo, cx=W / 2 + 3, cy=H / 2 - 2, W=W, H=H, mvp=mvp)


@pytest.mark.gpu
def test_hip_canonicalize_matches_oracle_and_recovers_canonical_normals():
    from oracle import raster
    from avatarcap_amd import config, _lib
    from avatarcap_amd.normal_fusion.normal_fusion import canonicalize_normal_map, canonicalize_normal_map_device
    config.device = torch.device('cuda')
    s = _scene()
    pos = raster.render_mesh(s['live'], None, s['f'], s['mvp'], s['W'], s['H'])
    ref = nfo.canonicalize_vertex_normals(s['live'], s['M'], pos, s['obs'], s['mv'], s['fx'], s['fy'], s['cx'], s['cy'])
    seen = np.linalg.norm(ref, axis=1) > 0
    assert 0.2 < seen.mean() < 0.5                                                    # the near side, minus the unobserved strip
    err = np.abs(ref[seen] - s['n'][seen])                                            # observed normals come back in the canonical pose
    assert err.max() < 0.2 and err.mean() < 0.02                                      # (nearest-pixel sampling: worst near the silhouette)
    t = lambda x, d=torch.float32: torch.from_numpy(np.ascontiguousarray(x)).to('cuda', d)
    out = torch.empty((s['v'].shape[0], 3), device='cuda')
    d_live, d_M, d_pos, d_obs = t(s['live']), t(s['M']), t(pos), t(s['obs'])            # (kept alive across the call)
    _lib.check(_lib.lib().avc_canonicalize_normals(_lib.ctx(out.device), d_live.data_ptr(), d_M.data_ptr(), s['v'].shape[0], d_pos.data_ptr(),
                                                   d_obs.data_ptr(), s['H'], s['W'], _lib.f3(s['mv'].reshape(16)), s['fx'], s['fy'], s['cx'], s['cy'],
                                                   out.data_ptr(), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    same = (np.linalg.norm(got, axis=1) > 0) == seen
    assert same.mean() > 0.999                                                        # visibility decisions (|v - p| < 0.05) agree except at the threshold
    assert np.abs(got[same & seen] - ref[same & seen]).max() < 2e-5
    c = np.float32([0.01, -0.02, 0.0])
    fr, bk = canonicalize_normal_map(None, None, s['v'], s['live'], s['f'], s['obs'], torch.from_numpy(s['M']), s['mv'], s['fx'], s['fy'], s['cx'], s['cy'], c)
    ofr, obk = raster.render_cano_mesh(s['v'], got, s['f'], c, 512)
    assert fr.shape == (512, 512, 3) and np.array_equal(fr, ofr) and np.array_equal(bk, obk)


@pytest.mark.gpu
@pytest.mark.parametrize('size,iters,neck', [(128, 30, (64, 100)), (512, 100, (-256, 150))])
def test_hip_merge_normal_images_matches_oracle(size, iters, neck, fusion_golden):
    from avatarcap_amd import config
    from avatarcap_amd.normal_fusion.normal_fusion import merge_normal_images, merge_normal_images_cover
    config.device = torch.device('cuda')
    rot, src, tar = _case(5, H=size)
    src, tar = src.astype(np.float32), tar.astype(np.float32)
    ref = nfo.merge_normal_images(src, tar, iters, neck, np.float64)
    out = merge_normal_images(src, tar, iters, neck)
    assert out.dtype == np.float32 and out.shape == src.shape
    # Yardsticks, all measured on these very inputs: (1) the oracle's own fp32 run against its fp64 run -- it shares the fp64 run's operation order and so
    # understates what ANOTHER order loses (the reference's autograd run deviates 50x more in the mean); (2) how the recursion amplifies rounding-sized
    # noise wherever it enters: the fp64 oracle on inputs perturbed by a few fp32 ulps (2^-20 relative) -- order-independent, worst pixel included
    # (1.4e-3 at 512^2 / 100 iterations: Adam's normalised step is chaotic where a gradient is near zero); (3) at 512^2 the REFERENCE's own fp32 run.
    rs = np.random.RandomState(size)
    noise = lambda a: a.astype(np.float64) * (1.0 + 2.0 ** -20 * rs.uniform(-1, 1, a.shape))      # noqa: E731
    cond = np.abs(nfo.merge_normal_images(noise(src), noise(tar), iters, neck, np.float64) - ref)
    if size == 512:                                                                   # the reference's own run of these very inputs (lattice of every third pixel)
        ref_slack = np.abs(fusion_golden['G15_a_lattice'] - ref[::3, ::3])
        _within_measured_slack(np.abs(out - ref)[::3, ::3], [cond[::3, ::3], ref_slack], 'HIP vs fp64 oracle on the lattice (slacks: 8-ulp conditioning, reference run)')
        _within_measured_slack(np.abs(out[::3, ::3] - fusion_golden['G15_a_lattice']), [cond[::3, ::3], ref_slack], 'HIP vs the reference run', factor=8.0)
        _within_measured_slack(np.abs(out - ref), [cond], f'HIP vs fp64 oracle ({size}^2, {iters} iterations)')
    else:
        slack = np.abs(nfo.merge_normal_images(src, tar, iters, neck, np.float32) - ref)
        _within_measured_slack(np.abs(out - ref), [slack, cond], f'HIP vs fp64 oracle ({size}^2, {iters} iterations)')
    obs = nfo.erode3x3(np.linalg.norm(tar, axis=-1) > 0, 3) > 0
    assert np.array_equal(out[~obs], src[~obs])                                       # erosion / distance transform agree exactly
    assert np.array_equal(merge_normal_images(src, tar, iters, neck), out)            # deterministic
    assert np.array_equal(merge_normal_images_cover(src, tar), nfo.merge_normal_images_cover(src, tar))
    assert np.array_equal(merge_normal_images(src, np.zeros_like(tar), 4, neck), src)


@pytest.mark.gpu
def test_hip_merge_non_square_ragged_image():
    """Rows and columns that are no multiple of anything the kernels tile by (wave-wide row segments, 64-column blocks, 16 row segments of the
    distance transform): a 93 x 75 crop, HIP vs the oracle; the distance-transform blend outside the observed region is exact."""
    from avatarcap_amd import config
    from avatarcap_amd.normal_fusion.normal_fusion import merge_normal_images
    config.device = torch.device('cuda')
    _, src, tar = _case(9, H=96)
    src, tar = np.ascontiguousarray(src[2:95, 11:86], np.float32), np.ascontiguousarray(tar[2:95, 11:86], np.float32)
    assert src.shape == (93, 75, 3)
    ref = nfo.merge_normal_images(src, tar, 12, (30, 70), np.float64)
    out = merge_normal_images(src, tar, 12, (30, 70))
    _within_measured_slack(np.abs(out - ref), np.abs(nfo.merge_normal_images(src, tar, 12, (30, 70), np.float32) - ref), 'HIP vs fp64 oracle (93 x 75)')
    obs = nfo.erode3x3(np.linalg.norm(tar, axis=-1) > 0, 3) > 0
    assert np.array_equal(out[~obs], src[~obs])
    # no observed pixel at all / every pixel observed: the distance transform's two saturated ends
    assert np.array_equal(merge_normal_images(src, np.zeros_like(tar), 3, (30, 70)), src)
    full = np.ascontiguousarray(np.where(np.linalg.norm(tar, axis=-1, keepdims=True) > 0, tar, np.float32([0, 0, 1])), np.float32)
    o2, r2 = merge_normal_images(src, full, 3, (30, 70)), nfo.merge_normal_images(src, full, 3, (30, 70), np.float64)
    _within_measured_slack(np.abs(o2 - r2), np.abs(nfo.merge_normal_images(src, full, 3, (30, 70), np.float32) - r2), 'HIP vs fp64 oracle (every pixel observed)')
