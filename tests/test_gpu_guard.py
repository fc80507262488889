"""Guard-band check of the C-ABI's device kernels at ragged sizes (sanitizers; the AddressSanitizer flavour of the library
builds -- `python -m avatarcap_amd.build --asan` -- but cannot be driven on this image: tools/sanitize/README.md).

Every OUTPUT of a call lives in the middle of a larger allocation whose margins carry a bit pattern that no kernel produces; every INPUT is framed by
NaNs.  After the call the margins must be untouched (an out-of-bounds WRITE would change them) and every output finite (an out-of-bounds READ of an
input's frame would poison it).  The sizes are the awkward ones: 0, 1, one short of / one past the tile and wave sizes, non-cubic volumes, images
that do not fill their tiles.  Values are not checked here -- the parity tests do that -- only where the kernels read and write."""
import ctypes as C

import numpy as np
import pytest
import torch

import golden_inputs as gi
from avatarcap_amd import _lib, config, synthetic as syn

pytestmark = pytest.mark.gpu
CANARY = 0x7FA5C3E1             # a NaN payload no arithmetic produces
MARGIN = 4096                   # 32-bit words on either side


class Guarded:
    """A device buffer of `n` 32-bit words (viewed as `dtype`) between two margins of CANARY words."""

    def __init__(self, n, dtype=torch.float32, fill=None):
        self.n = int(n)
        self.raw = torch.full((self.n + 2 * MARGIN,), CANARY, dtype=torch.int32, device='cuda')
        self.t = self.raw[MARGIN:MARGIN + self.n].view(dtype)
        if fill is not None and self.n:
            self.t.copy_(fill.reshape(-1).view(dtype) if fill.dtype != dtype else fill.reshape(-1))

    @property
    def ptr(self):
        return self.t.data_ptr()

    def intact(self):
        return bool((self.raw[:MARGIN] == CANARY).all()) and bool((self.raw[MARGIN + self.n:] == CANARY).all())


def _framed(x):
    """An input tensor framed by NaNs (float) -- reading past its ends shows up as a non-finite output."""
    return Guarded(x.numel(), x.dtype, fill=x.contiguous())


def _ok(*bufs, finite=()):
    torch.cuda.synchronize()
    for b in bufs:
        assert b.intact(), 'a margin was written'
    for b in finite:
        assert bool(torch.isfinite(b.t).all()), 'an output is not finite (an input was read out of bounds?)'


@pytest.mark.parametrize('nr', [1, 5, 63, 6890])
@pytest.mark.parametrize('nq', [0, 1, 63, 65, 257, 4099])
def test_knn_lbs_skinning_stay_inside(nq, nr):
    L, ctx = _lib.lib(), _lib.ctx(0)
    g = torch.Generator().manual_seed(nq * 7 + nr)
    q = _framed(torch.rand(nq, 3, generator=g).cuda() * 2 - 1)
    ref = _framed(torch.rand(nr, 3, generator=g).cuda() * 2 - 1)
    sw = _framed(torch.rand(nr, 24, generator=g).cuda())
    jm = _framed(torch.rand(24, 4, 4, generator=g).cuda())
    for K in (1, 4):
        if K > nr:
            continue
        d2, idx = Guarded(nq * K), Guarded(2 * nq * K)                  # int64 indices: two words each
        _lib.check(L.avc_knn(ctx, q.ptr, nq, ref.ptr, nr, K, d2.ptr, idx.ptr, None))
        _ok(d2, idx, q, ref, finite=(d2,))
        if nq:
            assert int(idx.t.view(torch.int64).max()) < nr and int(idx.t.view(torch.int64).min()) >= 0
    if nr >= 4:
        lbs, po, no, mo = Guarded(nq * 24), Guarded(nq * 3), Guarded(nq * 3), Guarded(nq * 16)
        _lib.check(L.avc_calculate_lbs(ctx, q.ptr, nq, ref.ptr, sw.ptr, nr, lbs.ptr, None))
        _lib.check(L.avc_skinning(ctx, q.ptr, q.ptr, nq, lbs.ptr, jm.ptr, po.ptr, no.ptr, mo.ptr, None))
        _ok(lbs, po, no, mo, q, ref, sw, jm, finite=(lbs, po, no, mo))
        # the bound form (round 5: avc_lbs_prepare builds the vertices' grid and per-cell candidate lists once): same outputs, nothing written elsewhere
        lbs2 = Guarded(nq * 24)
        _lib.check(L.avc_lbs_prepare(ctx, ref.ptr, nr, None))
        _lib.set_owner(ctx, 'lbs_bound', None)                             # (whatever SmplUtil bound before is gone: its token no longer owns the slot)
        _lib.check(L.avc_calculate_lbs_bound(ctx, q.ptr, nq, sw.ptr, lbs2.ptr, None))
        _ok(lbs2, q, ref, sw, finite=(lbs2,))
        assert torch.equal(lbs2.t, lbs.t)
        # the fused launch (round 6: avc_lbs_skin_bound): the same four outputs, nothing written elsewhere
        lbs3, po3, no3, mo3 = Guarded(nq * 24), Guarded(nq * 3), Guarded(nq * 3), Guarded(nq * 16)
        _lib.check(L.avc_lbs_skin_bound(ctx, q.ptr, q.ptr, nq, sw.ptr, jm.ptr, lbs3.ptr, po3.ptr, no3.ptr, mo3.ptr, None))
        _ok(lbs3, po3, no3, mo3, q, ref, sw, jm, finite=(lbs3, po3, no3, mo3))
        assert torch.equal(lbs3.t, lbs.t) and torch.equal(po3.t, po.t) and torch.equal(no3.t, no.t) and torch.equal(mo3.t, mo.t)


@pytest.mark.parametrize('N', [1, 1023, 1024, 1025, 70001])
def test_scatter_volume_stays_inside(N):
    L, ctx = _lib.lib(), _lib.ctx(0)
    g = torch.Generator().manual_seed(N)
    flag = torch.rand(N, generator=g) < 0.3
    nv = int(flag.sum())
    valid = flag.to(torch.uint8).cuda()
    vals, fill, vol = _framed(torch.rand(nv, generator=g).cuda()), _framed(torch.rand(N - nv, generator=g).cuda()), Guarded(N)
    _lib.check(L.avc_scatter_volume(ctx, valid.data_ptr(), N, vals.ptr, fill.ptr, vol.ptr, None))
    _ok(vol, vals, fill, finite=(vol,))


@pytest.mark.parametrize('res', [(2, 2, 2), (3, 70, 11), (33, 17, 9), (40, 96, 36), (65, 31, 130)])
def test_marching_cubes_and_rasterisers_stay_inside(res):
    L, ctx = _lib.lib(), _lib.ctx(0)
    g = [np.linspace(-0.5, 0.5, r, dtype=np.float32) for r in res]
    x, y, z = np.meshgrid(*g, indexing='ij')
    field = (0.33 - np.sqrt(x * x + y * y + z * z) + 0.05 * np.sin(40 * x) * np.cos(33 * y)).astype(np.float32)
    vol = _framed(torch.from_numpy(field).cuda())
    r3, b6 = (C.c_int32 * 3)(*res), (C.c_float * 6)(-1, -1, -0.3, 1, 0.9, 0.3)
    counts = (C.c_int64 * 2)()
    rc = L.avc_recon_mesh(ctx, vol.ptr, r3, b6, 0.0, None, None, None, 0, 0, counts, None)           # capacity query
    assert rc in (0, _lib.AVC_ERR_CAPACITY), L.avc_last_error()
    V, F = int(counts[0]), int(counts[1])
    verts, nrm, faces = Guarded(3 * V), Guarded(3 * V), Guarded(3 * F, torch.int32)
    _lib.check(L.avc_recon_mesh(ctx, vol.ptr, r3, b6, 0.0, verts.ptr, nrm.ptr, faces.ptr, V, F, counts, None))   # EXACT capacities
    _ok(verts, nrm, faces, vol, finite=(verts,))
    if F:
        assert int(faces.t.max()) < V and int(faces.t.min()) >= 0
        c3 = (C.c_float * 3)(0, -0.05, 0)
        for size in (33, 96):
            front, back = Guarded(size * size * 3), Guarded(size * size * 3)
            _lib.check(L.avc_render_cano_maps(ctx, verts.ptr, nrm.ptr, V, faces.ptr, F, c3, size, front.ptr, back.ptr, None))
            _ok(front, back, verts, nrm, faces)
        mvp = (C.c_float * 16)(1.2, 0, 0, 0, 0, 1.2, 0, 0, 0, 0, -1, -0.2, 0, 0, -1, 2.5)
        img = Guarded(47 * 29 * 4)
        _lib.check(L.avc_render_mesh(ctx, verts.ptr, verts.ptr, V, faces.ptr, F, mvp, 47, 29, img.ptr, None))
        _ok(img, verts, faces, finite=(img,))


@pytest.mark.parametrize('shape,G', [((1, 64, 128 * 128), 32), ((2, 96, 17 * 23), 32), ((3, 8, 35), 4), ((1, 256, 1), 32)])
def test_group_norm_stays_inside(shape, G):
    L, ctx = _lib.lib(), _lib.ctx(0)
    g = torch.Generator().manual_seed(sum(shape))
    x = _framed(torch.randn(shape, generator=g).cuda())
    ga, be = _framed(torch.randn(shape[1], generator=g).cuda()), _framed(torch.randn(shape[1], generator=g).cuda())
    y = Guarded(x.n)
    _lib.check(L.avc_group_norm(ctx, x.ptr, shape[0], shape[1], shape[2], G, ga.ptr, be.ptr, 1e-5, 1, y.ptr, None))
    _ok(y, x, ga, be, finite=(y,))


@pytest.mark.parametrize('hw', [(64, 64), (97, 61)])
def test_normal_fusion_stays_inside(hw):
    L, ctx = _lib.lib(), _lib.ctx(0)
    g = torch.Generator().manual_seed(hw[0])
    n = hw[0] * hw[1] * 3
    src, tar, out = _framed(torch.randn(n, generator=g).cuda()), _framed(torch.randn(n, generator=g).cuda()), Guarded(n)
    _lib.check(L.avc_merge_normal_images(ctx, src.ptr, tar.ptr, hw[0], hw[1], 6, -5, 20, out.ptr, None))
    _ok(out, src, tar, finite=(out,))
    _lib.check(L.avc_merge_normal_images_cover(ctx, src.ptr, tar.ptr, hw[0] * hw[1], out.ptr, None))
    _ok(out, src, tar, finite=(out,))


@pytest.mark.parametrize('n', [1, 31, 127, 129, 70001])
def test_fused_queries_stay_inside(n):
    """The avatar and recon queries on point lists that do not fill their 128-point tiles: occupancies, offsets and rgba end exactly where they should."""
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
    from avatarcap_amd.network.arch_recon import ReconNetwork
    from common import geotex_sd, recon_sd
    config.cfg = config.default_cfg()
    L, ctx = _lib.lib(), _lib.ctx(0)
    net = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume()).to('cuda').eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in geotex_sd().items()})
    net.warping_field.pose_feat_map = torch.from_numpy(gi.pose_feat_map()[None]).cuda()
    pts = _framed(torch.from_numpy(gi.query_points(5, n)).cuda())
    OccupancyNet(net).query({'cano_pts': pts.t.reshape(1, n, 3), 'cano_smpl_center': torch.from_numpy(gi.center()[None]).cuda()})       # packs + binds
    c3 = (C.c_float * 3)(*gi.center().tolist())
    occ, off, rgba = Guarded(n), Guarded(3 * n), Guarded(4 * n)
    _lib.check(L.avc_avatar_query(ctx, pts.ptr, n, c3, 0, occ.ptr, off.ptr, rgba.ptr, None))
    _ok(occ, off, rgba, pts, finite=(occ, off, rgba))
    rn = ReconNetwork().to('cuda').eval()
    rn.load_state_dict({k: torch.from_numpy(v) for k, v in recon_sd().items()})
    nm = torch.from_numpy(gi.normal_maps(64)[None]).cuda()
    rn.infer({'cano_pts': pts.t.reshape(1, n, 3), 'cano_smpl_center': torch.from_numpy(gi.center()[None]).cuda(), 'front_normal': nm[:, :3], 'back_normal': nm[:, 3:]})
    out = Guarded(n)
    _lib.check(L.avc_recon_query(ctx, pts.ptr, n, c3, out.ptr, None))
    _ok(out, pts, finite=(out,))


@pytest.mark.parametrize('hw', [(64, 64), (128, 64), (63, 64)])
def test_image_encoder_stays_inside(hw):
    """avc_hgfilter_forward on images whose feature maps do not fill the convolution tiles (partial tiles at every level), with and without
    split-K / the second stream / the hipGraph: the NCHW outputs end where they should and are finite."""
    from avatarcap_amd.network.HGFilters import HGFilter
    L = _lib.lib()
    hg = HGFilter(1, 4, 6, 32, 'group', 'no_down', False).to('cuda').eval()
    syn.load_synth(hg, gi.SEED_NET)
    ctx = hg._ctx(torch.device('cuda', 0))
    H1, W1 = (hw[0] - 1) // 2 + 1, (hw[1] - 1) // 2 + 1
    img = _framed(torch.randn(6, hw[0], hw[1], generator=torch.Generator().manual_seed(hw[0])).cuda())
    try:
        for graph, ksplit, fork in ((1, 1, 1), (0, 1, 1), (1, 0, 0)):
            for name, v in (('enc_graph', graph), ('enc_ksplit', ksplit), ('enc_fork', fork)):
                _lib.set_option(name, v)
            feat, normx = Guarded(32 * H1 * W1), Guarded(128 * H1 * W1)
            _lib.check(L.avc_hgfilter_forward(ctx, img.ptr, hw[0], hw[1], feat.ptr, normx.ptr, 0, None))
            _ok(feat, normx, img, finite=(feat, normx))
    finally:
        for name in ('enc_graph', 'enc_ksplit', 'enc_fork'):
            _lib.set_option(name, 1)


@pytest.mark.parametrize('hw', [(128, 128), (128, 256)])
def test_unet_stays_inside(hw):
    """avc_unet_forward on position maps whose deep levels are one or two pixels wide (every convolution tile partial), with and without split-K / the
    hipGraph: the NCHW output ends where it should and is finite; the framed input is not read past its ends."""
    from avatarcap_amd.network.unets import UnetNoCond7DS
    L = _lib.lib()
    un = UnetNoCond7DS(input_nc=6, output_nc=64, nf=32).to('cuda').eval()
    syn.load_synth(un, gi.SEED_NET)
    ctx = un._ctx(torch.device('cuda', 0))
    x = _framed(torch.randn(6, hw[0], hw[1], generator=torch.Generator().manual_seed(hw[1])).cuda())
    try:
        for graph, ksplit in ((1, 1), (0, 1), (1, 0)):
            _lib.set_option('enc_graph', graph)
            _lib.set_option('enc_ksplit', ksplit)
            out = Guarded(64 * hw[0] * hw[1])
            _lib.check(L.avc_unet_forward(ctx, x.ptr, hw[0], hw[1], out.ptr, 0, None))
            _ok(out, x, finite=(out,))
    finally:
        _lib.set_option('enc_graph', 1)
        _lib.set_option('enc_ksplit', 1)


@pytest.mark.parametrize('n_rays,n_samples', [(0, 64), (1, 64), (3, 2), (65, 64), (130, 33), (7, 200)])
def test_render_rays_and_blend_weights_stay_inside(n_rays, n_samples):
    """avc_render_rays_cano with every optional output requested, at ray / sample counts around the wavefront size (its composite kernel takes 64 samples
    per pass, 4 rays per workgroup), and avc_blend_weight_sample on a small volume with points at and beyond the borders."""
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
    from common import geotex_sd
    config.cfg = config.default_cfg()
    L, ctx = _lib.lib(), _lib.ctx(0)
    net = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume()).to('cuda').eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in geotex_sd().items()})
    net.warping_field.pose_feat_map = torch.from_numpy(gi.pose_feat_map()[None]).cuda()
    center = torch.from_numpy(gi.center()[None]).cuda()
    OccupancyNet(net).query({'cano_pts': torch.from_numpy(gi.query_points(5, 8)).cuda()[None], 'cano_smpl_center': center})     # packs + binds
    g = torch.Generator().manual_seed(n_rays * 131 + n_samples)
    P, S = n_rays, n_samples
    d = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=1)
    o = _framed((torch.rand(P, 3, generator=g) * 0.6 - 0.3 + d).cuda())
    dd = _framed((-d).cuda())
    near, far = _framed(torch.full((P,), 0.95).cuda()), _framed(torch.full((P,), 1.05).cuda())
    depth = _framed((torch.rand(P, generator=g) > 0.3).float().cuda())
    t = _framed(torch.linspace(0., 1., steps=S).cuda())
    smpl = _framed(torch.from_numpy(syn.synthetic_body()['cano_smpl_v']).cuda())
    rgb, acc, dep, disp, wts, raw = Guarded(3 * P), Guarded(P), Guarded(P), Guarded(P), Guarded(P * S), Guarded(4 * P * S)
    c3, b6 = (C.c_float * 3)(*gi.center().tolist()), (C.c_float * 6)(*syn.CANO_BOUNDS.reshape(-1).tolist())
    _lib.check(L.avc_render_rays_cano(ctx, o.ptr, dd.ptr, near.ptr, far.ptr, depth.ptr, 0.02, 0.05, t.ptr, P, S, c3, b6, smpl.ptr, smpl.t.numel() // 3, 0,
                                      rgb.ptr, acc.ptr, dep.ptr, disp.ptr, wts.ptr, raw.ptr, None))
    _ok(rgb, acc, dep, disp, wts, raw, o, dd, near, far, depth, t, smpl, finite=(rgb, acc, dep, wts, raw))
    vol = _framed(torch.from_numpy(gi.blend_weight_volume()).cuda())
    X, Y, Z, Cn = gi.blend_weight_volume().shape
    n = P * 3 + 1
    pts = _framed((torch.rand(n, 3, generator=g) * 1.4 - 0.2).cuda())              # inside, on and beyond the unit cube
    out = Guarded(n * Cn)
    _lib.check(L.avc_blend_weight_sample(ctx, vol.ptr, (C.c_int32 * 3)(X, Y, Z), Cn, pts.ptr, n, out.ptr, None))
    _ok(out, vol, pts, finite=(out,))
