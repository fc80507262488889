"""GPU parity of marching cubes + normals (csrc/mesh.hip) and KNN / LBS / skinning (csrc/knn_lbs.hip)
through the C-ABI against the CPU oracle and the reference goldens.
Marching cubes: vertices, faces and their numbering are compared bit for bit with oracle/mc_oracle.c AND directly with
outputs of the real scikit-image call (tests/golden/mc_golden.npz)."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi
from avatarcap_amd import _lib, config, synthetic as syn
from common import maxabs

pytestmark = pytest.mark.gpu


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to('cuda')


def _fields(res, kind, seed=0):
    g = [np.linspace(-1, 1, r, dtype=np.float32) for r in res]
    x, y, z = np.meshgrid(*g, indexing='ij')
    if kind == 'sphere':
        return (0.6 - np.sqrt(x * x + y * y + z * z)).astype(np.float32)
    if kind == 'torus':
        return (0.25 - np.sqrt((np.sqrt(x * x + y * y) - 0.55) ** 2 + z * z)).astype(np.float32)
    if kind == 'noise':     # exercises every ambiguous configuration
        return np.random.RandomState(seed).randn(*res).astype(np.float32)
    if kind == 'blobs':
        rs = np.random.RandomState(seed)
        v = np.zeros(res, np.float32)
        for _ in range(6):
            c = rs.uniform(-0.6, 0.6, 3); r = rs.uniform(0.15, 0.4)
            v = np.maximum(v, (r - np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2)).astype(np.float32))
        return v - 0.05
    raise ValueError(kind)


@pytest.mark.parametrize('res,kind,iso', [((24, 24, 24), 'sphere', 0.0), ((40, 33, 17), 'torus', 0.0), ((19, 23, 31), 'noise', 0.1),
                                          ((64, 64, 64), 'blobs', 0.0), ((33, 9, 130), 'noise', 0.5), ((2, 2, 2), 'noise', 0.0),
                                          ((48, 40, 56), 'noise', 0.0), ((21, 20, 19), 'ints', 0.0), ((3, 70, 1100), 'noise', 0.0)])
def test_recon_mesh_matches_oracle(res, kind, iso):
    """bit-exact: vertex positions (float32), faces, vertex numbering, face order"""
    from avatarcap_amd.utils import recon_util
    from oracle import avatarcap_oracle as orc
    if kind == 'ints':      # ties of the face / interior tests, values equal to iso
        vol = np.random.RandomState(sum(res)).randint(-2, 3, res).astype(np.float32)
    else:
        vol = _fields(res, kind, seed=sum(res))
    v, f, n = recon_util.recon_mesh(_t(vol), list(res), syn.CANO_BOUNDS, iso_value=iso)
    ov, of, on = orc.recon_mesh(vol, list(res), syn.CANO_BOUNDS, iso)
    assert v.dtype == np.float32 and f.dtype == np.int32 and n.dtype == np.float32
    assert f.shape == of.shape and np.array_equal(f, of), 'triangle connectivity differs from the oracle'
    assert v.shape == ov.shape and np.array_equal(v, ov)
    if kind not in ('noise', 'ints'):          # gradients of white noise are ill-conditioned under normalisation
        assert maxabs(n, on) < 1e-4


def test_recon_mesh_fuzz_matches_oracle(monkeypatch):
    """tests/tools/mc_fuzz_gpu.py, 250 cases: random thin / long / odd grid shapes around every tile edge, random iso values, noise, smooth fields, sparse volumes and
    volumes of small integers / half-integers (every test of the case analysis ties, samples equal the iso value); vertices bit for bit, faces, numbering, and the same
    exception type where the library raises.  (The round's campaign: 6,400 cases, 77 M vertices, no mismatch.)"""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location('mc_fuzz_gpu', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'mc_fuzz_gpu.py'))
    mod = importlib.util.module_from_spec(spec)
    dev, cfg = config.device, config.cfg
    try:
        spec.loader.exec_module(mod)
        monkeypatch.setattr(sys, 'argv', ['mc_fuzz_gpu.py', '250', '77'])
        assert mod.main() == 0
    finally:
        config.device, config.cfg = dev, cfg


_MC_GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mc_golden.npz'))


@pytest.mark.parametrize('k', range(len(_MC_GOLD['names'])), ids=lambda k: f"{k}-{_MC_GOLD['names'][k]}")
def test_recon_mesh_matches_scikit_image(k):
    """The device marching cubes against the REAL library (goldens of tests/golden/make_golden_mc.py): identical faces and
    vertex numbering; positions = library vertices + b0 + voxel/2 in float32 (recon_util.py:64-65) to the last bit."""
    from avatarcap_amd.utils import recon_util
    vol, iso, sp = _MC_GOLD[f'vol_{k}'], float(_MC_GOLD[f'iso_{k}']), _MC_GOLD[f'spacing_{k}']
    res = list(vol.shape)
    unit = bool(np.all(sp == 1))
    b0 = np.zeros(3, np.float32) if unit else np.float32([-0.3, 0.1, 0.7])      # unit spacing: (b1 - b0) / res == spacing exactly
    bounds = np.stack([b0, b0 + sp * np.float32(res)]).astype(np.float32)
    voxel = ((bounds[1] - bounds[0]) / np.array(res, np.float32)).astype(np.float32)    # recon_util.py:61-62
    v, f, n = recon_util.recon_mesh(_t(vol), res, bounds, iso_value=iso)
    assert np.array_equal(f, _MC_GOLD[f'faces_{k}'][:, [2, 1, 0]])                       # :69
    if unit:
        assert np.array_equal(voxel, sp)
        assert np.array_equal(v, _MC_GOLD[f'verts_{k}'] + bounds[0] + np.float32(0.5) * voxel)   # :65, float32 like the reference
    else:       # the golden's spacing and (b1 - b0) / res differ in the last bit: compare in index units
        gi_ = _MC_GOLD[f'verts_{k}'].astype(np.float64) / sp
        vi = (v.astype(np.float64) - bounds[0] - 0.5 * voxel.astype(np.float64)) / voxel
        assert v.shape == gi_.shape and maxabs(vi, gi_) < 2e-5 * max(res)


def test_recon_mesh_empty_and_capacity():
    from avatarcap_amd.utils import recon_util
    vol = np.full((8, 8, 8), -1.0, np.float32)
    v, f, n = recon_util.recon_mesh_device(_t(vol), [8, 8, 8], syn.CANO_BOUNDS, 0.0)
    assert v.shape == (0, 3) and f.shape == (0, 3) and n.shape == (0, 3)
    with pytest.raises(ValueError, match='Surface level must be within volume data range'):      # the library's errors (recon_util.py:64)
        recon_util.recon_mesh(_t(vol), [8, 8, 8], syn.CANO_BOUNDS, 0.0)
    with pytest.raises(RuntimeError, match='No surface found'):
        recon_util.recon_mesh(_t(vol), [8, 8, 8], syn.CANO_BOUNDS, -1.0)
    recon_util._cap.clear(); recon_util._cap[torch.cuda.current_device()] = (16, 16)   # force the grow-and-retry path
    big = _fields((48, 48, 48), 'sphere')
    v, f, n = recon_util.recon_mesh(_t(big), [48, 48, 48], syn.CANO_BOUNDS, 0.0)
    assert v.shape[0] > 16 and f.shape[0] > 16


def test_normals_match_reference_golden(golden):
    """G9 pins the Sobel + trilinear arithmetic to the reference; here the HIP kernel evaluates it at
    marching-cubes vertices, so compare through the oracle on the same volume."""
    from avatarcap_amd.utils import recon_util
    from oracle import avatarcap_oracle as orc
    vol, voxel = gi.sdf_volume(32)
    bounds = np.stack([np.zeros(3, np.float32), voxel * 32]).astype(np.float32)
    v, f, n = recon_util.recon_mesh(_t(vol), [32, 32, 32], bounds, 0.0)
    ov, of, on = orc.recon_mesh(vol, [32, 32, 32], bounds, 0.0)
    assert maxabs(n, on) < 1e-4
    assert np.abs(np.linalg.norm(n, axis=1) - 1).max() < 1e-5


@pytest.mark.parametrize('K', [1, 4, 8])
def test_knn_vs_oracle(body, K):
    from avatarcap_amd.utils.smpl_util import SmplUtil
    from oracle import avatarcap_oracle as orc
    su = SmplUtil()
    q = gi.surface_points(900 + K, 3000, body)
    d2, idx = su.knn_points(_t(q[None]), _t(body['cano_smpl_v'][None]), K=K)
    od2, oidx = orc.knn(q, body['cano_smpl_v'], K)
    assert idx.dtype == torch.int64 and np.array_equal(idx[0].cpu().numpy(), oidx)
    assert np.array_equal(d2[0].cpu().numpy(), od2)


@pytest.mark.parametrize('K', [1, 4, 8])
def test_knn_grid_equals_brute_force(body, K, monkeypatch):
    """The wave-cooperative grid search must return exactly what the exhaustive scan returns: scattered
    queries (large wave bounding boxes), grid lines far from the body, queries outside the reference box,
    duplicated reference points (ties -> lower index), and a reference set too small for the grid."""
    from avatarcap_amd.utils.smpl_util import SmplUtil
    from avatarcap_amd.grid import generate_volume_points_np
    su = SmplUtil()
    rs = np.random.RandomState(7 + K)
    ref = body['cano_smpl_v'].copy()
    ref[1000:1100] = ref[2000:2100]                                           # exact duplicates
    ref[3000:3010] = ref[3000]
    lines = generate_volume_points_np(syn.CANO_BOUNDS, [24, 24, 96])            # z-fastest runs, like the frame's grid
    q = np.concatenate([rs.uniform(-1.5, 1.5, (5000, 3)), lines, gi.surface_points(31, 4000, body), ref[990:1110], ref[2995:3015],
                        np.float32([[50, 0, 0], [-3, -3, -3], [0, 0, 0]])]).astype(np.float32)
    for refs in (ref, ref[:700], ref[:40]):
        qt, rt = _t(q[None]), _t(refs[None])
        d_g, i_g = su.knn_points(qt, rt, K=K)
        _lib.set_option('knn_search', 3)
        d_b, i_b = su.knn_points(qt, rt, K=K)
        _lib.set_option('knn_search', 0)
        assert torch.equal(i_g, i_b) and torch.equal(d_g, d_b)
        for path in ('lane', 'wave'):                                          # each of the two grid searches on its own, every wave
            _lib.set_option('knn_search', {'lane': 1, 'wave': 2}[path])
            d_p, i_p = su.knn_points(qt, rt, K=K)
            _lib.set_option('knn_search', 0)
            assert torch.equal(i_p, i_b) and torch.equal(d_p, d_b), path
        assert bool((d_g[0, :, 1:] >= d_g[0, :, :-1]).all())
        if K > 1:                                                              # ties resolved towards the lower index
            tie = d_g[0, :, 1:] == d_g[0, :, :-1]
            assert bool((i_g[0, :, 1:][tie] > i_g[0, :, :-1][tie]).all())


@pytest.mark.parametrize('K', [1, 3])
def test_knn_large_reference_set(body, K, monkeypatch):
    """main.py:478-481 looks every reconstructed vertex up among ~1e6 avatar vertices: the grid is built by
    several workgroups and is finer (up to 128 cells per axis); far, near and scattered queries stay exact."""
    from avatarcap_amd.utils.smpl_util import SmplUtil
    su = SmplUtil()
    rs = np.random.RandomState(5)
    bv = body['cano_smpl_v']
    ref = (bv[rs.randint(0, bv.shape[0], 400_000)] + rs.normal(0, 0.01, (400_000, 3))).astype(np.float32)
    ref[5000:5050] = ref[100:150]                                             # duplicates
    q = np.concatenate([bv[rs.randint(0, bv.shape[0], 30_000)] + rs.normal(0, 0.02, (30_000, 3)),
                        rs.uniform(-1.2, 1.2, (3000, 3)), ref[90:160], np.float32([[9, 9, 9]])]).astype(np.float32)
    q[:30_000] = q[:30_000][np.lexsort(np.floor(q[:30_000] * 20).T)]          # spatially coherent, like mesh vertices
    qt, rt = _t(q[None]), _t(ref[None])
    d_g, i_g = su.knn_points(qt, rt, K=K)
    _lib.set_option('knn_search', 3)
    d_b, i_b = su.knn_points(qt, rt, K=K)
    _lib.set_option('knn_search', 0)
    assert torch.equal(i_g, i_b) and torch.equal(d_g, d_b)
    for path in ('lane', 'wave'):
        _lib.set_option('knn_search', {'lane': 1, 'wave': 2}[path])
        d_p, i_p = su.knn_points(qt, rt, K=K)
        _lib.set_option('knn_search', 0)
        assert torch.equal(i_p, i_b) and torch.equal(d_p, d_b), path
    dd = torch.cdist(qt[0, :2000].double(), rt[0].double()) ** 2                # an independent check of a sample
    assert float((dd.min(1).values - d_g[0, :2000, 0].double()).abs().max()) < 1e-6


def test_lbs_skinning_matches_reference_golden(golden, body):
    from avatarcap_amd.utils.smpl_util import SmplUtil
    su = SmplUtil(body['skin_weights'])
    with pytest.raises(ValueError):
        su.calculate_lbs(_t(np.zeros((1, 4, 3), np.float32)))            # smpl_util.py:30-31
    su.set_cano_smpl_vertices(_t(body['cano_smpl_v']))
    vp = gi.surface_points(105, 700, body)
    lbs = su.calculate_lbs(_t(vp[None]))
    assert lbs.shape == (1, 700, 24)
    assert maxabs(lbs[0].cpu().numpy(), golden['G8_lbs']) < 2e-6
    jm = syn.random_pose_jnt_mats(gi.SEED_POSE)
    live, mats = su.skinning(_t(vp[None]), lbs, _t(jm[None]), True)
    assert maxabs(live[0].cpu().numpy(), golden['G8_live']) < 5e-6
    assert maxabs(mats[0].cpu().numpy(), golden['G8_mats']) < 5e-6
    nrm = gi.unit_vectors(106, 700)
    ln = su.skinning_normal(_t(nrm[None]), lbs, _t(jm[None]))
    assert maxabs(ln[0].cpu().numpy(), golden['G8_live_normals']) < 5e-6
    # far points underflow to all-zero rows by design
    far = su.calculate_lbs(_t(np.full((1, 3, 3), 5.0, np.float32)))
    assert float(far.abs().max()) == 0.0


def test_bound_lbs_equals_the_exhaustive_scan(body):
    """avc_lbs_prepare / avc_calculate_lbs_bound (round 5): one candidate list per 2 cm cell around the bound vertices instead of a search.  Held bit for
    bit to the exhaustive scan of avc_calculate_lbs on: points on and around the body (the lists), the vertices themselves and duplicated vertices (zero
    distances, ties -> lower index), points exactly on cell faces, points far from the body and outside the cells' box (no list: the grid search), and the
    points of a whole coarse grid over the canonical bounds."""
    import ctypes as C
    from avatarcap_amd.utils.smpl_util import SmplUtil
    from avatarcap_amd.grid import generate_volume_points_np
    rs = np.random.RandomState(11)
    ref = body['cano_smpl_v'].copy()
    ref[100:110] = ref[90:100]                                            # duplicated reference points
    su = SmplUtil(body['skin_weights'])
    su.set_cano_smpl_vertices(_t(ref))
    st = (C.c_int64 * 4)()
    _lib.check(_lib.lib().avc_lbs_bound_stats(_lib.ctx(0), st))
    assert st[0] == ref.shape[0] and st[3] == 1 and st[1] > 10000 and 20 * st[1] > st[2] > st[1] // 10, list(st)
    print(f'bound LBS: {st[1]} cells, {st[2]} list entries ({st[2] / st[1]:.1f} per cell over all cells)')
    surf = gi.surface_points(1234, 5000, body)
    lo = ref.min(0) - 0.16
    faces = lo + 0.02 * rs.randint(0, 40, (3000, 3)).astype(np.float32)   # on the faces / edges / corners of the 2 cm cells
    q = np.concatenate([surf, surf + 0.05 * rs.randn(*surf.shape).astype(np.float32), ref[:2000], ref[90:110], faces,
                        rs.uniform(-3, 3, (3000, 3)).astype(np.float32), np.full((3, 3), 50.0, np.float32),
                        generate_volume_points_np(syn.CANO_BOUNDS, (24, 24, 12))]).astype(np.float32)
    got = su.calculate_lbs(_t(q[None]))
    _lib.set_option('knn_search', 3)
    try:
        want = su._lbs(_t(q[None]), su.cano_smpl_vertices)                # avc_calculate_lbs, exhaustive scan
        want_b = su.calculate_lbs(_t(q[None]))                            # the bound entry with the same switch: the exhaustive scan of its own copy
    finally:
        _lib.set_option('knn_search', 0)
    assert torch.equal(got, want) and torch.equal(want_b, want)
    assert float(got[0, :5000].sum(1).min()) > 0.99                      # (points on the body: real weights, not the underflowed rows of far points)
    # another SmplUtil binding its own vertices takes the context's slot; the first one notices and rebinds
    su2 = SmplUtil(body['skin_weights'])
    su2.set_cano_smpl_vertices(_t(ref[::2].copy()))
    assert torch.equal(su.calculate_lbs(_t(q[None])), want)
    # in-place edits of the bound tensor are seen too (tensor version)
    su.cano_smpl_vertices += 0.01
    moved = su.calculate_lbs(_t(q[None]))
    assert not torch.equal(moved, want) and torch.equal(moved, su._lbs(_t(q[None]), su.cano_smpl_vertices))


def test_fused_lbs_skinning_equals_the_three_calls(body):
    """avc_lbs_skin_bound (round 6): calculate_lbs + skinning + skinning_normal of main.py:385-389 in one launch.  The same operations in the same order, so every
    output must equal the three calls' bit for bit: on and around the body (candidate lists), far and outside points (grid search), the exhaustive scan and the
    forced search paths, a ragged count, without normals / matrices, with the weights handed out; and whatever the reach of the lists (a dense dataset binds with
    1000 mm, FramePipeline.__init__)."""
    from avatarcap_amd.utils.smpl_util import SmplUtil
    from avatarcap_amd.grid import generate_volume_points_np
    rs = np.random.RandomState(12)
    ref = body['cano_smpl_v'].copy()
    surf = gi.surface_points(4321, 4000, body)
    q = np.concatenate([surf, surf + 0.05 * rs.randn(*surf.shape).astype(np.float32), ref[:1000], rs.uniform(-3, 3, (2000, 3)).astype(np.float32),
                        np.full((3, 3), 50.0, np.float32), generate_volume_points_np(syn.CANO_BOUNDS, (20, 20, 10))]).astype(np.float32)[:12345]
    nrm = gi.unit_vectors(107, q.shape[0])
    jm = _t(syn.random_pose_jnt_mats(gi.SEED_POSE)[None])
    tq, tn = _t(q[None]), _t(nrm[None])
    su = SmplUtil(body['skin_weights'])
    try:
        for reach in (140, 0, 1000):
            _lib.set_option('lbs_reach_mm', reach)
            su.set_cano_smpl_vertices(_t(ref))
            for search in (0, 1, 2, 3):
                _lib.set_option('knn_search', search)
                lbs = su.calculate_lbs(tq)
                live, mats = su.skinning(tq, lbs, jm, True)
                ln = su.skinning_normal(tn, lbs, jm)
                po, no, mo, lo = su.lbs_skinning(tq, tn, jm, return_pt_mats=True, return_lbs=True)
                assert torch.equal(po, live) and torch.equal(no, ln) and torch.equal(mo, mats) and torch.equal(lo, lbs), (reach, search)
            _lib.set_option('knn_search', 0)
            po2, no2, mo2, lo2 = su.lbs_skinning(tq, None, jm)                # points only, nothing else written
            assert no2 is None and mo2 is None and lo2 is None and torch.equal(po2, live)
            if reach == 140:
                want_lbs = lbs
            else:
                assert torch.equal(lbs, want_lbs), reach                      # same bits whatever the reach
        assert float(lo[0, :4000].sum(1).min()) > 0.99
        # an empty mesh
        e = torch.zeros((1, 0, 3), device='cuda')
        po, no, mo, _ = su.lbs_skinning(e, e, jm, return_pt_mats=True)
        assert po.shape == (1, 0, 3) and no.shape == (1, 0, 3) and mo.shape == (1, 0, 4, 4)
    finally:
        _lib.set_option('knn_search', 0)
        _lib.set_option('lbs_reach_mm', 140)


def test_knn_fuzz_every_path_equals_the_exhaustive_scan(monkeypatch):
    """tests/tools/knn_fuzz_gpu.py, 150 cases: reference sets of 4 .. 20,000 points -- clustered, collinear, coplanar, on a lattice, duplicated -- and queries inside,
    far outside and exactly on them; the default search, each grid search forced, and the bound LBS at a random list reach against the exhaustive scan bit for bit,
    small cases also against the fp32 oracle.  (The round's campaign: 4,500 cases / 4.7 M queries without a mismatch; it found avc_lbs_prepare asking for more
    memory than the device has when thousands of coincident vertices put every vertex on every cell's list: beyond 2^28 entries no lists are built.)"""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location('knn_fuzz_gpu', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'knn_fuzz_gpu.py'))
    mod = importlib.util.module_from_spec(spec)
    dev, cfg = config.device, config.cfg
    try:
        spec.loader.exec_module(mod)
        monkeypatch.setattr(sys, 'argv', ['knn_fuzz_gpu.py', '150', '9'])
        assert mod.main() == 0
    finally:
        config.device, config.cfg = dev, cfg


def test_scatter_volume():
    from avatarcap_amd import _lib
    N = 100003
    rs = np.random.RandomState(5)
    valid = rs.rand(N) < 0.3
    values = rs.randn(int(valid.sum())).astype(np.float32)
    fill = rs.randn(int((~valid).sum())).astype(np.float32)
    vol = torch.empty(N, dtype=torch.float32, device='cuda')
    tv, tf, tval = _t(valid.astype(np.uint8)), _t(fill), _t(values)
    _lib.check(_lib.lib().avc_scatter_volume(_lib.ctx(), tv.data_ptr(), N, tval.data_ptr(), tf.data_ptr(), vol.data_ptr(), _lib.stream_ptr()))
    ref = np.empty(N, np.float32); ref[valid] = values; ref[~valid] = fill          # main.py:362-363
    assert np.array_equal(vol.cpu().numpy(), ref)


@pytest.mark.parametrize('res,kind', [((64, 64, 64), 'blobs'), ((19, 32, 128), 'noise'), ((9, 8, 256), 'noise'), ((5, 6, 512), 'noise'), ((4, 3, 1024), 'noise'),
                                      ((33, 64, 16), 'noise'), ((40, 512, 4), 'ints'), ((2, 16, 64), 'noise'), ((17, 48, 128), 'torus')])
def test_recon_mesh_walking_classify_equals_the_general_one(res, kind):
    """Volumes whose 1024-point tiles are whole x rows of one z plane take the classify pass that walks z inside a workgroup (avc_set_option "mc_walk"):
    the mesh must be the oracle's bit for bit, and the general pass's, on shapes that put 1, 2, 4, 8, 16, 64 and 256 rows into a tile, a last axis shorter
    than a wavefront's reach, a first axis that is not a multiple of the z segment and the two-plane minimum."""
    from avatarcap_amd.utils import recon_util
    from oracle import avatarcap_oracle as orc
    assert 1024 % res[2] == 0 and res[1] % (1024 // res[2]) == 0
    vol = np.random.RandomState(sum(res)).randint(-2, 3, res).astype(np.float32) if kind == 'ints' else _fields(res, kind, seed=sum(res))
    iso = 0.1 if kind == 'noise' else 0.0
    out = {}
    for walk in (1, 0):
        _lib.set_option('mc_walk', walk)
        try:
            out[walk] = recon_util.recon_mesh(_t(vol), list(res), syn.CANO_BOUNDS, iso_value=iso)
        finally:
            _lib.set_option('mc_walk', 1)
    ov, of, _ = orc.recon_mesh(vol, list(res), syn.CANO_BOUNDS, iso)
    for walk in (1, 0):
        v, f, n = out[walk]
        assert f.shape == of.shape and np.array_equal(f, of) and v.shape == ov.shape and np.array_equal(v, ov), f'mc_walk {walk}'
    assert np.array_equal(out[0][2], out[1][2])
