"""The two per-frame convolutional producers ON THE GPU (the U-Net and HGFilter, both on the hand-written HIP encoder) against goldens produced
by the reference's own modules on the CPU: UnetNoCond7DS (network/unets.py:169-229) at 128^2 and at its real 256^2 input,
HGFilter (network/HGFilters.py:124-219) at 64^2 and at its real 512^2 input, and ReconNetwork.infer end to end (arch_recon.py:45-76).
Bar: north_star's 1e-4 on O(1) outputs (relative to max(1, |golden|_max)); the measured errors are printed.  Also: the producers must give
the same bits on the first and on later calls (vertex counts must not depend on call order)."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi
from avatarcap_amd import config, synthetic as syn
from common import maxabs

pytestmark = pytest.mark.gpu
PG = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'producers_golden.npz'))


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to('cuda')


def _rel(got, gold):
    return maxabs(got, gold) / max(1.0, float(np.abs(gold).max()))


def _unet(seed=gi.SEED_NET):
    from avatarcap_amd.network.unets import UnetNoCond7DS
    un = UnetNoCond7DS(input_nc=6, output_nc=64, nf=32).to('cuda').eval()
    syn.load_synth(un, seed)
    return un


def test_unet7ds_on_hip_matches_reference(golden):
    """UnetNoCond7DS.forward on the hand-written convolution kernel against goldens of the reference's own module on the CPU (128^2 and the real
    256^2 position map), and: first call == later calls bit for bit; the module refuses what it does not implement."""
    un = _unet()
    with torch.no_grad():
        for res, gold in ((128, golden['G7_unet_samples']), (256, PG['G7_unet256_samples'])):
            y = un(_t(gi.pos_map(res)[None]))[0]
            assert y.shape == (64, res, res)
            got = y[:, torch.from_numpy(gi.PIX[:, 0] % res).cuda(), torch.from_numpy(gi.PIX[:, 1] % res).cuda()].cpu().numpy()
            e = _rel(got, gold)
            print(f'UNet7DS {res}^2 on the HIP encoder vs reference (CPU): {e:.3e} relative to max(1, |g|max = {np.abs(gold).max():.2f})')
            assert e < 1e-4
        assert abs(float(y.abs().mean()) / float(PG['G7_unet256_absmean']) - 1) < 1e-5
        first = un(_t(gi.pos_map(256, seed=77)[None])).clone()
        again = un(_t(gi.pos_map(256, seed=77)[None]))
        assert torch.equal(first, again), float((first - again).abs().max())  # first call == later calls, bit for bit
        two = un(torch.cat([_t(gi.pos_map(256, seed=77)[None]), _t(gi.pos_map(256)[None])]))        # a batch is a loop over items
        assert torch.equal(two[0], first[0]) and torch.equal(two[1], y)
        with pytest.raises(RuntimeError, match='HIP device only'):
            un(torch.from_numpy(gi.pos_map(128)[None]))
        with pytest.raises(Exception, match='multiples of 128'):
            un(_t(gi.pos_map(256)[None, :, :192, :192]))
        un.train()
        with pytest.raises(RuntimeError, match='eval'):
            un(_t(gi.pos_map(128)[None]))


@pytest.mark.parametrize('res', [128, 256])
def test_unet_launch_by_launch(res):
    """Every launch of the U-Net's plan (csrc/conv_enc.hip: 1 space-to-depth, 7 + 4 + 3 convolutions, 3 bilinear upsamples) against the stock-torch
    restatement of the tensor it wrote (tests/torch_unet.py, fp32 on the same device).  The convolutions write channel slices of the concatenated
    decoder tensors: an encoder level is compared on its slice, a decoder level on the whole concatenation."""
    import ctypes as C
    from avatarcap_amd import _lib
    from torch_unet import unet7ds_trace
    un = _unet()
    x = _t(gi.pos_map(res)[None])
    with torch.no_grad():
        y = un(x)
        torch.cuda.synchronize()
        trace = unet7ds_trace(un, x)
    ctx, L = _lib.ctx(x.device), _lib.lib()

    def fetch(launch):
        c, h, w = C.c_int32(), C.c_int32(), C.c_int32()
        rc = L.avc_hgfilter_debug_tensor(ctx, launch, 2, None, C.byref(c), C.byref(h), C.byref(w), None)
        assert rc == 0, (launch, rc, L.avc_last_error())
        got = torch.empty((c.value & 0xffff, h.value, w.value), dtype=torch.float32, device='cuda')
        _lib.check(L.avc_hgfilter_debug_tensor(ctx, launch, 2, got.data_ptr(), C.byref(c), C.byref(h), C.byref(w), _lib.stream_ptr(x.device)))
        torch.cuda.synchronize()
        cfg = c.value >> 16
        return got, (f'CT{cfg & 15} PT{(cfg >> 4) & 15}{" splitK" if cfg >> 14 else ""}' if cfg else '')
    # launch index of every traced tensor: [s2d, conv1..7, upconv1, 2, 3, 3, up2, C5, up2, C6, up2, C7]
    launches = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17]
    worst, bad = 0.0, []
    for (name, ref), launch in zip(trace, launches):
        got, cfg = fetch(launch)
        ref = ref[0]
        if name.startswith('conv') and name != 'conv7':                       # the skip slice of the decoder tensor
            got = got[-ref.shape[0]:]
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        e = float((got - ref).abs().max()) / max(1.0, float(ref.abs().max()))
        print('  launch %3d %-18s %-18s %-14s %.3e' % (launch, name, tuple(ref.shape), cfg, e))
        if not e < 1e-4:
            bad.append((launch, name, e))
        worst = max(worst, e)
    assert not bad, f'first launch off: {bad[0]}'
    assert torch.equal(fetch(17)[0], y[0])
    print(f'UNet7DS {res}^2, HIP plan vs stock torch ops launch by launch: worst {worst:.3e} (relative to max(1, |ref|max))')


def test_unet_graph_replay_equals_plain_launches():
    """avc_unet_forward replays a hipGraph (config.hg_graph): bit for bit the plain launches of the same kernels, with and without split-K the results
    agree to rounding, fresh output tensors, new weights and a second module are picked up; the module deep-copies and pickles."""
    import copy
    import pickle
    from avatarcap_amd import _lib
    un = _unet()
    a, b = _t(gi.pos_map(256, seed=77)[None]), _t(gi.pos_map(256, seed=78)[None])
    with torch.no_grad():
        config.hg_graph = False
        try:
            ea, eb = un(a).clone(), un(b).clone()
        finally:
            config.hg_graph = True
        ga = un(a)
        gb = un(b)
        assert torch.equal(ga, ea) and torch.equal(gb, eb) and ga.data_ptr() != gb.data_ptr()
        _lib.set_option('enc_ksplit', 0, a.device)
        try:
            na = un(a)
        finally:
            _lib.set_option('enc_ksplit', 1, a.device)
        assert float((na - ea).abs().max()) < 1e-5 * max(1.0, float(ea.abs().max()))
        assert torch.equal(un(a), ea)
        un2 = pickle.loads(pickle.dumps(copy.deepcopy(un)))
        assert un2._packed is None and torch.equal(un2(a), ea)
        other = _unet(gi.SEED_NET + 5)
        oa = other(a)
        assert not torch.equal(oa, ea) and torch.equal(un(a), ea)             # two modules share the context: each call runs on its own weights
        syn.load_synth(un, gi.SEED_NET + 5)
        assert torch.equal(un(a), oa)


def _hg():
    from avatarcap_amd.network.HGFilters import HGFilter
    hg = HGFilter(1, 4, 6, 32, 'group', 'no_down', False).to('cuda').eval()
    syn.load_synth(hg, gi.SEED_NET)
    return hg


def _walk_plan(hg, x, tol=1e-4):
    """Every launch of the HIP encoder's plan against the stock-torch restatement of the same tensor (tests/torch_hgfilter.py, fp32 on the GPU):
    names the first launch that disagrees.  Returns the worst relative error and the plan's launch kinds."""
    import ctypes as C
    from avatarcap_amd import _lib
    from torch_hgfilter import hgfilter_trace
    with torch.no_grad():
        hg.encode(x)
        torch.cuda.synchronize()
        trace = hgfilter_trace(hg, x)
    ctx, L = _lib.ctx(x.device), _lib.lib()
    launch, worst, report = 0, 0.0, []
    for name, which, ref in trace:
        while True:                                   # skip launches without a tensor (separate statistics launches)
            c, h, w = C.c_int32(), C.c_int32(), C.c_int32()
            rc = L.avc_hgfilter_debug_tensor(ctx, launch, 1 if which == 'y' else 0, None, C.byref(c), C.byref(h), C.byref(w), None)
            assert rc >= 0, L.avc_last_error()
            if rc == 0:
                break
            launch += 1
        cfg = c.value >> 16
        ch = c.value & 0xffff
        assert (ch, h.value, w.value) == tuple(ref.shape[1:]), (name, launch, (ch, h.value, w.value), tuple(ref.shape))
        got = torch.empty_like(ref[0])
        _lib.check(L.avc_hgfilter_debug_tensor(ctx, launch, 1 if which == 'y' else 0, got.data_ptr(), C.byref(c), C.byref(h), C.byref(w), _lib.stream_ptr(x.device)))
        torch.cuda.synchronize()
        e = float((got - ref[0]).abs().max()) / max(1.0, float(ref.abs().max()))
        report.append((launch, name, tuple(ref.shape[1:]), f'CT{cfg & 15} PT{(cfg >> 4) & 15} taps{(cfg >> 8) & 31}{" splitK" if cfg >> 14 else ""}' if cfg else '', e))
        worst = max(worst, e)
        launch += 1
    for r in report:
        print('  launch %3d %-22s %-16s %-16s %.3e' % r)
    bad = [r for r in report if not r[4] < tol]
    assert not bad, f'first launch off: {bad[0]}'
    return worst


@pytest.mark.parametrize('res', [64, 512])
def test_hgfilter_launch_by_launch(res):
    """The hand-written encoder (csrc/conv_enc.hip), every intermediate tensor against stock torch ops on the same device."""
    hg = _hg()
    x = _t(gi.normal_maps(res)[None])
    worst = _walk_plan(hg, x)
    print(f'HGFilter {res}^2, HIP encoder vs stock torch ops launch by launch: worst {worst:.3e} (relative to max(1, |ref|max))')


@pytest.mark.parametrize('hw', [(96, 160), (32, 224), (160, 96)])
def test_hgfilter_launch_by_launch_on_other_image_shapes(hw):
    """Non-square and narrow images (tools/enc_shapes_probe.py also ran 64 x 128 .. 480 x 512: worst launch 2.8e-6): the plan's tilings, halos and statistics
    tables are sized per launch shape."""
    hg = _hg()
    x = _t(np.random.RandomState(sum(hw)).randn(1, 6, *hw).astype(np.float32))
    worst = _walk_plan(hg, x)
    print(f'HGFilter {hw[0]} x {hw[1]}, HIP encoder vs stock torch ops launch by launch: worst {worst:.3e}')


def test_hgfilter_matches_reference(golden):
    """HGFilter.forward on the HIP encoder against goldens of the reference's own module on the CPU (64^2 and the real 512^2 input)."""
    hg = _hg()
    with torch.no_grad():
        for res, gold in ((64, golden['G7_hg_samples']), (512, PG['G7_hg512_samples'])):
            outs, normx = hg(_t(gi.normal_maps(res)[None]))
            y = outs[-1][0]
            q = res // 2
            assert y.shape == (32, q, q) and normx.shape == (1, 128, q, q)
            got = y[:, torch.from_numpy(gi.PIX[:, 0] % q).cuda(), torch.from_numpy(gi.PIX[:, 1] % q).cuda()].cpu().numpy()
            e = _rel(got, gold)
            print(f'HGFilter {res}^2 on the HIP encoder vs reference (CPU): {e:.3e} relative to max(1, |g|max = {np.abs(gold).max():.2f})')
            assert e < 1e-4
        assert abs(float(y.abs().mean()) / float(PG['G7_hg512_absmean']) - 1) < 1e-5
        nm = _t(gi.normal_maps(512, seed=78)[None])
        first = hg(nm)[0][-1].clone()
        again = hg(nm)[0][-1]
        assert torch.equal(first, again), float((first - again).abs().max())          # deterministic: no floating-point atomics anywhere


def test_hgfilter_switches_change_nothing():
    """hipGraph replay vs plain launches, the hourglass' upper branches on a second stream or not: the same bits; new weights are picked up."""
    from avatarcap_amd import _lib
    hg = _hg()
    a = _t(gi.normal_maps(512, seed=78)[None])
    with torch.no_grad():
        base = hg(a)[0][-1].clone()
        try:
            for g, f in ((0, 1), (1, 0), (0, 0)):      # (split-K changes the rounding: see the next test)
                _lib.set_option('enc_graph', g)
                _lib.set_option('enc_fork', f)
                assert torch.equal(hg(a)[0][-1], base), (g, f)
        finally:
            _lib.set_option('enc_graph', 1)
            _lib.set_option('enc_fork', 1)
        b = _t(gi.normal_maps(512, seed=79)[None])
        assert not torch.equal(hg(b)[0][-1], base) and torch.equal(hg(a)[0][-1], base)     # a second input through the same graph
        syn.load_synth(hg, gi.SEED_NET + 3)
        assert not torch.equal(hg(a)[0][-1], base)                                          # new weights: packed again


def test_hgfilter_without_split_k():
    """avc_set_option "enc_ksplit" 0: every convolution walks its whole K in one workgroup per tile (the small hourglass levels then run on a handful of
    CUs).  Both forms are held to the launch-by-launch reference; they differ from each other by fp32 rounding only."""
    from avatarcap_amd import _lib
    hg = _hg()
    x = _t(gi.normal_maps(512)[None])
    with torch.no_grad():
        a = hg(x)[0][-1].clone()
        _lib.set_option('enc_ksplit', 0)
        try:
            worst = _walk_plan(hg, x)
            b = hg(x)[0][-1].clone()
        finally:
            _lib.set_option('enc_ksplit', 1)
    d = float((a - b).abs().max()) / max(1.0, float(a.abs().max()))
    print(f'without split-K: worst launch {worst:.3e}; feature map with vs without split-K {d:.3e}')
    assert d < 2e-5


def test_recon_infer_end_to_end_at_512(golden):
    """HGFilter on the HIP encoder + fused decoder vs the reference's infer() on the CPU, at the real 512^2 map size and at 64^2."""
    from avatarcap_amd.network.arch_recon import ReconNetwork
    rn = ReconNetwork().to('cuda').eval()
    syn.load_synth(rn, gi.SEED_NET)
    pts = gi.query_points(104, 2048)
    for res, gold in ((512, PG['G6_infer512']), (64, golden['G6_recon'])):
        nm = gi.normal_maps(res)
        items = {'cano_pts': _t(pts[None]), 'cano_smpl_center': _t(gi.center()[None]), 'front_normal': _t(nm[None, :3]), 'back_normal': _t(nm[None, 3:])}
        y = rn.infer(items)
        assert y.shape == gold.shape == (1, 2048)
        e = maxabs(y.cpu().numpy(), gold)
        print(f'ReconNetwork.infer ({res}^2 maps) vs reference (CPU): {e:.3e}')
        assert e < 1e-4




def test_hgfilter_range_check_trips():
    """config.check_range (avc_set_range_check): an encoder whose normalised activations leave the fp16 range of the split (a GroupNorm gain of 1e5: the
    staged value is 16 x the activation) returns AVC_ERR_RANGE instead of a feature map of infs the decoder would turn into plausible occupancies."""
    from avatarcap_amd import _lib
    hg = _hg()
    x = _t(gi.normal_maps(64)[None])
    config.check_range = True
    try:
        with torch.no_grad():
            hg(x)                                                             # in range: no error
            hg.conv3.bn2.weight.mul_(1e5)
            with pytest.raises(_lib.AvcapError) as ei:
                hg(x)
        assert ei.value.status == _lib.AVC_ERR_RANGE
    finally:
        config.check_range = False
        _lib.apply_range_check(_lib.ctx(x.device))


def test_hgfilter_one_or_two_workgroups_per_cu():
    """avc_set_option "enc_occ2": the 3x3 convolutions that have two to four half-height workgroups per CU run two per CU (half the LDS each, one staged
    chunk), the others one per CU.  The two forms cut the image into different tiles, so the fp32 per-tile partial sums of GroupNorm differ by rounding: both
    are held to the launch-by-launch reference, and to each other to a few 1e-6 of the map's scale; each is deterministic."""
    from avatarcap_amd import _lib
    hg = _hg()
    x = _t(gi.normal_maps(512)[None])
    with torch.no_grad():
        a = hg(x)[0][-1].clone()
        assert torch.equal(hg(x)[0][-1], a)
        _lib.set_option('enc_occ2', 0)
        try:
            worst = _walk_plan(hg, x)
            b = hg(x)[0][-1].clone()
            assert torch.equal(hg(x)[0][-1], b)
        finally:
            _lib.set_option('enc_occ2', 1)
        assert torch.equal(hg(x)[0][-1], a)                                       # back: the plan is rebuilt, same bits as before
    d = float((a - b).abs().max()) / max(1.0, float(a.abs().max()))
    print(f'one workgroup per CU: worst launch {worst:.3e}; feature map two-per-CU vs one-per-CU {d:.3e}')
    assert d < 2e-5
