"""The two per-frame convolutional producers ON THE GPU (PyTorch-ROCm / MIOpen + the fused GroupNorm op) against goldens produced
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


def test_unet7ds_on_miopen_matches_reference(golden):
    from avatarcap_amd.network.unets import UnetNoCond7DS
    un = UnetNoCond7DS(input_nc=6, output_nc=64, nf=32).to('cuda').eval()
    syn.load_synth(un, gi.SEED_NET)
    with torch.no_grad():
        for res, gold in ((128, golden['G7_unet_samples']), (256, PG['G7_unet256_samples'])):
            y = un(_t(gi.pos_map(res)[None]))[0]
            assert y.shape == (64, res, res)
            got = y[:, torch.from_numpy(gi.PIX[:, 0] % res).cuda(), torch.from_numpy(gi.PIX[:, 1] % res).cuda()].cpu().numpy()
            e = _rel(got, gold)
            print(f'UNet7DS {res}^2 on MIOpen vs reference (CPU): {e:.3e} relative to max(1, |g|max = {np.abs(gold).max():.2f})')
            assert e < 1e-4
        assert abs(float(y.abs().mean()) / float(PG['G7_unet256_absmean']) - 1) < 1e-5
        first = un(_t(gi.pos_map(256, seed=77)[None])).clone()               # a shape / input MIOpen has not seen in this process
        again = un(_t(gi.pos_map(256, seed=77)[None]))
        assert torch.equal(first, again), float((first - again).abs().max())  # first call == later calls, bit for bit


def test_hgfilter_on_miopen_matches_reference(golden):
    from avatarcap_amd.network.HGFilters import HGFilter
    hg = HGFilter(1, 4, 6, 32, 'group', 'no_down', False).to('cuda').eval()
    syn.load_synth(hg, gi.SEED_NET)
    with torch.no_grad():
        for res, gold in ((64, golden['G7_hg_samples']), (512, PG['G7_hg512_samples'])):
            y = hg(_t(gi.normal_maps(res)[None]))[0][-1][0]
            assert y.shape == (32, res // 2, res // 2)
            q = res // 2
            got = y[:, torch.from_numpy(gi.PIX[:, 0] % q).cuda(), torch.from_numpy(gi.PIX[:, 1] % q).cuda()].cpu().numpy()
            e = _rel(got, gold)
            print(f'HGFilter {res}^2 on MIOpen + avc_group_norm vs reference (CPU): {e:.3e} relative to max(1, |g|max = {np.abs(gold).max():.2f})')
            assert e < 1e-4
        assert abs(float(y.abs().mean()) / float(PG['G7_hg512_absmean']) - 1) < 1e-5
        nm = _t(gi.normal_maps(512, seed=78)[None])
        first = hg(nm)[0][-1].clone()
        again = hg(nm)[0][-1]
        assert torch.equal(first, again), float((first - again).abs().max())


def test_recon_infer_end_to_end_at_512(golden):
    """HGFilter on MIOpen + fused decoder vs the reference's infer() on the CPU, at the real 512^2 map size and at 64^2."""
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


def test_hgfilter_graph_replay_equals_eager_launches():
    """ReconNetwork.get_feat_maps replays the encoder as a hipGraph (config.hg_graph): the same kernels on the same arguments -- the feature map is bit
    for bit the eager call's, for a second input through the same graph, after new weights (re-capture), and the replay really is taken."""
    from avatarcap_amd import config
    from avatarcap_amd.network.arch_recon import ReconNetwork
    rn = ReconNetwork().to('cuda').eval()
    syn.load_synth(rn, gi.SEED_NET)
    a, b = _t(gi.normal_maps(512, seed=78)[None]), _t(gi.normal_maps(512, seed=79)[None])
    with torch.no_grad():
        config.hg_graph = False
        try:
            ea, eb = rn.get_feat_maps(a)[-1].clone(), rn.get_feat_maps(b)[-1].clone()
        finally:
            config.hg_graph = True
        ga = rn.get_feat_maps(a)[-1]
        assert rn._hg_graph is not None and getattr(rn, '_hg_graph_failed', None) is None          # captured, not fallen back
        gb = rn.get_feat_maps(b)[-1]
        assert torch.equal(ga, ea) and torch.equal(gb, eb)
        assert torch.equal(rn.get_feat_maps(a)[-1], ea) and ga.data_ptr() != rn.get_feat_maps(a)[-1].data_ptr()   # fresh tensors, not the graph's buffers
        key = rn._hg_graph['key']
        syn.load_synth(rn, gi.SEED_NET + 3)                                                          # new weights: the graph is recorded again
        gc = rn.get_feat_maps(a)[-1]
        assert rn._hg_graph['key'] != key and not torch.equal(gc, ea)
        config.hg_graph = False
        try:
            assert torch.equal(rn.get_feat_maps(a)[-1], gc)
        finally:
            config.hg_graph = True
