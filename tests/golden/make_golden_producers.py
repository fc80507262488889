#!/usr/bin/env python3
"""Golden vectors of the two per-frame producers AT THEIR REAL INPUT SIZES, by importing the reference (/root/reference):
  G7_unet256_samples   UnetNoCond7DS(6 -> 64, nf 32) on a (6,256,256) pos map   (network/unets.py:169-229, arch_avatar.py:109-111)
  G7_hg512_samples     HGFilter(1,4,6,32,'group','no_down',False)[-1] on a (6,512,512) normal-map pair   (network/HGFilters.py:124-219,
                       arch_recon.py:51-52)
  G6_infer512          ReconNetwork.infer on those maps at 2048 points           (arch_recon.py:45-76)
sampled at the 24 fixed pixels of golden_inputs.PIX (all channels), plus their checksums.  tests/golden/reference_golden.npz keeps the
small-input versions (128^2 / 64^2); the GPU tests hold MIOpen's output to both.  Build container only.

    python tests/golden/make_golden_producers.py        ->  tests/golden/producers_golden.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                     # noqa: E402
from avatarcap_amd import synthetic as syn   # noqa: E402
import golden_inputs as gi                   # noqa: E402


def main():
    mg.install_stubs()
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    from network.arch_recon import ReconNetwork
    from network.unets import UnetNoCond7DS
    from network.HGFilters import HGFilter
    out = {}
    un = UnetNoCond7DS(input_nc=6, output_nc=64, nf=32, up_mode='upconv', use_dropout=False).eval()
    syn.load_synth(un, gi.SEED_NET)
    y = un(mg.t(gi.pos_map(256)[None])).numpy()[0]
    out['G7_unet256_samples'] = y[:, gi.PIX[:, 0] % 256, gi.PIX[:, 1] % 256]
    out['G7_unet256_absmean'] = np.float64(np.abs(y).mean())
    hgf = HGFilter(1, 4, 6, 32, 'group', 'no_down', False).eval()
    syn.load_synth(hgf, gi.SEED_NET)
    nm = gi.normal_maps(512)
    y = hgf(mg.t(nm[None]))[0][-1].numpy()[0]
    assert y.shape == (32, 256, 256)
    out['G7_hg512_samples'] = y[:, gi.PIX[:, 0] % 256, gi.PIX[:, 1] % 256]
    out['G7_hg512_absmean'] = np.float64(np.abs(y).mean())
    rn = ReconNetwork().eval()
    syn.load_synth(rn, gi.SEED_NET)
    pts = gi.query_points(104, 2048)
    items = {'cano_pts': mg.t(pts[None]), 'cano_smpl_center': mg.t(gi.center()[None]), 'front_normal': mg.t(nm[None, :3]), 'back_normal': mg.t(nm[None, 3:])}
    out['G6_infer512'] = rn.infer(items).numpy()
    path = os.path.join(HERE, 'producers_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == '__main__':
    main()
