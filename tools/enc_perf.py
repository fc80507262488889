"""Timing of the HIP image encoder (csrc/conv_enc.hip) at the reference's 512^2 input: hipGraph replay vs plain launches, split-K and the
second stream on / off.  `python tools/enc_perf.py [--iters N] [--once]` on the GPU box (--once: one forward per setting, for rocprofv3)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import golden_inputs as gi                                             # noqa: E402
from avatarcap_amd import _lib, synthetic as syn                       # noqa: E402
from avatarcap_amd.network.HGFilters import HGFilter                   # noqa: E402


def unet(a):
    from avatarcap_amd.network.unets import UnetNoCond7DS
    un = UnetNoCond7DS(input_nc=6, output_nc=64, nf=32).to('cuda').eval()
    syn.load_synth(un, gi.SEED_NET)
    x = torch.from_numpy(gi.pos_map(256)[None]).cuda()
    from avatarcap_amd import config
    with torch.no_grad():
        for graph, ksplit in ([(1, 1)] if a.once else [(1, 1), (0, 1), (1, 0)]):
            config.hg_graph = bool(graph)
            _lib.set_option('enc_ksplit', ksplit)
            for _ in range(3):
                un(x, bind=True)
            torch.cuda.synchronize()
            if a.once:
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                un(x, bind=True)
            e1.record()
            torch.cuda.synchronize()
            print(f'U-Net 256^2  graph={graph} ksplit={ksplit}: {e0.elapsed_time(e1) / a.iters:.3f} ms per frame (incl. the NCHW copy of the result)', flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--res', type=int, default=512)
    ap.add_argument('--once', action='store_true')
    ap.add_argument('--unet', action='store_true', help="time the warping field's U-Net (256^2 position map) instead of HGFilter")
    a = ap.parse_args()
    if a.unet:
        return unet(a)
    hg = HGFilter(1, 4, 6, 32, 'group', 'no_down', False).to('cuda').eval()
    syn.load_synth(hg, gi.SEED_NET)
    x = torch.from_numpy(gi.normal_maps(a.res)[None]).cuda()
    settings = [(1, 1, 1, 1)] if a.once else [(1, 1, 1, 1), (1, 1, 1, 0), (1, 1, 1, 1), (1, 1, 1, 0), (0, 1, 1, 1), (1, 1, 0, 1), (1, 0, 1, 1)]
    with torch.no_grad():
        for graph, ksplit, fork, occ2 in settings:
            _lib.set_option('enc_graph', graph)
            _lib.set_option('enc_ksplit', ksplit)
            _lib.set_option('enc_fork', fork)
            _lib.set_option('enc_occ2', occ2)
            for _ in range(1 if a.once else 3):
                hg.encode(x, want_feat=False, bind=True)
            torch.cuda.synchronize()
            if a.once:
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                hg.encode(x, want_feat=False, bind=True)
            e1.record()
            torch.cuda.synchronize()
            print(f'encoder {a.res}^2  graph={graph} ksplit={ksplit} fork={fork} occ2={occ2}: {e0.elapsed_time(e1) / a.iters:.3f} ms per frame', flush=True)


if __name__ == '__main__':
    main()
