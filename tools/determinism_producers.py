"""Are the two MIOpen producers run-to-run deterministic, and what does torch.backends.cudnn.deterministic cost?  (GPU box)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import golden_inputs as gi
from avatarcap_amd import config, synthetic as syn
config.cfg = config.default_cfg()
from avatarcap_amd.network.unets import UnetNoCond7DS
from avatarcap_amd.network.HGFilters import HGFilter
un = UnetNoCond7DS(input_nc=6, output_nc=64, nf=32).to('cuda').eval(); syn.load_synth(un, gi.SEED_NET)
hg = HGFilter(1, 4, 6, 32, 'group', 'no_down', False).to('cuda').eval(); syn.load_synth(hg, gi.SEED_NET)
x = torch.from_numpy(gi.pos_map(256)[None]).cuda(); nm = torch.from_numpy(gi.normal_maps(512)[None]).cuda()
for det in (False, True):
    with torch.no_grad(), torch.backends.cudnn.flags(enabled=True, benchmark=False, deterministic=det):
        for name, f in (('unet', lambda: un(x)), ('hgfilter', lambda: hg(nm)[0][-1])):
            outs = [f().clone() for _ in range(6)]
            diffs = [float((o - outs[0]).abs().max()) for o in outs[1:]]
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10): f()
            torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 10 * 1e3
            print(f'deterministic={det} {name}: max |call_k - call_0| = {max(diffs):.3e} ({diffs}), {ms:.2f} ms/call', flush=True)
# per-layer hunt: which module of the U-Net is not repeatable?
with torch.no_grad():
    acts = {}
    def hook(name):
        def h(m, i, o):
            acts.setdefault(name, []).append(o.detach().clone())
        return h
    hs = [m.register_forward_hook(hook(n)) for n, m in un.named_modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d))]
    for _ in range(4): un(x)
    for n, v in acts.items():
        d = max(float((a - v[0]).abs().max()) for a in v[1:] if a.shape == v[0].shape)       # upconv3 runs twice per forward with two shapes
        print('  layer', n, tuple(v[0].shape), 'max diff across calls', d)
