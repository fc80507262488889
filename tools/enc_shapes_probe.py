import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from avatarcap_amd import config
config.cfg = config.default_cfg(); config.device = torch.device('cuda')
import test_gpu_producers as T
hg = T._hg()
rs = np.random.RandomState(0)
for (H, W) in [(64, 128), (96, 160), (128, 64), (160, 96), (32, 32), (32, 224), (288, 352), (480, 512)]:
    x = torch.from_numpy(rs.randn(1, 6, H, W).astype(np.float32)).cuda()
    try:
        import io, contextlib
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            worst = T._walk_plan(hg, x)
        print(f'HGFilter {H}x{W}: worst {worst:.3e}')
    except AssertionError as e:
        print(f'HGFilter {H}x{W}: FAILED {str(e)[:300]}')
    except Exception as e:
        print(f'HGFilter {H}x{W}: {type(e).__name__} {str(e)[:300]}')
