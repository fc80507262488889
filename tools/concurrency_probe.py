"""Do kernels of a second HIP stream run beside the persistent fused query when it leaves CUs free?  Launches the dense 256^3 query on `mlp_blocks` CUs on
the current stream and a chain of small elementwise kernels on a side stream; reports when the side chain finishes relative to the query."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import _lib, config, synthetic as syn
config.cfg = config.default_cfg()
from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
from avatarcap_amd.grid import volume_axes
dev = torch.device('cuda')
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to(dev).eval(); syn.load_synth(net, syn.SEED)
net.warping_field.pose_feat_map = torch.randn(1, 64, 256, 256, device=dev)
axes = volume_axes(syn.CANO_BOUNDS, (256,) * 3, dev)
batch = {'cano_smpl_center': torch.zeros(1, 3, device=dev)}
q = OccupancyNet(net)
q.query_grid(batch, axes, (256,) * 3); torch.cuda.synchronize()
side = torch.cuda.Stream()
x = torch.randn(1 << 22, device=dev)
for blocks in (0, 240, 192):
    _lib.set_option('mlp_blocks', blocks)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    e[0].record()
    q.query_grid(batch, axes, (256,) * 3)
    e[1].record()
    with torch.cuda.stream(side):
        e[2].record(side)
        y = x
        for _ in range(200):
            y = y * 1.0001 + 0.5
        e[3].record(side)
    torch.cuda.synchronize()
    print(f'mlp_blocks {blocks or 256}: query {e[0].elapsed_time(e[1]):.2f} ms; side chain (200 small kernels) started {e[0].elapsed_time(e[2]):.2f} ms and ended '
          f'{e[0].elapsed_time(e[3]):.2f} ms after the query was enqueued')
_lib.set_option('mlp_blocks', 0)
