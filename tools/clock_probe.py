"""Is the fused query running at a power-limited clock?  Queues ~8 s of dense 256^3 queries and samples
rocm-smi (sclk, power) while they run; also times a pure-MFMA-free reference point (idle) before."""
import subprocess, sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import config, synthetic as syn
config.cfg = config.default_cfg()
from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
from avatarcap_amd.grid import generate_volume_points


def smi(tag):
    out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showmaxpower'], capture_output=True, text=True).stdout
    keep = [l.strip() for l in out.splitlines() if any(k in l for k in ('sclk', 'fclk', 'mclk', 'Power', 'power'))]
    print(tag, ' | '.join(' '.join(k.split()[2:]) for k in keep), flush=True)


net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to('cuda').eval()
sd = syn.synth_state_dict(syn.module_shapes(net), syn.SEED)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.warping_field.pose_feat_map = torch.randn(1, 64, 256, 256, device='cuda')
pts = generate_volume_points(syn.CANO_BOUNDS, (256, 256, 256), 'cuda')[None]
batch = {'cano_pts': pts, 'cano_smpl_center': torch.zeros(1, 3, device='cuda')}
q = OccupancyNet(net)
q.query(batch); torch.cuda.synchronize()
smi('idle :')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(130): q.query(batch)
e1.record()
for i in range(6):
    time.sleep(0.8); smi('busy%d:' % i)
torch.cuda.synchronize()
print('avg ms per query: %.2f' % (e0.elapsed_time(e1) / 130))
