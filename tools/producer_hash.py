"""sha256 of the U-Net's and the HGFilter's outputs on fixed seeded inputs and weights (a before / after check for changes that must not move a bit)."""
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import config, synthetic as syn
dev = torch.device('cuda'); config.device = dev; config.cfg = config.default_cfg()
from avatarcap_amd.network.arch_avatar import GeoTexAvatar
from avatarcap_amd.network.arch_recon import ReconNetwork
rs = np.random.RandomState(1)
net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to(dev).eval(); syn.load_synth(net, syn.SEED)
rn = ReconNetwork().to(dev).eval(); syn.load_synth(rn, syn.SEED)
pos_map = torch.from_numpy(rs.randn(1, 6, 256, 256).astype(np.float32)).cuda()
img = torch.from_numpy(rs.randn(1, 6, 512, 512).astype(np.float32)).cuda()
u = net.warping_field.unet(pos_map)
h = rn.image_filter(img) if hasattr(rn, 'image_filter') else rn.get_feat_maps(img)
h = h if isinstance(h, torch.Tensor) else h[-1]
torch.cuda.synchronize()
print('unet', tuple(u.shape), hashlib.sha256(u.contiguous().cpu().numpy().tobytes()).hexdigest()[:16])
print('hgfilter', tuple(h.shape), hashlib.sha256(h.contiguous().cpu().numpy().tobytes()).hexdigest()[:16])
