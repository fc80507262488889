"""What does assembling the PLY payload on the device cost?  (main.py --save-ply: two meshes of ~0.53 M vertices / 1.06 M faces per frame)"""
import sys, time
import torch
sys.path.insert(0, '.')
from avatarcap_amd.utils import obj_io
V, F = 530000, 1060000
v, n = torch.randn(V, 3, device='cuda'), torch.randn(V, 3, device='cuda')
f = torch.randint(0, V, (F, 3), dtype=torch.int32, device='cuda')
c = torch.rand(V, 3, device='cuda')
for name, cc in (('no colours', None), ('colours', c)):
    for _ in range(3): obj_io.ply_records_device(v, f, n, cc)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): obj_io.ply_records_device(v, f, n, cc)
    torch.cuda.synchronize(); print(name, '%.3f ms per mesh' % ((time.perf_counter() - t) / 20 * 1e3))
