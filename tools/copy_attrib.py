"""Who launches the copy / fill kernels of a frame?  Runs main.py's frame loop in-process (synthetic, example.yaml grid) under torch.profiler and prints, per ATen op
that issued device memcpy / memset / copy kernels, the count per frame and the Python line that called it.  argv: [frames] [extra main.py flags ...]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import main as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
extra = sys.argv[2:]
argv = ['-c', 'configs/example.yaml', '-m', 'test', '--synthetic', '--frames', str(n), '--output-dir', os.environ.get('TMPDIR', '/tmp') + '/avc_attrib'] + extra
M.main(['-c', 'configs/example.yaml', '-m', 'test', '--synthetic', '--frames', '2', '--no-npz', '--output-dir', os.environ.get('TMPDIR', '/tmp') + '/avc_attrib'])   # warm: plans, graphs
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=False) as prof:
    M.main(argv)
ev = prof.events()
rows = {}
for e in ev:
    name = e.name
    if e.device_type == torch.autograd.DeviceType.CUDA and ('copy' in name.lower() or 'fill' in name.lower() or 'memcpy' in name.lower() or 'memset' in name.lower()):
        rows.setdefault(name, [0, 0.0]); rows[name][0] += 1; rows[name][1] += e.device_time if hasattr(e, 'device_time') else e.cuda_time
print('device copy/fill activity over %d frames:' % n)
for k, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
    print('  %-60s %6d  (%.1f per frame)  %.1f us total' % (k[:60], c, c / n, t))
print(prof.key_averages(group_by_stack_n=6).table(sort_by='self_cuda_time_total', row_limit=60, max_name_column_width=50, max_src_column_width=110))
