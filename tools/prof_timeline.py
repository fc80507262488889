"""Kernel timeline of the last encoder frame in a rocprofv3 rocpd database (gpurun_out/prof_enc/enc_results.db): start, duration, gap to the
previous kernel's end on the same queue, grid, name.  `python tools/prof_timeline.py <db> [frame index, default the last]`"""
import sqlite3
import sys

db = sys.argv[1]
first = 's2d_kernel'
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x, workgroup_x, stream_id, queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if first in r[0]]
i0 = idx[which]
seq = rows[i0:(idx[which + 1] if which != -1 and which + 1 < len(idx) else len(rows))]
t0 = seq[0][1]
last_end = {}
tot = {}
for n, s, e, g, w, st, q in seq:
    short = n.split('(')[0].replace('avc::enc::', '').replace('void ', '')
    if 'copyBuffer' in short and s - t0 > 1e6:
        break
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    print(f'{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap:7.1f}  q{q} grid {g // max(w, 1):5d}  {short[:60]}')
    last_end[q] = e
    tot[short] = tot.get(short, 0.0) + (e - s) / 1e3
    end = e
print(f'frame span {(end - t0) / 1e3:.1f} us')
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f'{v:9.1f} us  {k}')
