#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --marker-trace --kernel-trace --output-format csv -d gpurun_out/prof_r3m -o mk -- python tools/full_frame_prof.py 3 merge > gpurun_out/r3m.log 2>&1
ls gpurun_out/prof_r3m/ gpurun_out/prof_r3m/* | head -20
python - <<'PY'
import csv, glob, collections
f = sorted(glob.glob('gpurun_out/prof_r3m/**/*marker*trace*.csv', recursive=True))
print(f)
if f:
    rows = list(csv.DictReader(open(f[0])))
    print(rows[0].keys())
    agg = collections.OrderedDict()
    for r in rows:
        name = r.get('Function') or r.get('Message') or r.get('Name') or str(r)
        try:
            d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        except Exception:
            d = 0.0
        a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += d
    for k, v in agg.items():
        print(f'{k:40s} n={v[0]:4d} total host ms {v[1]:.2f}')
PY
