#!/bin/bash
# Memory-side PMC passes of the dominant kernel (dense 256^3 avatar query), round 5: two rocprofv3 --pmc runs of their own (counters + --kernel-trace only).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
i=0
for C in "FETCH_SIZE TCC_REQ" "WRITE_SIZE TCC_HIT TCC_MISS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv --kernel-include-regex "avatar_kernel|column_terms_kernel" -d $O/pmc_r5a/p$i -o p$i -- python tools/pmc_probe.py 256 1 grid > $O/pmc_r5a_p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob('$O/pmc_r5a/p*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = ('column_terms_kernel' if 'column_terms' in r['Kernel_Name'] else 'avatar_kernel<true,false,1>')
        agg.setdefault((k, r['Counter_Name']), []).append(float(r['Counter_Value']))
with open('$O/r05_pmc_avatar.txt', 'w') as out:
    for (k, c), v in agg.items():
        out.write(f'{k:30s} {c:28s} rows={len(v):4d} last={v[-1]:.6g}\n')
print(open('$O/r05_pmc_avatar.txt').read())
PY
