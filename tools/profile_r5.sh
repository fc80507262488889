#!/bin/bash
# Round 5's profiles in one gpurun call: kernel stats of the bench command and of the full chained frame (BASELINE configs[2]); PMC passes (separate runs, counters +
# --kernel-trace only) on the band launches of both queries.  Outputs under gpurun_out/, to be copied into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r5b -o r5b -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-masked --no-configs \
    > $O/r05_bench_prof_line.json 2> $O/r5_bench_prof.err
python tools/summarize_prof.py $(ls $O/prof_r5b/*/r5b_kernel_stats.csv $O/prof_r5b/r5b_kernel_stats.csv 2>/dev/null | head -1) $O/r05_bench_kernel_stats.md \
    "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-masked --no-configs (MI355X, dense 256^3), round 5"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r5f -o r5f -- python tools/full_frame_prof.py 3 merge > $O/r5_ff.log 2>&1
python tools/summarize_prof.py $(ls $O/prof_r5f/*/r5f_kernel_stats.csv $O/prof_r5f/r5f_kernel_stats.csv 2>/dev/null | head -1) $O/r05_full_frame_kernel_stats.md \
    "BASELINE configs[2] chained frame x 4 (tools/full_frame_prof.py 3 merge), round 5"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "GRBM_GUI_ACTIVE FETCH_SIZE TCC_REQ" "WRITE_SIZE TCC_HIT TCC_MISS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv --kernel-include-regex "avatar_kernel|recon_fold_kernel|recon_kernel|column_terms|band_prepass" -d $O/pmc_r5/p$i -o p$i -- python tools/pmc_band.py 1 > $O/pmc_r5_p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob('$O/pmc_r5/p*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        k = ('avatar_kernel<.,.,2>' if 'avatar_kernel' in n else 'recon_fold_kernel<2>' if 'recon_fold_kernel' in n else 'recon_kernel (left-over tiles)' if 'recon_kernel' in n
             else 'recon_column_terms_kernel' if 'recon_column' in n else 'column_terms_kernel' if 'column_terms' in n else 'band_prepass_kernel')
        agg.setdefault((k, r['Counter_Name']), []).append(float(r['Counter_Value']))
with open('$O/r05_pmc_band.txt', 'w') as out:
    out.write('rocprofv3 --pmc passes over tools/pmc_band.py 1 (warm-up + 1 launch of each band query, 2,800,408 points at 256^3); per kernel and counter: dispatch rows, sum, last\n')
    for (k, c), v in agg.items():
        out.write(f'{k:34s} {c:28s} rows={len(v):4d} sum={sum(v):.6g} last={v[-1]:.6g}\n')
print(open('$O/r05_pmc_band.txt').read())
PY
python -c "
import json; d=json.load(open('$O/r05_bench_prof_line.json')); r=d['roofline']; print('profiled bench line: fps', d['value'], 'avg_launch_ms', r['avg_launch_ms'], 'launches', r['launches'])"
head -14 $O/r05_bench_kernel_stats.md | cut -c1-160
