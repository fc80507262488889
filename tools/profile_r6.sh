#!/bin/bash
# Round 6's profiles in one gpurun call (every step under its own `timeout`; rocprofv3 with --output-format csv; counters only ever with --kernel-trace):
#   1 kernel stats of the bench command                       -> r06_bench_kernel_stats.md + r06_bench_prof_line.json
#   2 kernel stats of main.py end to end (4 and 20 frames)    -> r06_main_kernel_stats_{4,20}.md  (copy / fill launches per frame = the difference / 16)
#   3 PMC traffic of the dominant launch                      -> profiles/pmc_traffic.json (tools/pmc_traffic.py)
#   4 PMC matrix-pipe counters of the encoder's convolutions  -> r06_pmc_encoder.txt
#   5 the power-wall closing measurement                      -> r06_mfma_order.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6p; mkdir -p $O
stats() { ls $O/$1/*/$2_kernel_stats.csv $O/$1/$2_kernel_stats.csv 2>/dev/null | head -1; }
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b -o b -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-masked --no-configs \
    > $O/r06_bench_prof_line.json 2> $O/bench_prof.err
f=$(stats prof_b b); [ -n "$f" ] && python tools/summarize_prof.py $f $O/r06_bench_kernel_stats.md "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-masked --no-configs (MI355X, dense 256^3), round 6"
for n in 4 20; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_m$n -o m$n -- python main.py -c configs/example.yaml -m test --synthetic --frames $n --no-npz --save-ply \
      --output-dir /tmp/avc_prof_o$n > $O/main$n.log 2>&1
  f=$(stats prof_m$n m$n); [ -n "$f" ] && python tools/summarize_prof.py $f $O/r06_main_kernel_stats_$n.md "python main.py -c configs/example.yaml -m test --synthetic --frames $n --no-npz --save-ply, round 6"
  rm -rf /tmp/avc_prof_o$n
done
timeout 700 python tools/pmc_traffic.py collect $O/pmc_traffic > $O/pmc_traffic.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    --kernel-include-regex conv_mfma_kernel -d $O/pmc_enc -o e -- python tools/enc_perf.py --once > $O/pmc_enc.log 2>&1
python - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob('$O/pmc_enc/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('conv_mfma_kernel')[-1].split('(')[0]
        a = agg.setdefault(k, collections.Counter()); a[r['Counter_Name']] += float(r['Counter_Value']); a['rows:' + r['Counter_Name']] += 1
with open('$O/r06_pmc_encoder.txt', 'w') as out:
    out.write('rocprofv3 --pmc over python tools/enc_perf.py --once (HGFilter forwards at 512^2): per kernel variant <CT, PT, TAPS, TWC, NORM, OCC2> the SUM over its dispatches\n')
    for k, a in agg.items():
        d = int(a['rows:SQ_INSTS_MFMA']); busy = a['SQ_VALU_MFMA_BUSY_CYCLES'] / max(1.0, 1024.0 * a['GRBM_GUI_ACTIVE'] / 8.0)
        out.write(f"{k:36s} dispatches {d:4d}  MFMA-busy share {100*busy:5.1f} %  MFMAs {a['SQ_INSTS_MFMA']:.3e}  GRBM_GUI_ACTIVE {a['GRBM_GUI_ACTIVE']:.3e}  MFMA_BUSY {a['SQ_VALU_MFMA_BUSY_CYCLES']:.3e}\n")
print(open('$O/r06_pmc_encoder.txt').read())
PY
timeout 200 ./tools/ubench/mfma_order > $O/r06_mfma_order.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python - <<PY
import json
try:
    d = json.load(open('$O/r06_bench_prof_line.json')); r = d['roofline']; print('profiled bench line: fps', d['value'], 'avg_launch_ms', r['avg_launch_ms'], 'launches', r['launches'], 'traffic', r['traffic'])
except Exception as e: print('bench line:', e)
PY
head -12 $O/r06_bench_kernel_stats.md | cut -c1-170
for n in 4 20; do echo "== main.py $n frames"; grep -i "copyBuffer\|fillBuffer" $O/r06_main_kernel_stats_$n.md | cut -c1-150; tail -2 $O/main$n.log | cut -c1-200; done
cat $O/r06_mfma_order.txt; tail -3 $O/pmc_traffic.log
