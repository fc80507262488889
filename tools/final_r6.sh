#!/bin/bash
# Round 6's closing GPU session: the whole GPU suite, smoke(), the PMC traffic of the final sources, the default bench line (what the driver runs), and the
# rocprofv3 kernel stats of the bench command with the final library.  Every step under its own timeout.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6z; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 700 python tools/pmc_traffic.py collect $O/pmc_traffic > $O/pmc_traffic.log 2>&1
timeout 900 python bench.py > $O/r06_bench_line.json 2> $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b -o b -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-masked --no-configs \
    > $O/r06_bench_prof_line.json 2> $O/bench_prof.err
f=$(ls $O/prof_b/*/b_kernel_stats.csv $O/prof_b/b_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/summarize_prof.py $f $O/r06_bench_kernel_stats.md "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-masked --no-configs (MI355X, dense 256^3), round 6, final library"
AVC_FORCE_DIST=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-masked --no-configs > $O/bench_rccl1.json 2> $O/bench_rccl1.err
find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/pytest_gpu.log | tail -4; tail -2 $O/smoke.log; tail -3 $O/pmc_traffic.log
python - <<PY
import json
for f in ('r06_bench_line.json', 'r06_bench_prof_line.json', 'bench_rccl1.json'):
    try:
        d = json.load(open('$O/' + f)); r = d['roofline']
        print(f, 'fps %.3f ms/step %.2f avg_launch_ms %.2f frac %.4f mfma_util %.3f traffic %s (%s) clock %.0f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['mfma_util'], r['traffic'], r['traffic_ref']['state'], r['clock_mhz']))
        c = d.get('configs')
        if c and 'main_py_e2e' in c: print('  main_py_e2e', {k: round(v['ms_per_frame'], 2) for k, v in c['main_py_e2e']['legs'].items() if 'ms_per_frame' in v}, 'device', round(c['main_py_e2e']['device_figure_ms'], 2))
        if c: print('  configs[2] %.2f ms, example.yaml %.2f ms, hgfilter %.3f ms, unet %.3f' % (c['configs[2]']['ms_per_frame'], c['example.yaml']['ms_per_frame'], c['configs[2]']['stage_ms']['of which hgfilter'], c['configs[2]']['stage_ms']['of which unet7ds (in avatar_frame)']))
        if d['config'].get('exchange_autotune'): print('  autotune', d['config']['exchange_autotune']['choice'], [round(x, 1) for x in d['config']['exchange_autotune']['warmup_ab_ms']])
    except Exception as e:
        print(f, 'ERR', repr(e))
PY
head -12 $O/r06_bench_kernel_stats.md | cut -c1-170
