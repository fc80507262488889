#!/bin/bash
# Round 4's profiles in one gpurun call: kernel stats of the bench command, the full chained frame, the encoder's and the U-Net's timelines, and the
# PMC passes (separate runs, counters + --kernel-trace only) on the convolution kernel.  Outputs under gpurun_out/, to be copied into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r4b -o r4b -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-masked --no-configs \
    > $O/r4_bench_prof_line.json 2> $O/r4_bench_prof.err
python tools/summarize_prof.py $(ls $O/prof_r4b/*/r4b_kernel_stats.csv $O/prof_r4b/r4b_kernel_stats.csv 2>/dev/null | head -1) $O/r04_bench_kernel_stats.md \
    "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-masked --no-configs (MI355X, dense 256^3), round 4"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r4f -o r4f -- python tools/full_frame_prof.py 3 merge > $O/r4_ff.log 2>&1
python tools/summarize_prof.py $(ls $O/prof_r4f/*/r4f_kernel_stats.csv $O/prof_r4f/r4f_kernel_stats.csv 2>/dev/null | head -1) $O/r04_full_frame_kernel_stats.md \
    "BASELINE configs[2] chained frame x 4 (tools/full_frame_prof.py 3 merge), round 4: every kernel is this repository's (no MIOpen / rocBLAS launch left on the path)"
timeout 100 rocprofv3 --kernel-trace -d $O/prof_r4e -o enc -- python tools/enc_perf.py --once > $O/r4_enc.log 2>&1
python tools/prof_timeline.py $O/prof_r4e/enc_results.db > $O/r04_encoder_timeline.txt 2>&1
timeout 100 rocprofv3 --kernel-trace -d $O/prof_r4u -o unet -- python tools/enc_perf.py --unet --once > $O/r4_unet.log 2>&1
python tools/prof_timeline.py $O/prof_r4u/unet_results.db > $O/r04_unet_timeline.txt 2>&1
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE WRITE_SIZE TCC_HIT TCC_MISS"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $C --kernel-trace --output-format csv --kernel-include-regex "conv_mfma_kernel" -d $O/pmc_r4e/p$i -o p$i -- python tools/enc_perf.py --once > $O/pmc_r4e_p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob('$O/pmc_r4e/p*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
with open('$O/r04_pmc_encoder.txt', 'w') as out:
    out.write('rocprofv3 --pmc passes over conv_mfma_kernel launches of tools/enc_perf.py --once (3 HGFilter forwards at 512^2; sums over all conv launches)\n')
    for k, v in agg.items():
        out.write(f'{k:28s} dispatch-rows={len(v):5d} sum={sum(v):.6g}\n')
    b, m, w = sum(agg.get('SQ_BUSY_CYCLES', [0])), sum(agg.get('SQ_VALU_MFMA_BUSY_CYCLES', [0])), sum(agg.get('SQ_WAVE_CYCLES', [0]))
    if b: out.write(f'SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES = {m / b:.4f}\n')
    g = sum(agg.get('GRBM_GUI_ACTIVE', [0]))
    if g: out.write(f'SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE = {m / g:.4f}\n')
print(open('$O/r04_pmc_encoder.txt').read())
PY
tail -8 $O/r04_encoder_timeline.txt
