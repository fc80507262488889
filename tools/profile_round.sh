#!/bin/bash
# usage (on the GPU box, through gpurun): tools/profile_round.sh <tag>
# Everything the round's profiles/ entries are made from, in one call: the bench line (with its configs[2] / configs[3] legs), rocprofv3 --kernel-trace
# --stats of the same command, the colour leg's kernel split.  PMC passes: tools/run_pmc.sh (separate passes, never combined with sys/hip traces).
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-masked --no-configs \
    > gpurun_out/${TAG}_bench_prof_line.json 2> gpurun_out/${TAG}_bench_prof.err
python tools/summarize_prof.py $(ls gpurun_out/prof_$TAG/*/${TAG}_kernel_stats.csv gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${TAG}_bench_kernel_stats.md \
    "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-masked --no-configs (MI355X, dense 256^3)"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_clr -o ${TAG}_clr -- python tools/colour_leg_probe.py > gpurun_out/${TAG}_clr.log 2>&1
python tools/summarize_prof.py $(ls gpurun_out/prof_${TAG}_clr/*/${TAG}_clr_kernel_stats.csv gpurun_out/prof_${TAG}_clr/${TAG}_clr_kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${TAG}_colour_leg_kernel_stats.md \
    "colour leg: avatar frame (band, 256^3) + 3 x colour_vertices on 200k vertices (tools/colour_leg_probe.py)"
