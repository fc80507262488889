#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r3o -o ov -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-masked --no-configs > gpurun_out/r3o_line.json 2> gpurun_out/r3o.err
python tools/overlap_trace.py $(ls gpurun_out/prof_r3o/*/ov_kernel_trace.csv gpurun_out/prof_r3o/ov_kernel_trace.csv 2>/dev/null | head -1)
python -c "import json; d=json.load(open('gpurun_out/r3o_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
