#!/bin/bash
# Round 5's profiles in one gpurun call: kernel stats of the full chained frame (BASELINE configs[2]) and of the bench command.  Outputs under gpurun_out/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r5f -o r5f -- python tools/full_frame_prof.py 3 merge > $O/r5_ff.log 2>&1
python tools/summarize_prof.py $(ls $O/prof_r5f/*/r5f_kernel_stats.csv $O/prof_r5f/r5f_kernel_stats.csv 2>/dev/null | head -1) $O/r05_full_frame_kernel_stats.md \
    "BASELINE configs[2] chained frame x 4 (tools/full_frame_prof.py 3 merge), round 5"
tail -3 $O/r5_ff.log
