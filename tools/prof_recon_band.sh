cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_query.py -m gpu -x -q -k "recon_grid_subset" -s 2>&1 | grep -E "^res|passed|failed|Error|assert" | head -20 > $O/r5_t6.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rb -o rb -- python tools/recon_band_perf.py > $O/r5_rb.log 2>&1
python tools/summarize_prof.py $(ls $O/prof_rb/*/rb_kernel_stats.csv $O/prof_rb/rb_kernel_stats.csv 2>/dev/null | head -1) $O/r05_recon_band_kernel_stats.md "tools/recon_band_perf.py (recon query: band / dense, folded / point-by-point; 256^3 and 384x384x128), round 5"
cat $O/r5_t6.log; cat $O/r05_recon_band_kernel_stats.md
