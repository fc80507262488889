#!/bin/bash
# The entry point end to end on the GPU box: `main.py -c configs/example.yaml -m test --synthetic --frames N` with the loop's own
# timing, in four shapes: no outputs, PLY (the reference's mesh output), npz + PLY, and the reference's blocking loop shape (--sync-io).
# usage: tools/main_e2e.sh [frames] [outdir]
N=${1:-32}; OUT=${2:-gpurun_out/e2e}
mkdir -p $OUT; TMP=${TMPDIR:-/tmp}/avc_e2e; rm -rf $TMP; mkdir -p $TMP
run() { tag=$1; shift; python main.py -c configs/example.yaml -m test --synthetic --frames $N --output-dir $TMP/$tag --timing-json $OUT/$tag.json "$@" > $OUT/$tag.log 2>&1; echo "$tag rc=$?"; du -sh $TMP/$tag 2>/dev/null | tail -1; rm -rf $TMP/$tag; }
run none --no-npz
run ply --no-npz --save-ply
run npz_ply --save-ply
run sync_npz_ply --save-ply --sync-io
run sync_none --no-npz --sync-io
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/*.json')):
    t=json.load(open(f)); print(f.split('/')[-1], {k: (round(v,2) if isinstance(v,float) else v) for k,v in t.items() if k in ('e2e_ms_per_frame','device_ms_per_frame','host_enqueue_ms_per_frame','writer_tail_ms','bytes_written','waited_for_writer_slot_ms','first_frame_ms')})
PY
df -h $TMP | tail -1
