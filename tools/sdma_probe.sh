cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6s; mkdir -p $O; T=/tmp/avc_sdma; rm -rf $T; mkdir -p $T
run() { tag=$1; shift; env "$@" python main.py -c configs/example.yaml -m test --synthetic --frames 24 --no-npz --save-ply --output-dir $T/$tag --timing-json $O/$tag.json > $O/$tag.log 2>&1; echo "$tag rc=$?"; rm -rf $T/$tag; }
run default A=1
run sdma1 HSA_ENABLE_SDMA=1
run sdma0 HSA_ENABLE_SDMA=0
run default2 A=1
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*.json')):
    t=json.load(open(f)); print(f.split('/')[-1], {k: (round(v,2) if isinstance(v,float) else v) for k,v in t.items() if k in ('e2e_ms_per_frame','device_ms_per_frame','writer_tail_ms','waited_for_writer_slot_ms')})
PY
