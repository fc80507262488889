#!/bin/bash
# Bisecting the irreproducible --sync-io loop: the flaky test's configuration, N runs per variant, every .npz member against the first ASYNC run.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-10}; T=/tmp/avc_flaky; rm -rf $T; mkdir -p $T
cat > $T/cfg.yaml <<Y
training: {training_data_dir: null}
testing: {vol_res: [48, 64, 32], recon_net_ckpt: null, net_ckpt: null, net_ckpt_finetuned: null, testing_data_dir: null, output_dir: null}
model: {cano_template: {pos_encoding: 10}, warping_field: {pos_encoding: 0}}
Y
BASE="python main.py -c $T/cfg.yaml -m test --synthetic --frames 6"
$BASE --save-ply --nerf --output-dir $T/ref > $T/ref.log 2>&1
variant() { tag=$1; envs=$2; shift 2
  for i in $(seq 0 $((N-1))); do env $envs $BASE "$@" --output-dir $T/$tag$i > $T/$tag$i.log 2>&1 || echo "$tag$i failed"; done
  python - <<PY
import numpy as np
T='$T'; N=$N; tag='$tag'; bad=0; what=[]
for i in range(N):
    b=[]
    for f in range(6):
        try: a=np.load(f'{T}/{tag}{i}/%04d_mesh.npz' % f); r=np.load(f'{T}/ref/%04d_mesh.npz' % f)
        except Exception as e: b.append(('missing', f)); continue
        for k in a.files:
            if k in r.files and (a[k].shape != r[k].shape or not np.array_equal(a[k], r[k])): b.append((f, k)); break
    if b: bad+=1; what.append(b[0])
print(f'{tag:28s} {bad} of {N} runs differ', what[:4])
PY
  rm -rf $T/$tag*
}
variant sync A=1 --save-ply --sync-io
variant sync_noside AVC_LOOKAHEAD_SIDE=0 --save-ply --sync-io
