#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-12}; T=/tmp/avc_flaky; rm -rf $T; mkdir -p $T
cat > $T/cfg.yaml <<Y
training: {training_data_dir: null}
testing: {vol_res: [48, 64, 32], recon_net_ckpt: null, net_ckpt: null, net_ckpt_finetuned: null, testing_data_dir: null, output_dir: null}
model: {cano_template: {pos_encoding: 10}, warping_field: {pos_encoding: 0}}
Y
BASE="python main.py -c $T/cfg.yaml -m test --synthetic --frames 6 --save-ply"
$BASE --output-dir $T/ref > $T/ref.log 2>&1
for i in $(seq 0 $((N-1))); do AVC_DEBUG_FILL=1 $BASE --sync-io --output-dir $T/s$i > $T/s$i.log 2>&1 || echo "s$i failed"; done
python - <<PY
import numpy as np
T='$T'; N=$N
for i in range(N):
    for f in range(6):
        a=np.load(f'{T}/s{i}/%04d_mesh.npz' % f); r=np.load(f'{T}/ref/%04d_mesh.npz' % f)
        for k in ('cano_v','cano_vn','live_v','live_vn'):
            if a[k].shape==r[k].shape and not np.array_equal(a[k], r[k], equal_nan=False):
                d=(a[k]!=r[k]).any(1); rows=np.nonzero(d)[0]
                print(f'run {i} frame {f} {k}: rows {rows[0]}..{rows[-1]} ({rows.size}); NaN rows {int(np.isnan(a[k][rows]).any(1).sum())}; got', a[k][rows[0]], 'want', r[k][rows[0]], 'prev-frame same row', (np.load(f'{T}/s{i}/%04d_mesh.npz' % (f-1))[k][rows[0]] if f>0 and np.load(f'{T}/s{i}/%04d_mesh.npz' % (f-1))[k].shape[0]>rows[0] else None))
                break
PY
