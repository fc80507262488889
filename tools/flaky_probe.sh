#!/bin/bash
# Which loop shape is not reproducible?  main.py (the flaky test's configuration) N times async and N times --sync-io; every .npz member compared against run 0 of --sync-io.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-8}; T=/tmp/avc_flaky; rm -rf $T; mkdir -p $T
cat > $T/cfg.yaml <<Y
training: {training_data_dir: null}
testing: {vol_res: [48, 64, 32], recon_net_ckpt: null, net_ckpt: null, net_ckpt_finetuned: null, testing_data_dir: null, output_dir: null}
model: {cano_template: {pos_encoding: 10}, warping_field: {pos_encoding: 0}}
Y
for i in $(seq 0 $((N-1))); do
  [ -n "$SKIP_ASYNC" ] && cp -r $T/sync0 $T/async$i 2>/dev/null || python main.py -c $T/cfg.yaml -m test --synthetic --frames 6 --save-ply --nerf --output-dir $T/async$i --io-slots 2 --io-threads 2 $EXTRA_ASYNC > $T/async$i.log 2>&1 || echo "async$i failed"
  python main.py -c $T/cfg.yaml -m test --synthetic --frames 6 --save-ply --nerf --output-dir $T/sync$i --sync-io > $T/sync$i.log 2>&1 || echo "sync$i failed"
done
python - <<PY
import numpy as np, glob, os
T='$T'; N=$N
ref={f: np.load(f'{T}/sync0/%04d_mesh.npz' % f) for f in range(6)}
for tag in ['sync%d' % i for i in range(1, N)] + ['async%d' % i for i in range(N)]:
    bad=[]
    for f in range(6):
        a=np.load(f'{T}/{tag}/%04d_mesh.npz' % f)
        for k in a.files:
            if a[k].shape != ref[f][k].shape: bad.append((f,k,'shape')); continue
            if not np.array_equal(a[k], ref[f][k]):
                d=np.abs(a[k].astype(np.float64)-ref[f][k].astype(np.float64)); rows=np.nonzero(d.reshape(d.shape[0],-1).max(1)>0)[0]
                bad.append((f,k,int(rows.size),float(d.max()),rows[:6].tolist(),rows[-1]))
    print(tag, 'identical' if not bad else bad)
PY
