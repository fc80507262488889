#!/bin/bash
# main.py --synthetic end to end (avatar + fusion + recon + colours, files written) on odd grid shapes, band and dense: completes, every frame done, and two
# runs of the same configuration write the same bytes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=/tmp/avc_shapes; rm -rf $T; mkdir -p $T
for res in "33 47 19" "50 70 33" "64 40 128" "31 31 31" "40 56 24" "96 80 48"; do
  set -- $res
  cat > $T/cfg.yaml <<Y
training: {training_data_dir: null}
testing: {vol_res: [$1, $2, $3], recon_net_ckpt: null, net_ckpt: null, net_ckpt_finetuned: null, testing_data_dir: null, output_dir: null}
model: {cano_template: {pos_encoding: 10}, warping_field: {pos_encoding: 0}}
Y
  for valid in band dense; do
    ok=1
    for r in a b; do
      python main.py -c $T/cfg.yaml -m test --synthetic --frames 3 --save-ply --nerf --valid $valid --output-dir $T/o_$r > $T/log_$r.txt 2>&1 || ok=0
      grep -q "3 of 3 frames done" $T/log_$r.txt || ok=0
    done
    same=$(python - <<PY
import numpy as np, glob, os
a=sorted(glob.glob('$T/o_a/*')); same=len(a)>0
for f in a:
    g=f.replace('/o_a/','/o_b/')
    if not os.path.exists(g): same=False; continue
    if f.endswith('.npz'):
        x,y=np.load(f),np.load(g); same &= sorted(x.files)==sorted(y.files) and all(np.array_equal(x[k],y[k]) for k in x.files)
    else: same &= open(f,'rb').read()==open(g,'rb').read()
v=[np.load(f)['cano_v'].shape[0] for f in a if f.endswith('.npz')]
print(same, 'files', len(a), 'avatar vertices', v)
PY
)
    echo "vol_res $res $valid: completed=$ok reproducible=$same"; [ $ok = 1 ] || tail -5 $T/log_a.txt
    rm -rf $T/o_a $T/o_b
  done
done
