#!/bin/bash
# usage: tools/run_pmc.sh <tag> [res] [mode]   -- separate rocprofv3 --pmc passes (never combined with sys/hip traces); mode: grid (avatar query) | recon
TAG=$1; RES=${2:-256}; MODE=${3:-grid}
if [ "$MODE" = recon ]; then KRE='recon_fold_kernel|recon_column_terms_kernel'; else KRE='avatar_kernel|column_terms_kernel'; fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE GRBM_COUNT" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_SMEM" \
         "FETCH_SIZE TCC_REQ" \
         "WRITE_SIZE TCC_HIT TCC_MISS" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv --kernel-include-regex "$KRE" -d $OUT/p$i -o p$i -- python tools/pmc_probe.py $RES 1 $MODE > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob('$OUT/p*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if 'avatar_kernel' not in name and 'column_terms_kernel' not in name and 'recon_fold_kernel' not in name: continue
        k = ('col:' if 'column_terms_kernel' in name else '') + r['Counter_Name']; agg.setdefault(k, []).append(float(r['Counter_Value']))
with open('$OUT/summary.txt', 'w') as out:
    for k, v in agg.items():
        # counters are reported per dispatch (possibly per dimension instance); sum instances of the LAST dispatch group
        line = f'{k:32s} n={len(v):4d} sum={sum(v):.6g} last={v[-1]:.6g} mean={sum(v)/len(v):.6g}'
        print(line); out.write(line + '\n')
PY
