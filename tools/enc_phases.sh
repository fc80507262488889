#!/bin/bash
# Where does a workgroup of the encoder's large convolution launches spend its time?  tools/ubench/enc_bench built with -DAVC_ENC_PHASES (s_memtime stamps per
# workgroup) on the launch shapes VERDICT round 5 names; XCD=1 repeats each with the band-per-XCD tile order.   usage: tools/enc_phases.sh [outfile]
OUT=${1:-gpurun_out/enc_phases.txt}
mkdir -p $(dirname $OUT); : > $OUT
for cfg in "conv 256 256 128 64 9 2 2 1 2" "conv 256 256 64 64 9 2 2 1 2" "conv 256 256 256 128 9 4 2 1 2" "conv 128 128 256 128 9 2 1 1 2" "conv 128 128 128 64 9 1 1 1 2" "conv 256 256 64 32 9 1 2 1 2" "conv 256 256 256 256 1 4 2 1 1"; do
  for x in 0 1; do echo "XCD=$x" >> $OUT; XCD=$x ./tools/ubench/enc_bench $cfg >> $OUT 2>&1; done
done
cat $OUT
