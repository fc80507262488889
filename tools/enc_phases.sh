#!/bin/bash
# Where does a workgroup of the encoder's large convolution launches spend its time?  tools/ubench/enc_bench built with -DAVC_ENC_PHASES (s_memtime stamps per
# workgroup) on the launch shapes VERDICT round 5 names, each as one workgroup per CU (the round-5 shape) and as two half-height workgroups per CU (OCC2).
# usage: tools/enc_phases.sh [outfile]
OUT=${1:-gpurun_out/enc_phases.txt}
mkdir -p $(dirname $OUT); : > $OUT
run() { echo "OCC2=$1: $2" >> $OUT; OCC2=$1 timeout 60 ./tools/ubench/enc_bench $2 >> $OUT 2>&1; }
run 0 "conv 256 256 128 64 9 2 2 1 2";  run 1 "conv 256 256 128 64 9 2 1 1 2"
run 0 "conv 256 256 64 64 9 2 2 1 2";   run 1 "conv 256 256 64 64 9 2 1 1 2"
run 0 "conv 256 256 256 128 9 4 2 1 2"; run 1 "conv 256 256 256 128 9 2 1 1 2"
run 0 "conv 128 128 256 128 9 2 1 1 2"; run 1 "conv 128 128 256 128 9 1 1 1 2"
run 0 "conv 128 128 128 64 9 1 1 1 2"
run 0 "conv 256 256 64 32 9 1 2 1 2";   run 1 "conv 256 256 64 32 9 1 1 1 2"
run 0 "conv 256 256 256 256 1 4 2 1 1"; run 1 "conv 256 256 256 256 1 2 1 1 1"
cat $OUT
