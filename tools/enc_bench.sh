#!/bin/bash
# Builds tools/ubench/enc_bench (here: hipcc cross-compiles) -- then `gpurun -- tools/enc_bench.sh run "<args>" ...` times launch configurations on the GPU box.
# usage: tools/enc_bench.sh build [extra hipcc flags]   |   tools/enc_bench.sh run "conv 256 256 256 128 9 4 2 1 2" "upadd 256 256 256" ...
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
    shift
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -I avatarcap_amd/csrc "$@" -o tools/ubench/enc_bench tools/ubench/enc_bench.hip
else
    shift
    for a in "$@"; do ./tools/ubench/enc_bench $a; done
fi
