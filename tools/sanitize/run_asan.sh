#!/bin/bash
# AddressSanitizer pass over the C-ABI library on the GPU box (sanitizers).
#   1. here (no GPU needed):  python -m avatarcap_amd.build --asan        -> avatarcap_amd/libavcap_hip_asan.so  (host AND device code instrumented,
#                             gfx950:xnack+; fused_mlp.hip is left plain -- hipcc 7.2 crashes instrumenting it)
#                             hipcc --offload-arch=gfx950:xnack+ -fsanitize=address -shared-libasan tools/sanitize/asan_probe.hip -o tools/sanitize/asan_probe
#   2. on the box:            gpurun -- tools/sanitize/run_asan.sh  -> gpurun_out/asan/{log.txt, summary.txt}
# This image has no ASAN build of the HIP runtime, so a device-side report arrives as "Hostcall: no handler found for service ID 4" (the report service);
# the control runs of asan_probe show the signal is there for a one-element overrun and absent for an in-bounds kernel.  Host-side findings come as the
# usual "ERROR: AddressSanitizer" reports.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/asan; mkdir -p $OUT
export HSA_XNACK=1
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=0
SIG='service ID 4|ERROR: AddressSanitizer'
bad=$(LD_PRELOAD=$RT ./tools/sanitize/asan_probe bad 2>&1 | grep -cE "$SIG")
ok=$(LD_PRELOAD=$RT ./tools/sanitize/asan_probe ok 2>&1 | grep -cE "$SIG")
# the ragged-size exercise of the C-ABI without Python (torch's bundled HIP runtime aborts under the sanitizer's hsa allocation interceptor):
#   hipcc --offload-arch=gfx950:xnack+ -fsanitize=address -shared-libasan -O1 -std=c++17 -I include tools/sanitize/asan_driver.cpp -L avatarcap_amd -lavcap_hip_asan \
#         -Wl,-rpath,'$ORIGIN/../../avatarcap_amd' -o tools/sanitize/asan_driver
LD_PRELOAD=$RT timeout 1500 ./tools/sanitize/asan_driver > $OUT/log.txt 2>&1
rc=$?
{
  echo "xnack: $(rocminfo | grep -m1 -i 'xnack enabled')"
  echo "control, one-element overrun : $bad report line(s) (expected >= 1)"
  echo "control, in-bounds kernel    : $ok report line(s) (expected 0)"
  echo "driver exit status           : $rc"
  echo "driver summary               : $(grep asan_driver: $OUT/log.txt | tail -1)"
  echo "device / host ASAN reports   : $(grep -cE "$SIG" $OUT/log.txt)"
  grep -E "$SIG" $OUT/log.txt | sort | uniq -c | head -20
  grep -E "SUMMARY: AddressSanitizer" $OUT/log.txt | sort | uniq -c | head -20
} > $OUT/summary.txt
cat $OUT/summary.txt
