// gfx950, ROCm 7.2: a VGPR written by a packed-f32 VALU instruction (v_pk_add_f32 / v_pk_mul_f32: hipcc packs adjacent scalar f32 operations under -O3) and read
// as the DATA of a multi-dword store one instruction later was seen to reach memory STALE in the wave's last 16 lanes -- the value before the instruction's last
// pass -- when other kernels share the CU (the look-ahead U-Net on a side stream beside marching cubes / LBS: one x component in 16 consecutive vertices of a mesh,
// once in a few hundred launches; tools/race_probe.py, profiles/r06_store_hazard.md).  The hazard recogniser pads one wait state there.  settle() pins the
// values in registers, spends five wait states, and only then lets the store issue: no arithmetic, no change of any result.
#pragma once
#include <hip/hip_runtime.h>
namespace avc {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(AVC_NO_SETTLE)
__device__ __forceinline__ void settle(float &a, float &b, float &c) { asm volatile("s_nop 4" : "+v"(a), "+v"(b), "+v"(c)); }
__device__ __forceinline__ void settle(float &a, float &b, float &c, float &d) { asm volatile("s_nop 4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
typedef float settle_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void settle(settle_f32x4 &v) { asm volatile("s_nop 4" : "+v"(v)); }
#else
typedef float settle_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void settle(settle_f32x4 &) {}
__device__ __forceinline__ void settle(float &, float &, float &) {}
__device__ __forceinline__ void settle(float &, float &, float &, float &) {}
#endif
}  // namespace avc
