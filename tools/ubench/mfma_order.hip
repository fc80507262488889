// Microbenchmark (round 3): at the power cap the fused query's wall time is energy, not cycles (profiles/r03_power_wall.md).  Does the ORDER of the
// six MFMAs of a k-step (two output tiles x three split-fp16 passes) or the ENTROPY of the `lo` operands change what the part sustains?
//   order 0  t0.hh t1.hh t0.hl t1.hl t0.lh t1.lh   (shipped: consecutive MFMAs never share an accumulator)
//   order 1  t0.hh t0.hl t0.lh t1.hh t1.hl t1.lh   (three-long dependent chains: the accumulator can stay in the matrix pipe's forwarding path)
//   trunc B  the B `lo` fragment (activation residuals) keeps 4 of its 11 significant bits
//   trunc A  the A `lo` fragments (weight residuals) keep 4 significant bits
//   no LDS   (round 6, TRUNC == 3) the A fragments are read from LDS ONCE, four k-steps' worth kept in registers and rotated: the same MFMAs on operands of the
//            same entropy without a single ds_read in the loop -- what the LDS operand path costs at the power cap
// Full-entropy fp16 operands (hashed), A fragments from LDS one k-step ahead (4 x ds_read_b128 per k-step), B fragments rotate through 8 registers sets,
// 256 workgroups x 4 waves, one wave per SIMD.  hipcc --offload-arch=gfx950 -O3 -o mfma_order mfma_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// a finite fp16 with a random sign / mantissa and an exponent in [2^-3, 2^0) (hi) -- `lo` operands are the same scaled by 2^-11
__device__ __forceinline__ unsigned short rnd_half(unsigned seed, int exp_bias, unsigned mant_mask)
{
    const unsigned h = hash(seed);
    return (unsigned short)(((h >> 31) << 15) | (((h >> 10) % 3 + 12 + exp_bias) << 10) | ((h & 0x3ff) & mant_mask));
}

template <int ORDER, int TRUNC>
__global__ __launch_bounds__(256, 1) void bench(float *out, long long *cyc, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const unsigned am = TRUNC == 2 ? 0x380u : 0x3ffu, bm = TRUNC == 1 ? 0x380u : 0x3ffu;
    // LDS: 32 units of [hi 1 KiB | lo 1 KiB]
    for (int i = threadIdx.x; i < 32768; i += 256) {
        const int unit = i >> 10, w = i & 1023, lo = w >= 512;
        reinterpret_cast<unsigned short *>(smem)[i] = rnd_half(i * 2654435761u + blockIdx.x, lo ? -11 : 0, lo ? am : 0x3ffu);
        (void)unit;
    }
    __syncthreads();
    half8 bh[8], bl[8];
    for (int s = 0; s < 8; ++s)
        for (int i = 0; i < 8; ++i) {
            unsigned short a = rnd_half((s * 64 + lane) * 8 + i + 77777u, 0, 0x3ffu), b = rnd_half((s * 64 + lane) * 8 + i + 99999u, -11, bm);
            bh[s][i] = __builtin_bit_cast(_Float16, a); bl[s][i] = __builtin_bit_cast(_Float16, b);
        }
    const unsigned addr = lane * 16;
    f32x16 acc[2];
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    half8 ah[2][2], al[2][2];
    for (int t = 0; t < 2; ++t) { ah[0][t] = *reinterpret_cast<const half8 *>(smem + addr + t * 2048); al[0][t] = *reinterpret_cast<const half8 *>(smem + addr + t * 2048 + 1024); }
    half8 rh[4][2], rl[4][2];                      // TRUNC == 3: four k-steps of A fragments, register-resident
    if (TRUNC == 3)
        for (int k = 0; k < 4; ++k)
            for (int t = 0; t < 2; ++t) { rh[k][t] = *reinterpret_cast<const half8 *>(smem + addr + (k * 2 + t) * 2048); rl[k][t] = *reinterpret_cast<const half8 *>(smem + addr + (k * 2 + t) * 2048 + 1024); }
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int cur = u & 1, nxt = cur ^ 1, s = u & 7;
            if (TRUNC == 3) {
                const int k = u & 3;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh[k][0], bh[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh[k][1], bh[s], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh[k][0], bl[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh[k][1], bl[s], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rl[k][0], bh[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rl[k][1], bh[s], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[nxt][t] = *reinterpret_cast<const half8 *>(smem + addr + ((((u + 1) & 15) * 2 + t) * 2048));
                al[nxt][t] = *reinterpret_cast<const half8 *>(smem + addr + ((((u + 1) & 15) * 2 + t) * 2048 + 1024));
            }
            if (ORDER == 0) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][0], bh[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][1], bh[s], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][0], bl[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][1], bl[s], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][0], bh[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][1], bh[s], acc[1], 0, 0, 0);
            } else {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][0], bh[s], acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][0], bl[s], acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][0], bh[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][1], bh[s], acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][1], bl[s], acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][1], bh[s], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // keep the accumulators bounded (and the loop honest): one cheap rescale per 96 MFMAs
        if ((it & 63) == 63) { for (int r = 0; r < 16; ++r) { acc[0][r] *= 1e-3f; acc[1][r] *= 1e-3f; } }
    }
    const long long t1 = clock64();
    float sum = 0;
    for (int r = 0; r < 16; ++r) sum += acc[0][r] + acc[1][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int ORDER, int TRUNC>
void run(const char *name)
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 40000;
    hipFuncSetAttribute((const void *)bench<ORDER, TRUNC>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((bench<ORDER, TRUNC>), dim3(256), dim3(256), 65536, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((bench<ORDER, TRUNC>), dim3(256), dim3(256), 65536, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= 256;
    const double per_mfma = avg / (iters * 96.0);
    printf("%-44s %6.2f cycles/MFMA  kernel %7.2f ms  clock %4.0f MHz  %5.0f TFLOP/s\n", name, per_mfma, ms, avg / (ms * 1e3),
           32.0 * 32 * 16 * 2 * 96 * 4 * 256 * (double)iters / (ms * 1e9));
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>("order 0 (alternating tiles), full entropy");
        run<1, 0>("order 1 (chains of 3 per tile), full entropy");
        run<0, 1>("order 0, B lo keeps 4 significant bits");
        run<0, 2>("order 0, A lo keeps 4 significant bits");
        run<0, 3>("order 0, A in registers (no ds_read), full entropy");
    }
    return 0;
}
