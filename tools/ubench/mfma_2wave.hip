// Microbenchmark: would 2 waves/SIMD with 16x16x32 MFMAs (16 points per wave) hide the VALU/LDS work
// that one wave per SIMD with 32x32x16 cannot?  Models the fused-MLP inner loop: per (tile, k-step) two
// ds_read_b128 (A hi, A lo) feed three MFMAs, plus F filler VALU per MFMA triple.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int F, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS / 256) void bench(float *out, long long *cyc, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 65536 / 4; i += THREADS) reinterpret_cast<float *>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    half8 bh, bl;
    for (int i = 0; i < 8; ++i) { bh[i] = (_Float16)(0.001f * (lane + i)); bl[i] = (_Float16)(0.0002f * (lane - i)); }
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = 0.5f + lane * 0.01f + i;
    const unsigned addr = lane * 16;
    long long t0 = clock64();
    float sum = 0;
    if (MODE == 0) {            // 16x16x32, 4 accumulators in flight
        f32x4 acc[4] = {{0}, {0}, {0}, {0}};
        half8 ah[2], al[2];
        ah[0] = *reinterpret_cast<const half8 *>(smem + addr); al[0] = *reinterpret_cast<const half8 *>(smem + addr + 1024);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int cur = u & 1, nxt = cur ^ 1;
                ah[nxt] = *reinterpret_cast<const half8 *>(smem + addr + (((it * 16 + u + 1) * 2048) & 65535));
                al[nxt] = *reinterpret_cast<const half8 *>(smem + addr + (((it * 16 + u + 1) * 2048 + 1024) & 65535));
                acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur], bh, acc[u & 3], 0, 0, 0);
                acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur], bl, acc[u & 3], 0, 0, 0);
                acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cur], bh, acc[u & 3], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < F; ++k) f[k & 7] = __builtin_fmaf(f[k & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int i = 0; i < 4; ++i) sum += acc[i][0] + acc[i][3];
    } else if (MODE == 2) {     // 32x32x16, operands stay in registers: what do the LDS reads cost in clock?
        f32x16 acc[2] = {{0}, {0}};
        half8 ah = *reinterpret_cast<const half8 *>(smem + addr), al = *reinterpret_cast<const half8 *>(smem + addr + 1024);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[u & 1], 0, 0, 0);
                acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[u & 1], 0, 0, 0);
                acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[u & 1], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < F; ++k) f[k & 7] = __builtin_fmaf(f[k & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int i = 0; i < 2; ++i) sum += acc[i][0] + acc[i][15];
    } else {                    // 32x32x16, 2 accumulators (the shipped kernel's shape)
        f32x16 acc[2] = {{0}, {0}};
        half8 ah[2], al[2];
        ah[0] = *reinterpret_cast<const half8 *>(smem + addr); al[0] = *reinterpret_cast<const half8 *>(smem + addr + 1024);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int cur = u & 1, nxt = cur ^ 1;
                ah[nxt] = *reinterpret_cast<const half8 *>(smem + addr + (((it * 16 + u + 1) * 2048) & 65535));
                al[nxt] = *reinterpret_cast<const half8 *>(smem + addr + (((it * 16 + u + 1) * 2048 + 1024) & 65535));
                acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bh, acc[u & 1], 0, 0, 0);
                acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bl, acc[u & 1], 0, 0, 0);
                acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur], bh, acc[u & 1], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < F; ++k) f[k & 7] = __builtin_fmaf(f[k & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int i = 0; i < 2; ++i) sum += acc[i][0] + acc[i][15];
    }
    long long t1 = clock64();
    for (int i = 0; i < 8; ++i) sum += f[i];
    out[blockIdx.x * THREADS + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int F, int THREADS>
void run(const char *name)
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * THREADS * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    hipFuncSetAttribute((const void *)bench<MODE, F, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((bench<MODE, F, THREADS>), dim3(256), dim3(THREADS), 65536, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((bench<MODE, F, THREADS>), dim3(256), dim3(THREADS), 65536, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= 256;
    const double per_triple = avg / (iters * 16.0);
    // useful MFMA flops per CU cycle: waves * 3 MFMAs * flops / cycles
    const double flops = (MODE == 0 ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2) * 3 * (THREADS / 64);
    printf("%-22s F=%2d fillers/triple: %6.1f cycles per triple per wave -> %5.0f MFMA flop/clk/CU (peak 4096); kernel %.3f ms, clock64 rate %.0f MHz, %.0f TFLOP/s\n",
           name, F, per_triple, flops / per_triple, ms, avg / (ms * 1e3), flops * iters * 16.0 * 256 / (ms * 1e9));
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<1, 0, 256>("32x32x16 4 waves/CU"); run<1, 6, 256>("32x32x16 4 waves/CU"); run<1, 12, 256>("32x32x16 4 waves/CU"); run<1, 24, 256>("32x32x16 4 waves/CU");
    run<0, 0, 512>("16x16x32 8 waves/CU"); run<0, 3, 512>("16x16x32 8 waves/CU"); run<0, 6, 512>("16x16x32 8 waves/CU"); run<0, 12, 512>("16x16x32 8 waves/CU");
    run<0, 0, 256>("16x16x32 4 waves/CU"); run<0, 6, 256>("16x16x32 4 waves/CU");
    run<2, 0, 256>("32x32x16 regs only"); run<2, 6, 256>("32x32x16 regs only");
    return 0;
}
