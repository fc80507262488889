// Microbenchmark: how many independent filler instructions hide behind v_mfma_f32_32x32x16_f16 when ONE
// wave per SIMD issues them (the fused-MLP regime)?  Prints cycles per MFMA for K fillers per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int K>
__global__ __launch_bounds__(256, 1) void bench(float *out, long long *cyc, int iters, const float4 *gsrc)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 acc0 = {0}, acc1 = {0};
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = 0.5f + lane * 0.01f + i;
    unsigned addr = lane * 16;
    half8 l0 = a, l1 = b;
    half8 ring[8]; float4 gring[8];
    for (int i = 0; i < 8; ++i) { ring[i] = a; gring[i] = make_float4(0, 0, 0, 0); }
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) f[k & 7] = __builtin_fmaf(f[k & 7], 1.0001f, 0.5f);                  // plain VALU
                if (KIND == 1) f[k & 7] = __builtin_amdgcn_exp2f(f[k & 7]);                         // transcendental
                if (KIND == 2) { l0 = *reinterpret_cast<const half8 *>(smem + addr + ((k * 1024 + u * 4096) & 32767)); asm volatile("" :: "v"(l0)); }   // ds_read_b128
                if (KIND == 3) { f[k & 7] = acc1[(k + u) & 15] + 1.0f; }                            // reads of the other accumulator
                if (KIND == 4) { ring[(u * K + k) & 7] = *reinterpret_cast<const half8 *>(smem + addr + ((k * 1024 + u * 4096) & 32767)); }   // ds_read_b128, use deferred
                if (KIND == 5) { gring[(u * K + k) & 7] = gsrc[(size_t)(((it * 8 + u) * K + k) & 1023) * 64 + lane]; }                        // global_load_dwordx4 (L2-resident), use deferred
                if (KIND == 6) { *reinterpret_cast<half8 *>(smem + 32768 + addr + ((k * 1024 + u * 4096) & 32767)) = a; }                    // ds_write_b128
            }
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) f[(k + 4) & 7] = __builtin_fmaf(f[(k + 4) & 7], 1.0001f, 0.5f);
                if (KIND == 1) f[(k + 4) & 7] = __builtin_amdgcn_exp2f(f[(k + 4) & 7]);
                if (KIND == 2) { l1 = *reinterpret_cast<const half8 *>(smem + addr + ((k * 1024 + u * 4096 + 512) & 32767)); asm volatile("" :: "v"(l1)); }
                if (KIND == 3) { f[(k + 4) & 7] = acc0[(k + u) & 15] + 1.0f; }
                if (KIND == 4) { ring[(u * K + k + 4) & 7] = *reinterpret_cast<const half8 *>(smem + addr + ((k * 1024 + u * 4096 + 512) & 32767)); }
                if (KIND == 5) { gring[(u * K + k + 4) & 7] = gsrc[(size_t)(((it * 8 + u) * K + k + 512) & 1023) * 64 + lane]; }
                if (KIND == 6) { *reinterpret_cast<half8 *>(smem + 32768 + addr + ((k * 1024 + u * 4096 + 512) & 32767)) = b; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) s += f[i];
    s += (float)l0[0] + (float)l1[1];
    for (int i = 0; i < 8; ++i) s += (float)ring[i][0] + gring[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int K>
void run(const char *name)
{
    float *out; long long *cyc; float4 *gsrc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&gsrc, 1024 * 64 * 16); hipMemset(gsrc, 0, 1024 * 64 * 16);
    const int iters = 2000;
    hipFuncSetAttribute((const void *)bench<KIND, K>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((bench<KIND, K>), dim3(256), dim3(256), 65536, 0, out, cyc, iters, gsrc);
    hipDeviceSynchronize();
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= 256;
    printf("%-10s K=%d fillers/MFMA: %.1f clock64 ticks per MFMA\n", name, K, avg / (iters * 16.0));
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0, 0>("none");
    run<0, 2>("fma"); run<0, 4>("fma"); run<0, 6>("fma"); run<0, 8>("fma"); run<0, 12>("fma");
    run<1, 1>("exp2"); run<1, 2>("exp2"); run<1, 4>("exp2");
    run<2, 1>("ds_read"); run<2, 2>("ds_read"); run<2, 4>("ds_read");
    run<4, 1>("ds_read_d"); run<4, 2>("ds_read_d"); run<4, 4>("ds_read_d");
    run<5, 1>("gload_d"); run<5, 2>("gload_d");
    run<6, 1>("ds_write"); run<6, 2>("ds_write");
    return 0;
}
