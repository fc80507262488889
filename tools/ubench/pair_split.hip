// Microbenchmark: would TWO waves per SIMD pay for the fused-MLP chunk loop?  A 256-wide layer's activations of 32 points are 2 x 128 registers, so a
// second wave per SIMD needs the layer split over a wave PAIR.  Model of the K-split form that fits the LDS:
//   * workgroup = 8 waves = 4 pairs; a pair owns 32 points; wave h of the pair holds the B fragments of half the k-steps;
//   * LDS: weight ring 2 x 32 KiB (a chunk = 8 k-steps x 2 output tiles), exchange buffers 8 x 2 x 4 KiB, (park area 32 KiB not modelled);
//   * per chunk a wave runs its 4 k-steps: 16 ds_read_b128 of A fragments, 24 MFMAs on two alternating accumulators, 4 LDS-DMA pieces of the next
//     chunk, its share of the epilogue (a tile per wave every two chunks: 1 value pair per k-step, 16 VALU of which 4 transcendental, as in the shipped
//     form), writes half a tile of partial sums for its partner (2 ds_write_b128) and, after the chunk barrier, reads its partner's (2 ds_read_b128 + 8 v_add_f32).
// Beside it the shipped form (copy_cost.hip fill 2 mode 6: one wave per SIMD, 64 KiB chunks, 96 MFMAs per chunk per wave).  Both run the same number
// of MFMAs per SIMD; compare the kernel milliseconds (the clock is power-managed).        hipcc --offload-arch=gfx950 -O3 pair_split.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int STREAM = 52 * 65536;

__device__ __forceinline__ void filler(float (&f)[8], int i)
{
    if ((i & 3) == 1) f[i & 7] = __builtin_amdgcn_exp2f(f[i & 7]);
    else if ((i & 3) == 3) f[i & 7] = __builtin_amdgcn_logf(f[i & 7] + 1.5f);
    else f[i & 7] = __builtin_fmaf(f[i & 7], 1.0001f, 0.5f);
}

// one k-step: 6 MFMAs (two tiles x hh, hl, lh) with VPG fillers behind each
template <int VPG>
__device__ __forceinline__ void kstep(f32x16 (&acc)[2], const half8 (&ah)[2], const half8 (&al)[2], half8 bh, half8 bl, float (&f)[8])
{
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh, acc[t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < VPG; ++i) filler(f, 6 * t + i);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl, acc[t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < VPG; ++i) filler(f, 6 * t + 3 + i);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh, acc[t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < VPG; ++i) filler(f, 12 + 2 * t + i);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// WPS = waves per SIMD (1: shipped form, 64 KiB chunks, 16 k-steps per wave; 2: K-split pairs, 32 KiB chunks, 4 k-steps per wave)
template <int WPS>
__global__ __launch_bounds__(256 * WPS, 1) void bench(float *out, long long *cyc, int chunks, const char *gsrc)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 4 * WPS, CHUNK = WPS == 1 ? 65536 : 32768, KS = WPS == 1 ? 16 : 4, PIECES = CHUNK / NW / 1024;
    constexpr int VPG = 3;                                                  // fillers per MFMA gap: the epilogue is ~16 VALU per k-step of a wave in both forms
    constexpr int EXCH = 2 * CHUNK;                                         // exchange area behind the ring (WPS == 2)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = wave & 1;
    for (int i = threadIdx.x; i < 2 * CHUNK / 4; i += 256 * WPS) reinterpret_cast<float *>(smem)[i] = 0.001f * ((i * 7) & 255) - 0.1f;
    if (WPS == 2) for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<float *>(smem + EXCH)[i] = 0.25f;
    __syncthreads();
    half8 bh, bl;
    for (int i = 0; i < 8; ++i) { bh[i] = (_Float16)(0.01f * (lane + i) - 0.3f); bl[i] = (_Float16)(0.0002f * (lane - i)); }
    f32x16 acc[2] = {{0}, {0}};
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = 0.5f + lane * 0.01f + i;
    const unsigned lane16 = lane * 16;
    unsigned parity = 0, pf = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(gsrc), 0, 0x7fffffff, 0x00027000);
    long long t0 = clock64();
    for (int c = 0; c < chunks; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned base = parity * CHUNK + lane16 + (WPS == 2 ? half * (CHUNK / 2) : 0);       // this wave's k-steps of the chunk
        unsigned so = pf;
        unsigned dst = (parity ^ 1u) * CHUNK + wave * (CHUNK / NW);
        asm volatile("" : "+v"(base), "+s"(so), "+s"(dst));
        if constexpr (WPS == 2) {
            // partner's partial sums of the tile this wave finishes: read, add, feed the epilogue
            const unsigned ex = EXCH + ((wave ^ 1) * 2 + parity) * 4096 + lane16;
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q) {                     // a 32 KiB chunk is half the K of two tiles: half a tile's partial sums per wave and chunk
                const f32x4 v = *reinterpret_cast<const f32x4 *>(smem + ex + q * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[0][4 * q + i] += v[i]; }
            }
            s = acc[0][0] * 1e-30f;
            f[0] += s;
        }
        half8 ah[2][2], al[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ah[0][t] = *reinterpret_cast<const half8 *>(smem + base + t * 2048);
            al[0][t] = *reinterpret_cast<const half8 *>(smem + base + t * 2048 + 1024);
        }
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int cur = k & 1, nxt = cur ^ 1;
            if (k + 1 < KS) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    ah[nxt][t] = *reinterpret_cast<const half8 *>(smem + base + ((k + 1) * 2 + t) * 2048);
                    al[nxt][t] = *reinterpret_cast<const half8 *>(smem + base + ((k + 1) * 2 + t) * 2048 + 1024);
                }
            }
            if (k < PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(smem + dst + k * 1024), 16, (int)lane16, (int)(so + wave * (CHUNK / NW) + k * 1024), 0, 0);
            kstep<VPG>(acc, ah[cur], al[cur], bh, bl, f);
        }
        if constexpr (WPS == 2) {
            // partial sums of the tile the partner finishes
            const unsigned ex = EXCH + (wave * 2 + (parity ^ 1u)) * 4096 + lane16;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 v = {acc[1][4 * q], acc[1][4 * q + 1], acc[1][4 * q + 2], acc[1][4 * q + 3]};
                *reinterpret_cast<f32x4 *>(smem + ex + q * 1024) = v;
            }
        }
        pf += CHUNK; if (pf >= STREAM) pf = 0;
        parity ^= 1u;
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int WPS>
void run(const char *name, const char *gsrc)
{
    float *out; long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 2048 * 8);
    (void)hipMemset(cyc, 0, 2048 * 8);
    const int chunks = WPS == 1 ? 4000 : 8000;                  // the same MFMAs per SIMD: 4000 x 96 = 8000 x 2 x 24
    const int lds = WPS == 1 ? 2 * 65536 : 2 * 32768 + 65536;
    hipFuncSetAttribute((const void *)bench<WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((bench<WPS>), dim3(256), dim3(256 * WPS), lds, 0, out, cyc, 200, gsrc);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((bench<WPS>), dim3(256), dim3(256 * WPS), lds, 0, out, cyc, chunks, gsrc);
    hipEventRecord(e1, 0);
    hipError_t err = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2048);
    hipMemcpy(h.data(), cyc, 2048 * 8, hipMemcpyDeviceToHost);
    double avg = 0; int n = 0; for (auto v : h) if (v) { avg += v; ++n; }
    avg /= n;
    const double mfma_per_simd = 4000.0 * 96.0;
    printf("%-58s: %9.0f cycles  %6.2f cycles per MFMA of the SIMD  kernel %7.3f ms  %5.2f ns per MFMA  clock %4.0f MHz  %s\n", name, avg, avg / mfma_per_simd, ms,
           ms * 1e6 / mfma_per_simd, avg / (ms * 1e3), err == hipSuccess ? "" : hipGetErrorString(err));
    hipFree(out); hipFree(cyc);
}

int main()
{
    char *gsrc; hipMalloc(&gsrc, STREAM + 65536);
    std::vector<unsigned short> h((STREAM + 65536) / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x2c00 + (i * 2654435761u >> 20) % 0x0fff);
    hipMemcpy(gsrc, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        run<1>("1 wave / SIMD, 64 KiB chunks (shipped form)", gsrc);
        run<2>("2 waves / SIMD, K-split pairs, 32 KiB chunks + exchange", gsrc);
    }
    return 0;
}
