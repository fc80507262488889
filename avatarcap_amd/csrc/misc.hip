// Small device helpers of libavcap_hip.so: NCHW -> channel-last relayout of the per-frame feature
// maps, and the valid/invalid scatter of main.py:362-363.  HBM-bound, trivially small.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>

#include "avcap_internal.h"

namespace avc {

// ------------------------------------------------------------------------------------------
// small helpers: NCHW -> HWC relayout, volume scatter
// ------------------------------------------------------------------------------------------
__global__ void nchw_to_hwc_kernel(const float *__restrict__ src, float *__restrict__ dst, int C, int HW)
{
    __shared__ float tile[64][65];
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 256 threads: 64 x 4
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, pp = p0 + tx;
        tile[r][tx] = (c < C && pp < HW) ? src[(size_t)c * HW + pp] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int pp = p0 + r, c = c0 + tx;
        if (c < C && pp < HW) dst[(size_t)pp * C + c] = tile[tx][r];
    }
}

int launch_nchw_to_hwc(const float *src, float *dst, int C, int H, int W, hipStream_t s)
{
    const int HW = H * W;
    dim3 grid((HW + 63) / 64, (C + 63) / 64);
    hipLaunchKernelGGL(nchw_to_hwc_kernel, grid, dim3(256), 0, s, src, dst, C, HW);
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

// occ_volume[valid] = values (compacted order); occ_volume[~valid] = fill  (main.py:362-363).
// Ranks come from a block-level prefix over the flags; three tiny kernels.
__global__ void scatter_count_kernel(const uint8_t *__restrict__ valid, int64_t N, unsigned *__restrict__ block_counts)
{
    __shared__ unsigned wsum[4];
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    unsigned c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) c += (i + k < N && valid[i + k]) ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void scan_blocks_kernel(unsigned *__restrict__ counts, int nblocks)
{
    // single workgroup exclusive scan (nblocks up to a few hundred thousand): serial over chunks of 1024
    __shared__ unsigned buf[1024];
    __shared__ unsigned carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblocks; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        const unsigned v = i < nblocks ? counts[i] : 0u;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            unsigned t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0u;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblocks) counts[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
}

__global__ void scatter_write_kernel(const uint8_t *__restrict__ valid, int64_t N, const unsigned *__restrict__ block_off,
                                     const float *__restrict__ values, const float *__restrict__ fill, float *__restrict__ vol)
{
    __shared__ unsigned wsum[4];
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    unsigned f[4], c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { f[k] = (i + k < N && valid[i + k]) ? 1u : 0u; c += f[k]; }
    // exclusive prefix of c within the block
    unsigned incl = c;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    unsigned wbase = 0;
    for (int k = 0; k < w; ++k) wbase += wsum[k];
    unsigned rank = block_off[blockIdx.x] + wbase + incl - c;      // # valid before element i
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i + k < N) {
            if (f[k]) { vol[i + k] = values[rank]; ++rank; }
            else vol[i + k] = fill[(i + k) - rank];
        }
    }
}

int launch_scatter(avc_ctx *ctx, const uint8_t *valid, int64_t N, const float *values, const float *fill, float *vol, hipStream_t s)
{
    const int nblocks = (int)((N + 1023) / 1024);
    if (sizeof(unsigned) * (size_t)nblocks > ctx->scatter_scratch_bytes) {        // per-context (per-device) block counts
        if (ctx->scatter_scratch) AVC_HIP(hipFree(ctx->scatter_scratch));
        ctx->scatter_scratch = nullptr; ctx->scatter_scratch_bytes = 0;
        AVC_HIP(hipMalloc(&ctx->scatter_scratch, sizeof(unsigned) * (size_t)nblocks));
        ctx->scatter_scratch_bytes = sizeof(unsigned) * (size_t)nblocks;
    }
    unsigned *g_scatter_scratch = (unsigned *)ctx->scatter_scratch;
    hipLaunchKernelGGL(scatter_count_kernel, dim3(nblocks), dim3(256), 0, s, valid, N, g_scatter_scratch);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, s, g_scatter_scratch, nblocks);
    hipLaunchKernelGGL(scatter_write_kernel, dim3(nblocks), dim3(256), 0, s, valid, N, g_scatter_scratch, values, fill, vol);
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}


// ---- GroupNorm (+ ReLU) for the HGFilter producer (reference network/HGFilters.py:46-49,64-66,141,165,178,204) -------
// Every GroupNorm of the image encoder is followed by a ReLU; eager PyTorch spends more time on their 55 tiny
// statistics launches (32 workgroups each) than on the convolutions between them.  Two launches per layer here:
// slice partial sums (double accumulation, fixed summation order -> deterministic), then a*x + b (+ max 0).
// x is (N, C, HW) contiguous; a group is C/G adjacent channels = one contiguous run of (C/G)*HW floats.
constexpr int GN_MAX_SPLIT = 64;

__global__ __launch_bounds__(256) void gn_stats_kernel(const float *__restrict__ x, int64_t L, int S, double *__restrict__ part)
{
    const int g = blockIdx.x, sl = blockIdx.y;
    const int64_t per = ((L + S - 1) / S + 3) & ~(int64_t)3, lo = sl * per, hi = min(L, lo + per);
    const float *p = x + (int64_t)g * L;
    double s1 = 0.0, s2 = 0.0;
    const bool vec = (L & 3) == 0 && ((uintptr_t)p & 15) == 0;
    if (vec) {
        for (int64_t i = lo + 4 * threadIdx.x; i < hi; i += 4 * 256) {
            const float4 v = *reinterpret_cast<const float4 *>(p + i);
            s1 += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
            s2 += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
        }
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) { const double v = p[i]; s1 += v; s2 += v * v; }
    }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    __shared__ double red[2][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((int64_t)g * S + sl) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        part[((int64_t)g * S + sl) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float *__restrict__ x, float *__restrict__ y, int C, int64_t HW, int cpg, int S,
                                                       const double *__restrict__ part, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, float eps, int relu, int P)
{
    const int plane = blockIdx.x;                       // n * C + c
    const int c = plane % C, g = plane / cpg;           // (n * C + c) / cpg == n * G + c / cpg
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < S; ++k) { s1 += part[((int64_t)g * S + k) * 2]; s2 += part[((int64_t)g * S + k) * 2 + 1]; }
    const double L = (double)cpg * (double)HW, mean = s1 / L, var = fmax(s2 / L - mean * mean, 0.0);
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float a = rstd * (gamma ? gamma[c] : 1.f), b = (beta ? beta[c] : 0.f) - (float)mean * a;
    const float *px = x + (int64_t)plane * HW;
    float *py = y + (int64_t)plane * HW;
    const int64_t per = ((HW + P - 1) / P + 3) & ~(int64_t)3, lo = blockIdx.y * per, hi = min(HW, lo + per);
    if ((HW & 3) == 0 && (((uintptr_t)px | (uintptr_t)py) & 15) == 0) {
        for (int64_t i = lo + 4 * threadIdx.x; i < hi; i += 4 * 256) {
            float4 v = *reinterpret_cast<const float4 *>(px + i);
            v.x = a * v.x + b; v.y = a * v.y + b; v.z = a * v.z + b; v.w = a * v.w + b;
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4 *>(py + i) = v;
        }
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) { const float v = a * px[i] + b; py[i] = relu ? fmaxf(v, 0.f) : v; }
    }
}

int launch_group_norm(avc_ctx *ctx, const float *x, int N, int C, int64_t HW, int G, const float *gamma, const float *beta, float eps,
                      int relu, float *y, hipStream_t s)
{
    const int cpg = C / G;
    const int64_t L = (int64_t)cpg * HW;
    const int S = (int)std::max<int64_t>(1, std::min<int64_t>(GN_MAX_SPLIT, std::min<int64_t>(2048 / ((int64_t)N * G) + 1, L / 4096 + 1)));
    const size_t bytes = sizeof(double) * 2 * (size_t)N * G * S;
    if (ctx->gn_scratch_bytes < bytes) {
        // never freed while the context lives: a caller may have captured launches that carry the old pointer into a hipGraph;
        // outgrown blocks are parked and released by avc_ctx_destroy
        if (ctx->gn_scratch) ctx->retired_scratch.push_back(ctx->gn_scratch);
        ctx->gn_scratch = nullptr; ctx->gn_scratch_bytes = 0;
        AVC_HIP(hipMalloc(&ctx->gn_scratch, bytes));
        ctx->gn_scratch_bytes = bytes;
    }
    double *part = static_cast<double *>(ctx->gn_scratch);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(N * G, S), dim3(256), 0, s, x, L, S, part);
    const int P = (int)std::max<int64_t>(1, std::min<int64_t>(64, std::min<int64_t>(4096 / ((int64_t)N * C) + 1, HW / 2048 + 1)));
    hipLaunchKernelGGL(gn_apply_kernel, dim3(N * C, P), dim3(256), 0, s, x, y, C, HW, cpg, S, part, gamma, beta, eps, relu, P);
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

}  // namespace avc
