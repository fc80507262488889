// Small device helpers of libavcap_hip.so: NCHW -> channel-last relayout of the per-frame feature
// maps, and the valid/invalid scatter of main.py:362-363.  HBM-bound, trivially small.
#include <hip/hip_runtime.h>
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

static unsigned *g_scatter_scratch = nullptr;
static size_t g_scatter_cap = 0;

int launch_scatter(const uint8_t *valid, int64_t N, const float *values, const float *fill, float *vol, hipStream_t s)
{
    const int nblocks = (int)((N + 1023) / 1024);
    if ((size_t)nblocks > g_scatter_cap) {
        if (g_scatter_scratch) hipFree(g_scatter_scratch);
        AVC_HIP(hipMalloc((void **)&g_scatter_scratch, sizeof(unsigned) * nblocks));
        g_scatter_cap = nblocks;
    }
    hipLaunchKernelGGL(scatter_count_kernel, dim3(nblocks), dim3(256), 0, s, valid, N, g_scatter_scratch);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, s, g_scatter_scratch, nblocks);
    hipLaunchKernelGGL(scatter_write_kernel, dim3(nblocks), dim3(256), 0, s, valid, N, g_scatter_scratch, values, fill, vol);
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

}  // namespace avc
