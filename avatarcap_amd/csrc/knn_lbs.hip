// SMPL utilities on the device (reference utils/smpl_util.py):
//   knn            : pytorch3d.ops.knn_points as the reference uses it (smpl_util.py:33,
//                    avatarcap_dataset.py:114, arch_avatar.py:190,208) -- brute force, the 6890
//                    reference points are staged through LDS in tiles and every thread keeps its
//                    query's K best in registers (sorted insertion, ties -> lower index).
//   calculate_lbs  : KNN-4 + Gaussian weights + gather/blend of the 24-wide skin weights (:24-39), fused
//   skinning       : per-point blend of the 24 joint 4x4s and its application to points / normals (:58-81)
// All HBM/VALU-bound elementwise work; queries are read once, coalesced.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "avcap_internal.h"

namespace avc {
namespace {

constexpr int REF_TILE = 2048;   // reference points per LDS tile (32 KiB as float4)

template <int K>
__device__ __forceinline__ void knn_insert(float d, int id, float (&bd)[K], int (&bi)[K])
{
    if (d < bd[K - 1]) {
        bd[K - 1] = d; bi[K - 1] = id;
#pragma unroll
        for (int k = K - 1; k > 0; --k) {
            if (bd[k] < bd[k - 1]) {
                const float td = bd[k]; bd[k] = bd[k - 1]; bd[k - 1] = td;
                const int ti = bi[k]; bi[k] = bi[k - 1]; bi[k - 1] = ti;
            }
        }
    }
}

// Reference points are staged as float4 {x,y,z,-} so one ds_read_b128 (a broadcast: every lane reads
// the same address) feeds a candidate, four candidates per loop trip; the running K-th best distance
// gates the (rare) sorted insertion.
template <int K>
__device__ __forceinline__ void knn_scan(const float *__restrict__ ref, int nr, float qx, float qy, float qz,
                                         float (&bd)[K], int (&bi)[K], float4 *lds)
{
#pragma unroll
    for (int k = 0; k < K; ++k) { bd[k] = __builtin_inff(); bi[k] = 0x7fffffff; }
    for (int r0 = 0; r0 < nr; r0 += REF_TILE) {
        const int cnt = min(REF_TILE, nr - r0);
        __syncthreads();
        for (int i = threadIdx.x; i < REF_TILE; i += blockDim.x) {
            float4 v = make_float4(3.0e18f, 3.0e18f, 3.0e18f, 0.f);       // padding: never among the K nearest
            if (i < cnt) v = make_float4(ref[(size_t)(r0 + i) * 3], ref[(size_t)(r0 + i) * 3 + 1], ref[(size_t)(r0 + i) * 3 + 2], 0.f);
            lds[i] = v;
        }
        __syncthreads();
        const int cnt4 = (cnt + 3) & ~3;
        for (int i = 0; i < cnt4; i += 4) {
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 r = lds[i + u];
                // (dx*dx + dy*dy) + dz*dz with separately rounded operations (file built with -ffp-contract=off)
                const float dx = qx - r.x, dy = qy - r.y, dz = qz - r.z;
                d[u] = (dx * dx + dy * dy) + dz * dz;
            }
            const float m = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
            if (m < bd[K - 1]) {
#pragma unroll
                for (int u = 0; u < 4; ++u) knn_insert<K>(d[u], r0 + i + u, bd, bi);
            }
        }
    }
}

template <int K>
__global__ __launch_bounds__(256) void knn_kernel(const float *__restrict__ q, int64_t nq, const float *__restrict__ ref, int nr,
                                                  float *__restrict__ d2, int64_t *__restrict__ idx)
{
    __shared__ float4 lds[REF_TILE];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ii = i < nq ? i : nq - 1;
    float bd[K]; int bi[K];
    knn_scan<K>(ref, nr, q[3 * ii], q[3 * ii + 1], q[3 * ii + 2], bd, bi, lds);
    if (i < nq) {
#pragma unroll
        for (int k = 0; k < K; ++k) { if (d2) d2[i * K + k] = bd[k]; if (idx) idx[i * K + k] = bi[k]; }
    }
}

__global__ __launch_bounds__(256) void lbs_kernel(const float *__restrict__ pts, int64_t n, const float *__restrict__ cano_v,
                                                  const float *__restrict__ skin_w, int nv, float *__restrict__ lbs)
{
    __shared__ float4 lds[REF_TILE];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ii = i < n ? i : n - 1;
    float bd[4]; int bi[4];
    knn_scan<4>(cano_v, nv, pts[3 * ii], pts[3 * ii + 1], pts[3 * ii + 2], bd, bi, lds);
    if (i >= n) return;
    // weights = exp(-dists / (2 r^2)), r = 0.05; weights /= sum + 1e-16     (smpl_util.py:34-36)
    const float denom = (float)(2 * 0.05 * 0.05);
    float w[4], sum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { w[k] = expf(-bd[k] / denom); sum += w[k]; }
    sum += 1e-16f;
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] /= sum;
    // lbs = sum_k skin_w[idx_k] * w_k   (smpl_util.py:37-38)
    float acc[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) acc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 *row = reinterpret_cast<const float4 *>(skin_w + (size_t)bi[k] * 24);
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {
            const float4 v = row[jj];
            acc[4 * jj + 0] += v.x * w[k]; acc[4 * jj + 1] += v.y * w[k]; acc[4 * jj + 2] += v.z * w[k]; acc[4 * jj + 3] += v.w * w[k];
        }
    }
    float4 *out = reinterpret_cast<float4 *>(lbs + (size_t)i * 24);
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) out[jj] = make_float4(acc[4 * jj], acc[4 * jj + 1], acc[4 * jj + 2], acc[4 * jj + 3]);
}

__global__ __launch_bounds__(256) void skinning_kernel(const float *__restrict__ pts, const float *__restrict__ nrm, int64_t n,
                                                       const float *__restrict__ lbs, const float *__restrict__ jm,
                                                       float *__restrict__ po, float *__restrict__ no, float *__restrict__ mo)
{
    __shared__ float J[24 * 16];
    for (int i = threadIdx.x; i < 24 * 16; i += blockDim.x) J[i] = jm[i];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // cano2live_pt_mats = einsum('bnj,bjxy->bnxy', lbs, jnt_mats)   (smpl_util.py:67)
    float M[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) M[e] = 0.f;
    const float4 *lrow = reinterpret_cast<const float4 *>(lbs + (size_t)i * 24);
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) {
        const float4 l4 = lrow[jj];
        const float l[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) M[e] += l[u] * J[(4 * jj + u) * 16 + e];
    }
    if (mo) {
        float4 *o = reinterpret_cast<float4 *>(mo + (size_t)i * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = make_float4(M[4 * r], M[4 * r + 1], M[4 * r + 2], M[4 * r + 3]);
    }
    if (pts) {   // live = M[:3,:3] p + M[:3,3]   (smpl_util.py:69)
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r) po[3 * i + r] = M[4 * r] * x + M[4 * r + 1] * y + M[4 * r + 2] * z + M[4 * r + 3];
    }
    if (nrm) {   // live_normals = M[:3,:3] n, no renormalisation   (smpl_util.py:80)
        const float x = nrm[3 * i], y = nrm[3 * i + 1], z = nrm[3 * i + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r) no[3 * i + r] = M[4 * r] * x + M[4 * r + 1] * y + M[4 * r + 2] * z;
    }
}

}  // namespace

int knn(const float *q, int64_t nq, const float *ref, int32_t nr, int K, float *d2, int64_t *idx, hipStream_t s)
{
    if (nq == 0) return AVC_OK;
    const dim3 grid((unsigned)((nq + 255) / 256)), block(256);
    switch (K) {
#define CASE(k) case k: hipLaunchKernelGGL(knn_kernel<k>, grid, block, 0, s, q, nq, ref, nr, d2, idx); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
        default: set_error("avc_knn: unsupported K %d", K); return AVC_ERR_ARG;
    }
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

int calculate_lbs(const float *pts, int64_t n, const float *cano_v, const float *skin_w, int32_t nv, float *lbs, hipStream_t s)
{
    if (n == 0) return AVC_OK;
    hipLaunchKernelGGL(lbs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pts, n, cano_v, skin_w, nv, lbs);
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

int skinning(const float *pts, const float *nrm, int64_t n, const float *lbs, const float *jm, float *po, float *no, float *mo, hipStream_t s)
{
    if (n == 0) return AVC_OK;
    hipLaunchKernelGGL(skinning_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pts, nrm, n, lbs, jm, po, no, mo);
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

}  // namespace avc
