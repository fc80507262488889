// Canonical normal fusion on the device (reference normal_fusion/normal_fusion.py) -- SURVEY.md section 8(f) item 2.
//   canonicalize_normals : the per-vertex part of canonicalize_normal_map (:27-62): project every posed vertex into the
//                          image, test it against the rendered position map (visibility), fetch the observed normal and
//                          carry it back through the camera and the vertex's skinning matrix to the canonical pose.
// One thread per vertex; 12 B + 64 B in, 12 B out per vertex, two dependent texel gathers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <utility>

#include "avcap_internal.h"

namespace avc {
namespace {

struct CanonArgs {
    float R[9], t[3];        // mv[:3,:3], mv[:3,3]
    float Ri[9];             // inv(mv)[:3,:3]
    float fx, fy, cx, cy;
    int H, W;
};

// F.grid_sample(..., 'nearest', 'border', align_corners=True): unnormalise, clamp, round half to even
__device__ __forceinline__ int nearest_px(float g, int n)
{
    float pix = (g + 1.0f) * 0.5f * (float)(n - 1);
    pix = fminf(fmaxf(pix, 0.0f), (float)(n - 1));
    return (int)rintf(pix);
}

__global__ __launch_bounds__(256) void canonicalize_kernel(const float *__restrict__ live_v, const float *__restrict__ vert_mats, int64_t nv,
                                                           const float *__restrict__ pos_map, const float *__restrict__ nrm_map, CanonArgs a,
                                                           float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const float x = live_v[3 * i], y = live_v[3 * i + 1], z = live_v[3 * i + 2];
    const float cx_ = ((a.R[0] * x + a.R[1] * y) + a.R[2] * z) + a.t[0];
    const float cy_ = ((a.R[3] * x + a.R[4] * y) + a.R[5] * z) + a.t[1];
    const float cz_ = ((a.R[6] * x + a.R[7] * y) + a.R[8] * z) + a.t[2];
    const float gx = 2.0f * ((cx_ / cz_ * a.fx + a.cx) / (float)a.W) - 1.0f;
    const float gy = 2.0f * ((cy_ / cz_ * a.fy + a.cy) / (float)a.H) - 1.0f;
    float n[3] = {0.f, 0.f, 0.f};
    bool valid = isfinite(gx) && isfinite(gy);
    if (valid) {
        const int ix = nearest_px(gx, a.W), iy = nearest_px(gy, a.H);
        const float4 p = reinterpret_cast<const float4 *>(pos_map)[(int64_t)iy * a.W + ix];
        const float dx = x - p.x, dy = y - p.y, dz = z - p.z;
        valid = sqrtf((dx * dx + dy * dy) + dz * dz) < 0.05f;
        const float *q = nrm_map + 3 * ((int64_t)iy * a.W + ix);
        const float ox = q[0], oy = -q[1], oz = -q[2];
        valid = valid && sqrtf((ox * ox + oy * oy) + oz * oz) > 1e-6f;
        // camera -> world
        const float wx = (a.Ri[0] * ox + a.Ri[1] * oy) + a.Ri[2] * oz;
        const float wy = (a.Ri[3] * ox + a.Ri[4] * oy) + a.Ri[5] * oz;
        const float wz = (a.Ri[6] * ox + a.Ri[7] * oy) + a.Ri[8] * oz;
        // posed -> canonical: inverse of the upper-left 3x3 of the vertex's cano2live matrix (adjugate / determinant)
        const float *M = vert_mats + 16 * i;
        const float m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[4], m11 = M[5], m12 = M[6], m20 = M[8], m21 = M[9], m22 = M[10];
        const float c00 = m11 * m22 - m12 * m21, c01 = m02 * m21 - m01 * m22, c02 = m01 * m12 - m02 * m11;
        const float c10 = m12 * m20 - m10 * m22, c11 = m00 * m22 - m02 * m20, c12 = m02 * m10 - m00 * m12;
        const float c20 = m10 * m21 - m11 * m20, c21 = m01 * m20 - m00 * m21, c22 = m00 * m11 - m01 * m10;
        const float det = (m00 * c00 + m01 * c10) + m02 * c20;
        valid = valid && fabsf(det) > 1e-12f;
        const float id = 1.0f / det;
        n[0] = ((c00 * wx + c01 * wy) + c02 * wz) * id;
        n[1] = ((c10 * wx + c11 * wy) + c12 * wz) * id;
        n[2] = ((c20 * wx + c21 * wy) + c22 * wz) * id;
    }
    out[3 * i] = valid ? n[0] : 0.f; out[3 * i + 1] = valid ? n[1] : 0.f; out[3 * i + 2] = valid ? n[2] : 0.f;
}

// 3x3 block of the inverse of a general 4x4 (double, Gauss-Jordan with partial pivoting); false if singular
static bool inv4_upper3(const float m[16], float out9[9])
{
    double a[4][8];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { a[r][c] = m[4 * r + c]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (std::abs(a[r][c]) > std::abs(a[piv][c])) piv = r;
        if (std::abs(a[piv][c]) < 1e-300) return false;
        if (piv != c) for (int k = 0; k < 8; ++k) std::swap(a[piv][k], a[c][k]);
        const double d = 1.0 / a[c][c];
        for (int k = 0; k < 8; ++k) a[c][k] *= d;
        for (int r = 0; r < 4; ++r) if (r != c) { const double f = a[r][c]; if (f != 0.0) for (int k = 0; k < 8; ++k) a[r][k] -= f * a[c][k]; }
    }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out9[3 * r + c] = (float)a[r][4 + c];
    return true;
}

}  // namespace

int canonicalize_normals(const float *live_v, const float *vert_mats, int64_t nv, const float *pos_map, const float *nrm_map, int H, int W,
                         const float mv[16], float fx, float fy, float cx, float cy, float *out, hipStream_t s)
{
    if (nv == 0) return AVC_OK;
    CanonArgs a;
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) a.R[3 * r + c] = mv[4 * r + c]; a.t[r] = mv[4 * r + 3]; }
    AVC_REQUIRE(inv4_upper3(mv, a.Ri), AVC_ERR_ARG, "avc_canonicalize_normals: the model-view matrix is singular");
    a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy; a.H = H; a.W = W;
    hipLaunchKernelGGL(canonicalize_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, s, live_v, vert_mats, nv, pos_map, nrm_map, a, out);
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

}  // namespace avc
