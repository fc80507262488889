// Orthographic front/back attribute rasteriser on the device: what the reference gets from OpenGL in
// utils/visualize_util.py:11-52 (render_cano_mesh: 'vertex_attribute' shader, orthographic projection,
// depth test, back-face culling, flipped read-back) -- SURVEY.md section 8(f) item 1.  It turns the
// marching-cubes mesh + its normals into the 512x512 front / back canonical normal maps that
// ReconNetwork consumes, without leaving the GPU (the reference needs a GL context and a host round
// trip).  Conventions and the unpinned-parity note: oracle/raster_oracle.c, which this file matches
// bit for bit (built with -ffp-contract=off).
//   pass 1: one thread per triangle; fixed-point (8 sub-pixel bits) edge functions, top-left rule,
//           64-bit atomicMax of (ordered depth, ~triangle id) per covered pixel, both views
//   pass 2: one thread per pixel; recompute the winner's barycentrics, interpolate the attribute
// Sub-pixel triangles dominate (1.2 M faces on 262 k pixels), so the per-triangle bounding-box loop
// is 1-4 pixels; HBM-bound on the 36 B/face + 24 B/vertex read.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "avcap_internal.h"

namespace avc {
namespace {

constexpr int SUB = 256;

__device__ __forceinline__ long long edge_fn(long long ax, long long ay, long long bx, long long by, long long px, long long py)
{
    return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}
__device__ __forceinline__ bool top_left(long long dx, long long dy) { return (dy < 0) || (dy == 0 && dx < 0); }
__device__ __forceinline__ unsigned ordered(float z)
{
    const unsigned u = __float_as_uint(z);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct Tri { long long fx[3], fy[3]; float z[3]; };

__device__ __forceinline__ void load_tri(const float *__restrict__ verts, const int32_t *__restrict__ faces, long long t,
                                         float cx, float cy, float cz, float half, Tri &T)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float *p = verts + 3 * (long long)faces[3 * t + k];
        const float x = p[0] - cx, y = p[1] - cy;
        T.fx[k] = (long long)floorf((x + 1.0f) * half * (float)SUB + 0.5f);
        T.fy[k] = (long long)floorf((1.0f - y) * half * (float)SUB + 0.5f);
        T.z[k] = p[2] - cz;
    }
}

// A sliver can win a pixel through the 1/256-pixel snap of its corners while its UNSNAPPED area is ~0 or of the other sign: the barycentric
// weights e_k / (e0 + e1 + e2) then leave [0, 1] by far, or are inf / NaN, and would extrapolate the attribute (normals that feed the fusion).
// Such weights -- any outside [-0.5, 1.5], or not finite -- are clamped to [0, 1] and renormalised (equal thirds if nothing is left); every
// other triangle, including the ordinary near-edge pixel whose weights overshoot by a sub-pixel's worth, is left exactly as it was.
__device__ __forceinline__ void bary_guard(double *e0, double *e1, double *e2)
{
    const double ar = (*e0 + *e1) + *e2;
    const double l0 = *e0 / ar, l1 = *e1 / ar, l2 = *e2 / ar;
    if (l0 >= -0.5 && l0 <= 1.5 && l1 >= -0.5 && l1 <= 1.5 && l2 >= -0.5 && l2 <= 1.5) return;
    double c0 = l0 > 0.0 ? (l0 < 1.0 ? l0 : 1.0) : 0.0, c1 = l1 > 0.0 ? (l1 < 1.0 ? l1 : 1.0) : 0.0, c2 = l2 > 0.0 ? (l2 < 1.0 ? l2 : 1.0) : 0.0;
    double sum = (c0 + c1) + c2;
    if (!(sum > 0.0)) { c0 = c1 = c2 = 1.0; sum = 3.0; }
    *e0 = c0 / sum; *e1 = c1 / sum; *e2 = c2 / sum;
}

__global__ __launch_bounds__(256) void raster_depth_kernel(const float *__restrict__ verts, const int32_t *__restrict__ faces, long long nf,
                                                           float cx, float cy, float cz, int size, unsigned long long *__restrict__ keys)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nf) return;
    const float half = 0.5f * (float)size;
    Tri T;
    load_tri(verts, faces, t, cx, cy, cz, half, T);
    long long area = edge_fn(T.fx[0], T.fy[0], T.fx[1], T.fy[1], T.fx[2], T.fy[2]);
    if (area == 0) return;
    const int view = area < 0 ? 0 : 1;                    // front keeps area < 0, back keeps area > 0 (oracle)
    int a = 0, b = 1, c = 2;
    if (area < 0) { b = 2; c = 1; area = -area; }
    long long minx = min(T.fx[0], min(T.fx[1], T.fx[2])), maxx = max(T.fx[0], max(T.fx[1], T.fx[2]));
    long long miny = min(T.fy[0], min(T.fy[1], T.fy[2])), maxy = max(T.fy[0], max(T.fy[1], T.fy[2]));
    long long x0 = (minx - SUB / 2 + SUB - 1) / SUB, x1 = (maxx - SUB / 2) / SUB;
    long long y0 = (miny - SUB / 2 + SUB - 1) / SUB, y1 = (maxy - SUB / 2) / SUB;
    if (minx - SUB / 2 < 0) x0 = 0;
    if (miny - SUB / 2 < 0) y0 = 0;
    x0 = max(x0, 0ll); y0 = max(y0, 0ll); x1 = min(x1, (long long)size - 1); y1 = min(y1, (long long)size - 1);
    const bool tl0 = top_left(T.fx[c] - T.fx[b], T.fy[c] - T.fy[b]);
    const bool tl1 = top_left(T.fx[a] - T.fx[c], T.fy[a] - T.fy[c]);
    const bool tl2 = top_left(T.fx[b] - T.fx[a], T.fy[b] - T.fy[a]);
    const float za = view == 0 ? T.z[a] : -T.z[a], zb = view == 0 ? T.z[b] : -T.z[b], zc = view == 0 ? T.z[c] : -T.z[c];
    const float inv = 1.0f / (float)area;
    unsigned long long *kv = keys + (size_t)view * size * size;
    for (long long py = y0; py <= y1; ++py)
        for (long long px = x0; px <= x1; ++px) {
            const long long sx = px * SUB + SUB / 2, sy = py * SUB + SUB / 2;
            const long long w0 = edge_fn(T.fx[b], T.fy[b], T.fx[c], T.fy[c], sx, sy);
            const long long w1 = edge_fn(T.fx[c], T.fy[c], T.fx[a], T.fy[a], sx, sy);
            const long long w2 = edge_fn(T.fx[a], T.fy[a], T.fx[b], T.fy[b], sx, sy);
            if (w0 < 0 || w1 < 0 || w2 < 0) continue;
            if ((w0 == 0 && !tl0) || (w1 == 0 && !tl1) || (w2 == 0 && !tl2)) continue;
            const float l0 = (float)w0 * inv, l1 = (float)w1 * inv, l2 = (float)w2 * inv;
            const float z = (l0 * za + l1 * zb) + l2 * zc;
            const unsigned long long key = ((unsigned long long)ordered(z) << 32) | (unsigned long long)(0xffffffffu - (unsigned)t);
            atomicMax(kv + py * size + px, key);
        }
}

__global__ __launch_bounds__(256) void raster_resolve_kernel(const float *__restrict__ verts, const float *__restrict__ attrs,
                                                             const int32_t *__restrict__ faces, float cx, float cy, float cz, int size,
                                                             const unsigned long long *__restrict__ keys, float *__restrict__ front, float *__restrict__ back)
{
    const long long pi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long npix = (long long)size * size;
    if (pi >= 2 * npix) return;
    const int view = pi >= npix;
    const long long p = view ? pi - npix : pi;
    float *out = (view ? back : front) + 3 * p;
    const unsigned long long key = keys[pi];
    if (key == 0ull) { out[0] = 0.f; out[1] = 0.f; out[2] = 0.f; return; }
    const long long t = (long long)(0xffffffffu - (unsigned)(key & 0xffffffffull));
    const float half = 0.5f * (float)size;
    const long long px = p % size, py = p / size;
    // Attribute interpolation from the UNSNAPPED window positions, in double, exactly as oracle/raster_oracle.c does it (coverage and depth
    // use the 1/256-pixel fixed-point positions; OpenGL -- Mesa llvmpipe, tests/golden/gl_golden.npz -- derives the attribute planes from the
    // float positions, and on a grazing triangle the snap would move an attribute by up to 1e-3)
    double wx[3], wy[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float *q = verts + 3 * (long long)faces[3 * t + k];
        wx[k] = (double)((q[0] - cx + 1.0f) * half);
        wy[k] = (double)((1.0f - (q[1] - cy)) * half);
    }
    const double qx = (double)px + 0.5, qy = (double)py + 0.5;
    const double e0 = (wx[2] - wx[1]) * (qy - wy[1]) - (wy[2] - wy[1]) * (qx - wx[1]);
    const double e1 = (wx[0] - wx[2]) * (qy - wy[2]) - (wy[0] - wy[2]) * (qx - wx[2]);
    const double e2 = (wx[1] - wx[0]) * (qy - wy[0]) - (wy[1] - wy[0]) * (qx - wx[0]);
    double g0 = e0, g1 = e1, g2 = e2;
    bary_guard(&g0, &g1, &g2);                          // slivers only (oracle/raster_oracle.c)
    const double ar = (g0 + g1) + g2;
    const double l0 = g0 / ar, l1 = g1 / ar, l2 = g2 / ar;
    const float *A = attrs + 3 * (long long)faces[3 * t + 0], *B = attrs + 3 * (long long)faces[3 * t + 1], *C = attrs + 3 * (long long)faces[3 * t + 2];
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = (float)((l0 * (double)A[k] + l1 * (double)B[k]) + l2 * (double)C[k]);
}


// ---- general model-view-projection view (oracle/raster_oracle.c: raster_mvp_oracle) ----------------------------
struct Mvp { float m[16]; };
struct TriP { long long fx[3], fy[3]; float zn[3], iw[3]; };

__device__ __forceinline__ bool load_tri_mvp(const float *__restrict__ verts, const int32_t *__restrict__ faces, long long t, const Mvp &M,
                                             int W, int H, TriP &T)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float *p = verts + 3 * (long long)faces[3 * t + k];
        float c[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = ((M.m[4 * r] * p[0] + M.m[4 * r + 1] * p[1]) + M.m[4 * r + 2] * p[2]) + M.m[4 * r + 3];
        if (!(c[3] > 0.0f)) return false;
        T.iw[k] = 1.0f / c[3];
        const float nx = c[0] * T.iw[k], ny = c[1] * T.iw[k];
        T.zn[k] = c[2] * T.iw[k];
        const float px = (nx + 1.0f) * (0.5f * (float)W) * (float)SUB + 0.5f, py = (1.0f - ny) * (0.5f * (float)H) * (float)SUB + 0.5f;
        if (!(fabsf(px) < 1.0e12f) || !(fabsf(py) < 1.0e12f)) return false;
        T.fx[k] = (long long)floorf(px); T.fy[k] = (long long)floorf(py);
    }
    return true;
}

__global__ __launch_bounds__(256) void raster_mvp_depth_kernel(const float *__restrict__ verts, const int32_t *__restrict__ faces, long long nf,
                                                               Mvp M, int W, int H, unsigned long long *__restrict__ keys)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nf) return;
    TriP T;
    if (!load_tri_mvp(verts, faces, t, M, W, H, T)) return;
    long long area = edge_fn(T.fx[0], T.fy[0], T.fx[1], T.fy[1], T.fx[2], T.fy[2]);
    if (area >= 0) return;                                   // back-facing or degenerate
    const int a = 0, b = 2, c = 1; area = -area;
    const long long minx = min(T.fx[0], min(T.fx[1], T.fx[2])), maxx = max(T.fx[0], max(T.fx[1], T.fx[2]));
    const long long miny = min(T.fy[0], min(T.fy[1], T.fy[2])), maxy = max(T.fy[0], max(T.fy[1], T.fy[2]));
    if (maxx < 0 || maxy < 0 || minx > (long long)W * SUB || miny > (long long)H * SUB) return;
    long long x0 = (minx - SUB / 2 + SUB - 1) / SUB, x1 = (maxx - SUB / 2) / SUB;
    long long y0 = (miny - SUB / 2 + SUB - 1) / SUB, y1 = (maxy - SUB / 2) / SUB;
    if (minx - SUB / 2 < 0) x0 = 0;
    if (miny - SUB / 2 < 0) y0 = 0;
    x0 = max(x0, 0ll); y0 = max(y0, 0ll); x1 = min(x1, (long long)W - 1); y1 = min(y1, (long long)H - 1);
    const bool tl0 = top_left(T.fx[c] - T.fx[b], T.fy[c] - T.fy[b]);
    const bool tl1 = top_left(T.fx[a] - T.fx[c], T.fy[a] - T.fy[c]);
    const bool tl2 = top_left(T.fx[b] - T.fx[a], T.fy[b] - T.fy[a]);
    const float inv = 1.0f / (float)area;
    for (long long py = y0; py <= y1; ++py)
        for (long long px = x0; px <= x1; ++px) {
            const long long sx = px * SUB + SUB / 2, sy = py * SUB + SUB / 2;
            const long long w0 = edge_fn(T.fx[b], T.fy[b], T.fx[c], T.fy[c], sx, sy);
            const long long w1 = edge_fn(T.fx[c], T.fy[c], T.fx[a], T.fy[a], sx, sy);
            const long long w2 = edge_fn(T.fx[a], T.fy[a], T.fx[b], T.fy[b], sx, sy);
            if (w0 < 0 || w1 < 0 || w2 < 0) continue;
            if ((w0 == 0 && !tl0) || (w1 == 0 && !tl1) || (w2 == 0 && !tl2)) continue;
            const float l0 = (float)w0 * inv, l1 = (float)w1 * inv, l2 = (float)w2 * inv;
            const float z = (l0 * T.zn[a] + l1 * T.zn[b]) + l2 * T.zn[c];
            if (!(z >= -1.0f && z <= 1.0f)) continue;
            // GL_LESS: the smallest ndc.z wins, ties go to the lower triangle id
            const unsigned long long key = ((unsigned long long)ordered(-z) << 32) | (unsigned long long)(0xffffffffu - (unsigned)t);
            atomicMax(keys + py * W + px, key);
        }
}

__global__ __launch_bounds__(256) void raster_mvp_resolve_kernel(const float *__restrict__ verts, const float *__restrict__ attrs,
                                                                 const int32_t *__restrict__ faces, Mvp M, int W, int H,
                                                                 const unsigned long long *__restrict__ keys, float *__restrict__ out)
{
    const long long pi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= (long long)W * H) return;
    float4 *o = reinterpret_cast<float4 *>(out) + pi;
    const unsigned long long key = keys[pi];
    if (key == 0ull) { *o = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const long long t = (long long)(0xffffffffu - (unsigned)(key & 0xffffffffull));
    const long long px = pi % W, py = pi / W;
    double wx[3], wy[3], iw[3];        // perspective-correct attribute from the unsnapped window positions, in double (oracle/raster_oracle.c)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float *q = verts + 3 * (long long)faces[3 * t + k];
        float c[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = ((M.m[4 * r] * q[0] + M.m[4 * r + 1] * q[1]) + M.m[4 * r + 2] * q[2]) + M.m[4 * r + 3];
        const float w1 = 1.0f / c[3];
        iw[k] = (double)w1;
        wx[k] = (double)((c[0] * w1 + 1.0f) * (0.5f * (float)W));
        wy[k] = (double)((1.0f - c[1] * w1) * (0.5f * (float)H));
    }
    const double qx = (double)px + 0.5, qy = (double)py + 0.5;
    const double e0 = (wx[2] - wx[1]) * (qy - wy[1]) - (wy[2] - wy[1]) * (qx - wx[1]);
    const double e1 = (wx[0] - wx[2]) * (qy - wy[2]) - (wy[0] - wy[2]) * (qx - wx[2]);
    const double e2 = (wx[1] - wx[0]) * (qy - wy[0]) - (wy[1] - wy[0]) * (qx - wx[0]);
    double g0 = e0, g1 = e1, g2 = e2;
    bary_guard(&g0, &g1, &g2);
    const double u0 = g0 * iw[0], u1 = g1 * iw[1], u2 = g2 * iw[2];
    const double den = (u0 + u1) + u2;
    const float *A = attrs + 3 * (long long)faces[3 * t + 0], *B = attrs + 3 * (long long)faces[3 * t + 1], *C = attrs + 3 * (long long)faces[3 * t + 2];
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) r[k] = (float)(((u0 * (double)A[k] + u1 * (double)B[k]) + u2 * (double)C[k]) / den);
    *o = make_float4(r[0], r[1], r[2], 1.0f);
}

}  // namespace

int render_cano_maps(avc_ctx *ctx, const float *verts, const float *attrs, const int32_t *faces, int64_t nf, const float center[3],
                     int size, float *front, float *back, hipStream_t s)
{
    const size_t kbytes = sizeof(unsigned long long) * 2 * (size_t)size * size;
    if (kbytes > ctx->raster_scratch_bytes) {
        if (ctx->raster_scratch) AVC_HIP(hipFree(ctx->raster_scratch));
        ctx->raster_scratch = nullptr; ctx->raster_scratch_bytes = 0;
        AVC_HIP(hipMalloc(&ctx->raster_scratch, kbytes));
        ctx->raster_scratch_bytes = kbytes;
    }
    unsigned long long *keys = (unsigned long long *)ctx->raster_scratch;
    AVC_HIP(hipMemsetAsync(keys, 0, kbytes, s));
    if (nf > 0)
        hipLaunchKernelGGL(raster_depth_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, verts, faces, (long long)nf,
                           center[0], center[1], center[2], size, keys);
    hipLaunchKernelGGL(raster_resolve_kernel, dim3((unsigned)((2ll * size * size + 255) / 256)), dim3(256), 0, s, verts, attrs, faces,
                       center[0], center[1], center[2], size, keys, front, back);
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

int render_mesh(avc_ctx *ctx, const float *verts, const float *attrs, const int32_t *faces, int64_t nf, const float mvp[16],
                int W, int H, float *out, hipStream_t s)
{
    const size_t kbytes = sizeof(unsigned long long) * (size_t)W * H;
    if (kbytes > ctx->raster_scratch_bytes) {
        if (ctx->raster_scratch) AVC_HIP(hipFree(ctx->raster_scratch));
        ctx->raster_scratch = nullptr; ctx->raster_scratch_bytes = 0;
        AVC_HIP(hipMalloc(&ctx->raster_scratch, kbytes));
        ctx->raster_scratch_bytes = kbytes;
    }
    unsigned long long *keys = (unsigned long long *)ctx->raster_scratch;
    AVC_HIP(hipMemsetAsync(keys, 0, kbytes, s));
    Mvp M;
    for (int i = 0; i < 16; ++i) M.m[i] = mvp[i];
    if (nf > 0) hipLaunchKernelGGL(raster_mvp_depth_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, verts, faces, (long long)nf, M, W, H, keys);
    hipLaunchKernelGGL(raster_mvp_resolve_kernel, dim3((unsigned)(((long long)W * H + 255) / 256)), dim3(256), 0, s, verts, attrs ? attrs : verts,
                       faces, M, W, H, keys, out);
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

}  // namespace avc
