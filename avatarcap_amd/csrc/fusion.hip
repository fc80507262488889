// Canonical normal fusion on the device (reference normal_fusion/normal_fusion.py) -- SURVEY.md section 8(f) item 2.
//   canonicalize_normals : the per-vertex part of canonicalize_normal_map (:27-62): project every posed vertex into the
//                          image, test it against the rendered position map (visibility), fetch the observed normal and
//                          carry it back through the camera and the vertex's skinning matrix to the canonical pose.
// One thread per vertex; 12 B + 64 B in, 12 B out per vertex, two dependent texel gathers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <cstdlib>
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


// ---- merge_normal_images (normal_fusion.py:89-155) ------------------------------------------------------------------
// The reference runs 100 autograd iterations of two Adam optimisers (a 64x64 axis-angle grid, then the normal map itself)
// from Python; here every iteration is one or two launches with the gradients written out by hand (oracle/
// normal_fusion_oracle.py holds the same formulas, pinned against torch.autograd), and the OpenCV pre-processing
// (3x3 erosion x3, L1 distance transform) runs on the device too.  Everything is fp32, deterministic (gathers, no atomics
// in the iteration), HBM/latency-bound and tiny next to the two network passes it sits between.
constexpr int GRID = 64;            // rot_aa_img is (64, 64, 3), normal_fusion.py:114
constexpr float DT_CAP = 8192.0f;   // OpenCV's chamfer saturates there when the image holds no zero pixel

__global__ void fus_masks_kernel(const float *__restrict__ src, const float *__restrict__ tar, int n, uint8_t *__restrict__ smask, uint8_t *__restrict__ tmask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *s = src + 3 * (size_t)i, *t = tar + 3 * (size_t)i;
    smask[i] = sqrtf((s[0] * s[0] + s[1] * s[1]) + s[2] * s[2]) > 0.f;
    tmask[i] = sqrtf((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]) > 0.f;
}

// cv.erode(3x3 rectangle, iterations = 3) == all set in the 7x7 window; pixels outside the image never erode
__global__ void fus_erode_kernel(const uint8_t *__restrict__ in, int H, int W, int r, uint8_t *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i % W;
    bool all = true;
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) all = all && in[yy * W + xx];
        }
    out[i] = all;
}

// cv.distanceTransform(DIST_L1, 3): the city-block distance is separable.  Along a row it is the distance to the nearest zero pixel on
// either side: one wave per row, 64 pixels at a time, the zero pixels of a segment as a ballot mask (highest zero bit at or below the
// lane, lowest at or above it), the nearest zero of the segments already passed as a carry.  Counts of pixels, exact in fp32; rows
// without a zero pixel keep the 1e18 the sequential recurrence d = mask ? d + 1 : 0 would leave there.
constexpr float DT_FAR = 1e18f;
constexpr int DT_SEGS = 16;
__global__ __launch_bounds__(64) void fus_dt_rows_kernel(const uint8_t *__restrict__ mask, int H, int W, float *__restrict__ g)
{
    const int y = blockIdx.x, lane = threadIdx.x;
    const uint8_t *row = mask + (size_t)y * W;
    float *out = g + (size_t)y * W;
    int carry = -1;                                       // x of the nearest zero pixel left of the segment
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        const unsigned long long z = __ballot(x < W && row[x] == 0);
        const unsigned long long below = z & (~0ull >> (63 - lane));
        const int last = below ? x0 + 63 - __clzll((long long)below) : carry;
        if (x < W) out[x] = last >= 0 ? (float)(x - last) : DT_FAR;
        if (z) carry = x0 + 63 - __clzll((long long)z);
    }
    carry = -1;                                           // ... and right of it
    for (int x0 = ((W - 1) / 64) * 64; x0 >= 0; x0 -= 64) {
        const int x = x0 + lane;
        const unsigned long long z = __ballot(x < W && row[x] == 0);
        const unsigned long long above = z >> lane;
        const int next = above ? x + __ffsll((unsigned long long)above) - 1 : carry;
        if (x < W && next >= 0) out[x] = fminf(out[x], (float)(next - x));
        if (z) carry = x0 + __ffsll((unsigned long long)z) - 1;
    }
}
// ... then down the columns: d(y) = min over yy of g(yy) + |y - yy| = min( y + min_{yy <= y} (g(yy) - yy), -y + min_{yy >= y} (g(yy) + yy) ), two
// running minima.  A workgroup takes 64 columns x DT_SEGS row segments: segment minima through LDS give every segment its carry-in, then one
// downward and one upward sweep over the segment's rows.  Integer-valued floats (or 1e18, which absorbs the +-y): the same value as the
// exhaustive minimum, bit for bit.
__global__ __launch_bounds__(64 * DT_SEGS) void fus_dt_cols_kernel(const float *__restrict__ g, int H, int W, float *__restrict__ out)
{
    __shared__ float lo[DT_SEGS][64], hi[DT_SEGS][64];
    const int lx = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lx;
    const int rows = (H + DT_SEGS - 1) / DT_SEGS, y0 = seg * rows, y1 = min(H, y0 + rows);
    float a = DT_FAR, b = DT_FAR;
    if (x < W)
        for (int y = y0; y < y1; ++y) { const float v = g[(size_t)y * W + x]; a = fminf(a, v - (float)y); b = fminf(b, v + (float)y); }
    lo[seg][lx] = a; hi[seg][lx] = b;
    __syncthreads();
    if (x >= W) return;
    float run = DT_FAR;
    for (int q = 0; q < seg; ++q) run = fminf(run, lo[q][lx]);
    for (int y = y0; y < y1; ++y) { run = fminf(run, g[(size_t)y * W + x] - (float)y); out[(size_t)y * W + x] = run + (float)y; }
    run = DT_FAR;
    for (int q = DT_SEGS - 1; q > seg; --q) run = fminf(run, hi[q][lx]);
    for (int y = y1 - 1; y >= y0; --y) {
        run = fminf(run, g[(size_t)y * W + x] + (float)y);
        out[(size_t)y * W + x] = fminf(fminf(out[(size_t)y * W + x], run - (float)y), DT_CAP);
    }
}

__global__ void fus_valid_kernel(const uint8_t *__restrict__ smask, const uint8_t *__restrict__ emask, int n, uint8_t *__restrict__ valid, int *__restrict__ count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool v = i < n && smask[i] && emask[i];
    if (i < n) valid[i] = v;
    const unsigned long long b = __ballot(v);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, __popcll(b));
}

struct Rot { float r, i, j, k, s2, sh, ch, kf, dk, th; };

// pytorch3d axis_angle_to_matrix = axis_angle_to_quaternion + quaternion_to_matrix, with what the backward needs
__device__ __forceinline__ void aa_forward(const float aa[3], float R[9], Rot &q)
{
    const float th = sqrtf((aa[0] * aa[0] + aa[1] * aa[1]) + aa[2] * aa[2]);
    const float half = th * 0.5f;
    const bool small = th < 1e-6f;
    q.th = th; q.sh = sinf(half); q.ch = cosf(half);
    q.kf = small ? 0.5f - th * th / 48.f : q.sh / th;
    q.dk = small ? -th / 24.f : (q.ch * 0.5f * th - q.sh) / (th * th);
    q.r = q.ch; q.i = aa[0] * q.kf; q.j = aa[1] * q.kf; q.k = aa[2] * q.kf;
    const float N = (q.r * q.r + q.i * q.i) + (q.j * q.j + q.k * q.k);
    q.s2 = 2.f / N;
    const float s2 = q.s2, r = q.r, i = q.i, j = q.j, k = q.k;
    R[0] = 1.f - s2 * (j * j + k * k); R[1] = s2 * (i * j - k * r); R[2] = s2 * (i * k + j * r);
    R[3] = s2 * (i * j + k * r); R[4] = 1.f - s2 * (i * i + k * k); R[5] = s2 * (j * k - i * r);
    R[6] = s2 * (i * k - j * r); R[7] = s2 * (j * k + i * r); R[8] = 1.f - s2 * (i * i + j * j);
}

// dL/daa from G = dL/dR (oracle: axis_angle_to_matrix_backward)
__device__ __forceinline__ void aa_backward(const float aa[3], const Rot &q, const float G[9], float g[3])
{
    const float r = q.r, i = q.i, j = q.j, k = q.k, s2 = q.s2;
    const float N = 2.f / s2;
    const float P[9] = {-(j * j + k * k), i * j - k * r, i * k + j * r, i * j + k * r, -(i * i + k * k), j * k - i * r, i * k - j * r, j * k + i * r, -(i * i + j * j)};
    float gp = 0.f;
#pragma unroll
    for (int e = 0; e < 9; ++e) gp += G[e] * P[e];
    const float dN = gp * (-s2 / N);
    const float dr = s2 * (-k * G[1] + j * G[2] + k * G[3] - i * G[5] - j * G[6] + i * G[7]) + dN * 2.f * r;
    const float di = s2 * (-2.f * i * (G[4] + G[8]) + j * (G[1] + G[3]) + k * (G[2] + G[6]) + r * (G[7] - G[5])) + dN * 2.f * i;
    const float dj = s2 * (-2.f * j * (G[0] + G[8]) + i * (G[1] + G[3]) + k * (G[5] + G[7]) + r * (G[2] - G[6])) + dN * 2.f * j;
    const float dkk = s2 * (-2.f * k * (G[0] + G[4]) + i * (G[2] + G[6]) + j * (G[5] + G[7]) + r * (G[3] - G[1])) + dN * 2.f * k;
    const float dth = dr * (-q.sh * 0.5f) + ((di * aa[0] + dj * aa[1]) + dkk * aa[2]) * q.dk;
    const float inv = q.th > 0.f ? 1.f / q.th : 0.f;       // the norm's gradient at the zero vector is 0 (torch)
    g[0] = di * q.kf + dth * aa[0] * inv;
    g[1] = dj * q.kf + dth * aa[1] * inv;
    g[2] = dkk * q.kf + dth * aa[2] * inv;
}

struct AdamK { float step, bc2s; };     // lr / (1 - beta1^t), sqrt(1 - beta2^t)
__device__ __forceinline__ float adam_update(float p, float g, float &m, float &v, AdamK a)
{
    m = m * 0.9f + g * (1.f - 0.9f);
    v = v * 0.999f + g * g * (1.f - 0.999f);
    return p - a.step * (m / (sqrtf(v) / a.bc2s + 1e-8f));
}

// one pixel of one iteration: rotation field sample -> residual -> gradient w.r.t. the sampled axis-angle (first half,
// UPDATE_SRC = false) or Adam step of the normal itself (second half)
template <bool UPDATE_SRC>
__global__ __launch_bounds__(256) void fus_pixel_kernel(const float *__restrict__ rot, float *__restrict__ src, const float *__restrict__ tar,
                                                        const uint8_t *__restrict__ valid, const int *__restrict__ count, int H, int W,
                                                        float *__restrict__ g_up, float *__restrict__ am, float *__restrict__ av, AdamK ak)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    if (!valid[p]) {
        if (!UPDATE_SRC) { g_up[3 * (size_t)p] = 0.f; g_up[3 * (size_t)p + 1] = 0.f; g_up[3 * (size_t)p + 2] = 0.f; }
        return;
    }
    const int y = p / W, x = p % W;
    // resize_img: bilinear, align_corners=True
    const float py = (float)y * ((float)(GRID - 1) / (float)(H - 1)), px = (float)x * ((float)(GRID - 1) / (float)(W - 1));
    const int y0 = min((int)floorf(py), GRID - 1), x0 = min((int)floorf(px), GRID - 1);
    const int y1 = min(y0 + 1, GRID - 1), x1 = min(x0 + 1, GRID - 1);
    const float ty = py - (float)y0, tx = px - (float)x0;
    float aa[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a00 = rot[(y0 * GRID + x0) * 3 + c], a01 = rot[(y0 * GRID + x1) * 3 + c];
        const float a10 = rot[(y1 * GRID + x0) * 3 + c], a11 = rot[(y1 * GRID + x1) * 3 + c];
        aa[c] = (1.f - ty) * ((1.f - tx) * a00 + tx * a01) + ty * ((1.f - tx) * a10 + tx * a11);
    }
    float R[9]; Rot q;
    aa_forward(aa, R, q);
    float s[3] = {src[3 * (size_t)p], src[3 * (size_t)p + 1], src[3 * (size_t)p + 2]};
    const float *t = tar + 3 * (size_t)p;
    const float sc = 2.f / (3.f * (float)*count);
    float gr[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) gr[a] = (((R[3 * a] * s[0] + R[3 * a + 1] * s[1]) + R[3 * a + 2] * s[2]) - t[a]) * sc;
    if (!UPDATE_SRC) {
        float G[9], g[3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) G[3 * a + b] = gr[a] * s[b];
        aa_backward(aa, q, G, g);
        g_up[3 * (size_t)p] = g[0]; g_up[3 * (size_t)p + 1] = g[1]; g_up[3 * (size_t)p + 2] = g[2];
    } else {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float gs = (R[b] * gr[0] + R[3 + b] * gr[1]) + R[6 + b] * gr[2];
            float m = am[3 * (size_t)p + b], v = av[3 * (size_t)p + b];
            src[3 * (size_t)p + b] = adam_update(s[b], gs, m, v, ak);
            am[3 * (size_t)p + b] = m; av[3 * (size_t)p + b] = v;
        }
    }
}

// The SECOND half of the iterations (normal_fusion.py:134-139: only the normals are stepped, the rotation field is frozen) has no coupling between
// pixels at all: a pixel's Adam steps read its own normal, its own target and the rotation sampled at its own position.  Up to SRC_STEPS of them are
// therefore ONE launch -- the rotation matrix built once, normal and moments in registers -- instead of one launch of fus_pixel_kernel<true> each
// (round 4: 50 launches of 4.7 us behind one another).  Same operations in the same order per pixel: the same bits.
constexpr int SRC_STEPS = 64;
struct AdamSteps { int n; AdamK ak[SRC_STEPS]; };
__global__ __launch_bounds__(256) void fus_src_steps_kernel(const float *__restrict__ rot, float *__restrict__ src, const float *__restrict__ tar,
                                                            const uint8_t *__restrict__ valid, const int *__restrict__ count, int H, int W,
                                                            float *__restrict__ am, float *__restrict__ av, AdamSteps st)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W || !valid[p]) return;
    const int y = p / W, x = p % W;
    const float py = (float)y * ((float)(GRID - 1) / (float)(H - 1)), px = (float)x * ((float)(GRID - 1) / (float)(W - 1));
    const int y0 = min((int)floorf(py), GRID - 1), x0 = min((int)floorf(px), GRID - 1);
    const int y1 = min(y0 + 1, GRID - 1), x1 = min(x0 + 1, GRID - 1);
    const float ty = py - (float)y0, tx = px - (float)x0;
    float aa[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a00 = rot[(y0 * GRID + x0) * 3 + c], a01 = rot[(y0 * GRID + x1) * 3 + c];
        const float a10 = rot[(y1 * GRID + x0) * 3 + c], a11 = rot[(y1 * GRID + x1) * 3 + c];
        aa[c] = (1.f - ty) * ((1.f - tx) * a00 + tx * a01) + ty * ((1.f - tx) * a10 + tx * a11);
    }
    float R[9]; Rot q;
    aa_forward(aa, R, q);
    float s[3] = {src[3 * (size_t)p], src[3 * (size_t)p + 1], src[3 * (size_t)p + 2]};
    const float t[3] = {tar[3 * (size_t)p], tar[3 * (size_t)p + 1], tar[3 * (size_t)p + 2]};
    float m[3] = {am[3 * (size_t)p], am[3 * (size_t)p + 1], am[3 * (size_t)p + 2]}, v[3] = {av[3 * (size_t)p], av[3 * (size_t)p + 1], av[3 * (size_t)p + 2]};
    const float sc = 2.f / (3.f * (float)*count);
    for (int it = 0; it < st.n; ++it) {
        float gr[3], sn[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) gr[a] = (((R[3 * a] * s[0] + R[3 * a + 1] * s[1]) + R[3 * a + 2] * s[2]) - t[a]) * sc;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float gs = (R[b] * gr[0] + R[3 + b] * gr[1]) + R[6 + b] * gr[2];
            sn[b] = adam_update(s[b], gs, m[b], v[b], st.ak[it]);
        }
        s[0] = sn[0]; s[1] = sn[1]; s[2] = sn[2];
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) { src[3 * (size_t)p + b] = s[b]; am[3 * (size_t)p + b] = m[b]; av[3 * (size_t)p + b] = v[b]; }
}

// one node of the rotation grid per wave: the lanes share the node's transposed-bilinear footprint of g_up (up to ~19 x 19 pixels at
// 512 / 64): lane & 31 walks the columns, lane >> 5 the rows two at a time, ten rows per lane in flight before the first is consumed (one
// memory latency per node instead of one per 64 pixels); a fixed-order butterfly sums the lanes (deterministic).  The eight neighbours of
// the smoothness term go to eight lanes, the three components of the Adam step to three.
__global__ __launch_bounds__(64) void fus_grid_kernel(const float *__restrict__ rot_in, float *__restrict__ rot_out, const float *__restrict__ g_up,
                                                      int H, int W, float *__restrict__ am, float *__restrict__ av, AdamK ak)
{
    const int n = blockIdx.x, lane = threadIdx.x;
    const int Y = n / GRID, X = n % GRID;
    const float sy = (float)(GRID - 1) / (float)(H - 1), sx = (float)(GRID - 1) / (float)(W - 1);
    const int ya = max(0, (int)ceilf((float)(Y - 1) / sy) - 1), yb = min(H - 1, (int)floorf((float)(Y + 1) / sy) + 1);
    const int xa = max(0, (int)ceilf((float)(X - 1) / sx) - 1), xb = min(W - 1, (int)floorf((float)(X + 1) / sx) + 1);
    constexpr int ROWS = 10;
    float g[3] = {0.f, 0.f, 0.f};
    for (int x = xa + (lane & 31); x <= xb; x += 32) {
        const float px = (float)x * sx;
        const int x0 = min((int)floorf(px), GRID - 1), x1 = min(x0 + 1, GRID - 1);
        const float tx = px - (float)x0;
        const float wx = (x0 == X ? 1.f - tx : 0.f) + (x1 == X ? tx : 0.f);
        for (int yy = ya + (lane >> 5); yy <= yb; yy += 2 * ROWS) {
            float v[ROWS][3], w[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const int y = yy + 2 * r;
                const bool in = y <= yb;
                const int yc = in ? y : yb;
                const float py = (float)yc * sy;
                const int y0 = min((int)floorf(py), GRID - 1), y1 = min(y0 + 1, GRID - 1);
                const float ty = py - (float)y0;
                const float wy = (y0 == Y ? 1.f - ty : 0.f) + (y1 == Y ? ty : 0.f);
                w[r] = in ? wy * wx : 0.f;
                const float *gp = g_up + 3 * ((size_t)yc * W + x);
                v[r][0] = gp[0]; v[r][1] = gp[1]; v[r][2] = gp[2];
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { g[0] += w[r] * v[r][0]; g[1] += w[r] * v[r][1]; g[2] += w[r] * v[r][2]; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { g[0] += __shfl_xor(g[0], o); g[1] += __shfl_xor(g[1], o); g[2] += __shfl_xor(g[2], o); }
    // smoothness: sum over the 8 neighbour images of mean((shift(rot) - rot)^2), zero padding (normal_fusion.py:66-78,127-131)
    const float cM = 2.f / (float)(GRID * GRID * 3);
    const float *c = rot_in + 3 * n;
    float t[3] = {0.f, 0.f, 0.f};
    if (lane < 8) {
        const int idx = lane < 4 ? lane : lane + 1, di = idx / 3 - 1, dj = idx % 3 - 1;
        const int yp = Y + di, xp = X + dj, ym = Y - di, xm = X - dj;
        const bool inp = yp >= 0 && yp < GRID && xp >= 0 && xp < GRID, inm = ym >= 0 && ym < GRID && xm >= 0 && xm < GRID;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float fwd = c[k] - (inp ? rot_in[(yp * GRID + xp) * 3 + k] : 0.f);
            const float bwd = inm ? c[k] - rot_in[(ym * GRID + xm) * 3 + k] : 0.f;
            t[k] = cM * (fwd + bwd);
        }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) { t[0] += __shfl_xor(t[0], o); t[1] += __shfl_xor(t[1], o); t[2] += __shfl_xor(t[2], o); }
    if (lane >= 3) return;
    const int k = lane;
    const float gk = (k == 0 ? g[0] : k == 1 ? g[1] : g[2]) + (k == 0 ? t[0] : k == 1 ? t[1] : t[2]);
    float m = am[3 * n + k], v = av[3 * n + k];
    rot_out[3 * n + k] = adam_update(c[k], gk, m, v, ak);
    am[3 * n + k] = m; av[3 * n + k] = v;
}

// merge_normal_images_cover (normal_fusion.py:158-167): the observed normal wherever there is one
__global__ void fus_cover_kernel(const float *__restrict__ src, const float *__restrict__ tar, int64_t n, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *t = tar + 3 * i, *a = src + 3 * i;
    const bool m = sqrtf((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]) > 1e-6f;
    out[3 * i] = m ? t[0] : a[0]; out[3 * i + 1] = m ? t[1] : a[1]; out[3 * i + 2] = m ? t[2] : a[2];
}

// Python's slice(start, stop) on an axis of length n
__host__ __device__ inline void py_slice(int start, int stop, int n, int &lo, int &hi)
{
    lo = start < 0 ? start + n : start; hi = stop < 0 ? stop + n : stop;
    lo = lo < 0 ? 0 : (lo > n ? n : lo); hi = hi < 0 ? 0 : (hi > n ? n : hi);
}

// distance-transform blend and the face rectangle (normal_fusion.py:143-153)
__global__ void fus_blend_kernel(const float *__restrict__ src, const float *__restrict__ init, const float *__restrict__ dtm, int H, int W,
                                 int r0, int r1, int c0, int c1, float *__restrict__ out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p % W;
    const float d = dtm[p] / 5.f;
    const float w0 = d > 1.f ? 0.f : 1.f;
    const bool face = y >= r0 && y < r1 && x >= c0 && x < c1;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = init[3 * (size_t)p + k];
        out[3 * (size_t)p + k] = face ? a : (src[3 * (size_t)p + k] * d + a * w0) / (d + w0);
    }
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


// The iteration loop is launch-bound (150 dependent launches of a few microseconds each): it is built once per
// (image size, iteration count, scratch allocation) as a hipGraph of kernel nodes whose arguments only reference the
// context's scratch, cached in the context and replayed with one hipGraphLaunch per call.
namespace {
struct FusionBuffers { float *src, *tar, *sm, *sv, *g_up, *dt_g, *dtm, *rot_a, *rot_b, *rm, *rv; int *count; uint8_t *smask, *tmask, *emask, *valid; };

template <typename... Args>
int add_kernel_node(hipGraph_t g, hipGraphNode_t &prev, bool &has_prev, const void *func, dim3 grid, dim3 block, Args... args)
{
    void *params[] = {(void *)&args...};
    hipKernelNodeParams p{};
    p.func = const_cast<void *>(func); p.gridDim = grid; p.blockDim = block; p.sharedMemBytes = 0; p.kernelParams = params; p.extra = nullptr;
    hipGraphNode_t node;
    AVC_HIP(hipGraphAddKernelNode(&node, g, has_prev ? &prev : nullptr, has_prev ? 1 : 0, &p));
    prev = node; has_prev = true;
    return AVC_OK;
}

// enqueue (graph == nullptr) or record (graph != nullptr) the iterations
int fusion_iterations(const FusionBuffers &B, int H, int W, int iter_num, hipStream_t s, hipGraph_t graph)
{
    const dim3 blk(256), grd((unsigned)(((size_t)H * W + 255) / 256));
    float *rin = B.rot_a, *rout = B.rot_b;
    int t_rot = 0, t_src = 0;
    hipGraphNode_t prev{}; bool has_prev = false;
    for (int it = 0; it < iter_num; ++it) {
        if (it < iter_num / 2.0) {                                                               // normal_fusion.py:134
            ++t_rot;
            const AdamK ak{(float)(1e-2 / (1.0 - std::pow(0.9, t_rot))), (float)std::sqrt(1.0 - std::pow(0.999, t_rot))};
            if (graph) {
                if (int rc = add_kernel_node(graph, prev, has_prev, (const void *)fus_pixel_kernel<false>, grd, blk, (const float *)rin, B.src, (const float *)B.tar,
                                             (const uint8_t *)B.valid, (const int *)B.count, H, W, B.g_up, B.sm, B.sv, ak)) return rc;
                if (int rc = add_kernel_node(graph, prev, has_prev, (const void *)fus_grid_kernel, dim3(GRID * GRID), dim3(64), (const float *)rin, rout,
                                             (const float *)B.g_up, H, W, B.rm, B.rv, ak)) return rc;
            } else {
                hipLaunchKernelGGL(fus_pixel_kernel<false>, grd, blk, 0, s, rin, B.src, B.tar, B.valid, B.count, H, W, B.g_up, B.sm, B.sv, ak);
                hipLaunchKernelGGL(fus_grid_kernel, dim3(GRID * GRID), dim3(64), 0, s, rin, rout, B.g_up, H, W, B.rm, B.rv, ak);
            }
            std::swap(rin, rout);
        } else {
            // the remaining iterations step the normals only: batches of up to SRC_STEPS steps per launch (fus_src_steps_kernel)
            AdamSteps st{};
            for (; it < iter_num && st.n < SRC_STEPS; ++it) {
                ++t_src;
                st.ak[st.n++] = AdamK{(float)(1e-1 / (1.0 - std::pow(0.9, t_src))), (float)std::sqrt(1.0 - std::pow(0.999, t_src))};
            }
            --it;                                                                                // (the loop's own increment)
            if (graph) {
                if (int rc = add_kernel_node(graph, prev, has_prev, (const void *)fus_src_steps_kernel, grd, blk, (const float *)rin, B.src, (const float *)B.tar,
                                             (const uint8_t *)B.valid, (const int *)B.count, H, W, B.sm, B.sv, st)) return rc;
            } else {
                hipLaunchKernelGGL(fus_src_steps_kernel, grd, blk, 0, s, rin, B.src, B.tar, B.valid, B.count, H, W, B.sm, B.sv, st);
            }
        }
    }
    return AVC_OK;
}
}  // namespace

void release_fusion_graph(avc_ctx *ctx)
{
    if (ctx->fusion_graph_exec) hipGraphExecDestroy(static_cast<hipGraphExec_t>(ctx->fusion_graph_exec));
    if (ctx->fusion_graph) hipGraphDestroy(static_cast<hipGraph_t>(ctx->fusion_graph));
    ctx->fusion_graph_exec = nullptr; ctx->fusion_graph = nullptr;
}

int merge_normal_images(avc_ctx *ctx, const float *src_in, const float *tar_in, int H, int W, int iter_num, int neck_x, int neck_y,
                        float *out, hipStream_t s)
{
    const size_t np = (size_t)H * W;
    // scratch: working copies of both maps, Adam moments of src, g_up, dt (2 buffers), rot ping-pong + moments, counter, masks (4)
    const size_t fbytes = sizeof(float) * (3 * np * 5 + 2 * np + 4 * GRID * GRID * 3) + 4 * np + 256;
    if (ctx->fusion_scratch_bytes < fbytes) {
        release_fusion_graph(ctx);
        if (ctx->fusion_scratch) AVC_HIP(hipFree(ctx->fusion_scratch));
        ctx->fusion_scratch = nullptr; ctx->fusion_scratch_bytes = 0;
        AVC_HIP(hipMalloc(&ctx->fusion_scratch, fbytes));
        ctx->fusion_scratch_bytes = fbytes;
    }
    FusionBuffers B;
    float *f = static_cast<float *>(ctx->fusion_scratch);
    B.src = f; f += 3 * np;
    B.tar = f; f += 3 * np;
    B.sm = f; f += 3 * np;
    B.sv = f; f += 3 * np;
    B.g_up = f; f += 3 * np;
    B.dt_g = f; f += np;
    B.dtm = f; f += np;
    B.rot_a = f; f += GRID * GRID * 3;
    B.rot_b = f; f += GRID * GRID * 3;
    B.rm = f; f += GRID * GRID * 3;
    B.rv = f; f += GRID * GRID * 3;
    B.count = reinterpret_cast<int *>(f); f += 64;
    B.smask = reinterpret_cast<uint8_t *>(f); B.tmask = B.smask + np; B.emask = B.tmask + np; B.valid = B.emask + np;
    AVC_HIP(hipMemcpyAsync(B.src, src_in, sizeof(float) * 3 * np, hipMemcpyDeviceToDevice, s));
    AVC_HIP(hipMemcpyAsync(B.tar, tar_in, sizeof(float) * 3 * np, hipMemcpyDeviceToDevice, s));
    AVC_HIP(hipMemsetAsync(B.sm, 0, sizeof(float) * 6 * np, s));                                  // sm, sv
    AVC_HIP(hipMemsetAsync(B.rot_a, 0, sizeof(float) * 4 * GRID * GRID * 3 + 256, s));            // rot_a, rot_b, rm, rv, count
    const dim3 blk(256), grd((unsigned)((np + 255) / 256));
    hipLaunchKernelGGL(fus_masks_kernel, grd, blk, 0, s, src_in, tar_in, (int)np, B.smask, B.tmask);
    hipLaunchKernelGGL(fus_erode_kernel, grd, blk, 0, s, B.tmask, H, W, 3, B.emask);
    hipLaunchKernelGGL(fus_dt_rows_kernel, dim3(H), dim3(64), 0, s, B.emask, H, W, B.dt_g);
    hipLaunchKernelGGL(fus_dt_cols_kernel, dim3((W + 63) / 64), dim3(64 * DT_SEGS), 0, s, B.dt_g, H, W, B.dtm);
    hipLaunchKernelGGL(fus_valid_kernel, grd, blk, 0, s, B.smask, B.emask, (int)np, B.valid, B.count);
    if (iter_num > 0) {
        if (!ctx->opt.fusion_graph) {                                                            // avc_set_option "fusion_graph" 0: plain launches (A/B)
            if (int rc = fusion_iterations(B, H, W, iter_num, s, nullptr)) return rc;
        } else {
            if (!ctx->fusion_graph_exec || ctx->fusion_graph_H != H || ctx->fusion_graph_W != W || ctx->fusion_graph_iters != iter_num) {
                release_fusion_graph(ctx);
                hipGraph_t g;
                AVC_HIP(hipGraphCreate(&g, 0));
                ctx->fusion_graph = g;
                if (int rc = fusion_iterations(B, H, W, iter_num, s, g)) { release_fusion_graph(ctx); return rc; }
                hipGraphExec_t ex;
                AVC_HIP(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
                ctx->fusion_graph_exec = ex; ctx->fusion_graph_H = H; ctx->fusion_graph_W = W; ctx->fusion_graph_iters = iter_num;
            }
            AVC_HIP(hipGraphLaunch(static_cast<hipGraphExec_t>(ctx->fusion_graph_exec), s));
        }
    }
    int r0, r1, c0, c1;
    py_slice(neck_y - 90, neck_y, H, r0, r1);
    py_slice(neck_x - 35, neck_x + 35, W, c0, c1);
    hipLaunchKernelGGL(fus_blend_kernel, grd, blk, 0, s, B.src, src_in, B.dtm, H, W, r0, r1, c0, c1, out);
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

int merge_normal_images_cover(const float *src, const float *tar, int64_t npix, float *out, hipStream_t s)
{
    if (npix == 0) return AVC_OK;
    hipLaunchKernelGGL(fus_cover_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, src, tar, npix, out);
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

}  // namespace avc
