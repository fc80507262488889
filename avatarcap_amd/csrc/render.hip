// The colour path of the test loop on the device (reference main.py:464-477): one ray per avatar vertex through the texture template,
// alpha-composited.  NerfRenderer.render(pts_space='cano') in eval mode (network/arch_avatar.py:240-349) is four launches per chunk of rays:
//   ray_points_kernel     get_wsampling_points (:248-254): z = near (1 - t) + far t, t = linspace(0, 1, S); pts = o + d z
//   the fused query       GeoTexAvatar.forward's network part (:210-213): occupancy, offsets and rgba of every sample (fused_mlp.hip, colour head on)
//   near_flag_kernel      the near flag (:208-209): is a canonical SMPL vertex closer than 0.08 (knn_lbs.hip: the K = 1 search's d2 < 0.08^2)
//   composite_kernel      the rest of forward (:213, :222-230: the offset point inside the canonical bounds, near the body, alpha = 1 - exp(-sigma dist))
//                         and raw2outputs (utils/nerf_util.py:185-212: transmittance as the running product, weights, rgb / depth / acc / disp maps)
// -- one wavefront per ray, 64 samples = its 64 lanes; the running product is taken in sample order, as torch.cumprod does on the reference's CPU path.
// Also here: the canonical blend-weight volume's trilinear fetch (CanoBlendWeightVolume.forward, arch_avatar.py:143-165) for the posed branch.
// Built with -ffp-contract=off: the sample positions are the reference's float32 expressions, operation by operation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "avcap_internal.h"

namespace avc {
namespace {

// torch.linspace(0, 1, S) in float32 (ATen RangeFactories: step = (end - start) / (steps - 1); the first half counts up from start, the second down from end)
__device__ __forceinline__ float linspace01(int i, int S)
{
    if (S == 1) return 0.0f;
    const float step = 1.0f / (float)(S - 1);
    return i < S / 2 ? step * (float)i : 1.0f - step * (float)(S - i - 1);
}

struct RayArgs {
    const float *ray_o, *ray_d, *near, *far, *depth;       // (P, 3), (P, 3), (P), (P), (P) or null
    const float *t_vals;                                   // (S) or null: linspace(0, 1, S) as the caller's host computed it
    float near_dist, far_dist;
    int64_t p0, np;                                        // this chunk: rays [p0, p0 + np)
    int S;
    float *pts, *z;                                        // (np S, 3), (np S)
};

__global__ __launch_bounds__(256) void ray_points_kernel(const RayArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.np * a.S) return;
    const int64_t r = i / a.S, p = a.p0 + r;
    const int s = (int)(i - r * a.S);
    float nr = a.near[p], fr = a.far[p];
    if (a.depth) {                                         // get_pixel_value (:289-291): rays with a depth sample around it
        const float d = a.depth[p];
        if (d > 1e-6f) { nr = d - a.near_dist; fr = d + a.far_dist; }
    }
    const float t = a.t_vals ? a.t_vals[s] : linspace01(s, a.S);
    const float z = nr * (1.0f - t) + fr * t;              // :252
    a.z[i] = z;
#pragma unroll
    for (int c = 0; c < 3; ++c) a.pts[3 * i + c] = a.ray_o[3 * p + c] + a.ray_d[3 * p + c] * z;      // :253
}

struct CompArgs {
    int64_t p0, np;
    int S;
    const float *pts, *z, *off, *rgba, *d2;                // per sample of the chunk
    float lo[3], hi[3];                                    // batch['cano_bounds']
    float thr2;                                            // 0.08^2
    float *rgb_map, *acc_map, *depth_map, *disp_map;       // (P, 3), (P), (P), (P); each may be null
    float *weights, *raw;                                  // (P, S), (P S, 4); may be null
};

__global__ __launch_bounds__(256) void composite_kernel(const CompArgs a)
{
    __shared__ float fac[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= a.np) return;
    const int64_t p = a.p0 + r;
    float carry = 1.0f, sr = 0.0f, sg = 0.0f, sb = 0.0f, sd = 0.0f, sa = 0.0f;
    for (int base = 0; base < a.S; base += 64) {
        const int s = base + lane;
        const bool live = s < a.S;
        float alpha = 0.0f, z = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
        if (live) {
            const int64_t i = r * a.S + s;
            const float4 c = *reinterpret_cast<const float4 *>(a.rgba + 4 * i);
            cr = c.x; cg = c.y; cb = c.z; alpha = c.w;
            bool keep = a.d2[i] < a.thr2;                                      // near_flag (:208-209)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float q = a.pts[3 * i + k] + a.off[3 * i + k];            // cano_pts + offsets (:213)
                keep = keep && q > a.lo[k] && q < a.hi[k];                     // inside the canonical bounds (:222-224)
            }
            if (!keep) alpha = 0.0f;
            z = a.z[i];
            // dists (:279-281): to the next sample; the last one repeats its predecessor's
            const float dist = s + 1 < a.S ? a.z[i + 1] - z : (a.S > 1 ? z - a.z[i - 1] : 0.0f);
            alpha = 1.0f - expf(-alpha * dist);                                // :228-230
            if (a.raw) *reinterpret_cast<float4 *>(a.raw + 4 * (p * a.S + s)) = make_float4(cr, cg, cb, alpha);
        }
        fac[wave][lane] = live ? 1.0f - alpha + 1e-10f : 1.0f;
        __builtin_amdgcn_wave_barrier();
        // transmittance: the product of the factors before this sample, multiplied in sample order (torch.cumprod of [1, f0, f1, ...][:-1])
        float T = carry, all = carry;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) {
            const float f = fac[wave][k];
            if (k < lane) T *= f;
            all *= f;
        }
        carry = all;
        __builtin_amdgcn_wave_barrier();
        const float w = alpha * T;
        if (live && a.weights) a.weights[p * a.S + s] = w;
        sr += w * cr; sg += w * cg; sb += w * cb; sd += w * z; sa += w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sr += __shfl_xor(sr, o); sg += __shfl_xor(sg, o); sb += __shfl_xor(sb, o); sd += __shfl_xor(sd, o); sa += __shfl_xor(sa, o);
    }
    if (lane == 0) {
        if (a.rgb_map) { a.rgb_map[3 * p] = sr; a.rgb_map[3 * p + 1] = sg; a.rgb_map[3 * p + 2] = sb; }
        if (a.depth_map) a.depth_map[p] = sd;
        if (a.acc_map) a.acc_map[p] = sa;
        if (a.disp_map) a.disp_map[p] = 1.0f / fmaxf(1e-10f, sd / sa);         // nerf_util.py:207
    }
}

// F.grid_sample(volume (1, C, X, Y, Z), grid = (2 pts - 1)[..., [2, 1, 0]], padding_mode='border', align_corners=True) on the channel-last volume:
// thread = one point x 4 channels.  Coordinates, corner weights and the order of the eight products are ATen's (GridSampler: unnormalize, clip, trilinear).
__global__ __launch_bounds__(256) void blend_weight_kernel(const float *__restrict__ vol, int X, int Y, int Z, int C, const float *__restrict__ pts, int64_t n,
                                                           float *__restrict__ out)
{
    const int c4n = C >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * c4n) return;
    const int64_t pi = i / c4n;
    const int c4 = (int)(i - pi * c4n);
    auto coord = [](float u, int size) {
        const float g = 2.0f * u - 1.0f;
        float x = ((g + 1.0f) / 2.0f) * (float)(size - 1);
        x = fminf((float)(size - 1), fmaxf(x, 0.0f));
        return x;
    };
    const float fx = coord(pts[3 * pi], X), fy = coord(pts[3 * pi + 1], Y), fz = coord(pts[3 * pi + 2], Z);
    const float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
    const int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
    const float ax = fx - x0f, ay = fy - y0f, az = fz - z0f;           // distance to the low corner; (corner + 1) - f to the high one
    const float bx = (x0f + 1.0f) - fx, by = (y0f + 1.0f) - fy, bz = (z0f + 1.0f) - fz;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // ATen's order: D (here X) low then high; within it H (Y) low / high; within it W (Z) low / high.  Weight = (w term)(h term)(d term).
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int dx = k >> 2, dy = (k >> 1) & 1, dz = k & 1;
        const int xi = x0 + dx, yi = y0 + dy, zi = z0 + dz;
        if (xi >= X || yi >= Y || zi >= Z) continue;                   // within_bounds (the weight of such a corner is zero)
        const float w = ((dz ? az : bz) * (dy ? ay : by)) * (dx ? ax : bx);
        const float4 v = *reinterpret_cast<const float4 *>(vol + (((size_t)xi * Y + yi) * Z + zi) * C + 4 * c4);
        acc.x += v.x * w; acc.y += v.y * w; acc.z += v.z * w; acc.w += v.w * w;
    }
    *reinterpret_cast<float4 *>(out + pi * C + 4 * c4) = acc;
}

}  // namespace

int render_rays_cano(avc_ctx *ctx, const float *ray_o, const float *ray_d, const float *near, const float *far, const float *depth, float near_dist,
                     float far_dist, const float *t_vals, int64_t P, int S, const float center[3], const float bounds[6], const float *smpl_v, int32_t n_smpl, int occ_sigmoid,
                     float *rgb_map, float *acc_map, float *depth_map, float *disp_map, float *weights, float *raw, hipStream_t s)
{
    if (P == 0) return AVC_OK;
    // rays per pass: bounds the per-sample scratch (points, z, offsets, rgba, occupancy, distances: 52 bytes per sample) to ~220 MB
    const int64_t chunk = std::max<int64_t>(1, ((int64_t)1 << 22) / S);
    const int64_t ns = std::min(P, chunk) * S;
    const size_t bytes = (size_t)ns * 52 + 256;
    if (ctx->render_scratch_bytes < bytes) {
        AVC_HIP(hipStreamSynchronize(s));
        if (ctx->render_scratch) AVC_HIP(hipFree(ctx->render_scratch));
        ctx->render_scratch = nullptr; ctx->render_scratch_bytes = 0;
        AVC_HIP(hipMalloc(&ctx->render_scratch, bytes));
        ctx->render_scratch_bytes = bytes;
    }
    float *rgba = static_cast<float *>(ctx->render_scratch);          // 16-byte aligned first
    float *pts = rgba + 4 * ns, *off = pts + 3 * ns, *z = off + 3 * ns, *occ = z + ns, *d2 = occ + ns;
    for (int64_t p0 = 0; p0 < P; p0 += chunk) {
        const int64_t np = std::min(chunk, P - p0), n = np * S;
        RayArgs ra{ray_o, ray_d, near, far, depth, t_vals, near_dist, far_dist, p0, np, S, pts, z};
        hipLaunchKernelGGL(ray_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ra);
        const int rc = ctx->check_range ? checked::launch_avatar(ctx, pts, nullptr, n, center, occ_sigmoid, occ, off, rgba, false, s)
                       : needs_scaled_kernels(ctx) ? scaled::launch_avatar(ctx, pts, nullptr, n, center, occ_sigmoid, occ, off, rgba, false, s)
                                                   : plain::launch_avatar(ctx, pts, nullptr, n, center, occ_sigmoid, occ, off, rgba, false, s);
        if (rc) return rc;
        if (int rk = near_flags(ctx, pts, n, smpl_v, n_smpl, (float)(0.08 * 0.08), d2, s)) return rk;
        CompArgs ca{p0, np, S, pts, z, off, rgba, d2, {bounds[0], bounds[1], bounds[2]}, {bounds[3], bounds[4], bounds[5]}, (float)(0.08 * 0.08),
                    rgb_map, acc_map, depth_map, disp_map, weights, raw};
        hipLaunchKernelGGL(composite_kernel, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, s, ca);
    }
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

int blend_weight_sample(const float *vol, const int32_t res[3], int C, const float *pts01, int64_t n, float *out, hipStream_t s)
{
    if (n == 0) return AVC_OK;
    const int64_t threads = n * (C / 4);
    hipLaunchKernelGGL(blend_weight_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, vol, res[0], res[1], res[2], C, pts01, n, out);
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

}  // namespace avc
