// SMPL utilities on the device (reference utils/smpl_util.py):
//   knn            : pytorch3d.ops.knn_points as the reference uses it (smpl_util.py:33,
//                    avatarcap_dataset.py:114, arch_avatar.py:190,208).  Exact K nearest, squared L2
//                    ascending, ties -> lower index.  The reference points are binned into a uniform grid
//                    (<= 32^3 cells, rebuilt on the device per call: one workgroup, tens of us for the 6890
//                    SMPL vertices).  A WAVE of coherent queries searches together: it takes the cell bounding box
//                    of its 64 queries and scans the cells of that box, then ring after ring around it, until every
//                    lane's K-th best is closer than the nearest unscanned cell face; control flow and candidate
//                    addresses are wave-uniform, so candidates arrive through scalar loads at ~9 VALU per lane each.
//                    A wave whose queries span a large box (marching-cubes vertices in the library's order trace the
//                    contour of a slice; scattered queries) lets every lane search the 27 cells around its own query
//                    instead (knn_lane_scan).  Reference sets too small for a grid use the exhaustive scan
//                    (reference points staged through LDS in tiles).
//   calculate_lbs  : KNN-4 + Gaussian weights + gather/blend of the 24-wide skin weights (:24-39), fused
//   skinning       : per-point blend of the 24 joint 4x4s and its application to points / normals (:58-81)
// All HBM/VALU-bound elementwise work; queries are read once, coalesced.
#include "store_settle.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <vector>

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


// ---- uniform grid over the reference points ----------------------------------------------------
constexpr int GRID_MAX_AXIS = 128;              // cells per axis (upper bound)
constexpr int GRID_MIN_REFS = 512;               // below: exhaustive scan only
constexpr int LANE_BOX = 64;                     // cells of a wave's box (+ one ring) above which its lanes search on their own (knn_lane_scan)
struct GridHdr { float ox, oy, oz, h, inv_h, eps; int nx, ny, nz, ncell; };

__device__ __forceinline__ int cell_coord(float v, float o, float inv_h, int n)
{
    const int c = (int)floorf((v - o) * inv_h);
    return min(max(c, 0), n - 1);
}

__device__ __forceinline__ float wave_minf(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_maxf(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ int wave_mini(int v) { for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o)); return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int wave_maxi(int v) { for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o)); return __builtin_amdgcn_readfirstlane(v); }

// ---- grid construction (stream-ordered, no host round trip) ---------------------------------------
//   bbox (atomics on order-preserving keys) -> header -> histogram -> exclusive scan (one workgroup)
//   -> scatter of {x, y, z, index} by cell (z fastest, so a run of cells along z is one contiguous range)
__device__ __forceinline__ unsigned f2key(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float key2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ __launch_bounds__(256) void grid_bbox_kernel(const float *__restrict__ ref, int nr, unsigned *keys /* [6] min xyz, max xyz */)
{
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += gridDim.x * blockDim.x)
        for (int a = 0; a < 3; ++a) { const float v = ref[(size_t)i * 3 + a]; mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v); }
    for (int a = 0; a < 3; ++a) {
        const float lo = wave_minf(mn[a]), hi = wave_maxf(mx[a]);
        if ((threadIdx.x & 63) == 0) { atomicMin(&keys[a], f2key(lo)); atomicMax(&keys[3 + a], f2key(hi)); }
    }
}

__global__ void grid_header_kernel(const unsigned *keys, int axis, GridHdr *hdr)
{
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = key2f(keys[a]); hi[a] = key2f(keys[3 + a]); }
    GridHdr H;
    const float ext = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
    float h = ext / (float)axis;
    if (!(h > 0.f) || !(h < __builtin_inff())) h = 1.f;            // all points identical / non-finite input: one cell per axis
    H.ox = lo[0]; H.oy = lo[1]; H.oz = lo[2]; H.h = h; H.inv_h = 1.f / h;
    const float mag = fmaxf(fmaxf(fmaxf(fabsf(lo[0]), fabsf(hi[0])), fmaxf(fabsf(lo[1]), fabsf(hi[1]))), fmaxf(fabsf(lo[2]), fabsf(hi[2])));
    H.eps = 8e-6f * (mag + ext) + 1e-30f;                           // rounding slop of the cell assignment (see knn_grid_scan)
    H.nx = min(axis, (int)floorf((hi[0] - lo[0]) * H.inv_h) + 1);
    H.ny = min(axis, (int)floorf((hi[1] - lo[1]) * H.inv_h) + 1);
    H.nz = min(axis, (int)floorf((hi[2] - lo[2]) * H.inv_h) + 1);
    if (!(H.nx >= 1)) H.nx = 1;
    if (!(H.ny >= 1)) H.ny = 1;
    if (!(H.nz >= 1)) H.nz = 1;
    H.ncell = H.nx * H.ny * H.nz;
    *hdr = H;
}

__device__ __forceinline__ int cell_of(const GridHdr &H, float x, float y, float z)
{
    return (cell_coord(x, H.ox, H.inv_h, H.nx) * H.ny + cell_coord(y, H.oy, H.inv_h, H.ny)) * H.nz + cell_coord(z, H.oz, H.inv_h, H.nz);
}

__global__ __launch_bounds__(256) void grid_count_kernel(const float *__restrict__ ref, int nr, const GridHdr *__restrict__ hdr, int *start)
{
    const GridHdr H = *hdr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += gridDim.x * blockDim.x)
        atomicAdd(&start[cell_of(H, ref[(size_t)i * 3], ref[(size_t)i * 3 + 1], ref[(size_t)i * 3 + 2]) + 1], 1);
}

// start[c + 1] holds the count of cell c on entry; on exit start[c] .. start[c + 1] is cell c's range and cursor[c] = start[c]
__global__ __launch_bounds__(1024) void grid_scan_kernel(const GridHdr *__restrict__ hdr, int *start, int *cursor)
{
    __shared__ int part[1024];
    const int tid = threadIdx.x, ncell = hdr->ncell;
    const int per = (ncell + 1023) / 1024;
    int sum = 0;
    for (int c = tid * per; c < min((tid + 1) * per, ncell); ++c) sum += start[c + 1];
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - sum;
    for (int c = tid * per; c < min((tid + 1) * per, ncell); ++c) { const int n = start[c + 1]; cursor[c] = run; run += n; start[c + 1] = run; }
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(const float *__restrict__ ref, int nr, const GridHdr *__restrict__ hdr, int *cursor,
                                                           float4 *sorted)
{
    const GridHdr H = *hdr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += gridDim.x * blockDim.x) {
        const float x = ref[(size_t)i * 3], y = ref[(size_t)i * 3 + 1], z = ref[(size_t)i * 3 + 2];
        sorted[atomicAdd(&cursor[cell_of(H, x, y, z)], 1)] = make_float4(x, y, z, __int_as_float(i));
    }
    if (blockIdx.x == 0 && threadIdx.x < 8)                         // read, masked, by the last trips of scan_range
        sorted[nr + threadIdx.x] = make_float4(0.f, 0.f, 0.f, __int_as_float(0x7fffffff));
}

// insertion under the total order (distance, index): candidates arrive in cell order, not index order.  A candidate that is already in the
// list is skipped: a wave whose lanes first looked around themselves (knn_lane_ring1) rescans those cells when it goes on cooperatively.
template <int K>
__device__ __forceinline__ void knn_insert_lex(float d, int id, float (&bd)[K], int (&bi)[K])
{
    if (d < bd[K - 1] || (d == bd[K - 1] && id < bi[K - 1])) {
        bool seen = false;
#pragma unroll
        for (int k = 0; k < K - 1; ++k) seen = seen || bi[k] == id;
        if (seen) return;
        bd[K - 1] = d; bi[K - 1] = id;
#pragma unroll
        for (int k = K - 1; k > 0; --k) {
            if (bd[k] < bd[k - 1] || (bd[k] == bd[k - 1] && bi[k] < bi[k - 1])) {
                const float td = bd[k]; bd[k] = bd[k - 1]; bd[k - 1] = td;
                const int ti = bi[k]; bi[k] = bi[k - 1]; bi[k - 1] = ti;
            }
        }
    }
}

__device__ __forceinline__ float cand_d2(const float4 r, float qx, float qy, float qz)
{
    // (dx*dx + dy*dy) + dz*dz with separately rounded operations (file built with -ffp-contract=off)
    const float dx = qx - r.x, dy = qy - r.y, dz = qz - r.z;
    return (dx * dx + dy * dy) + dz * dz;
}

template <int K>
__device__ __forceinline__ void scan_range(const float4 *__restrict__ sorted, int s, int e, float qx, float qy, float qz,
                                           float (&bd)[K], int (&bi)[K])
{
    // s, e, j are wave-uniform: the candidates come through scalar loads, 4 a trip, the next trip's load in
    // flight while this trip's candidates are tested (the array is padded by 8 entries: the loads may run
    // past `e`; what they bring from there is masked out)
    if (s >= e) return;
    float4 nxt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) nxt[u] = sorted[s + u];
    for (int j = s; j < e; j += 4) {
        float4 c[4]; float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = nxt[u];
#pragma unroll
        for (int u = 0; u < 4; ++u) nxt[u] = sorted[j + 4 + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = cand_d2(c[u], qx, qy, qz);
        if (j + 4 > e) {
#pragma unroll
            for (int u = 1; u < 4; ++u) if (j + u >= e) { d[u] = __builtin_inff(); c[u].w = __int_as_float(0x7fffffff); }   // never inserted
        }
        const float m = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
        if (m <= bd[K - 1]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) knn_insert_lex<K>(d[u], __float_as_int(c[u].w), bd, bi);
        }
    }
}

struct GridDims { int nx, ny, nz; };

// all reference points in the cell box [x0,x1] x [y0,y1] x [z0,z1]; cells are stored z fastest, then y, then x,
// so full-depth / full-height boxes collapse into long contiguous runs
template <int K>
__device__ __forceinline__ void scan_box(const int *__restrict__ start, const float4 *__restrict__ sorted, GridDims g, int x0, int x1, int y0, int y1,
                                         int z0, int z1, float qx, float qy, float qz, float (&bd)[K], int (&bi)[K])
{
    if (x0 > x1 || y0 > y1 || z0 > z1) return;
    auto cell = [&](int x, int y, int z) { return (x * g.ny + y) * g.nz + z; };
    auto run = [&](int ca, int cb) {
        scan_range<K>(sorted, __builtin_amdgcn_readfirstlane(start[ca]), __builtin_amdgcn_readfirstlane(start[cb + 1]), qx, qy, qz, bd, bi);
    };
    const bool zfull = z0 == 0 && z1 == g.nz - 1, yfull = y0 == 0 && y1 == g.ny - 1;
    if (zfull && yfull) { run(cell(x0, 0, 0), cell(x1, g.ny - 1, g.nz - 1)); return; }
    for (int x = x0; x <= x1; ++x) {
        if (zfull) { run(cell(x, y0, 0), cell(x, y1, g.nz - 1)); continue; }
        for (int y = y0; y <= y1; ++y) run(cell(x, y, z0), cell(x, y, z1));
    }
}

// Exact KNN of one query per lane, searched by the whole wave (see the header comment).  Correctness of the
// stopping rule: after ring r every cell of the box [lo-r, hi+r] (clipped to the grid) has been scanned; a
// reference point outside it lies beyond one of the box's faces that is not a grid boundary, hence at least
// the lane's distance to the nearest such face away (less `eps` for the rounding of the cell assignment).
template <int K>
__device__ __forceinline__ void knn_grid_scan(const GridHdr *__restrict__ hdr, const int *__restrict__ start, const float4 *__restrict__ sorted,
                                              float qx, float qy, float qz, float (&bd)[K], int (&bi)[K], bool settled)
{
    // `settled` lanes already hold their exact answer (knn_lane_ring1): they ride along, but neither shape the box nor keep the wave going
    const float ox = hdr->ox, oy = hdr->oy, oz = hdr->oz, h = hdr->h, inv_h = hdr->inv_h, eps = hdr->eps;
    const int nx = hdr->nx, ny = hdr->ny, nz = hdr->nz;
    const GridDims g{nx, ny, nz};
    const int cx = cell_coord(qx, ox, inv_h, nx), cy = cell_coord(qy, oy, inv_h, ny), cz = cell_coord(qz, oz, inv_h, nz);
    const int big = 0x3fffffff;
    const int lx = wave_mini(settled ? big : cx), hx = wave_maxi(settled ? -big : cx), ly = wave_mini(settled ? big : cy), hy = wave_maxi(settled ? -big : cy),
              lz = wave_mini(settled ? big : cz), hz = wave_maxi(settled ? -big : cz);
    scan_box<K>(start, sorted, g, lx, hx, ly, hy, lz, hz, qx, qy, qz, bd, bi);
    for (int r = 0, rp = 0;;) {                  // box radius scanned so far (r) and before that (rp), in cells around [l, h]
        const int X0 = max(lx - r, 0), X1 = min(hx + r, nx - 1), Y0 = max(ly - r, 0), Y1 = min(hy + r, ny - 1),
                  Z0 = max(lz - r, 0), Z1 = min(hz + r, nz - 1);
        if (r > rp) {
            // the shell between radius rp and r as six disjoint slabs: two x slabs (whole y, z extent of the new box),
            // two y slabs (x extent of the old box), two z slabs (x and y extent of the old box)
            const int xi0 = max(lx - rp, 0), xi1 = min(hx + rp, nx - 1), yi0 = max(ly - rp, 0), yi1 = min(hy + rp, ny - 1);
            scan_box<K>(start, sorted, g, X0, min(lx - rp - 1, nx - 1), Y0, Y1, Z0, Z1, qx, qy, qz, bd, bi);
            scan_box<K>(start, sorted, g, max(hx + rp + 1, 0), X1, Y0, Y1, Z0, Z1, qx, qy, qz, bd, bi);
            scan_box<K>(start, sorted, g, xi0, xi1, Y0, min(ly - rp - 1, ny - 1), Z0, Z1, qx, qy, qz, bd, bi);
            scan_box<K>(start, sorted, g, xi0, xi1, max(hy + rp + 1, 0), Y1, Z0, Z1, qx, qy, qz, bd, bi);
            scan_box<K>(start, sorted, g, xi0, xi1, yi0, yi1, Z0, min(lz - rp - 1, nz - 1), qx, qy, qz, bd, bi);
            scan_box<K>(start, sorted, g, xi0, xi1, yi0, yi1, max(hz + rp + 1, 0), Z1, qx, qy, qz, bd, bi);
        }
        if (X0 == 0 && X1 == nx - 1 && Y0 == 0 && Y1 == ny - 1 && Z0 == 0 && Z1 == nz - 1) break;          // everything scanned
        float b = __builtin_inff();
        if (X0 > 0) b = fminf(b, qx - (ox + (float)X0 * h));
        if (X1 < nx - 1) b = fminf(b, (ox + (float)(X1 + 1) * h) - qx);
        if (Y0 > 0) b = fminf(b, qy - (oy + (float)Y0 * h));
        if (Y1 < ny - 1) b = fminf(b, (oy + (float)(Y1 + 1) * h) - qy);
        if (Z0 > 0) b = fminf(b, qz - (oz + (float)Z0 * h));
        if (Z1 < nz - 1) b = fminf(b, (oz + (float)(Z1 + 1) * h) - qz);
        b = fmaxf(b - eps, 0.f);
        const bool done = settled || bd[K - 1] < b * b;
        if (__all(done)) break;
        // next radius: grow by a quarter (at least one cell), but never beyond the radius that settles the worst unsettled
        // lane even if its K-th best does not improve any more
        const float need = done ? 0.f : sqrtf(bd[K - 1]) + 2.f * eps;                   // inf while fewer than K candidates seen
        const float worst = wave_maxf(need);
        const int cap = worst < 1e30f ? (int)fminf(ceilf(worst * inv_h) + 1.f, 1e6f) : 0x7fffffff;
        const int rn = __builtin_amdgcn_readfirstlane(max(r + 1, min(cap, r + max(1, r / 4))));
        rp = r; r = rn;
    }
}

// First look of an INCOHERENT wave: every lane scans the 3 x 3 x 3 cells around its own query (per-lane vector loads, divergent trip counts:
// ~1.5x the cost per candidate of the cooperative scan) and applies the stopping rule to that box.  Marching-cubes vertices arrive in the
// library's order -- rows of cells along the last axis, a few crossings per row -- so the 64 queries of a wave trace a whole contour of a slice
// and their common box holds hundreds of cells; but a vertex of the avatar is within a cell or two of the SMPL surface and settles here.
// Returns whether the lane's answer is final; the lanes that are not go on together (knn_grid_scan, which skips what is already listed).
template <int K>
__device__ __forceinline__ bool knn_lane_ring1(const GridHdr *__restrict__ hdr, const int *__restrict__ start, const float4 *__restrict__ sorted,
                                               float qx, float qy, float qz, float (&bd)[K], int (&bi)[K])
{
    const float ox = hdr->ox, oy = hdr->oy, oz = hdr->oz, h = hdr->h, inv_h = hdr->inv_h, eps = hdr->eps;
    const int nx = hdr->nx, ny = hdr->ny, nz = hdr->nz;
    const int cx = cell_coord(qx, ox, inv_h, nx), cy = cell_coord(qy, oy, inv_h, ny), cz = cell_coord(qz, oz, inv_h, nz);
    const int X0 = max(cx - 1, 0), X1 = min(cx + 1, nx - 1), Y0 = max(cy - 1, 0), Y1 = min(cy + 1, ny - 1), Z0 = max(cz - 1, 0), Z1 = min(cz + 1, nz - 1);
    for (int x = X0; x <= X1; ++x)
        for (int y = Y0; y <= Y1; ++y) {
            const int c0 = (x * ny + y) * nz;
            const int s = start[c0 + Z0], e = start[c0 + Z1 + 1];               // cells are stored z fastest: one contiguous range per column
            for (int j = s; j < e; ++j) {
                const float4 c = sorted[j];
                const float d = cand_d2(c, qx, qy, qz);
                if (d <= bd[K - 1]) knn_insert_lex<K>(d, __float_as_int(c.w), bd, bi);
            }
        }
    if (X0 == 0 && X1 == nx - 1 && Y0 == 0 && Y1 == ny - 1 && Z0 == 0 && Z1 == nz - 1) return true;
    float b = __builtin_inff();
    if (X0 > 0) b = fminf(b, qx - (ox + (float)X0 * h));
    if (X1 < nx - 1) b = fminf(b, (ox + (float)(X1 + 1) * h) - qx);
    if (Y0 > 0) b = fminf(b, qy - (oy + (float)Y0 * h));
    if (Y1 < ny - 1) b = fminf(b, (oy + (float)(Y1 + 1) * h) - qy);
    if (Z0 > 0) b = fminf(b, qz - (oz + (float)Z0 * h));
    if (Z1 < nz - 1) b = fminf(b, (oz + (float)(Z1 + 1) * h) - qz);
    b = fmaxf(b - eps, 0.f);
    return bd[K - 1] < b * b;
}

struct GridView { const GridHdr *hdr; const int *start; const float4 *sorted; int lane_box; };   // hdr == nullptr: brute force

// ---- per-cell candidate lists of a BOUND reference set (avc_lbs_prepare: the canonical SMPL vertices of a sequence) -------------------------------
// The vertices a frame's marching-cubes output is skinned with are the same 6890 points for the whole sequence (main.py:335), and the queries lie on
// the body's surface.  The space within the reach (avc_set_option "lbs_reach_mm") of the vertices is cut into cells of edge h (2 cm); for the cell with centre c the list holds every
// vertex v with |v - c| <= d_K(c) + diag(cell) (+ a rounding allowance), d_K(c) the distance of c's K-th nearest vertex.  That list contains the K nearest
// of EVERY point q of the cell: |q - c| <= diag / 2, so d_K(q) <= d_K(c) + diag / 2 (c's K nearest are that close to q), and a vertex among q's K nearest
// -- ties included -- has |v - c| <= |v - q| + |q - c| <= d_K(c) + diag.  A lane therefore reads ONE range and scans some 40 - 100 candidates with the
// same cand_d2 / (distance, index) insertion as the grid search: same bits as the exhaustive scan, without the dependent chain of 27 cell ranges of
// knn_lane_ring1 (round 4: 317 us for the 190 k vertices of a band frame, three waves per SIMD waiting on memory).  Cells farther than the reach (avc_set_option "lbs_reach_mm", 140 by default) from
// every vertex, and points outside the cells' box, have no list: their lanes go on with the grid search.
struct CandView { const int *cstart; const float4 *cand; float ox, oy, oz, inv_h; int nx, ny, nz; };      // cstart == nullptr: no lists
constexpr long long LIST_MAX_ENTRIES = 1ll << 28;             // 4 GiB of candidates: beyond it avc_lbs_prepare builds no lists (a body's 6890 vertices with lists over a whole volume: 1e8)
constexpr float LIST_CELL = 0.02f, LIST_MARGIN = 0.16f;       // (the reach -- how far from the vertices cells still get a list -- is Options::lbs_reach_mm)

template <int K>
__device__ __forceinline__ bool knn_lane_list(const CandView &cv, float qx, float qy, float qz, float (&bd)[K], int (&bi)[K])
{
    const float fx = (qx - cv.ox) * cv.inv_h, fy = (qy - cv.oy) * cv.inv_h, fz = (qz - cv.oz) * cv.inv_h;
    const bool inside = fx >= 0.f && fy >= 0.f && fz >= 0.f && fx < (float)cv.nx && fy < (float)cv.ny && fz < (float)cv.nz;     // (false for NaN)
    const int c = inside ? ((int)fx * cv.ny + (int)fy) * cv.nz + (int)fz : 0;
    const int s = cv.cstart[c], e = inside ? cv.cstart[c + 1] : s;
    // (Per lane: a wave-level rule -- lists only when every lane has one -- was tried for the dense stress frame, whose surface fills the volume; it left
    // that frame's LBS unchanged and sent 86 % of a band frame's waves, each with a lane or two at the band's edge, back to the search: 165 -> 326 us.)
    if (e <= s) return false;
    for (int j = s; j < e; j += 4) {                       // four loads in flight; the tail repeats the last candidate (a listed index is skipped on insertion)
        float4 c4[4]; float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) c4[u] = cv.cand[min(j + u, e - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = cand_d2(c4[u], qx, qy, qz);
        const float m = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
        if (m <= bd[K - 1]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) knn_insert_lex<K>(d[u], __float_as_int(c4[u].w), bd, bi);
        }
    }
    return true;
}
template <int K, bool GRID>
__device__ __forceinline__ void knn_any(const float *__restrict__ ref, int nr, const GridView &g, float qx, float qy, float qz,
                                        float (&bd)[K], int (&bi)[K], const CandView &cv = CandView{})
{
    if constexpr (GRID) {
        const GridHdr *hdr = g.hdr;
#pragma unroll
        for (int k = 0; k < K; ++k) { bd[k] = __builtin_inff(); bi[k] = 0x7fffffff; }
        bool settled = false;
        if (cv.cstart) {                                   // a bound reference set: one candidate list per lane (knn_lane_list)
            settled = knn_lane_list<K>(cv, qx, qy, qz, bd, bi);
            if (__all(settled)) return;
        }
        const int cx = cell_coord(qx, hdr->ox, hdr->inv_h, hdr->nx), cy = cell_coord(qy, hdr->oy, hdr->inv_h, hdr->ny),
                  cz = cell_coord(qz, hdr->oz, hdr->inv_h, hdr->nz);
        const int big = 0x3fffffff;                        // the lanes that still search shape the box
        const int bx = wave_maxi(settled ? -big : cx) - wave_mini(settled ? big : cx) + 1, by = wave_maxi(settled ? -big : cy) - wave_mini(settled ? big : cy) + 1,
                  bz = wave_maxi(settled ? -big : cz) - wave_mini(settled ? big : cz) + 1;
        if ((long long)(bx + 2) * (by + 2) * (bz + 2) > g.lane_box && !settled) settled = knn_lane_ring1<K>(hdr, g.start, g.sorted, qx, qy, qz, bd, bi);
        if (__all(settled)) return;
        knn_grid_scan<K>(hdr, g.start, g.sorted, qx, qy, qz, bd, bi, settled);
    } else {
        __shared__ float4 lds[REF_TILE];
        knn_scan<K>(ref, nr, qx, qy, qz, bd, bi, lds);
    }
}

template <int K, bool GRID>
__global__ __launch_bounds__(256) void knn_kernel(const float *__restrict__ q, int64_t nq, const float *__restrict__ ref, int nr, GridView g,
                                                  float *__restrict__ d2, int64_t *__restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ii = i < nq ? i : nq - 1;
    float bd[K]; int bi[K];
    knn_any<K, GRID>(ref, nr, g, q[3 * ii], q[3 * ii + 1], q[3 * ii + 2], bd, bi);
    if (i < nq) {
#pragma unroll
        for (int k = 0; k < K; ++k) { if (d2) d2[i * K + k] = bd[k]; if (idx) idx[i * K + k] = bi[k]; }
    }
}

template <bool GRID>
__global__ __launch_bounds__(256) void lbs_kernel(const float *__restrict__ pts, int64_t n, const float *__restrict__ cano_v,
                                                  const float *__restrict__ skin_w, int nv, GridView g, CandView cv, float *__restrict__ lbs)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ii = i < n ? i : n - 1;
    float bd[4]; int bi[4];
    knn_any<4, GRID>(cano_v, nv, g, pts[3 * ii], pts[3 * ii + 1], pts[3 * ii + 2], bd, bi, cv);
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
        for (int r = 0; r < 4; ++r) { settle(M[4 * r], M[4 * r + 1], M[4 * r + 2], M[4 * r + 3]); o[r] = make_float4(M[4 * r], M[4 * r + 1], M[4 * r + 2], M[4 * r + 3]); }
    }
    if (pts) {   // live = M[:3,:3] p + M[:3,3]   (smpl_util.py:69)
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        float o[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = M[4 * r] * x + M[4 * r + 1] * y + M[4 * r + 2] * z + M[4 * r + 3];
        settle(o[0], o[1], o[2]);                // store_settle.h
        po[3 * i] = o[0]; po[3 * i + 1] = o[1]; po[3 * i + 2] = o[2];
    }
    if (nrm) {   // live_normals = M[:3,:3] n, no renormalisation   (smpl_util.py:80)
        const float x = nrm[3 * i], y = nrm[3 * i + 1], z = nrm[3 * i + 2];
        float o[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = M[4 * r] * x + M[4 * r + 1] * y + M[4 * r + 2] * z;
        settle(o[0], o[1], o[2]);
        no[3 * i] = o[0]; no[3 * i + 1] = o[1]; no[3 * i + 2] = o[2];
    }
}

// calculate_lbs + skinning + skinning_normal of main.py:385-389 in ONE launch (avc_lbs_skin_bound; lbs_skin_grid_kernel / lbs_skin_scan_kernel below): the blend weights of a vertex go from lbs_kernel's registers
// straight into skinning_kernel's sums -- the same operations in the same order as the two kernels, so the same bits -- and the (n,24) weights, which the frame
// loop never looks at, are written only on request: 112 B per vertex through HBM instead of 412 (lbs written once and read twice, points read twice).
template <bool GRID>
__device__ __forceinline__ void lbs_skin_body(const float *__restrict__ pts, const float *__restrict__ nrm, int64_t n, const float *__restrict__ cano_v,
                                              const float *__restrict__ skin_w, int nv, const GridView &g, const CandView &cv, const float *__restrict__ jm,
                                              float *__restrict__ lbs, float *__restrict__ po, float *__restrict__ no, float *__restrict__ mo)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ii = i < n ? i : n - 1;
    float bd[4]; int bi[4];
    knn_any<4, GRID>(cano_v, nv, g, pts[3 * ii], pts[3 * ii + 1], pts[3 * ii + 2], bd, bi, cv);
    if (i >= n) return;
    const float denom = (float)(2 * 0.05 * 0.05);                     // lbs_kernel, line for line
    float w[4], sum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { w[k] = expf(-bd[k] / denom); sum += w[k]; }
    sum += 1e-16f;
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] /= sum;
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
    if (lbs) {
        float4 *out = reinterpret_cast<float4 *>(lbs + (size_t)i * 24);
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) out[jj] = make_float4(acc[4 * jj], acc[4 * jj + 1], acc[4 * jj + 2], acc[4 * jj + 3]);
    }
    float M[16];                                                      // skinning_kernel, line for line
#pragma unroll
    for (int e = 0; e < 16; ++e) M[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 24; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) M[e] += acc[j] * jm[j * 16 + e];           // wave-uniform addresses: the joint matrices come through scalar loads, sixteen SGPRs a row
    if (mo) {
        float4 *o = reinterpret_cast<float4 *>(mo + (size_t)i * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) { settle(M[4 * r], M[4 * r + 1], M[4 * r + 2], M[4 * r + 3]); o[r] = make_float4(M[4 * r], M[4 * r + 1], M[4 * r + 2], M[4 * r + 3]); }
    }
    if (po) {                                                         // (the point is read again rather than kept alive across the search)
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        float o[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = M[4 * r] * x + M[4 * r + 1] * y + M[4 * r + 2] * z + M[4 * r + 3];
        settle(o[0], o[1], o[2]);                                     // store_settle.h
        po[3 * i] = o[0]; po[3 * i + 1] = o[1]; po[3 * i + 2] = o[2];
    }
    if (nrm) {
        const float x = nrm[3 * i], y = nrm[3 * i + 1], z = nrm[3 * i + 2];
        float o[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = M[4 * r] * x + M[4 * r + 1] * y + M[4 * r + 2] * z;
        settle(o[0], o[1], o[2]);
        no[3 * i] = o[0]; no[3 * i + 1] = o[1]; no[3 * i + 2] = o[2];
    }
}
// The search is a chain of dependent loads: it lives on waves in flight.  lbs_kernel fits 76 registers (six waves per SIMD); the fused body's tail -- 24 blend
// weights, 16 matrix sums, the skin-weight rows in flight -- would set the whole kernel's budget at 94 - 142 (3 - 5 waves: the search ran 1.5x slower than
// lbs_kernel's on the dense frame's 1.9 M vertices), so the grid form is held to six waves (80 registers, four dwords spilled once in the tail).  The exhaustive
// form is bound by its LDS tile.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void lbs_skin_grid_kernel(
    const float *__restrict__ pts, const float *__restrict__ nrm, int64_t n, const float *__restrict__ cano_v, const float *__restrict__ skin_w, int nv, GridView g, CandView cv,
    const float *__restrict__ jm, float *__restrict__ lbs, float *__restrict__ po, float *__restrict__ no, float *__restrict__ mo)
{
    lbs_skin_body<true>(pts, nrm, n, cano_v, skin_w, nv, g, cv, jm, lbs, po, no, mo);
}
__global__ __launch_bounds__(256) void lbs_skin_scan_kernel(
    const float *__restrict__ pts, const float *__restrict__ nrm, int64_t n, const float *__restrict__ cano_v, const float *__restrict__ skin_w, int nv, GridView g, CandView cv,
    const float *__restrict__ jm, float *__restrict__ lbs, float *__restrict__ po, float *__restrict__ no, float *__restrict__ mo)
{
    lbs_skin_body<false>(pts, nrm, n, cano_v, skin_w, nv, g, cv, jm, lbs, po, no, mo);
}

// near[i] = 0 when a reference point lies closer than sqrt(thr2) to query i, +inf otherwise -- the only thing the colour path wants of
// knn_points(wpts, cano_smpl_vertices, K=1) (arch_avatar.py:208-209: near_flag = d2 < 0.08^2).  The same squared distance (cand_d2) against the
// same threshold as a comparison of the K = 1 result gives, so the flags are the exact search's; but only the cells the ball touches are
// visited and the first hit ends the search (the exact K = 1 kernel spends 3.3 ms on the 12.8 M samples of 200 k rays, this one a third).
__global__ __launch_bounds__(256) void near_flag_kernel(const float *__restrict__ q, int64_t nq, GridView g, float thr2, float radius, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    const GridHdr H = *g.hdr;
    const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
    const float r = radius + H.eps;
    const int x0 = cell_coord(qx - r, H.ox, H.inv_h, H.nx), x1 = cell_coord(qx + r, H.ox, H.inv_h, H.nx);
    const int y0 = cell_coord(qy - r, H.oy, H.inv_h, H.ny), y1 = cell_coord(qy + r, H.oy, H.inv_h, H.ny);
    const int z0 = cell_coord(qz - r, H.oz, H.inv_h, H.nz), z1 = cell_coord(qz + r, H.oz, H.inv_h, H.nz);
    bool hit = false;
    for (int cx = x0; cx <= x1 && !hit; ++cx)
        for (int cy = y0; cy <= y1 && !hit; ++cy) {
            const int row = (cx * H.ny + cy) * H.nz;
            const int b = g.start[row + z0], e = g.start[row + z1 + 1];          // z runs fastest: the cells z0..z1 of this (x, y) are one range
            for (int k = b; k < e; ++k)
                if (cand_d2(g.sorted[k], qx, qy, qz) < thr2) { hit = true; break; }
        }
    out[i] = hit ? 0.0f : __builtin_inff();
}

// ---- building the candidate lists (once per bound reference set) ------------------------------------------------------------------------------
struct CandGrid { float ox, oy, oz, h, inv_h; int nx, ny, nz; };

// radius[c] = d_K(centre of cell c) + diag + allowance, or -1 when the centre is farther than `reach` from its K-th nearest vertex (no list)
template <int K>
__global__ __launch_bounds__(256) void cand_radius_kernel(CandGrid cg, const float *__restrict__ ref, int nr, GridView g, float reach, float *__restrict__ radius)
{
    const int ncell = cg.nx * cg.ny * cg.nz;
    const int c = blockIdx.x * 256 + threadIdx.x, cc = min(c, ncell - 1);
    const int x = cc / (cg.ny * cg.nz), y = (cc / cg.nz) % cg.ny, z = cc % cg.nz;
    const float qx = cg.ox + ((float)x + 0.5f) * cg.h, qy = cg.oy + ((float)y + 0.5f) * cg.h, qz = cg.oz + ((float)z + 0.5f) * cg.h;
    float bd[K]; int bi[K];
    knn_any<K, true>(ref, nr, g, qx, qy, qz, bd, bi);
    if (c >= ncell) return;
    const float dk = sqrtf(bd[K - 1]);
    // allowance: the rounding of the cell assignment and of the centre (a few ulp of coordinates of order 1) -- 1e-5 m is a thousand times that
    radius[c] = dk <= reach ? (dk + 1.7320508f * cg.h) * 1.00001f + 1e-5f : -1.0f;
}

// FILL == false: count[c] = number of vertices within radius[c] of the centre of cell c;  FILL == true: writes them, in index order, at cstart[c]
template <bool FILL>
__global__ __launch_bounds__(256) void cand_list_kernel(CandGrid cg, const float *__restrict__ ref, int nr, const float *__restrict__ radius,
                                                        int *__restrict__ count, const int *__restrict__ cstart, float4 *__restrict__ cand)
{
    __shared__ float4 lds[REF_TILE];
    const int ncell = cg.nx * cg.ny * cg.nz;
    const int c = blockIdx.x * 256 + threadIdx.x, cc = min(c, ncell - 1);
    const int x = cc / (cg.ny * cg.nz), y = (cc / cg.nz) % cg.ny, z = cc % cg.nz;
    const float qx = cg.ox + ((float)x + 0.5f) * cg.h, qy = cg.oy + ((float)y + 0.5f) * cg.h, qz = cg.oz + ((float)z + 0.5f) * cg.h;
    const float r = c < ncell ? radius[cc] : -1.0f, r2 = r * r;
    // a workgroup of far cells has nothing to list
    if (!__syncthreads_or(r >= 0.f)) { if (!FILL && c < ncell) count[c] = 0; return; }
    int n = 0, w = FILL && c < ncell ? cstart[cc] : 0;
    for (int r0 = 0; r0 < nr; r0 += REF_TILE) {
        const int cnt = min(REF_TILE, nr - r0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += 256) lds[i] = make_float4(ref[(size_t)(r0 + i) * 3], ref[(size_t)(r0 + i) * 3 + 1], ref[(size_t)(r0 + i) * 3 + 2], __int_as_float(r0 + i));
        __syncthreads();
        if (r >= 0.f)
            for (int i = 0; i < cnt; ++i) {
                const float4 v = lds[i];
                if (cand_d2(v, qx, qy, qz) <= r2) {
                    if constexpr (FILL) cand[w++] = v;
                    else ++n;
                }
            }
    }
    if (!FILL && c < ncell) count[c] = n;
}

// exclusive scan of count[0 .. n) in place into cstart[0 .. n] (one workgroup; once per sequence).  The sums are 64-bit: a[n + 2], a[n + 3] hold the exact total
// (low, high word) whatever happens to the 32-bit offsets -- a degenerate vertex set (thousands of coincident points) lists every vertex in every cell, and
// avc_lbs_prepare then does without lists instead of asking for the memory.
__global__ __launch_bounds__(1024) void cand_scan_kernel(int *__restrict__ a, int n)
{
    // batches of 16384 counts through LDS (coalesced in and out; a thread sums its 16 consecutive ones, the block scans the thread sums): 0.3 ms for the 3.4 M
    // cells of a whole volume (a thread walking its own contiguous share of the array, uncoalesced, took 6 ms)
    constexpr int PER = 16, BATCH = 1024 * PER;
    __shared__ int buf[BATCH + BATCH / 16];                  // padded: a thread's 16 consecutive words start 17 words apart
    __shared__ long long wsum[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    auto at = [](int i) { return i + (i >> 4); };
    long long carry = 0;
    for (int b0 = 0; b0 < n; b0 += BATCH) {
        for (int j = 0; j < PER; ++j) { const int i = j * 1024 + threadIdx.x; buf[at(i)] = b0 + i < n ? a[b0 + i] : 0; }
        __syncthreads();
        int v[PER]; long long sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { v[j] = buf[at(threadIdx.x * PER + j)]; sum += v[j]; }
        long long incl = sum;
        for (int o = 1; o < 64; o <<= 1) { const long long t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        long long base = 0, all = 0;
        for (int k = 0; k < 16; ++k) { if (k < w) base += wsum[k]; all += wsum[k]; }
        long long run = carry + base + incl - sum;
#pragma unroll
        for (int j = 0; j < PER; ++j) { buf[at(threadIdx.x * PER + j)] = (int)run; run += v[j]; }
        carry += all;
        __syncthreads();
        for (int j = 0; j < PER; ++j) { const int i = j * 1024 + threadIdx.x; if (b0 + i < n) a[b0 + i] = buf[at(i)]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { a[n] = (int)carry; a[n + 2] = (int)(carry & 0xffffffffll); a[n + 3] = (int)(carry >> 32); }
}

}  // namespace

// Builds the grid over `ref` in the context's scratch (stream-ordered; no host synchronisation).
static int make_grid(avc_ctx *ctx, const float *ref, int32_t nr, int64_t nq, GridView &g, hipStream_t s, void **scratch = nullptr, size_t *scratch_bytes = nullptr)
{
    if (!scratch) { scratch = &ctx->knn_scratch; scratch_bytes = &ctx->knn_scratch_bytes; }       // (a bound reference set keeps its grid in memory of its own)
    g = GridView{nullptr, nullptr, nullptr, 0};
    if (nr < GRID_MIN_REFS || ctx->opt.knn_search == 3) return AVC_OK;
    // cells per axis: about one occupied cell per few reference points for surface-like sets (6890 -> 32, 1e6 -> 128)
    const int axis = std::min(GRID_MAX_AXIS, std::max(8, (int)(1.7 * cbrt((double)nr))));
    const size_t ncell = (size_t)axis * axis * axis;
    const size_t cells = (ncell + 64 + 1) & ~(size_t)1;      // even: the two int arrays together stay a multiple of 16 bytes, so `sorted` (float4) is aligned
    const size_t bytes = 256 + 2 * sizeof(int) * cells + sizeof(float4) * ((size_t)nr + 8);
    if (*scratch_bytes < bytes) {
        if (*scratch) AVC_HIP(hipFree(*scratch));
        *scratch = nullptr; *scratch_bytes = 0;
        AVC_HIP(hipMalloc(scratch, bytes));
        *scratch_bytes = bytes;
    }
    char *base = static_cast<char *>(*scratch);
    GridHdr *hdr = reinterpret_cast<GridHdr *>(base);
    unsigned *keys = reinterpret_cast<unsigned *>(base + 128);
    int *start = reinterpret_cast<int *>(base + 256);
    int *cursor = start + cells;
    float4 *sorted = reinterpret_cast<float4 *>(cursor + cells);
    AVC_HIP(hipMemsetAsync(keys, 0xff, 3 * sizeof(unsigned), s));                 // running minima start at the largest key
    AVC_HIP(hipMemsetAsync(keys + 3, 0x00, 3 * sizeof(unsigned), s));
    AVC_HIP(hipMemsetAsync(start, 0, sizeof(int) * (ncell + 1), s));
    const dim3 blocks((unsigned)std::min<int64_t>(1024, (nr + 255) / 256)), threads(256);
    hipLaunchKernelGGL(grid_bbox_kernel, blocks, threads, 0, s, ref, nr, keys);
    hipLaunchKernelGGL(grid_header_kernel, dim3(1), dim3(1), 0, s, keys, axis, hdr);
    hipLaunchKernelGGL(grid_count_kernel, blocks, threads, 0, s, ref, nr, hdr, start);
    hipLaunchKernelGGL(grid_scan_kernel, dim3(1), dim3(1024), 0, s, hdr, start, cursor);
    hipLaunchKernelGGL(grid_scatter_kernel, blocks, threads, 0, s, ref, nr, hdr, cursor, sorted);
    AVC_HIP(hipGetLastError());
    // avc_set_option "knn_search" (debugging / test switch): 1 or 2 force the per-lane or the cooperative search on every wave
    g = GridView{hdr, start, sorted, ctx->opt.knn_search == 1 ? 0 : (ctx->opt.knn_search == 2 ? 0x7fffffff : LANE_BOX)};
    return AVC_OK;
}

int knn(avc_ctx *ctx, const float *q, int64_t nq, const float *ref, int32_t nr, int K, float *d2, int64_t *idx, hipStream_t s)
{
    if (nq == 0) return AVC_OK;
    GridView g;
    if (int rc = make_grid(ctx, ref, nr, nq, g, s)) return rc;
    const dim3 grid((unsigned)((nq + 255) / 256)), block(256);
    switch (K) {
#define CASE(k) case k: if (g.hdr) hipLaunchKernelGGL((knn_kernel<k, true>), grid, block, 0, s, q, nq, ref, nr, g, d2, idx); \
                       else hipLaunchKernelGGL((knn_kernel<k, false>), grid, block, 0, s, q, nq, ref, nr, g, d2, idx); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
        default: set_error("avc_knn: unsupported K %d", K); return AVC_ERR_ARG;
    }
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

int near_flags(avc_ctx *ctx, const float *q, int64_t nq, const float *ref, int32_t nr, float thr2, float *out, hipStream_t s)
{
    if (nq == 0) return AVC_OK;
    GridView g;
    if (int rc = make_grid(ctx, ref, nr, nq, g, s)) return rc;
    if (!g.hdr) return knn(ctx, q, nq, ref, nr, 1, out, nullptr, s);          // no grid (a handful of reference points): the exact distances serve as flags
    hipLaunchKernelGGL(near_flag_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, q, nq, g, thr2, std::sqrt(thr2) * 1.0001f, out);
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

int calculate_lbs(avc_ctx *ctx, const float *pts, int64_t n, const float *cano_v, const float *skin_w, int32_t nv, float *lbs, hipStream_t s)
{
    if (n == 0) return AVC_OK;
    GridView g;
    if (int rc = make_grid(ctx, cano_v, nv, n, g, s)) return rc;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (g.hdr) hipLaunchKernelGGL(lbs_kernel<true>, grid, block, 0, s, pts, n, cano_v, skin_w, nv, g, CandView{}, lbs);
    else hipLaunchKernelGGL(lbs_kernel<false>, grid, block, 0, s, pts, n, cano_v, skin_w, nv, g, CandView{}, lbs);
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

// ---- a bound reference set: SmplUtil.set_cano_smpl_vertices (utils/smpl_util.py:21) -----------------------------------------------------------------
struct LbsBound {
    float *ref = nullptr; int32_t nr = 0;                 // the context's own copy of the vertices (nr x 3)
    void *grid_mem = nullptr; size_t grid_bytes = 0;      // the uniform grid over them (GridView: what lanes without a list search)
    GridView grid{nullptr, nullptr, nullptr, 0};
    void *list_mem = nullptr; size_t list_bytes = 0;      // [radius (ncell floats) | cstart (ncell + 1 ints)]
    float4 *cand = nullptr; size_t cand_cap = 0;
    CandView cv{};
    int64_t ncand = 0; int ncell = 0, nlisted = 0;
};

void release_lbs_bound(avc_ctx *ctx)
{
    LbsBound *b = static_cast<LbsBound *>(ctx->lbs_bound);
    if (!b) return;
    if (b->ref) hipFree(b->ref);
    if (b->grid_mem) hipFree(b->grid_mem);
    if (b->list_mem) hipFree(b->list_mem);
    if (b->cand) hipFree(b->cand);
    delete b;
    ctx->lbs_bound = nullptr;
}

int lbs_prepare(avc_ctx *ctx, const float *cano_v, int32_t nv, hipStream_t s)
{
    if (!ctx->lbs_bound) ctx->lbs_bound = new LbsBound();
    LbsBound *b = static_cast<LbsBound *>(ctx->lbs_bound);
    b->cv = CandView{}; b->grid = GridView{nullptr, nullptr, nullptr, 0};
    if (b->nr < nv) { if (b->ref) AVC_HIP(hipFree(b->ref)); b->ref = nullptr; AVC_HIP(hipMalloc((void **)&b->ref, sizeof(float) * 3 * (size_t)nv)); }
    b->nr = nv;
    std::vector<float> host((size_t)nv * 3);
    AVC_HIP(hipMemcpyAsync(b->ref, cano_v, sizeof(float) * 3 * (size_t)nv, hipMemcpyDeviceToDevice, s));
    AVC_HIP(hipMemcpyAsync(host.data(), cano_v, sizeof(float) * 3 * (size_t)nv, hipMemcpyDeviceToHost, s));
    AVC_HIP(hipStreamSynchronize(s));                     // once per sequence: the box of the cells is sized on the host
    if (int rc = make_grid(ctx, b->ref, nv, nv, b->grid, s, &b->grid_mem, &b->grid_bytes)) return rc;
    if (!b->grid.hdr || ctx->opt.knn_search != 0 || ctx->opt.lbs_reach_mm <= 0) { AVC_HIP(hipStreamSynchronize(s)); return AVC_OK; }      // a handful of vertices, or a forced search path (tests): no lists, the generic search serves
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int32_t i = 0; i < nv; ++i)
        for (int a = 0; a < 3; ++a) { const float v = host[(size_t)i * 3 + a]; if (v < mn[a]) mn[a] = v; if (v > mx[a]) mx[a] = v; }
    for (int a = 0; a < 3; ++a) AVC_REQUIRE(std::isfinite(mn[a]) && std::isfinite(mx[a]), AVC_ERR_ARG, "avc_lbs_prepare: the vertices are not finite");
    float h = LIST_CELL;
    const float reach = 1e-3f * (float)ctx->opt.lbs_reach_mm, margin = std::max(LIST_MARGIN, reach + 0.04f);
    CandGrid cg{};
    for (;; h *= 1.26f) {                                 // (a reference set far larger than a body: coarser cells rather than more than 4 M of them)
        cg = CandGrid{mn[0] - margin, mn[1] - margin, mn[2] - margin, h, 1.0f / h,
                      (int)std::ceil((mx[0] - mn[0] + 2 * margin) / h), (int)std::ceil((mx[1] - mn[1] + 2 * margin) / h), (int)std::ceil((mx[2] - mn[2] + 2 * margin) / h)};
        if ((double)cg.nx * cg.ny * cg.nz <= 4.0e6) break;
    }
    const int ncell = cg.nx * cg.ny * cg.nz;
    const size_t lbytes = sizeof(float) * ((size_t)ncell + 4) + sizeof(int) * ((size_t)ncell + 4);
    if (b->list_bytes < lbytes) {
        if (b->list_mem) AVC_HIP(hipFree(b->list_mem));
        b->list_mem = nullptr; b->list_bytes = 0;
        AVC_HIP(hipMalloc(&b->list_mem, lbytes));
        b->list_bytes = lbytes;
    }
    float *radius = static_cast<float *>(b->list_mem);
    int *cstart = reinterpret_cast<int *>(radius + ncell + 4);
    const dim3 blocks((unsigned)((ncell + 255) / 256)), threads(256);
    hipLaunchKernelGGL(cand_radius_kernel<4>, blocks, threads, 0, s, cg, b->ref, nv, b->grid, reach, radius);
    hipLaunchKernelGGL(cand_list_kernel<false>, blocks, threads, 0, s, cg, b->ref, nv, radius, cstart, (const int *)nullptr, (float4 *)nullptr);
    hipLaunchKernelGGL(cand_scan_kernel, dim3(1), dim3(1024), 0, s, cstart, ncell);
    int tail[4] = {0, 0, 0, 0};
    AVC_HIP(hipMemcpyAsync(tail, cstart + ncell, sizeof tail, hipMemcpyDeviceToHost, s));
    AVC_HIP(hipStreamSynchronize(s));
    AVC_HIP(hipGetLastError());
    const long long total64 = (long long)(unsigned)tail[2] | ((long long)tail[3] << 32);
    if (total64 > LIST_MAX_ENTRIES) {                     // a degenerate vertex set: every cell would list (nearly) every vertex -- the search serves, same results
        b->ncand = 0; b->ncell = ncell;
        return AVC_OK;
    }
    const int total = (int)total64;
    if (b->cand_cap < (size_t)total + 8) {
        if (b->cand) AVC_HIP(hipFree(b->cand));
        b->cand = nullptr; b->cand_cap = 0;
        AVC_HIP(hipMalloc((void **)&b->cand, sizeof(float4) * ((size_t)total + 8)));
        b->cand_cap = (size_t)total + 8;
    }
    hipLaunchKernelGGL(cand_list_kernel<true>, blocks, threads, 0, s, cg, b->ref, nv, radius, (int *)nullptr, cstart, b->cand);
    AVC_HIP(hipGetLastError());
    AVC_HIP(hipStreamSynchronize(s));                     // the tables belong to the context and are read on whatever stream calculate_lbs_bound is given: complete on return
    b->cv = CandView{cstart, b->cand, cg.ox, cg.oy, cg.oz, cg.inv_h, cg.nx, cg.ny, cg.nz};
    b->ncand = total; b->ncell = ncell;
    return AVC_OK;
}

int calculate_lbs_bound(avc_ctx *ctx, const float *pts, int64_t n, const float *skin_w, float *lbs, hipStream_t s)
{
    LbsBound *b = static_cast<LbsBound *>(ctx->lbs_bound);
    AVC_REQUIRE(b && b->nr >= 4, AVC_ERR_STATE, "Canonical smpl vertices are invalid!");      // smpl_util.py:31 (avc_lbs_prepare was not called)
    if (n == 0) return AVC_OK;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    GridView g = b->grid;
    if (g.hdr) g.lane_box = ctx->opt.knn_search == 1 ? 0 : (ctx->opt.knn_search == 2 ? 0x7fffffff : LANE_BOX);
    const bool brute = ctx->opt.knn_search == 3 || !g.hdr;
    if (!brute) hipLaunchKernelGGL(lbs_kernel<true>, grid, block, 0, s, pts, n, b->ref, skin_w, b->nr, g, ctx->opt.knn_search == 0 ? b->cv : CandView{}, lbs);
    else hipLaunchKernelGGL(lbs_kernel<false>, grid, block, 0, s, pts, n, b->ref, skin_w, b->nr, g, CandView{}, lbs);
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

int lbs_skin_bound(avc_ctx *ctx, const float *pts, const float *nrm, int64_t n, const float *skin_w, const float *jm, float *lbs, float *po, float *no, float *mo,
                   hipStream_t s)
{
    LbsBound *b = static_cast<LbsBound *>(ctx->lbs_bound);
    AVC_REQUIRE(b && b->nr >= 4, AVC_ERR_STATE, "Canonical smpl vertices are invalid!");      // smpl_util.py:31 (avc_lbs_prepare was not called)
    if (n == 0) return AVC_OK;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    GridView g = b->grid;
    if (g.hdr) g.lane_box = ctx->opt.knn_search == 1 ? 0 : (ctx->opt.knn_search == 2 ? 0x7fffffff : LANE_BOX);
    const bool brute = ctx->opt.knn_search == 3 || !g.hdr;
    if (!brute) hipLaunchKernelGGL(lbs_skin_grid_kernel, grid, block, 0, s, pts, nrm, n, b->ref, skin_w, b->nr, g, ctx->opt.knn_search == 0 ? b->cv : CandView{}, jm, lbs, po, no, mo);
    else hipLaunchKernelGGL(lbs_skin_scan_kernel, grid, block, 0, s, pts, nrm, n, b->ref, skin_w, b->nr, g, CandView{}, jm, lbs, po, no, mo);
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

int lbs_bound_stats(avc_ctx *ctx, int64_t out[4])
{
    LbsBound *b = static_cast<LbsBound *>(ctx->lbs_bound);
    out[0] = b ? b->nr : 0; out[1] = b ? b->ncell : 0; out[2] = b ? b->ncand : 0; out[3] = b && b->cv.cstart ? 1 : 0;
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
