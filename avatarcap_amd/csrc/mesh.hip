// Device-resident replacement of utils/recon_util.py:51-70 (recon_mesh):
//   marching cubes (the reference's skimage.measure.marching_cubes call, :64)  +  vertex offset/scale (:62,65)
//   + Sobel-gradient normals sampled trilinearly at the vertices (:9-48, :66-68) + face flip (:69)
// with the occupancy volume never leaving HBM (the reference copies it to the host, runs single-thread
// Cython, and copies the vertices back).
//
// Marching cubes = scikit-image's Lewiner implementation (method='lewiner', the library default), restated:
// 15 base cases, face tests, interior tests, sub-case tilings with the optional centre vertex, the library's
// 1/(eps+|v|) edge interpolation in double, AND the library's output order, so that vertices, faces and their
// numbering are identical to what the reference's call returns (oracle/mc_oracle.c is the sequential
// restatement, pinned against the real library: tests/golden/mc_golden.npz).
//
// The library is sequential: cells in (axis0, axis1, axis2) order, a vertex is appended the first time a
// triangle refers to its grid edge.  Parallel form of the same order:
//   * every cut edge of a cell appears in the cell's tiling, so the vertex of a grid edge is created by the
//     FIRST cell in traversal order among the (up to 4) cells around it -- a function of the cell's position
//     only (creator_mask);  inside a cell new vertices are numbered by first appearance in its triangle list;
//   pass A1 classify: the one streaming read of the volume: a thread takes four consecutive cells (their four corner rows as 16-byte loads, the fifth
//                     column from the next lane) and one comparison of 20 "above the level" bits tells that none is crossed (most are not).  Tiles
//                     (1024 consecutive grid points) with a crossed cell get their cells' cube indices stored (1 byte each) and a place in the list of
//                     crossed tiles; the volume is not walked again.  No tables, 32 registers: bound by the cache's request rate (four rows per cell row)
//   pass A2 count   : per crossed tile, the crossed cells compacted per wave (every lane takes one crossed cell at a time): cube index -> tiling-table row
//                     (the face / interior tests in fp64 read the 8 corner values of the few cells
//                     that need them), stored as 2 bytes per cell; #vertices created, #triangles, #crossed cells -> tile sums
//   scan            : exclusive scan of the tile sums (one workgroup, through LDS), totals and "fits the caller's capacity" left on the device
//   pass B  verts   : per crossed tile, in any order: the stored rows, in-tile scan (per-cell offsets in LDS), the crossed cells compacted per wave, number the new vertices, record their ids in the edge map (3 ints
//                     per grid point, sparsely written), write one 16-byte record per crossed cell {cell, table row, first face, centre-vertex id}
//                     and, into each new vertex's output slot, which (cell, edge) it is
//   pass B' eval    : one thread per vertex: position (the library's fp64 formula) + normal (64-tap stencil)
//   pass C  faces   : one thread per crossed cell: ids from the edge map -> triangles
// A2, B, B' and C take their sizes from the device: the six launches are enqueued back to back and the host waits once, at the end, for the counts.
// HBM-bound: the volume is read once (A1) + 3 B per cell of the crossed tiles + 24 B/vertex + 12 B/face written.
// The 18 KB of look-up tables live in LDS.  Face / interior tests and the interpolation run in fp64 exactly as
// the library's C code does (translation unit built with -ffp-contract=off).
#include "store_settle.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "avcap_internal.h"
#include "mc_tables.h"

namespace avc {
namespace {

constexpr int TILE = 1024;          // grid points per tile: 256 threads x 4 consecutive points
constexpr double LIB_EPS = 2.220446049250313e-16;   // the library's "FLT_EPSILON" = numpy.spacing(1.0)

struct McArgs {
    const float *vol;
    int n0, n1, n2;                 // volume shape (axis 0 slowest); the library calls the axes z, y, x
    int64_t N;
    float iso_f;
    double iso;
    int ntiles;
    const uint32_t *tables;         // mc::BLOB, device copy
};

__device__ __forceinline__ void load_tables(const uint32_t *__restrict__ g, uint32_t *lds)
{
    for (int i = threadIdx.x; i < mc::BLOB_WORDS; i += blockDim.x) lds[i] = g[i];
    __syncthreads();
}
__device__ __forceinline__ int tab_i8(const uint32_t *lds, int byte_off) { return (int)reinterpret_cast<const int8_t *>(lds)[byte_off]; }
__device__ __forceinline__ int tab_u8(const uint32_t *lds, int byte_off) { return (int)reinterpret_cast<const uint8_t *>(lds)[byte_off]; }
__device__ __forceinline__ int row_ntri(const uint32_t *lds, int row) { return tab_u8(lds, mc::OFF_ROWS + row * mc::ROW_BYTES); }
__device__ __forceinline__ unsigned row_mask(const uint32_t *lds, int row)
{
    return (unsigned)tab_u8(lds, mc::OFF_ROWS + row * mc::ROW_BYTES + 1) | ((unsigned)tab_u8(lds, mc::OFF_ROWS + row * mc::ROW_BYTES + 2) << 8);
}
__device__ __forceinline__ int row_edge(const uint32_t *lds, int row, int k)   // k-th edge id of the row's triangle list
{
    return (tab_u8(lds, mc::OFF_ROWS + row * mc::ROW_BYTES + 4 + (k >> 1)) >> (4 * (k & 1))) & 0xf;
}

// ---- the library's tests, in double (oracle/mc_oracle.c test_face / test_internal) ----
__device__ __forceinline__ double sel8(const double *v, int i)
{
    double r = v[0];
#pragma unroll
    for (int c = 1; c < 8; ++c) r = (i == c) ? v[c] : r;
    return r;
}

__device__ bool test_face(const double *v, int face)
{
    const int af = face < 0 ? -face : face;
    // corners A B C D of faces 1..6, 3 bits each
    const unsigned FC[6] = {0u | 4u << 3 | 5u << 6 | 1u << 9, 1u | 5u << 3 | 6u << 6 | 2u << 9, 2u | 6u << 3 | 7u << 6 | 3u << 9,
                            3u | 7u << 3 | 4u << 6 | 0u << 9, 0u | 3u << 3 | 2u << 6 | 1u << 9, 4u | 7u << 3 | 6u << 6 | 5u << 9};
    const unsigned pc = FC[af - 1];
    const double A = sel8(v, pc & 7), B = sel8(v, (pc >> 3) & 7), C = sel8(v, (pc >> 6) & 7), D = sel8(v, (pc >> 9) & 7);
    const double acbd = A * C - B * D;
    if (acbd > -LIB_EPS && acbd < LIB_EPS) return face >= 0;
    return (double)face * A * acbd >= 0;
}

__device__ bool test_internal(const double *v, int kase, int edge, int s)
{
    double t, At = 0, Bt = 0, Ct = 0, Dt = 0;
    if (kase == 4 || kase == 10) {
        const double a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
        const double b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
        t = -b / (2 * a + LIB_EPS);
        if (t < 0 || t > 1) return s > 0;
        At = v[0] + (v[4] - v[0]) * t;
        Bt = v[3] + (v[7] - v[3]) * t;
        Ct = v[2] + (v[6] - v[2]) * t;
        Dt = v[1] + (v[5] - v[1]) * t;
    } else {
        // reference edge -> corner indices {p, q, b0, b1, c0, c1, d0, d1}, 3 bits each
        const unsigned TI[12] = {
            0u | 1u << 3 | 3u << 6 | 2u << 9 | 7u << 12 | 6u << 15 | 4u << 18 | 5u << 21, 1u | 2u << 3 | 0u << 6 | 3u << 9 | 4u << 12 | 7u << 15 | 5u << 18 | 6u << 21,
            2u | 3u << 3 | 1u << 6 | 0u << 9 | 5u << 12 | 4u << 15 | 6u << 18 | 7u << 21, 3u | 0u << 3 | 2u << 6 | 1u << 9 | 6u << 12 | 5u << 15 | 7u << 18 | 4u << 21,
            4u | 5u << 3 | 7u << 6 | 6u << 9 | 3u << 12 | 2u << 15 | 0u << 18 | 1u << 21, 5u | 6u << 3 | 4u << 6 | 7u << 9 | 0u << 12 | 3u << 15 | 1u << 18 | 2u << 21,
            6u | 7u << 3 | 5u << 6 | 4u << 9 | 1u << 12 | 0u << 15 | 2u << 18 | 3u << 21, 7u | 4u << 3 | 6u << 6 | 5u << 9 | 2u << 12 | 1u << 15 | 3u << 18 | 0u << 21,
            0u | 4u << 3 | 3u << 6 | 7u << 9 | 2u << 12 | 6u << 15 | 1u << 18 | 5u << 21, 1u | 5u << 3 | 0u << 6 | 4u << 9 | 3u << 12 | 7u << 15 | 2u << 18 | 6u << 21,
            2u | 6u << 3 | 1u << 6 | 5u << 9 | 0u << 12 | 4u << 15 | 3u << 18 | 7u << 21, 3u | 7u << 3 | 2u << 6 | 6u << 9 | 1u << 12 | 5u << 15 | 0u << 18 | 4u << 21};
        if (edge >= 0 && edge < 12) {
            const unsigned q = TI[edge];
            const double vp = sel8(v, q & 7), vq = sel8(v, (q >> 3) & 7);
            t = vp / (vp - vq + LIB_EPS);
            const double b0 = sel8(v, (q >> 6) & 7), b1 = sel8(v, (q >> 9) & 7), c0 = sel8(v, (q >> 12) & 7), c1 = sel8(v, (q >> 15) & 7);
            const double d0 = sel8(v, (q >> 18) & 7), d1 = sel8(v, (q >> 21) & 7);
            Bt = b0 + (b1 - b0) * t; Ct = c0 + (c1 - c0) * t; Dt = d0 + (d1 - d0) * t;
        }
    }
    const int test = (At >= 0 ? 1 : 0) + (Bt >= 0 ? 2 : 0) + (Ct >= 0 ? 4 : 0) + (Dt >= 0 ? 8 : 0);
    switch (test) {
    case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
    // the library's if-chain returns 0 when the inner condition of 5 / 10 fails (Lewiner's original falls through to s < 0)
    case 5: return (At * Ct - Bt * Dt < LIB_EPS) ? s > 0 : false;
    case 10: return (At * Ct - Bt * Dt >= LIB_EPS) ? s > 0 : false;
    default: return s < 0;      // 7, 11, 13, 14, 15
    }
}

// the big switch: cube index -> row of the tiling table (oracle/mc_oracle.c resolve); -1 = nothing to add
__device__ int resolve_row(const float *val, const McArgs &a, int idx, const uint32_t *lds)
{
    const int kase = tab_u8(lds, mc::OFF_CASES + 2 * idx), cf = tab_i8(lds, mc::OFF_CASES + 2 * idx + 1);
    switch (kase) {         // cases without tests need no arithmetic at all
    case 1: return mc::ROW_TILING1 + cf;
    case 2: return mc::ROW_TILING2 + cf;
    case 5: return mc::ROW_TILING5 + cf;
    case 8: return mc::ROW_TILING8 + cf;
    case 9: return mc::ROW_TILING9 + cf;
    case 11: return mc::ROW_TILING11 + cf;
    case 14: return mc::ROW_TILING14 + cf;
    default: break;
    }
    double v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = (double)val[c] - a.iso;
    int sub = 0;
    switch (kase) {
    case 3:
        return test_face(v, tab_i8(lds, mc::OFF_TEST3 + cf)) ? mc::ROW_TILING3_2 + cf : mc::ROW_TILING3_1 + cf;
    case 4:
        return test_internal(v, 4, -1, tab_i8(lds, mc::OFF_TEST4 + cf)) ? mc::ROW_TILING4_1 + cf : mc::ROW_TILING4_2 + cf;
    case 6:
        if (test_face(v, tab_i8(lds, mc::OFF_TEST6 + 3 * cf))) return mc::ROW_TILING6_2 + cf;
        return test_internal(v, 6, tab_i8(lds, mc::OFF_TEST6 + 3 * cf + 2), tab_i8(lds, mc::OFF_TEST6 + 3 * cf + 1)) ? mc::ROW_TILING6_1_1 + cf
                                                                                                                       : mc::ROW_TILING6_1_2 + cf;
    case 7:
        for (int k = 0; k < 3; ++k) if (test_face(v, tab_i8(lds, mc::OFF_TEST7 + 5 * cf + k))) sub |= 1 << k;
        switch (sub) {
        case 0: return mc::ROW_TILING7_1 + cf;
        case 1: return mc::ROW_TILING7_2 + 3 * cf + 0;
        case 2: return mc::ROW_TILING7_2 + 3 * cf + 1;
        case 3: return mc::ROW_TILING7_3 + 3 * cf + 0;
        case 4: return mc::ROW_TILING7_2 + 3 * cf + 2;
        case 5: return mc::ROW_TILING7_3 + 3 * cf + 1;
        case 6: return mc::ROW_TILING7_3 + 3 * cf + 2;
        default:
            return test_internal(v, 7, tab_i8(lds, mc::OFF_TEST7 + 5 * cf + 4), tab_i8(lds, mc::OFF_TEST7 + 5 * cf + 3)) ? mc::ROW_TILING7_4_2 + cf
                                                                                                                           : mc::ROW_TILING7_4_1 + cf;
        }
    case 10:
    case 12: {
        const bool is10 = kase == 10;
        const int toff = is10 ? mc::OFF_TEST10 + 3 * cf : mc::OFF_TEST12 + 4 * cf;
        const bool f0 = test_face(v, tab_i8(lds, toff)), f1 = test_face(v, tab_i8(lds, toff + 1));
        if (f0 && f1) return (is10 ? mc::ROW_TILING10_1_1_ : mc::ROW_TILING12_1_1_) + cf;
        if (f0) return (is10 ? mc::ROW_TILING10_2 : mc::ROW_TILING12_2) + cf;
        if (f1) return (is10 ? mc::ROW_TILING10_2_ : mc::ROW_TILING12_2_) + cf;
        const bool in = test_internal(v, kase, is10 ? -1 : tab_i8(lds, toff + 3), tab_i8(lds, toff + 2));
        if (in) return (is10 ? mc::ROW_TILING10_1_1 : mc::ROW_TILING12_1_1) + cf;
        return (is10 ? mc::ROW_TILING10_1_2 : mc::ROW_TILING12_1_2) + cf;
    }
    case 13: {
        for (int k = 0; k < 6; ++k) if (test_face(v, tab_i8(lds, mc::OFF_TEST13 + 7 * cf + k))) sub |= 1 << k;
        const int sc = tab_i8(lds, mc::OFF_SUBCONFIG13 + sub);
        if (sc == 0) return mc::ROW_TILING13_1 + cf;
        if (sc >= 1 && sc <= 6) return mc::ROW_TILING13_2 + 6 * cf + sc - 1;
        if (sc >= 7 && sc <= 18) return mc::ROW_TILING13_3 + 12 * cf + sc - 7;
        if (sc >= 19 && sc <= 22) return mc::ROW_TILING13_4 + 4 * cf + sc - 19;
        if (sc >= 23 && sc <= 26) {
            const int r51 = mc::ROW_TILING13_5_1 + 4 * cf + sc - 23;
            return test_internal(v, 13, row_edge(lds, r51, 0), tab_i8(lds, mc::OFF_TEST13 + 7 * cf + 6)) ? r51 : mc::ROW_TILING13_5_2 + 4 * cf + sc - 23;
        }
        if (sc >= 27 && sc <= 38) return mc::ROW_TILING13_3_ + 12 * cf + sc - 27;
        if (sc >= 39 && sc <= 44) return mc::ROW_TILING13_2_ + 6 * cf + sc - 39;
        if (sc == 45) return mc::ROW_TILING13_1_ + cf;
        return -1;      // combination of face tests the tables mark impossible: the library adds nothing
    }
    default: return -1;
    }
}

// which of the cell's 13 possible vertices it is the FIRST cell (in traversal order) to refer to: edges 5, 6, 10 and the centre
// always; the others only when the earlier neighbours that share them do not exist (see the header)
__device__ __forceinline__ unsigned creator_mask(int z, int y, int x)
{
    unsigned m = (1u << 5) | (1u << 6) | (1u << 10) | (1u << 12);
    if (z == 0) m |= (1u << 1) | (1u << 2);
    if (y == 0) m |= (1u << 4) | (1u << 9);
    if (x == 0) m |= (1u << 7) | (1u << 11);
    if (y == 0 && z == 0) m |= 1u << 0;
    if (x == 0 && z == 0) m |= 1u << 3;
    if (x == 0 && y == 0) m |= 1u << 8;
    return m;
}

// cube edge e -> base grid point offset (dx, dy, dz) and axis (0 = x, 1 = y, 2 = z), packed 5 bits per edge
__device__ __forceinline__ void edge_base(int e, int &dx, int &dy, int &dz, int &axis)
{
    //            e:  0  1  2  3  4  5  6  7  8  9 10 11      bits: dx | dy<<1 | dz<<2 | axis<<3
    const unsigned long long T = 0ull | (1ull | 1ull << 3) << 5 | (2ull) << 10 | (0ull | 1ull << 3) << 15 | (4ull) << 20 | (5ull | 1ull << 3) << 25 |
                                 (6ull) << 30 | (4ull | 1ull << 3) << 35 | (0ull | 2ull << 3) << 40 | (1ull | 2ull << 3) << 45 | (3ull | 2ull << 3) << 50 |
                                 (2ull | 2ull << 3) << 55;
    const unsigned q = (unsigned)(T >> (5 * e)) & 31u;
    dx = q & 1; dy = (q >> 1) & 1; dz = (q >> 2) & 1; axis = q >> 3;
}

// exclusive prefix sum of one value per thread across a 256-thread block; returns the block total in `total`
__device__ __forceinline__ unsigned block_exclusive(unsigned v, unsigned *lds4, unsigned &total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    __syncthreads();
    if (lane == 63) lds4[w] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int k = 0; k < w; ++k) base += lds4[k];
    total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    return base + incl - v;
}

// ---------------- pass A1: classify ----------------
// The streaming pass over the volume: a thread takes four consecutive cells -- the rows (z, y), (z, y + 1), (z + 1, y), (z + 1, y + 1) as one 16-byte load
// each, the fifth column from the next lane --, and one comparison of the 20 "above the level" bits tells that none of them is crossed (most are not).
// For a tile with a crossed cell: the 8-bit cube index of each of its cells (idx8, 0 = not crossed / not a cell) and its number in the list of crossed
// tiles, which the later passes walk instead of the volume.  Every tile's three counters are zeroed here (pass A2 fills in the crossed ones).
// Few registers, no tables: bound by the read of the volume.
__global__ __launch_bounds__(256) void mc_classify_kernel(McArgs a, unsigned *__restrict__ tile_v, unsigned *__restrict__ tile_t, unsigned *__restrict__ tile_c,
                                                          uint8_t *__restrict__ idx8)
{
    const int yx = a.n1 * a.n2;
    const bool vec = (a.n2 & 3) == 0 && (reinterpret_cast<uintptr_t>(a.vol) & 15) == 0;      // a thread's four cells then lie in one x row, 16-byte aligned
    const int lane = threadIdx.x & 63;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int64_t li0 = (int64_t)tile * TILE + threadIdx.x * 4;
        unsigned word = 0;                                   // the four cube indices, cell k in byte k
        if (li0 < a.N) {
            const int z = (int)(li0 / yx), rem = (int)(li0 - (int64_t)z * yx), y = rem / a.n2, x0 = rem - y * a.n2;
            if (vec) {
                const bool cellrow = z + 1 < a.n0 && y + 1 < a.n1;
                const bool fifth = x0 + 4 < a.n2;            // the cell at x0 + 3 exists
                unsigned above = 0;                          // bit q * 5 + i: value (row q, x0 + i) > iso  (== ((double)val - iso > 0), the library's test)
                if (cellrow) {
                    const float *p = a.vol + li0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float *pq = p + (q >> 1) * (int64_t)yx + (q & 1) * a.n2;
                        const float4 v4 = *reinterpret_cast<const float4 *>(pq);
                        above |= ((v4.x > a.iso_f ? 1u : 0u) | (v4.y > a.iso_f ? 2u : 0u) | (v4.z > a.iso_f ? 4u : 0u) | (v4.w > a.iso_f ? 8u : 0u)) << (q * 5);
                    }
                }
                // fifth column = the next lane's first (the same rows, x0 + 4), unless that lane is in another wave or on another x row
                const unsigned nb = __shfl_down(above, 1, 64);
                const bool nb_ok = lane < 63 && fifth;       // (lane + 1 then holds x0 + 4 of the same (z, y): its li0 is this one's + 4)
                unsigned col5 = (nb & 1u) | (((nb >> 5) & 1u) << 1) | (((nb >> 10) & 1u) << 2) | (((nb >> 15) & 1u) << 3);
                if (cellrow && fifth && !nb_ok) {
                    const float *p = a.vol + li0 + 4;
                    col5 = (p[0] > a.iso_f ? 1u : 0u) | (p[a.n2] > a.iso_f ? 2u : 0u) | (p[yx] > a.iso_f ? 4u : 0u) | (p[(int64_t)yx + a.n2] > a.iso_f ? 8u : 0u);
                }
                if (!fifth) col5 = ((above >> 3) & 1u) | (((above >> 8) & 1u) << 1) | (((above >> 13) & 1u) << 2) | (((above >> 18) & 1u) << 3);
                above |= ((col5 & 1u) << 4) | (((col5 >> 1) & 1u) << 9) | (((col5 >> 2) & 1u) << 14) | (((col5 >> 3) & 1u) << 19);
                if (cellrow && above != 0u && above != 0xfffffu) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (k == 3 && !fifth) break;
                        // corners 0..7 of the cell at x0 + k: (z,y,x) (z,y,x+1) (z,y+1,x+1) (z,y+1,x) (z+1,y,x) (z+1,y,x+1) (z+1,y+1,x+1) (z+1,y+1,x)
                        const unsigned idx = ((above >> k) & 1u) | (((above >> (k + 1)) & 1u) << 1) | (((above >> (5 + k + 1)) & 1u) << 2) | (((above >> (5 + k)) & 1u) << 3) |
                                             (((above >> (10 + k)) & 1u) << 4) | (((above >> (10 + k + 1)) & 1u) << 5) | (((above >> (15 + k + 1)) & 1u) << 6) |
                                             (((above >> (15 + k)) & 1u) << 7);
                        if (idx != 255u) word |= idx << (8 * k);
                    }
                }
            } else {
#pragma unroll 1
                for (int k = 0; k < 4; ++k) {
                    const int64_t li = li0 + k;
                    if (li >= a.N) break;
                    const int zz = (int)(li / yx), r2 = (int)(li - (int64_t)zz * yx), yy = r2 / a.n2, xx = r2 - yy * a.n2;
                    if (zz + 1 >= a.n0 || yy + 1 >= a.n1 || xx + 1 >= a.n2) continue;
                    const float *p = a.vol + li;
                    const float val[8] = {p[0], p[1], p[a.n2 + 1], p[a.n2], p[yx], p[yx + 1], p[yx + a.n2 + 1], p[yx + a.n2]};
                    unsigned idx = 0;
#pragma unroll
                    for (int c = 0; c < 8; ++c) idx |= (val[c] > a.iso_f ? 1u : 0u) << c;
                    if (idx != 255u) word |= idx << (8 * k);
                }
            }
        }
        const int any = __syncthreads_or(word != 0u);
        if (threadIdx.x == 0) { tile_v[tile] = 0; tile_t[tile] = 0; tile_c[tile] = any ? 1u : 0u; }      // tile_c: the "crossed" flag until pass A2 counts
        if (any && li0 < a.N) *reinterpret_cast<unsigned *>(idx8 + li0) = word;       // (the array is padded to whole tiles)
    }
}

// Pass A1 for volumes whose tiles are whole x rows of one z plane (n2 divides 1024, n1 a multiple of the 1024 / n2 rows of a tile: 256^3, 512^3,
// 384 x 384 x 128 ...): a workgroup walks ZS consecutive z planes of one tile column and keeps the "above the level" bits of the plane it has just
// read as the lower plane of the next cell layer, and a thread gets the bits of the row above its own (y + 1) from the thread that read it, through LDS:
// one 16-byte load per thread and plane (plus one for the threads of a tile's last row) instead of four -- the general kernel above is bound by the
// caches' request rate, four corner rows per cell row.  Same outputs, bit for bit (the bits are the same comparisons).
template <int ZS>
__global__ __launch_bounds__(256) void mc_classify_walk_kernel(McArgs a, unsigned *__restrict__ tile_v, unsigned *__restrict__ tile_t, unsigned *__restrict__ tile_c,
                                                               uint8_t *__restrict__ idx8)
{
    __shared__ unsigned sh[2][256];
    const int yx = a.n1 * a.n2, R = TILE / a.n2, TP = a.n1 / R, q4 = a.n2 >> 2;         // rows per tile, tiles per z plane, threads per row
    const int lane = threadIdx.x & 63;
    const int row = threadIdx.x / q4, x0 = (threadIdx.x - row * q4) * 4;
    const bool fifth = x0 + 4 < a.n2;                      // the cell at x0 + 3 exists
    const bool last_row = row + 1 == R;
    const int nseg = (a.n0 + ZS - 1) / ZS;
    unsigned flip = 0;
    for (int work = blockIdx.x; work < TP * nseg; work += gridDim.x) {
        const int yt = work % TP, z0 = (work / TP) * ZS, z1 = min(a.n0, z0 + ZS);
        const int y = yt * R + row;
        const bool has_y1 = y + 1 < a.n1;
        // every load of the segment first (ZS + 1 planes of this thread's row; for a tile's last row also the row above, which belongs to the next tile):
        // the pass is a latency chain otherwise -- one 16-byte load per thread between two barriers moves 1 TB/s, a stock reduction 2.7
        const float *p0 = a.vol + ((int64_t)z0 * yx + (int64_t)y * a.n2 + x0);
        float4 own[ZS + 1], upr[ZS + 1];
#pragma unroll
        for (int i = 0; i <= ZS; ++i) {
            const bool in = z0 + i < a.n0;
            own[i] = in ? *reinterpret_cast<const float4 *>(p0 + (int64_t)i * yx) : make_float4(0.f, 0.f, 0.f, 0.f);
            upr[i] = in && last_row && has_y1 ? *reinterpret_cast<const float4 *>(p0 + (int64_t)i * yx + a.n2) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // bits 0..4: a row at x0 .. x0 + 4 (the fifth column from the next lane, which holds x0 + 4 of the same row when that column exists and the lane is in this wave)
        auto row_bits = [&](const float4 v4, const float *pr) -> unsigned {
            unsigned b = (v4.x > a.iso_f ? 1u : 0u) | (v4.y > a.iso_f ? 2u : 0u) | (v4.z > a.iso_f ? 4u : 0u) | (v4.w > a.iso_f ? 8u : 0u);
            const unsigned nb = __shfl_down(b, 1, 64);
            unsigned c5 = (b >> 3) & 1u;                       // no fifth column: repeat the fourth (the cell at x0 + 3 is not a cell)
            if (fifth) c5 = lane < 63 ? (nb & 1u) : (pr[4] > a.iso_f ? 1u : 0u);
            return b | (c5 << 4);
        };
        // bits 0..4: row (z0 + i, y);  bits 5..9: row (z0 + i, y + 1), from the thread that read it (LDS) or from this thread's second load
        auto plane_bits = [&](const float4 vo, const float4 vu, int i) -> unsigned {
            const float *pr = p0 + (int64_t)i * yx;
            const unsigned mine = row_bits(vo, pr);
            sh[flip][threadIdx.x] = mine;
            __syncthreads();
            unsigned up = 0;
            if (!last_row) up = sh[flip][threadIdx.x + q4];
            else if (has_y1) up = row_bits(vu, pr + a.n2);
            flip ^= 1u;
            return mine | (up << 5);
        };
        unsigned lower = plane_bits(own[0], upr[0], 0);
#pragma unroll
        for (int i = 0; i < ZS; ++i) {
            const int z = z0 + i;
            if (z >= z1) break;                                  // (block-uniform)
            const bool cells = z + 1 < a.n0 && has_y1;
            const unsigned upper = z + 1 < a.n0 ? plane_bits(own[i + 1], upr[i + 1], i + 1) : 0u;
            const unsigned above = lower | (upper << 10);
            unsigned word = 0;
            if (cells && above != 0u && above != 0xfffffu) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k == 3 && !fifth) break;
                    const unsigned idx = ((above >> k) & 1u) | (((above >> (k + 1)) & 1u) << 1) | (((above >> (5 + k + 1)) & 1u) << 2) | (((above >> (5 + k)) & 1u) << 3) |
                                         (((above >> (10 + k)) & 1u) << 4) | (((above >> (10 + k + 1)) & 1u) << 5) | (((above >> (15 + k + 1)) & 1u) << 6) |
                                         (((above >> (15 + k)) & 1u) << 7);
                    if (idx != 255u) word |= idx << (8 * k);
                }
            }
            const int tile = z * TP + yt;
            const int any = __syncthreads_or(word != 0u);
            if (threadIdx.x == 0) { tile_v[tile] = 0; tile_t[tile] = 0; tile_c[tile] = any ? 1u : 0u; }
            if (any) *reinterpret_cast<unsigned *>(idx8 + (int64_t)tile * TILE + threadIdx.x * 4) = word;
            lower = upper;
        }
    }
}

// The list of crossed tiles from their flags, in tile order: one workgroup, 16 consecutive flags per thread, a block scan of the thread sums.  (An atomic
// counter in the classify pass did this for free in the source and for 40 us on the device: ~5,000 same-address atomics are served one after the other.)
__global__ __launch_bounds__(1024) void mc_compact_kernel(const unsigned *__restrict__ flags, int n, unsigned *__restrict__ list, unsigned *__restrict__ list_n)
{
    constexpr int PER = 16;
    __shared__ unsigned wsum[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned carry = 0;
    for (int b0 = 0; b0 < n; b0 += 1024 * PER) {
        const int i0 = b0 + threadIdx.x * PER;
        unsigned f[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { f[j] = i0 + j < n ? flags[i0 + j] : 0u; sum += f[j] != 0u; }
        unsigned incl = sum;
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        __syncthreads();
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        unsigned base = 0, all = 0;
        for (int k = 0; k < 16; ++k) { if (k < w) base += wsum[k]; all += wsum[k]; }
        unsigned pos = carry + base + incl - sum;
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (f[j]) list[pos++] = (unsigned)(i0 + j);
        carry += all;
    }
    if (threadIdx.x == 0) *list_n = carry;
}

// ---------------- pass A2: the crossed tiles' cells -> tiling rows, counts ----------------
// rows16: one int16 per grid point of a crossed tile (the tiling row of the cell whose low corner it is, -1 = not crossed / not a cell)
// Crossed cells are sparse even in a crossed tile (~11 % of the cells of a dense frame's tiles): walking a thread's four cells in lockstep ran the case
// analysis (resolve_row: table look-ups, the fp64 face and interior tests) four times per wave with a tenth of the lanes alive.  Each wave now COMPACTS the
// crossed cells of its 256 cells (ballot-free: a lane prefix over the per-lane counts, the cells listed in LDS in traversal order) and gives every lane one
// crossed cell at a time: one or two passes of the case analysis per wave with most lanes alive (94 -> see profiles/r06_mc_kernels.md).
// wave-level exclusive prefix of one small count per lane; returns the lane's base, `total` = the wave's sum
__device__ __forceinline__ unsigned wave_exclusive(unsigned v, unsigned &total)
{
    const int lane = threadIdx.x & 63;
    unsigned incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    total = __shfl(incl, 63, 64);
    return incl - v;
}

__global__ __launch_bounds__(256) void mc_count_kernel(McArgs a, const unsigned *__restrict__ list, const unsigned *__restrict__ list_n,
                                                       const uint8_t *__restrict__ idx8, unsigned *__restrict__ tile_v, unsigned *__restrict__ tile_t,
                                                       unsigned *__restrict__ tile_c, int16_t *__restrict__ rows16)
{
    __shared__ uint32_t tab[mc::BLOB_WORDS];
    __shared__ unsigned red[12];
    __shared__ uint16_t items[TILE];            // per wave [256]: (lane << 2 | k) of its crossed cells, in traversal order; low byte pair reused below
    __shared__ uint8_t cube[TILE];              // the tile's cube indices
    __shared__ int16_t rows_l[TILE];            // the tile's tiling rows (-1: not crossed)
    const unsigned n = *list_n;
    if (blockIdx.x >= n) return;
    load_tables(a.tables, tab);
    const int yx = a.n1 * a.n2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (unsigned it = blockIdx.x; it < n; it += gridDim.x) {
        const int tile = (int)list[it];
        const int64_t li0 = (int64_t)tile * TILE + threadIdx.x * 4;
        const unsigned word = li0 < a.N ? *reinterpret_cast<const unsigned *>(idx8 + li0) : 0u;
        *reinterpret_cast<unsigned *>(cube + threadIdx.x * 4) = word;
        *reinterpret_cast<uint2 *>(rows_l + threadIdx.x * 4) = make_uint2(0xffffffffu, 0xffffffffu);
        const unsigned mine = (word & 0xffu ? 1u : 0u) + (word & 0xff00u ? 1u : 0u) + (word & 0xff0000u ? 1u : 0u) + (word & 0xff000000u ? 1u : 0u);
        unsigned total;
        unsigned base = wave_exclusive(mine, total);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((word >> (8 * k)) & 255u) items[wave * 256 + base++] = (uint16_t)(lane * 4 + k);
        __builtin_amdgcn_wave_barrier();        // a wave reads only what it wrote itself (LDS operations of a wave complete in order)
        unsigned nv = 0, nt = 0, nc = 0;
        for (unsigned i = lane; i < total; i += 64) {
            const int c = wave * 256 + items[wave * 256 + i];               // the cell's place in the tile
            const int64_t li = (int64_t)tile * TILE + c;
            const int z = (int)(li / yx), r2 = (int)(li - (int64_t)z * yx), y = r2 / a.n2, x = r2 - y * a.n2;
            const float *p = a.vol + li;                                    // (a crossed cell exists: its +1 neighbours along every axis do)
            const float val[8] = {p[0], p[1], p[a.n2 + 1], p[a.n2], p[yx], p[yx + 1], p[yx + a.n2 + 1], p[yx + a.n2]};
            const int row = resolve_row(val, a, (int)cube[c], tab);
            rows_l[c] = (int16_t)row;
            if (row >= 0) {
                nt += row_ntri(tab, row);
                nv += __popc(row_mask(tab, row) & creator_mask(z, y, x));
                nc += 1;
            }
        }
        for (int o = 32; o > 0; o >>= 1) { nv += __shfl_down(nv, o, 64); nt += __shfl_down(nt, o, 64); nc += __shfl_down(nc, o, 64); }
        __syncthreads();
        if (lane == 0) { red[wave] = nv; red[4 + wave] = nt; red[8 + wave] = nc; }
        __syncthreads();
        if (threadIdx.x == 0) {
            tile_v[tile] = red[0] + red[1] + red[2] + red[3];
            tile_t[tile] = red[4] + red[5] + red[6] + red[7];
            tile_c[tile] = red[8] + red[9] + red[10] + red[11];
        }
        if (li0 < a.N) *reinterpret_cast<uint2 *>(rows16 + li0) = *reinterpret_cast<const uint2 *>(rows_l + threadIdx.x * 4);
        __syncthreads();                         // the LDS arrays are rewritten by the next tile
    }
}

// exclusive scan of three arrays by one workgroup of 1024: a batch of 16384 elements goes through LDS (coalesced in and out), a thread sums its 16
// consecutive ones, the block scans the thread sums.  totals[0..2] = the sums, totals[3] = 1 when the mesh fits the caller's capacity (and 32-bit
// indices), which the later passes require before they write anything.
__global__ __launch_bounds__(1024) void mc_scan_kernel(unsigned *__restrict__ t0, unsigned *__restrict__ t1, unsigned *__restrict__ t2, int n,
                                                       unsigned long long cap_v, unsigned long long cap_f, unsigned long long *__restrict__ totals)
{
    constexpr int PER = 16, BATCH = 1024 * PER;
    __shared__ unsigned buf[BATCH + BATCH / 16];             // padded: a thread's 16 consecutive words start 17 words apart
    __shared__ unsigned wsum[16];
    unsigned *arr[3] = {t0, t1, t2};
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    auto at = [](int i) { return i + (i >> 4); };
    unsigned long long tot[3];
    for (int q = 0; q < 3; ++q) {
        unsigned long long carry = 0;
        for (int b0 = 0; b0 < n; b0 += BATCH) {
            for (int j = 0; j < PER; ++j) { const int i = j * 1024 + threadIdx.x; buf[at(i)] = b0 + i < n ? arr[q][b0 + i] : 0u; }
            __syncthreads();
            unsigned v[PER], sum = 0;
#pragma unroll
            for (int j = 0; j < PER; ++j) { v[j] = buf[at(threadIdx.x * PER + j)]; sum += v[j]; }
            unsigned incl = sum;
            for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            if (lane == 63) wsum[w] = incl;
            __syncthreads();
            unsigned base = 0, all = 0;
            for (int k = 0; k < 16; ++k) { if (k < w) base += wsum[k]; all += wsum[k]; }
            unsigned run = (unsigned)carry + base + incl - sum;
#pragma unroll
            for (int j = 0; j < PER; ++j) { buf[at(threadIdx.x * PER + j)] = run; run += v[j]; }
            carry += all;
            __syncthreads();
            for (int j = 0; j < PER; ++j) { const int i = j * 1024 + threadIdx.x; if (b0 + i < n) arr[q][b0 + i] = buf[at(i)]; }
            __syncthreads();
        }
        tot[q] = carry;
    }
    if (threadIdx.x == 0) {
        totals[0] = tot[0]; totals[1] = tot[1]; totals[2] = tot[2];
        totals[3] = (tot[0] <= cap_v && tot[1] <= cap_f && tot[0] < (1ull << 31) && tot[1] < (1ull << 31)) ? 1ull : 0ull;
    }
}

// ---------------- pass B: vertices + normals ----------------
struct EmitArgs {
    float vox[3];        // voxel_size = (b1 - b0) / res            (recon_util.py:62)
    float b0[3];         // bounds[0]
    float len[3];        // bounds[1] - bounds[0]
};

// per-axis 4-tap weights of (trilinear sample o 1-D Sobel factor) around base index i0 - 1
__device__ __forceinline__ void axis_weights(float pix, int n, int &ibase, float wS[4], float wD[4])
{
    // pix already clamped to [0, n-1] (padding_mode='border', align_corners=True)
    const float fl = floorf(pix);
    const int i0 = (int)fl;
    const int i1 = i0 + 1 < n ? i0 + 1 : n - 1;
    const float t = pix - fl;
    ibase = i0 - 1;
#pragma unroll
    for (int m = 0; m < 4; ++m) { wS[m] = 0.f; wD[m] = 0.f; }
    const int cs[2] = {i0, i1};
    const float cw[2] = {1.0f - t, t};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int o = cs[c] - ibase;     // 1 or 2
        // smoothing [1,2,1] and central difference [-1,0,1] centred on cs[c]; conv3d zero padding =>
        // taps outside [0, n-1] contribute nothing (they are skipped at read time)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int d = m - o;
            if (d == -1) { wS[m] += cw[c]; wD[m] -= cw[c]; }
            else if (d == 0) { wS[m] += 2.0f * cw[c]; }
            else if (d == 1) { wS[m] += cw[c]; wD[m] += cw[c]; }
        }
    }
}

__device__ __forceinline__ void vertex_normal(const McArgs &a, const EmitArgs &e, const float vidx[3], float nrm[3])
{
    // reference arithmetic, in its order (recon_util.py:65-66, F.grid_sample unnormalise)
    const int dim[3] = {a.n0, a.n1, a.n2};
    int ib[3];
    float wS[3][4], wD[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = vidx[c] * e.vox[c] + e.b0[c] + 0.5f * e.vox[c];
        const float g = 2.0f * (v - e.b0[c]) / e.len[c] - 1.0f;
        float pix = (g + 1.0f) * 0.5f * (float)(dim[c] - 1);
        pix = fminf(fmaxf(pix, 0.0f), (float)(dim[c] - 1));
        axis_weights(pix, dim[c], ib[c], wS[c], wD[c]);
    }
    // conv3d's zero padding: a tap outside the volume contributes nothing -- its weight is zeroed and its address clamped, so that all 64 loads are
    // unconditional and in flight together (one memory latency per vertex instead of one per (i, j) row of the stencil)
    int off[3][4];
    const int stride[3] = {a.n1 * a.n2, a.n2, 1};
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int idx = ib[c] + m;
            const bool in = idx >= 0 && idx < dim[c];
            if (!in) { wS[c][m] = 0.f; wD[c][m] = 0.f; }
            off[c][m] = (in ? idx : 0) * stride[c];
        }
    float val[4][4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) val[i][j][k] = a.vol[(int64_t)off[0][i] + off[1][j] + off[2][k]];
    float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float sS = 0.f, sD = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { sS += val[i][j][k] * wS[2][k]; sD += val[i][j][k] * wD[2][k]; }
            gx += wD[0][i] * wS[1][j] * sS;
            gy += wS[0][i] * wD[1][j] * sS;
            gz += wS[0][i] * wS[1][j] * sD;
        }
    gx /= 32.0f * e.vox[0]; gy /= 32.0f * e.vox[1]; gz /= 32.0f * e.vox[2];     // recon_util.py:18-20
    const float nn = sqrtf(gx * gx + gy * gy + gz * gz);                          // :45-47 (no epsilon)
    nrm[0] = -(gx / nn); nrm[1] = -(gy / nn); nrm[2] = -(gz / nn);                // :68
}


// The library's vertex of cube edge e (or the centre, e == 12) of cell (z, y, x), in index units (axis0, axis1, axis2),
// computed in double and rounded to float exactly as its C code does.
__device__ __forceinline__ void vertex_position(const McArgs &a, int z, int y, int x, int e, float vidx[3])
{
    const int yx = a.n1 * a.n2;
    const float *p = a.vol + ((int64_t)z * yx + (int64_t)y * a.n2 + x);
    if (e < 12) {
        int dx, dy, dz, axis;
        edge_base(e, dx, dy, dz, axis);
        const float *q = p + dz * yx + dy * a.n2 + dx;
        const int step = axis == 0 ? 1 : (axis == 1 ? a.n2 : yx);
        const double w_lo = 1.0 / (LIB_EPS + fabs((double)q[0] - a.iso));
        const double w_hi = 1.0 / (LIB_EPS + fabs((double)q[step] - a.iso));
        const double t = w_hi / (w_lo + w_hi);
        const int bz = z + dz, by = y + dy, bx = x + dx;
        vidx[0] = axis == 2 ? (float)((double)bz + t) : (float)bz;
        vidx[1] = axis == 1 ? (float)((double)by + t) : (float)by;
        vidx[2] = axis == 0 ? (float)((double)bx + t) : (float)bx;
    } else {
        double w[8];
        w[0] = 1.0 / (LIB_EPS + fabs((double)p[0] - a.iso));
        w[1] = 1.0 / (LIB_EPS + fabs((double)p[1] - a.iso));
        w[2] = 1.0 / (LIB_EPS + fabs((double)p[a.n2 + 1] - a.iso));
        w[3] = 1.0 / (LIB_EPS + fabs((double)p[a.n2] - a.iso));
        w[4] = 1.0 / (LIB_EPS + fabs((double)p[yx] - a.iso));
        w[5] = 1.0 / (LIB_EPS + fabs((double)p[yx + 1] - a.iso));
        w[6] = 1.0 / (LIB_EPS + fabs((double)p[yx + a.n2 + 1] - a.iso));
        w[7] = 1.0 / (LIB_EPS + fabs((double)p[yx + a.n2] - a.iso));
        const double fx = ((w[1] + w[2]) + w[5]) + w[6];
        const double fy = ((w[2] + w[3]) + w[6]) + w[7];
        const double fz = ((w[4] + w[5]) + w[6]) + w[7];
        const double ff = ((((((w[0] + w[1]) + w[2]) + w[3]) + w[4]) + w[5]) + w[6]) + w[7];
        vidx[0] = (float)((double)z + fz / ff);
        vidx[1] = (float)((double)y + fy / ff);
        vidx[2] = (float)((double)x + fx / ff);
    }
}

struct CellRec { uint32_t li, row, face0; int32_t cvid; };     // one per crossed cell, in traversal order

// Pass B only NUMBERS: creators describe their vertices -- (cell << 4) | edge id, 64 bits, in output order -- in the first two words of the
// vertex's own output slot (verts[3 * id]); positions and normals are evaluated by pass B' below, one vertex per thread.  (First cut: descriptors
// in a worst-case LDS array and the evaluation at the end of this kernel, by the lanes of the tile's workgroup: the case logic's 170 registers
// held the 64-tap normal stencil at three waves per SIMD, and half of the lanes had no vertex.)
__global__ __launch_bounds__(256) void mc_verts_kernel(McArgs a, EmitArgs e, const unsigned *__restrict__ tile_voff, const unsigned *__restrict__ tile_toff,
                                                       const unsigned *__restrict__ tile_coff, const unsigned long long *__restrict__ totals,
                                                       const unsigned *__restrict__ list, const unsigned *__restrict__ list_n,
                                                       const int16_t *__restrict__ rows16, int32_t *__restrict__ edge_map, CellRec *__restrict__ cells,
                                                       float *__restrict__ verts)
{
    __shared__ uint32_t tab[mc::BLOB_WORDS];
    __shared__ unsigned red[4];
    __shared__ uint16_t items[TILE];            // per wave [256]: its crossed cells (lane << 2 | k) in traversal order (as in mc_count_kernel)
    __shared__ int16_t rows_l[TILE];
    __shared__ unsigned loc_l[TILE], floc_l[TILE], cloc_l[TILE];      // per cell: where its first new vertex, first face and its record go
    if (!totals[3]) return;                                  // the mesh does not fit the caller's buffers: nothing is written (AVC_ERR_CAPACITY)
    const unsigned nlist = *list_n;
    if (blockIdx.x >= nlist) return;
    load_tables(a.tables, tab);
    const int yx = a.n1 * a.n2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (unsigned it = blockIdx.x; it < nlist; it += gridDim.x) {          // the crossed tiles, in any order: their offsets say where they write
        const int tile = (int)list[it];
        const unsigned vbase = tile_voff[tile], tbase = tile_toff[tile], cbase = tile_coff[tile];
        int rows[4] = {-1, -1, -1, -1};
        unsigned cv[4] = {0, 0, 0, 0}, ct[4] = {0, 0, 0, 0};
        unsigned nv = 0, nt = 0, nc = 0;
        const int64_t li0 = (int64_t)tile * TILE + threadIdx.x * 4;
        if (li0 < a.N) {
            const uint2 rw = *reinterpret_cast<const uint2 *>(rows16 + li0);
            rows[0] = (int16_t)(rw.x & 0xffffu); rows[1] = (int16_t)(rw.x >> 16); rows[2] = (int16_t)(rw.y & 0xffffu); rows[3] = (int16_t)(rw.y >> 16);
            *reinterpret_cast<uint2 *>(rows_l + threadIdx.x * 4) = rw;
        } else {
            *reinterpret_cast<uint2 *>(rows_l + threadIdx.x * 4) = make_uint2(0xffffffffu, 0xffffffffu);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (rows[k] >= 0) {
                const int64_t li = li0 + k;
                const int cz = (int)(li / yx), r = (int)(li - (int64_t)cz * yx), cy = r / a.n2, cx = r - cy * a.n2;
                ct[k] = row_ntri(tab, rows[k]);
                cv[k] = __popc(row_mask(tab, rows[k]) & creator_mask(cz, cy, cx));
                nt += ct[k]; nv += cv[k]; nc += 1;
            }
        }
        unsigned total, tt, tc;
        unsigned loc = vbase + block_exclusive(nv, red, total);
        unsigned floc = tbase + block_exclusive(nt, red, tt);
        unsigned cloc = cbase + block_exclusive(nc, red, tc);
        // the crossed cells of this wave, compacted in traversal order, with their offsets (the serial emission loop below then runs with most lanes alive
        // instead of four times per wave with a tenth of them)
        unsigned wtotal;
        unsigned base = wave_exclusive(nc, wtotal);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (rows[k] >= 0) {
                const int c = threadIdx.x * 4 + k;
                items[wave * 256 + base++] = (uint16_t)(lane * 4 + k);
                loc_l[c] = loc; floc_l[c] = floc; cloc_l[c] = cloc;
                loc += cv[k]; floc += ct[k]; ++cloc;
            }
        __builtin_amdgcn_wave_barrier();
        for (unsigned i = lane; i < wtotal; i += 64) {
            const int c = wave * 256 + items[wave * 256 + i];
            const int64_t li = (int64_t)tile * TILE + c;
            const int cz = (int)(li / yx), r2 = (int)(li - (int64_t)cz * yx), cy = r2 / a.n2, cx = r2 - cy * a.n2;
            const int row = rows_l[c], n = 3 * row_ntri(tab, row);
            const unsigned creator = creator_mask(cz, cy, cx);
            unsigned vloc = loc_l[c];
            unsigned seen = 0;
            int32_t cvid = -1;
            for (int k = 0; k < n; ++k) {
                const int ed = row_edge(tab, row, k);
                if (seen & (1u << ed)) continue;
                seen |= 1u << ed;
                if (!(creator & (1u << ed))) continue;
                const int32_t id = (int32_t)vloc;
                const uint64_t d = ((uint64_t)li << 4) | (uint64_t)ed;
                reinterpret_cast<uint32_t *>(verts)[3 * (size_t)id] = (uint32_t)d; reinterpret_cast<uint32_t *>(verts)[3 * (size_t)id + 1] = (uint32_t)(d >> 32);
                ++vloc;
                if (ed == 12) { cvid = id; continue; }
                int dx, dy, dz, axis;
                edge_base(ed, dx, dy, dz, axis);
                edge_map[3 * (li + (int64_t)dz * yx + dy * a.n2 + dx) + axis] = id;
            }
            CellRec rec;
            rec.li = (uint32_t)li; rec.row = (uint32_t)row; rec.face0 = floc_l[c]; rec.cvid = cvid;
            cells[cloc_l[c]] = rec;
        }
        __syncthreads();                         // the LDS arrays are rewritten by the next tile
    }
}

// ---------------- pass B': positions + normals, one vertex per thread ----------------
__global__ __launch_bounds__(256) void mc_eval_kernel(McArgs a, EmitArgs e, const unsigned long long *__restrict__ totals, float *__restrict__ verts,
                                                      float *__restrict__ normals)
{
    if (!totals[3]) return;
    const size_t total_v = (size_t)totals[0];
    const int yx = a.n1 * a.n2;
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < total_v; id += (size_t)gridDim.x * 256) {
        const uint32_t *dw = reinterpret_cast<const uint32_t *>(verts) + 3 * id;
        const uint64_t d = (uint64_t)dw[0] | ((uint64_t)dw[1] << 32);
        const int64_t li = (int64_t)(d >> 4);
        const int z = (int)(li / yx), rem = (int)(li - (int64_t)z * yx);
        float vidx[3], out[3];
        vertex_position(a, z, rem / a.n2, rem % a.n2, (int)(d & 15u), vidx);
#pragma unroll
        for (int c = 0; c < 3; ++c)     // vertices = mc * voxel + b0 + 0.5 * voxel   (library: * spacing; recon_util.py:65), float32
            out[c] = __fadd_rn(__fadd_rn(__fmul_rn(vidx[c], e.vox[c]), e.b0[c]), __fmul_rn(0.5f, e.vox[c]));
        settle(out[0], out[1], out[2]);                                      // store_settle.h
        verts[3 * id + 0] = out[0]; verts[3 * id + 1] = out[1]; verts[3 * id + 2] = out[2];
        if (normals) {
            float n[3];
            vertex_normal(a, e, vidx, n);
            settle(n[0], n[1], n[2]);
            normals[3 * id + 0] = n[0]; normals[3 * id + 1] = n[1]; normals[3 * id + 2] = n[2];
        }
    }
}

// ---------------- pass C: faces ----------------
__global__ __launch_bounds__(256) void mc_faces_kernel(McArgs a, const CellRec *__restrict__ cells, const unsigned long long *__restrict__ totals,
                                                       const int32_t *__restrict__ edge_map, int32_t *__restrict__ faces)
{
    __shared__ uint32_t tab[mc::BLOB_WORDS];
    if (!totals[3]) return;
    const unsigned ncells = (unsigned)totals[2];
    load_tables(a.tables, tab);
    const int yx = a.n1 * a.n2;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < ncells; i += gridDim.x * 256u) {
        const CellRec r = cells[i];
        const int n = 3 * row_ntri(tab, (int)r.row);
        int32_t *out = faces + 3 * (size_t)r.face0;
        // the library emits (a, b, c), its wrapper flips for gradient_direction='descent', the reference flips back
        // (recon_util.py:69: faces[:, [2, 1, 0]]) => the tiling order itself
        for (int k = 0; k < n; ++k) {
            const int ed = row_edge(tab, (int)r.row, k);
            int32_t id = r.cvid;
            if (ed != 12) {
                int dx, dy, dz, axis;
                edge_base(ed, dx, dy, dz, axis);
                id = edge_map[3 * ((int64_t)r.li + (int64_t)dz * yx + dy * a.n2 + dx) + axis];
            }
            out[k] = id;
        }
    }
}

}  // namespace

int recon_mesh(avc_ctx *ctx, const float *vol, const int32_t res[3], const float bounds[6], float iso,
               float *verts, float *normals, int32_t *faces, int64_t cap_v, int64_t cap_f, int64_t counts[2], hipStream_t s)
{
    McArgs a{};
    a.vol = vol; a.n0 = res[0]; a.n1 = res[1]; a.n2 = res[2];
    a.N = (int64_t)a.n0 * a.n1 * a.n2; a.iso_f = iso; a.iso = (double)iso;
    a.ntiles = (int)((a.N + TILE - 1) / TILE);
    if (!ctx->mc_tables_dev) {
        AVC_HIP(hipMalloc((void **)&ctx->mc_tables_dev, sizeof(mc::BLOB)));
        AVC_HIP(hipMemcpy(ctx->mc_tables_dev, mc::BLOB, sizeof(mc::BLOB), hipMemcpyHostToDevice));
    }
    a.tables = ctx->mc_tables_dev;
    // scratch: 3 x tile sums + the list of crossed tiles [ntiles each], totals[4] (u64) + the list's length, the cells' cube indices (1 byte) and rows
    // (int16) per grid point of whole tiles, edge map 3 x int32 per grid point (sparsely written, never cleared)
    const size_t off_t = sizeof(unsigned) * (size_t)a.ntiles;
    const size_t off_tot = (4 * off_t + 15) & ~(size_t)15;
    const size_t off_idx = off_tot + 48;
    const size_t off_rows = (off_idx + (size_t)a.ntiles * TILE + 15) & ~(size_t)15;
    const size_t off_map = (off_rows + sizeof(int16_t) * (size_t)a.ntiles * TILE + 15) & ~(size_t)15;
    const size_t need = off_map + sizeof(int32_t) * 3 * (size_t)a.N;
    if (need > ctx->mc_scratch_bytes) {
        AVC_HIP(hipStreamSynchronize(s));
        if (ctx->mc_scratch) AVC_HIP(hipFree(ctx->mc_scratch));
        ctx->mc_scratch = nullptr; ctx->mc_scratch_bytes = 0;
        AVC_HIP(hipMalloc(&ctx->mc_scratch, need));
        ctx->mc_scratch_bytes = need;
    }
    char *base = (char *)ctx->mc_scratch;
    unsigned *tile_v = (unsigned *)base, *tile_t = (unsigned *)(base + off_t), *tile_c = (unsigned *)(base + 2 * off_t);
    unsigned *list = (unsigned *)(base + 3 * off_t);
    unsigned long long *totals = (unsigned long long *)(base + off_tot);
    unsigned *list_n = (unsigned *)(base + off_tot + 32);
    uint8_t *idx8 = (uint8_t *)(base + off_idx);
    int16_t *rows16 = (int16_t *)(base + off_rows);
    int32_t *edge_map = (int32_t *)(base + off_map);
    // what the later passes may write: the caller's capacities (none without output buffers); a crossed cell has at least one triangle, so cap_f bounds
    // the cell records too
    const unsigned long long eff_v = verts ? (unsigned long long)std::max<int64_t>(cap_v, 0) : 0ull;
    const unsigned long long eff_f = (verts && faces) ? (unsigned long long)std::max<int64_t>(cap_f, 0) : 0ull;
    const size_t cell_bytes = sizeof(CellRec) * (size_t)eff_f;
    if (cell_bytes > ctx->mc_cells_bytes) {
        AVC_HIP(hipStreamSynchronize(s));
        if (ctx->mc_cells) AVC_HIP(hipFree(ctx->mc_cells));
        ctx->mc_cells = nullptr; ctx->mc_cells_bytes = 0;
        AVC_HIP(hipMalloc(&ctx->mc_cells, cell_bytes + 4096));
        ctx->mc_cells_bytes = cell_bytes + 4096;
    }
    CellRec *cells = (CellRec *)ctx->mc_cells;
    EmitArgs e{};
    for (int c = 0; c < 3; ++c) {
        e.b0[c] = bounds[c];
        e.len[c] = bounds[3 + c] - bounds[c];
        e.vox[c] = e.len[c] / (float)res[c];
    }

    const int grid = std::min(a.ntiles, ctx->num_cus * 8), lgrid = std::min(a.ntiles, ctx->num_cus * 4);
    // whole x rows of one z plane per tile, 16-byte aligned rows: the walking classify (see its header); anything else: the general one
    const bool walk = a.n2 >= 4 && TILE % a.n2 == 0 && a.n1 % (TILE / a.n2) == 0 && (reinterpret_cast<uintptr_t>(a.vol) & 15) == 0 && ctx->opt.mc_walk;
    if (walk) {
        const int TP = a.n1 / (TILE / a.n2);
        int ZS = 8;                                            // planes per workgroup: 1 / ZS extra reads at the start of a segment against workgroups to fill the chip
        while (ZS > 2 && TP * ((a.n0 + ZS - 1) / ZS) < ctx->num_cus * 4) ZS /= 2;
        const int wgrid = std::min(TP * ((a.n0 + ZS - 1) / ZS), ctx->num_cus * 8);
        if (ZS == 8) hipLaunchKernelGGL(mc_classify_walk_kernel<8>, dim3(wgrid), dim3(256), 0, s, a, tile_v, tile_t, tile_c, idx8);
        else if (ZS == 4) hipLaunchKernelGGL(mc_classify_walk_kernel<4>, dim3(wgrid), dim3(256), 0, s, a, tile_v, tile_t, tile_c, idx8);
        else hipLaunchKernelGGL(mc_classify_walk_kernel<2>, dim3(wgrid), dim3(256), 0, s, a, tile_v, tile_t, tile_c, idx8);
    } else {
        hipLaunchKernelGGL(mc_classify_kernel, dim3(grid), dim3(256), 0, s, a, tile_v, tile_t, tile_c, idx8);
    }
    hipLaunchKernelGGL(mc_compact_kernel, dim3(1), dim3(1024), 0, s, tile_c, a.ntiles, list, list_n);
    hipLaunchKernelGGL(mc_count_kernel, dim3(lgrid), dim3(256), 0, s, a, list, list_n, idx8, tile_v, tile_t, tile_c, rows16);
    hipLaunchKernelGGL(mc_scan_kernel, dim3(1), dim3(1024), 0, s, tile_v, tile_t, tile_c, a.ntiles, eff_v, eff_f, totals);
    if (eff_v && eff_f) {
        hipLaunchKernelGGL(mc_verts_kernel, dim3(grid), dim3(256), 0, s, a, e, tile_v, tile_t, tile_c, totals, list, list_n, rows16, edge_map, cells, verts);
        const int egrid = (int)std::min<unsigned long long>((eff_v + 255ull) / 256ull, (unsigned long long)ctx->num_cus * 16ull);
        hipLaunchKernelGGL(mc_eval_kernel, dim3(egrid), dim3(256), 0, s, a, e, totals, verts, normals);
        const int fgrid = (int)std::min<unsigned long long>((eff_f + 255ull) / 256ull, (unsigned long long)ctx->num_cus * 8ull);
        hipLaunchKernelGGL(mc_faces_kernel, dim3(fgrid), dim3(256), 0, s, a, cells, totals, edge_map, faces);
    }
    unsigned long long h_tot[4];
    AVC_HIP(hipMemcpyAsync(h_tot, totals, sizeof h_tot, hipMemcpyDeviceToHost, s));
    AVC_HIP(hipStreamSynchronize(s));
    counts[0] = (int64_t)h_tot[0]; counts[1] = (int64_t)h_tot[1];
    AVC_REQUIRE(counts[0] <= cap_v && counts[1] <= cap_f, AVC_ERR_CAPACITY,
                "avc_recon_mesh: need capacity for %lld vertices / %lld faces, got %lld / %lld",
                (long long)counts[0], (long long)counts[1], (long long)cap_v, (long long)cap_f);
    if (counts[1] == 0) return AVC_OK;
    AVC_REQUIRE(counts[0] < ((int64_t)1 << 31) && counts[1] < ((int64_t)1 << 31), AVC_ERR_ARG, "avc_recon_mesh: mesh too large for 32-bit indices");
    AVC_REQUIRE(verts && faces, AVC_ERR_ARG, "avc_recon_mesh: verts/faces output is NULL");
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

}  // namespace avc
