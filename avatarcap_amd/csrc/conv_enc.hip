// The image encoder of ReconNetwork on gfx950 (MI355X): PIFu's stacked hourglass as the reference configures it,
//   HGFilter(stack 1, depth 4, in 6, out 32, GroupNorm(32), 'no_down', no tanh)      network/HGFilters.py:124-219
//   ConvBlock (three pre-activated 3x3 convs c/2, c/4, c/4, concatenated, + residual)  network/HGFilters.py:33-75
//   HourGlass (avg_pool2d down, bicubic align_corners=True up)                          network/HGFilters.py:77-121
// 232.3 GFLOP per 512^2 frame.  Rounds 1-3 ran it on MIOpen (~200 launches, 5.3 ms of kernels, 1-2 % of the MFMA peak).
//
// Design:
//   * activations live in HBM channel-last (H, W, C) fp32, RAW (pre-normalisation): a GroupNorm needs the statistics of the whole
//     tensor, so it cannot be applied by the producer; the CONSUMER applies y = relu(a_c x + b_c) while it stages its input tile,
//     with (a_c, b_c) from the group statistics the producer left behind.
//   * every convolution is an implicit GEMM  D[pixel][co] = sum_{tap, ci} X[pixel + tap][ci] W[co][ci][tap]  on
//     v_mfma_f32_32x32x16_f16, A = activations (rows = 32 pixels of one image row), B = weights (columns = 32 output channels),
//     fp32 products as three fp16 passes (hi*hi + hi*lo + lo*hi, fp32 accumulation -- the arithmetic of fused_mlp.hip).
//     One workgroup = 4 waves = (4 PT * 32 / TWC) image rows x TWC columns x (32 CT) output channels, K walked as
//     (32-channel chunk) x (tap) steps of two k-steps each:
//       - the chunk's halo tile goes HBM -> registers -> GroupNorm + ReLU + fp16 split -> LDS ([pixel][32 hi | 32 lo | pad], 144 B per
//         pixel: a lane's 8 consecutive channels are one ds_read_b128 and 32 lanes on consecutive pixels hit all banks once),
//         double-buffered one chunk ahead;
//       - the step's weights (pre-split, pre-ordered fragments: pack_conv) travel L2 -> LDS by buffer LDS-DMA into a 3-slot ring two
//         steps ahead; one barrier per step.
//   * the epilogue writes the raw output and/or its slice of the block's concatenated output + residual, and accumulates the
//     GroupNorm statistics of both: with D transposed (pixels in rows) a lane owns one channel and 16 PT pixels of it, so the pixel
//     sum is in-lane; per-workgroup group partials go to HBM with plain stores and the CONSUMER's prologue folds them in a fixed order, in
//     double, into (mean, rstd) per group: deterministic, no atomics of any kind, no extra launch, nothing left to wait for at a launch's end.
//   * avg_pool2d and bicubic-upsample + add are two small HBM-bound kernels that also leave statistics behind.
//   * the whole encoder (~65 launches) is recorded once per input size as a hipGraph and replayed per frame.
#include "store_settle.h"
#include <hip/hip_runtime.h>
#include <thread>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdint.h>
#include <type_traits>
#include <utility>

#include "avcap_internal.h"

namespace avc {
namespace enc {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

__device__ f32x4 raw_buffer_load_f32x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// ---- LDS map of conv_mfma_kernel ---------------------------------------------------------------------------------------
constexpr int PIXB = 144;                       // bytes per staged pixel: 32 channels x (hi, lo) halves + 16 pad (odd multiple of 16: conflict-free b128 reads)
constexpr int RING_SLOT = 16384;                // one slot of the weight ring: a group of 4 / CT taps
constexpr int MAX_CIN = 1024;               // input channels of one convolution (the U-Net's space-to-depth layers reach 1024)
constexpr int MAX_NORM_CIN = 256;           // ... of one whose input goes through a GroupNorm: its (a, b) per channel sit in LDS

// GroupNorm statistics (torch.nn.GroupNorm: mean and biased variance over (C / G, H, W)) travel from the launch that produces a tensor to the
// launches that consume it as ONE fp32 (sum, sum of squares) pair per (group, tile of the producing launch): `part[group][ntiles]`.  A producing
// workgroup stores the pairs of its tile's groups with plain stores and is done -- no ticket, no atomic, no device-scope round trip at the end of
// the launch (rounds 4-5 folded tiles into <= 32 buckets through per-bucket tickets: two dependent device-scope round trips, 3 - 5 us at the end of
// every one of the encoder's ~65 launches; `profiles/r06_enc_phases.md`).  The CONSUMER is a later launch, so the kernel boundary orders the
// stores; its prologue folds the tiles of a group with eight threads per group, in double, in a fixed order (fold_groups): the loads fly beside the
// first activation tile's, and the result is bitwise deterministic (fixed tile order, fixed tree).
struct StatOut {
    f32x2 *part;         // [rows][ntiles] (sum, sum of squares) per (group, tile), fp32; the launch's first row first.  null = the launch leaves no statistics of this kind
    int cpg;             // channels per group
    int ntiles;          // row pitch of the table (>= the tiles of any launch that produces the tensor; entries no launch writes stay zero)
};

// A destination of a convolution's output in one of the layouts the U-Net's layers hand to each other (all channel-last):
//   OUT_NORMAL  pixel (y, x), channel coff + co of a (H, W, C) tensor -- a slice of the tensor a torch.cat would have built
//   OUT_S2D     the space-to-depth form the next stride-2 convolution reads: pixel (y / 2, x / 2), channel ((y & 1) 2 + (x & 1)) Cout + co of (H/2, W/2, 4 Cout)
//   OUT_D2S     a transposed convolution computed as a 3x3 convolution with 4 Cq parity-major output channels: output channel par Cq + c of pixel (y, x) is
//               channel coff + c of pixel (2 y + par / 2, 2 x + par % 2) of a (2 H, 2 W, C) tensor
enum { OUT_NONE = 0, OUT_NORMAL = 1, OUT_S2D = 2, OUT_D2S = 3 };
struct OutSpec { float *ptr; int layout, C, coff; };

struct ConvArgs {
    const float *x;              // input (H, W, Cin) channel-last
    int H, W, Cin;
    const f32x2 *in_part;        // the producers' [group][in_nt] partials of x (NORM launches)
    int in_nt; float in_inv_n, in_eps;           // tiles of the producing launches; 1 / (cpg H W), the GroupNorm's eps
    const float *gamma, *beta;   // the consumer's GroupNorm affine
    int in_cpg;
    float in_scale;              // power of two folded into (a, b): keeps small activations' lo halves normal
    float in_slope;              // raw inputs (NORM == false): x <- max(x, in_slope x) while staging: 1 none, 0 ReLU, 0.2 LeakyReLU(0.2) (the U-Net's pre-activations)
    const char *wstream;         // packed weights: slices of 32 CT output channels, each [chunk][tap][k-step][tile][hi 1 KiB | lo 1 KiB]
    unsigned wbytes, slice_bytes;
    const float *bias;           // (Cout) or null
    float out_scale;             // undoes the weight and activation scales
    int Cout;
    float *raw;                  // (H, W, Cout) or null
    float *y;                    // (H, W, yC): channels [ycoff, ycoff + Cout) <- conv + res[same channels], or null
    const float *res;            // (H, W, yC)
    int yC, ycoff;
    StatOut st_raw, st_y;
    int tiles_x, tiles_y;
    int ksplit;                  // workgroups per (tile, slice): each walks Cin / 32 / ksplit chunks of K (1: no split)
    float *kpart;                // split-K: [tile x slice][ksplit][accumulator registers][256 threads] raw sums
    unsigned *kcounter;          // split-K: one ticket per (tile, slice)
    unsigned *range_flag;        // set to 1 when a staged value exceeds the fp16 range of the split (avc_set_range_check reads it)
    OutSpec oa, ob;              // generic outputs (the U-Net's layouts); when oa.ptr != null they replace raw / y and no statistics are produced
#ifdef AVC_ENC_PHASES
    unsigned long long *phases;  // tools/ubench/enc_bench: [workgroup][8] s_memtime stamps (start, prologue done, main loop done, outputs stored, end)
#endif
    int tap_mode, tap_div;       // TAPS == 4: where the 2 x 2 taps start in the 3 x 3 neighbourhood: 1 = by input parity (chunk / tap_div: stride-2 convolution
                                 // of a space-to-depth tensor, origin (1 - py, 1 - px)), 2 = by output parity (output channel / tap_div: transposed convolution, (a, b))
};

__device__ __forceinline__ float relu_bits(float x)
{
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// (x0, x1) -> packed fp16 pair hi = RN(x), lo = RN(x - hi)   (fused_mlp.hip split2)
__device__ __forceinline__ void split2(float x0, float x1, unsigned &hi, unsigned &lo)
{
    const half2_t hv = {(_Float16)x0, (_Float16)x1};
    hi = __builtin_bit_cast(unsigned, hv);
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(lo)
        : "v"(hi), "v"(x0), "v"(x1));
}

// D-tile register r (0..15) of lane-half h -> row of the 32x32 tile (CDNA4 C/D map; mlp_layout.h d_row)
__device__ __forceinline__ constexpr int d_row0(int r) { return (r & 3) + 8 * (r >> 2); }      // + 4 h

// ---- statistics ------------------------------------------------------------------------------------------------------------------
// The consumer's side.  (mean, rstd) of the `ngroups` (<= 32) groups of a tensor from its [group][ntiles] partials into LDS (`mr[group]`), by all 256
// threads of the workgroup: thread t sums the tiles k = t % 8, t % 8 + 8, ... of group t / 8 in double (fixed order), the eight partial sums of a group
// meet in a three-step butterfly (a fixed tree).  `issue` runs between the issue of the loads and their first use: whatever else the caller wants in
// flight meanwhile.  Ends with a __syncthreads().
template <class F>
__device__ __forceinline__ void fold_groups(const f32x2 *__restrict__ part, int ngroups, int ntiles, float inv_n, float eps, f32x2 *mr, F &&issue)
{
    const int tid = threadIdx.x, g = tid >> 3, sub = tid & 7;
    const f32x2 *row = part + (size_t)(g < ngroups ? g : 0) * ntiles;
    double S = 0.0, Q = 0.0;
    constexpr int B = 32;                                  // loads in flight per thread: ALL of a launch of <= 256 tiles in one round trip (512 tiles: two)
    f32x2 v[B];
#pragma unroll
    for (int k = 0; k < B; ++k) { const int t = sub + 8 * k; v[k] = t < ntiles ? row[t] : f32x2{0.0f, 0.0f}; }
    issue();
#pragma unroll
    for (int k = 0; k < B; ++k) { S += (double)v[k][0]; Q += (double)v[k][1]; }
    for (int t0 = 8 * B; t0 < ntiles; t0 += 8 * B) {
#pragma unroll
        for (int k = 0; k < B; ++k) { const int t = t0 + sub + 8 * k; v[k] = t < ntiles ? row[t] : f32x2{0.0f, 0.0f}; }
#pragma unroll
        for (int k = 0; k < B; ++k) { S += (double)v[k][0]; Q += (double)v[k][1]; }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { S += __shfl_xor(S, o); Q += __shfl_xor(Q, o); }
    if (sub == 0 && g < ngroups) {
        const double m = S * (double)inv_n, var = fmax(Q * (double)inv_n - m * m, 0.0);
        mr[g] = f32x2{(float)m, (float)(1.0 / sqrt(var + (double)eps))};
    }
    __syncthreads();
}

// The producer's side.  Thread tid < nrows holds (s, q) of row row0 + tid of the launch's table for this workgroup's tile.
struct StatRows { int row0, nrows; float s, q; };
__device__ __forceinline__ void stats_store(const StatOut &st, const StatRows &r, int tile, int tid)
{
    if (st.part && tid < r.nrows) st.part[(size_t)(r.row0 + tid) * st.ntiles + tile] = f32x2{r.s, r.q};
}

// ---- the convolution ---------------------------------------------------------------------------------------------------
// CT: 32-channel output tiles per workgroup (B operands), PT: 32-pixel tiles per wave (A operands), TAPS: 9 (3x3, pad 1) or 1,
// TWC: image columns of a pixel tile (32: one image row; 16: two rows of 16 -- images narrower than 32),
// NORM: the input goes through relu(GroupNorm(.)) while it is staged (else: raw).
// LDS map (per instantiation): [staged chunk A | staged chunk B | weight ring of RS 16-KiB slots | (a, b) table | flag].
// OCC2: TWO workgroups per CU.  A launch of one workgroup per CU runs its phases in lockstep on every CU -- the prologue's fetch burst (every workgroup asks for
// its first activation tile and weight groups at once: ~75 KB per CU at the ~11 B/cycle/CU such a burst gets, 12 - 15 k cycles), the main loop, the output burst --
// and the matrix pipe idles through the first and the last (profiles/r06_enc_phases.md: 46 - 58 % of a large launch's cycles are its main loop).  With half the LDS
// each (ONE staged chunk instead of two: the next chunk waits in registers and is written between two chunks, which is when the OTHER workgroup's MFMAs run) two
// workgroups of half the pixel tile share a CU and fill each other's memory phases.
template <int PT, int TAPS, int TWC, bool OCC2 = false>
struct ConvGeo {
    static constexpr int PTR = 32 / TWC;                          // image rows of one pixel tile
    static constexpr int ROWS = 4 * PT * PTR;                     // image rows of the workgroup's tile
    // halo: 3x3 pad 1; 4x4 (the space-to-depth form of conv1's 7x7 stride 2 pad 3) 2 before and 1 after; 1x1 none; TAPS == 4: the 2 x 2 taps of a 3 x 3
    // neighbourhood that a stride-2 4 x 4 kernel leaves non-zero for one input (output) parity -- the 3 x 3 halo, walked from a runtime origin
    static constexpr int PAD = (TAPS == 9 || TAPS == 4) ? 1 : (TAPS == 16 ? 2 : 0), PADH = (TAPS == 9 || TAPS == 4) ? 1 : (TAPS == 16 ? 1 : 0);
    static constexpr int KW = TAPS == 9 ? 3 : (TAPS == 16 ? 4 : (TAPS == 4 ? 2 : 1));
    static constexpr int HR = ROWS + PAD + PADH, HC = TWC + PAD + PADH;
    static constexpr int RP = TAPS == 1 ? TWC : (TWC == 32 ? HC : 32);       // LDS row pitch in pixels (32 for the 16-wide tiles: bank note in DESIGN.md)
    static constexpr int NPIX = HR * RP;
    static constexpr int ACTB = (NPIX * PIXB + 1023) & ~1023;
    static constexpr int NBUF = OCC2 ? 1 : 2, LDS_BUDGET = OCC2 ? 81920 : 163840;
    static constexpr int RS_FIT = (LDS_BUDGET - 8 * MAX_NORM_CIN - 64 - 256 - NBUF * ACTB) / RING_SLOT;
    static constexpr int RS = RS_FIT > 6 ? 6 : RS_FIT;            // ring slots; RS - 1 groups of weights are in flight
    static constexpr int L_ACT0 = 0, L_ACT1 = OCC2 ? 0 : ACTB, L_RING = NBUF * ACTB, L_AB = L_RING + RS * RING_SLOT, L_FLAG = L_AB + 8 * MAX_NORM_CIN, L_TOTAL = L_FLAG + 64 + 256;     // (flag, then (mean, rstd) of the input's 32 groups)
    static_assert(RS >= 3, "no room for the weight ring");
};

#ifdef AVC_ENC_PHASES
#define AVC_PHASE(k) do { if (p.phases && threadIdx.x == 0) p.phases[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AVC_PHASE(k) do { } while (0)
#endif

template <int CT, int PT, int TAPS, int TWC, bool NORM, bool OCC2 = false>
__global__ __launch_bounds__(256, OCC2 ? 2 : 1) void conv_mfma_kernel(const ConvArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    AVC_PHASE(0);
    using Geo = ConvGeo<PT, TAPS, TWC, OCC2>;
    constexpr int PTR = Geo::PTR, ROWS = Geo::ROWS, PAD = Geo::PAD, RP = Geo::RP, HC = Geo::HC, NPIX = Geo::NPIX, KW = Geo::KW;
    constexpr int LDS_ACT0 = Geo::L_ACT0, LDS_ACT1 = Geo::L_ACT1, LDS_RING = Geo::L_RING, LDS_AB = Geo::L_AB, LDS_FLAG = Geo::L_FLAG;
    constexpr int RS = Geo::RS, LA = RS - 1;                // ring slots, groups of weights in flight
    constexpr int NPIECE = (NPIX * 8 + 255) / 256;         // 16-byte pieces (4 channels of one pixel) per thread and chunk

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int slices = p.Cout / (32 * CT);
    // blockIdx -> (k slice, channel slice, pixel tile): a workgroup walks the chunks [c0, c1) of the input channels (split-K launches: ksplit > 1)
    // (a band of the image per XCD -- workgroup b on XCD b % 8 taking the tiles of ITS eighth, so that vertically adjacent tiles share their halo rows in one
    // L2 -- was measured and changes nothing: 46.3 / 46.9, 136.5 / 135.6, 51.8 / 51.1 us; the input fetch is not what a launch waits for, profiles/r06_enc_phases.md)
    // (cutting a launch into two launches of every other tile on two streams, so that the halves start a launch latency apart, costs its fork and join: +15 - 20 us
    // on every shape tried, profiles/r06_enc_phases.md)
    const int ks = blockIdx.x % p.ksplit, wg = blockIdx.x / p.ksplit;
    const int slice = wg % slices, tile = wg / slices;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const int y0 = ty * ROWS, x0 = tx * TWC;
    const int cpk = (p.Cin >> 5) / p.ksplit, c0 = ks * cpk, c1 = c0 + cpk;

    // ---- staging geometry of this thread: piece i = 4 channels `sub` of staged pixel sp = tid / 8 + 32 i
    const int sub = tid & 7;
    int goff[NPIECE];                                      // byte offset of the piece in x for chunk 0; negative = outside the image (zero padding)
    {
        int sp = tid >> 3;
        int hr = sp / RP, hc = sp - hr * RP;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const int iy = y0 - PAD + hr, ix = x0 - PAD + hc;
            const bool ok = (i * 32 + (tid >> 3)) < NPIX && hc < HC && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            goff[i] = ok ? ((iy * p.W + ix) * p.Cin + sub * 4) * 4 : (int)0x80000000;
            hc += 32;                                      // next piece: 32 staged pixels on (RP is 32, 34 or 35: at most one row wrap)
            if (hc >= RP) { hc -= RP; ++hr; }
            if (RP < 32 && hc >= RP) { hc -= RP; ++hr; }  // RP == 16 (1x1 on the narrow tile)
        }
    }
    i32x4 xrs;
    {
        const unsigned long long a = reinterpret_cast<unsigned long long>(p.x);
        xrs[0] = (int)(unsigned)a; xrs[1] = (int)((unsigned)(a >> 32) & 0xffffu);
        xrs[2] = p.H * p.W * p.Cin * 4; xrs[3] = 0x00027000;
    }
    f32x4 stage[NPIECE];
    float amax = 0.0f;                                     // largest magnitude that went through the fp16 split
    auto load_acts = [&](int c) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) stage[i] = raw_buffer_load_f32x4(xrs, goff[i], c * 128, 0);
    };
    // pieces [i0, i1) of the staged chunk c: affine + ReLU + split -> LDS buffer `buf`
    auto store_acts = [&](int c, unsigned buf, int i0, int i1) {
        f32x4 ab0 = {p.in_scale, 0.0f, p.in_scale, 0.0f}, ab1 = ab0;
        if constexpr (NORM) {
            ab0 = *reinterpret_cast<const f32x4 *>(smem + LDS_AB + (c * 32 + sub * 4) * 8);
            ab1 = *reinterpret_cast<const f32x4 *>(smem + LDS_AB + (c * 32 + sub * 4) * 8 + 16);
        }
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            if (i < i0 || i >= i1) continue;
            if ((i + 1) * 32 > NPIX && i * 32 + (tid >> 3) >= NPIX) continue;
            float v0 = ab0[0] * stage[i][0] + ab0[1], v1 = ab0[2] * stage[i][1] + ab0[3];
            float v2 = ab1[0] * stage[i][2] + ab1[1], v3 = ab1[2] * stage[i][3] + ab1[3];
            if constexpr (NORM) { v0 = relu_bits(v0); v1 = relu_bits(v1); v2 = relu_bits(v2); v3 = relu_bits(v3); }
            else {
                v0 = __builtin_fmaxf(v0, p.in_slope * v0); v1 = __builtin_fmaxf(v1, p.in_slope * v1);
                v2 = __builtin_fmaxf(v2, p.in_slope * v2); v3 = __builtin_fmaxf(v3, p.in_slope * v3);
            }
            amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v0), __builtin_fabsf(v1)), __builtin_fmaxf(__builtin_fabsf(v2), __builtin_fabsf(v3))));
            unsigned h01, l01, h23, l23;
            split2(v0, v1, h01, l01);
            split2(v2, v3, h23, l23);
            const bool in = goff[i] >= 0;                  // zero padding is applied to the normalised tensor: out-of-image pieces are zeros, not relu(b)
            h01 = in ? h01 : 0u; l01 = in ? l01 : 0u; h23 = in ? h23 : 0u; l23 = in ? l23 : 0u;
            const unsigned a = buf + ((tid >> 3) + 32 * i) * PIXB + sub * 8;
            *reinterpret_cast<u32x2 *>(smem + a) = u32x2{h01, h23};
            *reinterpret_cast<u32x2 *>(smem + a + 64) = u32x2{l01, l23};
        }
    };

    // ---- weights: buffer LDS-DMA into the ring.  A ring slot (16 KiB) holds a GROUP of G = 4 / CT consecutive taps of one chunk (the stream
    // is [chunk][tap][k-step][tile], so a group is contiguous); groups restart with every chunk (3x3: CT 4 -> 9 groups of 1 tap, CT 2 -> 5 of
    // 2, 2, 2, 2, 1, CT 1 -> 3 of 4, 4, 1), one barrier per group, LA = RS - 1 groups in flight: the small-tile kernels are bound by how many
    // bytes of weights a CU has in flight (their workgroups are few and each streams its slice once, mostly from HBM).
    // Wave w moves the w-th quarter of a group, 1 KiB per instruction.
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(p.wstream), 0, (int)p.wbytes, 0x00027000);
    const unsigned wbase = slice * p.slice_bytes;
    const unsigned lane16 = lane * 16u;
    constexpr int TAP_BYTES = 2 * CT * 2048;                // one (chunk, tap): 2 k-steps x CT tiles x [hi | lo]
    constexpr int G = TAPS == 1 ? 1 : 4 / CT, NG = (TAPS + G - 1) / G;
    auto group_taps = [](int g) constexpr { return (g + 1) * G <= TAPS ? G : TAPS - g * G; };
    const int ngroups = cpk * NG;
    // group gi = (c - c0) NG + g (g compile-time) -> ring slot gi % RS
    auto dma_group = [&](auto gc, int gi) {
        constexpr int g = decltype(gc)::value, PW = group_taps(g) * CT;                   // 1 KiB pieces per wave
        const int c = c0 + gi / NG;
        const unsigned so = wbase + (unsigned)(c * TAPS + g * G) * TAP_BYTES + wave * (PW * 1024);
        const unsigned dst = LDS_RING + (unsigned)(gi % RS) * RING_SLOT + wave * (PW * 1024);
        static_for<PW>([&](auto ic) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void *)(smem + dst), 16, (int)lane16, (int)so, decltype(ic)::value * 1024, 0);
        });
    };
    auto dma_any = [&](int gi) {                            // run-time group residue
        const int g = gi % NG;
        static_for<NG>([&](auto gc) { if (g == decltype(gc)::value) dma_group(gc, gi); });
    };
    // "group gi + 1 has landed": in-order completion -> at most the operations issued after its last piece may still be outstanding: the groups
    // gi + 2 .. gi + min(LA, groups left) (g = gi % NG compile-time, `left` = groups after gi) and, when they were issued behind it, the staging
    // loads of the next chunk
    auto wait_next = [&](auto gc, int left, bool staged) {
        constexpr int g = decltype(gc)::value;
        bool done = false;
        static_for<LA>([&](auto kc) {
            constexpr int k = LA - decltype(kc)::value;     // LA, LA - 1, ..., 1
            constexpr int n = [&] { int a = 0; for (int i = 2; i <= k; ++i) a += group_taps((g + i) % NG) * CT; return a; }();
            if (!done && left >= k) {
                done = true;
                if (staged) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(n + NPIECE) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(n) : "memory");
            }
        });
        if (!done) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };

    // ---- A-operand addresses: pixel tile n of this wave = tile row q = wave PT + n; lane (j, h) reads 8 channels of pixel j
    unsigned abase[PT];
#pragma unroll
    for (int n = 0; n < PT; ++n) {
        const int q = wave * PT + n, r = j / TWC, cc = j - r * TWC;
        abase[n] = (unsigned)(((q * PTR + r) * RP + cc) * PIXB + h * 16);
    }

    // ---- prologue.  Everything the workgroup has to fetch before its first MFMA is requested up front, in one burst: the first activation tile, the first
    // weight group, the GroupNorm's per-channel affine and the producers' per-tile statistics (fold_groups) -- one memory round trip instead of the four
    // dependent ones of rounds 4-5 (statistics -> gamma / beta -> activations -> weights: 11.6 k cycles before the first MFMA, profiles/r06_enc_phases.md).
    load_acts(c0);
    dma_any(0);
    // the affine map per input channel: relu(a x + b), a = gamma rstd, b = beta - mean a (both times in_scale); raw inputs: a = in_scale, b = 0, no table
    if constexpr (NORM) {
        float ga = 0.0f, be = 0.0f;
        if (tid < p.Cin) { ga = p.gamma[tid]; be = p.beta[tid]; }
        f32x2 *mr = reinterpret_cast<f32x2 *>(smem + LDS_FLAG + 64);
        fold_groups(p.in_part, p.Cin / p.in_cpg, p.in_nt, p.in_inv_n, p.in_eps, mr, [] {});
        if (tid < p.Cin) {
            const f32x2 m = mr[tid / p.in_cpg];
            const float a = ga * m[1];
            *reinterpret_cast<f32x2 *>(smem + LDS_AB + tid * 8) = f32x2{a * p.in_scale, (be - m[0] * a) * p.in_scale};
        }
    }
    __syncthreads();                                        // the (a, b) table
    store_acts(c0, LDS_ACT0, 0, NPIECE);
    for (int gi = 1; gi < LA && gi < ngroups; ++gi) dma_any(gi);
    wait_next(std::integral_constant<int, NG - 1>{}, ngroups < LA ? ngroups : LA, false);        // group 0 (= "group -1 + 1")
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    AVC_PHASE(1);
    f32x16 acc[PT][CT];
#pragma unroll
    for (int n = 0; n < PT; ++n)
#pragma unroll
        for (int m = 0; m < CT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.0f;

    // the next chunk's staged tile is transformed and written late: consuming a staging load waits for every weight group requested before it
    // (vmcnt completes in order), so it is consumed where those groups are due anyway: from the second group on when two are in flight, in the
    // chunk's last group otherwise
    constexpr int T0 = (TAPS == 1 || NG == 1) ? 0 : (LA == 2 ? G : G * (NG - 1));
    // one chunk of K; `more` (a next chunk exists: its tile is fetched, transformed and written while this one is multiplied) is a compile-time flag --
    // as a run-time `if` it cut the loop body into basic blocks at every tap, and the staging arithmetic could not be scheduled between the MFMAs
    // (profiles/r06_enc_phases.md: 21 % of the main loop)
    auto chunk = [&](const int c, auto more_c) {
        constexpr bool more = decltype(more_c)::value;
        const unsigned abuf = ((c - c0) & 1) ? LDS_ACT1 : LDS_ACT0, nbuf = ((c - c0) & 1) ? LDS_ACT0 : LDS_ACT1;
        unsigned org = 0;
        if constexpr (TAPS == 4) {
            const int par = p.tap_mode == 1 ? c / p.tap_div : (int)(slice * CT * 32) / p.tap_div;
            const int ty0 = p.tap_mode == 1 ? 1 - (par >> 1) : (par >> 1), tx0 = p.tap_mode == 1 ? 1 - (par & 1) : (par & 1);
            org = (unsigned)((ty0 * RP + tx0) * PIXB);
        }
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value, NT = group_taps(g);
            const int gi = (c - c0) * NG + g;
            if constexpr (g == 0 && more) load_acts(c + 1);
            // group gi + LA goes into the slot group gi - 1 was read from (every wave is past the barrier that ended it)
            if (gi + LA < ngroups) dma_group(std::integral_constant<int, (g + LA) % NG>{}, gi + LA);
            const unsigned wb = LDS_RING + (unsigned)(gi % RS) * RING_SLOT + lane16;
            static_for<NT>([&](auto tc) {
                constexpr int tl = decltype(tc)::value, t = g * G + tl;
                constexpr int toff = ((t / KW) * RP + (t % KW)) * PIXB;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    half8 bh[CT], bl[CT], ah[PT], al[PT];
#pragma unroll
                    for (int m = 0; m < CT; ++m) {
                        bh[m] = *reinterpret_cast<const half8 *>(smem + wb + ((tl * 2 + kk) * CT + m) * 2048);
                        bl[m] = *reinterpret_cast<const half8 *>(smem + wb + ((tl * 2 + kk) * CT + m) * 2048 + 1024);
                    }
#pragma unroll
                    for (int n = 0; n < PT; ++n) {
                        ah[n] = *reinterpret_cast<const half8 *>(smem + abuf + abase[n] + org + toff + kk * 32);
                        al[n] = *reinterpret_cast<const half8 *>(smem + abuf + abase[n] + org + toff + kk * 32 + 64);
                    }
#pragma unroll
                    for (int n = 0; n < PT; ++n)
#pragma unroll
                        for (int m = 0; m < CT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[n], bh[m], acc[n][m], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < PT; ++n)
#pragma unroll
                        for (int m = 0; m < CT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[n], bl[m], acc[n][m], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < PT; ++n)
#pragma unroll
                        for (int m = 0; m < CT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[n], bh[m], acc[n][m], 0, 0, 0);
                }
                // (a sched_group_barrier pipeline that pins the pieces' VALU instructions between the tap's MFMAs was measured: no change -- at one wave per SIMD
                // a VALU instruction costs MFMA issue time wherever it stands, profiles/r06_enc_phases.md)
                if constexpr (!OCC2 && more && t >= T0) store_acts(c + 1, nbuf, ((t - T0) * NPIECE) / (TAPS - T0), ((t - T0 + 1) * NPIECE) / (TAPS - T0));
            });
            // the staging loads count as "issued behind group gi + 1" while the group that was requested in their own step (group gi0 + LA)
            // is later than gi + 1
            wait_next(gc, ngroups - 1 - gi, more && g < LA - 1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        });
        if constexpr (OCC2) {
            // one staged chunk: every wave is past the chunk's last read (the barrier above); the next chunk goes from the registers into the same buffer
            // while the CU's other workgroup has the matrix pipe
            if constexpr (more) {
                store_acts(c + 1, nbuf, 0, NPIECE);
                __syncthreads();
            }
        }
    };
    for (int c = c0; c + 1 < c1; ++c) chunk(c, std::true_type{});
    chunk(c1 - 1, std::false_type{});

    AVC_PHASE(2);
    // a staged value beyond 65504 became +-inf in its `hi` half: the launch's output is not to be trusted (avc_set_range_check reports it)
    if (!(amax <= 65504.0f) && p.range_flag) atomicOr(p.range_flag, 1u);

    // ---- split-K: every k slice leaves its raw accumulators in HBM; the last one to arrive (ticket) adds all of them in slice order -- its
    // own included, re-read, so that the sum does not depend on who is last -- and goes on to the epilogue.  Device-scope accesses, no fences
    // (the partial sums cross XCDs, whose L2s are not coherent with each other).
    if (p.ksplit > 1) {
        constexpr int NV = PT * CT * 8;                        // 8-byte pieces per thread (device-scope accesses are at most 64 bits wide)
        typedef unsigned long long u64;
        u64 *mine = reinterpret_cast<u64 *>(p.kpart) + ((size_t)wg * p.ksplit + ks) * (NV * 256) + tid;
#pragma unroll
        for (int n = 0; n < PT; ++n)
#pragma unroll
            for (int m = 0; m < CT; ++m)
#pragma unroll
                for (int v = 0; v < 8; ++v) {
                    // (through named floats: __builtin_bit_cast applied to a vector-element lvalue reads element 0 -- hipcc 7.2)
                    const float a0 = acc[n][m][2 * v], a1 = acc[n][m][2 * v + 1];
                    const u64 val = (u64)__builtin_bit_cast(unsigned, a0) | ((u64)__builtin_bit_cast(unsigned, a1) << 32);
                    __hip_atomic_store(mine + ((n * CT + m) * 8 + v) * 256, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(p.kcounter + wg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *reinterpret_cast<volatile unsigned *>(smem + LDS_FLAG) = (t == (unsigned)p.ksplit - 1) ? 1u : 0u;
            if (t == (unsigned)p.ksplit - 1) __hip_atomic_store(p.kcounter + wg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!*reinterpret_cast<volatile unsigned *>(smem + LDS_FLAG)) return;
        const u64 *all = reinterpret_cast<const u64 *>(p.kpart) + (size_t)wg * p.ksplit * (NV * 256) + tid;
#pragma unroll
        for (int n = 0; n < PT; ++n)
#pragma unroll
            for (int m = 0; m < CT; ++m) {
                float lo[8], hi[8];
#pragma unroll
                for (int v = 0; v < 8; ++v) lo[v] = hi[v] = 0.0f;
                for (int kb = 0; kb < p.ksplit; kb += 8) {         // 8 slices at a time: every load of a batch is issued before the first add
                    u64 val[8][8];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
#pragma unroll
                        for (int v = 0; v < 8; ++v)
                            val[k][v] = kb + k < p.ksplit ? __hip_atomic_load(all + ((size_t)(kb + k) * NV + (n * CT + m) * 8 + v) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
                    for (int v = 0; v < 8; ++v)
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            lo[v] += __builtin_bit_cast(float, (unsigned)val[k][v]);
                            hi[v] += __builtin_bit_cast(float, (unsigned)(val[k][v] >> 32));
                        }
                }
#pragma unroll
                for (int v = 0; v < 8; ++v) { acc[n][m][2 * v] = lo[v]; acc[n][m][2 * v + 1] = hi[v]; }
            }
        __syncthreads();
    }

    // ---- epilogue: lane (j, h) owns channel co = 32 (slice CT + m) + j and, per pixel tile, the 16 pixels d_row0(r) + 4 h
    const bool full = y0 + ROWS <= p.H && x0 + TWC <= p.W;
    float sr[CT], qr[CT], sy[CT], qy[CT];
#pragma unroll
    for (int m = 0; m < CT; ++m) sr[m] = qr[m] = sy[m] = qy[m] = 0.0f;
    // per output tile m: first ALL residual values of the tile (one batch of loads in flight -- written element by element the compiler
    // serialises every load behind the previous store: it cannot know that res, raw and y do not alias, and a store is in vmcnt too), then the
    // stores.  32-bit element offsets from uniform bases (every tensor of the encoder is below 2^32 bytes).
    const float *__restrict__ resp = p.res;
    float *__restrict__ rawp = p.raw;
    float *__restrict__ yp = p.y;
    // MODE 0: partial tile (any outputs, run-time checks per element -- rare: every level of the 512^2 path tiles exactly); 1: raw + y; 2: y only;
    // 3: raw only.  The full-tile forms are straight-line code: a run-time `if (raw)` per element ends the basic block and the compiler then
    // drains vmcnt at every join.
    auto emit = [&](auto mode) {
        constexpr int MODE = decltype(mode)::value;
        constexpr bool MASK = MODE == 0;
#pragma unroll
        for (int m = 0; m < CT; ++m) {
            const unsigned co = (slice * CT + m) * 32 + j;
            const float bias = p.bias ? p.bias[co] : 0.0f;
            float rv[PT][16];
            bool ok[PT][16];
            unsigned ybase[PT], rbase[PT];
            const bool has_raw = MODE == 0 ? rawp != nullptr : (MODE == 1 || MODE == 3), has_y = MODE == 0 ? yp != nullptr : (MODE == 1 || MODE == 2);
            // element offset of register r = lane base (this lane's first pixel and channel) + dpix(r) * channels, dpix(r) wave-uniform:
            // one VALU add per element instead of a multiply-add chain and a 64-bit address each
#pragma unroll
            for (int n = 0; n < PT; ++n) {
                const int q = wave * PT + n;
                const unsigned pix0 = (unsigned)((y0 + q * PTR) * p.W + x0 + 4 * h);
                ybase[n] = pix0 * (unsigned)p.yC + (unsigned)p.ycoff + co;
                rbase[n] = pix0 * (unsigned)p.Cout + co;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int iy = y0 + q * PTR + (TWC == 32 ? 0 : (r >> 3)), ix = x0 + 4 * h + (TWC == 32 ? d_row0(r) : (d_row0(r) & 15));
                    ok[n][r] = !MASK || (iy < p.H && ix < p.W);
                    const unsigned dpix = TWC == 32 ? (unsigned)d_row0(r) : (unsigned)((r >> 3) * p.W + (d_row0(r) & 15));
                    rv[n][r] = (has_y && ok[n][r]) ? resp[ybase[n] + dpix * (unsigned)p.yC] : 0.0f;
                }
            }
#pragma unroll
            for (int n = 0; n < PT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (MASK && !ok[n][r]) continue;
                    const unsigned dpix = TWC == 32 ? (unsigned)d_row0(r) : (unsigned)((r >> 3) * p.W + (d_row0(r) & 15));
                    const float v = acc[n][m][r] * p.out_scale + bias;
                    if (has_raw) rawp[rbase[n] + dpix * (unsigned)p.Cout] = v;
                    sr[m] += v; qr[m] += v * v;
                    if (has_y) {
                        const float w = v + rv[n][r];
                        yp[ybase[n] + dpix * (unsigned)p.yC] = w;
                        sy[m] += w; qy[m] += w * w;
                    }
                }
        }
    };
    // the U-Net's outputs: any layout, no residual, no statistics; these launches are small (the whole U-Net is 10 GFLOP), the addressing is per element
    if constexpr (!NORM && (TAPS == 9 || TAPS == 4)) {     // the U-Net's layouts (only its launches set oa)
        if (p.oa.ptr) {
            // element offset = f(iy) + g(ix) + k(co) in every layout (see OutSpec); k per channel tile here, f + g per pixel below
            auto chan = [&](const OutSpec &o, unsigned co) -> int {
                if (o.layout != OUT_D2S) return o.coff + (int)co;
                const unsigned cq = (unsigned)p.Cout >> 2, par = co / cq;
                return (int)(((par >> 1) * 2 * p.W + (par & 1)) * o.C + o.coff + (co - par * cq));
            };
            auto pixel = [&](const OutSpec &o, int iy, int ix) -> int {
                if (o.layout == OUT_NORMAL) return (iy * p.W + ix) * o.C;
                if (o.layout == OUT_S2D) return ((iy >> 1) * (p.W >> 1) + (ix >> 1)) * o.C + ((iy & 1) * 2 + (ix & 1)) * (o.C >> 2);
                return (4 * iy * p.W + 2 * ix) * o.C;
            };
            int ka[CT], kb[CT]; float bias[CT];
            static_for<CT>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                const unsigned co = (slice * CT + m) * 32 + j;
                ka[m] = chan(p.oa, co); kb[m] = p.ob.ptr ? chan(p.ob, co) : 0;
                bias[m] = p.bias ? p.bias[co] : 0.0f;
            });
            static_for<PT>([&](auto nc) {
                constexpr int n = decltype(nc)::value;
                const int q = wave * PT + n;
                static_for<16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const int iy = y0 + q * PTR + (TWC == 32 ? 0 : (r >> 3)), ix = x0 + 4 * h + (TWC == 32 ? d_row0(r) : (d_row0(r) & 15));
                    if (iy < p.H && ix < p.W) {
                        const int pa = pixel(p.oa, iy, ix), pb = p.ob.ptr ? pixel(p.ob, iy, ix) : 0;
                        static_for<CT>([&](auto mc) {
                            constexpr int m = decltype(mc)::value;
                            const float v = acc[n][m][r] * p.out_scale + bias[m];
                            p.oa.ptr[pa + ka[m]] = v;
                            if (p.ob.ptr) p.ob.ptr[pb + kb[m]] = v;
                        });
                    }
                });
            });
            return;
        }
    }
    if (!full) emit(std::integral_constant<int, 0>{});
    else if (rawp && yp) emit(std::integral_constant<int, 1>{});
    else if (yp) emit(std::integral_constant<int, 2>{});
    else emit(std::integral_constant<int, 3>{});
#ifdef AVC_ENC_PHASES
    if (p.phases) { AVC_PHASE(3); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); AVC_PHASE(4); }
#endif

    // ---- statistics: the two pixel halves, then the cpg adjacent channel lanes, then the four waves through LDS
    if (p.st_raw.part || p.st_y.part) {
        float *red = reinterpret_cast<float *>(smem + LDS_ACT0);          // [kind][wave][32 CT groups max](s, q); the staged tiles are dead
        auto reduce = [&](const StatOut &st, float *s, float *q, int kind) {
            if (!st.part) return;
#pragma unroll
            for (int m = 0; m < CT; ++m) {
                s[m] += __shfl_xor(s[m], 32); q[m] += __shfl_xor(q[m], 32);
                for (int o = 1; o < st.cpg; o <<= 1) { s[m] += __shfl_xor(s[m], o); q[m] += __shfl_xor(q[m], o); }
                if (h == 0 && (j & (st.cpg - 1)) == 0) {
                    const int g = (m * 32 + j) / st.cpg;
                    *reinterpret_cast<f32x2 *>(red + ((kind * 4 + wave) * 32 * CT + g) * 2) = f32x2{s[m], q[m]};
                }
            }
        };
        reduce(p.st_raw, sr, qr, 0);
        reduce(p.st_y, sy, qy, 1);
        __syncthreads();
        auto rows = [&](const StatOut &st, int kind) {
            StatRows r{0, 0, 0.0f, 0.0f};
            if (!st.part) return r;
            r.nrows = 32 * CT / st.cpg;                                    // groups of this workgroup's slice
            r.row0 = slice * r.nrows;
            if (tid < r.nrows)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const f32x2 v = *reinterpret_cast<const f32x2 *>(red + ((kind * 4 + w) * 32 * CT + tid) * 2);
                    r.s += v[0]; r.q += v[1];
                }
            return r;
        };
        const StatRows ra = rows(p.st_raw, 0), rb = rows(p.st_y, 1);
        stats_store(p.st_raw, ra, tile, tid);
        stats_store(p.st_y, rb, tile, tid);
    }
    AVC_PHASE(5);
}

// ---- conv1: 7x7, stride 2, pad 3, 6 -> 64 channels, bias (HGFilters.py:134) -------------------------------------------------
// A stride-2 convolution is a stride-1 convolution of the space-to-depth image: (6, Hin, Win) -> (H1, W1, 24 = 2 x 2 x 6) with the 7x7 kernel
// zero-extended to 8x8 = 4x4 taps over the 24 channels (input row 2 o - 3 + k, k = 0..6, is row o - 2 + ty of parity py with k = 2 ty + py - 1).
// The 24 channels are padded to the 32 of one K chunk; conv_mfma_kernel<.., TAPS = 16, .., NORM = false> does the rest (1.3x the MACs of the
// direct form, on the matrix pipe: the fp32-FMA kernel of the first cut was 60 - 70 us, 2.5 % of the encoder's FLOPs in 3 % of its time).
struct S2dArgs { const float *img; int Hin, Win, H, W; float *out; };      // out (H, W, 32) channel-last

__global__ __launch_bounds__(256) void s2d_kernel(const S2dArgs p)
{
    const int i = blockIdx.x * 256 + threadIdx.x, quad = i & 7, pix = i >> 3;
    if (pix >= p.H * p.W) return;
    const int oy = pix / p.W, ox = pix - oy * p.W;
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ch = quad * 4 + e;                      // (py * 2 + px) * 6 + c
        if (ch < 24) {
            const int par = ch / 6, c = ch - par * 6, iy = 2 * oy + (par >> 1), ix = 2 * ox + (par & 1);
            if (iy < p.Hin && ix < p.Win) v[e] = p.img[((size_t)c * p.Hin + iy) * p.Win + ix];
        }
    }
    *reinterpret_cast<f32x4 *>(p.out + (size_t)pix * 32 + quad * 4) = v;
}

// ---- avg_pool2d(2, stride 2) + statistics of the pooled tensor (HGFilters.py:103) ------------------------------------------------
// x (H, W, C) -> out (H/2, W/2, C); thread = 4 channels, loops over pixels; a workgroup covers PPW output pixels.
struct EltArgs {
    const float *a, *b;          // pool: a = x.  upsample-add: a = up1 (H, W, C), b = low3 (H/2, W/2, C)
    float *out;
    int H, W, C;                 // OUTPUT size
    StatOut st;
    int Hb, Wb;                  // size of the other tensor (pool: the input; upsample-add: low3)
    const f32x2 *in_part; const float *gamma, *beta; int in_cpg, in_nt; float in_inv_n, in_eps;     // norm-relu: GroupNorm of a
    int ppw;                     // output pixels per workgroup
    int ntiles;
};

template <class F>
__device__ __forceinline__ void elt_body(const EltArgs &p, F &&value)
{
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x, c4n = p.C >> 2;         // C in {32..256}: c4n in {8..64}
    const int c4 = tid % c4n, sub = tid / c4n, nsub = 256 / c4n;
    const int npix = p.H * p.W, pix0 = blockIdx.x * p.ppw, pix1 = min(npix, pix0 + p.ppw);
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    for (int pix = pix0 + sub; pix < pix1; pix += nsub) {
        const int oy = pix / p.W, ox = pix - oy * p.W;
        f32x4 v = value(oy, ox, c4);
        settle(v);                                       // store_settle.h: packed-f32 results as store data
        *reinterpret_cast<f32x4 *>(p.out + (size_t)pix * p.C + 4 * c4) = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
    }
    if (!p.st.part) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[tid * 8 + 2 * e] = s[e]; red[tid * 8 + 2 * e + 1] = q[e]; }
    __syncthreads();
    // group g = channels [g cpg, (g + 1) cpg): cpg in {1, 2, 4, 8}
    const int groups = p.C / p.st.cpg;
    StatRows r{0, groups, 0.0f, 0.0f};
    if (tid < groups)
        for (int c = tid * p.st.cpg; c < (tid + 1) * p.st.cpg; ++c)
            for (int k = 0; k < nsub; ++k) { const float *rr = red + ((k * c4n + (c >> 2)) * 8 + 2 * (c & 3)); r.s += rr[0]; r.q += rr[1]; }
    stats_store(p.st, r, blockIdx.x, tid);
}

__global__ __launch_bounds__(256) void avgpool_kernel(const EltArgs p)
{
    elt_body(p, [&](int oy, int ox, int c4) {
        // torch's order: ((x00 + x01) + x10) + x11, then / 4; rows / columns beyond 2 H, 2 W (odd inputs) are dropped
        const float *r0 = p.a + ((size_t)(2 * oy) * p.Wb + 2 * ox) * p.C + 4 * c4;
        const float *r1 = r0 + (size_t)p.Wb * p.C;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(r0), b = *reinterpret_cast<const f32x4 *>(r0 + p.C);
        const f32x4 c = *reinterpret_cast<const f32x4 *>(r1), d = *reinterpret_cast<const f32x4 *>(r1 + p.C);
        return (((a + b) + c) + d) * 0.25f;
    });
}

// ---- up1 + interpolate(low3, scale_factor=2, mode='bicubic', align_corners=True) + statistics (HGFilters.py:113-116) ------------
// torch's upsample_bicubic2d: source coordinate = dst * (in - 1) / (out - 1), taps floor - 1 .. floor + 2 clamped to the image,
// cubic convolution coefficients with A = -0.75, rows first (along x), then along y.
__device__ __forceinline__ void cubic_coeffs(float t, float c[4])
{
    const float A = -0.75f;
    const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = x2 + 1.0f;
    c[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    c[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
    c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    c[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

__global__ __launch_bounds__(256) void upadd_kernel(const EltArgs p)
{
    const float sy = p.H > 1 ? (float)(p.Hb - 1) / (float)(p.H - 1) : 0.0f, sx = p.W > 1 ? (float)(p.Wb - 1) / (float)(p.W - 1) : 0.0f;
    elt_body(p, [&](int oy, int ox, int c4) {
        const float ry = sy * (float)oy, rx = sx * (float)ox;
        const int iy = min((int)floorf(ry), p.Hb - 1), ix = min((int)floorf(rx), p.Wb - 1);
        const float ty = fminf(fmaxf(ry - (float)iy, 0.0f), 1.0f), tx = fminf(fmaxf(rx - (float)ix, 0.0f), 1.0f);
        float cy[4], cx[4];
        cubic_coeffs(ty, cy);
        cubic_coeffs(tx, cx);
        int xs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xs[k] = min(max(ix - 1 + k, 0), p.Wb - 1);
        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int yy = min(max(iy - 1 + m, 0), p.Hb - 1);
            const float *row = p.b + (size_t)yy * p.Wb * p.C + 4 * c4;
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(row + (size_t)xs[0] * p.C), v1 = *reinterpret_cast<const f32x4 *>(row + (size_t)xs[1] * p.C);
            const f32x4 v2 = *reinterpret_cast<const f32x4 *>(row + (size_t)xs[2] * p.C), v3 = *reinterpret_cast<const f32x4 *>(row + (size_t)xs[3] * p.C);
            const f32x4 r = v0 * cx[0] + v1 * cx[1] + v2 * cx[2] + v3 * cx[3];
            acc = m == 0 ? r * cy[0] : acc + r * cy[m];
        }
        const f32x4 u = *reinterpret_cast<const f32x4 *>(p.a + ((size_t)oy * p.W + ox) * p.C + 4 * c4);
        return u + acc;
    });
}

// The same, tiled: an 8 x 16 tile of output pixels x 64 channels per workgroup.  The 8 x 12 source pixels the tile's taps can touch are staged
// once in LDS (clamped: the border replication is baked into the staged tile) and every output takes its 16 taps from there -- 16 LDS reads
// instead of 16 global loads per output float4; 24 KiB of LDS and one barrier, so six workgroups share a CU and hide each other's latencies
// (a two-pass separable form through a second LDS buffer was slower: three dependent phases per workgroup, two workgroups per CU).
// Same operation order per output as the direct kernel.
constexpr int UT_H = 8, UT_W = 16, UT_C = 64, UT_SR = 8, UT_SC = 12;
struct UpTiledArgs { EltArgs e; int tiles_x, tiles_y; };

__global__ __launch_bounds__(256) void upadd_tiled_kernel(const UpTiledArgs a)
{
    const EltArgs &p = a.e;
    __shared__ __attribute__((aligned(16))) float src[UT_SR * UT_SC * UT_C];     // 24 KiB
    const int tid = threadIdx.x;
    const int nt = a.tiles_x * a.tiles_y, tile = blockIdx.x % nt, chunk = blockIdx.x / nt;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = ty * UT_H, ox0 = tx * UT_W, c0 = chunk * UT_C;
    const float sy = p.H > 1 ? (float)(p.Hb - 1) / (float)(p.H - 1) : 0.0f, sx = p.W > 1 ? (float)(p.Wb - 1) / (float)(p.W - 1) : 0.0f;
    const int ylo = min((int)floorf(sy * (float)oy0), p.Hb - 1) - 1, xlo = min((int)floorf(sx * (float)ox0), p.Wb - 1) - 1;
    const int quad = tid & 15;
    // up1 of this thread's 8 outputs: in flight while the source tile is staged
    f32x4 up[UT_H * UT_W / 16];
#pragma unroll
    for (int i = 0; i < UT_H * UT_W / 16; ++i) {
        const int pp = (tid >> 4) + 16 * i, oy = oy0 + (pp >> 4), ox = ox0 + (pp & 15);
        up[i] = (oy < p.H && ox < p.W) ? *reinterpret_cast<const f32x4 *>(p.a + ((size_t)oy * p.W + ox) * p.C + c0 + quad * 4) : f32x4{0, 0, 0, 0};
    }
    for (int i = tid; i < UT_SR * UT_SC * (UT_C / 4); i += 256) {
        const int qd = i & 15, cell = i >> 4, r = cell / UT_SC, c = cell - r * UT_SC;
        const int yy = min(max(ylo + r, 0), p.Hb - 1), xx = min(max(xlo + c, 0), p.Wb - 1);
        *reinterpret_cast<f32x4 *>(src + cell * UT_C + qd * 4) = *reinterpret_cast<const f32x4 *>(p.b + ((size_t)yy * p.Wb + xx) * p.C + c0 + qd * 4);
    }
    __syncthreads();
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < UT_H * UT_W / 16; ++i) {
        const int pp = (tid >> 4) + 16 * i, y = pp >> 4, x = pp & 15;
        const int oy = oy0 + y, ox = ox0 + x;
        if (oy >= p.H || ox >= p.W) continue;
        const float ry = sy * (float)oy, rx = sx * (float)ox;
        const int iy = min((int)floorf(ry), p.Hb - 1), ix = min((int)floorf(rx), p.Wb - 1);
        float cy[4], cx[4];
        cubic_coeffs(fminf(fmaxf(ry - (float)iy, 0.0f), 1.0f), cy);
        cubic_coeffs(fminf(fmaxf(rx - (float)ix, 0.0f), 1.0f), cx);
        const int rb = min(max(iy - 1 - ylo, 0), UT_SR - 4), cb = min(max(ix - 1 - xlo, 0), UT_SC - 4);
        const float *base = src + (rb * UT_SC + cb) * UT_C + quad * 4;
        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float *row = base + m * UT_SC * UT_C;
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(row), v1 = *reinterpret_cast<const f32x4 *>(row + UT_C);
            const f32x4 v2 = *reinterpret_cast<const f32x4 *>(row + 2 * UT_C), v3 = *reinterpret_cast<const f32x4 *>(row + 3 * UT_C);
            const f32x4 r = v0 * cx[0] + v1 * cx[1] + v2 * cx[2] + v3 * cx[3];
            acc = m == 0 ? r * cy[0] : acc + r * cy[m];
        }
        f32x4 v = up[i] + acc;
        settle(v);
        *reinterpret_cast<f32x4 *>(p.out + ((size_t)oy * p.W + ox) * p.C + c0 + quad * 4) = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
    }
    if (!p.st.part) return;
    __syncthreads();
    float *red = src;                                              // [16 pixel lanes][64 channels](s, q)
#pragma unroll
    for (int e = 0; e < 4; ++e) *reinterpret_cast<f32x2 *>(red + (((tid >> 4) * UT_C) + quad * 4 + e) * 2) = f32x2{s[e], q[e]};
    __syncthreads();
    const int gpc = UT_C / p.st.cpg;                               // groups of this workgroup's 64 channels
    StatRows r{chunk * gpc, gpc, 0.0f, 0.0f};
    if (tid < gpc)
        for (int c = tid * p.st.cpg; c < (tid + 1) * p.st.cpg; ++c)
            for (int k = 0; k < 16; ++k) { const f32x2 v = *reinterpret_cast<const f32x2 *>(red + (k * UT_C + c) * 2); r.s += v[0]; r.q += v[1]; }
    stats_store(p.st, r, tile, tid);
}

// relu(GroupNorm(a)) materialised (HGFilters.py:178: the block that follows normalises THIS tensor again and needs its statistics)
__global__ __launch_bounds__(256) void normrelu_kernel(const EltArgs p)
{
    __shared__ f32x2 mr[32];
    fold_groups(p.in_part, p.C / p.in_cpg, p.in_nt, p.in_inv_n, p.in_eps, mr, [] {});
    const int c4 = threadIdx.x % (p.C >> 2);
    float a[4], b[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * c4 + e, g = c / p.in_cpg;
        a[e] = p.gamma[c] * mr[g][1];
        b[e] = p.beta[c] - mr[g][0] * a[e];
    }
    elt_body(p, [&](int oy, int ox, int c4_) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(p.a + ((size_t)oy * p.W + ox) * p.C + 4 * c4_);
        return f32x4{relu_bits(a[0] * v[0] + b[0]), relu_bits(a[1] * v[1] + b[1]), relu_bits(a[2] * v[2] + b[2]), relu_bits(a[3] * v[3] + b[3])};
    });
}

// (H, W, C) channel-last -> the reference's (C, H, W)
__global__ void hwc_to_nchw_kernel(const float *__restrict__ src, float *__restrict__ dst, int C, int HW)
{
    __shared__ float tile[64][65];
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int pp = p0 + r, c = c0 + tx;
        tile[r][tx] = (c < C && pp < HW) ? src[(size_t)pp * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, pp = p0 + tx;
        if (c < C && pp < HW) dst[(size_t)c * HW + pp] = tile[tx][r];
    }
}


// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) of relu(x) (network/unets.py:42-45, the ReLU of :52): out[2 i] = 0.25 x[i - 1] + 0.75 x[i],
// out[2 i + 1] = 0.75 x[i] + 0.25 x[i + 1], indices clamped -- torch's source index (dst + 0.5) / 2 - 0.5 floored at 0.  x (H, W, C) -> out (2 H, 2 W, C).
struct Up2Args { const float *x; float *out; int H, W, C; };
__global__ __launch_bounds__(256) void up2_kernel(const Up2Args p)
{
    const int c4n = p.C >> 2;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)4 * p.H * p.W * c4n;
    if (i >= total) return;
    const int c4 = (int)(i % c4n);
    const size_t pix = i / c4n;
    const int ox = (int)(pix % (2 * p.W)), oy = (int)(pix / (2 * p.W));
    // torch: real = max((dst + 0.5) * 0.5 - 0.5, 0); i0 = floor(real); lambda1 = real - i0; i1 = min(i0 + 1, size - 1)
    const float ry = fmaxf((oy + 0.5f) * 0.5f - 0.5f, 0.0f), rx = fmaxf((ox + 0.5f) * 0.5f - 0.5f, 0.0f);
    const int y0 = (int)ry, x0 = (int)rx, y1 = min(y0 + 1, p.H - 1), x1 = min(x0 + 1, p.W - 1);
    const float ly = ry - (float)y0, lx = rx - (float)x0;
    auto at = [&](int y, int x) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(p.x + ((size_t)y * p.W + x) * p.C + 4 * c4);
        return f32x4{relu_bits(v[0]), relu_bits(v[1]), relu_bits(v[2]), relu_bits(v[3])};
    };
    const f32x4 v00 = at(y0, x0), v01 = at(y0, x1), v10 = at(y1, x0), v11 = at(y1, x1);
    const float w00 = (1.0f - ly) * (1.0f - lx), w01 = (1.0f - ly) * lx, w10 = ly * (1.0f - lx), w11 = ly * lx;
    f32x4 vo = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
    settle(vo);
    *reinterpret_cast<f32x4 *>(p.out + pix * p.C + 4 * c4) = vo;
}

// =====================================================================================================================
// host side: weight packing, the launch plan of one input size, the hipGraph
// =====================================================================================================================
struct DevConv {            // a packed convolution
    char *wstream = nullptr; unsigned wbytes = 0;
    float *bias = nullptr;  // device (cout) or null
    float wscale_inv = 1.0f;     // 2^-sw
    int cout = 0, cin = 0, taps = 0;
    unsigned off[3] = {0, 0, 0}; // byte offset of the stream packed for CT = 1, 2, 4
    int ct_mask = 0;             // which of them were packed
};
struct DevNorm { float *gamma = nullptr, *beta = nullptr; int C = 0, groups = 32; float eps = 1e-5f; };
struct DevBlock { DevConv conv[3], ds; bool has_ds = false; DevNorm bn[4]; int cin = 0, cout = 0; };

struct Tensor { float *data = nullptr; int H = 0, W = 0, C = 0;
                f32x2 *part = nullptr; int nt = 0; };       // GroupNorm partials [32 groups][nt tiles] its producers leave, its consumers fold

enum LaunchKind { L_S2D, L_CONV, L_POOL, L_UPADD, L_UPADD_TILED, L_NORMRELU, L_UP2, L_FORK, L_JOIN };
struct Launch {
    LaunchKind kind;
    ConvArgs conv; int CT = 0, PT = 0, TAPS = 0, TWC = 0; bool norm = false;
    bool occ2 = false;               // the two-workgroups-per-CU flavour of the kernel (ConvGeo: OCC2)
    S2dArgs s2d;
    Up2Args up2;
    EltArgs elt; int ut_x = 0, ut_y = 0;     // (L_UPADD_TILED: the tile grid)
    unsigned grid = 0;
    int side = 0;                    // 1: the launch goes to the side stream (the hourglass' upper branches run beside the lower ones)
    int event = -1;                  // L_FORK: the side stream waits for the main stream here; L_JOIN: the main stream waits for the side stream
};

struct Encoder {
    bool packed = false;
    DevConv conv1, conv_last, l;
    DevNorm bn1, bn_end;
    DevBlock conv2, conv3, conv4, top_m;
    std::vector<DevBlock> hg;        // b1_d, b2_d, ..., b1_1, b2_1, b2_plus_1, b3_1 .. b3_d
    int depth = 0;
    DevConv u_down[7], u_up[3], u_upc[3];       // the U-Net's convolutions (a second Encoder object holds them: avc_ctx::unet)
    std::vector<void *> weight_allocs;
    // the plan of one input size
    int Hin = 0, Win = 0;
    std::vector<Launch> plan;
    std::vector<void *> plan_allocs;
    float *in_buf = nullptr;
    unsigned *range_flag = nullptr;
    Tensor out, normx;
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    hipStream_t cap_stream = nullptr, side_stream = nullptr;
    std::vector<hipEvent_t> events;
    int fork = -1, ksplit = -1, occ2 = -1;
};

static void free_plan(Encoder *e)
{
    if (e->exec) { hipGraphExecDestroy(e->exec); e->exec = nullptr; }
    if (e->graph) { hipGraphDestroy(e->graph); e->graph = nullptr; }
    for (void *p : e->plan_allocs) hipFree(p);
    for (hipEvent_t ev : e->events) hipEventDestroy(ev);
    e->events.clear();
    e->plan_allocs.clear(); e->plan.clear(); e->Hin = e->Win = 0; e->in_buf = nullptr; e->range_flag = nullptr;
}
static void free_weights(Encoder *e)
{
    for (void *p : e->weight_allocs) hipFree(p);
    e->weight_allocs.clear();
    e->hg.clear();
    e->packed = false;
}
void release_encoder(avc_ctx *ctx)
{
    Encoder *e = static_cast<Encoder *>(ctx->encoder);
    if (!e) return;
    free_plan(e);
    free_weights(e);
    if (e->cap_stream) hipStreamDestroy(e->cap_stream);
    if (e->side_stream) hipStreamDestroy(e->side_stream);
    delete e;
    ctx->encoder = nullptr;
}

template <class T>
static int upload_vec(Encoder *e, const std::vector<T> &v, T **dst)
{
    void *d = nullptr;
    AVC_HIP(hipMalloc(&d, v.size() * sizeof(T)));
    e->weight_allocs.push_back(d);
    AVC_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<T *>(d);
    return AVC_OK;
}

constexpr int conv_ct(int cout) { return cout >= 128 ? 4 : (cout >= 64 ? 2 : 1); }

// [slice][chunk][tap][k-step][tile][hi 1 KiB | lo 1 KiB]; element (lane, e) of a fragment: co = 32 (slice CT + tile) + (lane & 31),
// ci = 32 chunk + 16 kstep + 8 (lane >> 5) + e.  Any CT that divides cout / 32 reads the same stream (a slice of CT tiles is CT
// consecutive slices of one tile only when the tile index is the outermost axis -- it is not; so the stream is packed per CT).
static int pack_conv(Encoder *e, const avc_conv2d &c, int taps_expected, DevConv &d, const char *name, int ct_mask = 7)
{
    AVC_REQUIRE(c.w, AVC_ERR_ARG, "avc_hgfilter_pack: %s: NULL weight", name);
    AVC_REQUIRE(c.kh * c.kw == taps_expected && c.kh == c.kw, AVC_ERR_ARG, "avc_hgfilter_pack: %s: kernel %dx%d, expected %d taps", name, c.kh, c.kw, taps_expected);
    AVC_REQUIRE(c.cin % 32 == 0 && c.cin <= MAX_CIN && c.cout % 32 == 0 && c.cout <= 1024, AVC_ERR_ARG,
                "pack: %s: (%d <- %d) channels; multiples of 32 up to 1024 are supported", name, c.cout, c.cin);
    d.cout = c.cout; d.cin = c.cin; d.taps = taps_expected; d.ct_mask = 0;
    const int taps = taps_expected, nchunk = c.cin / 32;
    double m = 0.0;
    const size_t nw = (size_t)c.cout * c.cin * taps;
    for (size_t i = 0; i < nw; ++i) {
        AVC_REQUIRE(std::isfinite(c.w[i]), AVC_ERR_ARG, "avc_hgfilter_pack: %s: non-finite weight", name);
        m = std::fmax(m, std::fabs((double)c.w[i]));
    }
    int sw = 0;
    if (m > 0.0) { int ex; std::frexp(m, &ex); sw = 10 - ex; }      // largest |w| 2^sw in [2^9, 2^10): the lo halves stay normal fp16 numbers
    sw = std::max(-40, std::min(40, sw));
    d.wscale_inv = (float)std::ldexp(1.0, -sw);
    // one stream per CT in {1, 2, 4} that divides the tile count: [CT == 1 | CT == 2 | CT == 4] back to back.  Every 2 KiB unit [hi | lo] is a function of its
    // index alone: the units are filled by a few host threads (the U-Net's 10 M weights in three slice widths took 0.57 s of the first frame on one).
    const int ntile = c.cout / 32;
    const size_t units_per_ct = (size_t)ntile * nchunk * taps * 2;       // the same for every CT that divides ntile
    unsigned off[3] = {0, 0, 0};
    size_t total = 0;
    for (int v = 0; v < 3; ++v) {
        const int CT = 1 << v;
        off[v] = (unsigned)total;
        if (ntile % CT || !(ct_mask & CT)) continue;
        d.ct_mask |= CT;
        total += units_per_ct * 2048;
    }
    AVC_REQUIRE(total < (1ull << 32), AVC_ERR_ARG, "avc_hgfilter_pack: %s: a weight stream of %zu bytes", name, total);
    std::vector<uint8_t> stream(total);
    const float scale = std::ldexp(1.0f, sw);                            // a power of two: w * scale is exact (|w| 2^sw < 2^10)
    auto fill = [&](int v, size_t u0, size_t u1) {
        const int CT = 1 << v;
        for (size_t u = u0; u < u1; ++u) {
            // u = ((((slice * nchunk + ch) * taps + t) * 2 + kk) * CT + tl
            size_t r = u;
            const int tl = (int)(r % CT); r /= CT;
            const int kk = (int)(r % 2); r /= 2;
            const int t = (int)(r % taps); r /= taps;
            const int ch = (int)(r % nchunk); r /= nchunk;
            const int slice = (int)r;
            _Float16 *hi = reinterpret_cast<_Float16 *>(stream.data() + off[v] + u * 2048), *lo = hi + 512;
            for (int lane = 0; lane < 64; ++lane)
                for (int el = 0; el < 8; ++el) {
                    const int co = 32 * (slice * CT + tl) + (lane & 31), ci = 32 * ch + 16 * kk + 8 * (lane >> 5) + el;
                    const float w = c.w[((size_t)co * c.cin + ci) * taps + t] * scale;
                    const _Float16 hh = (_Float16)w;
                    hi[lane * 8 + el] = hh;
                    lo[lane * 8 + el] = (_Float16)(w - (float)hh);
                }
        }
    };
    {
        const unsigned hw = std::thread::hardware_concurrency();
        const int nthr = (int)std::max<size_t>(1, std::min<size_t>({(size_t)(hw ? hw : 4), (size_t)16, units_per_ct / 64 + 1}));
        for (int v = 0; v < 3; ++v) {
            if (!(d.ct_mask & (1 << v))) continue;
            if (nthr == 1) { fill(v, 0, units_per_ct); continue; }
            std::vector<std::thread> pool;
            for (int k = 0; k < nthr; ++k)
                pool.emplace_back(fill, v, units_per_ct * k / nthr, units_per_ct * (k + 1) / nthr);
            for (auto &th : pool) th.join();
        }
    }
    d.wbytes = (unsigned)stream.size();
    uint8_t *dev = nullptr;
    if (int rc = upload_vec(e, stream, &dev)) return rc;
    d.wstream = reinterpret_cast<char *>(dev);
    d.off[0] = off[0]; d.off[1] = off[1]; d.off[2] = off[2];
    if (c.b) { std::vector<float> b(c.b, c.b + c.cout); if (int rc = upload_vec(e, b, &d.bias)) return rc; }
    return AVC_OK;
}


static int pack_norm(Encoder *e, const avc_groupnorm &g, int C, DevNorm &d, const char *name)
{
    AVC_REQUIRE(g.gamma && g.beta, AVC_ERR_ARG, "avc_hgfilter_pack: %s: NULL gamma / beta", name);
    AVC_REQUIRE(g.channels == C && g.groups == 32 && C % 32 == 0 && C <= MAX_NORM_CIN, AVC_ERR_ARG,
                "avc_hgfilter_pack: %s: GroupNorm(%d, %d), expected GroupNorm(32, %d)", name, g.groups, g.channels, C);
    d.C = C; d.groups = g.groups; d.eps = g.eps;
    std::vector<float> ga(g.gamma, g.gamma + C), be(g.beta, g.beta + C);
    if (int rc = upload_vec(e, ga, &d.gamma)) return rc;
    return upload_vec(e, be, &d.beta);
}

static int pack_block(Encoder *e, const avc_convblock &b, int cin, int cout, DevBlock &d, const char *name)
{
    char nm[96];
    const int co[3] = {cout / 2, cout / 4, cout / 4}, ci[3] = {cin, cout / 2, cout / 4};
    d.cin = cin; d.cout = cout;
    for (int i = 0; i < 3; ++i) {
        snprintf(nm, sizeof nm, "%s.conv%d", name, i + 1);
        AVC_REQUIRE(b.conv[i].cout == co[i] && b.conv[i].cin == ci[i], AVC_ERR_ARG, "avc_hgfilter_pack: %s: weight (%d,%d,..), expected (%d,%d,3,3)", nm,
                    b.conv[i].cout, b.conv[i].cin, co[i], ci[i]);
        AVC_REQUIRE(!b.conv[i].b, AVC_ERR_ARG, "avc_hgfilter_pack: %s has no bias in the reference (HGFilters.py:29-31)", nm);
        if (int rc = pack_conv(e, b.conv[i], 9, d.conv[i], nm)) return rc;
        snprintf(nm, sizeof nm, "%s.bn%d", name, i + 1);
        if (int rc = pack_norm(e, b.bn[i], ci[i], d.bn[i], nm)) return rc;
    }
    d.has_ds = cin != cout;
    AVC_REQUIRE(d.has_ds == (b.downsample.w != nullptr), AVC_ERR_ARG, "avc_hgfilter_pack: %s: a 1x1 projection exists exactly when in_planes != out_planes (HGFilters.py:51-58)", name);
    if (d.has_ds) {
        snprintf(nm, sizeof nm, "%s.downsample.2", name);
        AVC_REQUIRE(b.downsample.cout == cout && b.downsample.cin == cin && !b.downsample.b, AVC_ERR_ARG, "avc_hgfilter_pack: %s: bad shape or a bias", nm);
        if (int rc = pack_conv(e, b.downsample, 1, d.ds, nm)) return rc;
        snprintf(nm, sizeof nm, "%s.bn4", name);
        if (int rc = pack_norm(e, b.bn[3], cin, d.bn[3], nm)) return rc;
    }
    return AVC_OK;
}

int pack_encoder(avc_ctx *ctx, const avc_hgfilter *net)
{
    if (!ctx->encoder) ctx->encoder = new Encoder();
    Encoder *e = static_cast<Encoder *>(ctx->encoder);
    free_plan(e);
    free_weights(e);
    AVC_REQUIRE(net->depth >= 1 && net->depth <= 6 && net->hourglass, AVC_ERR_ARG, "avc_hgfilter_pack: depth %d", net->depth);
    // conv1: 7x7 stride 2 pad 3, 6 -> 64, bias -> 4x4 taps over the 24 (padded to 32) space-to-depth channels
    const avc_conv2d &c1 = net->conv1;
    AVC_REQUIRE(c1.w && c1.b && c1.cout == 64 && c1.cin == 6 && c1.kh == 7 && c1.kw == 7, AVC_ERR_ARG,
                "avc_hgfilter_pack: conv1 must be Conv2d(6, 64, 7, stride 2, padding 3) with a bias (HGFilters.py:134)");
    {
        std::vector<float> w((size_t)64 * 32 * 16, 0.0f);
        for (int co = 0; co < 64; ++co)
            for (int par = 0; par < 4; ++par)
                for (int c = 0; c < 6; ++c)
                    for (int ty = 0; ty < 4; ++ty)
                        for (int tx = 0; tx < 4; ++tx) {
                            const int ky = 2 * ty + (par >> 1) - 1, kx = 2 * tx + (par & 1) - 1;
                            if (ky < 0 || ky > 6 || kx < 0 || kx > 6) continue;
                            w[((size_t)co * 32 + par * 6 + c) * 16 + ty * 4 + tx] = c1.w[((size_t)co * 6 + c) * 49 + ky * 7 + kx];
                        }
        const avc_conv2d s2d{w.data(), c1.b, 64, 32, 4, 4};
        if (int rc = pack_conv(e, s2d, 16, e->conv1, "conv1")) return rc;
    }
    if (int rc = pack_norm(e, net->bn1, 64, e->bn1, "bn1")) return rc;
    if (int rc = pack_block(e, net->conv2, 64, 128, e->conv2, "conv2")) return rc;
    if (int rc = pack_block(e, net->conv3, 128, 128, e->conv3, "conv3")) return rc;
    if (int rc = pack_block(e, net->conv4, 128, 256, e->conv4, "conv4")) return rc;
    e->depth = net->depth;
    e->hg.resize(3 * net->depth + 1);
    for (int i = 0; i < 3 * net->depth + 1; ++i) {
        char nm[32]; snprintf(nm, sizeof nm, "m0[%d]", i);
        if (int rc = pack_block(e, net->hourglass[i], 256, 256, e->hg[i], nm)) return rc;
    }
    if (int rc = pack_block(e, net->top_m, 256, 256, e->top_m, "top_m_0")) return rc;
    AVC_REQUIRE(net->conv_last.cout == 256 && net->conv_last.cin == 256 && net->conv_last.b, AVC_ERR_ARG, "avc_hgfilter_pack: conv_last0 must be Conv2d(256, 256, 1) with a bias");
    if (int rc = pack_conv(e, net->conv_last, 1, e->conv_last, "conv_last0")) return rc;
    if (int rc = pack_norm(e, net->bn_end, 256, e->bn_end, "bn_end0")) return rc;
    AVC_REQUIRE(net->l.cin == 256 && net->l.cout == 32 && net->l.b, AVC_ERR_ARG, "avc_hgfilter_pack: l0 must be Conv2d(256, 32, 1) with a bias (the decoder samples 32 channels)");
    if (int rc = pack_conv(e, net->l, 1, e->l, "l0")) return rc;
    e->packed = true;
    return AVC_OK;
}

// ---- launch plan -----------------------------------------------------------------------------------------------------
struct Planner {
    avc_ctx *ctx; Encoder *e; std::vector<void *> allocs; int rc = AVC_OK;
    float gn_eps = 1e-5f; int gn_groups = 32;
    int side = 0; bool fork = false;

    void push(Launch L) { L.side = side; e->plan.push_back(L); }
    void sync(LaunchKind kind)
    {
        if (!fork || rc) return;
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { set_error("avc_hgfilter_forward: hipEventCreate failed"); rc = AVC_ERR_HIP; return; }
        e->events.push_back(ev);
        Launch L{}; L.kind = kind; L.event = (int)e->events.size() - 1;
        e->plan.push_back(L);
    }

    void *alloc(size_t bytes, bool zero = false)
    {
        void *d = nullptr;
        if (rc) return nullptr;
        if (hipMalloc(&d, std::max<size_t>(bytes, 16)) != hipSuccess) { set_error("avc_hgfilter_forward: out of device memory"); rc = AVC_ERR_HIP; return nullptr; }
        allocs.push_back(d);
        if (zero) hipMemset(d, 0, std::max<size_t>(bytes, 16));
        return d;
    }
    Tensor tensor(int H, int W, int C)
    {
        Tensor t; t.H = H; t.W = W; t.C = C;
        t.data = static_cast<float *>(alloc(sizeof(float) * (size_t)H * W * C));
        return t;
    }
    // the statistics a launch of `ntiles` tiles leaves for the channels from first_channel on of t: its rows of the tensor's [group][pitch] table.  The
    // producers of one tensor fill different rows and need not cut the image into the same tiles: the pitch is the largest tile count any launch shape
    // has on a tensor of this size, the table is zeroed once when the plan is built, a producer of fewer tiles leaves the rest of its rows zero, and the
    // consumers fold whole rows.
    static int stat_pitch(int H, int W)
    {
        const int twc = W >= 32 ? 32 : 16, rows = 4 * (32 / twc), npix = H * W;
        const int conv_tiles = ((H + rows - 1) / rows) * ((W + twc - 1) / twc);                                      // a convolution's smallest tile (PT = 1)
        const int ppw = npix <= 1024 ? std::max(16, (npix + 31) / 32) : std::max(16, (npix + 511) / 512);           // elementwise()
        const int elt_tiles = std::max((npix + ppw - 1) / ppw, ((W + UT_W - 1) / UT_W) * ((H + UT_H - 1) / UT_H));
        return (std::max(conv_tiles, elt_tiles) + 7) & ~7;
    }
    StatOut stat(Tensor &t, int first_channel, int ntiles)
    {
        StatOut s{};
        s.cpg = t.C / gn_groups;
        if (!t.part) { t.nt = stat_pitch(t.H, t.W); t.part = static_cast<f32x2 *>(alloc(sizeof(f32x2) * (size_t)gn_groups * t.nt, true)); }
        if (ntiles > t.nt && !rc) { set_error("avc_hgfilter_forward: internal: a launch of %d tiles on a statistics table of pitch %d", ntiles, t.nt); rc = AVC_ERR_STATE; }
        s.ntiles = t.nt;
        s.part = t.part + (size_t)(first_channel / s.cpg) * t.nt;
        return s;
    }

    // conv: x (through gn + ReLU when gn != null) -> raw (with statistics when raw_stats) and / or y[:, ycoff ...] = conv + res
    // (force_pt: the tile height the block chose for its three convolutions; a two-per-CU launch takes its own)
    void conv(const DevConv &w, const Tensor &x, const DevNorm *gn, Tensor *raw, bool raw_stats, Tensor *y, const Tensor *res, int ycoff, float raw_in_scale = 1.0f,
              int force_pt = 0)
    {
        if (rc) return;
        Launch L{}; L.kind = L_CONV;
        L.TAPS = w.taps; L.norm = gn != nullptr;
        L.TWC = x.W >= 32 ? 32 : 16;
        L.CT = conv_ct(w.cout); L.PT = (L.TWC == 32 && w.taps != 16) ? 2 : 1;
        auto wgs = [&](int CT, int PT) { const int rows = 4 * PT * (32 / L.TWC); return ((x.H + rows - 1) / rows) * ((x.W + L.TWC - 1) / L.TWC) * (w.cout / (32 * CT)); };
        if (force_pt) L.PT = force_pt;
        // two half-height workgroups per CU (ConvGeo: OCC2) where the launch then has two to four of them per CU: 3x3 on 32-wide tiles, channel slices
        // of 64 or 32.  Measured on the encoder's shapes (profiles/r06_enc_phases.md): 44.8 -> 41.8 us (256^2 128 -> 64), 29.9 -> 27.6 (64 -> 64), 20.0 -> 18.9
        // (64 -> 32), 48.5 -> 43.6 (128^2 256 -> 128 as 32-channel slices); NOT where it takes narrower slices AND more rounds (256^2 256 -> 128: 133 -> 147 us as
        // 1024 workgroups of 64 channels) nor for the 1x1 convolutions (59.7 -> 77.7 us: they are output-bound, and half tiles double the weight stream).
        if (ctx->opt.enc_occ2 && L.TWC == 32 && w.taps == 9)
            for (int ct = std::min(L.CT, 2); ct >= 1 && !L.occ2; ct /= 2)
                if (wgs(ct, 1) >= 2 * ctx->num_cus && wgs(ct, 1) < 4 * ctx->num_cus) { L.occ2 = true; L.PT = 1; L.CT = ct; }
        while (!L.occ2 && wgs(L.CT, L.PT) < ctx->num_cus) {
            if (L.PT == 2 && !force_pt) L.PT = 1;
            else if (L.CT > 1 && w.taps != 16) L.CT /= 2;
            else break;
        }
        const int rows = 4 * L.PT * (32 / L.TWC);
        ConvArgs &a = L.conv;
        a.x = x.data; a.H = x.H; a.W = x.W; a.Cin = x.C;
        a.in_part = gn ? x.part : nullptr; a.in_nt = x.nt; a.in_eps = gn_eps;
        a.in_inv_n = gn ? 1.0f / ((float)(x.C / gn->groups) * (float)x.H * (float)x.W) : 0.0f;
        if (gn && !x.part && !rc) { set_error("avc_hgfilter_forward: internal: a normalised input without statistics"); rc = AVC_ERR_STATE; }
        a.gamma = gn ? gn->gamma : nullptr; a.beta = gn ? gn->beta : nullptr; a.in_cpg = gn ? x.C / gn->groups : 1;
        a.in_scale = gn ? 16.0f : raw_in_scale;
        a.in_slope = 1.0f;
        const int v = L.CT == 4 ? 2 : (L.CT == 2 ? 1 : 0);
        a.slice_bytes = (unsigned)(x.C / 32) * w.taps * 2 * L.CT * 2048;
        a.wstream = w.wstream + w.off[v]; a.wbytes = a.slice_bytes * (w.cout / (32 * L.CT));
        a.bias = w.bias; a.out_scale = w.wscale_inv / a.in_scale; a.Cout = w.cout;
        a.raw = raw ? raw->data : nullptr;
        a.y = y ? y->data : nullptr; a.res = res ? res->data : nullptr; a.yC = y ? y->C : 0; a.ycoff = ycoff;
        a.tiles_x = (x.W + L.TWC - 1) / L.TWC; a.tiles_y = (x.H + rows - 1) / rows;
        const int ntiles = a.tiles_x * a.tiles_y;
        const int slices = w.cout / (32 * L.CT);
        if (raw && raw_stats) a.st_raw = stat(*raw, 0, ntiles);
        if (y) a.st_y = stat(*y, ycoff, ntiles);
        // few workgroups, each streaming its whole K serially, are bound by the latency of their weight stream: split K over more of them
        const int wg = ntiles * (w.cout / (32 * L.CT)), nchunk = x.C / 32;
        a.range_flag = e->range_flag;
        a.ksplit = 1;
        if (ctx->opt.enc_ksplit)
            while (nchunk % (2 * a.ksplit) == 0 && a.ksplit < 8 && 2 * wg * a.ksplit <= ctx->num_cus) a.ksplit *= 2;       // slices of whole chunks
        if (a.ksplit > 1) {
            a.kpart = static_cast<float *>(alloc(sizeof(float) * (size_t)wg * a.ksplit * 256 * L.PT * L.CT * 16));
            a.kcounter = static_cast<unsigned *>(alloc(sizeof(unsigned) * wg, true));
        }
        L.grid = (unsigned)(wg * a.ksplit);
        push(L);
    }

    Tensor block(const DevBlock &b, const Tensor &x)
    {
        Tensor y = tensor(x.H, x.W, b.cout), o1 = tensor(x.H, x.W, b.cout / 2), o2 = tensor(x.H, x.W, b.cout / 4), r;
        const Tensor *res = &x;
        if (b.has_ds) { r = tensor(x.H, x.W, b.cout); conv(b.ds, x, &b.bn[3], &r, false, nullptr, nullptr, 0); res = &r; }
        // the tile height of the block: what its narrowest convolution (cout / 4 channels) would pick on its own
        const int twc = x.W >= 32 ? 32 : 16;
        int pt = twc == 32 ? 2 : 1;
        if (pt == 2 && ((x.H + 7) / 8) * ((x.W + 31) / 32) * std::max(1, b.cout / 4 / 32) < ctx->num_cus) pt = 1;
        conv(b.conv[0], x, &b.bn[0], &o1, true, &y, res, 0, 1.0f, pt);
        conv(b.conv[1], o1, &b.bn[1], &o2, true, &y, res, b.cout / 2, 1.0f, pt);
        conv(b.conv[2], o2, &b.bn[2], nullptr, false, &y, res, b.cout / 2 + b.cout / 4, 1.0f, pt);
        return y;
    }

    Tensor elementwise(LaunchKind kind, const Tensor &a, const Tensor *b, int H, int W, const DevNorm *gn = nullptr)
    {
        Tensor out = tensor(H, W, a.C);
        if (rc) return out;
        Launch L{}; L.kind = kind;
        EltArgs &g = L.elt;
        g.a = a.data; g.b = b ? b->data : nullptr; g.out = out.data; g.H = H; g.W = W; g.C = a.C;
        g.Hb = b ? b->H : a.H; g.Wb = b ? b->W : a.W;
        if (gn) {
            g.in_part = a.part; g.in_nt = a.nt; g.gamma = gn->gamma; g.beta = gn->beta; g.in_cpg = a.C / gn->groups;
            g.in_inv_n = 1.0f / ((float)g.in_cpg * (float)a.H * (float)a.W); g.in_eps = gn_eps;
        }
        const int npix = H * W;
        if (kind == L_UPADD && a.C % UT_C == 0 && H >= 2 * UT_H && W >= UT_W) {       // the tiled form (every level of the 512^2 path but the lowest)
            L.kind = L_UPADD_TILED;
            L.ut_x = (W + UT_W - 1) / UT_W; L.ut_y = (H + UT_H - 1) / UT_H;
            g.ntiles = L.ut_x * L.ut_y;
            g.ppw = UT_H * UT_W;
            g.st = stat(out, 0, g.ntiles);
            L.grid = (unsigned)(g.ntiles * (a.C / UT_C));
        } else {
            g.ppw = npix <= 1024 ? std::max(16, (npix + 31) / 32) : std::max(16, (npix + 511) / 512);      // small tensors: <= 32 tiles, no tickets
            g.ntiles = (npix + g.ppw - 1) / g.ppw;
            g.st = stat(out, 0, g.ntiles);
            L.grid = (unsigned)g.ntiles;
        }
        push(L);
        return out;
    }

    Tensor level(int lvl, const Tensor &x)
    {
        const int d = e->depth;
        // the upper branch (b1) needs only x: it runs on the side stream beside the whole lower branch (pool, b2, the inner levels, b3), whose
        // small launches leave most of the chip idle (HGFilters.py:98-118: up1 and low1..low3 meet at up1 + up2)
        // (the pool goes first: a small launch that shares the chip with a one-workgroup-per-CU convolution waits for that launch's end)
        Tensor low = elementwise(L_POOL, x, nullptr, x.H / 2, x.W / 2);
        sync(L_FORK);
        side = fork ? 1 : 0;
        Tensor up1 = block(e->hg[2 * (d - lvl)], x);
        side = 0;
        low = block(e->hg[2 * (d - lvl) + 1], low);
        low = lvl > 1 ? level(lvl - 1, low) : block(e->hg[2 * d], low);
        low = block(e->hg[2 * d + lvl], low);
        sync(L_JOIN);
        return elementwise(L_UPADD, up1, &low, x.H, x.W);
    }
};

template <int CT, int PT, int TAPS, int TWC, bool NORM, bool OCC2 = false>
static int launch_conv_t(const ConvArgs &a, unsigned grid, hipStream_t s)
{
    static bool attr[64] = {};                               // per device: the attribute belongs to the function on the current device
    int dev = 0;
    AVC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr[dev]) {
        AVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_mfma_kernel<CT, PT, TAPS, TWC, NORM, OCC2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    ConvGeo<PT, TAPS, TWC, OCC2>::L_TOTAL));
        if (dev >= 0 && dev < 64) attr[dev] = true;
    }
    constexpr int lds = ConvGeo<PT, TAPS, TWC, OCC2>::L_TOTAL;
    static_assert(!OCC2 || 2 * lds <= 163840, "two workgroups of this variant do not fit a CU's LDS");
    hipLaunchKernelGGL((conv_mfma_kernel<CT, PT, TAPS, TWC, NORM, OCC2>), dim3(grid), dim3(256), lds, s, a);
    return AVC_OK;
}

static int launch_conv(const Launch &L, hipStream_t s)
{
#define AVC_ENC_OCC2(CT_, TAPS_, NORM_) \
    if (L.occ2 && L.CT == CT_ && L.PT == 1 && L.TAPS == TAPS_ && L.TWC == 32 && L.norm == NORM_) return launch_conv_t<CT_, 1, TAPS_, 32, NORM_, true>(L.conv, L.grid, s);
    AVC_ENC_OCC2(1, 9, true) AVC_ENC_OCC2(2, 9, true) AVC_ENC_OCC2(1, 1, true) AVC_ENC_OCC2(2, 1, true) AVC_ENC_OCC2(1, 1, false) AVC_ENC_OCC2(2, 1, false)
#undef AVC_ENC_OCC2
    if (L.occ2) { set_error("avc_hgfilter_forward: no two-per-CU kernel for CT %d PT %d taps %d tile width %d norm %d", L.CT, L.PT, L.TAPS, L.TWC, (int)L.norm); return AVC_ERR_STATE; }
#define AVC_ENC_CASE(CT_, PT_, TAPS_, TWC_, NORM_) \
    if (L.CT == CT_ && L.PT == PT_ && L.TAPS == TAPS_ && L.TWC == TWC_ && L.norm == NORM_) return launch_conv_t<CT_, PT_, TAPS_, TWC_, NORM_>(L.conv, L.grid, s);
#define AVC_ENC_CT(PT_, TAPS_, TWC_, NORM_) AVC_ENC_CASE(1, PT_, TAPS_, TWC_, NORM_) AVC_ENC_CASE(2, PT_, TAPS_, TWC_, NORM_) AVC_ENC_CASE(4, PT_, TAPS_, TWC_, NORM_)
#define AVC_ENC_GEO(TAPS_, NORM_) AVC_ENC_CT(2, TAPS_, 32, NORM_) AVC_ENC_CT(1, TAPS_, 32, NORM_) AVC_ENC_CT(1, TAPS_, 16, NORM_)
    AVC_ENC_GEO(9, true)
    AVC_ENC_GEO(9, false)
    AVC_ENC_GEO(4, false)
    AVC_ENC_GEO(1, true)
    AVC_ENC_GEO(1, false)
    AVC_ENC_CASE(2, 1, 16, 32, false)
    AVC_ENC_CASE(2, 1, 16, 16, false)
#undef AVC_ENC_GEO
#undef AVC_ENC_CT
#undef AVC_ENC_CASE
    set_error("avc_hgfilter_forward: no kernel for CT %d PT %d taps %d tile width %d norm %d", L.CT, L.PT, L.TAPS, L.TWC, (int)L.norm);
    return AVC_ERR_STATE;
}

static int run_plan(Encoder *e, hipStream_t main_stream)
{
    for (const Launch &L : e->plan) {
        hipStream_t s = L.side ? e->side_stream : main_stream;
        switch (L.kind) {
        case L_FORK: AVC_HIP(hipEventRecord(e->events[L.event], main_stream)); AVC_HIP(hipStreamWaitEvent(e->side_stream, e->events[L.event], 0)); break;
        case L_JOIN: AVC_HIP(hipEventRecord(e->events[L.event], e->side_stream)); AVC_HIP(hipStreamWaitEvent(main_stream, e->events[L.event], 0)); break;
        case L_S2D: hipLaunchKernelGGL(s2d_kernel, dim3(L.grid), dim3(256), 0, s, L.s2d); break;
        case L_CONV: if (int rc = launch_conv(L, s)) return rc; break;
        case L_POOL: hipLaunchKernelGGL(avgpool_kernel, dim3(L.grid), dim3(256), 0, s, L.elt); break;
        case L_UPADD: hipLaunchKernelGGL(upadd_kernel, dim3(L.grid), dim3(256), 0, s, L.elt); break;
        case L_UPADD_TILED: { UpTiledArgs u{L.elt, L.ut_x, L.ut_y}; hipLaunchKernelGGL(upadd_tiled_kernel, dim3(L.grid), dim3(256), 0, s, u); break; }
        case L_NORMRELU: hipLaunchKernelGGL(normrelu_kernel, dim3(L.grid), dim3(256), 0, s, L.elt); break;
        case L_UP2: hipLaunchKernelGGL(up2_kernel, dim3(L.grid), dim3(256), 0, s, L.up2); break;
        }
    }
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

static int build_plan(avc_ctx *ctx, Encoder *e, int Hin, int Win)
{
    free_plan(e);
    const int H1 = (Hin - 1) / 2 + 1, W1 = (Win - 1) / 2 + 1;
    AVC_REQUIRE(H1 % (1 << e->depth) == 0 && W1 % (1 << e->depth) == 0, AVC_ERR_ARG,
                "avc_hgfilter_forward: a %d x %d image gives %d x %d features, which the depth-%d hourglass cannot halve %d times and add back "
                "(up1 + up2, HGFilters.py:118: the reference raises a size mismatch)", Hin, Win, H1, W1, e->depth, e->depth);
    Planner P{ctx, e, {}};
    P.gn_eps = e->bn1.eps; P.gn_groups = e->bn1.groups;
    P.fork = ctx->opt.enc_fork != 0;
    if (P.fork && !e->side_stream) AVC_HIP(hipStreamCreateWithFlags(&e->side_stream, hipStreamNonBlocking));
    e->in_buf = static_cast<float *>(P.alloc(sizeof(float) * 6 * (size_t)Hin * Win));
    e->range_flag = static_cast<unsigned *>(P.alloc(sizeof(unsigned), true));
    // conv1 (space-to-depth, then 4x4 taps on the matrix pipe) + statistics of bn1
    Tensor sd = P.tensor(H1, W1, 32), t0 = P.tensor(H1, W1, 64);
    if (!P.rc) {
        Launch L{}; L.kind = L_S2D;
        L.s2d = S2dArgs{e->in_buf, Hin, Win, H1, W1, sd.data};
        L.grid = (unsigned)((H1 * W1 * 8 + 255) / 256);
        P.push(L);
    }
    P.conv(e->conv1, sd, nullptr, &t0, true, nullptr, nullptr, 0, 16.0f);         // the image is a normal map in [-1, 1]
    Tensor x = P.elementwise(L_NORMRELU, t0, nullptr, H1, W1, &e->bn1);          // relu(bn1(conv1 x))        HGFilters.py:178
    x = P.block(e->conv2, x);                                                    // 'no_down'                  :184-185
    e->normx = x;
    x = P.block(e->conv3, x);
    x = P.block(e->conv4, x);
    x = P.level(e->depth, x);                                                    // the hourglass              :199
    x = P.block(e->top_m, x);                                                    //                            :202
    Tensor cl = P.tensor(H1, W1, 256), out = P.tensor(H1, W1, 32);
    P.conv(e->conv_last, x, nullptr, &cl, true, nullptr, nullptr, 0);            // conv_last0 on the raw block output   :204-205
    P.conv(e->l, cl, &e->bn_end, &out, false, nullptr, nullptr, 0);              // l0(relu(bn_end0(.)))                  :207
    e->out = out;
    e->plan_allocs = P.allocs;
    if (P.rc) { free_plan(e); return P.rc; }
    e->Hin = Hin; e->Win = Win; e->fork = ctx->opt.enc_fork; e->ksplit = ctx->opt.enc_ksplit; e->occ2 = ctx->opt.enc_occ2;
    // record the launches once as a hipGraph (replayed with one hipGraphLaunch per frame)
    if (ctx->opt.enc_graph) {
        if (!e->cap_stream) AVC_HIP(hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
        // first a plain run: hipFuncSetAttribute calls and module loading must not happen inside a capture
        AVC_HIP(hipDeviceSynchronize());                    // the zero-fills of the plan's counters (null stream) are done: the capture stream does not wait for them
        if (int rc = run_plan(e, e->cap_stream)) { free_plan(e); return rc; }
        AVC_HIP(hipStreamSynchronize(e->cap_stream));
        AVC_HIP(hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeThreadLocal));
        const int rc = run_plan(e, e->cap_stream);
        hipGraph_t g = nullptr;
        const hipError_t ee = hipStreamEndCapture(e->cap_stream, &g);
        if (rc) { if (g) hipGraphDestroy(g); free_plan(e); return rc; }
        AVC_HIP(ee);
        e->graph = g;
        AVC_HIP(hipGraphInstantiate(&e->exec, e->graph, nullptr, nullptr, 0));
    } else {
        // plain launches on the caller's stream: the null-stream zero-fills of the ticket counters / part2 tables must have landed first -- a non-blocking
        // stream does not wait for them, and a ticket that starts non-zero never reaches its count (the GroupNorm partials would never fold)
        AVC_HIP(hipDeviceSynchronize());
    }
    return AVC_OK;
}

int encoder_forward(avc_ctx *ctx, const float *image, int H, int W, float *feat_out, float *normx_out, int bind, hipStream_t s)
{
    Encoder *e = static_cast<Encoder *>(ctx->encoder);
    AVC_REQUIRE(e && e->packed, AVC_ERR_STATE, "avc_hgfilter_forward: no encoder weights (call avc_hgfilter_pack first)");
    AVC_REQUIRE(image && H >= 2 && W >= 2 && (int64_t)H * W <= (1 << 22), AVC_ERR_ARG, "avc_hgfilter_forward: NULL image or unsupported size %d x %d", H, W);
    if (e->Hin != H || e->Win != W || e->fork != ctx->opt.enc_fork || e->ksplit != ctx->opt.enc_ksplit || e->occ2 != ctx->opt.enc_occ2 || (ctx->opt.enc_graph != 0) != (e->exec != nullptr)) {
        // (re)building frees buffers a replay in flight may still use
        AVC_HIP(hipDeviceSynchronize());
        if (int rc = build_plan(ctx, e, H, W)) return rc;
    }
    AVC_HIP(hipMemcpyAsync(e->in_buf, image, sizeof(float) * 6 * (size_t)H * W, hipMemcpyDeviceToDevice, s));
    if (ctx->check_range) AVC_HIP(hipMemsetAsync(e->range_flag, 0, sizeof(unsigned), s));
    if (e->exec) AVC_HIP(hipGraphLaunch(e->exec, s));
    else if (int rc = run_plan(e, s)) return rc;
    if (ctx->check_range) {                                // avc_set_range_check: synchronous, like the queries'
        unsigned flag = 0;
        AVC_HIP(hipMemcpyAsync(&flag, e->range_flag, sizeof flag, hipMemcpyDeviceToHost, s));
        AVC_HIP(hipStreamSynchronize(s));
        AVC_REQUIRE(flag == 0, AVC_ERR_RANGE, "avc_hgfilter_forward: a normalised activation exceeded 65504 / 16 in magnitude (or a raw one 65504) -- outside the "
                    "range of the split-fp16 arithmetic (include/avcap.h, 'numeric range'); the feature map of this call is not valid");
    }
    const Tensor &o = e->out;
    const int HW = o.H * o.W;
    if (feat_out) hipLaunchKernelGGL(hwc_to_nchw_kernel, dim3((HW + 63) / 64, (o.C + 63) / 64), dim3(256), 0, s, o.data, feat_out, o.C, HW);
    if (normx_out) hipLaunchKernelGGL(hwc_to_nchw_kernel, dim3((HW + 63) / 64, (e->normx.C + 63) / 64), dim3(256), 0, s, e->normx.data, normx_out, e->normx.C, HW);
    if (bind) {
        if (!ctx->img_feat_hwc || ctx->img_C != o.C || ctx->img_H != o.H || ctx->img_W != o.W) {
            AVC_HIP(hipStreamSynchronize(s));          // a query on another stream may still read the old map
            if (ctx->img_feat_hwc) AVC_HIP(hipFree(ctx->img_feat_hwc));
            ctx->img_feat_hwc = nullptr;
            AVC_HIP(hipMalloc((void **)&ctx->img_feat_hwc, sizeof(float) * (size_t)o.C * HW));
            ctx->img_C = o.C; ctx->img_H = o.H; ctx->img_W = o.W;
        }
        AVC_HIP(hipMemcpyAsync(ctx->img_feat_hwc, o.data, sizeof(float) * (size_t)o.C * HW, hipMemcpyDeviceToDevice, s));
    }
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

// a tensor of the last forward, for tests: index into the launch plan -> the launch's raw / y output as NCHW
int encoder_debug_tensor(avc_ctx *ctx, int launch, int which, float *out, int *C, int *H, int *W, hipStream_t s)
{
    Encoder *e = static_cast<Encoder *>(which == 2 ? ctx->unet : ctx->encoder);
    AVC_REQUIRE(e && launch >= 0 && launch < (int)e->plan.size(), AVC_ERR_ARG, "avc_hgfilter_debug_tensor: launch %d of %d", launch, e ? (int)e->plan.size() : 0);
    const Launch &L = e->plan[launch];
    const float *src = nullptr; int c = 0, h = 0, w = 0;
    if (L.kind == L_CONV && which == 2) {                  // the U-Net's destination tensor (whole: the caller knows the channel slice the launch wrote)
        const OutSpec &o = L.conv.oa;
        const int up = o.layout == OUT_D2S ? 2 : 1;
        src = o.ptr; c = o.C; h = up * L.conv.H; w = up * L.conv.W;
    }
    else if (L.kind == L_CONV) { src = which ? L.conv.y : L.conv.raw; c = which ? L.conv.yC : L.conv.Cout; h = L.conv.H; w = L.conv.W; }
    else if (L.kind == L_POOL || L.kind == L_UPADD || L.kind == L_UPADD_TILED || L.kind == L_NORMRELU) { src = L.elt.out; c = L.elt.C; h = L.elt.H; w = L.elt.W; }
    else if (L.kind == L_UP2 && which == 2) { src = L.up2.out; c = L.up2.C; h = 2 * L.up2.H; w = 2 * L.up2.W; }
    else if (L.kind == L_S2D && which == 2) { src = L.s2d.out; c = 32; h = L.s2d.H; w = L.s2d.W; }
    *C = c; *H = h; *W = w;
    if (L.kind == L_CONV) *C = c | (L.CT << 16) | (L.PT << 20) | (L.TAPS << 24) | ((L.conv.ksplit > 1 ? 1 : 0) << 30);
    if (out && src) hipLaunchKernelGGL(hwc_to_nchw_kernel, dim3((h * w + 63) / 64, (c + 63) / 64), dim3(256), 0, s, src, out, c, h * w);
    AVC_HIP(hipGetLastError());
    return src ? AVC_OK : 1;
}


// =====================================================================================================================
// The warping field's U-Net (UnetNoCond7DS, network/unets.py:169-229; blocks :10-60) on the same convolution kernel
// =====================================================================================================================
// Every layer becomes ONE launch of conv_mfma_kernel<.., TAPS = 9, .., NORM = false> on tensors that are already in the layout the next layer reads:
//   (Of the 3x3 neighbourhood of a space-to-depth pixel only 2 x 2 taps meet the 4x4 kernel for a given input parity -- likewise for a given output parity
//   of a transposed convolution: those layers are packed and walked as 4 taps from a per-chunk / per-slice origin, TAPS == 4.)
//   Conv2DBlock (LeakyReLU(0.2) -> Conv2d k4 s2 p1 -> BatchNorm2d(affine=False)):  a stride-2 4x4 convolution is a 3x3 convolution (pad 1) of the
//     space-to-depth tensor (4 C channels per pixel: input row 2 o - 1 + k is row o - 1 + ty of parity py with k = 2 ty + py - 1); its producer writes the
//     s2d form directly (OUT_S2D), plus the plain form into its channel slice of the decoder tensor it will be concatenated into (OUT_NORMAL: torch.cat
//     costs nothing).  The BatchNorm (eval, running statistics) is folded into weights and bias; the pre-activation is applied while the input is staged.
//   UpConv2DBlock 'upconv' (ReLU -> ConvTranspose2d k4 s2 p1 -> BN -> cat):  output row 2 i + a takes input rows i - 1 + ty with k = a + 3 - 2 ty: a 3x3
//     convolution with 4 Cout parity-major output channels, scattered to the double-resolution tensor by the epilogue (OUT_D2S).
//   UpConv2DBlock 'upsample' (ReLU -> bilinear x2 -> Conv2d 3x3 + bias -> BN -> cat):  up2_kernel, then a plain 3x3 convolution.
// 18 launches (one space-to-depth of the input, 7 + 4 + 3 convolutions, 3 upsamples) instead of MIOpen's ~70, replayed as a hipGraph; the last
// convolution's output IS the channel-last pose feature map the avatar query samples.  The reference's quirk is kept: upconv3 runs twice, upconv4 never
// (unets.py:213-214).
static int fold_bn(const avc_bn2d &bn, int cout, std::vector<float> &scale, std::vector<float> &shift, const char *name)
{
    scale.assign(cout, 1.0f); shift.assign(cout, 0.0f);
    if (!bn.mean) return AVC_OK;
    AVC_REQUIRE(bn.var, AVC_ERR_ARG, "avc_unet_pack: %s: running_var is NULL", name);
    for (int c = 0; c < cout; ++c) {
        AVC_REQUIRE(bn.var[c] + bn.eps > 0.0f, AVC_ERR_ARG, "avc_unet_pack: %s: running_var[%d] + eps <= 0", name, c);
        scale[c] = 1.0f / std::sqrt(bn.var[c] + bn.eps);
        shift[c] = -bn.mean[c] * scale[c];
    }
    return AVC_OK;
}

int pack_unet(avc_ctx *ctx, const avc_unet7ds *net)
{
    if (!ctx->unet) ctx->unet = new Encoder();
    Encoder *e = static_cast<Encoder *>(ctx->unet);
    free_plan(e);
    free_weights(e);
    char nm[48];
    std::vector<float> sc, sh;
    // ---- conv1..7: Conv2d(ci, co, 4, 2, 1, bias=False) as 3x3 over the 4 ci (first layer: 24 padded to 32) space-to-depth channels
    for (int l = 0; l < 7; ++l) {
        const avc_conv2d &c = net->down[l];
        snprintf(nm, sizeof nm, "conv%d", l + 1);
        AVC_REQUIRE(c.w && c.kh == 4 && c.kw == 4 && !c.b && c.cout % 32 == 0, AVC_ERR_ARG, "avc_unet_pack: %s must be Conv2d(k4, s2, p1, bias=False) with cout a multiple of 32", nm);
        AVC_REQUIRE(l == 0 ? c.cin * 4 <= 32 : (c.cin == net->down[l - 1].cout), AVC_ERR_ARG, "avc_unet_pack: %s: %d input channels", nm, c.cin);
        if (int rc = fold_bn(net->down_bn[l], c.cout, sc, sh, nm)) return rc;
        const int ci4 = l == 0 ? 32 : 4 * c.cin;
        if (l > 0 && c.cin % 32 == 0) {
            // a chunk of 32 space-to-depth channels lies inside one parity (py, px): only the 2 x 2 taps (1 - py + a, 1 - px + b) of its 3 x 3 neighbourhood
            // meet the 4 x 4 kernel (k = 1 - py + 2 a): packed and walked as 4 taps from that origin -- 4/9 of the bytes and MFMAs of the 3 x 3 form
            std::vector<float> w((size_t)c.cout * ci4 * 4, 0.0f);
            for (int co = 0; co < c.cout; ++co)
                for (int par = 0; par < 4; ++par)
                    for (int ci = 0; ci < c.cin; ++ci)
                        for (int t = 0; t < 4; ++t) {
                            const int ky = 1 - (par >> 1) + 2 * (t >> 1), kx = 1 - (par & 1) + 2 * (t & 1);
                            w[((size_t)co * ci4 + par * c.cin + ci) * 4 + t] = c.w[(((size_t)co * c.cin + ci) * 4 + ky) * 4 + kx] * sc[co];
                        }
            const avc_conv2d c2{w.data(), sh.data(), c.cout, ci4, 2, 2};
            if (int rc = pack_conv(e, c2, 4, e->u_down[l], nm, 2 * conv_ct(c.cout) - 1)) return rc;
            continue;
        }
        std::vector<float> w((size_t)c.cout * ci4 * 9, 0.0f);
        for (int co = 0; co < c.cout; ++co)
            for (int par = 0; par < 4; ++par)
                for (int ci = 0; ci < c.cin; ++ci)
                    for (int ty = 0; ty < 3; ++ty)
                        for (int tx = 0; tx < 3; ++tx) {
                            const int ky = 2 * ty + (par >> 1) - 1, kx = 2 * tx + (par & 1) - 1;
                            if (ky < 0 || ky > 3 || kx < 0 || kx > 3) continue;
                            w[((size_t)co * ci4 + par * c.cin + ci) * 9 + ty * 3 + tx] = c.w[(((size_t)co * c.cin + ci) * 4 + ky) * 4 + kx] * sc[co];
                        }
        const avc_conv2d c3{w.data(), sh.data(), c.cout, ci4, 3, 3};
        if (int rc = pack_conv(e, c3, 9, e->u_down[l], nm, 2 * conv_ct(c.cout) - 1)) return rc;
    }
    // ---- upconv1..3: ConvTranspose2d(ci, co, 4, 2, 1, bias=False), weight (ci, co, 4, 4), as 3x3 with 4 co parity-major outputs
    for (int l = 0; l < 3; ++l) {
        const avc_conv2d &c = net->up[l];
        snprintf(nm, sizeof nm, "upconv%d", l + 1);
        AVC_REQUIRE(c.w && c.kh == 4 && c.kw == 4 && !c.b && c.cout % 32 == 0 && c.cin % 32 == 0, AVC_ERR_ARG, "avc_unet_pack: %s must be ConvTranspose2d(k4, s2, p1, bias=False)", nm);
        if (int rc = fold_bn(net->up_bn[l], c.cout, sc, sh, nm)) return rc;
        // output parity (a, b) takes the 2 x 2 taps (a + ta, b + tb) of the 3 x 3 neighbourhood, with k = 3 - a - 2 ta: 4 taps from an origin per output slice
        std::vector<float> w((size_t)4 * c.cout * c.cin * 4, 0.0f), b((size_t)4 * c.cout);
        for (int par = 0; par < 4; ++par)
            for (int co = 0; co < c.cout; ++co) {
                b[(size_t)par * c.cout + co] = sh[co];
                for (int ci = 0; ci < c.cin; ++ci)
                    for (int t = 0; t < 4; ++t) {
                        const int ky = 3 - (par >> 1) - 2 * (t >> 1), kx = 3 - (par & 1) - 2 * (t & 1);
                        w[(((size_t)par * c.cout + co) * c.cin + ci) * 4 + t] = c.w[(((size_t)ci * c.cout + co) * 4 + ky) * 4 + kx] * sc[co];
                    }
            }
        const avc_conv2d c2{w.data(), b.data(), 4 * c.cout, c.cin, 2, 2};
        if (int rc = pack_conv(e, c2, 4, e->u_up[l], nm, 7)) return rc;
    }
    // ---- upconvC5..C7: Conv2d(ci, co, 3, 1, 1, bias=True) behind the bilinear upsample
    for (int l = 0; l < 3; ++l) {
        const avc_conv2d &c = net->upc[l];
        snprintf(nm, sizeof nm, "upconvC%d", l + 5);
        AVC_REQUIRE(c.w && c.b && c.kh == 3 && c.kw == 3 && c.cout % 32 == 0 && c.cin % 32 == 0, AVC_ERR_ARG, "avc_unet_pack: %s must be Conv2d(3x3, p1) with a bias", nm);
        if (int rc = fold_bn(net->upc_bn[l], c.cout, sc, sh, nm)) return rc;
        std::vector<float> w((size_t)c.cout * c.cin * 9), b(c.cout);
        for (int co = 0; co < c.cout; ++co) {
            b[co] = c.b[co] * sc[co] + sh[co];
            for (size_t i = 0; i < (size_t)c.cin * 9; ++i) w[(size_t)co * c.cin * 9 + i] = c.w[(size_t)co * c.cin * 9 + i] * sc[co];
        }
        const avc_conv2d c3{w.data(), b.data(), c.cout, c.cin, 3, 3};
        if (int rc = pack_conv(e, c3, 9, e->u_upc[l], nm, 2 * conv_ct(c.cout) - 1)) return rc;
    }
    // the decoder's concatenations must fit together (unets.py:209-219)
    const int c6 = e->u_down[5].cout, c5 = e->u_down[4].cout, c4 = e->u_down[3].cout, c3 = e->u_down[2].cout, c2 = e->u_down[1].cout, c1 = e->u_down[0].cout;
    const int u1 = e->u_up[0].cout / 4, u2 = e->u_up[1].cout / 4, u3 = e->u_up[2].cout / 4;
    AVC_REQUIRE(e->u_up[0].cin == e->u_down[6].cout && e->u_up[1].cin == u1 + c6 && e->u_up[2].cin == u2 + c5 && e->u_up[2].cin == u3 + c4 &&
                e->u_upc[0].cin == u3 + c3 && e->u_upc[1].cin == e->u_upc[0].cout + c2 && e->u_upc[2].cin == e->u_upc[1].cout + c1, AVC_ERR_ARG,
                "avc_unet_pack: the layers' channel counts do not chain as UnetNoCond7DS.forward concatenates them (unets.py:201-219)");
    e->packed = true;
    return AVC_OK;
}

static int build_unet_plan(avc_ctx *ctx, Encoder *e, int Hin, int Win)
{
    free_plan(e);
    AVC_REQUIRE(Hin % 128 == 0 && Win % 128 == 0 && Hin >= 128 && Win >= 128, AVC_ERR_ARG,
                "avc_unet_forward: a %d x %d position map; seven stride-2 levels need multiples of 128 (the reference runs 256 x 256)", Hin, Win);
    Planner P{ctx, e, {}};
    e->in_buf = static_cast<float *>(P.alloc(sizeof(float) * 6 * (size_t)Hin * Win));
    e->range_flag = static_cast<unsigned *>(P.alloc(sizeof(unsigned), true));
    // one convolution: x (raw, pre-activation `slope`) -> the two generic outputs
    auto conv = [&](const DevConv &w, const Tensor &x, float slope, OutSpec oa, OutSpec ob) {
        if (P.rc) return;
        Launch L{}; L.kind = L_CONV; L.TAPS = w.taps; L.norm = false;
        L.TWC = x.W >= 32 ? 32 : 16;
        const int nchunk = x.C / 32;
        auto tiles = [&](int PT) { const int rows = 4 * PT * (32 / L.TWC); return ((x.H + rows - 1) / rows) * ((x.W + L.TWC - 1) / L.TWC); };
        L.CT = conv_ct(w.cout);
        L.PT = (L.TWC == 32 && tiles(2) * (w.cout / (32 * L.CT)) >= ctx->num_cus) ? 2 : 1;
        int ks = 1;
        // The deep levels are a few pixels wide and stream megabytes of weights: a workgroup moves ~15 GB/s of them through its LDS ring, so the
        // launch is as fast as it has workgroups.  Narrower channel slices (CT) and a deeper split of K until the chip is full; the larger CT on a tie.
        if (ctx->opt.enc_ksplit && tiles(L.PT) * (w.cout / (32 * L.CT)) < ctx->num_cus) {
            int best = 0;
            constexpr int ks_cap = 8;                          // beyond 8 slices the exchange of partial sums costs more than the shorter K walk saves (4 / 8 / 16: 0.282 / 0.263 / 0.297 ms per map)
            for (int ct = L.CT; ct >= 1; ct >>= 1) {
                if (!(w.ct_mask & ct)) continue;
                const int wg0 = tiles(1) * (w.cout / (32 * ct));
                int k = 1;
                while (nchunk % (2 * k) == 0 && k < ks_cap && 2 * wg0 * k <= ctx->num_cus) k *= 2;
                if (wg0 * k > best) { best = wg0 * k; L.CT = ct; ks = k; }
            }
            L.PT = 1;
        }
        const int rows = 4 * L.PT * (32 / L.TWC);
        ConvArgs &a = L.conv;
        a.x = x.data; a.H = x.H; a.W = x.W; a.Cin = x.C;
        a.in_cpg = 1; a.in_scale = 16.0f; a.in_slope = slope;
        const int v = L.CT == 4 ? 2 : (L.CT == 2 ? 1 : 0);
        a.slice_bytes = (unsigned)nchunk * w.taps * 2 * L.CT * 2048;
        a.wstream = w.wstream + w.off[v]; a.wbytes = a.slice_bytes * (w.cout / (32 * L.CT));
        a.bias = w.bias; a.out_scale = w.wscale_inv / a.in_scale; a.Cout = w.cout;
        a.tiles_x = (x.W + L.TWC - 1) / L.TWC; a.tiles_y = (x.H + rows - 1) / rows;
        a.oa = oa; a.ob = ob;
        if (w.taps == 4) {                                     // by input parity for the stride-2 convolutions, by output parity for the transposed ones
            a.tap_mode = oa.layout == OUT_D2S ? 2 : 1;
            a.tap_div = oa.layout == OUT_D2S ? w.cout / 4 : nchunk / 4;
        }
        a.range_flag = e->range_flag;
        const int wg = a.tiles_x * a.tiles_y * (w.cout / (32 * L.CT));
        a.ksplit = ks;
        if (a.ksplit > 1) {
            a.kpart = static_cast<float *>(P.alloc(sizeof(float) * (size_t)wg * a.ksplit * 256 * L.PT * L.CT * 16));
            a.kcounter = static_cast<unsigned *>(P.alloc(sizeof(unsigned) * wg, true));
        }
        L.grid = (unsigned)(wg * a.ksplit);
        P.push(L);
    };
    const int H1 = Hin / 2, W1 = Win / 2;
    // the concatenated decoder tensors [up part | skip] at each resolution, and the space-to-depth forms of the encoder's outputs
    const DevConv *dn = e->u_down;
    const int uq[3] = {e->u_up[0].cout / 4, e->u_up[1].cout / 4, e->u_up[2].cout / 4};
    Tensor x0 = P.tensor(H1, W1, 32);                                          // s2d of the (6, H, W) input
    Tensor cat6 = P.tensor(H1, W1, e->u_upc[1].cout + dn[0].cout);             // [upC6 | d1]   128^2
    Tensor cat5 = P.tensor(H1 / 2, W1 / 2, e->u_upc[0].cout + dn[1].cout);     // [upC5 | d2]   64^2
    Tensor cat4 = P.tensor(H1 / 4, W1 / 4, uq[2] + dn[2].cout);                // [upconv3' | d3]
    Tensor cat3 = P.tensor(H1 / 8, W1 / 8, uq[2] + dn[3].cout);                // [upconv3 | d4]
    Tensor cat2 = P.tensor(H1 / 16, W1 / 16, uq[1] + dn[4].cout);              // [upconv2 | d5]
    Tensor cat1 = P.tensor(H1 / 32, W1 / 32, uq[0] + dn[5].cout);              // [upconv1 | d6]
    Tensor d7 = P.tensor(H1 / 64, W1 / 64, dn[6].cout);
    Tensor *cats[6] = {&cat6, &cat5, &cat4, &cat3, &cat2, &cat1};
    Tensor s2d[6];                                                              // d1..d6 in the layout conv2..conv7 read
    for (int l = 0; l < 6; ++l) s2d[l] = P.tensor(H1 >> (l + 1), W1 >> (l + 1), 4 * dn[l].cout);
    if (!P.rc) {
        Launch L{}; L.kind = L_S2D;
        L.s2d = S2dArgs{e->in_buf, Hin, Win, H1, W1, x0.data};
        L.grid = (unsigned)((H1 * W1 * 8 + 255) / 256);
        P.push(L);
    }
    // encoder: conv1 has no activation (use_relu=False), the others LeakyReLU(0.2) on their input (unets.py:72-78, 22-23)
    for (int l = 0; l < 6; ++l) {
        const Tensor &in = l == 0 ? x0 : s2d[l - 1];
        Tensor &cat = *cats[l];
        conv(dn[l], in, l == 0 ? 1.0f : 0.2f, OutSpec{cat.data, OUT_NORMAL, cat.C, cat.C - dn[l].cout}, OutSpec{s2d[l].data, OUT_S2D, s2d[l].C, 0});
    }
    conv(dn[6], s2d[5], 0.2f, OutSpec{d7.data, OUT_NORMAL, d7.C, 0}, OutSpec{});
    // decoder: ReLU on the (concatenated) input, transposed convolutions scattered into the up part of the next concatenation
    conv(e->u_up[0], d7, 0.0f, OutSpec{cat1.data, OUT_D2S, cat1.C, 0}, OutSpec{});
    conv(e->u_up[1], cat1, 0.0f, OutSpec{cat2.data, OUT_D2S, cat2.C, 0}, OutSpec{});
    conv(e->u_up[2], cat2, 0.0f, OutSpec{cat3.data, OUT_D2S, cat3.C, 0}, OutSpec{});
    conv(e->u_up[2], cat3, 0.0f, OutSpec{cat4.data, OUT_D2S, cat4.C, 0}, OutSpec{});               // upconv3 again, not upconv4 (unets.py:213-214)
    auto up2 = [&](const Tensor &x) {
        Tensor u = P.tensor(2 * x.H, 2 * x.W, x.C);
        if (P.rc) return u;
        Launch L{}; L.kind = L_UP2;
        L.up2 = Up2Args{x.data, u.data, x.H, x.W, x.C};
        L.grid = (unsigned)(((size_t)4 * x.H * x.W * (x.C / 4) + 255) / 256);
        P.push(L);
        return u;
    };
    Tensor out = P.tensor(Hin, Win, e->u_upc[2].cout);
    conv(e->u_upc[0], up2(cat4), 1.0f, OutSpec{cat5.data, OUT_NORMAL, cat5.C, 0}, OutSpec{});      // the ReLU went into the upsample
    conv(e->u_upc[1], up2(cat5), 1.0f, OutSpec{cat6.data, OUT_NORMAL, cat6.C, 0}, OutSpec{});
    conv(e->u_upc[2], up2(cat6), 1.0f, OutSpec{out.data, OUT_NORMAL, out.C, 0}, OutSpec{});
    e->out = out;
    e->plan_allocs = P.allocs;
    if (P.rc) { free_plan(e); return P.rc; }
    e->Hin = Hin; e->Win = Win; e->fork = ctx->opt.enc_fork; e->ksplit = ctx->opt.enc_ksplit; e->occ2 = ctx->opt.enc_occ2;
    if (ctx->opt.enc_graph) {
        if (!e->cap_stream) AVC_HIP(hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
        AVC_HIP(hipDeviceSynchronize());                    // the zero-fills of the plan's buffers (null stream) are done
        if (int rc = run_plan(e, e->cap_stream)) { free_plan(e); return rc; }
        AVC_HIP(hipStreamSynchronize(e->cap_stream));
        AVC_HIP(hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeThreadLocal));
        const int rc = run_plan(e, e->cap_stream);
        hipGraph_t g = nullptr;
        const hipError_t ee = hipStreamEndCapture(e->cap_stream, &g);
        if (rc) { if (g) hipGraphDestroy(g); free_plan(e); return rc; }
        AVC_HIP(ee);
        e->graph = g;
        AVC_HIP(hipGraphInstantiate(&e->exec, e->graph, nullptr, nullptr, 0));
    } else {
        AVC_HIP(hipDeviceSynchronize());
    }
    return AVC_OK;
}

int unet_forward(avc_ctx *ctx, const float *pos_map, int H, int W, float *out_nchw, int bind, hipStream_t s)
{
    Encoder *e = static_cast<Encoder *>(ctx->unet);
    AVC_REQUIRE(e && e->packed, AVC_ERR_STATE, "avc_unet_forward: no U-Net weights (call avc_unet_pack first)");
    AVC_REQUIRE(pos_map, AVC_ERR_ARG, "avc_unet_forward: NULL position map");
    if (e->Hin != H || e->Win != W || e->ksplit != ctx->opt.enc_ksplit || e->occ2 != ctx->opt.enc_occ2 || (ctx->opt.enc_graph != 0) != (e->exec != nullptr)) {
        AVC_HIP(hipDeviceSynchronize());
        if (int rc = build_unet_plan(ctx, e, H, W)) return rc;
    }
    AVC_HIP(hipMemcpyAsync(e->in_buf, pos_map, sizeof(float) * 6 * (size_t)H * W, hipMemcpyDeviceToDevice, s));
    if (ctx->check_range) AVC_HIP(hipMemsetAsync(e->range_flag, 0, sizeof(unsigned), s));
    if (e->exec) AVC_HIP(hipGraphLaunch(e->exec, s));
    else if (int rc = run_plan(e, s)) return rc;
    if (ctx->check_range) {
        unsigned flag = 0;
        AVC_HIP(hipMemcpyAsync(&flag, e->range_flag, sizeof flag, hipMemcpyDeviceToHost, s));
        AVC_HIP(hipStreamSynchronize(s));
        AVC_REQUIRE(flag == 0, AVC_ERR_RANGE, "avc_unet_forward: an activation exceeded 65504 / 16 in magnitude -- outside the range of the split-fp16 arithmetic "
                    "(include/avcap.h, 'numeric range'); the pose feature map of this call is not valid");
    }
    const Tensor &o = e->out;
    const int HW = o.H * o.W;
    if (out_nchw) hipLaunchKernelGGL(hwc_to_nchw_kernel, dim3((HW + 63) / 64, (o.C + 63) / 64), dim3(256), 0, s, o.data, out_nchw, o.C, HW);
    if (bind) {
        AVC_REQUIRE(o.C == 64, AVC_ERR_ARG, "avc_unet_forward: the avatar query samples 64 pose-feature channels, this U-Net produces %d", o.C);
        if (!ctx->pose_feat_hwc || ctx->pose_C != o.C || ctx->pose_H != o.H || ctx->pose_W != o.W) {
            AVC_HIP(hipStreamSynchronize(s));
            if (ctx->pose_feat_hwc) AVC_HIP(hipFree(ctx->pose_feat_hwc));
            ctx->pose_feat_hwc = nullptr;
            AVC_HIP(hipMalloc((void **)&ctx->pose_feat_hwc, sizeof(float) * (size_t)o.C * HW));
            ctx->pose_C = o.C; ctx->pose_H = o.H; ctx->pose_W = o.W;
        }
        AVC_HIP(hipMemcpyAsync(ctx->pose_feat_hwc, o.data, sizeof(float) * (size_t)o.C * HW, hipMemcpyDeviceToDevice, s));
    }
    AVC_HIP(hipGetLastError());
    return AVC_OK;
}

void release_unet(avc_ctx *ctx)
{
    Encoder *e = static_cast<Encoder *>(ctx->unet);
    if (!e) return;
    free_plan(e);
    free_weights(e);
    if (e->cap_stream) hipStreamDestroy(e->cap_stream);
    if (e->side_stream) hipStreamDestroy(e->side_stream);
    delete e;
    ctx->unet = nullptr;
}

}  // namespace enc
}  // namespace avc
