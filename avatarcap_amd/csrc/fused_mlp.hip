// Fused per-point MLP queries for gfx950 (MI355X):
//   avatar_kernel     : OccupancyNet.query  = WarpingField.query + DoubleTNet.forward
//                       (reference network/arch_avatar.py:356-381, :113-140, :65-83)
//   recon_kernel      : ReconNetwork.infer's decoder loop (network/arch_recon.py:55-73), point by point
//   recon_fold_kernel : the same decoder on a dense grid, the image-feature columns folded per (x, y) column
//
// Design:
//   * one workgroup = 4 waves (one per SIMD, ~400 of the 512 VGPR+AGPR each), persistent over 128-point tiles;
//     a wave owns 32 points and ALL hidden channels of them, so the whole 17-layer chain runs
//     out of registers: the D tile of one layer is, register for register, the B operand of the
//     next (mlp_layout.h).  No activation ever touches LDS or HBM.
//   * arithmetic: every fp32 product a*b is evaluated as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with
//     (hi, lo) fp16 pairs on v_mfma_f32_32x32x16_f16, fp32 accumulation (22+ significant bits,
//     ~1e-6 relative; the parity bar is 1e-4 absolute).  3 MFMA passes at 16x the fp32-MFMA rate.
//   * weights (3.4 MB, pre-split and pre-permuted by pack.cpp) stream L2 -> LDS by LDS-DMA in its buffer form
//     (`buffer_load_dwordx4 ... offen lds`: no staging VGPRs, no ds_write) into a 2 x 64 KiB ring, one chunk ahead of
//     the MFMAs that read it (ds_read_b128, lane-linear => conflict-free); one barrier per chunk.
//   * everything is a compile-time unrolled sequence of CHUNK STEPS.  Inside a chunk every k-step
//     issues, in the shadow of its 3*TPC MFMAs: the LDS reads of the next k-step's A fragments, a
//     slice of the next chunk's LDS-DMA, and a slice of the PREVIOUS tile pair's epilogue (bias or column term is
//     the accumulator init; activation, fp16 re-split), each pinned behind one MFMA, so VALU / LDS / VMEM work hides
//     behind the matrix pipe instead of serialising with it.  Layers with a short K (conv1 of a folded launch,
//     shared.0) are ONE wide chunk of all eight tiles; layers with a second input segment (conv5, shared.4) walk a
//     tile pair's k-steps as two chunks of equal size.
//   * fused prologue: bilinear gather of the channel-last feature map (point-by-point launches; dense grids take the
//     feature part of conv1 / conv5 as one fp32 vector per (x, y) column), positional encoding (accurate sincos),
//     the p + offset hand-off in fp32.
//   * at the board's power cap the launch time is set by energy, not cycles (profiles/r03_power_wall.md): the kernel
//     issues 92 - 94 % of the split-fp16 MFMA rate the part sustains with nothing else in the instruction stream.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "avcap_internal.h"
#include "mlp_layout.h"

// (Rounds 2 - 5 carried compile-time ablation knobs here and at ~35 sites of the kernel bodies -- no prefetch, no barrier, per-chunk s_memtime stamps, ... -- that
// attributed time and produced wrong results; what they measured is in profiles/r02_* / r03_avatar_time_split.md / r03_recon_time_split.md.  Removed in round 6:
// the product kernels contain no debugging code path; the device code of the default build is unchanged by the removal, checked instruction for instruction.)

// Opt-in range check (a second build of this file with -DAVC_CHECK_RANGE=1, selected at run time by avc_set_range_check): every value
// that is about to be split into fp16 halves -- sampled features, positional encodings, every post-activation value of every layer -- feeds
// a running max of magnitudes; a value above 65504 would become +-inf in its `hi` half and silently poison what follows (a ReLU swallows the
// NaN again), so the launch raises a flag instead and the query returns AVC_ERR_RANGE.  Costs one VALU per value pair: off by default.
#ifndef AVC_CHECK_RANGE
#define AVC_CHECK_RANGE 0
#endif
// Softplus layers whose weights were packed with a power-of-two scale (pack.cpp add_warp: a BatchNorm fold that leaves the fp16 range): a THIRD build of this
// file with -DAVC_LAYER_SCALE=1 multiplies the accumulator by the inverse (one v_mul per value, exact) before the Softplus; the default build has no such
// instruction and no such kernels -- a checkpoint inside the range runs the very code it always ran.  The range-checking build carries the multiply too.
#ifndef AVC_LAYER_SCALE
#define AVC_LAYER_SCALE AVC_CHECK_RANGE
#endif
#if AVC_CHECK_RANGE
#define AVC_FLAVOUR checked
#elif AVC_LAYER_SCALE
#define AVC_FLAVOUR scaled
#else
#define AVC_FLAVOUR plain
#endif

namespace avc {
namespace AVC_FLAVOUR {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// B operand of one k-step: 16 K-slots x 32 points as split fp16; each dword packs two halves
struct Frag { u32x4 hi, lo; };
__device__ __forceinline__ half8 as_half8(u32x4 v) { return __builtin_bit_cast(half8, v); }

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2, ACT_SOFTPLUS = 3 };

constexpr int WAVES = 4;
constexpr int TILE_PTS = 32 * WAVES;
constexpr int PARK_BASE = 2 * layout::SLOT_BYTES;     // per-wave 8 KiB: 4 input k-steps parked in LDS (see ParkIn)
constexpr int PARK_PER_WAVE = 4 * layout::UNIT_BYTES;
constexpr int LDS_BYTES = PARK_BASE + WAVES * PARK_PER_WAVE;   // 128 KiB weight ring + 32 KiB = the whole 160 KiB

constexpr int chunk_bytes(int ks, int tpc) { return ks * tpc * layout::UNIT_BYTES; }

struct QueryParams {
    const float *pts;        // (n,3)
    int64_t n;
    const float *feat;       // channel-last feature map (H, W, C)
    int H, W;
    float cx, cy, cz;
    const char *wstream;
    unsigned stream_bytes;   // bytes of one pass over the network (the prefetcher wraps here)
    const float *bias;
    float *out0;             // occ / recon value (n)
    float *out1;             // offsets (n,3) or null
    float *out2;             // rgba (n,4) or null
    int sigmoid_occ;
    int64_t ntiles;
    // dense-grid mode (pts == nullptr): point i = (gx[i / (gry*grz)], gy[(i / grz) % gry], gz[i % grz]) -- the flat order of
    // AvatarCapDataset.generate_volume_points (dataset/avatarcap_dataset.py:312-326); the three axis tables hold lin * (b1 - b0) + b0
    const float *gx, *gy, *gz;
    unsigned gry, grz;
    const int32_t *gidx;     // dense-grid mode, optional: the launch covers the n grid points gidx[0..n) (the valid band) instead of all of them
    unsigned *range_flag;    // AVC_CHECK_RANGE builds: set to 1 when a value left the fp16 range
    float sp_unscale;        // AVC_LAYER_SCALE builds: what undoes the Softplus layers' weight scale (1 when there is none)
    const float *colterms;   // column-folded dense launches: per (x, y) column 512 floats [conv1 | conv5] (column_terms_kernel), else null
    long long *clk;          // timed launches (avc_timing_enable): workgroup 0 stores its s_memtime at entry and exit here, else null
    // subset launches of the reconstruction query (band_prepass_kernel): the folded kernel leaves the tiles flagged in tile_skip (a wavefront with more than
    // two runs of columns) to the point-by-point kernel, which then runs on tile_list[0 .. *tile_count)
    const int32_t *tile_skip;
    const int32_t *tile_list, *tile_count;
};

// flat grid index of query point pidx (grid modes): the point's own number, or -- a SUBSET launch -- one load from the index list
__device__ __forceinline__ unsigned load_index(const QueryParams &p, int64_t pidx) { return p.gidx ? (unsigned)p.gidx[pidx] : (unsigned)pidx; }
// the point of flat grid index i from the three axis tables; returns its (x, y) column
__device__ __forceinline__ unsigned point_of_index(const QueryParams &p, unsigned i, float pt[3])
{
    const unsigned yz = p.gry * p.grz;
    const unsigned ix = i / yz, r = i - ix * yz, iy = r / p.grz, iz = r - iy * p.grz;
    pt[0] = p.gx[ix]; pt[1] = p.gy[iy]; pt[2] = p.gz[iz];
    return ix * p.gry + iy;
}
__device__ __forceinline__ unsigned load_point(const QueryParams &p, int64_t pidx, float pt[3])
{
    // returns the point's (x, y) column of the grid (grid modes), 0 otherwise
    if (p.pts) {
        pt[0] = p.pts[pidx * 3 + 0]; pt[1] = p.pts[pidx * 3 + 1]; pt[2] = p.pts[pidx * 3 + 2];
        return 0u;
    }
    return point_of_index(p, load_index(p, pidx), pt);
}
// A subset launch reads its points through TWO dependent loads (index list -> axis tables).  Fetched one tile ahead in one go, the second waits for the first
// at the top of every tile -- a full memory latency with nothing else to issue (round 4: the band launches ran 8 % (avatar) and 15 % (recon) more cycles per
// tile than the dense ones).  PointAhead keeps the chain two tiles deep: the INDEX of the tile after next is requested while the point of the next tile is
// built from the index requested a tile ago.
struct PointAhead {
    unsigned idx_next;           // flat grid index of this lane's point in the tile after next
    __device__ __forceinline__ int64_t lane_point(const QueryParams &p, int64_t tile, int wave, int j) const
    {
        const int64_t i = tile * TILE_PTS + wave * 32 + j;
        return i < p.n ? i : p.n - 1;
    }
    // before the loop: -> point of the first tile, index of the second requested
    __device__ __forceinline__ unsigned start(const QueryParams &p, int64_t tile0, int64_t stride, int wave, int j, float pt_next[3])
    {
        const unsigned i0 = load_index(p, lane_point(p, tile0, wave, j));
        const int64_t t1 = tile0 + stride < p.ntiles ? tile0 + stride : tile0;
        idx_next = load_index(p, lane_point(p, t1, wave, j));
        return point_of_index(p, i0, pt_next);
    }
    // top of tile `tile`: -> point of tile + stride (its index arrived during the previous tile), index of tile + 2 stride requested
    __device__ __forceinline__ unsigned advance(const QueryParams &p, int64_t tile, int64_t stride, int wave, int j, float pt_next[3])
    {
        const unsigned col = point_of_index(p, idx_next, pt_next);
        const int64_t t1 = tile + stride < p.ntiles ? tile + stride : tile;
        const int64_t t2 = t1 + stride < p.ntiles ? t1 + stride : t1;
        idx_next = load_index(p, lane_point(p, t2, wave, j));
        return col;
    }
};

// running max of the magnitudes that go through an fp16 split (AVC_CHECK_RANGE builds only)
struct RangeTrack {
    float amax = 0.0f;
#if AVC_LAYER_SCALE
    float unscale = 1.0f;      // 2^-s of the Softplus layers' weight scale (QueryParams::sp_unscale)
#endif
    __device__ __forceinline__ void see(float a, float b)
    {
#if AVC_CHECK_RANGE
        amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)));
#endif
    }
    __device__ __forceinline__ void report(const QueryParams &p) const
    {
#if AVC_CHECK_RANGE
        if (!(amax <= 65504.0f) && p.range_flag) atomicOr(p.range_flag, 1u);
#endif
    }
};

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}
struct NoSide { template <class K, class R> __device__ __forceinline__ void operator()(K, R) const {} };
template <int R> using RegionC = std::integral_constant<int, R>;

// B-operand providers of a chunk: k-step K -> fragment.
struct RegIn {                       // activations living in registers
    const Frag *f;
    template <int K> __device__ __forceinline__ Frag get() const { return f[K]; }
};
// The network inputs (sampled features / positional encoding) are needed twice, several layers
// apart (conv1 & conv5, shared.0 & shared.4).  Keeping them in VGPRs across the layers in between
// pushes the kernel over the 512-register budget, so their first 4 k-steps are parked in the
// wave's private 8 KiB of LDS (lane-linear 1 KiB blocks, conflict-free) and re-read just in time;
// an optional 5th k-step (raw xyz, mostly zeros) stays in registers.
struct ParkIn {
    unsigned base;                   // PARK_BASE + wave * PARK_PER_WAVE + lane * 16
    const Frag *extra;               // k-step 4 (may be null when KS == 4)
    template <int K> __device__ __forceinline__ Frag get() const
    {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        if constexpr (K < 4) {
            Frag r;
            r.hi = *reinterpret_cast<const u32x4 *>(smem + base + K * layout::UNIT_BYTES);
            r.lo = *reinterpret_cast<const u32x4 *>(smem + base + K * layout::UNIT_BYTES + 1024);
            return r;
        } else {
            return *extra;
        }
    }
};
struct ParkPeIn {                    // a warping field with a positional encoding in front (pos_encoding > 0): 4 parked feature k-steps + 4 k-steps of posenc(xyz) in registers
    unsigned base;
    const Frag *pe;
    template <int K> __device__ __forceinline__ Frag get() const
    {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        if constexpr (K < 4) {
            Frag r;
            r.hi = *reinterpret_cast<const u32x4 *>(smem + base + K * layout::UNIT_BYTES);
            r.lo = *reinterpret_cast<const u32x4 *>(smem + base + K * layout::UNIT_BYTES + 1024);
            return r;
        } else {
            return pe[K - 4];
        }
    }
};
struct SameIn {                      // every k-step takes the same fragment
    const Frag *f;
    template <int K> __device__ __forceinline__ Frag get() const { return *f; }
};
// Two providers back to back: k-steps 0 .. N0-1 are provider A's k-steps OFF .. OFF+N0-1, the following ones provider B's from 0 (the second chunk of
// a layer with a concatenated input: the tail of the hidden activations, then the raw network input -- see dense()).
template <int OFF, int N0, class A, class B>
struct CatIn {
    const A &a; const B &b;
    template <int K> __device__ __forceinline__ Frag get() const
    {
        if constexpr (K < N0) return a.template get<OFF + K>();
        else return b.template get<K - N0>();
    }
};
__device__ __forceinline__ void park_store(unsigned base, int k, const Frag &f)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    *reinterpret_cast<u32x4 *>(smem + base + k * layout::UNIT_BYTES) = f.hi;
    *reinterpret_cast<u32x4 *>(smem + base + k * layout::UNIT_BYTES + 1024) = f.lo;
}

// ------------------------------------------------------------------------------------------
// weight stream: 2-slot LDS ring, one chunk of prefetch, sizes known at compile time
// ------------------------------------------------------------------------------------------
struct Stream {
    RangeTrack range;
    __amdgpu_buffer_rsrc_t rs;   // buffer resource over the weight stream (LDS-DMA source)
    const char *gs;          // weight stream (wave-uniform)
    unsigned total;          // bytes per pass
    unsigned pf_off;         // offset of the next chunk to prefetch
    unsigned parity;         // ring slot of the chunk about to be consumed
    unsigned wave;           // wave index in the workgroup
    unsigned lane_off;       // lane * 16
};


// The next chunk travels L2 -> LDS by LDS-DMA in its BUFFER form, `buffer_load_dwordx4 v_off32, s[rsrc], s_off offen lds`: the data never
// touches a VGPR, there is no ds_write, and the only per-lane operand is one 32-bit offset register (lane * 16) that never changes.
// Measured beside the MFMA stream of this kernel's chunk step (tools/ubench/copy_cost.hip, profiles/r02_ubench_copy_cost.md): 35.9 cycles
// per MFMA against 34.6 with no copy at all -- while `global_load_lds_dwordx4` (64-bit address VGPR pair) costs 41.3, the saddr-form
// global_load + ds_write_b128 40.9, and what round 1 shipped (global_load with a 64-bit vaddr + ds_write_b128) cost the kernel 28 % of its
// time (round 2's ablation builds).  A chunk is covered in GROUPS of 16 KiB (a shorter last group for sizes that are not a multiple of 16 KiB);
// inside a group wave w owns a contiguous quarter, one 1 KiB piece per instruction: M0 = LDS destination, soffset = stream offset.
// Completion: every wave drains its own pieces (`s_waitcnt vmcnt(0)`) right before the chunk barrier that publishes them.
constexpr int GROUP = 16384;
constexpr int pf_group_bytes(int bytes, int g) { return bytes - g * GROUP >= GROUP ? GROUP : bytes - g * GROUP; }
constexpr int pf_slots(int bytes) { return 4 * ((bytes + GROUP - 1) / GROUP); }          // piece ids incl. holes
constexpr bool pf_valid(int bytes, int i) { return (i % 4) * 1024 < pf_group_bytes(bytes, i / 4) / WAVES; }
constexpr int pf_count(int bytes) { int n = 0; for (int i = 0; i < pf_slots(bytes); ++i) n += pf_valid(bytes, i); return n; }
constexpr int pf_nth(int bytes, int n) { for (int i = 0; i < pf_slots(bytes); ++i) { if (pf_valid(bytes, i)) { if (n == 0) return i; --n; } } return -1; }

// schedule of the n-th piece over the 3*KS issue slots of a chunk: issued at slot n * (slots - tail) / npw, i.e. spread evenly with the last
// `tail` slots left free so that the final piece has landed when the wave reaches the barrier
struct PfPlan { int slots, npw, tail; };
constexpr PfPlan pf_plan(int ks, int bytes)
{
    PfPlan p{3 * ks, pf_count(bytes), 0};
    if (p.npw == 0) return p;
    p.tail = p.slots >= 36 ? 12 : (p.slots >= 18 ? 6 : (p.slots >= 9 ? 4 : 2));
#ifdef AVC_PF_DIST_PCT
    p.tail = p.tail * AVC_PF_DIST_PCT / 100; if (p.tail < 1) p.tail = 1; if (p.tail > p.slots - 2) p.tail = p.slots - 2;
#endif
    return p;
}
constexpr int pf_load_slot(const PfPlan &p, int n) { return n * (p.slots - p.tail) / p.npw; }

typedef __attribute__((address_space(3))) void lds_void;

// piece I of the next chunk: stream offset `so` (+ group / wave / piece) -> LDS ring slot `dst` (same layout)
template <int BYTES, int I>
__device__ __forceinline__ void pf_dma(__amdgpu_buffer_rsrc_t rs, unsigned so, unsigned dst, unsigned lane16, unsigned wave)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int g = I / 4, j = I % 4, pw = pf_group_bytes(BYTES, g) / WAVES;
    // the instruction's immediate offset is added to BOTH the LDS and the memory address (tools/ubench/dma_semantics.hip), and the stream
    // layout is the LDS layout: M0 and soffset are set once per group, the four pieces differ only in the immediate
    const unsigned rel = g * GROUP + wave * pw;                                    // scalar
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(smem + (dst + rel)), 16, (int)lane16, (int)(so + rel), j * 1024, 0);
}

// work of issue slot SLOT: start the pieces scheduled here
template <int KS, int NEXT_BYTES, int SLOT>
__device__ __forceinline__ void pf_step(__amdgpu_buffer_rsrc_t rs, unsigned so, unsigned dst, unsigned lane16, unsigned wave)
{
    constexpr PfPlan P = pf_plan(KS, NEXT_BYTES);
    static_for<P.npw>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        if constexpr (pf_load_slot(P, n) == SLOT) pf_dma<NEXT_BYTES, pf_nth(NEXT_BYTES, n)>(rs, so, dst, lane16, wave);
    });
}

// every piece this wave has issued is in LDS (LDS-DMA completes in vmcnt order); the barrier that follows publishes them
__device__ __forceinline__ void pf_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// One chunk step: KS k-steps x TPC output tiles (units k-major in LDS).  Every k-step is three issue slots -- (hi,hi), (hi,lo), (lo,hi)
// products, one MFMA per output tile each, consecutive MFMAs never sharing an accumulator.  The side work is pinned BEHIND individual MFMAs
// with sched_barriers, a few instructions per MFMA: with one wave per SIMD about five single-issue instructions hide in the 32-cycle shadow of
// an MFMA, and whatever is clustered beyond that idles the matrix pipe (tools/ubench/copy_cost.hip: the same 16 VALU per k-step cost 43.2
// cycles per MFMA when clustered in one slot and 36.3 when spread over the six MFMAs).  Slot 0 issues the LDS reads of the next k-step's
// operands, every slot its share of the next chunk's LDS-DMA, and region r = 2 * slot + tile (TPC == 2) stage r of the PREVIOUS tile pair's
// epilogue slice (`side(k, r)`: accumulator read -> activation -> fp16 split, six stages).
// KACC: the accumulators of k-step k are acc[k * KACC ...] (0: every k-step accumulates into the same TPC tiles; 8: a chunk whose k-steps are
// DIFFERENT layers fed by the same B operand -- the folded recon kernel's [fc0 half | fc1's z column] chunk)
// `ap(step, tile, a_hi)` may replace the `hi` A fragment of one (step, tile) in the (hi, hi) pass: a k-step whose input has free slots -- the z k-step of
// the folded recon kernel -- carries a per-column term there, as fp16 (hi, lo) halves in two K slots against 1.0 in the B operand (ColPatch below).
struct NoPatch { template <class Q, class T> __device__ __forceinline__ half8 operator()(Q, T, half8 a) const { return a; } };
template <int KS, int TPC, int NEXT_BYTES, int KACC = 0, class In, class Side, class AP = NoPatch>
__device__ __forceinline__ void chunk(Stream &s, const In &in, f32x16 *__restrict__ acc, Side &&side, AP &&ap = AP{})
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // acquire: every wave has stored its share of this chunk (ds_write) and finished reading the other slot
    pf_drain();
    __syncthreads();
    // Values every address of this chunk derives from are made opaque HERE, after the barrier: with
    // everything unrolled and compile-time, the compiler otherwise hoists the address arithmetic of all
    // ~60 chunks of a tile (hundreds of values) to the top of the tile loop and spills them.
    unsigned base = s.parity * layout::SLOT_BYTES + s.lane_off;
    unsigned so = s.pf_off;
    unsigned dst = (s.parity ^ 1u) * layout::SLOT_BYTES;
    asm volatile("" : "+v"(base), "+s"(so), "+s"(dst));

    // The chunk is walked as Q steps of T2 (two, or one for the single-tile heads) output tiles: step q = k-step q / G, tile group q % G.  Units are
    // k-major in LDS, so step q's A fragments are units q * T2 .. q * T2 + T2 - 1 whatever TPC is -- an 8-tile chunk (recon fc1) is the same loop as
    // a 2-tile chunk with four times the k-steps, its B fragment changing every G steps: only the fragments of one tile pair and of the next are
    // live (32 registers; holding a whole 8-tile k-step and its successor was 128, and the recon kernel spilled 350 registers for it).
    constexpr int T2 = TPC >= 2 ? 2 : 1, G = TPC / T2, Q = KS * G;
    static_assert(TPC == 1 || TPC % 2 == 0, "tiles per chunk: 1 or even");
    constexpr bool SIDE = TPC == 2 || (T2 == 2 && !std::is_same<typename std::decay<Side>::type, NoSide>::value);   // side(step, region) hooks behind every MFMA
    half8 ah[2][T2], al[2][T2];
    Frag b[2];
    b[0] = in.template get<0>();
#pragma unroll
    for (int t = 0; t < T2; ++t) {
        ah[0][t] = *reinterpret_cast<const half8 *>(smem + base + t * layout::UNIT_BYTES);
        al[0][t] = *reinterpret_cast<const half8 *>(smem + base + t * layout::UNIT_BYTES + 1024);
    }
    static_for<Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int cur = q & 1, nxt = cur ^ 1;
        f32x16 *const a2 = acc + (q % G) * T2 + (q / G) * KACC;
        // ---- slot 0
        // ONE wait for this step's operands (requested a whole step ago), before the next step's reads go out: hipcc would otherwise put
        // a counted `s_waitcnt lgkmcnt(n)` in front of every MFMA that touches a freshly read fragment, four issue slots per step
        __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0), vmcnt / expcnt untouched
        if constexpr (q + 1 < Q) {
            b[nxt] = in.template get<(q + 1) / G>();
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                ah[nxt][t] = *reinterpret_cast<const half8 *>(smem + base + ((q + 1) * T2 + t) * layout::UNIT_BYTES);
                al[nxt][t] = *reinterpret_cast<const half8 *>(smem + base + ((q + 1) * T2 + t) * layout::UNIT_BYTES + 1024);
            }
        }
        pf_step<Q, NEXT_BYTES, 3 * q + 0>(s.rs, so, dst, s.lane_off, s.wave);
        static_for<T2>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            a2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap(qc, tc, ah[cur][t]), as_half8(b[cur].hi), a2[t], 0, 0, 0);
            if constexpr (SIDE) { side(qc, RegionC<t>{}); __builtin_amdgcn_sched_barrier(0); }
        });
        __builtin_amdgcn_sched_barrier(0);
        // ---- slot 1
        pf_step<Q, NEXT_BYTES, 3 * q + 1>(s.rs, so, dst, s.lane_off, s.wave);
        static_for<T2>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            a2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][t], as_half8(b[cur].lo), a2[t], 0, 0, 0);
            if constexpr (SIDE) { side(qc, RegionC<2 + t>{}); __builtin_amdgcn_sched_barrier(0); }
        });
        __builtin_amdgcn_sched_barrier(0);
        // ---- slot 2
        pf_step<Q, NEXT_BYTES, 3 * q + 2>(s.rs, so, dst, s.lane_off, s.wave);
        static_for<T2>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            a2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][t], as_half8(b[cur].hi), a2[t], 0, 0, 0);
            if constexpr (SIDE) { side(qc, RegionC<4 + t>{}); __builtin_amdgcn_sched_barrier(0); }
        });
        __builtin_amdgcn_sched_barrier(0);
    });
    const unsigned no = s.pf_off + NEXT_BYTES;
    s.pf_off = no >= s.total ? 0u : no;
    s.parity ^= 1u;
}

// ------------------------------------------------------------------------------------------
// epilogue pieces
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 bias_tile(const float *bias_rows, int h)
{
    // rows d_row(r,h) = (r&3) + 8*(r>>2) + 4h : four aligned float4 per lane
    f32x16 a;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) f32x4 *>(
            (const __attribute__((address_space(1))) float *)bias_rows + 8 * m + 4 * h);
        a[4 * m + 0] = v[0]; a[4 * m + 1] = v[1]; a[4 * m + 2] = v[2]; a[4 * m + 3] = v[3];
    }
    return a;
}

// Bias sources for the accumulator init of a tile pair.
//   BiasDirect : load when needed.  The loads sit right in front of the chunk's `vmcnt(0)` + barrier,
//                so their full L2 latency is exposed once per pair (recon kernel: not the headline).
//   BiasQueue  : the 64 floats of the NEXT pair are fetched right after the barrier of the CURRENT
//                chunk and sit in 32 registers until needed; the table is in consumption order (pack.cpp)
//                so "next" is simply the following block.  `rewind` restarts at the table head for the
//                next point tile.  Removes a ~1-2 k cycle stall in front of every chunk.
struct BiasDirect {
    const float *p;
    __device__ __forceinline__ void take(f32x16 *acc, int ntiles, int h)
    {
        acc[0] = bias_tile(p, h);
        if (ntiles > 1) acc[1] = bias_tile(p + 32, h);
        p += 32 * ntiles;
    }
    __device__ __forceinline__ void after_barrier(int) {}
    __device__ __forceinline__ void rewind(const float *) {}
};
// The queue's loads are BUFFER loads: resource built from the (wave-uniform) block pointer, one per-lane byte offset register that never changes
// (16 h) and immediates.  The flat form `global_load_dwordx4 v, v_off, s[base:base+1]` of round 2 had its
// address registers parked in AGPRs by the allocator and read back in front of every load: two v_accvgpr_read per load, ~900 per point tile.
// (declared as the LLVM intrinsic itself: hipcc 7.2's __builtin_amdgcn_raw_buffer_load_b128 selects a ONE-dword load for the 128-bit result)
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 raw_buffer_load_f32x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ float raw_buffer_load_f32(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ __forceinline__ f32x16 bias_tile_buf(i32x4 rs, unsigned voff, int byte0)
{
    f32x16 a;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f32x4 v = raw_buffer_load_f32x4(rs, (int)voff + byte0 + 32 * m, 0, 0);
        a[4 * m + 0] = v[0]; a[4 * m + 1] = v[1]; a[4 * m + 2] = v[2]; a[4 * m + 3] = v[3];
    }
    return a;
}
__device__ __forceinline__ i32x4 bias_rsrc(const float *block)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(block);
    i32x4 rs;
    rs[0] = (int)(unsigned)a; rs[1] = (int)((unsigned)(a >> 32) & 0xffffu);     // base, stride 0
    rs[2] = -16; rs[3] = 0x00027000;                                              // raw buffer, offsets up to 4 GiB
    return rs;
}
struct BiasQueue {
    const float *next;       // block the registers were loaded from
    unsigned hoff;           // 16 h: rows d_row(r, h) = (r & 3) + 8 (r >> 2) + 4 h of a tile are four 16-byte pieces, 32 bytes apart, from byte 16 h
    f32x16 nb[2];
    __device__ __forceinline__ void fetch(int)
    {
        const i32x4 rs = bias_rsrc(next);
        nb[0] = bias_tile_buf(rs, hoff, 0); nb[1] = bias_tile_buf(rs, hoff, 128);
    }
    __device__ __forceinline__ void take(f32x16 *acc, int ntiles, int)
    {
        acc[0] = nb[0];
        if (ntiles > 1) acc[1] = nb[1];
        next += 32 * ntiles;
    }
    __device__ __forceinline__ void after_barrier(int h) { fetch(h); }
    __device__ __forceinline__ void rewind(const float *head) { next = head; }
    // tile `t` of the current block's layer, loaded directly (wide layers: tiles 2..7 are requested early, tiles 0 / 1 come through the queue)
    __device__ __forceinline__ f32x16 tile_at(int t) const { return bias_tile_buf(bias_rsrc(next), hoff, 128 * t); }
    __device__ __forceinline__ void skip(int nfloats) { next += nfloats; }
};
// max(x, 0) as a signed-integer max on the bit pattern (negative floats are negative integers; -0 and negative NaNs -> +0): one v_max_i32 with an
// inline constant.  fmaxf() would put a canonicalising v_max_f32 in front, and an inline-asm v_max_f32 (round 2) is opaque to the hazard recogniser,
// which then pads every use of a transcendental result behind it with an s_nop.
__device__ __forceinline__ float relu_bits(float x)
{
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

__device__ __forceinline__ float softplus_f(float m)
{
    // torch.nn.Softplus(beta=1, threshold=20)   (network/mlp.py:99).  pack.cpp folds log2(e) into the
    // weights/bias of every Softplus layer and ln(2) into its consumers, so the accumulator already is
    // m = x*log2(e) and the activation carried between layers is y/ln2 = log2(1 + 2^m), evaluated as
    //     max(m, 0) + log2(1 + 2^-|m|)
    // on v_exp_f32 / v_log_f32 (the -|m| is a free source modifier): exact for any magnitude (large x
    // returns x, the reference's threshold branch; nothing overflows), absolute error ~1e-7.
    // The max is a bare v_max_f32: fmaxf() would add a canonicalising v_max in front of it, and at one
    // wave per SIMD every VALU instruction costs ~2.5 cycles of MFMA issue (tools/ubench/mfma_fill.hip).
    return relu_bits(m) + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-__builtin_fabsf(m)));
}

template <int ACT>
__device__ __forceinline__ float act_f(float x)
{
    if constexpr (ACT == ACT_RELU) return relu_bits(x);
    else if constexpr (ACT == ACT_LEAKY) return __builtin_fmaxf(x, 0.02f * x);    // LeakyReLU(0.02), network/mlp.py:11: max(x, 0.02 x) -- one multiply, one max
    else if constexpr (ACT == ACT_SOFTPLUS) return softplus_f(x);
    else return x;
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

// (x0, x1) -> packed fp16 pair `hi` = RN(x) and packed `lo` = RN(x - hi): 3 VALU for two values
// (v_cvt_pk_f16_f32 + two v_fma_mix*_f16 that take hi as an fp16 operand and write one half each).
// The conversion is left to the compiler on purpose: x may come straight out of v_log_f32, and the
// transcendental -> VALU wait state is inserted by hipcc for its own instructions only (an asm
// statement that consumed x first would read a stale register -- seen as NaN offsets).
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float x0, float x1, unsigned &hi, unsigned &lo)
{
    const half2_t hv = {(_Float16)x0, (_Float16)x1};
    hi = __builtin_bit_cast(unsigned, hv);
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(lo)
        : "v"(hi), "v"(x0), "v"(x1));
}

__device__ __forceinline__ void split8(const float *v, u32x4 &hi, u32x4 &lo, RangeTrack &range)
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned h, l;
        range.see(v[2 * e], v[2 * e + 1]);
        split2(v[2 * e], v[2 * e + 1], h, l);
        hi[e] = h; lo[e] = l;
    }
}

// Epilogue of a tile pair: accumulators -> activation -> split fp16, written into the 4 B fragments (k-steps) the pair becomes for the next
// layer; 32 values per lane.  The work is cut into NS slices (one per k-step of the chunk it hides in) of PPS value pairs, and every slice into
// SIX STAGES, one per MFMA of the k-step (chunk() pins stage r behind MFMA r).  Intermediate values travel in EpiRegs.
//   Softplus (log2 domain, see softplus_f):  0: x <- acc, e = 2^-|x|   1: e += 1   2: e = log2 e   3: y = max(x, 0) + e   4: hi = cvt_pk(y)   5: lo
//   ReLU / LeakyReLU / none:                 0: x <- acc               1: y = act(x)   2: hi = cvt_pk(y)   3: lo (low half)   4: lo (high half)
// Scalar f32 adds on purpose: packed v_pk_add_f32 beside MFMAs costs ~13 cycles each (MI355X_MICROARCH.md).
template <int PPS>
struct EpiRegs { float x[2 * PPS]; float e[2 * PPS]; unsigned hi[PPS], lo[PPS]; };

template <int ACT, int NS, int K, int R>
__device__ __forceinline__ void epi_part(const f32x16 *__restrict__ acc, Frag *__restrict__ out4, EpiRegs<(16 + NS - 1) / NS> &st, RangeTrack &range)
{
    constexpr int PPS = (16 + NS - 1) / NS;          // value PAIRS per slice (16 pairs per lane)
    static_for<PPS>([&](auto ic) {
        constexpr int i = decltype(ic)::value, pr = K * PPS + i;
        if constexpr (K < NS && pr < 16) {
            constexpr int v = 2 * pr, t = v >> 4, r = v & 15;
            float &x0 = st.x[2 * i], &x1 = st.x[2 * i + 1], &e0 = st.e[2 * i], &e1 = st.e[2 * i + 1];
            auto write_out = [&]() { out4[2 * t + (r >> 3)].hi[(r & 7) >> 1] = st.hi[i]; out4[2 * t + (r >> 3)].lo[(r & 7) >> 1] = st.lo[i]; };
            auto cvt = [&]() { range.see(x0, x1); const half2_t hv = {(_Float16)x0, (_Float16)x1}; st.hi[i] = __builtin_bit_cast(unsigned, hv); };
            if constexpr (ACT == ACT_SOFTPLUS) {
#if AVC_LAYER_SCALE
                if constexpr (R == 0) { x0 = acc[t][r] * range.unscale; x1 = acc[t][r + 1] * range.unscale;
                                        e0 = __builtin_amdgcn_exp2f(-__builtin_fabsf(x0)); e1 = __builtin_amdgcn_exp2f(-__builtin_fabsf(x1)); }
#else
                if constexpr (R == 0) { x0 = acc[t][r]; x1 = acc[t][r + 1]; e0 = __builtin_amdgcn_exp2f(-__builtin_fabsf(x0)); e1 = __builtin_amdgcn_exp2f(-__builtin_fabsf(x1)); }
#endif
                else if constexpr (R == 1) { e0 = e0 + 1.0f; e1 = e1 + 1.0f; }
                else if constexpr (R == 2) { e0 = __builtin_amdgcn_logf(e0); e1 = __builtin_amdgcn_logf(e1); }
                else if constexpr (R == 3) {
                    x0 = relu_bits(x0) + e0; x1 = relu_bits(x1) + e1;
                }
                else if constexpr (R == 4) cvt();
                else { unsigned l; asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                                        : "=&v"(l) : "v"(st.hi[i]), "v"(x0), "v"(x1)); st.lo[i] = l; write_out(); }
            } else {
                if constexpr (R == 0) { x0 = acc[t][r]; x1 = acc[t][r + 1]; }
                else if constexpr (R == 1) { x0 = act_f<ACT>(x0); x1 = act_f<ACT>(x1); }
                else if constexpr (R == 2) cvt();
                else if constexpr (R == 3) { unsigned l; asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(st.hi[i]), "v"(x0)); st.lo[i] = l; }
                else if constexpr (R == 4) { unsigned l = st.lo[i]; asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(st.hi[i]), "v"(x1));
                                             st.lo[i] = l; write_out(); }
            }
        }
    });
}

// deferred epilogue of a layer's LAST tile pair, to be run inside the first chunk of the next layer
template <int ACT, int NS>
struct Pending {
    const f32x16 *acc;
    Frag *out4;
    RangeTrack *range;
    EpiRegs<(16 + NS - 1) / NS> st;
    template <class KC, class RC>
    __device__ __forceinline__ void operator()(KC, RC) { epi_part<ACT, NS, KC::value, RC::value>(acc, out4, st, *range); }
};

// ------------------------------------------------------------------------------------------
// dense layers
// ------------------------------------------------------------------------------------------
// NT output tiles (even), evaluated two at a time; up to two input segments accumulate into the
// same tiles (the reference's torch.cat on the channel axis).  The epilogue of pair p runs inside
// the chunk(s) of pair p+1; the last pair's accumulators are handed back in `pend` and the caller
// schedules their epilogue (Pending) into whatever chunk comes next.  `pre` is such deferred work
// from the previous layer.  NEXT_BYTES = size of the chunk that follows this layer in the stream.
// A layer with a second segment (KS1 k-steps of raw network input behind KS0 k-steps of hidden activations: conv5, shared.4, recon fc2) walks a
// pair's KS0 + KS1 k-steps as TWO chunks of about equal size -- the units of a pair are one k-major run in the stream, so where it is cut is the
// kernel's choice.  Round 2 cut it at the segment boundary: a 64 KiB chunk followed by one of 1 .. 5 k-steps, whose 6 .. 30 MFMAs had to cover the
// LDS-DMA of the NEXT 64 KiB chunk: 1.4 k cycles for 6 MFMAs (profiles/r03_avatar_time_split.md).
constexpr int split_first_ks(int ks0, int ks1) { return ks1 > 0 ? (ks0 + ks1 + 1) / 2 : ks0; }
constexpr int first_chunk_bytes(int ks0, int ks1) { return chunk_bytes(split_first_ks(ks0, ks1), 2); }
struct NoPairPatch { template <class P, class T> __device__ __forceinline__ half8 operator()(P, T, half8 a) const { return a; } };
template <int NT, int KS0, int KS1, int ACT, int NEXT_BYTES, class In0, class In1, class Bias, class Pre, class PP = NoPairPatch>
__device__ __forceinline__ void dense(Stream &s, const In0 &in0, const In1 &in1,
                                      Frag *__restrict__ out, Bias &bias, int h,
                                      Pre &&pre, f32x16 *__restrict__ pend, const float *jump = nullptr, PP &&ppatch = PP{})
{
    // `ppatch(pair, tile, a_hi)`: A-fragment patch of the FIRST k-step of the second segment (see chunk())
    // `jump`: where the bias blocks continue after this layer's last pair, when not at the following block of the table (column-folded
    // launches take conv1's and conv5's blocks from the per-column table and everything else from the layer table)
    constexpr int NPAIR = NT / 2;
    constexpr int KT = KS0 + KS1, KA = split_first_ks(KS0, KS1), KB = KT - KA;
    static_assert(KA <= KS0, "the first chunk of a pair stays inside the first segment");
    constexpr int BA = chunk_bytes(KA, 2), BB = chunk_bytes(KB, 2);
    constexpr int NS = KT < 16 ? KT : 16;
    f32x16 prev[2];
    EpiRegs<(16 + NS - 1) / NS> est;
    static_for<NPAIR>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        f32x16 acc[2];
        bias.take(acc, 2, h);
        if constexpr (p == NPAIR - 1) { if (jump) bias.rewind(jump); }
        constexpr int afterA = KB > 0 ? BB : (p + 1 < NPAIR ? BA : NEXT_BYTES);
        chunk<KA, 2, afterA>(s, in0, acc, [&](auto kc, auto rc) {
            if constexpr (decltype(kc)::value == 0 && decltype(rc)::value == 5) bias.after_barrier(h);
            if constexpr (p == 0) pre(kc, rc);
            else epi_part<ACT, NS, decltype(kc)::value, decltype(rc)::value>(prev, out + 4 * (p - 1), est, s.range);
        });
        if constexpr (KB > 0) {
            constexpr int afterB = p + 1 < NPAIR ? BA : NEXT_BYTES;
            const CatIn<KA, KS0 - KA, In0, In1> tail{in0, in1};
            chunk<KB, 2, afterB>(s, tail, acc, [&](auto kc, auto rc) {
                if constexpr (p == 0) pre(std::integral_constant<int, KA + decltype(kc)::value>{}, rc);
                else epi_part<ACT, NS, KA + decltype(kc)::value, decltype(rc)::value>(prev, out + 4 * (p - 1), est, s.range);
            }, [&](auto qc, auto tc, half8 a) {
                if constexpr (decltype(qc)::value == KS0 - KA) return ppatch(pc, tc, a);
                else return a;
            });
        }
        prev[0] = acc[0]; prev[1] = acc[1];
    });
    pend[0] = prev[0]; pend[1] = prev[1];
}

// A layer with a SHORT K (conv1 of a column-folded launch: the xyz k-step; shared.0: the four k-steps of the positional encoding): all eight
// output tiles in ONE chunk (wide8), instead of four chunks of 6 .. 24 MFMAs that each carried a full epilogue and a barrier (round 2: 7.8 k cycles
// for conv1's 24 MFMAs).  Pair 0 is finished right away (flush), pairs 1 .. 3 inside the first chunk of the NEXT layer: pair q's four fragments
// out[4q .. 4q+3] are produced during k-steps 4(q-1) .. 4(q-1)+3, one fragment per k-step, four k-steps before they are read.
template <int ACT>
struct PendWide {
    const f32x16 *acc8;
    Frag *out;
    RangeTrack *range;
    EpiRegs<4> st;
    template <class KC, class RC>
    __device__ __forceinline__ void operator()(KC, RC)
    {
        constexpr int k = KC::value;
        if constexpr (k < 12) epi_part<ACT, 4, k % 4, RC::value>(acc8 + 2 * (1 + k / 4), out + 4 * (1 + k / 4), st, *range);
    }
};
// bias blocks of a wide layer: tiles 2 .. 7 are requested directly -- call this EARLY, their latency then hides behind the prologue work in front
// of the layer --, tiles 0 / 1 arrive through the queue (wide8)
template <class Bias>
__device__ __forceinline__ void wide_bias_early(const Bias &bias, f32x16 *acc8)
{
#pragma unroll
    for (int t = 2; t < 8; ++t) acc8[t] = bias.tile_at(t);
}
template <int KS, int NEXT_BYTES, class In, class Bias>
__device__ __forceinline__ void wide8(Stream &s, const In &in, f32x16 *__restrict__ acc8, Bias &bias, int h, const float *jump = nullptr)
{
    bias.take(acc8, 2, h);
    bias.skip(192);
    if (jump) bias.rewind(jump);
    chunk<KS, 8, NEXT_BYTES>(s, in, acc8, [&](auto qc, auto rc) {
        if constexpr (decltype(qc)::value == 0 && decltype(rc)::value == 5) bias.after_barrier(h);
    });
}

// run a deferred epilogue right away (no chunk to hide it in)
template <int ACT>
__device__ __forceinline__ void flush(const f32x16 *__restrict__ pend, Frag *__restrict__ out4, RangeTrack &range)
{
    EpiRegs<4> st;
    static_for<4>([&](auto kc) { static_for<6>([&](auto rc) { epi_part<ACT, 4, decltype(kc)::value, decltype(rc)::value>(pend, out4, st, range); }); });
}

// one-tile linear head (rows 0..31 of which only the first few are real): the three products of a
// k-step go to three accumulators, so no MFMA depends on its predecessor.  Same slot structure as chunk().
template <int KS, int NEXT_BYTES, bool LAST, class Bias, class Pre>
__device__ __forceinline__ f32x16 head(Stream &s, const Frag *__restrict__ in, Bias &bias, const float *bias_head, int h, Pre &&pre)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x16 a0, a1, a2;
    bias.take(&a0, 1, h);
    if constexpr (LAST) bias.rewind(bias_head);      // the next block is the first one of the next point tile
#pragma unroll
    for (int r = 0; r < 16; ++r) { a1[r] = 0.f; a2[r] = 0.f; }
    pf_drain();
    __syncthreads();
    unsigned base = s.parity * layout::SLOT_BYTES + s.lane_off;
    unsigned so = s.pf_off;
    unsigned dst = (s.parity ^ 1u) * layout::SLOT_BYTES;
    asm volatile("" : "+v"(base), "+s"(so), "+s"(dst));
    static_for<KS>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const half8 ah = *reinterpret_cast<const half8 *>(smem + base + k * layout::UNIT_BYTES);
        const half8 al = *reinterpret_cast<const half8 *>(smem + base + k * layout::UNIT_BYTES + 1024);
        if constexpr (k == 0) bias.after_barrier(h);
        pf_step<KS, NEXT_BYTES, 3 * k + 0>(s.rs, so, dst, s.lane_off, s.wave);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, as_half8(in[k].hi), a0, 0, 0, 0);
        pre(kc, RegionC<0>{}); pre(kc, RegionC<1>{});
        __builtin_amdgcn_sched_barrier(0);
        pf_step<KS, NEXT_BYTES, 3 * k + 1>(s.rs, so, dst, s.lane_off, s.wave);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, as_half8(in[k].lo), a1, 0, 0, 0);
        pre(kc, RegionC<2>{}); pre(kc, RegionC<3>{});
        __builtin_amdgcn_sched_barrier(0);
        pf_step<KS, NEXT_BYTES, 3 * k + 2>(s.rs, so, dst, s.lane_off, s.wave);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, as_half8(in[k].hi), a2, 0, 0, 0);
        pre(kc, RegionC<4>{}); pre(kc, RegionC<5>{});
        __builtin_amdgcn_sched_barrier(0);
    });
    const unsigned no = s.pf_off + NEXT_BYTES;
    s.pf_off = no >= s.total ? 0u : no;
    s.parity ^= 1u;
    return a0 + a1 + a2;
}

// ------------------------------------------------------------------------------------------
// input builders
// ------------------------------------------------------------------------------------------
// F.grid_sample(map, (gx, gy), 'bilinear', 'border', align_corners=True) of a channel-last map
// (arch_avatar.py:133, arch_recon.py:68): corner addresses + weights once, then 8 channels at a time
struct Bilinear {
    const float *p00, *p01, *p10, *p11;
    float w00, w01, w10, w11;
};

template <int C>
__device__ __forceinline__ Bilinear bilinear_setup(const float *__restrict__ feat, int H, int W, float gx, float gy, int c0)
{
    float ix = (gx + 1.0f) * 0.5f * (float)(W - 1);
    float iy = (gy + 1.0f) * 0.5f * (float)(H - 1);
    ix = __builtin_fminf(__builtin_fmaxf(ix, 0.0f), (float)(W - 1));
    iy = __builtin_fminf(__builtin_fmaxf(iy, 0.0f), (float)(H - 1));
    const float fx = __builtin_floorf(ix), fy = __builtin_floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const int x1 = x0 + 1 < W ? x0 + 1 : W - 1, y1 = y0 + 1 < H ? y0 + 1 : H - 1;
    const float tx = ix - fx, ty = iy - fy;
    Bilinear b;
    b.w00 = (1.0f - tx) * (1.0f - ty); b.w01 = tx * (1.0f - ty); b.w10 = (1.0f - tx) * ty; b.w11 = tx * ty;
    b.p00 = feat + ((size_t)y0 * W + x0) * C + c0;
    b.p01 = feat + ((size_t)y0 * W + x1) * C + c0;
    b.p10 = feat + ((size_t)y1 * W + x0) * C + c0;
    b.p11 = feat + ((size_t)y1 * W + x1) * C + c0;
    return b;
}

// 8 consecutive channels starting at channel offset c (relative to c0) -> one split fragment
__device__ __forceinline__ void bilinear_frag(const Bilinear &b, int c, Frag &f, RangeTrack &range)
{
    float v[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(b.p00 + c + 4 * q);
        const f32x4 bb = *reinterpret_cast<const f32x4 *>(b.p01 + c + 4 * q);
        const f32x4 cc = *reinterpret_cast<const f32x4 *>(b.p10 + c + 4 * q);
        const f32x4 d = *reinterpret_cast<const f32x4 *>(b.p11 + c + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[4 * q + i] = a[i] * b.w00 + bb[i] * b.w01 + cc[i] * b.w10 + d[i] * b.w11;
    }
    split8(v, f.hi, f.lo, range);
}

// sin and cos of q * S for S = 2^f, to ~1.6e-7 absolute (tools: see DESIGN.md section 2) in ~25 VALU
// instead of the ~300 of ocml's sincosf for arguments of a few hundred radians: the angle is reduced
// in REVOLUTIONS, q/(2 pi) as an unevaluated fp32 pair (p + e) via one FMA error term, so that the
// power-of-two scaling, the round-to-nearest and the subtraction are all exact; then quadrant + the
// classic degree-7/8 minimax polynomials on [-pi/4, pi/4].
__device__ __forceinline__ void sincos_pow2(float q, float S, float &sn, float &cs)
{
    const float C_HI = 0.15915494f, C_LO = 4.5929136e-09f;           // 1/(2 pi) = C_HI + C_LO
    const float p = q * C_HI;
    float e = __builtin_fmaf(q, C_HI, -p);
    e = __builtin_fmaf(q, C_LO, e);
    const float ps = p * S, es = e * S;
    const float r = (ps - __builtin_rintf(ps)) + es;                   // revolutions, |r| <= 0.5 (+ eps)
    const float kq = __builtin_rintf(4.0f * r);
    const float x = __builtin_fmaf(kq, -0.25f, r) * 6.2831853071795865f;   // [-pi/4, pi/4]
    const float z = x * x;
    const float sp = (-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f;
    const float sinp = __builtin_fmaf(sp * z, x, x);
    const float cp = (2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f;
    const float cosp = __builtin_fmaf(cp * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    const int iq = (int)kq & 3;
    const float a = (iq & 1) ? cosp : sinp, b = (iq & 1) ? sinp : cosp;
    sn = (iq & 2) ? -a : a;
    cs = ((iq + 1) & 2) ? -b : b;
}

// NeRF positional encoding of q (3 floats) into the 4 k-steps of the PE layout (mlp_layout.h):
// lane-half h evaluates arguments 15h .. 15h+14: coordinate i%3, frequency 2^(5h + i/3) -- exact
// power-of-two scaling like the reference's x * freq (net_util.py:27-33).
__device__ __forceinline__ void posenc_values(const float q[3], int h, float v[32])
{
    const float hs = h ? 32.0f : 1.0f;
#pragma unroll
    for (int i = 0; i < 15; ++i) {
        float sn, cs;
        sincos_pow2(q[i % 3], (float)(1 << (i / 3)) * hs, sn, cs);
        v[2 * i] = sn;
        v[2 * i + 1] = cs;
    }
    v[30] = h ? q[2] : q[0];
    v[31] = h ? 0.0f : q[1];
}
__device__ __forceinline__ void posenc(const float q[3], int h, unsigned park, RangeTrack &range)
{
    float v[32];
    posenc_values(q, h, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        Frag f;
        split8(v + 8 * k, f.hi, f.lo, range);
        park_store(park, k, f);
    }
}
// the same four k-steps into registers (the warping field's own positional encoding: its park area holds the sampled features)
__device__ __forceinline__ void posenc_frags(const float q[3], int h, Frag out[4], RangeTrack &range)
{
    float v[32];
    posenc_values(q, h, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) split8(v + 8 * k, out[k].hi, out[k].lo, range);
}

__device__ __forceinline__ Stream stream_init(const QueryParams &p, int wave, int lane, int first_bytes)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Stream s;
#if AVC_LAYER_SCALE
    s.range.unscale = p.sp_unscale;
#endif
    s.wave = wave; s.lane_off = lane * 16u;
    s.gs = p.wstream;
    s.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(p.wstream), 0, (int)p.stream_bytes, 0x00027000);   // raw buffer, range-checked at stream_bytes
    s.total = p.stream_bytes; s.parity = 0;
    // chunk 0 -> slot 0, synchronously (same group layout as the pipelined copy; run-time sizes)
    for (int g0 = 0; g0 < first_bytes; g0 += GROUP) {
        const int gb = first_bytes - g0 >= GROUP ? GROUP : first_bytes - g0, per_wave = gb / WAVES;
        for (int o = 0; o < per_wave; o += 1024) {
            const unsigned off = g0 + wave * per_wave + o + s.lane_off;
            *reinterpret_cast<u32x4 *>(smem + off) = *reinterpret_cast<const u32x4 *>(p.wstream + off);
        }
    }
    s.pf_off = first_bytes;
    return s;
}

// ------------------------------------------------------------------------------------------
// avatar query kernel
// ------------------------------------------------------------------------------------------
constexpr int B_MAIN = chunk_bytes(16, 2);       // 64 KiB: two tiles x 16 k-steps
constexpr int B_IN67 = chunk_bytes(layout::IN67_KS, 2);
constexpr int B_PE = chunk_bytes(layout::PE_KS, 2);
constexpr int B_HEAD16 = chunk_bytes(16, 1), B_HEAD8 = chunk_bytes(8, 1);
constexpr int B_XYZ8 = chunk_bytes(1, 8);        // column-folded launches: conv1 is the xyz k-step alone, all eight tiles in one chunk (wide8)
constexpr int B_PE8 = chunk_bytes(layout::PE_KS, 8);     // shared.0: four k-steps, all eight tiles in one chunk
constexpr int B_INPE = chunk_bytes(layout::INPE_KS, 2), B_CONV5PE = first_chunk_bytes(16, layout::INPE_KS);       // warping field with pos_encoding > 0
constexpr int B_CONV5 = first_chunk_bytes(16, layout::IN67_KS), B_CONV5F = first_chunk_bytes(16, 1), B_SHARED4 = first_chunk_bytes(16, layout::PE_KS);

// ---- column folding of a DENSE launch -----------------------------------------------------------------------------------------------
// The points of a dense grid run along the last axis, and the pose feature of WarpingField.query (arch_avatar.py:125-133) is sampled at
// (x, y) only: the 128 points of a tile share it when the last axis holds a multiple of 128 points.  What conv1 and conv5 (the skip
// connection, mlp.py:106) do with those 64 channels is then one vector per COLUMN and layer -- W[:, feat] f(x, y) + b, fp32, 65,536 columns
// at 256^3 -- computed once by column_terms_kernel and handed to the two layers as their accumulator init in place of the bias.  Their
// K shrinks from 5 k-steps to the xyz k-step: 192 of the 4,920 MFMAs of a tile, and the per-tile gather, split and parking of the 64
// channels, are gone.  Same algebra as the reference, different rounding (fp32 dot products instead of three fp16 products): a folded
// launch agrees with the point-by-point query to ~1e-6, not bit for bit (tests/test_gpu_query.py).
// (PackedNet::colw: [conv1 | conv5][out channel][feature channel], 2 x 256 x 64 floats)
__global__ __launch_bounds__(256) void column_terms_kernel(const float *__restrict__ feat, int H, int W, const float *__restrict__ gx,
                                                           const float *__restrict__ gy, int nx, int ny, float cx, float cy,
                                                           const float *__restrict__ colw, const float *__restrict__ bias1,
                                                           const float *__restrict__ bias5, float *__restrict__ out,
                                                           const uint8_t *__restrict__ colflag)
{
    // colflag (subset launches, else null): one byte per column, set for the columns the subset touches (band_prepass_kernel); the others are skipped
    constexpr int CPB = 4;                       // columns per trip
    __shared__ __attribute__((aligned(16))) float f[CPB][64];
    const int o = threadIdx.x;
    float w1[64], w5[64];
#pragma unroll
    for (int c = 0; c < 64; c += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(colw + (size_t)o * 64 + c), b = *reinterpret_cast<const f32x4 *>(colw + 256 * 64 + (size_t)o * 64 + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) { w1[c + i] = a[i]; w5[c + i] = b[i]; }
    }
    const float b1 = bias1[o], b5 = bias5[o];
    const int ncol = nx * ny;
    for (int c0 = blockIdx.x * CPB; c0 < ncol; c0 += gridDim.x * CPB) {
        if (colflag && *reinterpret_cast<const unsigned *>(colflag + c0) == 0u) continue;       // (CPB == 4 flags, table padded: workgroup-uniform)
        {   // the bilinear sample of arch_avatar.py:125-133, as the point-by-point kernel takes it: thread = (column, channel)
            const int q = threadIdx.x >> 6, ch = threadIdx.x & 63, col = min(c0 + q, ncol - 1);
            const Bilinear bl = bilinear_setup<64>(feat, H, W, gx[col / ny] - cx, -(gy[col % ny] - cy), 0);
            f[q][ch] = bl.p00[ch] * bl.w00 + bl.p01[ch] * bl.w01 + bl.p10[ch] * bl.w10 + bl.p11[ch] * bl.w11;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CPB; ++q) {
            if (c0 + q >= ncol) break;
            float a1 = b1, a5 = b5;
#pragma unroll
            for (int c = 0; c < 64; c += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(&f[q][c]);
#pragma unroll
                for (int i = 0; i < 4; ++i) { a1 = __builtin_fmaf(w1[c + i], v[i], a1); a5 = __builtin_fmaf(w5[c + i], v[i], a5); }
            }
            out[(size_t)(c0 + q) * 512 + o] = a1;
            out[(size_t)(c0 + q) * 512 + 256 + o] = a5;
        }
        __syncthreads();
    }
}


// ---- column terms of a SUBSET launch (FOLD == 2): several columns in a wave ------------------------------------------------------------------
// The 32 points of a wave lie in whatever (x, y) columns the valid band gives them; a band's runs along z keep a wave inside one or two.  The wave's
// points are cut into RUNS of equal adjacent columns (a column met twice is two runs: no ordering is assumed).  conv1's and conv5's accumulators are
// INITIALISED by one MFMA per tile: run r owns a pair of K slots; the A fragment holds, at that pair, its column's term of the tile's row split into
// fp16 halves (c_hi, c_lo), the B fragment 1.0 at the pair for exactly the lanes whose point belongs to the run -- D = c_hi + c_lo, an exact fp32 sum,
// for those points and 0 for the others, whatever pair the run got: a point's value does not depend on which points share its launch.  Six pairs per
// pass (slots 4..7 of the lanes h == 0, 8..15 of the lanes h == 1); a wave with more than six runs repeats the MFMA with the next six.  A lane fetches
// four floats per tile (256-byte wave-instructions).  Round 2 gathered a per-lane accumulator init instead: four `dwordx4` of 64 distinct addresses per
// tile through the texture path -- 3.9 ns per point against the dense launch's 3.4.
constexpr int SEG_PER_PASS = 6;
struct ColSegs {
    unsigned long long first;    // lanes 0..31: first point of a run
    unsigned nseg;               // runs in the wave (1..32)
    unsigned seg;                // the run of this lane's point
    unsigned col;                // this lane's column
};
__device__ __forceinline__ ColSegs col_segments(unsigned col, int j, int h)
{
    ColSegs c;
    c.col = col;
    const unsigned prev = (unsigned)__shfl_up((int)col, 1, 32);
    const bool first = (j == 0) || (prev != col);
    c.first = __builtin_amdgcn_ballot_w64(first && h == 0) & 0xffffffffull;
    c.nseg = (unsigned)__builtin_popcountll(c.first);
    c.seg = (unsigned)__builtin_popcountll(c.first & ((2ull << j) - 1ull)) - 1u;
    return c;
}
// what pass `pass` needs: the B-side ones of this lane (dword d of the fragment: 0x3C003C00 where the lane's run owns pair d of its half) and, per
// dword, the byte offset of the owning run's column in the table (512 floats per column; runs beyond the last: run 0's column, with no ones)
struct SegPass { u32x4 ones; unsigned colbyte[4]; };
template <unsigned COLBYTES = 2048u>          // bytes of one column in the table (avatar: 512 floats, recon: RCOL floats)
__device__ __forceinline__ SegPass seg_pass(const ColSegs &c, unsigned pass, int h)
{
    unsigned long long rest = c.first;
    for (unsigned n = 0; n < pass * SEG_PER_PASS; ++n) rest &= rest - 1ull;
    unsigned colr[SEG_PER_PASS];
#pragma unroll
    for (int i = 0; i < SEG_PER_PASS; ++i) {
        const int pos = rest ? __builtin_ctzll(rest) : 0;
        colr[i] = (unsigned)__builtin_amdgcn_readlane((int)c.col, pos);
        rest &= rest - 1ull;
    }
    SegPass sp;
    const unsigned mine = c.seg - pass * SEG_PER_PASS;        // this lane's run as a pair of this pass (>= 6, incl. wrapped: none)
    // pair i -> (half, dword): 0, 1 -> h == 0 dwords 2, 3 (slots 4..7); 2..5 -> h == 1 dwords 0..3 (slots 8..15)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const unsigned pair = h ? (unsigned)(d + 2) : (d >= 2 ? (unsigned)(d - 2) : 99u);
        sp.ones[d] = (mine == pair) ? 0x3C003C00u : 0u;
        sp.colbyte[d] = (h ? colr[d + 2] : colr[d >= 2 ? d - 2 : 0]) * COLBYTES;
    }
    return sp;
}
__device__ __forceinline__ unsigned seg_patch_word(float c)     // (c_hi | c_lo << 16)
{
    const _Float16 hi = (_Float16)c, lo = (_Float16)(c - (float)hi);
    const half2_t hv = {hi, lo};
    return __builtin_bit_cast(unsigned, hv);
}
// the four floats a lane needs for tile `t` of a pass: rows row0 + 32 t + j of the columns owning its dwords
__device__ __forceinline__ void seg_fetch(const SegPass &sp, const i32x4 &rs, int j, int row, float *c4)
{
#pragma unroll
    for (int d = 0; d < 4; ++d) c4[d] = raw_buffer_load_f32(rs, (int)(sp.colbyte[d] + 4u * (unsigned)(row + j)), 0, 0);
}
__device__ __forceinline__ f32x16 seg_mfma(const float *c4, const SegPass &sp, int h, f32x16 acc)
{
    const u32x4 a = {h ? seg_patch_word(c4[0]) : 0u, h ? seg_patch_word(c4[1]) : 0u, seg_patch_word(c4[2]), seg_patch_word(c4[3])};
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, sp.ones), acc, 0, 0, 0);
}
// accumulator init of NT tiles starting at table row `row0` (floats; the layer's offset included): pass 0 from floats requested ahead (`c0`), the
// passes of a wave with more than six runs fetched here
template <int NT>
__device__ __forceinline__ void seg_init(const ColSegs &c, const SegPass &sp0, const float (*c0)[4], const i32x4 &rs, int j, int h, int row0, f32x16 *acc)
{
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.0f;
        acc[t] = seg_mfma(c0[t], sp0, h, z);
    }
    const unsigned npass = (c.nseg + SEG_PER_PASS - 1) / SEG_PER_PASS;
    for (unsigned pass = 1; pass < npass; ++pass) {
        const SegPass sp = seg_pass(c, pass, h);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float c4[4];
            seg_fetch(sp, rs, j, row0 + 32 * t, c4);
            acc[t] = seg_mfma(c4, sp, h, acc[t]);
        }
    }
}
// the Bias interface of dense() for a layer initialised this way
struct BiasSegs {
    const ColSegs *c; const SegPass *sp0; const float (*c0)[4]; i32x4 rs; int j, h, row;      // c0: pass-0 floats of the layer's tiles, row: next tile's table row
    __device__ __forceinline__ void take(f32x16 *acc, int ntiles, int)
    {
        seg_init<2>(*c, *sp0, c0, rs, j, h, row, acc);          // (dense() takes tile pairs)
        c0 += ntiles; row += 32 * ntiles;
    }
    __device__ __forceinline__ void after_barrier(int) {}
    __device__ __forceinline__ void rewind(const float *) {}
};

// FOLD: 0 = point by point; 1 = column-folded dense grid (every tile in one column: wave-uniform column blocks as accumulator init); 2 = column-folded
// SUBSET of the grid (p.gidx, the valid band: the 32 points of a wave lie in whatever columns they lie; accumulator init by runs of columns, ColSegs)
// WPE: model.warping_field.pos_encoding > 0 -- the field's input is [posenc(xyz) | feat] (arch_avatar.py:122,136): eight k-steps, the encoding of the raw point
// evaluated in front of conv1 and again in front of conv5 (32 registers that are not kept alive across conv2 .. conv4); point-by-point launches only.
template <bool WARP, bool COLOUR, int FOLD = 0, bool WPE = false>
__global__ __launch_bounds__(256, 1) void avatar_kernel(const QueryParams p)
{
    static_assert(!FOLD || (WARP && !COLOUR), "column folding: the geometry-only warped query of a dense grid");
    static_assert(!WPE || (WARP && FOLD == 0), "a warping field with a positional encoding runs point by point");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    constexpr int B_FIRST = WARP ? (FOLD ? B_XYZ8 : (WPE ? B_INPE : B_IN67)) : B_PE8;     // first chunk of a pass (the prefetcher wraps to it)

    const long long tk0 = clock64();
    Stream s = stream_init(p, wave, lane, B_FIRST);
    using BiasQ = BiasQueue;
    BiasQ bias;
    bias.hoff = 16u * h;
    const unsigned tiles_per_col = FOLD == 1 ? p.grz / TILE_PTS : 1u;
    float pt_next[3];                    // FOLD == 2: the next tile's point is loaded a tile ahead
    unsigned col_next = 0;
    PointAhead ahead{};
    if constexpr (FOLD == 2) {
        col_next = ahead.start(p, blockIdx.x, gridDim.x, wave, j, pt_next);
        bias.rewind(p.bias + 256);           // conv1's column terms ride its k-step: the queue starts at conv2's block
    } else {
        bias.rewind(FOLD == 1 ? p.colterms + (size_t)(blockIdx.x / tiles_per_col) * 512 : p.bias);
    }
    bias.fetch(h);                       // first block; afterwards every chunk fetches its successor's

    for (int64_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int64_t pidx_raw = tile * TILE_PTS + wave * 32 + j;
        const int64_t pidx = pidx_raw < p.n ? pidx_raw : p.n - 1;
        float pt[3];
        unsigned col = 0;
        if constexpr (FOLD == 2) {
            pt[0] = pt_next[0]; pt[1] = pt_next[1]; pt[2] = pt_next[2]; col = col_next;
            col_next = ahead.advance(p, tile, gridDim.x, wave, j, pt_next);
        } else {
            load_point(p, pidx, pt);
        }

        Frag X[16], Y[16];
        unsigned park = PARK_BASE + wave * PARK_PER_WAVE + lane * 16;
        asm volatile("" : "+v"(park));   // opaque per tile: otherwise every park address is hoisted out of the loop as its own VGPR
        f32x16 pa[2], pb[2];             // deferred accumulators of a layer's last tile pair (ping/pong)
        f32x16 w8[8];                    // the eight accumulators of a wide layer (conv1 of a folded launch, shared.0)
        if constexpr (WARP && FOLD == 1) wide_bias_early(bias, w8);      // conv1's column blocks, tiles 2 .. 7: in flight during the prologue
        const float *bias_head = p.bias;
        asm volatile("" : "+s"(bias_head));   // opaque per tile: keeps bias addresses from being hoisted out of the loop
        float q[3] = {pt[0], pt[1], pt[2]};
        float off[3] = {0.f, 0.f, 0.f};

        if constexpr (WARP) {
            // ---- WarpingField.query (arch_avatar.py:113-140) ----
            Frag S4;                         // k-step 4: raw xyz (pos_encoding 0); k-steps 0..3 are parked in LDS
            if constexpr (FOLD == 0) {
                const Bilinear bl = bilinear_setup<64>(p.feat, p.H, p.W, pt[0] - p.cx, -(pt[1] - p.cy), 32 * h);   // :125-133
#pragma unroll
                for (int k = 0; k < 4; ++k) { Frag f; bilinear_frag(bl, 8 * k, f, s.range); park_store(park, k, f); __builtin_amdgcn_sched_barrier(0); }
            }
            {
                float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (h == 0) { z[0] = pt[0]; z[1] = pt[1]; z[2] = pt[2]; }
                split8(z, S4.hi, S4.lo, s.range);
            }
            // FOLD == 2: the wave's runs of columns; conv1's column terms of the first six runs requested
            ColSegs segs{};
            SegPass sp0{};
            float c1[8][4], c5[8][4];
            const i32x4 crs = bias_rsrc(p.colterms);
            if constexpr (FOLD == 2) {
                segs = col_segments(col, j, h);
                sp0 = seg_pass(segs, 0, h);
#pragma unroll
                for (int t = 0; t < 8; ++t) seg_fetch(sp0, crs, j, 32 * t, c1[t]);
            }
            const ParkIn S{park, &S4};
            Frag WP[4];                      // WPE: posenc(xyz), k-steps 4..7 of the field's input
            if constexpr (WPE) posenc_frags(pt, h, WP, s.range);                                                        // arch_avatar.py:122
            const ParkPeIn SW{park, WP};
            const RegIn RX{X}, RY{Y}, R4{&S4};
            using SP = Pending<ACT_SOFTPLUS, 8>;
            // column-folded launch: conv1's and conv5's blocks come from the column table, everything else from the layer table
            const float *after1 = nullptr, *col5 = nullptr, *after5 = nullptr;
            if constexpr (FOLD != 0) {
                after1 = bias_head + 256; after5 = bias_head + 5 * 256;
                if constexpr (FOLD == 1) {
                    col5 = p.colterms + (size_t)(tile / tiles_per_col) * 512 + 256;
                    const int64_t nt = tile + gridDim.x < p.ntiles ? tile + gridDim.x : p.ntiles - 1;
                    bias_head = p.colterms + (size_t)(nt / tiles_per_col) * 512;                                           // where the next tile starts
                    asm volatile("" : "+s"(after1), "+s"(col5), "+s"(after5), "+s"(bias_head));
                    wide8<1, B_MAIN>(s, R4, w8, bias, h, after1);                                                                   // conv1+bn1 on xyz (+ column term)
                } else {
                    // column terms in the k-step (ColSegs): accumulators start at zero, the bias queue never sees conv1 / conv5
                    col5 = after5;                   // conv4's last pair sends the queue to conv6's block
                    bias_head = after1;              // ... and the last head to conv2's: where the next tile starts
                    asm volatile("" : "+s"(after1), "+s"(col5), "+s"(after5), "+s"(bias_head));
                    seg_init<8>(segs, sp0, c1, crs, j, h, 0, w8);
                    chunk<1, 8, B_MAIN>(s, R4, w8, NoSide{});                 // (the queue already holds conv2's first block: nothing to request here)
                }
                flush<ACT_SOFTPLUS>(w8, X, s.range);
                dense<8, 16, 0, ACT_SOFTPLUS, B_MAIN>(s, RX, RX, Y, bias, h, PendWide<ACT_SOFTPLUS>{w8, X, &s.range}, pb);         // conv2 (+ conv1's pairs 1 .. 3)
            } else {
                if constexpr (WPE) dense<8, layout::INPE_KS, 0, ACT_SOFTPLUS, B_MAIN>(s, SW, SW, X, bias, h, NoSide{}, pa);     // conv1+bn1 on [posenc | feat]
                else dense<8, layout::IN67_KS, 0, ACT_SOFTPLUS, B_MAIN>(s, S, S, X, bias, h, NoSide{}, pa);              // conv1+bn1
                dense<8, 16, 0, ACT_SOFTPLUS, B_MAIN>(s, RX, RX, Y, bias, h, SP{pa, X + 12, &s.range}, pb);                    // conv2
            }
            dense<8, 16, 0, ACT_SOFTPLUS, B_MAIN>(s, RY, RY, X, bias, h, SP{pb, Y + 12, &s.range}, pa);                        // conv3
            if constexpr (FOLD == 2) {
                // conv5's column terms of the first six runs: requested here, in flight during conv4
#pragma unroll
                for (int t = 0; t < 8; ++t) seg_fetch(sp0, crs, j, 256 + 32 * t, c5[t]);
                dense<8, 16, 0, ACT_SOFTPLUS, B_CONV5F>(s, RX, RX, Y, bias, h, SP{pa, X + 12, &s.range}, pb, col5);            // conv4
                BiasSegs b5{&segs, &sp0, c5, crs, j, h, 256};
                dense<8, 16, 1, ACT_SOFTPLUS, B_MAIN>(s, RY, R4, X, b5, h, SP{pb, Y + 12, &s.range}, pa);                      // conv5 on [xyz | x4]
            } else if constexpr (FOLD == 1) {
                dense<8, 16, 0, ACT_SOFTPLUS, B_CONV5F>(s, RX, RX, Y, bias, h, SP{pa, X + 12, &s.range}, pb, col5);            // conv4
                dense<8, 16, 1, ACT_SOFTPLUS, B_MAIN>(s, RY, R4, X, bias, h, SP{pb, Y + 12, &s.range}, pa, after5);            // conv5 on [xyz | x4] (+ column term)
            } else if constexpr (WPE) {
                dense<8, 16, 0, ACT_SOFTPLUS, B_CONV5PE>(s, RX, RX, Y, bias, h, SP{pa, X + 12, &s.range}, pb);                 // conv4
                float pr[3] = {pt[0], pt[1], pt[2]};
                asm volatile("" : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]));      // opaque: evaluated again instead of carried across conv2 .. conv4
                posenc_frags(pr, h, WP, s.range);
                dense<8, 16, layout::INPE_KS, ACT_SOFTPLUS, B_MAIN>(s, RY, SW, X, bias, h, SP{pb, Y + 12, &s.range}, pa);      // conv5 on [x0|x4]
            } else {
                dense<8, 16, 0, ACT_SOFTPLUS, B_CONV5>(s, RX, RX, Y, bias, h, SP{pa, X + 12, &s.range}, pb);                   // conv4
                dense<8, 16, layout::IN67_KS, ACT_SOFTPLUS, B_MAIN>(s, RY, S, X, bias, h, SP{pb, Y + 12, &s.range}, pa);       // conv5 on [x0|x4]
            }
            dense<8, 16, 0, ACT_SOFTPLUS, B_MAIN>(s, RX, RX, Y, bias, h, SP{pa, X + 12, &s.range}, pb);                        // conv6
            dense<8, 16, 0, ACT_SOFTPLUS, B_HEAD16>(s, RY, RY, X, bias, h, SP{pb, Y + 12, &s.range}, pa);                      // conv7
            const f32x16 o = head<16, COLOUR ? B_PE : B_PE8, false>(s, X, bias, bias_head, h, SP{pa, X + 12, &s.range});                                 // out_layer_coord_affine
            // rows 0..2 live in lanes h == 0, regs 0..2: broadcast to the other half
            off[0] = __shfl(o[0], j, 64); off[1] = __shfl(o[1], j, 64); off[2] = __shfl(o[2], j, 64);
            q[0] = pt[0] + off[0]; q[1] = pt[1] + off[1]; q[2] = pt[2] + off[2];                                       // arch_avatar.py:372 (fp32 add)
        }

        // ---- DoubleTNet.forward (arch_avatar.py:65-83) ----
        constexpr bool WIDE0 = !(WARP && COLOUR);      // (the warped colour kernel sits at the register limit: it keeps shared.0 as four chunks of a tile pair)
        if constexpr (WIDE0) wide_bias_early(bias, w8);                                                                         // shared.0's blocks, tiles 2 .. 7
        posenc(q, h, park, s.range);                                                                                            // :70 (parked in LDS)
        const ParkIn P{park, nullptr};
        const RegIn TX{X}, TY{Y};
        using RP = Pending<ACT_RELU, 8>;
        if constexpr (WIDE0) {
            wide8<layout::PE_KS, B_MAIN>(s, P, w8, bias, h);                                                             // shared 0
            flush<ACT_RELU>(w8, X, s.range);
            dense<8, 16, 0, ACT_RELU, B_MAIN>(s, TX, TX, Y, bias, h, PendWide<ACT_RELU>{w8, X, &s.range}, pb);           // shared 1 (+ shared 0's pairs 1 .. 3)
        } else {
            dense<8, layout::PE_KS, 0, ACT_RELU, B_MAIN>(s, P, P, X, bias, h, NoSide{}, pa);                             // shared 0
            dense<8, 16, 0, ACT_RELU, B_MAIN>(s, TX, TX, Y, bias, h, RP{pa, X + 12, &s.range}, pb);                      // shared 1
        }
        dense<8, 16, 0, ACT_RELU, B_MAIN>(s, TY, TY, X, bias, h, RP{pb, Y + 12, &s.range}, pa);
        dense<8, 16, 0, ACT_RELU, B_SHARED4>(s, TX, TX, Y, bias, h, RP{pa, X + 12, &s.range}, pb);
        dense<8, 16, layout::PE_KS, ACT_RELU, B_MAIN>(s, TY, P, X, bias, h, RP{pb, Y + 12, &s.range}, pa);                     // shared 4 on [x|x0]
        dense<8, 16, 0, ACT_RELU, B_MAIN>(s, TX, TX, Y, bias, h, RP{pa, X + 12, &s.range}, pb);                                                       // shared 5
        f32x16 g;
        if constexpr (COLOUR) {
            dense<8, 16, 0, ACT_NONE, B_MAIN>(s, TY, TY, X, bias, h, RP{pb, Y + 12, &s.range}, pa);                                // shared 6: no activation (mlp.py:46,64)
            dense<4, 16, 0, ACT_LEAKY, B_HEAD8>(s, TX, TX, Y, bias, h, Pending<ACT_NONE, 8>{pa, X + 12, &s.range}, pb);           // geo 0
            g = head<8, B_MAIN, false>(s, Y, bias, bias_head, h, Pending<ACT_LEAKY, 4>{pb, Y + 4, &s.range});                      // geo 1: row 0 = occ/sdf, row 1 = sigma
        } else {
            // geometry only: pack.cpp folded shared.6 (linear) into geo.0 -- one 256->128 layer instead of two
            dense<4, 16, 0, ACT_LEAKY, B_HEAD8>(s, TY, TY, X, bias, h, RP{pb, Y + 12, &s.range}, pa);                              // geo 0 o shared 6
            g = head<8, B_FIRST, true>(s, X, bias, bias_head, h, Pending<ACT_LEAKY, 4>{pa, X + 4, &s.range});                      // geo 1
        }

        const bool writer = (h == 0) && (pidx_raw < p.n);
        if (writer) {
            p.out0[pidx_raw] = p.sigmoid_occ ? sigmoid_f(g[0]) : g[0];                                                 // :77-80
            if (WARP && p.out1) { p.out1[pidx_raw * 3 + 0] = off[0]; p.out1[pidx_raw * 3 + 1] = off[1]; p.out1[pidx_raw * 3 + 2] = off[2]; }
        }
        if constexpr (COLOUR) {
            // X (the shared feature) is complete: its last pair was finished inside geo 0's first chunk
            dense<8, 16, 0, ACT_RELU, B_MAIN>(s, TX, TX, Y, bias, h, NoSide{}, pa);                                      // clr 0
            dense<4, 16, 0, ACT_RELU, B_HEAD8>(s, TY, TY, X, bias, h, RP{pa, Y + 12, &s.range}, pb);                           // clr 1
            const f32x16 c = head<8, B_FIRST, true>(s, X, bias, bias_head, h, Pending<ACT_RELU, 4>{pb, X + 4, &s.range});                        // clr 2
            if (writer && p.out2) {
                f32x4 rgba = {sigmoid_f(c[0]), sigmoid_f(c[1]), sigmoid_f(c[2]), __builtin_fmaxf(g[1], 0.0f)};         // :75-76
                *reinterpret_cast<f32x4 *>(p.out2 + pidx_raw * 4) = rgba;
            }
        }
    }
    s.range.report(p);
    if (p.clk && blockIdx.x == 0 && threadIdx.x == 0) { p.clk[0] = tk0; p.clk[1] = clock64(); }     // workgroup 0 is there from the first tile to the last
}

// ------------------------------------------------------------------------------------------
// recon query kernel (arch_recon.py:55-73): [feat(32) | z] -> 512 -> 256 -> 128 -> 1, LeakyReLU(0.02),
// res @ 1,2, sigmoid.  fc0 is produced in two 256-channel halves so that fc1 can consume each half
// while it is the only hidden state alive (see pack.cpp::pack_recon for the matching stream order).
// ------------------------------------------------------------------------------------------
constexpr int B_IN33 = chunk_bytes(layout::IN33_KS, 2);
constexpr int B_WIDE4 = chunk_bytes(4, 8), B_WIDE_IN33 = chunk_bytes(layout::IN33_KS, 8);

__global__ __launch_bounds__(256, 1) void recon_kernel(const QueryParams p)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;

    const long long tk0 = clock64();
    Stream s = stream_init(p, wave, lane, B_IN33);

    // (a launch on a LIST of tiles: what the folded subset launch left over -- see launch_recon)
    const int64_t ntl = p.tile_list ? (int64_t)*p.tile_count : p.ntiles;
    for (int64_t it = blockIdx.x; it < ntl; it += gridDim.x) {
        const int64_t tile = p.tile_list ? (int64_t)p.tile_list[it] : it;
        const int64_t pidx_raw = tile * TILE_PTS + wave * 32 + j;
        const int64_t pidx = pidx_raw < p.n ? pidx_raw : p.n - 1;
        float pt[3];
        load_point(p, pidx, pt);
        const float px = pt[0] - p.cx, py = pt[1] - p.cy, pz = pt[2] - p.cz;   // :62

        Frag I[layout::IN33_KS];
        {
            const Bilinear bl = bilinear_setup<32>(p.feat, p.H, p.W, px, -py, 16 * h);            // :63-68
            bilinear_frag(bl, 0, I[0], s.range);
            bilinear_frag(bl, 8, I[1], s.range);
            float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (h == 0) z[0] = pz;                                                               // :69
            split8(z, I[2].hi, I[2].lo, s.range);
        }
        Frag X[16], Y[16];
        const RegIn RI{I}, RY{Y};
        f32x16 pend[2];
        BiasDirect bias{p.bias};
        asm volatile("" : "+s"(bias.p));   // see avatar_kernel
        f32x16 acc[8];
        // fc0 rows 0..255
        dense<8, layout::IN33_KS, 0, ACT_LEAKY, B_WIDE4>(s, RI, RI, X, bias, h, NoSide{}, pend);
        flush<ACT_LEAKY>(pend, X + 12, s.range);
        // fc1 partial over x[0..255]: 8 tiles live, 4 k-steps per chunk
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = bias_tile(bias.p + 32 * t, h);
        bias.p += 256;
        chunk<4, 8, B_WIDE4>(s, RegIn{X}, acc, NoSide{});
        chunk<4, 8, B_WIDE4>(s, RegIn{X + 4}, acc, NoSide{});
        chunk<4, 8, B_WIDE4>(s, RegIn{X + 8}, acc, NoSide{});
        chunk<4, 8, B_IN33>(s, RegIn{X + 12}, acc, NoSide{});
        // fc0 rows 256..511
        dense<8, layout::IN33_KS, 0, ACT_LEAKY, B_WIDE4>(s, RI, RI, X, bias, h, NoSide{}, pend);
        flush<ACT_LEAKY>(pend, X + 12, s.range);
        bias.p += 256;   // (zero bias block of the second fc1 pack call)
        chunk<4, 8, B_WIDE4>(s, RegIn{X}, acc, NoSide{});
        chunk<4, 8, B_WIDE4>(s, RegIn{X + 4}, acc, NoSide{});
        chunk<4, 8, B_WIDE4>(s, RegIn{X + 8}, acc, NoSide{});
        chunk<4, 8, B_WIDE_IN33>(s, RegIn{X + 12}, acc, NoSide{});
        chunk<layout::IN33_KS, 8, first_chunk_bytes(16, layout::IN33_KS)>(s, RI, acc, NoSide{});
#pragma unroll
        for (int t = 0; t < 8; t += 2) flush<ACT_LEAKY>(acc + t, Y + 2 * t, s.range);
        // fc2 on [x(256) | in(33)] -> 128
        dense<4, 16, layout::IN33_KS, ACT_LEAKY, B_HEAD8>(s, RY, RI, X, bias, h, NoSide{}, pend);
        const f32x16 o = head<8, B_IN33, true>(s, X, bias, p.bias, h, Pending<ACT_LEAKY, 4>{pend, X + 4, &s.range});
        if (h == 0 && pidx_raw < p.n) p.out0[pidx_raw] = sigmoid_f(o[0]);                         // last_op sigmoid
    }
    s.range.report(p);
    if (p.clk && blockIdx.x == 0 && threadIdx.x == 0) { p.clk[0] = tk0; p.clk[1] = clock64(); }
}

// ---- column folding of the recon query on a grid ------------------------------------------------------------------------------------------
// The decoder's input is [img_feat(32) at (x, y) | z] (arch_recon.py:63-70) and enters fc0, fc1 and fc2 (res_layers, mlp.py:61): on a grid the 32
// feature channels are one vector per (x, y) column, so what the three layers do with them -- W[:, feat] f(x, y) + b, 896 = 512 + 256 + 128 rows --
// is computed once per column in fp32 (recon_column_terms_kernel) and enters as the accumulator init; only the z column stays a k-step.
// 1,068 instead of 1,236 MFMAs per 32 points, no per-point gather, and fc0 -- three k-steps for 512 outputs, epilogue-bound in the point-by-point
// kernel -- becomes two wide chunks whose epilogues hide in fc1's chunks.  Same algebra as the reference, other rounding: ~1e-6 from recon_kernel<0>.
constexpr int RCOL = 896;                         // floats per column: [fc0 rows 0..511 | fc1 | fc2], biases included
__global__ __launch_bounds__(256) void recon_column_terms_kernel(const float *__restrict__ feat, int H, int W, const float *__restrict__ gx,
                                                                 const float *__restrict__ gy, int nx, int ny, float cx, float cy,
                                                                 const float *__restrict__ colw, float *__restrict__ out,
                                                                 const uint8_t *__restrict__ colflag)
{
    // colw: [896][32] fp32 weights of the feature columns, then [896] biases;  colflag (subset launches, else null): the columns the subset touches
    constexpr int CPB = 8;                        // columns per trip
    __shared__ __attribute__((aligned(16))) float f[CPB][32];
    const int o = threadIdx.x;
    float w[4][32], b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = o + 256 * r;
        b[r] = row < RCOL ? colw[RCOL * 32 + row] : 0.0f;
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
            const f32x4 a = row < RCOL ? *reinterpret_cast<const f32x4 *>(colw + (size_t)row * 32 + c) : f32x4{0, 0, 0, 0};
            w[r][c] = a[0]; w[r][c + 1] = a[1]; w[r][c + 2] = a[2]; w[r][c + 3] = a[3];
        }
    }
    const int ncol = nx * ny;
    for (int c0 = blockIdx.x * CPB; c0 < ncol; c0 += gridDim.x * CPB) {
        if (colflag && *reinterpret_cast<const unsigned long long *>(colflag + c0) == 0ull) continue;      // (CPB == 8 flags, table padded: workgroup-uniform)
        {   // the bilinear sample of arch_recon.py:63-68 as recon_kernel<0> takes it: thread = (column, channel)
            const int q = threadIdx.x >> 5, ch = threadIdx.x & 31, col = min(c0 + q, ncol - 1);
            const Bilinear bl = bilinear_setup<32>(feat, H, W, gx[col / ny] - cx, -(gy[col % ny] - cy), 0);
            f[q][ch] = bl.p00[ch] * bl.w00 + bl.p01[ch] * bl.w01 + bl.p10[ch] * bl.w10 + bl.p11[ch] * bl.w11;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CPB; ++q) {
            if (c0 + q >= ncol) break;
            float a[4] = {b[0], b[1], b[2], b[3]};
#pragma unroll
            for (int c = 0; c < 32; c += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(&f[q][c]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] = __builtin_fmaf(w[r][c + i], v[i], a[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (o + 256 * r < RCOL) out[(size_t)(c0 + q) * RCOL + o + 256 * r] = a[r];
        }
        __syncthreads();
    }
}

// accumulator inits out of registers, in consumption order (a BiasQueue whose blocks were all loaded ahead)
struct BiasRegs {
    const f32x16 *cur;
    __device__ __forceinline__ void take(f32x16 *acc, int ntiles, int)
    {
        acc[0] = cur[0];
        if (ntiles > 1) acc[1] = cur[1];
        cur += ntiles;
    }
    __device__ __forceinline__ void after_barrier(int) {}
    __device__ __forceinline__ void rewind(const float *) {}
};

template <class F>
struct BiasRegsThen {                // BiasRegs whose after-the-barrier slot runs `then` (requests that must not sit in front of the chunk's vmcnt(0))
    const f32x16 *cur;
    F then;
    __device__ __forceinline__ void take(f32x16 *acc, int ntiles, int) { acc[0] = cur[0]; if (ntiles > 1) acc[1] = cur[1]; cur += ntiles; }
    __device__ __forceinline__ void after_barrier(int) { then(); }
    __device__ __forceinline__ void rewind(const float *) {}
};

constexpr int B_Z8 = chunk_bytes(1, 8), B_ZZ8 = chunk_bytes(2, 8), B_FC2F = first_chunk_bytes(16, 1);

// FOLD == 2 (a SUBSET of the grid, p.gidx: the valid band): the 32 points of a wave lie in whatever (x, y) columns the band gives them -- along z a band's
// runs are long, so in one or two.  The wave's points are cut into runs of equal adjacent columns (col_segments, as in avatar_kernel<.., 2>).  Runs 0 and
// 1 ride the z k-step exactly like the single column of a dense launch: the K slots of a k-step are split between the lane halves (h == 0: slots 0..7,
// h == 1: slots 8..15), so the patched dword of the lanes h == 0 (slots 2, 3) carries run 0's column term of the lane's row and that of the lanes h == 1
// (slots 10, 11) run 1's, each against 1.0 in the B operand of exactly the points of its run -- still ONE float per lane and 32-row tile, one 256-byte
// wave-instruction, whatever the wave's points do.  A tile with a wave of MORE than two runs (band edges, tangential cuts: 0.2 - 1.4 % of the waves of a
// band) is not evaluated here at all: band_prepass_kernel flags it (p.tile_skip) and the point-by-point recon_kernel runs on the list of flagged tiles
// afterwards.  (A first cut added the terms of runs 2.. to the zeroed accumulators in place, six runs per MFMA: the code of that rare path cost the kernel
// 128 registers and 1,340 v_accvgpr moves per tile on the common one -- 12 % more cycles per tile than the dense launch, profiles/r05_band_split.md.)
constexpr unsigned RCOL_BYTES = RCOL * 4u;
constexpr int RECON_MAX_RUNS = 2;
struct ReconRuns {
    unsigned voff;               // byte offset of (this lane's run column, row j) in the table
    bool one;                    // this lane's point belongs to the run its lane half carries
};
__device__ __forceinline__ ReconRuns recon_runs(unsigned col, int j, int h)
{
    const ColSegs c = col_segments(col, j, h);
    const unsigned long long r1 = c.first & (c.first - 1ull);
    const int p0 = __builtin_ctzll(c.first | (1ull << 63)), p1 = r1 ? __builtin_ctzll(r1) : p0;
    const unsigned col0 = (unsigned)__builtin_amdgcn_readlane((int)col, p0), col1 = (unsigned)__builtin_amdgcn_readlane((int)col, p1);
    ReconRuns r;
    r.voff = (h ? col1 : col0) * RCOL_BYTES + 4u * (unsigned)j;
    r.one = c.seg == (unsigned)h;
    return r;
}

// One workgroup = one 128-point tile of a subset launch (the lanes beyond n repeat point n - 1, as in the kernels): marks the (x, y) columns the subset
// touches (colflag: the column pass then skips the others -- a band covers about a third of them), and flags / lists the tiles in which some wavefront's
// 32 points fall into more than max_runs runs of equal adjacent columns.
__global__ __launch_bounds__(128) void band_prepass_kernel(const int32_t *__restrict__ gidx, int64_t n, unsigned grz, int max_runs,
                                                           uint8_t *__restrict__ colflag, int32_t *__restrict__ tile_skip,
                                                           int32_t *__restrict__ slow_list, int32_t *__restrict__ slow_count)
{
    const int64_t tile = blockIdx.x, i = tile * TILE_PTS + threadIdx.x;
    const unsigned col = (unsigned)gidx[i < n ? i : n - 1] / grz;
    const int j = threadIdx.x & 31;
    const unsigned prev = (unsigned)__shfl_up((int)col, 1, 32);
    const bool first = j == 0 || prev != col;
    if (first) colflag[col] = 1;
    const unsigned long long b = __builtin_amdgcn_ballot_w64(first);
    const int slow = __syncthreads_or(__builtin_popcountll(b & 0xffffffffull) > max_runs || __builtin_popcountll(b >> 32) > max_runs);
    if (threadIdx.x == 0) {
        tile_skip[tile] = slow;
        if (slow) slow_list[atomicAdd(slow_count, 1)] = (int32_t)tile;
    }
}

// FOLD: 1 = dense grid whose tiles lie in one (x, y) column each; 2 = subset of the grid by flat indices (p.gidx), the column terms by runs of columns
#if !AVC_CHECK_RANGE
template <int FOLD>
__global__ __launch_bounds__(256, 1) void recon_fold_kernel(const QueryParams p)
{
    static_assert(FOLD == 1 || FOLD == 2, "recon_fold_kernel: dense grid or subset");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const long long tk0 = clock64();
    Stream s = stream_init(p, wave, lane, B_Z8);
    const unsigned tiles_per_col = FOLD == 1 ? p.grz / TILE_PTS : 1u;
    // How a column term reaches the accumulators: every layer that takes one also has the z k-step, whose input fragment uses ONE of its 16 K slots.
    // The term c of (row, column) is split into fp16 halves (c_hi, c_lo) and put into K slots 2 and 3 of the `hi` A fragment of that k-step -- the second
    // dword of the lane's four -- against 1.0 in slots 2 and 3 of the B operand: the (hi, hi) MFMA of the z k-step adds c_hi + c_lo = c (to 2^-22) for free.
    // Accumulators start at zero, and what a lane fetches per 32-row tile is ONE float (row j of the tile: 256 bytes per wave-instruction) instead of the
    // sixteen of an accumulator-init block (round-3 first cut: 4 x dwordx4 per tile, 1 KiB per instruction through the texture path whatever the
    // duplication -- 2 k cycles per eight tiles, profiles/r03_recon_time_split.md).
    float pt_next[3];
    unsigned col_next = 0;
    auto col_rsrc = [&](int64_t t) { return bias_rsrc(FOLD == 1 ? p.colterms + (size_t)(t / tiles_per_col) * RCOL : p.colterms); };
    auto col_voff = [&](unsigned col) { if constexpr (FOLD == 1) return 4u * (unsigned)j; else return recon_runs(col, j, h).voff; };
    auto fetch8 = [&](float *dst, const i32x4 &rs, unsigned voff, int first_row, int ntiles) {          // row j of `ntiles` consecutive tiles
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < ntiles) dst[t] = raw_buffer_load_f32(rs, (int)voff + 4 * (first_row + 32 * t), 0, 0);
    };
    auto split8p = [&](const float *c, unsigned *pd, int ntiles) {                                        // -> packed (c_hi | c_lo << 16)
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < ntiles) {
                const _Float16 hi = (_Float16)c[t], lo = (_Float16)(c[t] - (float)hi);
                const half2_t hv = {hi, lo};
                pd[t] = __builtin_bit_cast(unsigned, hv);
            }
    };
    float cn[8];                          // fc0 rows 0..255 of the NEXT tile, requested behind the fc3 head of the current one
    PointAhead ahead{};
    col_next = ahead.start(p, blockIdx.x, gridDim.x, wave, j, pt_next);
    fetch8(cn, col_rsrc(blockIdx.x), col_voff(col_next), 0, 8);
    int skip_next = FOLD == 2 ? p.tile_skip[blockIdx.x] : 0;

    for (int64_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int64_t pidx_raw = tile * TILE_PTS + wave * 32 + j;
        float pt[3] = {pt_next[0], pt_next[1], pt_next[2]};
        const unsigned col = col_next;
        const int64_t tile_n = tile + gridDim.x < p.ntiles ? tile + gridDim.x : tile;           // the tile this workgroup runs next (itself: the last one)
        col_next = ahead.advance(p, tile, gridDim.x, wave, j, pt_next);
        if constexpr (FOLD == 2) {
            const int skip = skip_next;                                                          // (requested a tile ago: a scalar load, never waited for)
            skip_next = p.tile_skip[tile_n];
            if (skip) {                                                                          // left to the point-by-point kernel (workgroup-uniform)
                fetch8(cn, col_rsrc(tile_n), col_voff(col_next), 0, 8);                          // what the skipped body would have requested for the next tile
                continue;
            }
        }
        const float *cb = FOLD == 1 ? p.colterms + (size_t)(tile / tiles_per_col) * RCOL : p.colterms;
        asm volatile("" : "+s"(cb));
        const i32x4 crs = bias_rsrc(cb);
        ReconRuns runs{};
        if constexpr (FOLD == 2) runs = recon_runs(col, j, h);
        const unsigned voff = FOLD == 1 ? 4u * (unsigned)j : runs.voff;
        unsigned pd0[8], pdz[16], pd2[4];  // patches: fc0 rows 0..255 | [fc0 rows 256..511, fc1] | fc2
        float cz[16], c2f[4];
        split8p(cn, pd0, 8);
        Frag Z;                            // the z k-step's B fragment: z in slot 0, 1.0 in the slots of the run this lane half carries (dense: slots 2, 3 of h == 0)
        {
            float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (h == 0) z[0] = pt[2] - p.cz;                                                      // arch_recon.py:62,69
            if (FOLD == 1 ? h == 0 : runs.one) { z[2] = 1.0f; z[3] = 1.0f; }
            split8(z, Z.hi, Z.lo, s.range);
        }
        const SameIn RZ{&Z};
        auto patched = [](half8 a, unsigned d) { u32x4 v = __builtin_bit_cast(u32x4, a); v[1] = d; return __builtin_bit_cast(half8, v); };
        Frag X[16], Y[16];
        using LP = Pending<ACT_LEAKY, 16>;
        f32x16 acc16[16];                  // [0..7]: the 256-row half of fc0 in flight; [8..15]: fc1's accumulators
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc16[t][r] = 0.0f;
        // fc0 rows 0..255: z column + column term, all eight tiles in one chunk
        chunk<1, 8, B_WIDE4>(s, RZ, acc16, [&](auto qc, auto rc) {
            if constexpr (decltype(qc)::value == 0 && decltype(rc)::value == 5) { fetch8(cz, crs, voff, 256, 8); fetch8(cz + 8, crs, voff, 512, 8); }
        }, [&](auto qc, auto tc, half8 a) { return patched(a, pd0[2 * decltype(qc)::value + decltype(tc)::value]); });
        flush<ACT_LEAKY>(acc16, X, s.range);
        // fc1 over x[0..255]: chunk c consumes the fragments of fc0's pair c and finishes pair c + 1 in its shadow
        chunk<4, 8, B_WIDE4>(s, RegIn{X}, acc16 + 8, LP{acc16 + 2, X + 4, &s.range});
        chunk<4, 8, B_WIDE4>(s, RegIn{X + 4}, acc16 + 8, LP{acc16 + 4, X + 8, &s.range});
        chunk<4, 8, B_WIDE4>(s, RegIn{X + 8}, acc16 + 8, LP{acc16 + 6, X + 12, &s.range});
        chunk<4, 8, B_ZZ8>(s, RegIn{X + 12}, acc16 + 8, NoSide{});
        split8p(cz, pdz, 8); split8p(cz + 8, pdz + 8, 8);
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc16[t][r] = 0.0f;
        // [fc0 rows 256..511 | fc1's z column]: two k-steps on the same B operand, two sets of accumulators, each with its column term
        chunk<2, 8, B_WIDE4, 8>(s, RZ, acc16, [&](auto qc, auto rc) {
            if constexpr (decltype(qc)::value == 0 && decltype(rc)::value == 5) fetch8(c2f, crs, voff, 768, 4);
        }, [&](auto qc, auto tc, half8 a) { return patched(a, pdz[2 * decltype(qc)::value + decltype(tc)::value]); });
        flush<ACT_LEAKY>(acc16, X, s.range);
        chunk<4, 8, B_WIDE4>(s, RegIn{X}, acc16 + 8, LP{acc16 + 2, X + 4, &s.range});
        chunk<4, 8, B_WIDE4>(s, RegIn{X + 4}, acc16 + 8, LP{acc16 + 4, X + 8, &s.range});
        chunk<4, 8, B_WIDE4>(s, RegIn{X + 8}, acc16 + 8, LP{acc16 + 6, X + 12, &s.range});
        f32x16 b3;                         // fc3's bias tile (layer table), requested a chunk ahead
        chunk<4, 8, B_FC2F>(s, RegIn{X + 12}, acc16 + 8, [&](auto qc, auto rc) {
            if constexpr (decltype(qc)::value == 0 && decltype(rc)::value == 5)
                b3 = bias_tile_buf(bias_rsrc(p.bias), 16u * h, 4 * (5 * 256 + 128));              // fc3's block of the layer table (pack.cpp: pack_recon)
        });
        split8p(c2f, pd2, 4);
        // fc1's eight tiles -> Y: pair 0 now, pairs 1..3 inside fc2
        flush<ACT_LEAKY>(acc16 + 8, Y, s.range);
        f32x16 zero2[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) zero2[t][r] = 0.0f;
        BiasRegs bias{zero2};
        f32x16 pend[2];
        dense<4, 16, 1, ACT_LEAKY, B_HEAD8>(s, RegIn{Y}, RZ, X, bias, h, PendWide<ACT_LEAKY>{acc16 + 8, Y, &s.range}, pend, nullptr,
                                            [&](auto pc, auto tc, half8 a) { return patched(a, pd2[2 * decltype(pc)::value + decltype(tc)::value]); });      // fc2 on [x(256) | z]
        // the next tile's first column terms: requested right after the head's barrier, in flight during the fc3 head, the output store and the loop's turn-around
        auto prefetch_next = [&]() { fetch8(cn, col_rsrc(tile_n), col_voff(col_next), 0, 8); };
        BiasRegsThen<decltype(prefetch_next)> hbias{&b3, prefetch_next};
        const f32x16 o = head<8, B_Z8, true>(s, X, hbias, p.bias, h, Pending<ACT_LEAKY, 4>{pend, X + 4, &s.range});
        if (h == 0 && pidx_raw < p.n) p.out0[pidx_raw] = sigmoid_f(o[0]);                                                          // last_op sigmoid
    }
    s.range.report(p);
    if (p.clk && blockIdx.x == 0 && threadIdx.x == 0) { p.clk[0] = tk0; p.clk[1] = clock64(); }
}
#endif

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static void timing_begin(avc_ctx *ctx, int which, hipStream_t s, hipEvent_t &e0, hipEvent_t &e1, QueryParams &p)
{
    e0 = e1 = nullptr;
    if (!ctx->timing.enabled) return;
    auto &t = ctx->timing;
    if (!t.clk_dev[which] && hipMalloc((void **)&t.clk_dev[which], 2 * sizeof(long long) * Timing::CLK_SLOTS) != hipSuccess) t.clk_dev[which] = nullptr;
    if (t.clk_dev[which]) p.clk = t.clk_dev[which] + 2 * (t.clk_count[which]++ % Timing::CLK_SLOTS);
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
}
static void timing_end(avc_ctx *ctx, int which, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    if (!e0) return;
    hipEventRecord(e1, s);
    ctx->timing.pending[which].push_back({e0, e1});
}

template <typename K>
static int set_lds(K kernel)
{
    AVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    return AVC_OK;
}

// every kernel of this flavour gets its dynamic-LDS attribute once, BEFORE a launch function starts its timing events: a failure then returns without
// leaking them
static int set_all_lds()
{
    static bool done = false;
    if (done) return AVC_OK;
    if (int rc = set_lds(avatar_kernel<false, true, 0>)) return rc;
    if (int rc = set_lds(avatar_kernel<false, false, 0>)) return rc;
    if (int rc = set_lds(avatar_kernel<true, true, 0>)) return rc;
    if (int rc = set_lds(avatar_kernel<true, false, 0>)) return rc;
    if (int rc = set_lds(avatar_kernel<true, true, 0, true>)) return rc;
    if (int rc = set_lds(avatar_kernel<true, false, 0, true>)) return rc;
#if !AVC_CHECK_RANGE
    if (int rc = set_lds(avatar_kernel<true, false, 1>)) return rc;
    if (int rc = set_lds(avatar_kernel<true, false, 2>)) return rc;
    if (int rc = set_lds(recon_fold_kernel<1>)) return rc;
    if (int rc = set_lds(recon_fold_kernel<2>)) return rc;
#endif
    if (int rc = set_lds(recon_kernel)) return rc;
    done = true;
    return AVC_OK;
}

static unsigned bytes_until(const PackedNet &net, size_t nchunks)
{
    unsigned b = 0;
    for (size_t i = 0; i < nchunks && i < net.chunks.size(); ++i) b += net.chunks[i].bytes;
    return b;
}

static int fill_points(QueryParams &p, avc_ctx *ctx, const float *pts, const GridDesc *grid, int64_t n, const char *who)
{
    p.pts = pts; p.n = n;
    if (grid) {
        AVC_REQUIRE(grid->x && grid->y && grid->z && grid->res[0] > 0 && grid->res[1] > 0 && grid->res[2] > 0, AVC_ERR_ARG, "%s: bad grid descriptor", who);
        const int64_t total = (int64_t)grid->res[0] * grid->res[1] * grid->res[2];
        AVC_REQUIRE(total < ((int64_t)1 << 31) && (grid->idx ? n <= total : n == total), AVC_ERR_ARG,
                    "%s: grid of %d x %d x %d points does not match n = %lld (or exceeds 2^31)", who, grid->res[0], grid->res[1], grid->res[2], (long long)n);
        p.gidx = grid->idx;
        p.pts = nullptr; p.gx = grid->x; p.gy = grid->y; p.gz = grid->z; p.gry = (unsigned)grid->res[1]; p.grz = (unsigned)grid->res[2];
    }
#if AVC_CHECK_RANGE
    if (!ctx->range_flag_dev) AVC_HIP(hipMalloc((void **)&ctx->range_flag_dev, sizeof(unsigned)));
    p.range_flag = ctx->range_flag_dev;
#endif
    return AVC_OK;
}

// AVC_CHECK_RANGE builds: clear the flag before the launch, read it back (synchronously) after
static int range_begin(avc_ctx *ctx, hipStream_t s)
{
#if AVC_CHECK_RANGE
    AVC_HIP(hipMemsetAsync(ctx->range_flag_dev, 0, sizeof(unsigned), s));
#endif
    return AVC_OK;
}
static int range_end(avc_ctx *ctx, hipStream_t s, const char *who)
{
#if AVC_CHECK_RANGE
    unsigned flag = 0;
    AVC_HIP(hipMemcpyAsync(&flag, ctx->range_flag_dev, sizeof flag, hipMemcpyDeviceToHost, s));
    AVC_HIP(hipStreamSynchronize(s));
    AVC_REQUIRE(flag == 0, AVC_ERR_RANGE, "%s: a feature or activation exceeded 65504 in magnitude -- outside the range of the split-fp16 "
                "arithmetic (include/avcap.h, 'numeric range'); the outputs of this call are not valid", who);
#endif
    return AVC_OK;
}

int launch_avatar(avc_ctx *ctx, const float *pts, const GridDesc *grid, int64_t n, const float center[3], int occ_sigmoid,
                  float *occ, float *offset, float *rgba, bool template_only, hipStream_t s)
{
    const bool colour = rgba != nullptr;
    // a dense grid whose last axis holds a multiple of 128 points: every tile lies in one (x, y) column (see column_terms_kernel); a SUBSET of a grid
    // (grid->idx: the valid band) is folded whatever its shape, the column blocks then being gathered per lane
#if AVC_CHECK_RANGE
    const bool can_fold = false;               // the range-checked flavour (a debugging aid) keeps to the point-by-point kernels: half the build time
#else
    const bool can_fold = grid && !template_only && !colour && ctx->warp_tmpl_fold.ready && ctx->opt.column_fold && ctx->warp_pe == 0;
#endif
    const bool wpe = !template_only && ctx->warp_pe > 0;      // a positional encoding in front of the warping field: the eight-k-step input (avatar_kernel<.., WPE>)
    const int fold = !can_fold ? 0 : (grid->idx ? 2 : (grid->res[2] % TILE_PTS == 0 ? 1 : 0));
    PackedNet &net = template_only ? (colour ? ctx->tmpl_only_clr : ctx->tmpl_only) : (colour ? ctx->warp_tmpl_clr : (fold ? ctx->warp_tmpl_fold : ctx->warp_tmpl));
    AVC_REQUIRE(!colour || net.ready || !(template_only ? ctx->tmpl_only : ctx->warp_tmpl).ready, AVC_ERR_STATE,
                "avatar query: rgba requested but clr_mlp weights were not packed");
    AVC_REQUIRE(net.ready, AVC_ERR_STATE, "avatar query: weights not packed (call avc_pack_warp_weights and avc_pack_template_weights)");
    AVC_REQUIRE(template_only || ctx->pose_feat_hwc, AVC_ERR_STATE, "avatar query: pose feature map not set (WarpingField.precompute_conv)");
    if (n == 0) return AVC_OK;
    QueryParams p{};
    int rc = fill_points(p, ctx, pts, grid, n, "avatar query");
    if (rc) return rc;
    p.feat = ctx->pose_feat_hwc; p.H = ctx->pose_H; p.W = ctx->pose_W;
    p.cx = center ? center[0] : 0.f; p.cy = center ? center[1] : 0.f; p.cz = center ? center[2] : 0.f;
    p.wstream = (const char *)net.d_stream; p.bias = net.d_bias;
    p.sp_unscale = net.sp_unscale;
#if !AVC_LAYER_SCALE
    AVC_REQUIRE(net.sp_unscale == 1.0f, AVC_ERR_STATE, "avatar query: internal: a scaled weight stream on the unscaled kernels");
#endif
    p.out0 = occ; p.out1 = offset; p.out2 = rgba; p.sigmoid_occ = occ_sigmoid;
    p.ntiles = (n + TILE_PTS - 1) / TILE_PTS;
    p.stream_bytes = bytes_until(net, net.chunks.size());
    const int grid_dim = (int)std::min<int64_t>(p.ntiles, ctx->opt.mlp_blocks > 0 ? ctx->opt.mlp_blocks : ctx->num_cus);   // persistent workgroups
    rc = range_begin(ctx, s);
    if (rc) return rc;
    if (fold) {
        const size_t ncol = (size_t)grid->res[0] * grid->res[1], bytes = (ncol * 512 + 64) * sizeof(float);
        if (ctx->col_scratch_bytes < bytes) {
            if (ctx->col_scratch) { AVC_HIP(hipDeviceSynchronize()); AVC_HIP(hipFree(ctx->col_scratch)); }        // an earlier launch (any stream) may still read it
            ctx->col_scratch = nullptr; ctx->col_scratch_bytes = 0;
            AVC_HIP(hipMalloc(&ctx->col_scratch, bytes));
            ctx->col_scratch_bytes = bytes;
        }
        p.colterms = static_cast<const float *>(ctx->col_scratch);
    }
    uint8_t *colflag = nullptr;
#if !AVC_CHECK_RANGE
    size_t band_head = 0;
    if (fold == 2) {
        // the columns the subset touches (band_prepass_kernel; no tile is left out here: this kernel takes any number of runs per wave)
        const size_t ncol = (size_t)grid->res[0] * grid->res[1], ncol_pad = (ncol + 15) / 16 * 16;
        band_head = 16 + ncol_pad;
        const size_t bytes = band_head + (size_t)p.ntiles * 8;
        if (ctx->band_scratch_bytes < bytes) {
            if (ctx->band_scratch) { AVC_HIP(hipDeviceSynchronize()); AVC_HIP(hipFree(ctx->band_scratch)); }
            ctx->band_scratch = nullptr; ctx->band_scratch_bytes = 0;
            AVC_HIP(hipMalloc(&ctx->band_scratch, bytes));
            ctx->band_scratch_bytes = bytes;
        }
        colflag = reinterpret_cast<uint8_t *>(static_cast<char *>(ctx->band_scratch) + 16);
    }
#endif
    if (int rc0 = set_all_lds()) return rc0;
    hipEvent_t e0, e1;
    timing_begin(ctx, 0, s, e0, e1, p);        // (a folded launch is timed with its column pass, a subset launch also with its prepass)
#define LAUNCH(W_, C_, F_, ...)                                                                         \
    do {                                                                                                \
        rc = set_lds(avatar_kernel<W_, C_, F_, ##__VA_ARGS__>);                                         \
        if (rc) return rc;                                                                              \
        hipLaunchKernelGGL((avatar_kernel<W_, C_, F_, ##__VA_ARGS__>), dim3(grid_dim), dim3(256), LDS_BYTES, s, p);   \
    } while (0)
    if (template_only) { if (colour) LAUNCH(false, true, 0); else LAUNCH(false, false, 0); }
    else if (wpe) { if (colour) LAUNCH(true, true, 0, true); else LAUNCH(true, false, 0, true); }
    else if (colour) LAUNCH(true, true, 0);
    else if (fold) {
        const int ncol = grid->res[0] * grid->res[1];
#if !AVC_CHECK_RANGE
        if (fold == 2) {
            char *base = static_cast<char *>(ctx->band_scratch);
            int32_t *tile_flag = reinterpret_cast<int32_t *>(base + band_head);
            AVC_HIP(hipMemsetAsync(ctx->band_scratch, 0, band_head, s));
            hipLaunchKernelGGL(band_prepass_kernel, dim3((unsigned)p.ntiles), dim3(TILE_PTS), 0, s, p.gidx, p.n, p.grz, 64, colflag, tile_flag,
                               tile_flag + p.ntiles, reinterpret_cast<int32_t *>(base));
        }
#endif
        hipLaunchKernelGGL(column_terms_kernel, dim3(std::min((ncol + 3) / 4, ctx->num_cus * 4)), dim3(256), 0, s, p.feat, p.H, p.W, p.gx, p.gy,
                           (int)grid->res[0], (int)grid->res[1], p.cx, p.cy, (const float *)net.d_colw, p.bias, p.bias + 4 * 256,
                           static_cast<float *>(ctx->col_scratch), colflag);
#if !AVC_CHECK_RANGE
        if (fold == 2) LAUNCH(true, false, 2); else LAUNCH(true, false, 1);
#endif
    } else LAUNCH(true, false, 0);
#undef LAUNCH
    AVC_HIP(hipGetLastError());
    timing_end(ctx, 0, s, e0, e1);
    return range_end(ctx, s, "avatar query");
}

int launch_recon(avc_ctx *ctx, const float *pts, const GridDesc *grid, int64_t n, const float center[3], float *out, hipStream_t s)
{
    // a dense grid launch is column-folded (recon_fold_kernel<1>) when its tiles lie in one (x, y) column each, a SUBSET of a grid (grid->idx: the valid
    // band) by runs of columns (recon_fold_kernel<2>); point lists run recon_kernel
#if AVC_CHECK_RANGE
    const int fold = 0;                    // the range-checked flavour keeps to the point-by-point kernel
#else
    const int fold = !(grid && ctx->recon_fold.ready && ctx->opt.column_fold) ? 0 : (grid->idx ? 2 : (grid->res[2] % TILE_PTS == 0 ? 1 : 0));
#endif
    PackedNet &net = fold ? ctx->recon_fold : ctx->recon;
    AVC_REQUIRE(net.ready, AVC_ERR_STATE, "recon query: weights not packed (call avc_pack_recon_weights)");
    AVC_REQUIRE(ctx->img_feat_hwc, AVC_ERR_STATE, "recon query: image feature map not set");
    if (n == 0) return AVC_OK;
    QueryParams p{};
    int rc = fill_points(p, ctx, pts, grid, n, "recon query");
    if (rc) return rc;
    p.feat = ctx->img_feat_hwc; p.H = ctx->img_H; p.W = ctx->img_W;
    p.cx = center[0]; p.cy = center[1]; p.cz = center[2];
    p.wstream = (const char *)net.d_stream; p.bias = net.d_bias;
    p.stream_bytes = bytes_until(net, net.chunks.size());
    p.out0 = out;
    p.ntiles = (n + TILE_PTS - 1) / TILE_PTS;
    const int grid_dim = (int)std::min<int64_t>(p.ntiles, ctx->opt.mlp_blocks > 0 ? ctx->opt.mlp_blocks : ctx->num_cus);
    rc = range_begin(ctx, s);
    if (rc) return rc;
    if (fold) {
        const size_t ncol = (size_t)grid->res[0] * grid->res[1], bytes = (ncol * RCOL + 64) * sizeof(float);
        AVC_REQUIRE(ncol * RCOL * sizeof(float) < ((size_t)1 << 32) - 4096, AVC_ERR_ARG, "recon query: %zu columns exceed the 4 GiB column table of a folded launch", ncol);
        if (ctx->rcol_scratch_bytes < bytes) {
            if (ctx->rcol_scratch) { AVC_HIP(hipDeviceSynchronize()); AVC_HIP(hipFree(ctx->rcol_scratch)); }      // an earlier launch (any stream) may still read it
            ctx->rcol_scratch = nullptr; ctx->rcol_scratch_bytes = 0;
            AVC_HIP(hipMalloc(&ctx->rcol_scratch, bytes));
            ctx->rcol_scratch_bytes = bytes;
        }
        p.colterms = static_cast<const float *>(ctx->rcol_scratch);
    }
    // a subset launch: flags of the columns the subset touches, of the tiles the folded kernel leaves out, and their list (band_prepass_kernel)
    uint8_t *colflag = nullptr;
    int32_t *tile_skip = nullptr, *slow_list = nullptr, *slow_count = nullptr;
    size_t head_bytes = 0;
    if (fold == 2) {
        const size_t ncol = (size_t)grid->res[0] * grid->res[1], ncol_pad = (ncol + 15) / 16 * 16;
        head_bytes = 16 + ncol_pad;
        const size_t bytes = head_bytes + (size_t)p.ntiles * 8;
        if (ctx->band_scratch_bytes < bytes) {
            if (ctx->band_scratch) { AVC_HIP(hipDeviceSynchronize()); AVC_HIP(hipFree(ctx->band_scratch)); }
            ctx->band_scratch = nullptr; ctx->band_scratch_bytes = 0;
            AVC_HIP(hipMalloc(&ctx->band_scratch, bytes));
            ctx->band_scratch_bytes = bytes;
        }
        char *base = static_cast<char *>(ctx->band_scratch);
        slow_count = reinterpret_cast<int32_t *>(base);
        colflag = reinterpret_cast<uint8_t *>(base + 16);
        tile_skip = reinterpret_cast<int32_t *>(base + head_bytes);
        slow_list = tile_skip + p.ntiles;
        AVC_REQUIRE(ctx->recon.ready, AVC_ERR_STATE, "recon query: weights not packed (call avc_pack_recon_weights)");
    }
    if (int rc0 = set_all_lds()) return rc0;
    hipEvent_t e0, e1;
    timing_begin(ctx, 1, s, e0, e1, p);        // (a folded launch is timed with its column pass, a subset launch also with its prepass and its left-over tiles)
#if !AVC_CHECK_RANGE
    if (fold) {
        const int ncol = grid->res[0] * grid->res[1];
        if (fold == 2) {
            AVC_HIP(hipMemsetAsync(ctx->band_scratch, 0, head_bytes, s));
            hipLaunchKernelGGL(band_prepass_kernel, dim3((unsigned)p.ntiles), dim3(TILE_PTS), 0, s, p.gidx, p.n, p.grz, RECON_MAX_RUNS, colflag, tile_skip,
                               slow_list, slow_count);
        }
        hipLaunchKernelGGL(recon_column_terms_kernel, dim3(std::min((ncol + 7) / 8, ctx->num_cus * 4)), dim3(256), 0, s, p.feat, p.H, p.W, p.gx, p.gy,
                           (int)grid->res[0], (int)grid->res[1], p.cx, p.cy, (const float *)net.d_colw, static_cast<float *>(ctx->rcol_scratch), colflag);
        if (fold == 2) {
            p.tile_skip = tile_skip;
            rc = set_lds(recon_fold_kernel<2>);
            if (rc) return rc;
            hipLaunchKernelGGL(recon_fold_kernel<2>, dim3(grid_dim), dim3(256), LDS_BYTES, s, p);
            // the tiles it left out (a wavefront with more than two runs of columns), point by point on the generated coordinates
            QueryParams q = p;
            q.clk = nullptr; q.colterms = nullptr; q.tile_skip = nullptr;
            q.tile_list = slow_list; q.tile_count = slow_count;
            q.wstream = (const char *)ctx->recon.d_stream; q.bias = ctx->recon.d_bias;
            q.stream_bytes = bytes_until(ctx->recon, ctx->recon.chunks.size());
            rc = set_lds(recon_kernel);
            if (rc) return rc;
            hipLaunchKernelGGL(recon_kernel, dim3(grid_dim), dim3(256), LDS_BYTES, s, q);
        } else {
            rc = set_lds(recon_fold_kernel<1>);
            if (rc) return rc;
            hipLaunchKernelGGL(recon_fold_kernel<1>, dim3(grid_dim), dim3(256), LDS_BYTES, s, p);
        }
    } else
#endif
    {
        rc = set_lds(recon_kernel);
        if (rc) return rc;
        hipLaunchKernelGGL(recon_kernel, dim3(grid_dim), dim3(256), LDS_BYTES, s, p);
    }
    AVC_HIP(hipGetLastError());
    timing_end(ctx, 1, s, e0, e1);
    return range_end(ctx, s, "recon query");
}

}  // namespace AVC_FLAVOUR
}  // namespace avc
