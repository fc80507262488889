// Fused per-point MLP queries for gfx950 (MI355X):
//   avatar_kernel : OccupancyNet.query  = WarpingField.query + DoubleTNet.forward
//                   (reference network/arch_avatar.py:356-381, :113-140, :65-83)
//   recon_kernel  : ReconNetwork.infer's decoder loop (network/arch_recon.py:55-73)
//
// Design (DESIGN.md section "fused MLP"):
//   * one workgroup = 4 waves (one per SIMD, ~400 VGPRs each), persistent over 128-point tiles;
//     a wave owns 32 points and ALL hidden channels of them, so the whole 17-layer chain runs
//     out of registers: the D tile of one layer is, register for register, the B operand of the
//     next (mlp_layout.h).  No activation ever touches LDS or HBM.
//   * arithmetic: every fp32 product a*b is evaluated as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with
//     (hi, lo) fp16 pairs on v_mfma_f32_32x32x16_f16, fp32 accumulation (22+ significant bits,
//     ~1e-6 relative; the parity bar is 1e-4 absolute).  3 MFMA passes at 16x the fp32-MFMA rate.
//   * weights (3.5 MB, pre-split and pre-permuted by pack.cpp) stream L2 -> LDS with
//     global_load_lds_dwordx4 into a 2 x 64 KiB ring, one chunk ahead of the MFMAs that read it
//     (ds_read_b128, lane-linear => conflict-free); one barrier per chunk.
//   * prologue/epilogue work is fused: bilinear gather of the channel-last feature map,
//     positional encoding (accurate sincosf), bias (accumulator init), Softplus / ReLU /
//     LeakyReLU(0.02) / Sigmoid, the fp16 re-split, and the p + offset hand-off in fp32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "avcap_internal.h"
#include "mlp_layout.h"

namespace avc {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Frag { half8 hi, lo; };   // B operand of one k-step: 16 K-slots x 32 points, split fp16

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2, ACT_SOFTPLUS = 3 };

constexpr int WAVES = 4;
constexpr int TILE_PTS = 32 * WAVES;
constexpr int LDS_BYTES = 2 * layout::SLOT_BYTES;

struct QueryParams {
    const float *pts;        // (n,3)
    int64_t n;
    const float *feat;       // channel-last feature map (H, W, C)
    int H, W;
    float cx, cy, cz;
    const char *wstream;
    const ChunkDesc *chunks;
    int nchunks;
    const float *bias;
    float oscale[24];
    float *out0;             // occ / recon value (n)
    float *out1;             // offsets (n,3) or null
    float *out2;             // rgba (n,4) or null
    int sigmoid_occ;
    int64_t ntiles;
};

// ------------------------------------------------------------------------------------------
// weight stream: 2-slot LDS ring, one chunk of prefetch
// ------------------------------------------------------------------------------------------
struct Stream {
    const char *g;
    const ChunkDesc *tab;
    int nch, c;          // c = chunk about to be consumed
    unsigned parity;     // ring slot of chunk c
    int wave, lane;
};

__device__ __forceinline__ void stream_issue(const Stream &s, int chunk, unsigned slot)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ChunkDesc d = s.tab[chunk];
    const char *src = s.g + d.offset + s.lane * 16;
    char *dst = smem + slot * layout::SLOT_BYTES;
    for (unsigned o = s.wave * 1024u; o < d.bytes; o += WAVES * 1024u)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + o),
                                         (__attribute__((address_space(3))) void *)(dst + o), 16, 0, 0);
}

// Make chunk c readable (its loads were issued one chunk earlier), start loading chunk c+1 into
// the slot every wave has just finished reading, and return the LDS byte offset of chunk c.
__device__ __forceinline__ unsigned stream_acquire(Stream &s)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int nc = s.c + 1;
    if (nc == s.nch) nc = 0;
    stream_issue(s, nc, s.parity ^ 1u);
    const unsigned base = s.parity * layout::SLOT_BYTES + s.lane * 16;
    s.c = nc;
    s.parity ^= 1u;
    return base;
}

// ------------------------------------------------------------------------------------------
// MFMA over one chunk: KS k-steps x TPC output tiles, units ordered k-major
// ------------------------------------------------------------------------------------------
template <int KS, int TPC>
__device__ __forceinline__ void mma_chunk(unsigned base, const Frag *__restrict__ in, f32x16 *__restrict__ acc)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // software pipeline: the A fragments of k-step k+1 are fetched from LDS while the 3*TPC MFMAs of
    // k-step k issue; sched_barrier keeps the compiler from hoisting a whole chunk of ds_reads
    // (64 fragments = 256 VGPRs) above the first MFMA.
    half8 ah[2][TPC], al[2][TPC];
#pragma unroll
    for (int t = 0; t < TPC; ++t) {
        ah[0][t] = *reinterpret_cast<const half8 *>(smem + base + t * layout::UNIT_BYTES);
        al[0][t] = *reinterpret_cast<const half8 *>(smem + base + t * layout::UNIT_BYTES + 1024);
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        const int cur = k & 1, nxt = cur ^ 1;
        if (k + 1 < KS) {
#pragma unroll
            for (int t = 0; t < TPC; ++t) {
                ah[nxt][t] = *reinterpret_cast<const half8 *>(smem + base + ((k + 1) * TPC + t) * layout::UNIT_BYTES);
                al[nxt][t] = *reinterpret_cast<const half8 *>(smem + base + ((k + 1) * TPC + t) * layout::UNIT_BYTES + 1024);
            }
        }
#pragma unroll
        for (int t = 0; t < TPC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][t], in[k].hi, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TPC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][t], in[k].lo, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TPC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][t], in[k].hi, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// single output tile: two accumulators over even / odd k-steps break the dependent MFMA chain
template <int KS>
__device__ __forceinline__ f32x16 mma_head(unsigned base, const Frag *__restrict__ in, f32x16 init)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x16 a0 = init, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) a1[r] = 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        const half8 ah = *reinterpret_cast<const half8 *>(smem + base + k * layout::UNIT_BYTES);
        const half8 al = *reinterpret_cast<const half8 *>(smem + base + k * layout::UNIT_BYTES + 1024);
        if (k & 1) {
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, in[k].hi, a1, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, in[k].lo, a1, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, in[k].hi, a1, 0, 0, 0);
        } else {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, in[k].hi, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, in[k].lo, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, in[k].hi, a0, 0, 0, 0);
        }
    }
    return a0 + a1;
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
        const f32x4 v = *reinterpret_cast<const f32x4 *>(bias_rows + 8 * m + 4 * h);
        a[4 * m + 0] = v[0]; a[4 * m + 1] = v[1]; a[4 * m + 2] = v[2]; a[4 * m + 3] = v[3];
    }
    return a;
}

__device__ __forceinline__ float softplus_f(float x)
{
    // torch.nn.Softplus(beta=1, threshold=20): x > 20 ? x : log1p(exp(x))   (network/mlp.py:99)
    // = max(x,0) + ln2 * log2(1 + 2^(-|x| log2 e)); absolute error ~1e-7, see DESIGN.md
    const float t = __builtin_amdgcn_exp2f(-1.44269504088896341f * __builtin_fabsf(x));
    const float l = __builtin_amdgcn_logf(1.0f + t) * 0.69314718055994531f;
    return __builtin_fmaxf(x, 0.0f) + l;
}

template <int ACT>
__device__ __forceinline__ float act_f(float x)
{
    if constexpr (ACT == ACT_RELU) return __builtin_fmaxf(x, 0.0f);
    else if constexpr (ACT == ACT_LEAKY) return x > 0.0f ? x : 0.02f * x;     // network/mlp.py:11
    else if constexpr (ACT == ACT_SOFTPLUS) return softplus_f(x);
    else return x;
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ void split8(const float *v, half8 &hi, half8 &lo)
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h = (_Float16)v[e];
        hi[e] = h;
        lo[e] = (_Float16)(v[e] - (float)h);
    }
}

// accumulator tile -> activation -> the two B fragments it becomes for the next layer
template <int ACT>
__device__ __forceinline__ void tile_to_frags(const f32x16 &acc, float oscale, Frag &f0, Frag &f1)
{
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = act_f<ACT>(acc[r] * oscale);
    split8(v, f0.hi, f0.lo);
    split8(v + 8, f1.hi, f1.lo);
}

// ------------------------------------------------------------------------------------------
// dense layers
// ------------------------------------------------------------------------------------------
// NT output tiles (even), evaluated two at a time; up to two input segments accumulate into the
// same tiles (the reference's torch.cat on the channel axis).
template <int NT, int KS0, int KS1, int ACT>
__device__ __forceinline__ void dense(Stream &s, const Frag *__restrict__ in0, const Frag *__restrict__ in1,
                                      Frag *__restrict__ out, const float *bias, float oscale, int h)
{
#pragma unroll
    for (int p = 0; p < NT / 2; ++p) {
        f32x16 acc[2];
        acc[0] = bias_tile(bias + (2 * p) * 32, h);
        acc[1] = bias_tile(bias + (2 * p + 1) * 32, h);
        unsigned base = stream_acquire(s);
        mma_chunk<KS0, 2>(base, in0, acc);
        if constexpr (KS1 > 0) {
            base = stream_acquire(s);
            mma_chunk<KS1, 2>(base, in1, acc);
        }
        tile_to_frags<ACT>(acc[0], oscale, out[4 * p + 0], out[4 * p + 1]);
        tile_to_frags<ACT>(acc[1], oscale, out[4 * p + 2], out[4 * p + 3]);
    }
}

// one-tile linear head (rows 0..31 of which only the first few are real); returns scaled outputs
template <int KS>
__device__ __forceinline__ f32x16 head(Stream &s, const Frag *__restrict__ in, const float *bias, float oscale, int h)
{
    const f32x16 init = bias_tile(bias, h);
    const unsigned base = stream_acquire(s);
    f32x16 a = mma_head<KS>(base, in, init);
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] *= oscale;
    return a;
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
__device__ __forceinline__ void bilinear_frag(const Bilinear &b, int c, Frag &f)
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
    split8(v, f.hi, f.lo);
}

// NeRF positional encoding of q (3 floats) into the 4 k-steps of the PE layout (mlp_layout.h):
// lane-half h evaluates arguments 15h .. 15h+14: coordinate i%3, frequency 2^(5h + i/3) -- exact
// power-of-two scaling like the reference's x * freq (net_util.py:27-33), accurate sincosf.
__device__ __forceinline__ void posenc(const float q[3], int h, Frag *__restrict__ P)
{
    float v[32];
    const float hs = h ? 32.0f : 1.0f;
#pragma unroll
    for (int i = 0; i < 15; ++i) {
        const float arg = q[i % 3] * (float)(1 << (i / 3)) * hs;
        float sn, cs;
        sincosf(arg, &sn, &cs);
        v[2 * i] = sn;
        v[2 * i + 1] = cs;
    }
    v[30] = h ? q[2] : q[0];
    v[31] = h ? 0.0f : q[1];
#pragma unroll
    for (int k = 0; k < 4; ++k) split8(v + 8 * k, P[k].hi, P[k].lo);
}

// ------------------------------------------------------------------------------------------
// avatar query kernel
// ------------------------------------------------------------------------------------------
template <bool WARP, bool COLOUR>
__global__ __launch_bounds__(256, 1) void avatar_kernel(const QueryParams p)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;

    Stream s;
    s.g = p.wstream; s.tab = p.chunks; s.nch = p.nchunks; s.c = 0; s.parity = 0; s.wave = wave; s.lane = lane;
    stream_issue(s, 0, 0);

    for (int64_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int64_t pidx_raw = tile * TILE_PTS + wave * 32 + j;
        const int64_t pidx = pidx_raw < p.n ? pidx_raw : p.n - 1;
        float pt[3];
        pt[0] = p.pts[pidx * 3 + 0]; pt[1] = p.pts[pidx * 3 + 1]; pt[2] = p.pts[pidx * 3 + 2];

        Frag X[16], Y[16];
        const float *bias = p.bias;
        asm volatile("" : "+s"(bias));   // opaque per tile: stops LICM from hoisting ~60 tiles of bias loads out of the loop
        int li = 0;                      // layer counter (oscale index); compile-time after unrolling
        float q[3] = {pt[0], pt[1], pt[2]};
        float off[3] = {0.f, 0.f, 0.f};

        if constexpr (WARP) {
            // ---- WarpingField.query (arch_avatar.py:113-140) ----
            Frag S[layout::IN67_KS];
            {
                const Bilinear bl = bilinear_setup<64>(p.feat, p.H, p.W, pt[0] - p.cx, -(pt[1] - p.cy), 32 * h);   // :125-133
#pragma unroll
                for (int k = 0; k < 4; ++k) { bilinear_frag(bl, 8 * k, S[k]); __builtin_amdgcn_sched_barrier(0); }
                float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (h == 0) { z[0] = pt[0]; z[1] = pt[1]; z[2] = pt[2]; }                             // pos_encoding 0 => raw xyz
                split8(z, S[4].hi, S[4].lo);
            }
            dense<8, layout::IN67_KS, 0, ACT_SOFTPLUS>(s, S, nullptr, X, bias, p.oscale[li], h); bias += 256; ++li;   // conv1+bn1
            dense<8, 16, 0, ACT_SOFTPLUS>(s, X, nullptr, Y, bias, p.oscale[li], h); bias += 256; ++li;                 // conv2
            dense<8, 16, 0, ACT_SOFTPLUS>(s, Y, nullptr, X, bias, p.oscale[li], h); bias += 256; ++li;                 // conv3
            dense<8, 16, 0, ACT_SOFTPLUS>(s, X, nullptr, Y, bias, p.oscale[li], h); bias += 256; ++li;                 // conv4
            dense<8, 16, layout::IN67_KS, ACT_SOFTPLUS>(s, Y, S, X, bias, p.oscale[li], h); bias += 256; ++li;         // conv5 on [x0|x4]
            dense<8, 16, 0, ACT_SOFTPLUS>(s, X, nullptr, Y, bias, p.oscale[li], h); bias += 256; ++li;                 // conv6
            dense<8, 16, 0, ACT_SOFTPLUS>(s, Y, nullptr, X, bias, p.oscale[li], h); bias += 256; ++li;                 // conv7
            const f32x16 o = head<16>(s, X, bias, p.oscale[li], h); bias += 32; ++li;                                  // out_layer_coord_affine
            // rows 0..2 live in lanes h == 0, regs 0..2: broadcast to the other half
            off[0] = __shfl(o[0], j, 64); off[1] = __shfl(o[1], j, 64); off[2] = __shfl(o[2], j, 64);
            q[0] = pt[0] + off[0]; q[1] = pt[1] + off[1]; q[2] = pt[2] + off[2];                                       // arch_avatar.py:372 (fp32 add)
        }

        // ---- DoubleTNet.forward (arch_avatar.py:65-83) ----
        Frag P[layout::PE_KS];
        posenc(q, h, P);                                                                                               // :70
        dense<8, layout::PE_KS, 0, ACT_RELU>(s, P, nullptr, X, bias, p.oscale[li], h); bias += 256; ++li;             // shared 0
        dense<8, 16, 0, ACT_RELU>(s, X, nullptr, Y, bias, p.oscale[li], h); bias += 256; ++li;
        dense<8, 16, 0, ACT_RELU>(s, Y, nullptr, X, bias, p.oscale[li], h); bias += 256; ++li;
        dense<8, 16, 0, ACT_RELU>(s, X, nullptr, Y, bias, p.oscale[li], h); bias += 256; ++li;
        dense<8, 16, layout::PE_KS, ACT_RELU>(s, Y, P, X, bias, p.oscale[li], h); bias += 256; ++li;                  // shared 4 on [x|x0]
        dense<8, 16, 0, ACT_RELU>(s, X, nullptr, Y, bias, p.oscale[li], h); bias += 256; ++li;
        dense<8, 16, 0, ACT_NONE>(s, Y, nullptr, X, bias, p.oscale[li], h); bias += 256; ++li;                        // shared 6: no activation (mlp.py:46,64)
        dense<4, 16, 0, ACT_LEAKY>(s, X, nullptr, Y, bias, p.oscale[li], h); bias += 128; ++li;                       // geo 0
        const f32x16 g = head<8>(s, Y, bias, p.oscale[li], h); bias += 32; ++li;                                       // geo 1: row 0 = occ/sdf, row 1 = sigma

        const bool writer = (h == 0) && (pidx_raw < p.n);
        if (writer) {
            p.out0[pidx_raw] = p.sigmoid_occ ? sigmoid_f(g[0]) : g[0];                                                 // :77-80
            if (WARP && p.out1) { p.out1[pidx_raw * 3 + 0] = off[0]; p.out1[pidx_raw * 3 + 1] = off[1]; p.out1[pidx_raw * 3 + 2] = off[2]; }
        }
        if constexpr (COLOUR) {
            dense<8, 16, 0, ACT_RELU>(s, X, nullptr, Y, bias, p.oscale[li], h); bias += 256; ++li;                    // clr 0
            dense<4, 16, 0, ACT_RELU>(s, Y, nullptr, X, bias, p.oscale[li], h); bias += 128; ++li;                    // clr 1
            const f32x16 c = head<8>(s, X, bias, p.oscale[li], h); bias += 32; ++li;                                   // clr 2
            if (writer && p.out2) {
                f32x4 rgba = {sigmoid_f(c[0]), sigmoid_f(c[1]), sigmoid_f(c[2]), __builtin_fmaxf(g[1], 0.0f)};         // :75-76
                *reinterpret_cast<f32x4 *>(p.out2 + pidx_raw * 4) = rgba;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// recon query kernel (arch_recon.py:55-73): [feat(32) | z] -> 512 -> 256 -> 128 -> 1, LeakyReLU(0.02),
// res @ 1,2, sigmoid.  fc0 is produced in two 256-channel halves so that fc1 can consume each half
// while it is the only hidden state alive (see pack.cpp::pack_recon for the matching stream order).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void recon_kernel(const QueryParams p)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;

    Stream s;
    s.g = p.wstream; s.tab = p.chunks; s.nch = p.nchunks; s.c = 0; s.parity = 0; s.wave = wave; s.lane = lane;
    stream_issue(s, 0, 0);

    for (int64_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int64_t pidx_raw = tile * TILE_PTS + wave * 32 + j;
        const int64_t pidx = pidx_raw < p.n ? pidx_raw : p.n - 1;
        const float px = p.pts[pidx * 3 + 0] - p.cx, py = p.pts[pidx * 3 + 1] - p.cy, pz = p.pts[pidx * 3 + 2] - p.cz;   // :62

        Frag I[layout::IN33_KS];
        {
            const Bilinear bl = bilinear_setup<32>(p.feat, p.H, p.W, px, -py, 16 * h);            // :63-68
            bilinear_frag(bl, 0, I[0]);
            bilinear_frag(bl, 8, I[1]);
            float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (h == 0) z[0] = pz;                                                               // :69
            split8(z, I[2].hi, I[2].lo);
        }
        Frag X[16], Y[16];
        const float *bias = p.bias;
        asm volatile("" : "+s"(bias));   // see avatar_kernel
        f32x16 acc[8];
        // fc0 rows 0..255
        dense<8, layout::IN33_KS, 0, ACT_LEAKY>(s, I, nullptr, X, bias, p.oscale[0], h); bias += 256;
        // fc1 partial over x[0..255]: 8 tiles live, 4 k-steps per chunk
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = bias_tile(bias + 32 * t, h);
        bias += 256;
#pragma unroll
        for (int c = 0; c < 4; ++c) { const unsigned base = stream_acquire(s); mma_chunk<4, 8>(base, X + 4 * c, acc); }
        // fc0 rows 256..511
        dense<8, layout::IN33_KS, 0, ACT_LEAKY>(s, I, nullptr, X, bias, p.oscale[2], h); bias += 256;
        bias += 256;   // (zero bias block of the second fc1 pack call)
#pragma unroll
        for (int c = 0; c < 4; ++c) { const unsigned base = stream_acquire(s); mma_chunk<4, 8>(base, X + 4 * c, acc); }
        { const unsigned base = stream_acquire(s); mma_chunk<layout::IN33_KS, 8>(base, I, acc); }
#pragma unroll
        for (int t = 0; t < 8; ++t) tile_to_frags<ACT_LEAKY>(acc[t], p.oscale[1], Y[2 * t], Y[2 * t + 1]);
        // fc2 on [x(256) | in(33)] -> 128
        dense<4, 16, layout::IN33_KS, ACT_LEAKY>(s, Y, I, X, bias, p.oscale[4], h); bias += 128;
        const f32x16 o = head<8>(s, X, bias, p.oscale[5], h);
        if (h == 0 && pidx_raw < p.n) p.out0[pidx_raw] = sigmoid_f(o[0]);                         // last_op sigmoid
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

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

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static void timing_begin(avc_ctx *ctx, int which, hipStream_t s, hipEvent_t &e0, hipEvent_t &e1)
{
    e0 = e1 = nullptr;
    if (!ctx->timing.enabled) return;
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

int launch_avatar(avc_ctx *ctx, const float *pts, int64_t n, const float center[3], int occ_sigmoid,
                  float *occ, float *offset, float *rgba, bool template_only, hipStream_t s)
{
    PackedNet &net = template_only ? ctx->tmpl_only : ctx->warp_tmpl;
    AVC_REQUIRE(net.ready, AVC_ERR_STATE, "avatar query: weights not packed (call avc_pack_warp_weights and avc_pack_template_weights)");
    AVC_REQUIRE(template_only || ctx->pose_feat_hwc, AVC_ERR_STATE, "avatar query: pose feature map not set (WarpingField.precompute_conv)");
    AVC_REQUIRE(!rgba || net.has_colour, AVC_ERR_STATE, "avatar query: rgba requested but clr_mlp weights were not packed");
    if (n == 0) return AVC_OK;
    QueryParams p{};
    p.pts = pts; p.n = n; p.feat = ctx->pose_feat_hwc; p.H = ctx->pose_H; p.W = ctx->pose_W;
    p.cx = center ? center[0] : 0.f; p.cy = center ? center[1] : 0.f; p.cz = center ? center[2] : 0.f;
    p.wstream = (const char *)net.d_stream; p.chunks = net.d_chunks; p.bias = net.d_bias;
    AVC_REQUIRE(net.oscale.size() <= 24, AVC_ERR_STATE, "internal: too many layers");
    for (size_t i = 0; i < net.oscale.size(); ++i) p.oscale[i] = net.oscale[i];
    p.out0 = occ; p.out1 = offset; p.out2 = rgba; p.sigmoid_occ = occ_sigmoid;
    p.ntiles = (n + TILE_PTS - 1) / TILE_PTS;
    const bool colour = rgba != nullptr;
    // the chunk table of a colour-capable stream ends with the clr chunks; a geometry-only launch
    // simply wraps around before them (their count is fixed by pack.cpp: 2 + 2 + 1 chunks... see below)
    int nch = (int)net.chunks.size();
    if (net.has_colour && !colour) nch -= 4 + 2 + 1;   // clr0: 4 pair chunks, clr1: 2, clr2: 1
    p.nchunks = nch;
    const int grid = (int)std::min<int64_t>(p.ntiles, ctx->num_cus);
    hipEvent_t e0, e1;
    timing_begin(ctx, 0, s, e0, e1);
    int rc = AVC_OK;
#define LAUNCH(W_, C_)                                                                              \
    do {                                                                                            \
        rc = set_lds(avatar_kernel<W_, C_>);                                                        \
        if (rc) return rc;                                                                          \
        hipLaunchKernelGGL((avatar_kernel<W_, C_>), dim3(grid), dim3(256), LDS_BYTES, s, p);       \
    } while (0)
    if (template_only) { if (colour) LAUNCH(false, true); else LAUNCH(false, false); }
    else               { if (colour) LAUNCH(true, true);  else LAUNCH(true, false); }
#undef LAUNCH
    AVC_HIP(hipGetLastError());
    timing_end(ctx, 0, s, e0, e1);
    return AVC_OK;
}

int launch_recon(avc_ctx *ctx, const float *pts, int64_t n, const float center[3], float *out, hipStream_t s)
{
    PackedNet &net = ctx->recon;
    AVC_REQUIRE(net.ready, AVC_ERR_STATE, "recon query: weights not packed (call avc_pack_recon_weights)");
    AVC_REQUIRE(ctx->img_feat_hwc, AVC_ERR_STATE, "recon query: image feature map not set");
    if (n == 0) return AVC_OK;
    QueryParams p{};
    p.pts = pts; p.n = n; p.feat = ctx->img_feat_hwc; p.H = ctx->img_H; p.W = ctx->img_W;
    p.cx = center[0]; p.cy = center[1]; p.cz = center[2];
    p.wstream = (const char *)net.d_stream; p.chunks = net.d_chunks; p.bias = net.d_bias;
    for (size_t i = 0; i < net.oscale.size(); ++i) p.oscale[i] = net.oscale[i];
    p.nchunks = (int)net.chunks.size();
    p.out0 = out;
    p.ntiles = (n + TILE_PTS - 1) / TILE_PTS;
    const int grid = (int)std::min<int64_t>(p.ntiles, ctx->num_cus);
    int rc = set_lds(recon_kernel);
    if (rc) return rc;
    hipEvent_t e0, e1;
    timing_begin(ctx, 1, s, e0, e1);
    hipLaunchKernelGGL(recon_kernel, dim3(grid), dim3(256), LDS_BYTES, s, p);
    AVC_HIP(hipGetLastError());
    timing_end(ctx, 1, s, e0, e1);
    return AVC_OK;
}

}  // namespace avc
