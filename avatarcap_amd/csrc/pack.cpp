// Host-side weight packer: folds BatchNorm1d / weight_norm, splits every fp32 weight into an
// (fp16 hi, fp16 lo) pair, permutes the K axis into the kernels' register "slot" order and lays the
// result out as the chunk stream the fused kernels stream through LDS (see mlp_layout.h and
// fused_mlp.hip).  Accepts exactly the tensors of the reference checkpoints.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>

#include "avcap_internal.h"
#include "mlp_layout.h"

namespace avc {

static thread_local std::string g_err;
void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}
const char *last_error() { return g_err.c_str(); }

namespace {

// ---- weights beyond the fp16 range ------------------------------------------------------------------------------------------------------------------------
// The kernels multiply split-fp16 operands: a weight's `hi` half must be a finite fp16 number.  Round 5 refused any layer with |w| > 3e4 -- a BatchNorm fold with
// a small running variance, or a weight_norm gain, can produce one -- and had no remedy.  Two remedies, both exact (powers of two):
//   * ReLU / LeakyReLU / linear layers are positively homogeneous: act(2^s z) = 2^s act(z).  rebalance() scales an output ROW that is out of range (and its
//     bias) by 2^s, s < 0, down to the magnitude of the layer's ordinary rows, and the matching input COLUMN of every consumer by 2^-s: the network computes the
//     same function, only the offending channel and what reads it change (a trained network that carries a huge row carries a tiny column behind it: both come
//     back to ordinary magnitudes, where the split has its full precision and the channel's activation fits fp16 again), and the kernels know nothing of it -- the default build, no instruction added.  A consumer row pushed over the limit by the compensation
//     is rebalanced in its turn; a HEAD (its output is the result) has nowhere to pass a scale on to and is still refused.
//   * Softplus is not homogeneous: the warping field's seven Conv1d + BatchNorm1d + Softplus layers share ONE scale 2^s on weights and biases -- the mildest
//     that brings the largest weight under the limit, because it costs every other weight |s| bits of its `lo` half -- and the `scaled` build of the kernels
//     (fused_mlp.hip AVC_LAYER_SCALE) multiplies the accumulator by 2^-s in front of the Softplus (PackedNet::sp_unscale).
constexpr double W_LIMIT = 30000.0;

double max_abs(const std::vector<double> &v) { double m = 0; for (double x : v) m = std::fmax(m, std::fabs(x)); return m; }
// the s <= 0 closest to 0 with m 2^s <= W_LIMIT
int down_exponent(double m)
{
    int s = 0;
    while (m > W_LIMIT && s > -1000) { m *= 0.5; --s; }
    return s;
}
void scale_all(std::vector<double> &v, int s) { const double f = std::ldexp(1.0, s); for (double &x : v) x *= f; }

struct Link { int consumer, col0; };       // the producer's outputs are columns [col0, col0 + cout) of the consumer's input
struct Node { std::vector<double> *W, *b; int cout, cin; bool head; std::vector<Link> out; };
// in topological order; false: a head layer is out of range.  `rows_scaled`: how many rows were touched.
// An out-of-range row is brought to the magnitude of the layer's ORDINARY rows (the median row maximum), not merely under the limit: its activation shrinks by
// the same factor, and an activation has to fit fp16 as well (a row left at 2^14 would still put 2^18 times the ordinary values into the next layer's operands).
bool rebalance(std::vector<Node> &g, int *rows_scaled = nullptr)
{
    for (Node &n : g) {
        std::vector<double> rmax(n.cout, 0.0), ordinary;
        for (int r = 0; r < n.cout; ++r) {
            for (int i = 0; i < n.cin; ++i) rmax[r] = std::fmax(rmax[r], std::fabs((*n.W)[(size_t)r * n.cin + i]));
            if (rmax[r] <= W_LIMIT && rmax[r] > 0.0) ordinary.push_back(rmax[r]);
        }
        std::sort(ordinary.begin(), ordinary.end());
        const double med = ordinary.empty() ? 1.0 : ordinary[ordinary.size() / 2];
        for (int r = 0; r < n.cout; ++r) {
            if (!(rmax[r] > W_LIMIT)) continue;
            if (n.head) return false;
            const int s = (int)std::floor(std::log2(med / rmax[r]));                 // rmax 2^s in (med / 2, med]
            const double dn = std::ldexp(1.0, s), up = std::ldexp(1.0, -s);
            for (int i = 0; i < n.cin; ++i) (*n.W)[(size_t)r * n.cin + i] *= dn;
            if (n.b) (*n.b)[r] *= dn;
            if (rows_scaled) ++*rows_scaled;
            for (const Link &k : n.out) {
                Node &c = g[k.consumer];
                for (int q = 0; q < c.cout; ++q) (*c.W)[(size_t)q * c.cin + k.col0 + r] *= up;
            }
        }
    }
    return true;
}

struct Seg { int ks; std::function<int(int)> col; };   // slot -> column of W (or -1 = zero)

struct Builder {
    PackedNet &net;
    bool overflow = false;
    explicit Builder(PackedNet &n) : net(n) { net.stream.clear(); net.chunks.clear(); net.bias.clear(); net.colw.clear(); net.sp_unscale = 1.0f; }

    // W: (cout, cin) row-major effective weights, b: (cout).  tpc = tiles per chunk (accumulators
    // live at once in the kernel), kspc = k-steps per chunk.
    void layer(const std::vector<double> &W, const std::vector<double> &b, int cout, int cin,
               const std::vector<Seg> &segs, int tpc, int kspc)
    {
        const int nt_raw = (cout + 31) / 32;
        const int nt = ((nt_raw + tpc - 1) / tpc) * tpc;
        double m = 0;
        for (double w : W) m = std::fmax(m, std::fabs(w));
        // No per-layer rescale here: gfx950 MFMA does not flush fp16 subnormal inputs, a subnormal `lo` still carries the residual to 2^-25 absolute, and small
        // weights need none.  LARGE weights are dealt with before they get here (rebalance() / the warping field's Softplus scale, below); what still exceeds
        // the range at this point is a head layer, which has nowhere to put a scale.
        if (m > W_LIMIT) overflow = true;     // reported by the caller: |w| must stay below the fp16 range
        const double scale = 1.0;
        for (int r = 0; r < nt * 32; ++r) net.bias.push_back(r < cout ? (float)(b[r] * scale) : 0.0f);
        for (int g = 0; g < nt / tpc; ++g)
            for (const Seg &sg : segs)
                for (int k0 = 0; k0 < sg.ks; k0 += kspc) {
                    const int kn = std::min(kspc, sg.ks - k0);
                    ChunkDesc cd{(uint32_t)net.stream.size(), (uint32_t)(kn * tpc * layout::UNIT_BYTES)};
                    net.stream.resize(net.stream.size() + cd.bytes);
                    _Float16 *dst = reinterpret_cast<_Float16 *>(net.stream.data() + cd.offset);
                    for (int k = 0; k < kn; ++k)
                        for (int tt = 0; tt < tpc; ++tt) {
                            _Float16 *hi = dst + (size_t)(k * tpc + tt) * (layout::UNIT_BYTES / 2);
                            _Float16 *lo = hi + 512;
                            const int tile = g * tpc + tt;
                            for (int lane = 0; lane < 64; ++lane)
                                for (int e = 0; e < 8; ++e) {
                                    const int row = tile * 32 + (lane & 31);
                                    const int slot = (k0 + k) * 16 + (lane >> 5) * 8 + e;
                                    const int c = sg.col(slot);
                                    float w = 0.0f;
                                    if (row < cout && c >= 0 && c < cin) w = (float)(W[(size_t)row * cin + c] * scale);
                                    const _Float16 h = (_Float16)w;
                                    hi[lane * 8 + e] = h;
                                    lo[lane * 8 + e] = (_Float16)(w - (float)h);
                                }
                        }
                    net.chunks.push_back(cd);
                }
    }
};

Seg seg_d(int ks, int col_offset = 0)
{
    return Seg{ks, [col_offset](int s) { return col_offset + layout::slot_channel_d(s); }};
}
Seg seg_in67(int col_offset = 0)
{
    return Seg{layout::IN67_KS, [col_offset](int s) { int c = layout::in67_column(s); return c < 0 ? -1 : col_offset + c; }};
}
Seg seg_xyz()       // the xyz k-step of the 67-wide input alone (column-folded streams)
{
    return Seg{1, [](int s) { return layout::in67_column(4 * 16 + s); }};
}
Seg seg_pe(int col_offset = 0, int L = 10)      // the first 3 + 6 L columns of the embedding (cano_template.pos_encoding = L <= 10: net_util.py:40-55)
{
    return Seg{layout::PE_KS, [col_offset, L](int s) { int c = layout::pe_column_l(s, L); return c < 0 ? -1 : col_offset + c; }};
}
Seg seg_inpe(int L)       // [posenc_L(xyz) | feat(64)] of a warping field with pos_encoding = L > 0: 8 k-steps (mlp_layout.h)
{
    return Seg{layout::INPE_KS, [L](int s) { return layout::inpe_column(s, L); }};
}
Seg seg_z33(int col_offset = 0)       // the z k-step of the 33-wide input alone (column-folded recon stream)
{
    return Seg{1, [col_offset](int s) { int c = layout::in33_column(2 * 16 + s); return c < 0 ? -1 : col_offset + c; }};
}
Seg seg_in33(int col_offset = 0)
{
    return Seg{layout::IN33_KS, [col_offset](int s) { int c = layout::in33_column(s); return c < 0 ? -1 : col_offset + c; }};
}

}  // namespace

// Effective (cout x cin) double weights + bias of one Conv1d(k=1), weight_norm / BatchNorm folded.
int effective(const avc_dense &d, const avc_bn *bn, std::vector<double> &W, std::vector<double> &b)
{
    AVC_REQUIRE(d.w && d.b && d.cout > 0 && d.cin > 0, AVC_ERR_ARG, "avc_dense: null pointer or non-positive shape");
    W.assign((size_t)d.cout * d.cin, 0.0);
    b.assign(d.cout, 0.0);
    for (int o = 0; o < d.cout; ++o) {
        double s = 1.0;
        if (d.g) {   // w = g * v / ||v||  (torch weight_norm, dim 0)
            double n2 = 0;
            for (int i = 0; i < d.cin; ++i) n2 += (double)d.w[(size_t)o * d.cin + i] * d.w[(size_t)o * d.cin + i];
            s = (double)d.g[o] / std::sqrt(n2);
        }
        double bs = 1.0, bb = 0.0;
        if (bn) {   // y = (x - mean) / sqrt(var + eps) * gamma + beta
            bs = (double)bn->gamma[o] / std::sqrt((double)bn->var[o] + (double)bn->eps);
            bb = (double)bn->beta[o] - (double)bn->mean[o] * bs;
        }
        for (int i = 0; i < d.cin; ++i) W[(size_t)o * d.cin + i] = (double)d.w[(size_t)o * d.cin + i] * s * bs;
        b[o] = (double)d.b[o] * bs + bb;
    }
    return AVC_OK;
}

void release(PackedNet &net)
{
    if (net.d_stream) hipFree(net.d_stream);
    if (net.d_chunks) hipFree(net.d_chunks);
    if (net.d_bias) hipFree(net.d_bias);
    if (net.d_colw) hipFree(net.d_colw);
    net.d_stream = nullptr; net.d_chunks = nullptr; net.d_bias = nullptr; net.d_colw = nullptr; net.ready = false;
}

int upload(PackedNet &net)
{
    release(net);
    AVC_HIP(hipMalloc(&net.d_stream, net.stream.size()));
    AVC_HIP(hipMalloc((void **)&net.d_chunks, net.chunks.size() * sizeof(ChunkDesc)));
    net.bias.resize(net.bias.size() + 64, 0.0f);    // BiasQueue fetches 64 floats ahead of a 32-float head block
    AVC_HIP(hipMalloc((void **)&net.d_bias, net.bias.size() * sizeof(float)));
    AVC_HIP(hipMemcpy(net.d_stream, net.stream.data(), net.stream.size(), hipMemcpyHostToDevice));
    AVC_HIP(hipMemcpy(net.d_chunks, net.chunks.data(), net.chunks.size() * sizeof(ChunkDesc), hipMemcpyHostToDevice));
    AVC_HIP(hipMemcpy(net.d_bias, net.bias.data(), net.bias.size() * sizeof(float), hipMemcpyHostToDevice));
    if (!net.colw.empty()) {
        AVC_HIP(hipMalloc((void **)&net.d_colw, net.colw.size() * sizeof(float)));
        AVC_HIP(hipMemcpy(net.d_colw, net.colw.data(), net.colw.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    net.ready = true;
    return AVC_OK;
}

// Layer order here IS the kernel's consumption order (avatar_kernel in fused_mlp.hip).
//
// Softplus in the log2 domain: softplus(x) = ln2 * log2(1 + 2^(x log2 e)).  The kernel's epilogue only
// evaluates y' = log2(1 + 2^m) (v_exp_f32 / v_log_f32), so for every Softplus layer the weights and
// bias are pre-multiplied by log2(e) (the accumulator becomes m) and every consumer of a Softplus output
// absorbs the missing ln(2) into its weight columns.  For a Softplus layer fed by a Softplus layer the
// two factors cancel (log2 e * ln 2 = 1): only the bias changes.
static void add_warp(Builder &B, const avc_ctx::Staged &w, bool fold = false, int L = 0)
{
    const double LOG2E = 1.4426950408889634074, LN2 = 0.69314718055994530942;
    auto scaled = [](std::vector<double> v, double f, int cin = 0, int c0 = 0, int c1 = -1) {
        if (c1 < 0) { for (double &x : v) x *= f; return v; }
        for (size_t i = 0; i < v.size(); ++i) { const int c = (int)(i % cin); if (c >= c0 && c < c1) v[i] *= f; }
        return v;
    };
    // fold: a dense launch whose tiles lie in one (x, y) column gets the 64 pose-feature columns of conv1 / conv5 as a per-column accumulator
    // init (column_terms_kernel, fused_mlp.hip): the stream keeps the xyz k-step only, the fp32 weights of the 64 columns go to net.colw
    const int D = 3 + 6 * L + 64;          // conv1's input: [posenc_L(xyz) | feat(64)] (arch_avatar.py:100,136); L = 0: [xyz | feat] = 67
    const std::vector<double> W1 = scaled(w.W[0], LOG2E), W5 = scaled(w.W[4], LOG2E, D + 256, 0, D);
    if (fold) {           // (L == 0 only: pack_avatar)
        B.net.colw.assign((size_t)2 * 256 * 64, 0.0f);
        for (int o = 0; o < 256; ++o)
            for (int c = 0; c < 64; ++c) {
                B.net.colw[(size_t)o * 64 + c] = (float)W1[(size_t)o * 67 + 3 + c];                       // columns [xyz(3) | feat(64)] (arch_avatar.py:136)
                B.net.colw[(size_t)(256 + o) * 64 + c] = (float)W5[(size_t)o * 323 + 3 + c];              // conv5 sees cat([x0, x4]) (mlp.py:106)
            }
    }
    const Seg in = L > 0 ? seg_inpe(L) : (fold ? seg_xyz() : seg_in67());
    B.layer(W1, scaled(w.b[0], LOG2E), 256, D, {in}, fold ? 8 : 2, 16);                                     // conv1 (raw inputs; folded: one wide chunk)
    for (int i = 1; i <= 3; ++i) B.layer(w.W[i], scaled(w.b[i], LOG2E), 256, 256, {seg_d(16)}, 2, 16);      // conv2..4
    B.layer(W5, scaled(w.b[4], LOG2E), 256, D + 256, {seg_d(16, D), in}, 2, 16);                            // conv5: cat([x0 (raw), x4 (softplus)]) (mlp.py:106)
    for (int i = 5; i <= 6; ++i) B.layer(w.W[i], scaled(w.b[i], LOG2E), 256, 256, {seg_d(16)}, 2, 16);      // conv6..7
    B.layer(scaled(w.W[7], LN2), w.b[7], 3, 256, {seg_d(16)}, 1, 16);             // out_layer_coord_affine (linear consumer)
}

// `colour` streams keep shared.6 (its output feeds both heads).  Geometry-only streams fold shared.6
// (linear, no activation: mlp.py:46,64) into geo.0:  W_g0 (W_6 x + b_6) + b_g0 = (W_g0 W_6) x + (W_g0 b_6 + b_g0),
// an exact identity that removes 65,536 of the 886,784 MAC per point (and one epilogue).
static void add_template(Builder &B, const avc_ctx::Staged &t, bool colour, bool warp, int L = 10)
{
    const int P = 3 + 6 * L;               // embedding width of cano_template.pos_encoding = L (arch_avatar.py:33-36); the kernel evaluates all ten octaves,
                                           // the columns of octaves >= L pack as zero weights
    B.layer(t.W[0], t.b[0], 256, P, {seg_pe(0, L)}, warp && colour ? 2 : 8, 16);    // shared 0: one wide chunk (k-major over all eight tiles); the warped colour kernel keeps tile pairs
    for (int i = 1; i <= 3; ++i) B.layer(t.W[i], t.b[i], 256, 256, {seg_d(16)}, 2, 16);
    B.layer(t.W[4], t.b[4], 256, 256 + P, {seg_d(16), seg_pe(256, L)}, 2, 16);   // shared 4: cat([x, x0]) (mlp.py:61)
    B.layer(t.W[5], t.b[5], 256, 256, {seg_d(16)}, 2, 16);
    if (colour) {
        B.layer(t.W[6], t.b[6], 256, 256, {seg_d(16)}, 2, 16);                   // shared 6 (linear)
        B.layer(t.W[7], t.b[7], 128, 256, {seg_d(16)}, 2, 16);                   // geo 0
    } else {
        std::vector<double> Wf((size_t)128 * 256, 0.0), bf(128, 0.0);
        for (int o = 0; o < 128; ++o) {
            double acc_b = t.b[7][o];
            for (int m = 0; m < 256; ++m) {
                const double g = t.W[7][(size_t)o * 256 + m];
                acc_b += g * t.b[6][m];
                for (int i = 0; i < 256; ++i) Wf[(size_t)o * 256 + i] += g * t.W[6][(size_t)m * 256 + i];
            }
            bf[o] = acc_b;
        }
        B.layer(Wf, bf, 128, 256, {seg_d(16)}, 2, 16);                           // geo 0 o shared 6
    }
    B.layer(t.W[8], t.b[8], 2, 128, {seg_d(8)}, 1, 16);                          // geo 1
    if (colour) {
        B.layer(t.W[9], t.b[9], 256, 256, {seg_d(16)}, 2, 16);                   // clr 0
        B.layer(t.W[10], t.b[10], 128, 256, {seg_d(16)}, 2, 16);                 // clr 1
        B.layer(t.W[11], t.b[11], 3, 128, {seg_d(8)}, 1, 16);                    // clr 2
    }
}

int pack_avatar(avc_ctx *ctx)
{
    const bool has_clr = ctx->tmpl_st.W.size() == 12;
    // working copies: weights beyond the fp16 range are brought inside before anything is packed (see rebalance())
    avc_ctx::Staged tmpl = ctx->tmpl_st, warp_w = ctx->warp_st;
    float sp_unscale = 1.0f;
    if (ctx->tmpl_set) {
        // shared.0 .. 5 (ReLU) -> shared.6 (linear) -> geo.0 (LeakyReLU) -> geo.1 (head); shared.6 -> clr.0 -> clr.1 (ReLU) -> clr.2 (head); shared.4 = [x | posenc]
        std::vector<Node> g;
        const int n = (int)tmpl.W.size();
        for (int i = 0; i < n; ++i) {
            const int co = (int)tmpl.b[i].size(), ci = (int)(tmpl.W[i].size() / tmpl.b[i].size());          // (Staged keeps no shapes: rows = biases)
            g.push_back(Node{&tmpl.W[i], &tmpl.b[i], co, ci, i == 8 || i == 11, {}});
        }
        for (int i = 0; i < 6; ++i) g[i].out.push_back(Link{i + 1, 0});
        g[6].out.push_back(Link{7, 0});
        g[7].out.push_back(Link{8, 0});
        if (has_clr) { g[6].out.push_back(Link{9, 0}); g[9].out.push_back(Link{10, 0}); g[10].out.push_back(Link{11, 0}); }
        AVC_REQUIRE(rebalance(g), AVC_ERR_ARG, "cano_template: an output layer (geo_mlp / clr_mlp last fc) has weights beyond 3e4 in magnitude, also after the layers "
                    "in front of it were rebalanced: not representable by the split-fp16 kernels");
    }
    if (ctx->warp_set) {
        // conv1 .. conv7 + BatchNorm1d + Softplus: one scale for the seven (the accumulator becomes m log2(e) 2^s; the scaled kernels undo 2^s before the Softplus)
        double m = 0;
        for (int i = 0; i < 7; ++i) m = std::fmax(m, max_abs(warp_w.W[i]) * 1.4426950408889634074);
        if (const int s = down_exponent(m)) {
            for (int i = 0; i < 7; ++i) { scale_all(warp_w.W[i], s); scale_all(warp_w.b[i], s); }
            sp_unscale = (float)std::ldexp(1.0, -s);
        }
    }
    auto build = [&](PackedNet &net, bool warp, bool colour, bool fold = false) -> int {
        Builder B(net);
        if (warp) add_warp(B, warp_w, fold, ctx->warp_pe);
        add_template(B, tmpl, colour, warp, ctx->tmpl_pe);
        AVC_REQUIRE(!B.overflow, AVC_ERR_ARG, "weights exceed 3e4 in magnitude (warping_field.out_layer_coord_affine, or a warping-field layer beyond 2^40): not "
                    "representable by the split-fp16 kernels");
        net.has_colour = colour;
        net.sp_unscale = warp ? sp_unscale : 1.0f;
        return upload(net);
    };
    if (ctx->tmpl_set) {
        int rc = build(ctx->tmpl_only, false, false);
        if (rc) return rc;
        if (has_clr && (rc = build(ctx->tmpl_only_clr, false, true))) return rc;
    }
    if (ctx->warp_set && ctx->tmpl_set) {
        int rc = build(ctx->warp_tmpl, true, false);
        if (rc) return rc;
        // column folding takes the 64 feature columns out of conv1 / conv5 and leaves the xyz k-step: with a positional encoding in front of the warping
        // field (warp_pe > 0) the launches stay point by point (fused_mlp.hip launch_avatar)
        if (ctx->warp_pe == 0) { if ((rc = build(ctx->warp_tmpl_fold, true, false, true))) return rc; }
        else release(ctx->warp_tmpl_fold);
        if (has_clr && (rc = build(ctx->warp_tmpl_clr, true, true))) return rc;
    }
    return AVC_OK;
}

int pack_recon(avc_ctx *ctx, const avc_dense fc[4])
{
    static const int cout[4] = {512, 256, 128, 1}, cin[4] = {33, 545, 289, 128};
    std::vector<double> W[4], b[4];
    for (int i = 0; i < 4; ++i) {
        AVC_REQUIRE(fc[i].cout == cout[i] && fc[i].cin == cin[i], AVC_ERR_ARG,
                    "recon fc[%d]: expected (%d,%d), got (%d,%d)", i, cout[i], cin[i], fc[i].cout, fc[i].cin);
        int rc = effective(fc[i], nullptr, W[i], b[i]);
        if (rc) return rc;
    }
    {   // fc0 -> fc1 [x(512) | in33] -> fc2 [x(256) | in33] (LeakyReLU) -> fc3 (head): weight_norm gains beyond the fp16 range are rebalanced (see rebalance())
        std::vector<Node> g;
        for (int i = 0; i < 4; ++i) g.push_back(Node{&W[i], &b[i], cout[i], cin[i], i == 3, {}});
        for (int i = 0; i < 3; ++i) g[i].out.push_back(Link{i + 1, 0});
        AVC_REQUIRE(rebalance(g), AVC_ERR_ARG, "recon image_decoder: the output layer has weights beyond 3e4 in magnitude, also after the layers in front of it were "
                    "rebalanced: not representable by the split-fp16 kernels");
    }
    // Consumption order of recon_kernel (fused_mlp.hip): fc0 is evaluated in two halves of 256
    // channels so that fc1 (545 = [x(512) | in(33)] inputs, mlp.py:61) can accumulate over each half
    // while only 256 hidden channels are live in registers.
    Builder B(ctx->recon);
    auto rows = [&](const std::vector<double> &Wf, const std::vector<double> &bf, int cinf, int r0, int n,
                    std::vector<double> &Wo, std::vector<double> &bo) {
        Wo.assign(Wf.begin() + (size_t)r0 * cinf, Wf.begin() + (size_t)(r0 + n) * cinf);
        bo.assign(bf.begin() + r0, bf.begin() + r0 + n);
    };
    std::vector<double> Wa, ba, Wb, bb, zero256(256, 0.0);
    rows(W[0], b[0], 33, 0, 256, Wa, ba);
    rows(W[0], b[0], 33, 256, 256, Wb, bb);
    B.layer(Wa, ba, 256, 33, {seg_in33()}, 2, 16);                       // fc0 rows   0..255
    B.layer(W[1], b[1], 256, 545, {seg_d(16, 0)}, 8, 4);                 // fc1 over x[0..255]   (bias here)
    B.layer(Wb, bb, 256, 33, {seg_in33()}, 2, 16);                       // fc0 rows 256..511
    B.layer(W[1], zero256, 256, 545, {seg_d(16, 256), seg_in33(512)}, 8, 4);   // fc1 over x[256..511] and in(33)
    B.layer(W[2], b[2], 128, 289, {seg_d(16), seg_in33(256)}, 2, 16);    // fc2: [x(256) | in(33)]
    B.layer(W[3], b[3], 1, 128, {seg_d(8)}, 1, 16);                      // fc3
    AVC_REQUIRE(!B.overflow, AVC_ERR_ARG, "recon weights exceed 3e4 in magnitude: not representable by the split-fp16 kernel");
    if (int rc = upload(ctx->recon)) return rc;

    // The stream of recon_fold_kernel (grid launches): the 32 image-feature columns of fc0 / fc1 / fc2 leave the stream -- their fp32 weights go to
    // colw, [fc0 (512) | fc1 (256) | fc2 (128)] rows x 32 columns, followed by the 896 biases, for recon_column_terms_kernel -- and only the z column
    // stays a k-step.  Consumption order: fc0 rows 0..255 (8 tiles, one chunk); fc1 over x[0..255]; [fc0 rows 256..511 | fc1's z column] (one chunk of
    // two k-steps); fc1 over x[256..511]; fc2 on [x | z]; fc3.  The layer-table bias blocks are unused except fc3's (offset 5 * 256 + 128).
    Builder F(ctx->recon_fold);
    F.layer(Wa, ba, 256, 33, {seg_z33()}, 8, 16);
    F.layer(W[1], b[1], 256, 545, {seg_d(16, 0)}, 8, 4);
    F.layer(Wb, bb, 256, 33, {seg_z33()}, 8, 16);
    F.layer(W[1], zero256, 256, 545, {seg_z33(512)}, 8, 16);
    F.layer(W[1], zero256, 256, 545, {seg_d(16, 256)}, 8, 4);
    F.layer(W[2], b[2], 128, 289, {seg_d(16), seg_z33(256)}, 2, 16);
    F.layer(W[3], b[3], 1, 128, {seg_d(8)}, 1, 16);
    AVC_REQUIRE(!F.overflow, AVC_ERR_ARG, "recon weights exceed 3e4 in magnitude: not representable by the split-fp16 kernel");
    auto &cw = ctx->recon_fold.colw;
    cw.assign((size_t)896 * 32 + 896, 0.0f);
    const int row0[3] = {0, 512, 768}, nrow[3] = {512, 256, 128}, featc0[3] = {0, 512, 256};      // where [feat(32) | z] sits among each layer's input columns
    for (int l = 0; l < 3; ++l)
        for (int o = 0; o < nrow[l]; ++o) {
            for (int c = 0; c < 32; ++c) cw[(size_t)(row0[l] + o) * 32 + c] = (float)W[l][(size_t)o * cin[l] + featc0[l] + c];
            cw[(size_t)896 * 32 + row0[l] + o] = (float)b[l][o];
        }
    return upload(ctx->recon_fold);
}

}  // namespace avc
