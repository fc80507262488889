// Register/fragment layouts of the fused MLP kernels, shared by the device code (fused_mlp.hip)
// and the host-side weight packer (pack.cpp).  Everything here is plain constexpr arithmetic.
//
// One wave owns 32 points.  Every GEMM is D[ch][pt] = sum_k W[ch][k] * X[k][pt] on
// v_mfma_f32_32x32x16_f16, i.e. A = weights (rows = output channels), B = activations
// (columns = points).  Lane l = (j = l & 31, h = l >> 5):
//   B operand of k-step ks : 8 halves  X[slot = ks*16 + h*8 + e][pt j],  e = 0..7
//   A operand of (tile t, k-step ks): 8 halves W[row = 32 t + j][slot = ks*16 + h*8 + e]
//   D tile t: 16 floats, reg r -> channel 32 t + d_row(r, h), point j
// Because a lane's 16 outputs of tile t are exactly the 2 x 8 elements it must supply as B operand
// for k-steps 2t and 2t+1 of the next layer, activations never leave registers: the *weights* of the
// next layer are stored with their K axis permuted to "slot" order (slot_channel_d below).
#pragma once

namespace avc {
namespace layout {

// D-tile register r (0..15) of lane-half h -> row inside the 32-row tile (CDNA4 32x32 C/D map)
constexpr int d_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// K slot -> channel, for an activation that was produced as D tiles by the previous layer
constexpr int slot_channel_d(int slot)
{
    const int ks = slot >> 4, h = (slot >> 3) & 1, e = slot & 7;
    return 32 * (ks >> 1) + d_row((ks & 1) * 8 + e, h);
}

// ---- WarpingField input: [xyz(3) | pose_feat(64)] = 67 columns (arch_avatar.py:136), 5 k-steps.
// k-steps 0..3 carry the 64 sampled channels (lane-half h gathers channels 32h..32h+31, so a lane
// reads 128 contiguous bytes per texel of the channel-last map); k-step 4 carries xyz in h = 0.
constexpr int IN67_KS = 5;
constexpr int in67_column(int slot)   // -> column of conv1 / the [input | x4] part of conv5, or -1
{
    const int ks = slot >> 4, h = (slot >> 3) & 1, e = slot & 7;
    if (ks < 4) return 3 + h * 32 + ks * 8 + e;
    return (h == 0 && e < 3) ? e : -1;
}

// ---- DoubleTNet input: NeRF positional encoding, 63 columns (net_util.py:37), 4 k-steps.
// Argument a = 3 f + c (frequency 2^f, coordinate c), a = 0..29.  Lane-half h evaluates arguments
// 15 h .. 15 h + 14 with one sincos each: local slot m = ks*8 + e in 0..31,
//   m = 2 i, 2 i + 1 (i < 15): sin, cos of argument 15 h + i   (coordinate i % 3, frequency 5 h + i / 3)
//   m = 30, 31:                 h = 0: x, y      h = 1: z, (pad)
constexpr int PE_KS = 4;
constexpr int pe_column(int slot)     // -> column of the reference's 63-wide embedding, or -1
{
    const int ks = slot >> 4, h = (slot >> 3) & 1, e = slot & 7;
    const int m = ks * 8 + e;
    if (m < 30) {
        const int a = 15 * h + (m >> 1), f = a / 3, c = a % 3;
        return 3 + 6 * f + 3 * (m & 1) + c;
    }
    if (m == 30) return h == 0 ? 0 : 2;
    return h == 0 ? 1 : -1;
}

// ---- WarpingField input with model.warping_field.pos_encoding = L > 0: [posenc(xyz) (3 + 6 L) | pose_feat(64)] (arch_avatar.py:122,136), 8 k-steps:
// k-steps 0..3 the 64 sampled channels as above, k-steps 4..7 the positional encoding of the RAW point in the PE layout (all ten octaves are
// evaluated; the columns of octaves >= L do not exist in the weights and pack as zeros).  L = 0 is the 5-k-step layout above (xyz alone).
constexpr int INPE_KS = 4 + PE_KS;
constexpr int inpe_column(int slot, int L)   // -> column of conv1 / the [input | x4] part of conv5, or -1
{
    const int ks = slot >> 4, h = (slot >> 3) & 1, e = slot & 7;
    if (ks < 4) return 3 + 6 * L + h * 32 + ks * 8 + e;
    const int c = pe_column(slot - 64);
    return c >= 0 && c < 3 + 6 * L ? c : -1;
}
// the template's embedding with model.cano_template.pos_encoding = L <= 10: the first 3 + 6 L of the 63 columns
constexpr int pe_column_l(int slot, int L) { const int c = pe_column(slot); return c >= 0 && c < 3 + 6 * L ? c : -1; }

// ---- ReconNetwork decoder input: [img_feat(32) | z] = 33 columns (arch_recon.py:70), 3 k-steps.
// k-steps 0..1: lane-half h gathers channels 16h..16h+15; k-step 2: z in (h = 0, e = 0).
constexpr int IN33_KS = 3;
constexpr int in33_column(int slot)
{
    const int ks = slot >> 4, h = (slot >> 3) & 1, e = slot & 7;
    if (ks < 2) return h * 16 + ks * 8 + e;
    return (h == 0 && e == 0) ? 32 : -1;
}

constexpr int UNIT_BYTES = 2048;        // [hi 1 KiB | lo 1 KiB]
constexpr int SLOT_BYTES = 65536;       // one LDS ring slot = the largest chunk
}  // namespace layout
}  // namespace avc
