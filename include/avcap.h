/* avcap.h -- C ABI of libavcap_hip.so, the MI355X (gfx950) hot path of AvatarCap's per-frame
 * volumetric reconstruction.
 *
 * The reference (lizhe00/AvatarCap) is pure Python; it has no FFI.  Its seam for this path is a
 * handful of Python call signatures.  Each entry point below names the
 * reference interface it replaces (file:line under /root/reference); INTEGRATION.md shows the
 * ctypes stub a maintainer would add on the reference side.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  `*_dev` pointers are device (HBM) pointers
 *     the caller owns (e.g. tensor.data_ptr()); all other pointers are host memory.
 *   - every call returns 0 on success, a negative avc_status otherwise; avc_last_error() returns a
 *     thread-local description of the last failure.
 *   - `stream` is a hipStream_t (NULL = default stream).  Calls are asynchronous w.r.t. the host
 *     unless stated otherwise.  No ownership is transferred.  One context per device; a context
 *     is thread-compatible (use it from one thread at a time) and serves ONE stream at a time: its
 *     bound feature maps, per-column tables and mesh scratch are shared by the calls made on it,
 *     so two queries of one context must not overlap on different streams (use one context per
 *     concurrent stream, or order the streams with events).
 *   - all arithmetic is float32 in / float32 out; indices are int32 (faces) / int64 (KNN).
 */
#ifndef AVCAP_H
#define AVCAP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct avc_ctx avc_ctx;
typedef void *avc_stream; /* hipStream_t */

enum avc_status {
    AVC_OK = 0,
    AVC_ERR_ARG = -1,      /* invalid argument (the reference would raise ValueError/TypeError) */
    AVC_ERR_STATE = -2,    /* weights / feature map not set (reference: AttributeError / ValueError) */
    AVC_ERR_HIP = -3,      /* a HIP runtime call failed */
    AVC_ERR_CAPACITY = -4, /* caller-provided output capacity too small; required sizes are reported */
    AVC_ERR_RANGE = -5     /* range check on (avc_set_range_check): a value left the range of the split-fp16 arithmetic */
};

const char *avc_last_error(void);
int avc_version(void);

int avc_ctx_create(int device, avc_ctx **ctx_out);
int avc_ctx_destroy(avc_ctx *ctx);

/* ---- weights -------------------------------------------------------------------------------
 * A Conv1d(k=1) layer exactly as the reference stores it in net.pt / recon_net.pt:
 * w is (cout, cin) row-major (state_dict `...weight` squeezed); b is (cout).  When g != NULL the
 * layer is weight-normed: w holds `weight_v`, g holds `weight_g` (cout) and the effective weight
 * is g * v / ||v||_2 per output channel (network/mlp.py:24,36).  Host pointers; copied. */
typedef struct { const float *w, *b, *g; int32_t cout, cin; } avc_dense;
/* BatchNorm1d in eval mode (running statistics), network/mlp.py:88-96 */
typedef struct { const float *gamma, *beta, *mean, *var; float eps; } avc_bn;

/* WarpingField.mlp (OffsetDecoder conv1..7 + bn1..7) and out_layer_coord_affine
 * (network/arch_avatar.py:100-105; network/mlp.py:75-112).  BatchNorm is folded at pack time.
 * pos_encoding = cfg['model']['warping_field']['pos_encoding'] = L, 0 .. 10 (network/arch_avatar.py:97-100): conv1 is (256, 3 + 6 L + 64), conv5
 * (256, 3 + 6 L + 64 + 256) -- the field's input is [get_embedder(L)(xyz) | pose_feat(64)] (:122,136; utils/net_util.py:40-55).  L = 0 is what
 * configs/example.yaml ships (67-wide checkpoints): grid launches are then column-folded.  With L > 0 the encoding of the raw point is evaluated in the
 * kernel (all ten octaves; the columns of octaves >= L pack as zeros) and every launch runs point by point.  L > 10 -> AVC_ERR_ARG. */
int avc_pack_warp_weights(avc_ctx *ctx, const avc_dense conv[7], const avc_bn bn[7],
                          const avc_dense *out_affine, int pos_encoding);

/* DoubleTNet shared_mlp (7 layers, res @4), geo_mlp (2), clr_mlp (3, may be NULL)
 * (network/arch_avatar.py:37-58).  pos_encoding = cfg['model']['cano_template']['pos_encoding'] = L, 0 .. 10 (:33-36): shared[0] is (256, 3 + 6 L),
 * shared[4] (256, 256 + 3 + 6 L) (res layer, mlp.py:61); 10 = the example's 63-wide checkpoints.  The kernels evaluate ten octaves whatever L is;
 * the columns of octaves >= L do not exist in the checkpoint and pack as zero weights (same speed for every L).  L > 10 -> AVC_ERR_ARG. */
int avc_pack_template_weights(avc_ctx *ctx, const avc_dense shared[7], const avc_dense geo[2],
                              const avc_dense *clr /* [3] or NULL */, int pos_encoding);

/* ReconNetwork.image_decoder: MLP [33,512,256,128,1], res @ [1,2], weight_norm, LeakyReLU(0.02),
 * sigmoid (network/arch_recon.py:19-39).  weight_norm is folded at pack time. */
int avc_pack_recon_weights(avc_ctx *ctx, const avc_dense fc[4]);

/* ---- per-frame feature maps ----------------------------------------------------------------
 * WarpingField.precompute_conv caches self.pose_feat_map = unet(smpl_pos_map)
 * (network/arch_avatar.py:109-111).  The U-Net itself stays on PyTorch-ROCm; this call hands its
 * (1,C=64,H,W) NCHW output to the context, which re-lays it out channel-last for the gather. */
int avc_set_pose_feat_map(avc_ctx *ctx, const float *map_nchw_dev, int C, int H, int W, avc_stream stream);
/* img_feat_map = HGFilter(cat(front,back))[-1], (1,C=32,H,W) (network/arch_recon.py:51-52) */
int avc_set_img_feat_map(avc_ctx *ctx, const float *map_nchw_dev, int C, int H, int W, avc_stream stream);

/* ---- the image encoder -------------------------------------------------------------------------
 * ReconNetwork.image_encoder = HGFilter(1, 4, 6, 32, 'group', 'no_down', False) (network/arch_recon.py:29;
 * network/HGFilters.py:124-219 HGFilter, :33-75 ConvBlock, :77-121 HourGlass), hand-written for gfx950
 * (csrc/conv_enc.hip): implicit-GEMM convolutions on split-fp16 MFMA with GroupNorm + ReLU applied while
 * the input tile is staged, GroupNorm statistics produced by the convolution before, avg_pool / bicubic
 * up-sampling as small fused kernels, the whole encoder replayed as one hipGraph.
 *
 * Weights exactly as the state_dict stores them (host pointers, copied at pack time):
 *   avc_conv2d     w (cout, cin, kh, kw) row-major, b (cout) or NULL
 *   avc_groupnorm  gamma, beta (channels); torch.nn.GroupNorm(groups, channels, eps)
 *   avc_convblock  conv1..3 (3x3, no bias), downsample[2] (1x1, no bias; w == NULL when in == out planes),
 *                  bn1..bn4 (bn4 is only read when the projection exists)
 *   avc_hgfilter   conv1 (7x7 s2 p3), bn1, conv2..4, the hourglass' 3 depth + 1 blocks in the order
 *                  b1_d, b2_d, b1_{d-1}, b2_{d-1}, ..., b1_1, b2_1, b2_plus_1, b3_1, ..., b3_d, then top_m_0,
 *                  conv_last0 (1x1), bn_end0, l0 (1x1).  Only stack == 1, 'group' norm, 'no_down' are on the path. */
typedef struct { const float *w, *b; int32_t cout, cin, kh, kw; } avc_conv2d;
typedef struct { const float *gamma, *beta; int32_t channels, groups; float eps; } avc_groupnorm;
typedef struct { avc_conv2d conv[3]; avc_conv2d downsample; avc_groupnorm bn[4]; } avc_convblock;
typedef struct {
    avc_conv2d conv1; avc_groupnorm bn1;
    avc_convblock conv2, conv3, conv4;
    int32_t depth; const avc_convblock *hourglass;
    avc_convblock top_m;
    avc_conv2d conv_last; avc_groupnorm bn_end; avc_conv2d l;
} avc_hgfilter;
int avc_hgfilter_pack(avc_ctx *ctx, const avc_hgfilter *net);
/* HGFilter.forward (network/HGFilters.py:176-219) for one image: image_dev (6, H, W) NCHW as the reference passes
 * cat(front_normal, back_normal) -> outputs[-1] = l0(...) as feat_out_dev (32, H1, W1) NCHW (or NULL) and normx
 * (the conv2 block's output) as normx_out_dev (128, H1, W1) (or NULL), H1 = (H - 1) / 2 + 1.  With
 * bind_img_feat_map != 0 the context's image feature map (avc_set_img_feat_map) becomes this frame's feature map without
 * the NCHW round trip.  H1 and W1 must be divisible by 2^depth (the reference adds up1 + up2 of equal size). */
int avc_hgfilter_forward(avc_ctx *ctx, const float *image_dev, int H, int W, float *feat_out_dev, float *normx_out_dev,
                         int bind_img_feat_map, avc_stream stream);

/* ---- the warping field's U-Net -------------------------------------------------------------------
 * WarpingField.unet = UnetNoCond7DS(input_nc 6, output_nc 64, nf 32, 'upconv') (network/arch_avatar.py:95; network/unets.py:169-229, blocks :10-60) on
 * the encoder's convolution kernel: every stride-2 convolution is a 3x3 convolution of a space-to-depth tensor its producer wrote in that form, every
 * transposed convolution a 3x3 convolution with 4 x parity-major outputs scattered by the epilogue, BatchNorm2d(affine=False, eval) folded into the
 * weights, the pre-activations (LeakyReLU(0.2) / ReLU) applied while the input is staged, torch.cat free (producers write their channel slice): 18
 * launches in one hipGraph.  The reference's quirk is kept: upconv3 is applied twice, upconv4 never (unets.py:213-214).
 *   avc_bn2d     running mean / var of a BatchNorm2d(affine=False); mean == NULL: the layer has no norm (conv1, conv7, upconvC7)
 *   avc_unet7ds  down[7]: conv1..7 weights (cout, cin, 4, 4), no bias; up[3]: upconv1..3 ConvTranspose2d weights as stored, (cin, cout, 4, 4) with
 *                avc_conv2d.cin / cout the module's in / out channels; upc[3]: upconvC5..C7 (cout, cin, 3, 3) + bias. */
typedef struct { const float *mean, *var; float eps; } avc_bn2d;
typedef struct {
    avc_conv2d down[7]; avc_bn2d down_bn[7];
    avc_conv2d up[3]; avc_bn2d up_bn[3];
    avc_conv2d upc[3]; avc_bn2d upc_bn[3];
} avc_unet7ds;
int avc_unet_pack(avc_ctx *ctx, const avc_unet7ds *net);
/* UnetNoCond7DS.forward for one position map: pos_map_dev (6, H, W) NCHW (batch['smpl_pos_map'][b]; H, W multiples of 128: 256 in the reference) ->
 * out_nchw_dev (64, H, W) or NULL; with bind_pose_feat_map != 0 the result becomes the context's pose feature map (avc_set_pose_feat_map) without the
 * NCHW round trip -- WarpingField.precompute_conv (network/arch_avatar.py:109-111). */
int avc_unet_forward(avc_ctx *ctx, const float *pos_map_dev, int H, int W, float *out_nchw_dev, int bind_pose_feat_map, avc_stream stream);

/* Test hook: the output of launch `launch` of the last avc_hgfilter_forward's plan (which = 0: the launch's own output, 1: the block output y
 * it adds its slice to; which = 2: launch `launch` of the last avc_unet_forward's plan -- the WHOLE destination tensor the launch wrote its channel
 * slice of) as NCHW; *C, *H, *W receive its shape (for a convolution *C also carries the tile configuration in its high bits).
 * out_nchw_dev may be NULL (shape query).  Returns 1 when the launch has no such output, a negative status on errors. */
int avc_hgfilter_debug_tensor(avc_ctx *ctx, int launch, int which, float *out_nchw_dev, int32_t *C, int32_t *H, int32_t *W, avc_stream stream);

/* GroupNorm of the image encoder, with the ReLU that follows every one of them fused in
 * (network/HGFilters.py:46-49,64-66,141,165,178,204; torch.nn.GroupNorm semantics: per (n, group) mean and
 * biased variance over (C/G, H, W), y = (x - mean) / sqrt(var + eps) * gamma[c] + beta[c]).
 * x_dev, y_dev (N, C, HW) contiguous float32 (y_dev may equal x_dev); gamma_dev, beta_dev (C) or NULL. */
int avc_group_norm(avc_ctx *ctx, const float *x_dev, int N, int C, int64_t HW, int G, const float *gamma_dev,
                   const float *beta_dev, float eps, int relu, float *y_dev, avc_stream stream);

/* ---- queries -------------------------------------------------------------------------------
 * OccupancyNet.query (network/arch_avatar.py:356-381) = WarpingField.query (:113-140) +
 * DoubleTNet.forward (:65-83) on cano_pts + offset, for n points (any n >= 0; the reference's
 * 262,144-point chunking is an activation-memory device it no longer needs).
 *   pts_dev     (n,3)  cano_pts
 *   center      (3)    batch['cano_smpl_center']
 *   occ_out_dev (n)    'cano_pts_ov'   raw geo[0] when occupancy_sigmoid == 0 (config.if_type=='sdf'),
 *                                      sigmoid(geo[0]) otherwise (arch_avatar.py:77-80)
 *   offset_out_dev (n,3) or NULL   'nonrigid_offset'
 *   rgba_out_dev (n,4) or NULL     sigmoid(clr_mlp) and relu(geo[1]) -- the `rgb, alpha` of
 *                                  DoubleTNet.forward (:75-76); needs clr weights. */
int avc_avatar_query(avc_ctx *ctx, const float *pts_dev, int64_t n, const float center[3],
                     int occupancy_sigmoid, float *occ_out_dev, float *offset_out_dev,
                     float *rgba_out_dev, avc_stream stream);

/* The same query on the dense canonical grid of AvatarCapDataset.generate_volume_points (dataset/avatarcap_dataset.py:312-326)
 * WITHOUT materialising the (N,3) point array (201 MB at 256^3): point i = x*Ry*Rz + y*Rz + z has the coordinates
 * (axis_x[x], axis_y[y], axis_z[z]), where axis_a[k] = linspace(0,1,R_a)[k] * (b1_a - b0_a) + b0_a in float32 -- the caller
 * builds the three tables with the reference's own arithmetic (avatarcap_amd/grid.py: volume_axes), so the points are
 * bit-identical to the reference's.  Outputs as above; offset_out_dev may be NULL (main.py:360-364 never reads it).
 * When R_z is a multiple of 128 the launch is column-folded: the 64 pose-feature columns of the warping field's conv1 / conv5 enter as one fp32
 * vector per (x, y) column instead of per point (same algebra, other rounding: a few 1e-6 from avc_avatar_query on the same points, which an
 * unfolded launch -- any other R_z, or avc_set_option "column_fold" 0 -- reproduces bit for bit).  Scratch: 2 KB per column in the context. */
int avc_avatar_query_grid(avc_ctx *ctx, const float *axis_x_dev, const float *axis_y_dev, const float *axis_z_dev,
                          const int32_t res[3], const float center[3], int occupancy_sigmoid,
                          float *occ_out_dev, float *offset_out_dev, avc_stream stream);

/* The query on a SUBSET of that grid -- the valid band of the reference's test loop (avatarcap_dataset.py:114-116: infer_pts = vol_pts[infer_pts_flag],
 * queried at main.py:360) -- given as `n` flat grid indices (index_dev[k] = x*Ry*Rz + y*Rz + z, int32, any order): output k belongs to grid point
 * index_dev[k], coordinates as in avc_avatar_query_grid.  The launch is column-folded like a dense one (each point takes its column's vector), so it agrees
 * with avc_avatar_query on the same points to a few 1e-6 (bit for bit with "column_fold" 0), and the 12 bytes per point of coordinates are never read. */
int avc_avatar_query_grid_subset(avc_ctx *ctx, const float *axis_x_dev, const float *axis_y_dev, const float *axis_z_dev,
                                 const int32_t res[3], const int32_t *index_dev, int64_t n, const float center[3],
                                 int occupancy_sigmoid, float *occ_out_dev, float *offset_out_dev, avc_stream stream);

/* Numeric range.  The fused queries evaluate every float32 product as three fp16 x fp16 products with float32 accumulation
 * (DESIGN.md section 2): weights are split on the host -- a weight above 3e4 in magnitude does not fit the split; the packers bring
 * such layers inside the range exactly (powers of two): an out-of-range output row of a ReLU / LeakyReLU / linear layer (cano_template, the recon decoder:
 * a weight_norm gain) is scaled down to the layer's ordinary magnitude and its consumers' matching input column up, which no kernel notices; the
 * warping field's seven Conv1d + BatchNorm1d + Softplus layers (a BatchNorm fold with a small running variance) take ONE common scale that a separate
 * build of the query kernels undoes in front of the Softplus (chosen automatically; a checkpoint inside the range runs the default kernels unchanged);
 * an output layer (geo_mlp / clr_mlp / out_layer_coord_affine / the decoder's last fc) that is still out of range is refused with AVC_ERR_ARG --
 * and every sampled feature, positional-encoding value and post-activation value is split on the fly, which requires
 * |value| <= 65504 (Softplus layers carry y / ln 2).  The reference's float32 path has no such bound (network/mlp.py:90-110).
 * avc_set_range_check(ctx, 1) switches the queries of this context to a build of the kernels that tracks the largest
 * magnitude entering a split; a query then synchronises its stream and returns AVC_ERR_RANGE when the bound was exceeded
 * (its outputs are not valid).  Off by default: the check costs one VALU instruction per value pair and the synchronisation.
 * The image encoder (avc_hgfilter_forward) uses the same arithmetic: its kernels always track the largest staged value -- normalised
 * activations are staged times 16, so the bound is 4094 there, 65504 for the one raw input (conv_last0's) -- and with the check on the
 * call synchronises and returns AVC_ERR_RANGE likewise.
 * `main.py -m test` on checkpoints read from disk runs the first frame of every rank with the check on; a trip ends the run. */
int avc_set_range_check(avc_ctx *ctx, int enabled);

/* DoubleTNet.forward alone on given points (pts_space == 'temp', arch_avatar.py:216-219) */
int avc_template_query(avc_ctx *ctx, const float *pts_dev, int64_t n, int occupancy_sigmoid,
                       float *occ_out_dev, float *rgba_out_dev, avc_stream stream);

/* ---- the colour path ---------------------------------------------------------------------------------
 * NerfRenderer.render(batch, pts_space='cano') in eval mode followed by raw2outputs (network/arch_avatar.py:240-349 with the 'cano' branch of
 * GeoTexAvatar.forward :206-237; utils/nerf_util.py:185-212) -- what main.py:464-477 runs once per frame to colour the avatar's vertices:
 *   z = near (1 - t) + far t, t = t_vals_dev (n_samples) or, when NULL, linspace(0, 1, n_samples) by ATen's element formula (the reference takes
 *   torch.linspace on the host, whose vectorised last bits depend on the host: pass that tensor to follow it bit for bit); pts = ray_o + ray_d z; for rays with depth > 1e-6 (depth_dev != NULL) near / far are
 *   depth - near_dist / depth + far_dist first (:289-291);
 *   (occ, offset, rgba) = the fused avatar query at pts with the colour head (the pose feature map and the weights bound to the context);
 *   sigma = rgba.w, zeroed where pts + offset is not strictly inside `bounds` (lo xyz, hi xyz) or the nearest of the n_smpl canonical SMPL vertices is
 *   0.08 or farther (:208-209, :222-226); alpha = 1 - exp(-sigma dist), dist = z[s + 1] - z[s] (the last sample repeats its predecessor's);
 *   weights = alpha * cumprod([1, 1 - alpha + 1e-10])[:-1]; rgb_map = sum w rgb; depth_map = sum w z; acc_map = sum w; disp_map = 1 / max(1e-10, depth / acc).
 * Outputs (each may be NULL): rgb_map (n_rays, 3) in the network's channel order, acc / depth / disp maps (n_rays), weights (n_rays, n_samples),
 * raw (n_rays n_samples, 4) = [rgb, alpha].  The reference's 2048-ray chunking (:330) bounds its activation memory and does not change results; this
 * entry walks ~4 M samples at a time out of a context-owned scratch. */
int avc_render_rays_cano(avc_ctx *ctx, const float *ray_o_dev, const float *ray_d_dev, const float *near_dev, const float *far_dev, const float *depth_dev,
                         float near_dist, float far_dist, const float *t_vals_dev, int64_t n_rays, int n_samples, const float center[3], const float bounds[6],
                         const float *cano_smpl_v_dev, int32_t n_smpl, int occupancy_sigmoid, float *rgb_map_dev, float *acc_map_dev, float *depth_map_dev,
                         float *disp_map_dev, float *weights_dev, float *raw_dev, avc_stream stream);
/* CanoBlendWeightVolume.forward (network/arch_avatar.py:143-165): F.grid_sample(volume, (2 pts - 1)[..., [2, 1, 0]], padding_mode='border',
 * align_corners=True) of the canonical blend-weight volume.  vol_xyzc_dev (X, Y, Z, channels) channel-last float32 -- the layout of the .npy the
 * reference loads (:145) --, pts01_dev (n, 3) in [0, 1]^3 (x along X), out_dev (n, channels). */
int avc_blend_weight_sample(avc_ctx *ctx, const float *vol_xyzc_dev, const int32_t res[3], int channels, const float *pts01_dev, int64_t n, float *out_dev,
                            avc_stream stream);

/* ReconNetwork.infer's per-point part (network/arch_recon.py:55-73): bilinear 32-ch sample at
 * (x-cx, -(y-cy)), z = p.z-cz, decoder, sigmoid.  out_dev (n). */
int avc_recon_query(avc_ctx *ctx, const float *pts_dev, int64_t n, const float center[3],
                    float *out_dev, avc_stream stream);

/* The same decoder on the dense canonical grid / on a subset of it given by flat grid indices, without the (N,3) point array -- axis tables, point
 * order and `index_dev` exactly as for avc_avatar_query_grid / avc_avatar_query_grid_subset (ReconNetwork.infer is called with the same
 * items['cano_pts'] as the avatar query: main.py:440, arch_recon.py:48).  The decoder's input is [img_feat(32) sampled at (x, y) | z]
 * (arch_recon.py:63-70) and feeds fc0, fc1 and fc2 (res_layers).  A DENSE grid whose R_z is a multiple of 128 is column-folded: what the three layers do
 * with the 32 feature channels is one fp32 vector of 896 values per (x, y) column (3.5 KB per column of scratch in the context), which enters the
 * accumulators through free K slots of the z k-step; only the z column stays a per-point product: 14 % fewer MFMAs, no per-point gather, 13.2 instead of
 * 17.6 ms at 256^3.  A SUBSET (the valid band) is column-folded whatever its shape (round 5): the 32 points of a wavefront are cut into runs of equal
 * adjacent (x, y) columns; the first two runs ride the z k-step like the single column of a dense launch (one run per lane half), further runs -- rare
 * in a band, whose runs along z are long -- are added to the zeroed accumulators first.  Same algebra as avc_recon_query, other rounding: ~1e-6 from it;
 * bit-reproducible from call to call, and a point's value depends on which points share its wavefront only through the rounding of ONE fp32 addition
 * (measured: tests/test_gpu_query.py::test_recon_grid_subset_query).  Every other launch -- a dense grid with another R_z, or avc_set_option
 * "column_fold" 0 -- runs the point-by-point kernel on the generated points: bit-identical to avc_recon_query. */
int avc_recon_query_grid(avc_ctx *ctx, const float *axis_x_dev, const float *axis_y_dev, const float *axis_z_dev,
                         const int32_t res[3], const float center[3], float *out_dev, avc_stream stream);
int avc_recon_query_grid_subset(avc_ctx *ctx, const float *axis_x_dev, const float *axis_y_dev, const float *axis_z_dev,
                                const int32_t res[3], const int32_t *index_dev, int64_t n, const float center[3],
                                float *out_dev, avc_stream stream);

/* occ_volume[valid] = values; occ_volume[~valid] = fill  (main.py:362-363, 442-443).
 * valid_dev (N) uint8/bool; values_dev (n_valid) in flat order; fill_dev (N - n_valid). */
int avc_scatter_volume(avc_ctx *ctx, const uint8_t *valid_dev, int64_t N, const float *values_dev,
                       const float *fill_dev, float *volume_out_dev, avc_stream stream);

/* ---- meshing -------------------------------------------------------------------------------
 * recon_util.recon_mesh (utils/recon_util.py:51-70) with the volume kept on the device:
 * marching cubes at iso = scikit-image's Lewiner algorithm restated (skimage.measure.marching_cubes, :64): vertices, faces and their
 * numbering are those of the library call (pinned on outputs of the real library), vertices = index*voxel + b0 + voxel/2 (:62,65), normals = -normalize(trilinear
 * sample of the Sobel gradient volume) (:9-48,66-68), faces flipped [2,1,0] (:69).
 *   vol_dev (X,Y,Z) float32, res = {X,Y,Z}, bounds = {b0x,b0y,b0z,b1x,b1y,b1z}
 *   verts_out_dev (cap_v,3) f32, normals_out_dev (cap_v,3) f32 or NULL, faces_out_dev (cap_f,3) i32
 *   counts_out[2] (HOST) = {V, F}.  Synchronises the stream once (the counts decide the emit
 *   launch sizes).  Returns AVC_ERR_CAPACITY (with counts filled) if V > cap_v or F > cap_f.
 * Output order is the library's: cells in (axis 0, axis 1, axis 2) order, a vertex is numbered the first time a triangle refers to it.
 * counts_out = {0, 0} when no surface crosses iso (the library raises there; the Python mirror restates its exceptions). */
int avc_recon_mesh(avc_ctx *ctx, const float *vol_dev, const int32_t res[3], const float bounds[6],
                   float iso, float *verts_out_dev, float *normals_out_dev, int32_t *faces_out_dev,
                   int64_t cap_v, int64_t cap_f, int64_t counts_out[2], avc_stream stream);

/* ---- canonical normal maps -------------------------------------------------------------------
 * visualize_util.render_cano_mesh with the 'vertex_attribute' renderer (utils/visualize_util.py:11-52,
 * utils/renderer.py:10-29,316-323,432-451): orthographic front and back rasters of the mesh translated
 * by -center, x,y in [-1,1] -> size x size pixels (row 0 at y=+1), depth-tested, back faces culled,
 * per-vertex attribute (the canonical normal) interpolated, 0 background; the back map is already
 * mirrored so that both maps are pixel-aligned.  Replaces the reference's OpenGL context + read-back;

 * coverage identical to, values within 5e-5 of a real OpenGL implementation (Mesa llvmpipe, tests/golden/gl_golden.npz).
 *   verts_dev, attrs_dev (nv,3) f32; faces_dev (nf,3) i32; front_out_dev, back_out_dev (size,size,3) f32 */
int avc_render_cano_maps(avc_ctx *ctx, const float *verts_dev, const float *attrs_dev, int64_t nv,
                         const int32_t *faces_dev, int64_t nf, const float center[3], int size,
                         float *front_out_dev, float *back_out_dev, avc_stream stream);

/* Renderer.render with the 'position' / 'vertex_attribute' shaders for an arbitrary model-view-projection matrix
 * (utils/renderer.py:10-51,326-451), as normal_fusion.canonicalize_normal_map uses it with
 * gl_perspective_projection_matrix to obtain the posed mesh's position map (normal_fusion/normal_fusion.py:14-20).
 * mvp is row-major (the reference uploads it with transpose = GL_TRUE); attrs_dev == NULL renders the positions.
 * out_dev (height, width, 4) f32 RGBA = (perspective-correct attribute, 1), background 0, row 0 at ndc.y = +1 (the
 * reference flips the read-back, renderer.py:449).  Back faces culled, GL_LESS depth test on ndc.z in [-1, 1];
 * triangles with a vertex at w <= 0 are dropped (not clipped).  Pinned on Mesa llvmpipe like the call above. */
int avc_render_mesh(avc_ctx *ctx, const float *verts_dev, const float *attrs_dev, int64_t nv, const int32_t *faces_dev,
                    int64_t nf, const float mvp[16], int width, int height, float *out_dev, avc_stream stream);

/* ---- canonical normal fusion (normal_fusion/normal_fusion.py) ---------------------------------------------------------
 * The oracle (oracle/normal_fusion_oracle.py) is pinned against a run of the reference's own module with stand-ins for its
 * OpenCV / pytorch3d / OpenGL calls (tests/golden/make_golden_fusion.py) and against torch.autograd.
 *
 * canonicalize_normal_map's per-vertex part (:27-62): project each posed vertex with mv (row-major 4x4, world -> camera:
 * x right, y down, z forward) and the pinhole (fx, fy, cx, cy); sample the position map (the 'position' render of the
 * posed mesh, avc_render_mesh) and the observed normal map at the nearest pixel (grid_sample nearest / border /
 * align_corners=True); a vertex is visible if the rendered point lies within 0.05 of it; the observed normal has y, z
 * negated, is rotated by inv(mv)[:3,:3] and by the inverse of the vertex's cano2live matrix (vert_mats, (nv,4,4), upper
 * left 3x3), and is zeroed when the vertex is occluded or unobserved.  A singular vertex matrix gives 0 (the reference's
 * torch.linalg.inv raises).  pos_map_dev (H,W,4), normal_map_dev (H,W,3), normals_out_dev (nv,3). */
int avc_canonicalize_normals(avc_ctx *ctx, const float *live_v_dev, const float *vert_mats_dev, int64_t nv,
                             const float *pos_map_dev, const float *normal_map_dev, int height, int width,
                             const float mv[16], float fx, float fy, float cx, float cy, float *normals_out_dev,
                             avc_stream stream);

/* merge_normal_images (:89-155): src = avatar normal map, tar = image-observed canonical normal map, both (H,W,3);
 * 3x3 erosion x3 and L1 distance transform of the observation mask, iter_num iterations (the first half: Adam(lr 1e-2)
 * on a 64x64 axis-angle rotation grid, the second half: Adam(lr 1e-1) on the normals) of
 * mean|R(x) src - tar|^2 over observed pixels + the 8-neighbour smoothness of the grid, distance-transform blend with
 * the input, face rectangle [neck_y - 90, neck_y) x [neck_x - 35, neck_x + 35) (Python slice semantics) reset to the
 * input.  out_dev (H,W,3), may not alias src_dev's scratch but may equal src_dev. */
int avc_merge_normal_images(avc_ctx *ctx, const float *src_dev, const float *tar_dev, int height, int width, int iter_num,
                            int neck_x, int neck_y, float *out_dev, avc_stream stream);

/* merge_normal_images_cover (:158-167): out = tar where |tar| > 1e-6, src elsewhere; (npix,3) each */
int avc_merge_normal_images_cover(avc_ctx *ctx, const float *src_dev, const float *tar_dev, int64_t npix, float *out_dev,
                                  avc_stream stream);

/* ---- SMPL utilities ------------------------------------------------------------------------
 * K nearest of nr reference points per query, squared L2 ascending, ties -> lower index
 * (pytorch3d.ops.knn_points as used at utils/smpl_util.py:33, dataset/avatarcap_dataset.py:114,
 * network/arch_avatar.py:190,208).  K <= 8.  d2_out_dev (nq,K) f32, idx_out_dev (nq,K) i64. */
int avc_knn(avc_ctx *ctx, const float *query_dev, int64_t nq, const float *ref_dev, int32_t nr, int K,
            float *d2_out_dev, int64_t *idx_out_dev, avc_stream stream);

/* SmplUtil.calculate_lbs (utils/smpl_util.py:24-39): KNN-4 to the canonical SMPL vertices,
 * w = exp(-d2 / (2*0.05^2)), normalised (+1e-16), blended 24-wide skin weights.
 * cano_v_dev (nv,3), skin_w_dev (nv,24) -> lbs_out_dev (n,24) */
int avc_calculate_lbs(avc_ctx *ctx, const float *pts_dev, int64_t n, const float *cano_v_dev,
                      const float *skin_w_dev, int32_t nv, float *lbs_out_dev, avc_stream stream);

/* SmplUtil.set_cano_smpl_vertices (utils/smpl_util.py:21; main.py:335: once per sequence) and calculate_lbs against what it set (:24-39).
 * avc_lbs_prepare copies the (nv,3) vertices into the context and builds, once: the uniform grid of the KNN search and, for every 2 cm cell of the space
 * within 0.14 m of the vertices (avc_set_option "lbs_reach_mm"), the list of the vertices that can be among the four nearest of ANY point of the cell (all v with
 * |v - centre| <= d_4(centre) + cell diagonal).  avc_calculate_lbs_bound then reads one list per point instead of searching: bit-identical to
 * avc_calculate_lbs on the same vertices (same squared distances, same (distance, index) order; points without a list -- farther than that from the body,
 * outside the cells' box -- take the grid search), and without the per-call grid construction.  Host-synchronous (sizes its tables); a second call replaces
 * the first.  A vertex set whose lists would exceed 2^28 entries (thousands of coincident vertices: every cell lists every vertex) gets none: the search serves.
 * avc_calculate_lbs_bound before avc_lbs_prepare -> AVC_ERR_STATE ("Canonical smpl vertices are invalid!", smpl_util.py:31).
 * avc_lbs_bound_stats: out = {vertices bound, cells, list entries, 1 if lists exist}. */
int avc_lbs_prepare(avc_ctx *ctx, const float *cano_v_dev, int32_t nv, avc_stream stream);
int avc_calculate_lbs_bound(avc_ctx *ctx, const float *pts_dev, int64_t n, const float *skin_w_dev, float *lbs_out_dev, avc_stream stream);
int avc_lbs_bound_stats(avc_ctx *ctx, int64_t out[4]);

/* SmplUtil.skinning / skinning_normal (utils/smpl_util.py:58-81): M = sum_j lbs_j * J_j;
 * p' = M[:3,:3] p + M[:3,3]; n' = M[:3,:3] n (no renormalisation).
 * jnt_mats_dev (24,4,4).  nrm_dev/nrm_out_dev/mats_out_dev may be NULL.  mats_out_dev (n,4,4). */
int avc_skinning(avc_ctx *ctx, const float *pts_dev, const float *nrm_dev, int64_t n,
                 const float *lbs_dev, const float *jnt_mats_dev, float *pts_out_dev,
                 float *nrm_out_dev, float *mats_out_dev, avc_stream stream);

/* main.py:385-389 in one launch: lbs = calculate_lbs(v) (smpl_util.py:24-39, against the vertices avc_lbs_prepare bound), live_v / vert_mats = skinning(v, lbs,
 * jnt_mats, True) (:58-74), live_n = skinning_normal(n, lbs, jnt_mats) (:76-81).  The same operations in the same order as avc_calculate_lbs_bound followed by
 * avc_skinning -- bit-identical outputs --, but a vertex's 24 blend weights stay in registers: lbs_out_dev (n,24) is written only when it is not NULL (the frame
 * loop never reads it).  nrm_dev / nrm_out_dev, mats_out_dev (n,4,4), pts_out_dev and lbs_out_dev may be NULL (at least one output must not be).
 * Before avc_lbs_prepare -> AVC_ERR_STATE as avc_calculate_lbs_bound. */
int avc_lbs_skin_bound(avc_ctx *ctx, const float *pts_dev, const float *nrm_dev, int64_t n, const float *skin_w_dev, const float *jnt_mats_dev,
                       float *lbs_out_dev, float *pts_out_dev, float *nrm_out_dev, float *mats_out_dev, avc_stream stream);

/* ---- measurement hook ------------------------------------------------------------------------
 * Average device time (ms, HIP events on the launch stream) of the fused query kernel over the
 * launches since the last reset; used by bench.py for roofline.achieved.  which: 0 avatar, 1 recon. */
int avc_timing_enable(avc_ctx *ctx, int enable);
int avc_timing_read(avc_ctx *ctx, int which, double *avg_ms_out, int64_t *launches_out, int reset);
/* Shader cycles (s_memtime, workgroup 0 from its first tile to its last) of the same timed launches, averaged over the (at most 64) launches
 * since the last call; synchronises the device.  cycles / ms = the shader clock the power manager held DURING the launch (bench.py's
 * `roofline.clock_mhz`). */
int avc_timing_read_cycles(avc_ctx *ctx, int which, double *avg_cycles_out, int64_t *launches_out);

/* ---- switches of a context --------------------------------------------------------------------
 * No entry point reads the environment: avc_ctx_create reads the defaults ONCE (AVC_NO_FOLD, AVC_MLP_BLOCKS, AVC_KNN_BRUTE / AVC_KNN_PATH,
 * AVC_FUSION_NO_GRAPH), this call changes them afterwards.  Test / A-B switches; none changes what a caller may rely on:
 *   "column_fold"  1 (default) | 0   avc_avatar_query_grid[_subset] / avc_recon_query_grid[_subset] without column folding: bit-identical to the
 *                                    point queries instead of ~1e-6 from them
 *   "mlp_blocks"   0 (default: one persistent workgroup per CU) | n
 *   "knn_search"   0 (default: per wave) | 1 per-lane grid search | 2 cooperative grid search | 3 exhaustive scan -- all four return the same bits
 *                  (with 1, 2 or 3 avc_calculate_lbs_bound leaves its candidate lists alone and searches that way too)
 *   "lbs_reach_mm" 140 (default) | 0 .. 1000   read by avc_lbs_prepare: cells whose centre lies within this distance of its 4th nearest bound vertex
 *                  get a candidate list (0: no lists, avc_calculate_lbs_bound searches the grid); same bits whatever the value.  140 covers the valid band
 *                  (where a frame's surface lies); 1000 covers a whole canonical volume (1.7 GB of lists, 15 ms once per sequence) for meshes that fill it
 *   "fusion_graph" 1 (default: the fusion iterations replay a hipGraph) | 0 plain launches
 *   "enc_graph"    1 (default: avc_hgfilter_forward replays a hipGraph) | 0 plain launches -- same kernels, same bits
 *   "enc_ksplit"   1 (default: a convolution that would run on fewer than half the CUs splits its input channels over several workgroups per
 *                  tile, whose partial sums the last to arrive adds in a fixed order) | 0 -- deterministic either way; the two differ by fp32 rounding
 *   "mc_walk"      1 (default: marching cubes classifies volumes whose 1024-point tiles are whole x rows of one z plane -- 256^3, 512^3, 384 x 384 x 128 --
 *                  with a pass that walks z inside a workgroup: a quarter of the cache requests) | 0 (the general classify pass for every volume); same results
 *   "enc_fork"     1 (default: the hourglass' upper branches run on a second stream -- parallel branches of the hipGraph -- beside the lower ones) |
 *                  0 one stream -- same kernels, same bits
 *   "enc_occ2"     1 (default: an encoder convolution whose launch has at least two half-height workgroups per CU runs them two per CU, each with half
 *                  the LDS, so that one workgroup's fetch and output bursts fall into the other's matrix work) | 0 one workgroup per CU; the two cut the
 *                  image into different tiles, so GroupNorm's fp32 partial sums -- and with them the outputs -- differ by rounding (~1e-7) */
int avc_set_option(avc_ctx *ctx, const char *name, int value);

#ifdef __cplusplus
}
#endif
#endif /* AVCAP_H */
