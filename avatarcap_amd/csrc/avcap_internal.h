// Internal declarations shared by the translation units of libavcap_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/avcap.h"

namespace avc {

void set_error(const char *fmt, ...);
#define AVC_HIP(call)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            avc::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return AVC_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)
#define AVC_REQUIRE(cond, status, ...)                                                          \
    do {                                                                                        \
        if (!(cond)) { avc::set_error(__VA_ARGS__); return status; }                            \
    } while (0)

// ---- fused-MLP weight stream -----------------------------------------------------------------
// A network is packed into one contiguous byte stream of *chunks*, in exactly the order the kernel
// consumes them.  A chunk is a sequence of 2 KiB units [hi 1 KiB | lo 1 KiB]; each 1 KiB block is
// one MFMA A-operand fragment (64 lanes x 8 halves, lane-linear), so a wave's ds_read_b128 of it
// is conflict-free.  See fused_mlp.hip for the layouts.
struct ChunkDesc { uint32_t offset, bytes; };

struct PackedNet {
    std::vector<uint8_t> stream;      // host copy of the chunk stream
    std::vector<ChunkDesc> chunks;
    std::vector<float> bias;          // per-row bias, padded to the tile grid
    std::vector<float> colw;          // column-folded stream only: [conv1 | conv5][256][64] fp32 weights of the 64 pose-feature columns (fused_mlp.hip)
    float *d_colw = nullptr;
    void *d_stream = nullptr;
    ChunkDesc *d_chunks = nullptr;
    float *d_bias = nullptr;
    bool ready = false;
    bool has_colour = false;
    float sp_unscale = 1.0f;          // 2^-s: the Softplus layers (the warping field's conv1..7) were packed times 2^s and need the scaled kernels (fused_mlp.hip AVC_LAYER_SCALE)
};

struct Timing { bool enabled = false; double total_ms[2] = {0, 0}; int64_t launches[2] = {0, 0};
                std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[2];
                // shader-clock stamps of timed launches: workgroup 0 writes (s_memtime at entry, at exit) into slot `launch % CLK_SLOTS`
                static constexpr int CLK_SLOTS = 64;
                long long *clk_dev[2] = {nullptr, nullptr}; int64_t clk_count[2] = {0, 0}; };

// Switches of a context (avc_set_option).  Their defaults are read from the environment ONCE, by avc_ctx_create; no entry point reads the
// environment afterwards.
struct Options {
    int column_fold = 1;      // dense / band launches of the avatar query take conv1 / conv5's pose-feature part per (x, y) column (AVC_NO_FOLD=1 -> 0)
    int mlp_blocks = 0;       // persistent workgroups of the fused queries; 0 = one per CU (AVC_MLP_BLOCKS)
    int knn_search = 0;       // 0 automatic, 1 per-lane grid search, 2 cooperative grid search, 3 exhaustive scan (AVC_KNN_PATH=lane|wave, AVC_KNN_BRUTE=1)
    int fusion_graph = 1;     // normal-fusion iterations replayed as a hipGraph (AVC_FUSION_NO_GRAPH=1 -> 0)
    int enc_graph = 1;        // the image encoder's launches replayed as a hipGraph
    int enc_ksplit = 1;       // convolutions with few workgroups split K over several (partial sums in HBM, added in a fixed order by the last to arrive)
    int enc_fork = 1;         // the hourglass' upper branches (b1_k) run on a second stream beside the lower ones
    int enc_occ2 = 1;         // convolutions with at least two half-size workgroups per CU run two per CU (conv_enc.hip ConvGeo: OCC2)
    int lbs_reach_mm = 140;   // avc_lbs_prepare: cells whose centre lies within this distance of its 4th nearest vertex get a candidate list (0: no lists)
    int mc_walk = 1;          // marching cubes: the classify pass that walks z inside a workgroup, where the volume's shape allows it (0: the general one)
};

}  // namespace avc

struct avc_ctx {
    int device = 0;
    int num_cus = 256;
    avc::PackedNet warp_tmpl;      // warp + template packed as ONE stream (avatar query), geometry only (shared.6 folded into geo.0)
    avc::PackedNet warp_tmpl_clr;  // the same with the colour head (shared.6 kept)
    avc::PackedNet warp_tmpl_fold; // warp_tmpl for dense launches whose tiles lie in one (x, y) column: conv1 / conv5 without their 64 feature columns
    avc::PackedNet tmpl_only;      // template alone (pts_space == 'temp'), geometry only
    avc::PackedNet tmpl_only_clr;
    avc::PackedNet recon;
    avc::PackedNet recon_fold;     // the decoder for grid launches: the 32 image-feature columns of fc0 / fc1 / fc2 enter per (x, y) column (fused_mlp.hip)
    bool warp_set = false, tmpl_set = false;
    // staged host-side effective weights until both halves of the avatar net have arrived
    struct Staged { std::vector<std::vector<double>> W; std::vector<std::vector<double>> b; std::vector<int> cout, cin; };
    Staged warp_st, tmpl_st;
    int warp_pe = 0, tmpl_pe = 10;       // model.warping_field.pos_encoding / model.cano_template.pos_encoding of the packed weights (0 .. 10 each)
    float *pose_feat_hwc = nullptr; int pose_C = 0, pose_H = 0, pose_W = 0;
    float *img_feat_hwc = nullptr;  int img_C = 0, img_H = 0, img_W = 0;
    // scratch for meshing
    void *mc_scratch = nullptr; size_t mc_scratch_bytes = 0;
    void *mc_cells = nullptr; size_t mc_cells_bytes = 0;           // one record per crossed cell (mesh.hip)
    uint32_t *mc_tables_dev = nullptr;
    void *raster_scratch = nullptr; size_t raster_scratch_bytes = 0;
    void *fusion_scratch = nullptr; size_t fusion_scratch_bytes = 0;   // normal-fusion work buffers
    void *fusion_graph = nullptr, *fusion_graph_exec = nullptr;        // hipGraph_t / hipGraphExec_t of the fusion iterations
    int fusion_graph_H = 0, fusion_graph_W = 0, fusion_graph_iters = 0;
    void *unet = nullptr;                                          // enc::Encoder holding the warping field's U-Net
    void *encoder = nullptr;                                       // enc::Encoder: packed HGFilter weights + the launch plan of the last input size (conv_enc.hip)
    void *gn_scratch = nullptr; size_t gn_scratch_bytes = 0;       // GroupNorm slice sums
    std::vector<void *> retired_scratch;                           // outgrown blocks that a captured graph may still name: freed with the context
    void *scatter_scratch = nullptr; size_t scatter_scratch_bytes = 0;   // block counts of avc_scatter_volume
    void *render_scratch = nullptr; size_t render_scratch_bytes = 0;   // per-sample buffers of avc_render_rays_cano
    void *knn_scratch = nullptr; size_t knn_scratch_bytes = 0;     // uniform grid over the KNN reference points
    void *lbs_bound = nullptr;                                     // avc_lbs_prepare: the sequence's canonical SMPL vertices, their grid and per-cell candidate lists (knn_lbs.hip)
    void *col_scratch = nullptr; size_t col_scratch_bytes = 0;     // per-column terms of a column-folded dense query (512 floats per column)
    void *rcol_scratch = nullptr; size_t rcol_scratch_bytes = 0;   // ... of a column-folded recon query (896 floats per column)
    void *band_scratch = nullptr; size_t band_scratch_bytes = 0;   // subset launches of the recon query: [left-over tile count | column flags | tile flags | tile list]
    avc::Timing timing;
    avc::Options opt;
    int check_range = 0;                 // avc_set_range_check
    unsigned *range_flag_dev = nullptr;
};

namespace avc {
// pack.cpp
int pack_avatar(avc_ctx *ctx);   // builds warp_tmpl (and tmpl_only) from the staged weights
int pack_recon(avc_ctx *ctx, const avc_dense fc[4]);
int upload(PackedNet &net);
void release(PackedNet &net);
// fused_mlp.hip
// dense-grid point generator of the queries (pts == nullptr): three per-axis coordinate tables on the device
struct GridDesc { const float *x, *y, *z; int32_t res[3]; const int32_t *idx = nullptr; };   // idx: optional subset (flat indices, device) of the grid's points
namespace plain {       // fused_mlp.hip
int launch_avatar(avc_ctx *ctx, const float *pts, const GridDesc *grid, int64_t n, const float center[3], int occ_sigmoid,
                  float *occ, float *offset, float *rgba, bool template_only, hipStream_t s);
int launch_recon(avc_ctx *ctx, const float *pts, const GridDesc *grid, int64_t n, const float center[3], float *out, hipStream_t s);
}
namespace scaled {      // the same file built with -DAVC_LAYER_SCALE=1: Softplus layers packed with a power-of-two weight scale (pack.cpp add_warp)
int launch_avatar(avc_ctx *ctx, const float *pts, const GridDesc *grid, int64_t n, const float center[3], int occ_sigmoid,
                  float *occ, float *offset, float *rgba, bool template_only, hipStream_t s);
}
// which build of the avatar kernels serves the context's packed warping field
inline bool needs_scaled_kernels(const avc_ctx *ctx) { return ctx->warp_tmpl.sp_unscale != 1.0f; }
namespace checked {     // the same file built with -DAVC_CHECK_RANGE=1 (synchronous; returns AVC_ERR_RANGE; carries the Softplus multiply too)
int launch_avatar(avc_ctx *ctx, const float *pts, const GridDesc *grid, int64_t n, const float center[3], int occ_sigmoid,
                  float *occ, float *offset, float *rgba, bool template_only, hipStream_t s);
int launch_recon(avc_ctx *ctx, const float *pts, const GridDesc *grid, int64_t n, const float center[3], float *out, hipStream_t s);
}
int launch_nchw_to_hwc(const float *src, float *dst, int C, int H, int W, hipStream_t s);
int launch_group_norm(avc_ctx *ctx, const float *x, int N, int C, int64_t HW, int G, const float *gamma, const float *beta, float eps,
                      int relu, float *y, hipStream_t s);
int launch_scatter(avc_ctx *ctx, const uint8_t *valid, int64_t N, const float *values, const float *fill, float *vol, hipStream_t s);
// mesh.hip
int recon_mesh(avc_ctx *ctx, const float *vol, const int32_t res[3], const float bounds[6], float iso,
               float *verts, float *normals, int32_t *faces, int64_t cap_v, int64_t cap_f, int64_t counts[2], hipStream_t s);
// raster.hip
int render_cano_maps(avc_ctx *ctx, const float *verts, const float *attrs, const int32_t *faces, int64_t nf, const float center[3],
                     int size, float *front, float *back, hipStream_t s);
int render_mesh(avc_ctx *ctx, const float *verts, const float *attrs, const int32_t *faces, int64_t nf, const float mvp[16],
                int W, int H, float *out, hipStream_t s);
// fusion.hip
int canonicalize_normals(const float *live_v, const float *vert_mats, int64_t nv, const float *pos_map, const float *nrm_map, int H, int W,
                         const float mv[16], float fx, float fy, float cx, float cy, float *out, hipStream_t s);
int merge_normal_images(avc_ctx *ctx, const float *src, const float *tar, int H, int W, int iter_num, int neck_x, int neck_y, float *out, hipStream_t s);
int merge_normal_images_cover(const float *src, const float *tar, int64_t npix, float *out, hipStream_t s);
void release_fusion_graph(avc_ctx *ctx);
// conv_enc.hip
namespace enc {
int pack_encoder(avc_ctx *ctx, const avc_hgfilter *net);
int encoder_forward(avc_ctx *ctx, const float *image, int H, int W, float *feat_out, float *normx_out, int bind, hipStream_t s);
int encoder_debug_tensor(avc_ctx *ctx, int launch, int which, float *out, int *C, int *H, int *W, hipStream_t s);
void release_encoder(avc_ctx *ctx);
int pack_unet(avc_ctx *ctx, const avc_unet7ds *net);
int unet_forward(avc_ctx *ctx, const float *pos_map, int H, int W, float *out_nchw, int bind, hipStream_t s);
void release_unet(avc_ctx *ctx);
}
// render.hip
int render_rays_cano(avc_ctx *ctx, const float *ray_o, const float *ray_d, const float *near, const float *far, const float *depth, float near_dist,
                     float far_dist, const float *t_vals, int64_t P, int S, const float center[3], const float bounds[6], const float *smpl_v, int32_t n_smpl, int occ_sigmoid,
                     float *rgb_map, float *acc_map, float *depth_map, float *disp_map, float *weights, float *raw, hipStream_t s);
int blend_weight_sample(const float *vol, const int32_t res[3], int C, const float *pts01, int64_t n, float *out, hipStream_t s);
// knn_lbs.hip
int knn(avc_ctx *ctx, const float *q, int64_t nq, const float *ref, int32_t nr, int K, float *d2, int64_t *idx, hipStream_t s);
// out[i] = 0 if some reference point has cand d2 < thr2, +inf otherwise (the K = 1 search's d2 < thr2, without the search for the nearest)
int near_flags(avc_ctx *ctx, const float *q, int64_t nq, const float *ref, int32_t nr, float thr2, float *out, hipStream_t s);
int calculate_lbs(avc_ctx *ctx, const float *pts, int64_t n, const float *cano_v, const float *skin_w, int32_t nv, float *lbs, hipStream_t s);
int lbs_prepare(avc_ctx *ctx, const float *cano_v, int32_t nv, hipStream_t s);
int calculate_lbs_bound(avc_ctx *ctx, const float *pts, int64_t n, const float *skin_w, float *lbs, hipStream_t s);
int lbs_skin_bound(avc_ctx *ctx, const float *pts, const float *nrm, int64_t n, const float *skin_w, const float *jm, float *lbs, float *po, float *no, float *mo,
                   hipStream_t s);
int lbs_bound_stats(avc_ctx *ctx, int64_t out[4]);
void release_lbs_bound(avc_ctx *ctx);
int skinning(const float *pts, const float *nrm, int64_t n, const float *lbs, const float *jm, float *po, float *no, float *mo, hipStream_t s);
}  // namespace avc
