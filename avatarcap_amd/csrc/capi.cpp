// C ABI of libavcap_hip.so (include/avcap.h): argument checking, context state, dispatch.
#include <cstring>
#include <cstdlib>
#include <algorithm>

#include "avcap_internal.h"

namespace avc {
const char *last_error();
int effective(const avc_dense &d, const avc_bn *bn, std::vector<double> &W, std::vector<double> &b);
}  // namespace avc

using namespace avc;

extern "C" {

const char *avc_last_error(void) { return avc::last_error(); }
int avc_version(void) { return 100; }

int avc_ctx_create(int device, avc_ctx **ctx_out)
{
    AVC_REQUIRE(ctx_out, AVC_ERR_ARG, "avc_ctx_create: ctx_out is NULL");
    int count = 0;
    AVC_HIP(hipGetDeviceCount(&count));
    AVC_REQUIRE(device >= 0 && device < count, AVC_ERR_ARG, "avc_ctx_create: device %d out of range (%d visible)", device, count);
    AVC_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    AVC_HIP(hipGetDeviceProperties(&prop, device));
    AVC_REQUIRE(std::strncmp(prop.gcnArchName, "gfx950", 6) == 0, AVC_ERR_STATE,
                "avc_ctx_create: this library is built for gfx950 (MI355X) only, device %d is %s", device, prop.gcnArchName);
    avc_ctx *c = new avc_ctx();
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    // defaults of the context's switches: the environment is read here, once, and nowhere else (avc_set_option changes them afterwards)
    if (getenv("AVC_NO_FOLD")) c->opt.column_fold = 0;
    if (const char *e = getenv("AVC_MLP_BLOCKS")) c->opt.mlp_blocks = atoi(e) > 0 ? atoi(e) : 0;
    if (getenv("AVC_KNN_BRUTE")) c->opt.knn_search = 3;
    else if (const char *e = getenv("AVC_KNN_PATH")) c->opt.knn_search = !strcmp(e, "lane") ? 1 : (!strcmp(e, "wave") ? 2 : 0);
    if (getenv("AVC_FUSION_NO_GRAPH")) c->opt.fusion_graph = 0;
    *ctx_out = c;
    return AVC_OK;
}

int avc_ctx_destroy(avc_ctx *ctx)
{
    if (!ctx) return AVC_OK;
    hipSetDevice(ctx->device);
    release(ctx->warp_tmpl); release(ctx->warp_tmpl_clr); release(ctx->warp_tmpl_fold); release(ctx->tmpl_only); release(ctx->tmpl_only_clr); release(ctx->recon); release(ctx->recon_fold);
    if (ctx->pose_feat_hwc) hipFree(ctx->pose_feat_hwc);
    if (ctx->img_feat_hwc) hipFree(ctx->img_feat_hwc);
    if (ctx->mc_scratch) hipFree(ctx->mc_scratch);
    if (ctx->mc_cells) hipFree(ctx->mc_cells);
    if (ctx->scatter_scratch) hipFree(ctx->scatter_scratch);
    if (ctx->range_flag_dev) hipFree(ctx->range_flag_dev);
    if (ctx->mc_tables_dev) hipFree(ctx->mc_tables_dev);
    if (ctx->raster_scratch) hipFree(ctx->raster_scratch);
    if (ctx->knn_scratch) hipFree(ctx->knn_scratch);
    if (ctx->render_scratch) hipFree(ctx->render_scratch);
    if (ctx->col_scratch) hipFree(ctx->col_scratch);
    if (ctx->rcol_scratch) hipFree(ctx->rcol_scratch);
    if (ctx->band_scratch) hipFree(ctx->band_scratch);
    release_lbs_bound(ctx);
    if (ctx->gn_scratch) hipFree(ctx->gn_scratch);
    for (void *p : ctx->retired_scratch) hipFree(p);
    enc::release_encoder(ctx);
    enc::release_unet(ctx);
    release_fusion_graph(ctx);
    if (ctx->fusion_scratch) hipFree(ctx->fusion_scratch);
    for (int w = 0; w < 2; ++w) {
        for (auto &pr : ctx->timing.pending[w]) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
        if (ctx->timing.clk_dev[w]) hipFree(ctx->timing.clk_dev[w]);
    }
    delete ctx;
    return AVC_OK;
}

static int check_shape(const avc_dense &d, int cout, int cin, const char *what, int i)
{
    AVC_REQUIRE(d.cout == cout && d.cin == cin, AVC_ERR_ARG, "%s[%d]: expected weight (%d,%d), got (%d,%d)", what, i, cout, cin, d.cout, d.cin);
    return AVC_OK;
}

int avc_pack_warp_weights(avc_ctx *ctx, const avc_dense conv[7], const avc_bn bn[7], const avc_dense *out_affine, int pos_encoding)
{
    AVC_REQUIRE(ctx && conv && bn && out_affine, AVC_ERR_ARG, "avc_pack_warp_weights: NULL argument");
    AVC_REQUIRE(pos_encoding >= 0 && pos_encoding <= 10, AVC_ERR_ARG, "avc_pack_warp_weights: model.warping_field.pos_encoding must be 0 .. 10 (the kernels "
                "evaluate ten octaves; got %d)", pos_encoding);
    AVC_HIP(hipSetDevice(ctx->device));
    auto &st = ctx->warp_st;
    st.W.assign(8, {}); st.b.assign(8, {});
    for (int i = 0; i < 7; ++i) {
        const int D = 3 + 6 * pos_encoding + 64;                 // [posenc(xyz) | pose_feat(64)] (arch_avatar.py:97-100,136)
        const int cin = i == 0 ? D : (i == 4 ? D + 256 : 256);
        int rc = check_shape(conv[i], 256, cin, "warp conv", i + 1);
        if (rc) return rc;
        AVC_REQUIRE(bn[i].gamma && bn[i].beta && bn[i].mean && bn[i].var, AVC_ERR_ARG, "warp bn%d: NULL pointer", i + 1);
        rc = effective(conv[i], &bn[i], st.W[i], st.b[i]);
        if (rc) return rc;
    }
    int rc = check_shape(*out_affine, 3, 256, "out_layer_coord_affine", 0);
    if (rc) return rc;
    rc = effective(*out_affine, nullptr, st.W[7], st.b[7]);
    if (rc) return rc;
    ctx->warp_set = true;
    ctx->warp_pe = pos_encoding;
    return pack_avatar(ctx);
}

int avc_pack_template_weights(avc_ctx *ctx, const avc_dense shared[7], const avc_dense geo[2], const avc_dense *clr, int pos_encoding)
{
    AVC_REQUIRE(ctx && shared && geo, AVC_ERR_ARG, "avc_pack_template_weights: NULL argument");
    AVC_REQUIRE(pos_encoding >= 0 && pos_encoding <= 10, AVC_ERR_ARG, "avc_pack_template_weights: model.cano_template.pos_encoding must be 0 .. 10 (the kernels "
                "evaluate ten octaves; got %d)", pos_encoding);
    AVC_HIP(hipSetDevice(ctx->device));
    auto &st = ctx->tmpl_st;
    const int n = clr ? 12 : 9;
    st.W.assign(n, {}); st.b.assign(n, {});
    for (int i = 0; i < 7; ++i) {
        const int P = 3 + 6 * pos_encoding;                      // get_embedder(L): [x | sin, cos of 2^0 .. 2^(L-1) x] (net_util.py:40-55)
        const int cin = i == 0 ? P : (i == 4 ? 256 + P : 256);
        int rc = check_shape(shared[i], 256, cin, "shared_mlp", i);
        if (rc) return rc;
        rc = effective(shared[i], nullptr, st.W[i], st.b[i]);
        if (rc) return rc;
    }
    static const int gco[2] = {128, 2}, gci[2] = {256, 128}, cco[3] = {256, 128, 3}, cci[3] = {256, 256, 128};
    for (int i = 0; i < 2; ++i) {
        int rc = check_shape(geo[i], gco[i], gci[i], "geo_mlp", i);
        if (rc) return rc;
        rc = effective(geo[i], nullptr, st.W[7 + i], st.b[7 + i]);
        if (rc) return rc;
    }
    if (clr)
        for (int i = 0; i < 3; ++i) {
            int rc = check_shape(clr[i], cco[i], cci[i], "clr_mlp", i);
            if (rc) return rc;
            rc = effective(clr[i], nullptr, st.W[9 + i], st.b[9 + i]);
            if (rc) return rc;
        }
    ctx->tmpl_set = true;
    ctx->tmpl_pe = pos_encoding;
    return pack_avatar(ctx);
}

int avc_pack_recon_weights(avc_ctx *ctx, const avc_dense fc[4])
{
    AVC_REQUIRE(ctx && fc, AVC_ERR_ARG, "avc_pack_recon_weights: NULL argument");
    AVC_HIP(hipSetDevice(ctx->device));
    return pack_recon(ctx, fc);
}

static int set_map(avc_ctx *ctx, float **slot, int *sc, int *sh, int *sw, const float *map, int C, int H, int W, int wantC, hipStream_t s)
{
    AVC_REQUIRE(ctx && map, AVC_ERR_ARG, "set feature map: NULL argument");
    AVC_REQUIRE(C == wantC && H > 1 && W > 1, AVC_ERR_ARG, "set feature map: expected (%d,H,W) with H,W > 1, got (%d,%d,%d)", wantC, C, H, W);
    AVC_HIP(hipSetDevice(ctx->device));
    if (!*slot || *sc != C || *sh != H || *sw != W) {
        if (*slot) AVC_HIP(hipFree(*slot));
        *slot = nullptr;
        AVC_HIP(hipMalloc((void **)slot, sizeof(float) * (size_t)C * H * W));
        *sc = C; *sh = H; *sw = W;
    }
    return launch_nchw_to_hwc(map, *slot, C, H, W, s);
}

int avc_set_pose_feat_map(avc_ctx *ctx, const float *map, int C, int H, int W, avc_stream stream)
{
    return set_map(ctx, ctx ? &ctx->pose_feat_hwc : nullptr, &ctx->pose_C, &ctx->pose_H, &ctx->pose_W, map, C, H, W, 64, (hipStream_t)stream);
}
int avc_set_img_feat_map(avc_ctx *ctx, const float *map, int C, int H, int W, avc_stream stream)
{
    return set_map(ctx, ctx ? &ctx->img_feat_hwc : nullptr, &ctx->img_C, &ctx->img_H, &ctx->img_W, map, C, H, W, 32, (hipStream_t)stream);
}

// plain build of the fused kernels, or the range-checking one (avc_set_range_check)
static int run_avatar(avc_ctx *ctx, const float *pts, const GridDesc *grid, int64_t n, const float center[3], int sig, float *occ, float *offset,
                      float *rgba, bool template_only, hipStream_t s)
{
    if (ctx->check_range) return checked::launch_avatar(ctx, pts, grid, n, center, sig, occ, offset, rgba, template_only, s);
    if (!template_only && needs_scaled_kernels(ctx)) return scaled::launch_avatar(ctx, pts, grid, n, center, sig, occ, offset, rgba, template_only, s);
    return plain::launch_avatar(ctx, pts, grid, n, center, sig, occ, offset, rgba, template_only, s);
}

int avc_avatar_query(avc_ctx *ctx, const float *pts, int64_t n, const float center[3], int occupancy_sigmoid,
                     float *occ, float *offset, float *rgba, avc_stream stream)
{
    AVC_REQUIRE(ctx && center && n >= 0 && (n == 0 || (pts && occ)), AVC_ERR_ARG, "avc_avatar_query: NULL argument or negative n");
    AVC_HIP(hipSetDevice(ctx->device));
    return run_avatar(ctx, pts, nullptr, n, center, occupancy_sigmoid, occ, offset, rgba, false, (hipStream_t)stream);
}

int avc_avatar_query_grid(avc_ctx *ctx, const float *axis_x, const float *axis_y, const float *axis_z, const int32_t res[3], const float center[3],
                          int occupancy_sigmoid, float *occ, float *offset, avc_stream stream)
{
    AVC_REQUIRE(ctx && axis_x && axis_y && axis_z && res && center && occ, AVC_ERR_ARG, "avc_avatar_query_grid: NULL argument");
    AVC_REQUIRE(res[0] >= 1 && res[1] >= 1 && res[2] >= 1, AVC_ERR_ARG, "avc_avatar_query_grid: every resolution must be >= 1");
    AVC_HIP(hipSetDevice(ctx->device));
    const GridDesc g{axis_x, axis_y, axis_z, {res[0], res[1], res[2]}};
    return run_avatar(ctx, nullptr, &g, (int64_t)res[0] * res[1] * res[2], center, occupancy_sigmoid, occ, offset, nullptr, false, (hipStream_t)stream);
}

int avc_avatar_query_grid_subset(avc_ctx *ctx, const float *axis_x, const float *axis_y, const float *axis_z, const int32_t res[3], const int32_t *index,
                                 int64_t n, const float center[3], int occupancy_sigmoid, float *occ, float *offset, avc_stream stream)
{
    AVC_REQUIRE(ctx && axis_x && axis_y && axis_z && res && center && n >= 0 && (n == 0 || (index && occ)), AVC_ERR_ARG,
                "avc_avatar_query_grid_subset: NULL argument or negative n");
    AVC_REQUIRE(res[0] >= 1 && res[1] >= 1 && res[2] >= 1, AVC_ERR_ARG, "avc_avatar_query_grid_subset: every resolution must be >= 1");
    AVC_HIP(hipSetDevice(ctx->device));
    GridDesc g{axis_x, axis_y, axis_z, {res[0], res[1], res[2]}};
    g.idx = index;
    return run_avatar(ctx, nullptr, &g, n, center, occupancy_sigmoid, occ, offset, nullptr, false, (hipStream_t)stream);
}

int avc_template_query(avc_ctx *ctx, const float *pts, int64_t n, int occupancy_sigmoid, float *occ, float *rgba, avc_stream stream)
{
    AVC_REQUIRE(ctx && n >= 0 && (n == 0 || (pts && occ)), AVC_ERR_ARG, "avc_template_query: NULL argument or negative n");
    AVC_HIP(hipSetDevice(ctx->device));
    return run_avatar(ctx, pts, nullptr, n, nullptr, occupancy_sigmoid, occ, nullptr, rgba, true, (hipStream_t)stream);
}

int avc_render_rays_cano(avc_ctx *ctx, const float *ray_o, const float *ray_d, const float *near, const float *far, const float *depth, float near_dist,
                         float far_dist, const float *t_vals, int64_t n_rays, int n_samples, const float center[3], const float bounds[6], const float *cano_smpl_v, int32_t n_smpl,
                         int occupancy_sigmoid, float *rgb_map, float *acc_map, float *depth_map, float *disp_map, float *weights, float *raw,
                         avc_stream stream)
{
    AVC_REQUIRE(ctx && center && bounds && n_rays >= 0 && (n_rays == 0 || (ray_o && ray_d && near && far && cano_smpl_v)), AVC_ERR_ARG,
                "avc_render_rays_cano: NULL argument or negative n_rays");
    AVC_REQUIRE(n_samples >= 2 && n_samples <= 65536 && n_smpl >= 1, AVC_ERR_ARG, "avc_render_rays_cano: n_samples %d (>= 2), n_smpl %d (>= 1)", n_samples, n_smpl);
    AVC_HIP(hipSetDevice(ctx->device));
    return render_rays_cano(ctx, ray_o, ray_d, near, far, depth, near_dist, far_dist, t_vals, n_rays, n_samples, center, bounds, cano_smpl_v, n_smpl, occupancy_sigmoid,
                            rgb_map, acc_map, depth_map, disp_map, weights, raw, (hipStream_t)stream);
}

int avc_blend_weight_sample(avc_ctx *ctx, const float *vol_xyzc, const int32_t res[3], int channels, const float *pts01, int64_t n, float *out, avc_stream stream)
{
    AVC_REQUIRE(ctx && res && n >= 0 && (n == 0 || (vol_xyzc && pts01 && out)), AVC_ERR_ARG, "avc_blend_weight_sample: NULL argument or negative n");
    AVC_REQUIRE(res[0] >= 1 && res[1] >= 1 && res[2] >= 1 && channels >= 4 && channels % 4 == 0, AVC_ERR_ARG,
                "avc_blend_weight_sample: a (%d, %d, %d, %d) volume; every axis >= 1 and the channel count a multiple of 4 (the reference: 24)", res[0], res[1], res[2], channels);
    AVC_HIP(hipSetDevice(ctx->device));
    return blend_weight_sample(vol_xyzc, res, channels, pts01, n, out, (hipStream_t)stream);
}

int avc_recon_query(avc_ctx *ctx, const float *pts, int64_t n, const float center[3], float *out, avc_stream stream)
{
    AVC_REQUIRE(ctx && center && n >= 0 && (n == 0 || (pts && out)), AVC_ERR_ARG, "avc_recon_query: NULL argument or negative n");
    AVC_HIP(hipSetDevice(ctx->device));
    return ctx->check_range ? checked::launch_recon(ctx, pts, nullptr, n, center, out, (hipStream_t)stream)
                            : plain::launch_recon(ctx, pts, nullptr, n, center, out, (hipStream_t)stream);
}

static int run_recon(avc_ctx *ctx, const float *pts, const GridDesc *grid, int64_t n, const float center[3], float *out, hipStream_t s)
{
    return ctx->check_range ? checked::launch_recon(ctx, pts, grid, n, center, out, s) : plain::launch_recon(ctx, pts, grid, n, center, out, s);
}

int avc_recon_query_grid(avc_ctx *ctx, const float *axis_x, const float *axis_y, const float *axis_z, const int32_t res[3], const float center[3],
                         float *out, avc_stream stream)
{
    AVC_REQUIRE(ctx && axis_x && axis_y && axis_z && res && center && out, AVC_ERR_ARG, "avc_recon_query_grid: NULL argument");
    AVC_REQUIRE(res[0] >= 1 && res[1] >= 1 && res[2] >= 1, AVC_ERR_ARG, "avc_recon_query_grid: every resolution must be >= 1");
    AVC_HIP(hipSetDevice(ctx->device));
    const GridDesc g{axis_x, axis_y, axis_z, {res[0], res[1], res[2]}};
    return run_recon(ctx, nullptr, &g, (int64_t)res[0] * res[1] * res[2], center, out, (hipStream_t)stream);
}

int avc_recon_query_grid_subset(avc_ctx *ctx, const float *axis_x, const float *axis_y, const float *axis_z, const int32_t res[3], const int32_t *index,
                                int64_t n, const float center[3], float *out, avc_stream stream)
{
    AVC_REQUIRE(ctx && axis_x && axis_y && axis_z && res && center && n >= 0 && (n == 0 || (index && out)), AVC_ERR_ARG,
                "avc_recon_query_grid_subset: NULL argument or negative n");
    AVC_REQUIRE(res[0] >= 1 && res[1] >= 1 && res[2] >= 1, AVC_ERR_ARG, "avc_recon_query_grid_subset: every resolution must be >= 1");
    AVC_HIP(hipSetDevice(ctx->device));
    GridDesc g{axis_x, axis_y, axis_z, {res[0], res[1], res[2]}};
    g.idx = index;
    return run_recon(ctx, nullptr, &g, n, center, out, (hipStream_t)stream);
}

int avc_set_range_check(avc_ctx *ctx, int enabled)
{
    AVC_REQUIRE(ctx, AVC_ERR_ARG, "avc_set_range_check: NULL context");
    ctx->check_range = enabled ? 1 : 0;
    return AVC_OK;
}

int avc_group_norm(avc_ctx *ctx, const float *x, int N, int C, int64_t HW, int G, const float *gamma, const float *beta, float eps,
                   int relu, float *y, avc_stream stream)
{
    AVC_REQUIRE(ctx && x && y, AVC_ERR_ARG, "avc_group_norm: NULL argument");
    AVC_REQUIRE(N >= 1 && C >= 1 && HW >= 1 && G >= 1 && C % G == 0, AVC_ERR_ARG,
                "avc_group_norm: need N, C, HW >= 1 and C divisible by G (got N=%d C=%d HW=%lld G=%d)", N, C, (long long)HW, G);
    AVC_REQUIRE((int64_t)N * C <= 1 << 30, AVC_ERR_ARG, "avc_group_norm: N * C too large");
    AVC_HIP(hipSetDevice(ctx->device));
    return launch_group_norm(ctx, x, N, C, HW, G, gamma, beta, eps, relu, y, (hipStream_t)stream);
}

int avc_scatter_volume(avc_ctx *ctx, const uint8_t *valid, int64_t N, const float *values, const float *fill, float *vol, avc_stream stream)
{
    AVC_REQUIRE(ctx && valid && vol && N > 0, AVC_ERR_ARG, "avc_scatter_volume: NULL argument or N <= 0");
    AVC_HIP(hipSetDevice(ctx->device));
    return launch_scatter(ctx, valid, N, values, fill, vol, (hipStream_t)stream);
}

int avc_recon_mesh(avc_ctx *ctx, const float *vol, const int32_t res[3], const float bounds[6], float iso,
                   float *verts, float *normals, int32_t *faces, int64_t cap_v, int64_t cap_f, int64_t counts[2], avc_stream stream)
{
    AVC_REQUIRE(ctx && vol && res && bounds && counts, AVC_ERR_ARG, "avc_recon_mesh: NULL argument");
    AVC_REQUIRE(res[0] >= 2 && res[1] >= 2 && res[2] >= 2, AVC_ERR_ARG, "avc_recon_mesh: every resolution must be >= 2");
    AVC_REQUIRE((int64_t)res[0] * res[1] * res[2] < (int64_t)1 << 31, AVC_ERR_ARG, "avc_recon_mesh: volume too large for 32-bit voxel indices");
    AVC_HIP(hipSetDevice(ctx->device));
    return recon_mesh(ctx, vol, res, bounds, iso, verts, normals, faces, cap_v, cap_f, counts, (hipStream_t)stream);
}

int avc_render_cano_maps(avc_ctx *ctx, const float *verts, const float *attrs, int64_t nv, const int32_t *faces, int64_t nf,
                         const float center[3], int size, float *front, float *back, avc_stream stream)
{
    AVC_REQUIRE(ctx && center && front && back && nf >= 0 && nv >= 0 && (nf == 0 || (verts && attrs && faces)), AVC_ERR_ARG,
                "avc_render_cano_maps: NULL argument");
    AVC_REQUIRE(size >= 1 && size <= 8192, AVC_ERR_ARG, "avc_render_cano_maps: size must be in [1, 8192]");
    AVC_HIP(hipSetDevice(ctx->device));
    return render_cano_maps(ctx, verts, attrs, faces, nf, center, size, front, back, (hipStream_t)stream);
}

int avc_render_mesh(avc_ctx *ctx, const float *verts, const float *attrs, int64_t nv, const int32_t *faces, int64_t nf, const float mvp[16],
                    int width, int height, float *out, avc_stream stream)
{
    AVC_REQUIRE(ctx && mvp && out && nv >= 0 && nf >= 0 && (nf == 0 || (verts && faces)), AVC_ERR_ARG, "avc_render_mesh: NULL argument");
    AVC_REQUIRE(width >= 1 && height >= 1 && width <= 16384 && height <= 16384, AVC_ERR_ARG, "avc_render_mesh: image size must be in [1, 16384]");
    AVC_REQUIRE(nf < ((int64_t)1 << 32) - 1, AVC_ERR_ARG, "avc_render_mesh: too many faces for 32-bit triangle ids");
    AVC_HIP(hipSetDevice(ctx->device));
    return render_mesh(ctx, verts, attrs, faces, nf, mvp, width, height, out, (hipStream_t)stream);
}

int avc_canonicalize_normals(avc_ctx *ctx, const float *live_v, const float *vert_mats, int64_t nv, const float *pos_map, const float *normal_map,
                             int height, int width, const float mv[16], float fx, float fy, float cx, float cy, float *out, avc_stream stream)
{
    AVC_REQUIRE(ctx && mv && nv >= 0 && (nv == 0 || (live_v && vert_mats && pos_map && normal_map && out)), AVC_ERR_ARG,
                "avc_canonicalize_normals: NULL argument");
    AVC_REQUIRE(height >= 1 && width >= 1, AVC_ERR_ARG, "avc_canonicalize_normals: empty image");
    AVC_HIP(hipSetDevice(ctx->device));
    return canonicalize_normals(live_v, vert_mats, nv, pos_map, normal_map, height, width, mv, fx, fy, cx, cy, out, (hipStream_t)stream);
}

int avc_merge_normal_images(avc_ctx *ctx, const float *src, const float *tar, int height, int width, int iter_num, int neck_x, int neck_y,
                            float *out, avc_stream stream)
{
    AVC_REQUIRE(ctx && src && tar && out, AVC_ERR_ARG, "avc_merge_normal_images: NULL argument");
    AVC_REQUIRE(height >= 2 && width >= 2 && (int64_t)height * width < ((int64_t)1 << 28), AVC_ERR_ARG,
                "avc_merge_normal_images: image must be at least 2x2 and smaller than 2^28 pixels");
    AVC_REQUIRE(iter_num >= 0, AVC_ERR_ARG, "avc_merge_normal_images: iter_num < 0");
    AVC_HIP(hipSetDevice(ctx->device));
    return merge_normal_images(ctx, src, tar, height, width, iter_num, neck_x, neck_y, out, (hipStream_t)stream);
}

int avc_merge_normal_images_cover(avc_ctx *ctx, const float *src, const float *tar, int64_t npix, float *out, avc_stream stream)
{
    AVC_REQUIRE(ctx && npix >= 0 && (npix == 0 || (src && tar && out)), AVC_ERR_ARG, "avc_merge_normal_images_cover: NULL argument");
    AVC_HIP(hipSetDevice(ctx->device));
    return merge_normal_images_cover(src, tar, npix, out, (hipStream_t)stream);
}

int avc_knn(avc_ctx *ctx, const float *q, int64_t nq, const float *ref, int32_t nr, int K, float *d2, int64_t *idx, avc_stream stream)
{
    AVC_REQUIRE(ctx && nq >= 0 && nr > 0 && (nq == 0 || (q && ref)), AVC_ERR_ARG, "avc_knn: NULL argument");
    AVC_REQUIRE(K >= 1 && K <= 8 && K <= nr, AVC_ERR_ARG, "avc_knn: K must be in [1, min(8, nr)], got %d", K);
    AVC_HIP(hipSetDevice(ctx->device));
    return knn(ctx, q, nq, ref, nr, K, d2, idx, (hipStream_t)stream);
}

int avc_calculate_lbs(avc_ctx *ctx, const float *pts, int64_t n, const float *cano_v, const float *skin_w, int32_t nv, float *lbs, avc_stream stream)
{
    AVC_REQUIRE(ctx && n >= 0 && (n == 0 || (pts && lbs)), AVC_ERR_ARG, "avc_calculate_lbs: NULL argument");
    AVC_REQUIRE(cano_v && skin_w && nv >= 4, AVC_ERR_STATE, "Canonical smpl vertices are invalid!");   // smpl_util.py:31
    AVC_HIP(hipSetDevice(ctx->device));
    return calculate_lbs(ctx, pts, n, cano_v, skin_w, nv, lbs, (hipStream_t)stream);
}

int avc_lbs_prepare(avc_ctx *ctx, const float *cano_v, int32_t nv, avc_stream stream)
{
    AVC_REQUIRE(ctx && cano_v && nv >= 4, AVC_ERR_ARG, "avc_lbs_prepare: NULL argument or fewer than four vertices");
    AVC_HIP(hipSetDevice(ctx->device));
    return lbs_prepare(ctx, cano_v, nv, (hipStream_t)stream);
}

int avc_calculate_lbs_bound(avc_ctx *ctx, const float *pts, int64_t n, const float *skin_w, float *lbs, avc_stream stream)
{
    AVC_REQUIRE(ctx && n >= 0 && (n == 0 || (pts && lbs)) && skin_w, AVC_ERR_ARG, "avc_calculate_lbs_bound: NULL argument");
    AVC_HIP(hipSetDevice(ctx->device));
    return calculate_lbs_bound(ctx, pts, n, skin_w, lbs, (hipStream_t)stream);
}

int avc_lbs_skin_bound(avc_ctx *ctx, const float *pts, const float *nrm, int64_t n, const float *skin_w, const float *jm, float *lbs, float *po, float *no, float *mo,
                       avc_stream stream)
{
    AVC_REQUIRE(ctx && n >= 0 && skin_w && jm, AVC_ERR_ARG, "avc_lbs_skin_bound: NULL argument");
    AVC_REQUIRE(n == 0 || (pts && (po || mo || lbs)), AVC_ERR_ARG, "avc_lbs_skin_bound: no points, or nothing to write");
    AVC_REQUIRE(!nrm || no, AVC_ERR_ARG, "avc_lbs_skin_bound: output missing for the normals");
    AVC_HIP(hipSetDevice(ctx->device));
    return lbs_skin_bound(ctx, pts, nrm, n, skin_w, jm, lbs, po, no, mo, (hipStream_t)stream);
}

int avc_lbs_bound_stats(avc_ctx *ctx, int64_t out[4])
{
    AVC_REQUIRE(ctx && out, AVC_ERR_ARG, "avc_lbs_bound_stats: NULL argument");
    return lbs_bound_stats(ctx, out);
}

int avc_skinning(avc_ctx *ctx, const float *pts, const float *nrm, int64_t n, const float *lbs, const float *jm,
                 float *po, float *no, float *mo, avc_stream stream)
{
    AVC_REQUIRE(ctx && n >= 0 && jm, AVC_ERR_ARG, "avc_skinning: NULL argument");
    if (n == 0) return AVC_OK;                                   // an empty mesh: nothing to skin (the pointers may be NULL)
    AVC_REQUIRE(lbs && (pts || nrm), AVC_ERR_ARG, "avc_skinning: NULL argument");
    AVC_REQUIRE((!pts || po) && (!nrm || no), AVC_ERR_ARG, "avc_skinning: output missing for a given input");
    AVC_HIP(hipSetDevice(ctx->device));
    return skinning(pts, nrm, n, lbs, jm, po, no, mo, (hipStream_t)stream);
}

int avc_hgfilter_pack(avc_ctx *ctx, const avc_hgfilter *net)
{
    AVC_REQUIRE(ctx && net, AVC_ERR_ARG, "avc_hgfilter_pack: NULL argument");
    AVC_HIP(hipSetDevice(ctx->device));
    return enc::pack_encoder(ctx, net);
}

int avc_hgfilter_forward(avc_ctx *ctx, const float *image, int H, int W, float *feat_out, float *normx_out, int bind_img_feat_map, avc_stream stream)
{
    AVC_REQUIRE(ctx, AVC_ERR_ARG, "avc_hgfilter_forward: NULL ctx");
    AVC_HIP(hipSetDevice(ctx->device));
    return enc::encoder_forward(ctx, image, H, W, feat_out, normx_out, bind_img_feat_map, (hipStream_t)stream);
}

int avc_unet_pack(avc_ctx *ctx, const avc_unet7ds *net)
{
    AVC_REQUIRE(ctx && net, AVC_ERR_ARG, "avc_unet_pack: NULL argument");
    AVC_HIP(hipSetDevice(ctx->device));
    return enc::pack_unet(ctx, net);
}

int avc_unet_forward(avc_ctx *ctx, const float *pos_map, int H, int W, float *out_nchw, int bind_pose_feat_map, avc_stream stream)
{
    AVC_REQUIRE(ctx, AVC_ERR_ARG, "avc_unet_forward: NULL ctx");
    AVC_HIP(hipSetDevice(ctx->device));
    return enc::unet_forward(ctx, pos_map, H, W, out_nchw, bind_pose_feat_map, (hipStream_t)stream);
}

int avc_hgfilter_debug_tensor(avc_ctx *ctx, int launch, int which, float *out_nchw_dev, int32_t *C, int32_t *H, int32_t *W, avc_stream stream)
{
    AVC_REQUIRE(ctx && C && H && W, AVC_ERR_ARG, "avc_hgfilter_debug_tensor: NULL argument");
    AVC_HIP(hipSetDevice(ctx->device));
    return enc::encoder_debug_tensor(ctx, launch, which, out_nchw_dev, C, H, W, (hipStream_t)stream);
}

int avc_set_option(avc_ctx *ctx, const char *name, int value)
{
    AVC_REQUIRE(ctx && name, AVC_ERR_ARG, "avc_set_option: NULL argument");
    if (!strcmp(name, "column_fold")) { AVC_REQUIRE(value == 0 || value == 1, AVC_ERR_ARG, "avc_set_option: column_fold is 0 or 1"); ctx->opt.column_fold = value; }
    else if (!strcmp(name, "mlp_blocks")) { AVC_REQUIRE(value >= 0, AVC_ERR_ARG, "avc_set_option: mlp_blocks >= 0 (0: one workgroup per CU)"); ctx->opt.mlp_blocks = value; }
    else if (!strcmp(name, "knn_search")) { AVC_REQUIRE(value >= 0 && value <= 3, AVC_ERR_ARG, "avc_set_option: knn_search is 0 (auto), 1 (lane), 2 (wave) or 3 (exhaustive)"); ctx->opt.knn_search = value; }
    else if (!strcmp(name, "fusion_graph")) { AVC_REQUIRE(value == 0 || value == 1, AVC_ERR_ARG, "avc_set_option: fusion_graph is 0 or 1"); ctx->opt.fusion_graph = value; }
    else if (!strcmp(name, "enc_graph")) { AVC_REQUIRE(value == 0 || value == 1, AVC_ERR_ARG, "avc_set_option: enc_graph is 0 or 1"); ctx->opt.enc_graph = value; }
    else if (!strcmp(name, "enc_ksplit")) { AVC_REQUIRE(value == 0 || value == 1, AVC_ERR_ARG, "avc_set_option: enc_ksplit is 0 or 1"); ctx->opt.enc_ksplit = value; }
    else if (!strcmp(name, "mc_walk")) { AVC_REQUIRE(value == 0 || value == 1, AVC_ERR_ARG, "avc_set_option: mc_walk is 0 or 1"); ctx->opt.mc_walk = value; }
    else if (!strcmp(name, "lbs_reach_mm")) { AVC_REQUIRE(value >= 0 && value <= 1000, AVC_ERR_ARG, "avc_set_option: lbs_reach_mm is 0 .. 1000"); ctx->opt.lbs_reach_mm = value; }
    else if (!strcmp(name, "enc_occ2")) { AVC_REQUIRE(value == 0 || value == 1, AVC_ERR_ARG, "avc_set_option: enc_occ2 is 0 or 1"); ctx->opt.enc_occ2 = value; }
    else if (!strcmp(name, "enc_fork")) { AVC_REQUIRE(value == 0 || value == 1, AVC_ERR_ARG, "avc_set_option: enc_fork is 0 or 1"); ctx->opt.enc_fork = value; }
    else AVC_REQUIRE(false, AVC_ERR_ARG, "avc_set_option: unknown option '%s'", name);
    return AVC_OK;
}

int avc_timing_enable(avc_ctx *ctx, int enable)
{
    AVC_REQUIRE(ctx, AVC_ERR_ARG, "avc_timing_enable: NULL ctx");
    ctx->timing.enabled = enable != 0;
    return AVC_OK;
}

int avc_timing_read(avc_ctx *ctx, int which, double *avg_ms, int64_t *launches, int reset)
{
    AVC_REQUIRE(ctx && (which == 0 || which == 1) && avg_ms && launches, AVC_ERR_ARG, "avc_timing_read: bad argument");
    auto &t = ctx->timing;
    for (auto &pr : t.pending[which]) {
        AVC_HIP(hipEventSynchronize(pr.second));
        float ms = 0;
        AVC_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
        t.total_ms[which] += ms; t.launches[which] += 1;
        hipEventDestroy(pr.first); hipEventDestroy(pr.second);
    }
    t.pending[which].clear();
    *launches = t.launches[which];
    *avg_ms = t.launches[which] ? t.total_ms[which] / t.launches[which] : 0.0;
    if (reset) { t.total_ms[which] = 0; t.launches[which] = 0; }
    return AVC_OK;
}

int avc_timing_read_cycles(avc_ctx *ctx, int which, double *avg_cycles, int64_t *launches)
{
    AVC_REQUIRE(ctx && (which == 0 || which == 1) && avg_cycles && launches, AVC_ERR_ARG, "avc_timing_read_cycles: bad argument");
    auto &t = ctx->timing;
    *avg_cycles = 0.0; *launches = 0;
    const int64_t n = std::min<int64_t>(t.clk_count[which], Timing::CLK_SLOTS);
    if (n > 0 && t.clk_dev[which]) {
        AVC_HIP(hipSetDevice(ctx->device));
        AVC_HIP(hipDeviceSynchronize());
        std::vector<long long> h(2 * n);
        AVC_HIP(hipMemcpy(h.data(), t.clk_dev[which], sizeof(long long) * 2 * n, hipMemcpyDeviceToHost));
        double sum = 0;
        for (int64_t i = 0; i < n; ++i) sum += (double)(h[2 * i + 1] - h[2 * i]);
        *avg_cycles = sum / n; *launches = n;
    }
    t.clk_count[which] = 0;
    return AVC_OK;
}

}  // extern "C"
