// AddressSanitizer driver for the C-ABI (include/avcap.h), without Python: torch ships its own HIP runtime, and the sanitizer's interceptor of
// hsa_amd_memory_pool_allocate aborts inside it ("out of memory" at the first device allocation), so the ragged-size exercise of the library runs
// from this plain C++ program, linked with libavcap_hip_asan.so (host and device code instrumented; fused_mlp.hip excepted, see build.py) and
// ROCm's own runtime.  Every entry point that sizes a launch from its arguments is driven at awkward sizes -- 0, 1, one short of and one past the
// tile sizes, non-cubic volumes, images that do not fill their tiles -- with output buffers allocated to the exact size, so that an out-of-bounds
// access of a kernel or of the host code lands in a redzone.  Values are not checked here (tests/ does that against the oracle); the point is memory.
//   hipcc --offload-arch=gfx950:xnack+ -fsanitize=address -shared-libasan -I include tools/sanitize/asan_driver.cpp -L avatarcap_amd -lavcap_hip_asan ...
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "avcap.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define AV(x) do { int rc_ = (x); if (rc_ < 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, avc_last_error()); exit(3); } } while (0)

static std::mt19937 rng(7);
template <class T> static T *dev(size_t n)
{
    T *p = nullptr;
    CK(hipMalloc(&p, n ? n * sizeof(T) : 1));
    return p;
}
static float *dev_rand(size_t n, float lo = -1.f, float hi = 1.f)
{
    std::vector<float> h(n);
    std::uniform_real_distribution<float> d(lo, hi);
    for (auto &v : h) v = d(rng);
    float *p = dev<float>(n);
    if (n) CK(hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return p;
}
static std::vector<float> host_rand(size_t n, float lo, float hi)
{
    std::vector<float> h(n);
    std::uniform_real_distribution<float> d(lo, hi);
    for (auto &v : h) v = d(rng);
    return h;
}

// ---- the encoder's weights: random tensors of the reference's shapes
struct Net {
    std::vector<std::vector<float>> keep;
    const float *arr(size_t n, float s) { keep.push_back(host_rand(n, -s, s)); return keep.back().data(); }
    avc_conv2d conv(int co, int ci, int k, bool bias) { return avc_conv2d{arr((size_t)co * ci * k * k, 1.0f / std::sqrt((float)ci * k * k)), bias ? arr(co, 0.1f) : nullptr, co, ci, k, k}; }
    avc_groupnorm gn(int c) { keep.push_back(host_rand(c, 0.5f, 1.5f)); const float *g = keep.back().data(); return avc_groupnorm{g, arr(c, 0.2f), c, 32, 1e-5f}; }
    avc_convblock block(int ci, int co)
    {
        avc_convblock b{};
        b.conv[0] = conv(co / 2, ci, 3, false); b.conv[1] = conv(co / 4, co / 2, 3, false); b.conv[2] = conv(co / 4, co / 4, 3, false);
        b.bn[0] = gn(ci); b.bn[1] = gn(co / 2); b.bn[2] = gn(co / 4);
        if (ci != co) { b.downsample = conv(co, ci, 1, false); b.bn[3] = gn(ci); }
        return b;
    }
};

int main()
{
    avc_ctx *ctx = nullptr;
    AV(avc_ctx_create(0, &ctx));
    int done = 0;

    // KNN / LBS / skinning: query counts around the wave and workgroup sizes, reference counts around the grid's cell logic, K = 1 and 4
    for (int nr : {1, 5, 63, 6890}) {
        float *ref = dev_rand((size_t)nr * 3), *sw = dev_rand((size_t)nr * 24, 0.f, 1.f), *jm = dev_rand(24 * 16);
        for (int64_t nq : {0, 1, 63, 64, 65, 255, 257, 4099}) {
            float *q = dev_rand((size_t)nq * 3, -1.2f, 1.2f);
            for (int K : {1, 4}) {
                if (K > nr) continue;
                float *d2 = dev<float>((size_t)nq * K);
                int64_t *idx = dev<int64_t>((size_t)nq * K);
                AV(avc_knn(ctx, q, nq, ref, nr, K, d2, idx, nullptr));
                CK(hipFree(d2)); CK(hipFree(idx)); ++done;
            }
            if (nr >= 4) {
                float *lbs = dev<float>((size_t)nq * 24), *po = dev<float>((size_t)nq * 3), *no = dev<float>((size_t)nq * 3), *mo = dev<float>((size_t)nq * 16);
                AV(avc_calculate_lbs(ctx, q, nq, ref, sw, nr, lbs, nullptr));
                AV(avc_skinning(ctx, q, q, nq, lbs, jm, po, no, mo, nullptr));
                AV(avc_skinning(ctx, q, nullptr, nq, lbs, jm, po, nullptr, nullptr, nullptr));
                CK(hipFree(lbs)); CK(hipFree(po)); CK(hipFree(no)); CK(hipFree(mo)); done += 3;
            }
            CK(hipDeviceSynchronize());
            CK(hipFree(q));
        }
        CK(hipFree(ref)); CK(hipFree(sw)); CK(hipFree(jm));
    }

    // valid / invalid scatter at ragged N
    for (int64_t N : {1, 1023, 1024, 1025, 70001}) {
        std::vector<uint8_t> h((size_t)N);
        int64_t nv = 0;
        for (auto &b : h) { b = rng() % 3 == 0; nv += b; }
        uint8_t *valid = dev<uint8_t>((size_t)N);
        CK(hipMemcpy(valid, h.data(), (size_t)N, hipMemcpyHostToDevice));
        float *vals = dev_rand((size_t)nv), *fill = dev_rand((size_t)(N - nv)), *vol = dev<float>((size_t)N);
        AV(avc_scatter_volume(ctx, valid, N, vals, fill, vol, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipFree(valid)); CK(hipFree(vals)); CK(hipFree(fill)); CK(hipFree(vol)); ++done;
    }

    // marching cubes + normals on odd volumes (a blobby field), exact-size outputs obtained through the capacity protocol; then the rasterisers on the mesh
    const int shapes[][3] = {{2, 2, 2}, {3, 70, 11}, {33, 17, 9}, {40, 96, 36}, {65, 31, 130}};
    for (auto &sh : shapes) {
        const size_t n = (size_t)sh[0] * sh[1] * sh[2];
        std::vector<float> h(n);
        for (int x = 0; x < sh[0]; ++x)
            for (int y = 0; y < sh[1]; ++y)
                for (int z = 0; z < sh[2]; ++z) {
                    const float u = (x + 0.5f) / sh[0] - 0.5f, v = (y + 0.5f) / sh[1] - 0.5f, w = (z + 0.5f) / sh[2] - 0.5f;
                    h[((size_t)x * sh[1] + y) * sh[2] + z] = 0.33f - std::sqrt(u * u + v * v + w * w) + 0.05f * std::sin(40 * u) * std::cos(33 * v);
                }
        float *vol = dev<float>(n);
        CK(hipMemcpy(vol, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
        const int32_t res[3] = {sh[0], sh[1], sh[2]};
        const float bounds[6] = {-1, -1, -0.3f, 1, 0.9f, 0.3f};
        int64_t counts[2] = {0, 0};
        int rc = avc_recon_mesh(ctx, vol, res, bounds, 0.0f, nullptr, nullptr, nullptr, 0, 0, counts, nullptr);       // capacity query
        if (rc != AVC_ERR_CAPACITY && rc < 0 && counts[0] == 0) { fprintf(stderr, "recon_mesh %dx%dx%d: %d %s (skipped)\n", sh[0], sh[1], sh[2], rc, avc_last_error()); CK(hipFree(vol)); continue; }
        const int64_t V = counts[0], F = counts[1];
        float *verts = dev<float>((size_t)V * 3), *nrm = dev<float>((size_t)V * 3);
        int32_t *faces = dev<int32_t>((size_t)F * 3);
        AV(avc_recon_mesh(ctx, vol, res, bounds, 0.0f, verts, nrm, faces, V, F, counts, nullptr));
        ++done;
        if (V > 0 && F > 0) {
            const float center[3] = {0, -0.05f, 0};
            for (int size : {33, 96}) {
                float *front = dev<float>((size_t)size * size * 3), *back = dev<float>((size_t)size * size * 3);
                AV(avc_render_cano_maps(ctx, verts, nrm, V, faces, F, center, size, front, back, nullptr));
                CK(hipDeviceSynchronize());
                CK(hipFree(front)); CK(hipFree(back)); ++done;
            }
            const float mvp[16] = {1.2f, 0, 0, 0, 0, 1.2f, 0, 0, 0, 0, -1, -0.2f, 0, 0, -1, 2.5f};
            float *img = dev<float>((size_t)47 * 29 * 4);
            AV(avc_render_mesh(ctx, verts, verts, V, faces, F, mvp, 47, 29, img, nullptr));
            CK(hipDeviceSynchronize());
            CK(hipFree(img)); ++done;
        }
        CK(hipDeviceSynchronize());
        CK(hipFree(vol)); CK(hipFree(verts)); CK(hipFree(nrm)); CK(hipFree(faces));
    }

    // the stand-alone GroupNorm op at odd shapes
    const int gshapes[][4] = {{1, 64, 128 * 128, 32}, {2, 96, 17 * 23, 32}, {3, 8, 35, 4}, {1, 256, 1, 32}};
    for (auto &g : gshapes) {
        const size_t n = (size_t)g[0] * g[1] * g[2];
        float *x = dev_rand(n), *y = dev<float>(n), *ga = dev_rand(g[1]), *be = dev_rand(g[1]);
        AV(avc_group_norm(ctx, x, g[0], g[1], g[2], g[3], ga, be, 1e-5f, 1, y, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(ga)); CK(hipFree(be)); ++done;
    }

    // normal fusion on maps that are not a multiple of anything
    for (auto hw : {std::pair<int, int>{64, 64}, {97, 61}}) {
        const size_t n = (size_t)hw.first * hw.second * 3;
        float *src = dev_rand(n), *tar = dev_rand(n), *out = dev<float>(n);
        AV(avc_merge_normal_images(ctx, src, tar, hw.first, hw.second, 6, -5, 20, out, nullptr));
        AV(avc_merge_normal_images_cover(ctx, src, tar, (int64_t)hw.first * hw.second, out, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipFree(src)); CK(hipFree(tar)); CK(hipFree(out)); done += 2;
    }

    // the image encoder: images whose feature maps do not fill the tiles (32 x 32: partial tiles everywhere below; 64 x 32: non-square), both with and
    // without the hipGraph, split-K and the second stream
    {
        Net net;
        avc_hgfilter h{};
        h.conv1 = net.conv(64, 6, 7, true); h.bn1 = net.gn(64);
        h.conv2 = net.block(64, 128); h.conv3 = net.block(128, 128); h.conv4 = net.block(128, 256);
        std::vector<avc_convblock> hg;
        for (int i = 0; i < 13; ++i) hg.push_back(net.block(256, 256));
        h.depth = 4; h.hourglass = hg.data();
        h.top_m = net.block(256, 256);
        h.conv_last = net.conv(256, 256, 1, true); h.bn_end = net.gn(256); h.l = net.conv(32, 256, 1, true);
        AV(avc_hgfilter_pack(ctx, &h));
        for (auto hw : {std::pair<int, int>{64, 64}, {128, 64}, {63, 64}}) {
            const int H1 = (hw.first - 1) / 2 + 1, W1 = (hw.second - 1) / 2 + 1;
            float *img = dev_rand((size_t)6 * hw.first * hw.second), *feat = dev<float>((size_t)32 * H1 * W1), *normx = dev<float>((size_t)128 * H1 * W1);
            for (int variant = 0; variant < 3; ++variant) {
                AV(avc_set_option(ctx, "enc_graph", variant != 1));
                AV(avc_set_option(ctx, "enc_ksplit", variant != 2));
                AV(avc_set_option(ctx, "enc_fork", variant != 2));
                AV(avc_hgfilter_forward(ctx, img, hw.first, hw.second, feat, normx, 1, nullptr));
                CK(hipDeviceSynchronize());
                ++done;
            }
            CK(hipFree(img)); CK(hipFree(feat)); CK(hipFree(normx));
        }
    }
    AV(avc_ctx_destroy(ctx));
    printf("asan_driver: %d calls of the C-ABI completed\n", done);
    return 0;
}
