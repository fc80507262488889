// Microbenchmark of the image encoder's kernels (avatarcap_amd/csrc/conv_enc.hip), one launch configuration at a time, so that a kernel
// variant can be timed without the rest of the frame:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I avatarcap_amd/csrc -o tools/ubench/enc_bench
// tools/ubench/enc_bench.hip   (tools/enc_bench.sh builds and runs it on the GPU box).
//   enc_bench conv  H W Cin Cout taps CT PT ksplit stats(0|1|2: none | raw | raw + y)     one convolution of the encoder
//   enc_bench upadd H W C | pool H W C | normrelu H W C                                    the element-wise launches (H, W: output size)
// Prints microseconds per launch (hipEvents around `iters` back-to-back launches on one stream) and the implied rates.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "conv_enc.hip"

namespace avc {
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
}  // namespace avc

using namespace avc;
using namespace avc::enc;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float *dev_random(size_t n, float lo, float hi, unsigned seed)
{
    std::vector<float> h(n);
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> d(lo, hi);
    for (auto &v : h) v = d(g);
    float *p = nullptr;
    CK(hipMalloc(&p, n * sizeof(float)));
    CK(hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return p;
}
static float *dev_zero(size_t n) { float *p = nullptr; CK(hipMalloc(&p, std::max<size_t>(n, 4) * sizeof(float))); CK(hipMemset(p, 0, std::max<size_t>(n, 4) * sizeof(float))); return p; }

template <class F>
static double time_us(F &&launch, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3 * ms / iters;
}

static StatOut make_stat(int C, int first, int ntiles)
{
    StatOut s{};
    s.cpg = C / 32; s.ntiles = ntiles;
    f32x2 *t = nullptr; CK(hipMalloc(&t, sizeof(f32x2) * 32 * ntiles)); CK(hipMemset(t, 0, sizeof(f32x2) * 32 * ntiles));
    s.part = t + (size_t)(first / s.cpg) * ntiles;
    return s;
}
static f32x2 *dev_part(int nt)
{
    std::vector<float> h(2 * 32 * (size_t)nt);
    for (int i = 0; i < 32 * nt; ++i) { h[2 * i] = 0.01f * (i % 7); h[2 * i + 1] = 1.0f + 0.1f * (i % 5); }
    f32x2 *p = nullptr; CK(hipMalloc(&p, h.size() * sizeof(float))); CK(hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return p;
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: see the header of enc_bench.hip\n"); return 2; }
    const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 50;
    const std::string what = argv[1];
    const int H = atoi(argv[2]), W = atoi(argv[3]);
    if (what == "conv") {
        const int Cin = atoi(argv[4]), Cout = atoi(argv[5]), taps = atoi(argv[6]), CT = atoi(argv[7]), PT = atoi(argv[8]), ksplit = atoi(argv[9]), stats = atoi(argv[10]);
        Encoder e;
        std::vector<float> w((size_t)Cout * Cin * taps);
        std::mt19937 g(1);
        std::uniform_real_distribution<float> d(-0.05f, 0.05f);
        for (auto &v : w) v = d(g);
        const int k = taps == 9 ? 3 : (taps == 16 ? 4 : 1);
        avc_conv2d c{w.data(), nullptr, Cout, Cin, k, k};
        DevConv dc;
        if (pack_conv(&e, c, taps, dc, "bench")) return 1;
        Launch L{}; L.kind = L_CONV; L.TAPS = taps; L.norm = taps != 16; L.TWC = W >= 32 ? 32 : 16; L.CT = CT; L.PT = PT;
        L.occ2 = getenv("OCC2") && atoi(getenv("OCC2"));
        const int rows = 4 * PT * (32 / L.TWC);
        ConvArgs &a = L.conv;
        a.x = dev_random((size_t)H * W * Cin, -2.f, 2.f, 2); a.H = H; a.W = W; a.Cin = Cin;
        const int in_nt = getenv("IN_NT") ? atoi(getenv("IN_NT")) : 256;
        a.in_part = dev_part(in_nt); a.in_nt = in_nt; a.in_inv_n = 1.0f / in_nt; a.in_eps = 1e-5f;
        a.gamma = dev_random(Cin, 0.5f, 1.5f, 4); a.beta = dev_random(Cin, -0.5f, 0.5f, 5); a.in_cpg = Cin / 32; a.in_scale = 16.f; a.in_slope = 1.0f;
        const int v = CT == 4 ? 2 : (CT == 2 ? 1 : 0);
        a.slice_bytes = (unsigned)(Cin / 32) * taps * 2 * CT * 2048;
        a.wstream = dc.wstream + dc.off[v]; a.wbytes = a.slice_bytes * (Cout / (32 * CT));
        a.bias = nullptr; a.out_scale = dc.wscale_inv / a.in_scale; a.Cout = Cout;
        const int yC = 2 * Cout;
        a.raw = dev_zero((size_t)H * W * Cout);
        a.y = stats >= 2 ? dev_zero((size_t)H * W * yC) : nullptr; a.res = stats >= 2 ? dev_random((size_t)H * W * yC, -1.f, 1.f, 6) : nullptr; a.yC = yC; a.ycoff = 0;
        a.tiles_x = (W + L.TWC - 1) / L.TWC; a.tiles_y = (H + rows - 1) / rows;
        const int ntiles = a.tiles_x * a.tiles_y, wg = ntiles * (Cout / (32 * CT));
        if (stats >= 1) a.st_raw = make_stat(Cout, 0, ntiles);
        if (stats >= 2) a.st_y = make_stat(yC, 0, ntiles);
        a.ksplit = ksplit; a.range_flag = nullptr;
        if (ksplit > 1) {
            a.kpart = dev_zero((size_t)wg * ksplit * 256 * PT * CT * 16);
            CK(hipMalloc(&a.kcounter, sizeof(unsigned) * wg)); CK(hipMemset(a.kcounter, 0, sizeof(unsigned) * wg));
        }
        L.grid = (unsigned)(wg * ksplit);
        const double us = time_us([&] { if (launch_conv(L, 0)) exit(1); }, iters);
#ifdef AVC_ENC_PHASES
        {   // one more launch with s_memtime stamps per workgroup: where the time of a workgroup goes (cycles of the 100 MHz-independent shader counter)
            unsigned long long *ph = nullptr;
            CK(hipMalloc(&ph, sizeof(unsigned long long) * 8 * L.grid)); CK(hipMemset(ph, 0, sizeof(unsigned long long) * 8 * L.grid));
            a.phases = ph;
            if (launch_conv(L, 0)) exit(1);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> hph(8 * (size_t)L.grid);
            CK(hipMemcpy(hph.data(), ph, hph.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t1 = 0; double seg[5] = {0, 0, 0, 0, 0};
            for (unsigned b = 0; b < L.grid; ++b) {
                t0 = std::min(t0, hph[8 * b]); t1 = std::max(t1, hph[8 * b + 5]);
                for (int k = 0; k < 5; ++k) seg[k] += (double)(hph[8 * b + k + 1] - hph[8 * b + k]) / L.grid;
            }
            double first = 0; for (unsigned b = 0; b < L.grid; ++b) first += (double)(hph[8 * b] - t0) / L.grid;
            printf("   phases (mean ticks per workgroup; launch span %llu ticks): start skew %.0f | prologue %.0f | main loop %.0f | range flag+splitK %.0f+emit-issue %.0f | stores drain %.0f | stats %.0f\n",
                   t1 - t0, first, seg[0], seg[1], 0.0, seg[2], seg[3], seg[4]);
            a.phases = nullptr;
        }
#endif
        const double flop = 2.0 * H * W * Cin * Cout * taps;
        printf("conv %dx%d %d->%d taps %d CT%d PT%d ksplit %d stats %d: %d workgroups, %.1f us, %.1f TFLOP/s algorithmic (%.1f issued as 3 fp16 passes)\n", H, W, Cin, Cout, taps,
               CT, PT, ksplit, stats, L.grid, us, flop / us * 1e-6, 3 * flop / us * 1e-6);
        return 0;
    }
    const int C = atoi(argv[4]);
    EltArgs g{};
    g.H = H; g.W = W; g.C = C;
    g.out = dev_zero((size_t)H * W * C);
    const int npix = H * W;
    const bool stats = !(argc > 5 && atoi(argv[5]) == 0);
    if (what == "upadd") {
        g.a = dev_random((size_t)npix * C, -1.f, 1.f, 1); g.b = dev_random((size_t)npix / 4 * C, -1.f, 1.f, 2); g.Hb = H / 2; g.Wb = W / 2;
        UpTiledArgs u{g, (W + UT_W - 1) / UT_W, (H + UT_H - 1) / UT_H};
        u.e.ntiles = u.tiles_x * u.tiles_y; u.e.ppw = UT_H * UT_W;
        if (stats) u.e.st = make_stat(C, 0, u.e.ntiles);
        const unsigned grid = u.e.ntiles * (C / UT_C);
        const double us = time_us([&] { hipLaunchKernelGGL(upadd_tiled_kernel, dim3(grid), dim3(256), 0, 0, u); }, iters);
        printf("upadd (tiled) %dx%dx%d stats %d: %u workgroups, %.1f us, %.2f TB/s of (up1 + out + low3)\n", H, W, C, (int)stats, grid, us, 2.25 * npix * C * 4 / us * 1e-6);
        g.ppw = std::max(16, (npix + 511) / 512); g.ntiles = (npix + g.ppw - 1) / g.ppw;
        if (stats) g.st = make_stat(C, 0, g.ntiles);
        const double us2 = time_us([&] { hipLaunchKernelGGL(upadd_kernel, dim3(g.ntiles), dim3(256), 0, 0, g); }, iters);
        printf("upadd (direct) %dx%dx%d stats %d: %d workgroups, %.1f us\n", H, W, C, (int)stats, g.ntiles, us2);
        return 0;
    }
    g.ppw = std::max(16, (npix + 511) / 512); g.ntiles = (npix + g.ppw - 1) / g.ppw;
    if (stats) g.st = make_stat(C, 0, g.ntiles);
    if (what == "pool") {
        g.a = dev_random((size_t)npix * 4 * C, -1.f, 1.f, 1); g.Hb = 2 * H; g.Wb = 2 * W;
        const double us = time_us([&] { hipLaunchKernelGGL(avgpool_kernel, dim3(g.ntiles), dim3(256), 0, 0, g); }, iters);
        printf("avgpool -> %dx%dx%d stats %d: %d workgroups, %.1f us, %.2f TB/s\n", H, W, C, (int)stats, g.ntiles, us, 5.0 * npix * C * 4 / us * 1e-6);
    } else if (what == "normrelu") {
        g.a = dev_random((size_t)npix * C, -1.f, 1.f, 1); g.Hb = H; g.Wb = W;
        g.in_part = dev_part(256); g.in_nt = 256; g.in_inv_n = 1.0f / 256; g.in_eps = 1e-5f; g.gamma = dev_random(C, 0.5f, 1.5f, 4); g.beta = dev_random(C, -0.5f, 0.5f, 5); g.in_cpg = C / 32;
        const double us = time_us([&] { hipLaunchKernelGGL(normrelu_kernel, dim3(g.ntiles), dim3(256), 0, 0, g); }, iters);
        printf("normrelu %dx%dx%d stats %d: %d workgroups, %.1f us, %.2f TB/s\n", H, W, C, (int)stats, g.ntiles, us, 2.0 * npix * C * 4 / us * 1e-6);
    }
    return 0;
}
