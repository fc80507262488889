"""The algebra the HIP encoders rest on (csrc/conv_enc.hip: pack_unet, pack_encoder, the up-sampling kernels), checked on the CPU with stock torch ops:
  * Conv2d(k4, s2, p1) == a 3x3 convolution (pad 1) of the space-to-depth tensor, and only the 2 x 2 taps (1 - py + a, 1 - px + b) of the 3 x 3
    neighbourhood are non-zero for input parity (py, px): the 4-tap form with kernel index k = 1 - p + 2 a;
  * ConvTranspose2d(k4, s2, p1) == a 3x3 convolution with 4 Cout parity-major outputs scattered depth-to-space, and output parity (a, b) takes only the
    taps (a + ta, b + tb) with kernel index k = 3 - a - 2 ta;
  * BatchNorm2d(affine=False, eval) folds into weights and bias.
The index formulas below are the ones pack_unet uses; the GPU tests (tests/test_gpu_producers.py) hold the kernels to the reference's goldens."""
import numpy as np
import torch
import torch.nn.functional as F


def _s2d(x):
    """(B, C, H, W) -> (B, 4 C, H/2, W/2), channel (py 2 + px) C + c  (OUT_S2D / s2d_kernel)."""
    B, C, H, W = x.shape
    return x.reshape(B, C, H // 2, 2, W // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(B, 4 * C, H // 2, W // 2)


def _d2s(y, cout):
    """(B, 4 Cout, H, W) parity-major -> (B, Cout, 2 H, 2 W)  (OUT_D2S)."""
    B, _, H, W = y.shape
    return y.reshape(B, 2, 2, cout, H, W).permute(0, 3, 4, 1, 5, 2).reshape(B, cout, 2 * H, 2 * W)


def test_stride2_conv_is_a_3x3_conv_of_the_space_to_depth_tensor():
    g = torch.Generator().manual_seed(1)
    cin, cout = 8, 16
    x = torch.randn(2, cin, 12, 20, generator=g, dtype=torch.float64)
    w = torch.randn(cout, cin, 4, 4, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, w, stride=2, padding=1)
    # the 3 x 3 form: input row 2 o - 1 + k is row o - 1 + ty of parity py with k = 2 ty + py - 1
    w3 = torch.zeros(cout, 4 * cin, 3, 3, dtype=torch.float64)
    for par in range(4):
        for ty in range(3):
            for tx in range(3):
                ky, kx = 2 * ty + (par >> 1) - 1, 2 * tx + (par & 1) - 1
                if 0 <= ky <= 3 and 0 <= kx <= 3:
                    w3[:, par * cin:(par + 1) * cin, ty, tx] = w[:, :, ky, kx]
    assert torch.allclose(F.conv2d(_s2d(x), w3, padding=1), ref, atol=1e-12)
    assert int((w3 != 0).sum()) == w.numel()                                   # 16 of the 36 (parity, tap) slots
    # the 4-tap form: for parity (py, px) the taps (1 - py + a, 1 - px + b), kernel index k = 1 - p + 2 a
    for par in range(4):
        py, px = par >> 1, par & 1
        nz = (w3[:, par * cin:(par + 1) * cin] != 0).any(0).any(0)
        want = torch.zeros(3, 3, dtype=torch.bool)
        for t in range(4):
            a, b = t >> 1, t & 1
            want[1 - py + a, 1 - px + b] = True
            assert torch.equal(w3[:, par * cin:(par + 1) * cin, 1 - py + a, 1 - px + b], w[:, :, 1 - py + 2 * a, 1 - px + 2 * b])
        assert torch.equal(nz, want)


def test_transposed_conv_is_a_3x3_conv_with_parity_major_outputs():
    g = torch.Generator().manual_seed(2)
    cin, cout = 8, 6
    x = torch.randn(2, cin, 5, 7, generator=g, dtype=torch.float64)
    wt = torch.randn(cin, cout, 4, 4, generator=g, dtype=torch.float64)        # ConvTranspose2d stores (in, out, kh, kw)
    ref = F.conv_transpose2d(x, wt, stride=2, padding=1)
    w3 = torch.zeros(4 * cout, cin, 3, 3, dtype=torch.float64)
    for par in range(4):
        for ty in range(3):
            for tx in range(3):
                ky, kx = (par >> 1) + 3 - 2 * ty, (par & 1) + 3 - 2 * tx
                if 0 <= ky <= 3 and 0 <= kx <= 3:
                    w3[par * cout:(par + 1) * cout, :, ty, tx] = wt[:, :, ky, kx].t()
    assert torch.allclose(_d2s(F.conv2d(x, w3, padding=1), cout), ref, atol=1e-12)
    # the 4-tap form: output parity (a, b) takes the taps (a + ta, b + tb), kernel index k = 3 - a - 2 ta
    for par in range(4):
        a, b = par >> 1, par & 1
        nz = (w3[par * cout:(par + 1) * cout] != 0).any(0).any(0)
        want = torch.zeros(3, 3, dtype=torch.bool)
        for t in range(4):
            ta, tb = t >> 1, t & 1
            want[a + ta, b + tb] = True
            assert torch.equal(w3[par * cout:(par + 1) * cout, :, a + ta, b + tb], wt[:, :, 3 - a - 2 * ta, 3 - b - 2 * tb].t())
        assert torch.equal(nz, want)


def test_batchnorm_without_affine_folds_into_weights_and_bias():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 4, 8, 8, generator=g, dtype=torch.float64)
    conv = torch.nn.Conv2d(4, 5, 3, 1, 1).double()
    bn = torch.nn.BatchNorm2d(5, affine=False).double().eval()
    bn.running_mean.copy_(torch.randn(5, generator=g, dtype=torch.float64))
    bn.running_var.copy_(torch.rand(5, generator=g, dtype=torch.float64) + 0.5)
    sc = 1.0 / torch.sqrt(bn.running_var + bn.eps)
    w, b = conv.weight * sc[:, None, None, None], conv.bias * sc - bn.running_mean * sc
    with torch.no_grad():
        assert torch.allclose(F.conv2d(x, w, b, padding=1), bn(conv(x)), atol=1e-12)


def test_bilinear_x2_is_the_four_tap_formula_of_up2_kernel():
    """up2_kernel: real = max((dst + 0.5) / 2 - 0.5, 0), i0 = floor(real), i1 = min(i0 + 1, size - 1) -- F.interpolate(scale 2, bilinear, align_corners=False)."""
    x = torch.randn(1, 3, 5, 6, generator=torch.Generator().manual_seed(4), dtype=torch.float64)
    ref = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
    H, W = x.shape[2:]
    out = torch.zeros_like(ref)
    for oy in range(2 * H):
        ry = max((oy + 0.5) * 0.5 - 0.5, 0.0); y0 = int(ry); y1 = min(y0 + 1, H - 1); ly = ry - y0
        for ox in range(2 * W):
            rx = max((ox + 0.5) * 0.5 - 0.5, 0.0); x0 = int(rx); x1 = min(x0 + 1, W - 1); lx = rx - x0
            out[0, :, oy, ox] = (x[0, :, y0, x0] * (1 - ly) * (1 - lx) + x[0, :, y0, x1] * (1 - ly) * lx + x[0, :, y1, x0] * ly * (1 - lx) + x[0, :, y1, x1] * ly * lx)
    assert torch.allclose(out, ref, atol=1e-12)


def test_hgfilter_conv1_is_a_4x4_tap_conv_of_the_space_to_depth_image():
    """pack_encoder: Conv2d(6, 64, 7, stride 2, padding 3) (HGFilters.py:134) == a 4 x 4-tap convolution of the space-to-depth image, halo 2 before and 1
    after, with kernel index k = 2 t + p - 1 (csrc/conv_enc.hip, TAPS == 16)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 6, 20, 28, generator=g, dtype=torch.float64)
    w = torch.randn(8, 6, 7, 7, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, w, stride=2, padding=3)
    w4 = torch.zeros(8, 24, 4, 4, dtype=torch.float64)
    for par in range(4):
        for ty in range(4):
            for tx in range(4):
                ky, kx = 2 * ty + (par >> 1) - 1, 2 * tx + (par & 1) - 1
                if 0 <= ky <= 6 and 0 <= kx <= 6:
                    w4[:, par * 6:(par + 1) * 6, ty, tx] = w[:, :, ky, kx]
    got = F.conv2d(F.pad(_s2d(x), (2, 1, 2, 1)), w4)
    assert got.shape == ref.shape and torch.allclose(got, ref, atol=1e-12)


def test_bicubic_x2_align_corners_is_the_kernels_16_tap_formula():
    """upadd_kernel / upadd_tiled_kernel: F.interpolate(scale_factor=2, mode='bicubic', align_corners=True) (HGFilters.py:116) with ATen's cubic
    convolution coefficients (A = -0.75) and clamped taps."""
    def coeffs(t, A=-0.75):
        x0, x1, x2 = t + 1.0, t, 1.0 - t
        x3 = x2 + 1.0
        return [((A * x0 - 5 * A) * x0 + 8 * A) * x0 - 4 * A, ((A + 2) * x1 - (A + 3)) * x1 * x1 + 1, ((A + 2) * x2 - (A + 3)) * x2 * x2 + 1,
                ((A * x3 - 5 * A) * x3 + 8 * A) * x3 - 4 * A]
    x = torch.randn(1, 2, 6, 5, generator=torch.Generator().manual_seed(6), dtype=torch.float64)
    ref = F.interpolate(x, scale_factor=2, mode='bicubic', align_corners=True)
    Hb, Wb = x.shape[2:]
    H, W = 2 * Hb, 2 * Wb
    sy, sx = (Hb - 1) / (H - 1), (Wb - 1) / (W - 1)
    out = torch.zeros_like(ref)
    for oy in range(H):
        ry = sy * oy; iy = int(np.floor(ry)); cy = coeffs(ry - iy)
        for ox in range(W):
            rx = sx * ox; ix = int(np.floor(rx)); cx = coeffs(rx - ix)
            acc = 0
            for m in range(4):
                yy = min(max(iy - 1 + m, 0), Hb - 1)
                acc = acc + cy[m] * sum(cx[k] * x[0, :, yy, min(max(ix - 1 + k, 0), Wb - 1)] for k in range(4))
            out[0, :, oy, ox] = acc
    assert torch.allclose(out, ref, atol=1e-12)
