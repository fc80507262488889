"""The one function of the reference's `utils/nerf_util.py` on the test-mode path: alpha compositing along rays (`raw2outputs`,
nerf_util.py:185-212).  The test loop's vertex colours (pts_space='cano', main.py:464-477) do not come through here: `avc_render_rays_cano`
(csrc/render.hip) composites on the device, one wavefront per ray.  This torch form serves NerfRenderer.get_pixel_value for the 'posed' and
'temp' spaces, and the tests hold the device compositor to it."""
import torch


def raw2outputs(raw, z_vals, white_bkgd=False):
    """raw (R,S,4) = [rgb, alpha], z_vals (R,S) -> rgb_map (R,3), disp_map, acc_map, weights (R,S), depth_map."""
    rgb = raw[..., :-1]
    alpha = raw[..., -1]
    trans = torch.cumprod(torch.cat([torch.ones((alpha.shape[0], 1), dtype=alpha.dtype, device=alpha.device),
                                     1. - alpha + 1e-10], -1), -1)[:, :-1]
    weights = alpha * trans
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    acc_map = torch.sum(weights, -1)
    disp_map = 1. / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)
    if white_bkgd:
        rgb_map = rgb_map + (1. - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map
