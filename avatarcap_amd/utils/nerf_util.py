"""The one function of the reference's `utils/nerf_util.py` on the test-mode path: alpha compositing along rays (`raw2outputs`,
nerf_util.py:185-212).  The test loop's vertex colours (pts_space='cano', main.py:464-477) do not come through here: `avc_render_rays_cano`
(csrc/render.hip) composites on the device, one wavefront per ray.  This torch form serves NerfRenderer.get_pixel_value for the 'posed' and
'temp' spaces, and the tests hold the device compositor to it."""
import torch


def raw2outputs(raw, z_vals, white_bkgd=False):
    """Front-to-back alpha compositing of S samples per ray (the reference's signature and return order, nerf_util.py:185-212).
    raw (R,S,4): colour and opacity of every sample; z_vals (R,S): their depths
    -> rgb_map (R,3), disp_map (R,), acc_map (R,), weights (R,S), depth_map (R,)
    weight_s = alpha_s * prod_{t < s} (1 - alpha_t + 1e-10): the transmittance in front of a sample is the EXCLUSIVE running product, written here as
    a shifted inclusive one (multiplying by the leading 1 is exact, so the values are those of the reference's cumprod over [1, 1 - alpha + 1e-10])."""
    colour, alpha = raw[..., :3], raw[..., 3]
    through = (1. - alpha) + 1e-10
    transmittance = torch.ones_like(alpha)
    transmittance[:, 1:] = torch.cumprod(through[:, :-1], dim=-1)
    weights = alpha * transmittance
    acc_map = weights.sum(dim=-1)
    depth_map = (weights * z_vals).sum(dim=-1)
    rgb_map = (weights.unsqueeze(-1) * colour).sum(dim=-2)
    if white_bkgd:
        rgb_map = rgb_map + (1. - acc_map).unsqueeze(-1)
    floor = torch.full_like(depth_map, 1e-10)
    disp_map = 1. / torch.maximum(floor, depth_map / acc_map)
    return rgb_map, disp_map, acc_map, weights, depth_map
