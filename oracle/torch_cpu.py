"""The avatar occupancy query written with stock PyTorch CPU ops -- TEST / BASELINE INFRASTRUCTURE ONLY.

bench.py's `cpu_baseline` times this on the GPU box's host cores ("stock-PyTorch-CPU modules with
identical weights ... torch.set_num_threads(all physical cores)"): it is the closest thing to the reference's own CPU path that
can travel (the reference's modules cannot), a functional restatement over the reference-shaped state dict, one torch op per
reference op (Conv1d k=1 == matmul; BatchNorm1d eval; Softplus; grid_sample; the sin/cos embedder).  tests/test_oracle_golden.py
holds it to the NumPy oracle and, through it, to the goldens generated from the reference.
Functions cite the reference lines they follow.
"""
import torch
import torch.nn.functional as F


def _embed(p, multires):                                   # utils/net_util.py:5-55
    if multires == 0:
        return p
    out = [p]
    for i in range(multires):
        out += [torch.sin(p * 2.0 ** i), torch.cos(p * 2.0 ** i)]
    return torch.cat(out, -1)


def _lin(sd, key, x):                                      # Conv1d(kernel 1) on (n, C) rows
    return F.linear(x, sd[key + '.weight'][:, :, 0], sd[key + '.bias'])


def _offset_decoder(sd, p, x0):                            # network/mlp.py:75-112
    def block(i, x):
        y = _lin(sd, f'{p}.conv{i}', x)
        y = F.batch_norm(y, sd[f'{p}.bn{i}.running_mean'], sd[f'{p}.bn{i}.running_var'], sd[f'{p}.bn{i}.weight'], sd[f'{p}.bn{i}.bias'], False, 0.0, 1e-5)
        return F.softplus(y)
    x = x0
    for i in (1, 2, 3, 4):
        x = block(i, x)
    x = block(5, torch.cat([x0, x], -1))                   # input first (mlp.py:106)
    return block(7, block(6, x))


def _mlp(sd, p, x0, n_layers, res=(), act=F.relu):         # network/mlp.py:5-72
    x = x0
    for l in range(n_layers):
        last = l == n_layers - 1
        inp = torch.cat([x, x0], -1) if l in res else x
        x = _lin(sd, f'{p}.fc_list.{l}' + ('' if last else '.0'), inp)
        if not last:
            x = act(x)
    return x


@torch.no_grad()
def occupancy_query(cano_pts, pose_feat_map, center, sd, if_type='sdf', chunk=1 << 18):
    """cano_pts (n,3), pose_feat_map (64,H,W), center (3,), sd: dict of float32 CPU tensors keyed like net.pt['network'].
    -> cano_pts_ov (n,1), nonrigid_offset (n,3).  Chunked like the reference (arch_avatar.py:366)."""
    fmap = pose_feat_map[None]
    occs, offs = [], []
    for s in range(0, cano_pts.shape[0], chunk):
        p = cano_pts[s:s + chunk]
        q = p - center
        grid = torch.stack([q[:, 0], -q[:, 1]], -1)[None, :, None]                                # arch_avatar.py:125-132
        feat = F.grid_sample(fmap, grid, 'bilinear', 'border', True)[0, :, :, 0].T                # :133
        h = _offset_decoder(sd, 'warping_field.mlp', torch.cat([_embed(p, 0), feat], -1))         # :136-137
        off = _lin(sd, 'warping_field.out_layer_coord_affine', h)                                 # :138
        shared = _mlp(sd, 'cano_template.shared_mlp', _embed(p + off, 10), 7, res=(4,))           # :65-72
        geo = _mlp(sd, 'cano_template.geo_mlp', shared, 2, act=lambda t: F.leaky_relu(t, 0.02))
        occs.append(geo[:, 0:1] if if_type == 'sdf' else torch.sigmoid(geo[:, 0:1]))              # :77-80
        offs.append(off)
    return torch.cat(occs), torch.cat(offs)


@torch.no_grad()
def calculate_lbs(points, cano_smpl_v, skin_weights, chunk=16384):
    """utils/smpl_util.py:24-39 with torch CPU ops: KNN-4 by exhaustive squared distances, Gaussian weights (r = 0.05), blend."""
    out = []
    for s in range(0, points.shape[0], chunk):
        p = points[s:s + chunk]
        d2 = ((p[:, None, :] - cano_smpl_v[None, :, :]) ** 2).sum(-1)
        d, idx = torch.topk(d2, 4, dim=-1, largest=False, sorted=True)
        w = torch.exp(-d / (2 * 0.05 ** 2))
        w = w / (w.sum(-1, keepdim=True) + 1e-16)
        out.append((skin_weights[idx] * w[..., None]).sum(-2))
    return torch.cat(out)


@torch.no_grad()
def unet7ds(sd, x, prefix='warping_field.unet'):
    """UnetNoCond7DS.forward (network/unets.py:201-229; blocks :10-60) over the reference-shaped state dict: x (1,6,H,W) -> (1,64,H,W).
    Down blocks: LeakyReLU(0.2) before every convolution but the first, Conv2d(k4, s2, p1, no bias), BatchNorm2d(affine=False, eval) on conv2..6
    (:175-181).  Up blocks: ReLU, ConvTranspose2d(k4, s2, p1, no bias) + BatchNorm (upconv1..3) or bilinear x2 (align_corners=False) + Conv2d(k3, p1)
    (upconvC5..C7; C7 without BatchNorm), then the skip concatenation (:42-60).  upconv3 is applied twice and upconv4 never (:213-214).
    (The reference's in-place LeakyReLU also rewrites the tensors it later concatenates as skips; every consumer of a skip applies ReLU first and
    relu(leaky_relu(d)) == relu(d), so the outputs are the same.)"""
    def bn(name, y):
        return F.batch_norm(y, sd[f'{prefix}.{name}.bn.running_mean'], sd[f'{prefix}.{name}.bn.running_var'], None, None, False, 0.0, 1e-5)

    def down(i, y):
        if i > 1:
            y = F.leaky_relu(y, 0.2)
        y = F.conv2d(y, sd[f'{prefix}.conv{i}.conv.weight'], None, stride=2, padding=1)
        return bn(f'conv{i}', y) if 2 <= i <= 6 else y

    def up(name, y, skip=None, transposed=True, norm=True):
        y = F.relu(y)
        if transposed:
            y = F.conv_transpose2d(y, sd[f'{prefix}.{name}.up.weight'], None, stride=2, padding=1)
        else:
            y = F.interpolate(y, scale_factor=2, mode='bilinear', align_corners=False)
            y = F.conv2d(y, sd[f'{prefix}.{name}.up.1.weight'], sd[f'{prefix}.{name}.up.1.bias'], padding=1)
        if norm:
            y = bn(name, y)
        return y if skip is None else torch.cat([y, skip], 1)

    d = [x]
    for i in range(1, 8):
        d.append(down(i, d[-1]))
    u = up('upconv1', d[7], d[6])
    u = up('upconv2', u, d[5])
    u = up('upconv3', u, d[4])
    u = up('upconv3', u, d[3])
    u = up('upconvC5', u, d[2], transposed=False)
    u = up('upconvC6', u, d[1], transposed=False)
    return up('upconvC7', u, None, transposed=False, norm=False)


@torch.no_grad()
def vertex_normals(vol, voxel, grid_pts):
    """utils/recon_util.py:9-48 with torch CPU ops: the three 3x3x3 Sobel kernels as ONE conv3d (zero padding 1), each axis divided by 32 * voxel, then
    the trilinear fetch at the vertices (F.grid_sample: border, align_corners=True, xyz -> zyx) and n / ||n||.
    vol (X,Y,Z) f32, voxel (3,), grid_pts (V,3) in [-1,1]^3 -> (V,3)."""
    s, dd = torch.tensor([1.0, 2.0, 1.0]), torch.tensor([-1.0, 0.0, 1.0])
    k = torch.stack([dd[:, None, None] * s[None, :, None] * s[None, None, :], s[:, None, None] * dd[None, :, None] * s[None, None, :],
                     s[:, None, None] * s[None, :, None] * dd[None, None, :]])                                   # :10-21
    k = k / (32.0 * torch.as_tensor(voxel, dtype=torch.float32))[:, None, None, None]
    nv = F.conv3d(vol[None, None], k[:, None], padding=1)                                                         # (1,3,X,Y,Z)   :24-29
    g = grid_pts[:, [2, 1, 0]][None, :, None, None, :]                                                            # :41
    n = F.grid_sample(nv, g, 'bilinear', 'border', True)[0, :, :, 0, 0].T                                         # :42-45
    return n / n.norm(dim=1, keepdim=True)                                                                        # :46-47
