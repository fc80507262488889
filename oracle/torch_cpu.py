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
