"""The per-frame driver segment of `main.run_avatarcap` (main.py:357-367, 383-389, 438-453) with every
tensor kept on the device: no `.cpu()` of the volume for marching cubes, no host round trip of the
vertices for normals / LBS.  Steps 1-4 of the loop body are here: avatar geometry, canonical normal maps + fusion with the
image-observed normals, reconstruction, vertex colours; image file I/O stays with the caller.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import config
from . import _lib
from .utils import recon_util
from .utils.smpl_util import smpl_util


class _stage:
    """roctx range around one stage of a frame (tracing): `rocprofv3 --marker-trace` / rocprof-sys show the six stages -- U-Net,
    avatar query, marching cubes, LBS, HGFilter, recon query -- plus normal maps / fusion / colours as named spans.  torch.cuda.nvtx IS roctx on
    ROCm; where the extension is missing the range is a no-op."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        try:
            torch.cuda.nvtx.range_push(self.name)
            self.on = True
        except Exception:       # noqa: BLE001
            self.on = False
        return self

    def __exit__(self, *exc):
        if self.on:
            torch.cuda.nvtx.range_pop()
        return False


def fill_volume(values: torch.Tensor, valid_u8: torch.Tensor, invalid_ov: torch.Tensor, out: torch.Tensor | None = None):
    """occ_volume[valid] = values; occ_volume[~valid] = invalid_pts_ov   (main.py:362-363)."""
    N = valid_u8.numel()
    vol = out if out is not None else torch.empty(N, dtype=torch.float32, device=values.device)
    values = values.reshape(-1).contiguous()
    if values.numel() == N:
        return values if out is None else out.copy_(values)
    _lib.check(_lib.lib().avc_scatter_volume(_lib.ctx(values.device), valid_u8.data_ptr(), N, values.data_ptr(),
                                             invalid_ov.contiguous().data_ptr(), vol.data_ptr(), _lib.stream_ptr(values.device)))
    return vol


def _host_center(items: dict):
    """items['cano_smpl_center'] as the (1,3) host array it was uploaded from when the item dict carries it ('_host': dataset.to_cuda,
    frame_io.FramePrefetcher), else the device tensor (read back by the callee, one stream drain)."""
    c = _lib.host_mirror(items, 'cano_smpl_center')
    if c is not None and items['cano_smpl_center'].shape[0] == 1:
        return c.reshape(1, 3)
    return items['cano_smpl_center']


class FramePipeline:
    """Holds the networks and the per-sequence constants; `avatar_frame` / `recon_frame` are steps
    1 and 3 of main.py's loop body."""

    def __init__(self, network, dataset, recon_net=None, lbs_reach_mm=None):
        from .network.arch_avatar import OccupancyNet
        self.network = network
        self.occ_net = OccupancyNet(network)
        self.recon_net = recon_net
        self.ds = dataset
        self.vol_res = list(dataset.vol_res)
        self.exchange = None          # a parallel.MeshExchange while a sharded batch is running: avatar_frame pumps it behind its query launch
        self.after_query = []         # one-shot host work to run right behind the NEXT query launch (main.py: the previous frame's output section)
        self.lookahead_on_side_stream = os.environ.get('AVC_LOOKAHEAD_SIDE', '1') == '1'     # the next frame's U-Net beside this frame's tail (avatar_frame)
        self._side = None
        smpl_util.set_smpl_skinning_weights(dataset.body['skin_weights'])
        # How far from the body the bound vertices' candidate lists reach (csrc/knn_lbs.hip, avc_lbs_prepare) follows where a frame's SURFACE can lie: a
        # band dataset's marching-cubes vertices stay inside the valid band (0.1 m of the body: 140 mm with the cell's diagonal); a dense dataset's can
        # lie anywhere in the volume (the stress frame of bench.py: half of its 1.9 M vertices farther than 0.19 m, the farthest 0.8 m), and a lane
        # without a list pays the grid search -- lists everywhere: 1.7 GB, 15 ms once per sequence, LBS 0.93 -> 0.31 ms per frame, same bits
        # (tools/lbs_reach_dense.py).  `lbs_reach_mm` overrides.
        if lbs_reach_mm is None:
            lbs_reach_mm = 1000 if getattr(dataset, 'valid_mode', None) == 'dense' else 140
        self.lbs_reach_mm = int(lbs_reach_mm)
        if torch.device(config.device).type == 'cuda':
            _lib.set_option('lbs_reach_mm', self.lbs_reach_mm, config.device)
        smpl_util.set_cano_smpl_vertices(dataset.cano_smpl_v)                 # main.py:335

    def _grid_items(self, items: dict):
        """'dense' / 'band' when items['cano_pts'] IS the dataset's own point tensor (one frame, every grid point / the valid band in the order of
        dataset.infer_pts) -- the grid entry points then generate the same coordinates from the index --, else None (any other points: point queries)."""
        pts = items['cano_pts']
        ds = self.ds
        if pts.shape[0] != 1 or getattr(ds, 'infer_pts', None) is None or pts.data_ptr() != ds.infer_pts.data_ptr() or getattr(ds, 'grid_axes', None) is None:
            return None
        if getattr(ds, 'valid_mode', None) == 'dense' and pts.shape[1] == ds.valid_u8.numel():
            return 'dense'
        if getattr(ds, 'valid_idx', None) is not None and pts.shape[1] == ds.valid_idx.numel():
            return 'band'
        return None

    @torch.no_grad()
    def avatar_frame(self, items: dict, skin=True, next_items: dict | None = None):
        """1. geometric avatar in canonical space (main.py:357-367) + skinning to live space (:383-389).

        `next_items`: the frame that will be passed next, if the caller knows it (a sequence does).  Its pose feature map -- the U-Net, some
        seventy small launches that the host, not the device, paces (1.4 ms of wall clock for 0.7 ms of kernels) -- is then enqueued right
        behind this frame's query, while the host has nothing else to do, instead of at the start of the next call.  Same kernels, same
        inputs, same results; the next call recognises its input tensor and skips the U-Net."""
        wf = self.network.warping_field
        pre, self._next_map = getattr(self, '_next_map', None), None
        if pre is not None and pre[0] is items['smpl_pos_map']:
            wf.pose_feat_map, wf._map_on_device = pre[1], None
        else:
            with _stage('avc/unet7ds'):
                wf.precompute_conv(items)                                    # :359
        with _stage('avc/avatar_query'):
            out = self._avatar_query(items)
        side_done = None
        if next_items is not None:
            with _stage('avc/unet7ds (next frame)'):
                x = next_items['smpl_pos_map']
                if self.lookahead_on_side_stream and x.is_cuda:
                    # behind the query on a stream of its own: its 18 small launches then run BESIDE this frame's marching cubes / LBS (small launches
                    # too) instead of in front of them; this call's stream joins it before it returns
                    if self._side is None:
                        self._side = torch.cuda.Stream(x.device)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(x.device))           # = the query has finished
                    with torch.cuda.stream(self._side):
                        self._side.wait_event(ev)
                        m = wf.unet(x).contiguous()
                        done = torch.cuda.Event()
                        done.record(self._side)
                    self._next_map = (x, m)
                    side_done = done
                else:
                    self._next_map = (x, wf.unet(x).contiguous())
        if self.exchange is not None:
            # the query is enqueued and the host is idle until marching cubes' wait: the PREVIOUS frame's mesh goes out now, beside this frame's
            # kernels (parallel.MeshExchange.pump -- waits for the peers' counts of that step, never for this stream)
            with _stage('avc/mesh_exchange pump'):
                self.exchange.pump()
        self.run_after_query()
        try:
            with _stage('avc/marching_cubes'):
                vol = fill_volume(out['cano_pts_ov'][0, :, 0], self.ds.valid_u8, self.ds.invalid_pts_ov)   # :362-364
                v, f, n = recon_util.recon_mesh_device(vol, self.vol_res, self.ds.cano_bounds, iso_value=config.iso_value)   # :367
            res = {'cano_v': v, 'cano_vn': n, 'f': f, 'occ_volume': vol}
            if skin and v.shape[0] > 0:
                with _stage('avc/lbs'):
                    # :385 calculate_lbs, :386 skinning(.., True), :389 skinning_normal (einsum with vert_mats[:, :3, :3]) -- one launch, same bits
                    live_v, live_n, mats, _ = smpl_util.lbs_skinning(v[None], n[None], items['cano2live_jnt_mats'], return_pt_mats=True)
                res.update({'live_v': live_v[0], 'live_vn': live_n[0], 'vert_mats': mats[0]})
        finally:
            if side_done is not None:
                # whatever is enqueued after this call -- the next frame's query on the map, another pass of the U-Net (colour path, a frame that is not
                # the announced one) on the same launch plan -- comes behind the side stream's work; marching cubes and LBS above are already in the queue
                cur = torch.cuda.current_stream(self._next_map[1].device)
                cur.wait_event(side_done)
                self._next_map[1].record_stream(cur)
        return res

    def run_after_query(self):
        """Host work parked for the shadow of a query launch (`after_query`): the one place of the frame where the device has ~10 ms of work queued and the host
        nothing to do.  main.py assembles and hands over the PREVIOUS frame's outputs here: between two frames the device's queue is empty (marching cubes has
        just read its counts), and every host microsecond there is a device microsecond lost (1.3 - 2.4 ms per frame with PLY outputs, profiles/r06_main_e2e.md)."""
        hooks, self.after_query = self.after_query, []
        for fn in hooks:
            fn()

    def _avatar_query(self, items: dict):
        if self._grid_items(items) == 'dense':
            # every grid point is queried: the kernel generates the points from the grid index (no 12 B/point read), the offsets,
            # which :360-364 never read, are not written, and -- when the last axis holds a multiple of 128 points -- the 64 pose-feature
            # columns of conv1 / conv5 enter as one fp32 vector per (x, y) column (fused_mlp.hip: column folding; ~1e-6 from the
            # point-by-point query, bit-identical to it otherwise: tests/test_gpu_query.py)
            out = self.occ_net.query_grid(items, self.ds.grid_axes, self.vol_res)
        elif self._grid_items(items) == 'band':
            # the dataset's valid band (items['cano_pts'] IS dataset.infer_pts): the same points by their grid indices -- no coordinates read, column-folded
            out = self.occ_net.query_grid(items, self.ds.grid_axes, self.vol_res, index=self.ds.valid_idx)
        else:
            out = self.occ_net.query(items)                                  # :360
        return out

    @torch.no_grad()
    def avatar_frame_sharded(self, items: dict, group=None, skin=True):
        """Latency mode (optional): ONE frame across the ranks of `group`.  Every rank evaluates the fused
        query on its slab of the valid points, the occupancy slabs are all-gathered once (67 MB at 256^3), and every rank
        extracts the same mesh.  The throughput path (frames sharded, parallel.shard_frames) is what bench.py measures."""
        import torch.distributed as dist
        from .parallel import shard_range, all_gather_slabs
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.network.warping_field.precompute_conv(items)
        n = items['cano_pts'].shape[1]
        lo, hi = shard_range(n, rank, world)
        if hi <= lo:
            local = torch.empty(0, device=items['cano_pts'].device)
        elif getattr(self.ds, 'valid_idx', None) is not None and n == self.ds.valid_idx.numel() and items['cano_pts'].data_ptr() == self.ds.infer_pts.data_ptr():
            local = self.occ_net.query_grid(items, self.ds.grid_axes, self.vol_res, index=self.ds.valid_idx[lo:hi].contiguous())['cano_pts_ov'][0, :, 0]   # as avatar_frame
        else:
            sub = dict(items); sub['cano_pts'] = items['cano_pts'][:, lo:hi].contiguous()
            local = self.occ_net.query(sub)['cano_pts_ov'][0, :, 0]
        values = all_gather_slabs(local, n, group)
        vol = fill_volume(values, self.ds.valid_u8, self.ds.invalid_pts_ov)
        v, f, nrm = recon_util.recon_mesh_device(vol, self.vol_res, self.ds.cano_bounds, iso_value=config.iso_value)
        res = {'cano_v': v, 'cano_vn': nrm, 'f': f, 'occ_volume': vol}
        if skin and v.shape[0] > 0:
            live_v, live_n, mats, _ = smpl_util.lbs_skinning(v[None], nrm[None], items['cano2live_jnt_mats'], return_pt_mats=True)
            res.update({'live_v': live_v[0], 'live_vn': live_n[0], 'vert_mats': mats[0]})
        return res

    @torch.no_grad()
    def recon_frame(self, items: dict):
        """3. reconstruction network (main.py:438-453); items must hold front_normal / back_normal."""
        kind = self._grid_items(items)
        rn = self.recon_net
        center = _host_center(items)                                                               # by value into the C-ABI: no read-back of a device tensor
        with _stage('avc/hgfilter'):
            imgs = torch.cat([items['front_normal'], items['back_normal']], dim=1)                 # arch_recon.py:51-53
            img_feat_map = rn.bind_feat_map(imgs)                                                  # the encoder's channel-last output IS the decoder's map
        with _stage('avc/recon_query'):
            if kind == 'dense':
                out = rn.decode_grid(self.ds.grid_axes, self.vol_res, img_feat_map, center)                      # :440, on the grid (column-folded)
            elif kind == 'band':
                out = rn.decode_grid(self.ds.grid_axes, self.vol_res, img_feat_map, center, index=self.ds.valid_idx)
            else:
                out = rn.decode(items['cano_pts'].contiguous(), img_feat_map, center)                             # :440
        with _stage('avc/marching_cubes'):
            vol = fill_volume(out[0], self.ds.valid_u8, self.ds.invalid_pts_ov)                    # :442-443 (output[0])
            v, f, n = recon_util.recon_mesh_device(vol, self.vol_res, self.ds.cano_bounds)         # :444 (iso 0.5)
        res = {'cano_v': v, 'cano_vn': n, 'f': f, 'occ_volume': vol}
        if v.shape[0] > 0:
            with _stage('avc/lbs'):
                live_v, live_n, _, _ = smpl_util.lbs_skinning(v[None], n[None], items['cano2live_jnt_mats'])   # :451 calculate_lbs, :452 skinning, :453 skinning_normal
                res['live_v'], res['live_vn'] = live_v[0], live_n[0]
        return res

    @torch.no_grad()
    def avatarcap_frame(self, items: dict, observed_normal: torch.Tensor, w2c_RT, cam: dict, integrate_manner: str = 'merge', iter_num: int = 100,
                        next_items: dict | None = None):
        """Steps 1-3 of the reference's loop body for one frame (main.py:357-453; BASELINE configs[2], "AvatarCap full"): avatar geometry + skinning,
        canonical normal fusion with the image-observed normal map, HGFilter + reconstruction query + marching cubes + skinning."""
        a = self.avatar_frame(items, next_items=next_items)
        items = dict(items)
        if a['cano_v'].shape[0] > 0:
            with _stage('avc/normal_fusion'):
                items['front_normal'], items['back_normal'], _ = self.fuse_normals(a, observed_normal, w2c_RT, cam, integrate_manner, iter_num)
        else:
            items['front_normal'], items['back_normal'] = self.cano_normal_maps(a['cano_v'], a['cano_vn'], a['f'])
        return a, self.recon_frame(items)


    @torch.no_grad()
    def cano_normal_maps(self, cano_v, cano_vn, faces, size=512):
        """Front / back canonical normal maps of a mesh (visualize_util.render_cano_mesh, main.py:369):
        (1,3,H,W) tensors in the layout ReconNetwork.infer expects (main.py:432-433)."""
        from .utils.visualize_util import render_cano_mesh_device
        fr, bk = render_cano_mesh_device(cano_v, cano_vn, faces, self.ds.cano_smpl_center, size)
        return fr.permute(2, 0, 1)[None].contiguous(), bk.permute(2, 0, 1)[None].contiguous()

    @torch.no_grad()
    def fuse_normals(self, avatar: dict, observed_normal: torch.Tensor, w2c_RT, cam: dict, integrate_manner: str = 'merge', iter_num: int = 100):
        """2. canonical normal fusion (main.py:405-429): the image-observed normal map (H,W,3, device) is carried to the
        canonical pose through the posed avatar mesh (`avatar` = the dict avatar_frame returned, with live_v / vert_mats)
        and fused into the avatar's front map; the back map is the avatar's own (:427).  Returns (1,3,512,512) tensors."""
        from .normal_fusion.normal_fusion import canonicalize_normal_map_device, merge_normal_images_device, merge_normal_images_cover_device
        from .utils.visualize_util import render_cano_mesh_device
        center = self.ds.cano_smpl_center
        front_avatar, back_avatar = render_cano_mesh_device(avatar['cano_v'], avatar['cano_vn'], avatar['f'], center, 512)           # :369
        front_image, _ = canonicalize_normal_map_device(avatar['cano_v'], avatar['live_v'], avatar['f'], observed_normal, avatar['vert_mats'],
                                                        np.asarray(w2c_RT, np.float32), cam['fx'], cam['fy'], cam['cx'], cam['cy'], center)   # :413-415
        if integrate_manner == 'merge':
            cv = smpl_util.cano_smpl_vertices
            nk = getattr(self, '_neck', None)       # per sequence, not per frame (the read drains the stream); keyed on the tensor ITSELF (kept alive here) and its
            if nk is None or nk[0] is not cv or nk[1] != cv._version:      # version -- a device address is handed out again once the old vertex set is freed
                self._neck = nk = (cv, cv._version, cv[3068].cpu().numpy())
            neck_vert = nk[2] - np.asarray(center, np.float32)                                                                  # :418
            neck_y = int((1. - neck_vert[1]) / 2. * 512)                                                                                # :419
            neck_x = int((neck_vert[0] - 1) / 2. * 512)                                                                                 # :420 (negative: wraps, like the reference's slice)
            front = merge_normal_images_device(front_avatar, front_image, iter_num, (neck_x, neck_y))                                   # :421
        elif integrate_manner == 'cover':
            front = merge_normal_images_cover_device(front_avatar, front_image)                                                          # :423
        else:
            raise ValueError('Invalid integration manner!')                                                                              # :425
        return front.permute(2, 0, 1)[None].contiguous(), back_avatar.permute(2, 0, 1)[None].contiguous(), front_image

    @torch.no_grad()
    def full_frame(self, items: dict):
        """Steps 1 and 3 chained on the device: avatar geometry -> its normal maps -> reconstruction.
        Step 2 of the reference (fusion with image-observed normals, normal_fusion.py) needs a captured
        image and is not part of this path: the avatar's own maps are passed through, which is what
        `merge_normal_images_cover` yields when no pixel is observed (normal_fusion.py:158-167)."""
        a = self.avatar_frame(items)
        items = dict(items)
        items['front_normal'], items['back_normal'] = self.cano_normal_maps(a['cano_v'], a['cano_vn'], a['f'])
        r = self.recon_frame(items)
        return a, r

    @torch.no_grad()
    def transfer_colours(self, vertices: torch.Tensor, vertices_avatar: torch.Tensor, colour_avatar: torch.Tensor):
        """Colour of the nearest avatar vertex for every reconstructed vertex (main.py:478-482:
        knn_points(vertices, vertices_avatar) with K = 1, then knn_gather)."""
        _, idx = smpl_util.knn_points(vertices[None].contiguous(), vertices_avatar[None].contiguous(), K=1)
        return colour_avatar[idx[0, :, 0]]

    @torch.no_grad()
    def colour_vertices(self, items: dict, cano_v: torch.Tensor, cano_vn: torch.Tensor, renderer=None):
        """4. vertex colours from the texture template (main.py:464-477): one 64-sample ray per vertex,
        starting at v + n and marching along -n, alpha-composited.  Returns (V,3) in the reference's BGR order."""
        from .network.arch_avatar import NerfRenderer
        renderer = renderer or NerfRenderer(self.network)
        items = dict(items)
        items['ray_o'] = (cano_v + cano_vn)[None]
        items['ray_d'] = -cano_vn[None]
        items['depth'] = torch.ones((1, cano_v.shape[0]), dtype=cano_v.dtype, device=cano_v.device)
        items['near'] = items['depth'] - 0.05
        items['far'] = items['depth'] + 0.05
        items['occupancy'] = items['depth'].clone()
        renderer.net.warping_field.precompute_conv(items)                                           # :474 (nerf_renderer.net: the finetuned copy when one is loaded)
        with _stage('avc/colour_vertices'):
            out = renderer.render(items, pts_space='cano', near_dist=0.02, far_dist=0.05)           # :475
        return out['rgb_map'][0][:, [2, 1, 0]]                                                      # :476
