"""Host-side mirror of the reference's `utils/smpl_util.py` (SmplUtil, :12-84) on libavcap_hip.so.

The reference builds a module-level singleton at import from the licensed SMPL pickle
(`smpl_params.weights`, smpl_util.py:14,84).  Here the singleton starts empty and the (6890,24)
skinning weights are supplied with `set_smpl_skinning_weights` (real SMPL or the synthetic body).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import config
from .. import _lib


import itertools

_bind_tokens = itertools.count(1)


class SmplUtil:
    def __init__(self, smpl_skinning_weights=None):
        self.smpl_skinning_weights = None
        if smpl_skinning_weights is not None:
            self.set_smpl_skinning_weights(smpl_skinning_weights)
        self.cano_smpl_pose = np.zeros(75, dtype=np.float32)               # smpl_util.py:16-18
        self.cano_smpl_pose[3 + 3 * 1 + 2] = math.radians(25)
        self.cano_smpl_pose[3 + 3 * 2 + 2] = math.radians(-25)
        self.cano_smpl_vertices = None

    def set_smpl_skinning_weights(self, w):
        w = torch.as_tensor(np.asarray(w.cpu() if isinstance(w, torch.Tensor) else w, np.float32))
        self.smpl_skinning_weights = w.to(torch.float32).to(config.device).contiguous()

    def set_cano_smpl_vertices(self, cano_smpl_vertices: torch.Tensor):
        """smpl_util.py:21.  On the HIP device the vertices are also BOUND to the context (avc_lbs_prepare: their search grid and the per-cell candidate
        lists are built once per sequence, not once per calculate_lbs call)."""
        self.cano_smpl_vertices = cano_smpl_vertices.to(torch.float32).to(config.device).contiguous()
        self._bound = None
        v = self.cano_smpl_vertices
        if v.is_cuda and v.dim() == 2 and v.shape[0] >= 4:      # what calculate_lbs binds; anything else stays unbound and takes the search per call,
            try:                                                  # and the setter never raises (smpl_util.py:21-22): a set that cannot be bound (non-finite
                self._bind(v.device)                              # vertices) is reported by the calculate_lbs that needs it
            except _lib.AvcapError:
                self._bound = None

    def _bind(self, device):
        """The device context holds ONE bound vertex set.  The binder keeps a token of its own (a process-wide counter: neither `id()` nor a device address,
        both of which are recycled) together with the tensor version it bound; whoever calculates next rebinds when the context's slot carries another
        token (another SmplUtil has bound since) or the vertices were edited in place."""
        v = self.cano_smpl_vertices
        ctx = _lib.ctx(device)
        b = getattr(self, '_bound', None)
        if b is None or b[1] is not v or b[2] != v._version or not _lib.owns(ctx, 'lbs_bound', b[0]):
            _lib.set_owner(ctx, 'lbs_bound', None)       # lbs_prepare overwrites the context's tables before it can fail: nobody owns them until it has succeeded
            _lib.check(_lib.lib().avc_lbs_prepare(ctx, _lib.dev_ptr(v, name='cano_smpl_vertices'), v.shape[0], _lib.stream_ptr(device)))
            token = next(_bind_tokens)
            _lib.set_owner(ctx, 'lbs_bound', token)
            self._bound = (token, v, v._version)
        return ctx

    # pytorch3d.ops.knn_points stand-in (squared distances ascending, indices int64)
    def knn_points(self, p1, p2, K=1):
        B, N, _ = p1.shape
        p1 = p1.contiguous(); p2 = p2.contiguous()
        d2 = torch.empty((B, N, K), dtype=torch.float32, device=p1.device)
        idx = torch.empty((B, N, K), dtype=torch.int64, device=p1.device)
        ctx = _lib.ctx(p1.device)
        for b in range(B):
            _lib.check(_lib.lib().avc_knn(ctx, _lib.dev_ptr(p1[b], name='p1'), N, _lib.dev_ptr(p2[b], name='p2'), p2.shape[1], K,
                                          d2[b].data_ptr(), idx[b].data_ptr(), _lib.stream_ptr(p1.device)))
        return d2, idx

    def _lbs(self, points, verts):
        if self.smpl_skinning_weights is None:
            raise ValueError('SMPL skinning weights are not set!')
        B, N, _ = points.shape
        points = points.contiguous()
        lbs = torch.empty((B, N, 24), dtype=torch.float32, device=points.device)
        ctx = _lib.ctx(points.device)
        for b in range(B):
            v = verts[b] if verts.dim() == 3 else verts
            _lib.check(_lib.lib().avc_calculate_lbs(ctx, _lib.dev_ptr(points[b], name='points'), N, _lib.dev_ptr(v.contiguous(), name='smpl_v'),
                                                    _lib.dev_ptr(self.smpl_skinning_weights, name='skin_w'), v.shape[0],
                                                    lbs[b].data_ptr(), _lib.stream_ptr(points.device)))
        return lbs

    def calculate_lbs(self, points):
        """(B,N,3) -> (B,N,24): blend weights of points around canonical SMPL (smpl_util.py:24-39)."""
        if self.cano_smpl_vertices is None:
            raise ValueError('Canonical smpl vertices are invalid!')       # smpl_util.py:30-31
        v = self.cano_smpl_vertices
        if points.is_cuda and v.device == points.device and v.dim() == 2 and v.shape[0] >= 4 and self.smpl_skinning_weights is not None:
            ctx = self._bind(points.device)
            B, N, _ = points.shape
            points = points.contiguous()
            lbs = torch.empty((B, N, 24), dtype=torch.float32, device=points.device)
            for b in range(B):
                _lib.check(_lib.lib().avc_calculate_lbs_bound(ctx, _lib.dev_ptr(points[b], name='points'), N, _lib.dev_ptr(self.smpl_skinning_weights, name='skin_w'),
                                                              lbs[b].data_ptr(), _lib.stream_ptr(points.device)))
            return lbs
        return self._lbs(points, v)

    def calculate_lbs2(self, points, smpl_v):
        """Same against given vertices (B,N',3) (smpl_util.py:41-56)."""
        return self._lbs(points, smpl_v)

    def _skin(self, points, normals, lbs, jnt_mats, want_mats):
        ref = points if points is not None else normals
        B, N, _ = ref.shape
        lbs = lbs.contiguous(); jnt_mats = jnt_mats.contiguous()
        po = torch.empty_like(ref) if points is not None else None
        no = torch.empty_like(ref) if normals is not None else None
        mo = torch.empty((B, N, 4, 4), dtype=torch.float32, device=ref.device) if want_mats else None
        ctx = _lib.ctx(ref.device)
        for b in range(B):
            _lib.check(_lib.lib().avc_skinning(
                ctx, _lib.dev_ptr(points[b].contiguous(), name='points') if points is not None else None,
                _lib.dev_ptr(normals[b].contiguous(), name='normals') if normals is not None else None, N,
                _lib.dev_ptr(lbs[b], name='lbs'), _lib.dev_ptr(jnt_mats[b], name='jnt_mats'),
                po[b].data_ptr() if po is not None else None, no[b].data_ptr() if no is not None else None,
                mo[b].data_ptr() if mo is not None else None, _lib.stream_ptr(ref.device)))
        return po, no, mo

    def skinning(self, points, lbs, jnt_mats, return_pt_mats=False):
        """Forward skinning (smpl_util.py:58-74): points (B,N,3), lbs (B,N,24), jnt_mats (B,24,4,4)."""
        po, _, mo = self._skin(points.contiguous(), None, lbs, jnt_mats, return_pt_mats)
        return (po, mo) if return_pt_mats else po

    def skinning_normal(self, normals, lbs, cano2live_jnt_mats):
        """(smpl_util.py:76-81)"""
        return self._skin(None, normals.contiguous(), lbs, cano2live_jnt_mats, False)[1]

    def lbs_skinning(self, points, normals, jnt_mats, return_pt_mats=False, return_lbs=False):
        """What main.py:385-389 does with a frame's vertices -- `lbs = calculate_lbs(points)`, `skinning(points, lbs, jnt_mats, True)`,
        `skinning_normal(normals, lbs, jnt_mats)` -- as ONE launch on the bound vertices (avc_lbs_skin_bound): the same operations in the same order, the
        same bits, without the (B,N,24) weights going through HBM three times.  -> (live_points, live_normals | None, pt_mats | None, lbs | None).
        Anything the bound form does not serve (CPU tensors, batched vertices) is the three calls."""
        if self.cano_smpl_vertices is None:
            raise ValueError('Canonical smpl vertices are invalid!')       # smpl_util.py:30-31
        v = self.cano_smpl_vertices
        import os
        if os.environ.get('AVC_LBS_FUSED', '1') == '0' or not (     # AVC_LBS_FUSED=0: the three calls (A/B aid)
            points.is_cuda and v.device == points.device and v.dim() == 2 and v.shape[0] >= 4 and self.smpl_skinning_weights is not None):
            lbs = self.calculate_lbs(points)
            po, mo = self.skinning(points, lbs, jnt_mats, True)
            no = self.skinning_normal(normals, lbs, jnt_mats) if normals is not None else None
            return po, no, (mo if return_pt_mats else None), (lbs if return_lbs else None)
        ctx = self._bind(points.device)
        B, N, _ = points.shape
        points = points.contiguous(); jnt_mats = jnt_mats.contiguous()
        normals = normals.contiguous() if normals is not None else None
        po = torch.empty_like(points)
        no = torch.empty_like(normals) if normals is not None else None
        mo = torch.empty((B, N, 4, 4), dtype=torch.float32, device=points.device) if return_pt_mats else None
        lbs = torch.empty((B, N, 24), dtype=torch.float32, device=points.device) if return_lbs else None
        for b in range(B):
            _lib.check(_lib.lib().avc_lbs_skin_bound(
                ctx, _lib.dev_ptr(points[b], name='points'), _lib.dev_ptr(normals[b], name='normals') if normals is not None else None, N,
                _lib.dev_ptr(self.smpl_skinning_weights, name='skin_w'), _lib.dev_ptr(jnt_mats[b], name='jnt_mats'),
                lbs[b].data_ptr() if lbs is not None else None, po[b].data_ptr(), no[b].data_ptr() if no is not None else None,
                mo[b].data_ptr() if mo is not None else None, _lib.stream_ptr(points.device)))
        return po, no, mo, lbs


smpl_util = SmplUtil()
