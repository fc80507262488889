"""ctypes binding of libavcap_hip.so (include/avcap.h).

There is no fallback: if the library is missing or no gfx950 device is present every entry point
raises.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('AVCAP_LIB', os.path.join(_HERE, 'libavcap_hip.so'))   # override only for kernel A/B experiments

AVC_ERR_CAPACITY = -4
AVC_ERR_RANGE = -5


class avc_dense(C.Structure):
    _fields_ = [('w', C.c_void_p), ('b', C.c_void_p), ('g', C.c_void_p), ('cout', C.c_int32), ('cin', C.c_int32)]


class avc_bn(C.Structure):
    _fields_ = [('gamma', C.c_void_p), ('beta', C.c_void_p), ('mean', C.c_void_p), ('var', C.c_void_p), ('eps', C.c_float)]


class avc_conv2d(C.Structure):
    _fields_ = [('w', C.c_void_p), ('b', C.c_void_p), ('cout', C.c_int32), ('cin', C.c_int32), ('kh', C.c_int32), ('kw', C.c_int32)]


class avc_groupnorm(C.Structure):
    _fields_ = [('gamma', C.c_void_p), ('beta', C.c_void_p), ('channels', C.c_int32), ('groups', C.c_int32), ('eps', C.c_float)]


class avc_convblock(C.Structure):
    _fields_ = [('conv', avc_conv2d * 3), ('downsample', avc_conv2d), ('bn', avc_groupnorm * 4)]


class avc_hgfilter(C.Structure):
    _fields_ = [('conv1', avc_conv2d), ('bn1', avc_groupnorm), ('conv2', avc_convblock), ('conv3', avc_convblock), ('conv4', avc_convblock),
                ('depth', C.c_int32), ('hourglass', C.POINTER(avc_convblock)), ('top_m', avc_convblock),
                ('conv_last', avc_conv2d), ('bn_end', avc_groupnorm), ('l', avc_conv2d)]


class avc_bn2d(C.Structure):
    _fields_ = [('mean', C.c_void_p), ('var', C.c_void_p), ('eps', C.c_float)]


class avc_unet7ds(C.Structure):
    _fields_ = [('down', avc_conv2d * 7), ('down_bn', avc_bn2d * 7), ('up', avc_conv2d * 3), ('up_bn', avc_bn2d * 3),
                ('upc', avc_conv2d * 3), ('upc_bn', avc_bn2d * 3)]


_SIGNATURES = {
    'avc_last_error': (C.c_char_p, []),
    'avc_version': (C.c_int, []),
    'avc_ctx_create': (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    'avc_ctx_destroy': (C.c_int, [C.c_void_p]),
    'avc_pack_warp_weights': (C.c_int, [C.c_void_p, C.POINTER(avc_dense), C.POINTER(avc_bn), C.POINTER(avc_dense), C.c_int]),
    'avc_pack_template_weights': (C.c_int, [C.c_void_p, C.POINTER(avc_dense), C.POINTER(avc_dense), C.POINTER(avc_dense), C.c_int]),
    'avc_pack_recon_weights': (C.c_int, [C.c_void_p, C.POINTER(avc_dense)]),
    'avc_set_pose_feat_map': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'avc_set_img_feat_map': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'avc_avatar_query': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'avc_avatar_query_grid': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    'avc_avatar_query_grid_subset': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_int64, C.POINTER(C.c_float),
                                               C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'avc_set_range_check': (C.c_int, [C.c_void_p, C.c_int]),
    'avc_template_query': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'avc_recon_query': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.c_void_p, C.c_void_p]),
    'avc_recon_query_grid': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_void_p, C.c_void_p]),
    'avc_recon_query_grid_subset': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_int64,
                                              C.POINTER(C.c_float), C.c_void_p, C.c_void_p]),
    'avc_hgfilter_pack': (C.c_int, [C.c_void_p, C.POINTER(avc_hgfilter)]),
    'avc_hgfilter_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'avc_render_rays_cano': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_int64, C.c_int,
                                       C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, C.c_int32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    'avc_blend_weight_sample': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'avc_unet_pack': (C.c_int, [C.c_void_p, C.POINTER(avc_unet7ds)]),
    'avc_unet_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'avc_hgfilter_debug_tensor': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                            C.c_void_p]),
    'avc_group_norm': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                 C.c_int, C.c_void_p, C.c_void_p]),
    'avc_scatter_volume': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'avc_recon_mesh': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_float, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
    'avc_render_cano_maps': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    'avc_render_mesh': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p]),
    'avc_canonicalize_normals': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    'avc_merge_normal_images': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'avc_merge_normal_images_cover': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'avc_knn': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'avc_calculate_lbs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'avc_lbs_prepare': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    'avc_calculate_lbs_bound': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    'avc_lbs_bound_stats': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    'avc_lbs_skin_bound': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'avc_skinning': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'avc_timing_enable': (C.c_int, [C.c_void_p, C.c_int]),
    'avc_timing_read': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    'avc_timing_read_cycles': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    'avc_set_option': (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
}

_lib = None
_ctxs: dict[int, int] = {}


class AvcapError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f'libavcap_hip: {message} (status {status})')
        self.status = status


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    """The loaded library (raises if it has not been built: run `python -m avatarcap_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} is missing: build it with `python -m avatarcap_amd.build` '
                               '(there is no CPU/eager fallback for the hot path)')
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(status: int):
    if status != 0:
        raise AvcapError(status, lib().avc_last_error().decode())


def ctx(device=None) -> int:
    """Context handle of a HIP device (one per device, created on first use)."""
    if device is None:
        device = torch.cuda.current_device()
    elif isinstance(device, torch.device):
        if device.type != 'cuda':
            raise RuntimeError(f'avatarcap_amd hot path needs a HIP (cuda) device, got {device}')
        device = device.index if device.index is not None else torch.cuda.current_device()
    if device not in _ctxs:
        h = C.c_void_p()
        check(lib().avc_ctx_create(int(device), C.byref(h)))
        _ctxs[device] = h.value
        for kv in filter(None, os.environ.get('AVC_OPTIONS', '').split(',')):       # e.g. AVC_OPTIONS=enc_graph=0,enc_fork=0: avc_set_option on every new context (bisecting aid)
            k, _, v = kv.partition('=')
            check(lib().avc_set_option(h.value, k.strip().encode(), int(v)))
    return _ctxs[device]


# A context holds ONE packed avatar network, ONE packed recon decoder and ONE bound feature map of each kind, while the host-side
# modules cache "I have packed / bound already".  Two modules with different weights on the same device would otherwise evaluate
# with each other's weights: every slot records which module (and which version of it) the context currently holds.
_owners: dict[tuple[int, str], object] = {}


def owns(ctx_handle: int, slot: str, token) -> bool:
    return _owners.get((ctx_handle, slot)) == token


def set_owner(ctx_handle: int, slot: str, token) -> None:
    _owners[(ctx_handle, slot)] = token


def apply_range_check(ctx_handle: int) -> None:
    """config.check_range -> avc_set_range_check (include/avcap.h 'numeric range')."""
    from . import config
    check(lib().avc_set_range_check(ctx_handle, 1 if getattr(config, 'check_range', False) else 0))


def set_option(name: str, value: int, device=None) -> None:
    """avc_set_option on the device's context (include/avcap.h 'switches of a context')."""
    check(lib().avc_set_option(ctx(device), name.encode(), int(value)))


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def dev_ptr(t: torch.Tensor | None, dtype=torch.float32, name='tensor') -> int | None:
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name}: expected a torch.Tensor, got {type(t).__name__}')
    if t.device.type != 'cuda':
        raise RuntimeError(f'{name}: must live on the HIP device, got {t.device}')
    if t.dtype != dtype:
        raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name}: must be contiguous')
    return t.data_ptr()


def f3(v) -> C.Array:
    a = np.asarray(v.detach().cpu() if isinstance(v, torch.Tensor) else v, np.float32).reshape(-1)
    return (C.c_float * len(a))(*a.tolist())


def host_f3(batch: dict, key: str, b: int = 0) -> C.Array:
    """batch[key][b] as C floats.  The reference keeps such per-sequence constants (`cano_smpl_center`, `cano_bounds`) as device tensors of the item dict and
    the C-ABI takes them by value: reading one back costs a drain of the stream per query.  Item dicts made by this package's loaders carry the host
    arrays they were uploaded from under '_host' (dataset.to_cuda, frame_io.FramePrefetcher); any other dict takes the `.cpu()`."""
    a = host_mirror(batch, key)
    if a is not None and batch[key].shape[0] == 1:                                      # the un-batched host array of a one-frame batch
        return f3(a)
    return f3(batch[key][b])


def host_mirror(batch, key):
    """The host array batch[key] was uploaded from, or None: the dict must carry it under '_host' AND batch[key] must still be the very tensor the loader
    put there ('_host_ids': a caller that replaces an entry of the dict gets its own tensor read, not a stale mirror)."""
    if not isinstance(batch, dict):
        return None
    h, ids = batch.get('_host'), batch.get('_host_ids')
    if h is None or ids is None or key not in h or ids.get(key) != id(batch.get(key)):
        return None
    v = h[key]
    if isinstance(v, torch.Tensor):
        if v.is_cuda:
            return None
        v = v.detach().numpy()
    a = np.asarray(v, np.float32)
    return a if a.size == batch[key].numel() else None


class TensorWatch:
    """Has anything a module's packed weights were made from changed?  The modules re-pack when a parameter or buffer was edited in place (`_version`), moved or
    re-assigned (`data_ptr`, `id`) -- a check that runs at every query, at the START of a frame, where the device's queue is empty and host time is device time.
    `tuple(p._version for p in module.parameters())` walks the module tree (0.1 - 0.7 ms for the networks here, three times per signature); this resolves the
    (owner dict, key) slot of every parameter and buffer ONCE and re-reads the slots: ~30 us.  The slots are those of the module tree at construction -- the
    reference never adds or replaces sub-modules of a built network; a tensor re-assigned in its slot IS seen (the slot is read, not a cached tensor)."""

    def __init__(self, module, include=None):
        self.slots = []
        for name, m in module.named_modules():
            if include is not None and not include(name):
                continue
            for d in (m._parameters, m._buffers):
                for k in d:
                    self.slots.append((d, k))

    def signature(self):
        out = []
        for d, k in self.slots:
            t = d.get(k)
            out.append(None if t is None else (id(t), t._version, t.data_ptr()))
        return tuple(out)


# ---- weight marshalling --------------------------------------------------------------------
def _host(t) -> np.ndarray:
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    return np.ascontiguousarray(a, np.float32)


class DenseList:
    """Keeps the host arrays alive while the C side reads them."""

    def __init__(self, entries):
        self.keep = []
        arr = (avc_dense * len(entries))()
        for i, e in enumerate(entries):
            w = _host(e['w']); b = _host(e['b'])
            w = w.reshape(w.shape[0], -1)
            g = _host(e['g']).reshape(-1) if e.get('g') is not None else None
            self.keep += [w, b, g]
            arr[i].w = w.ctypes.data; arr[i].b = b.ctypes.data
            arr[i].g = g.ctypes.data if g is not None else None
            arr[i].cout, arr[i].cin = w.shape
        self.arr = arr


class BnList:
    def __init__(self, entries):
        self.keep = []
        arr = (avc_bn * len(entries))()
        for i, e in enumerate(entries):
            hs = [_host(e[k]) for k in ('gamma', 'beta', 'mean', 'var')]
            self.keep += hs
            arr[i].gamma, arr[i].beta, arr[i].mean, arr[i].var = (h.ctypes.data for h in hs)
            arr[i].eps = float(e['eps'])
        self.arr = arr


class UNetWeights:
    """avc_unet7ds for a module tree shaped like the reference's UnetNoCond7DS (network/unets.py:169-199); keeps the host arrays alive.
    `upconv4` exists in checkpoints and is never applied (unets.py:213-214): it is not marshalled."""

    def __init__(self, m):
        self.keep = []
        u = avc_unet7ds()
        for i in range(7):
            blk = getattr(m, f'conv{i + 1}')
            self._conv(u.down[i], blk.conv.weight, None, transposed=False)
            self._bn(u.down_bn[i], getattr(blk, 'bn', None))
        for i in range(3):
            blk = getattr(m, f'upconv{i + 1}')
            if blk.up.bias is not None:
                raise NotImplementedError('UnetNoCond7DS: the transposed convolutions carry no bias (unets.py:45)')
            self._conv(u.up[i], blk.up.weight, None, transposed=True)
            self._bn(u.up_bn[i], getattr(blk, 'bn', None))
        for i, n in enumerate(('upconvC5', 'upconvC6', 'upconvC7')):
            blk = getattr(m, n)
            self._conv(u.upc[i], blk.up[1].weight, blk.up[1].bias, transposed=False)
            self._bn(u.upc_bn[i], getattr(blk, 'bn', None))
        self.struct = u

    def _arr(self, t):
        a = _host(t)
        self.keep.append(a)
        return a.ctypes.data

    def _conv(self, dst, w, b, transposed):
        dst.w = self._arr(w)
        dst.b = self._arr(b) if b is not None else None
        o, i, kh, kw = (int(v) for v in w.shape)
        dst.cout, dst.cin = (i, o) if transposed else (o, i)          # ConvTranspose2d stores (in, out, kh, kw)
        dst.kh, dst.kw = kh, kw

    def _bn(self, dst, bn):
        if bn is None:
            dst.mean = dst.var = None
            dst.eps = 0.0
            return
        if bn.affine or bn.running_mean is None:
            raise NotImplementedError('UnetNoCond7DS: BatchNorm2d(affine=False) with running statistics (unets.py:26, :58)')
        dst.mean, dst.var, dst.eps = self._arr(bn.running_mean), self._arr(bn.running_var), float(bn.eps)


class HGFilterWeights:
    """avc_hgfilter for a module tree shaped like the reference's HGFilter (network/HGFilters.py:124-175); keeps the host arrays alive."""

    def __init__(self, m):
        self.keep = []
        h = avc_hgfilter()
        self._conv(h.conv1, m.conv1)
        self._norm(h.bn1, m.bn1)
        for name in ('conv2', 'conv3', 'conv4'):
            self._block(getattr(h, name), getattr(m, name))
        hg = m.m0
        d = hg.depth
        names = [f'b{k}_{lvl}' for lvl in range(d, 0, -1) for k in (1, 2)] + ['b2_plus_1'] + [f'b3_{lvl}' for lvl in range(1, d + 1)]
        self.blocks = (avc_convblock * len(names))()
        for i, n in enumerate(names):
            self._block(self.blocks[i], hg._modules[n])
        h.depth = d
        h.hourglass = C.cast(self.blocks, C.POINTER(avc_convblock))
        self._block(h.top_m, m.top_m_0)
        self._conv(h.conv_last, m.conv_last0)
        self._norm(h.bn_end, m.bn_end0)
        self._conv(h.l, m.l0)
        self.struct = h

    def _arr(self, t):
        a = _host(t)
        self.keep.append(a)
        return a.ctypes.data

    def _conv(self, dst, conv):
        w = conv.weight
        dst.w = self._arr(w)
        dst.b = self._arr(conv.bias) if conv.bias is not None else None
        dst.cout, dst.cin, dst.kh, dst.kw = (int(v) for v in w.shape)

    def _norm(self, dst, gn):
        if not isinstance(gn, torch.nn.GroupNorm) or gn.weight is None:
            raise NotImplementedError("the HIP encoder implements norm='group' with affine parameters (what ReconNetwork builds)")
        dst.gamma, dst.beta = self._arr(gn.weight), self._arr(gn.bias)
        dst.channels, dst.groups, dst.eps = int(gn.num_channels), int(gn.num_groups), float(gn.eps)

    def _block(self, dst, blk):
        for i, (cv, bn) in enumerate(((blk.conv1, blk.bn1), (blk.conv2, blk.bn2), (blk.conv3, blk.bn3))):
            self._conv(dst.conv[i], cv)
            self._norm(dst.bn[i], bn)
        if blk.downsample is not None:
            self._conv(dst.downsample, blk.downsample[2])
            self._norm(dst.bn[3], blk.bn4)
