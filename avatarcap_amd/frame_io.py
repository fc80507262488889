"""The host side of `main.run_avatarcap`'s frame loop, taken off the critical path.

The reference's loop (main.py:348-351, 491-504) loads a frame's item dict, uploads it tensor by tensor (`to_cuda`, dataset/avatarcap_dataset.py:329-346), computes,
pulls every mesh back with `.cpu().numpy()` and writes the files, all on one thread -- the device waits while the host reads and writes.  Here:

  `FramePrefetcher`  a worker thread runs the dataset's `__getitem__` (file reads, SMPL forward, EXR decode) `depth` frames ahead, packs every host array of the
                     item into ONE pinned staging buffer and issues ONE `non_blocking` host-to-device copy on a copy stream; the device tensors of the item dict
                     are views into that one block.  The compute stream waits for the copy's event, never the host.
  `MeshWriter`       the finished frame's tensors are copied device-to-host into a pinned slot on the copy stream behind an event recorded on the compute stream;
                     writer threads wait for the copy's event and write the files (NumPy / file writes release the GIL).  A bounded number of slots is the
                     back-pressure: when the disk is slower than the device, `submit` waits for a slot -- the frame's kernels are already enqueued by then.

Neither class contains a host synchronisation of the compute stream.  On a CPU device (main.py --dry-run) both degrade to plain threads over host tensors.
"""
from __future__ import annotations

import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

_ALIGN = 256


def _align(n: int) -> int:
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN


def _torch_dtype(dt: np.dtype) -> torch.dtype:
    return torch.from_numpy(np.empty(0, dtype=dt)).dtype


class _Pinned:
    """A pinned byte buffer that only grows (hipHostMalloc costs milliseconds per 100 MB: geometric growth, never per frame)."""

    def __init__(self):
        self.buf = None
        self.event = None           # the last copy that read from / wrote to the buffer

    def ensure(self, nbytes: int):
        if self.buf is None or self.buf.numel() < nbytes:
            if self.event is not None:
                self.event.synchronize()
                self.event = None
            cap = max(nbytes, 1 << 20)
            if self.buf is not None:
                cap = max(cap, self.buf.numel() * 3 // 2)
            self.buf = torch.empty(_align(cap), dtype=torch.uint8, pin_memory=True)
        return self.buf


class FramePrefetcher:
    """Item dicts `depth` frames ahead of the loop.

    `load_host(i)` returns the frame's item dict as the dataset hands it over: NumPy arrays and CPU tensors (uploaded), device tensors and everything else
    (passed through).  `order` is the sequence of frame indices this rank will ask for.  `get(i)` returns what `to_cuda(load_host(i), add_batch)` returns,
    plus `'_host'`: the dict as loaded (camera matrices and the like are read on the host too -- no `.cpu()` of something that was just uploaded).
    An exception of `load_host(i)` is raised by `get(i)`; `peek(i)` returns None instead (the look-ahead must not fail the frame in front of it)."""

    def __init__(self, load_host, order, device, depth: int = 2, add_batch: bool = True):
        self.load_host = load_host
        self.order = list(order)
        self.pos = {}
        for k, i in enumerate(self.order):
            self.pos.setdefault(i, k)
        self.device = torch.device(device) if device is not None else torch.device('cpu')
        self.on_gpu = self.device.type == 'cuda'
        self.depth = max(1, int(depth))
        self.add_batch = add_batch
        self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix='avc-prefetch')
        self._fut = {}
        self._done = {}
        self._slots = [_Pinned() for _ in range(self.depth + 1)]
        self._n = 0
        self._copy = torch.cuda.Stream(self.device) if self.on_gpu else None
        self.h2d_copies = 0           # one per frame, whatever the item holds
        self.h2d_bytes = 0

    # -- worker thread
    def _load(self, i, slot: _Pinned):
        host = self.load_host(i)
        out, parts, off = {}, [], 0
        for key, val in host.items():
            if isinstance(val, torch.Tensor) and val.device.type == 'cpu':
                arr = val.detach().contiguous().numpy()
            elif isinstance(val, np.ndarray):
                arr = np.ascontiguousarray(val)
            else:
                out[key] = val.unsqueeze(0) if (self.add_batch and isinstance(val, torch.Tensor)) else val
                continue
            off = _align(off)
            parts.append((key, arr, off))
            off += arr.nbytes
        if not self.on_gpu:
            for key, arr, _ in parts:
                t = torch.from_numpy(arr)
                out[key] = t.unsqueeze(0) if self.add_batch else t
            out['_host'] = host
            out['_host_ids'] = {k: id(v) for k, v in out.items() if isinstance(v, torch.Tensor)}
            return out, None, None
        torch.cuda.set_device(self.device)
        if slot.event is not None:                       # the copy that last read this staging buffer
            slot.event.synchronize()
        pinned = slot.ensure(max(off, _ALIGN))
        hp = pinned.numpy()
        for key, arr, o in parts:
            if arr.nbytes:
                hp[o:o + arr.nbytes] = arr.reshape(-1).view(np.uint8)
        with torch.cuda.stream(self._copy):
            dev = torch.empty(max(off, _ALIGN), dtype=torch.uint8, device=self.device)
            dev.copy_(pinned[:dev.numel()], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy)
        slot.event = ev
        self.h2d_copies += 1
        self.h2d_bytes += off
        for key, arr, o in parts:
            if arr.nbytes:
                t = dev[o:o + arr.nbytes].view(_torch_dtype(arr.dtype)).view(arr.shape)
            else:
                t = torch.empty(arr.shape, dtype=_torch_dtype(arr.dtype), device=self.device)
            out[key] = t.unsqueeze(0) if self.add_batch else t
        out['_host'] = host
        out['_host_ids'] = {k: id(v) for k, v in out.items() if isinstance(v, torch.Tensor)}
        return out, ev, dev

    # -- loop thread
    def _schedule(self, i):
        k = self.pos.get(i)
        want = self.order[k:k + self.depth] if k is not None else [i]
        for j in want:
            if j not in self._fut and j not in self._done:
                self._fut[j] = self._pool.submit(self._load, j, self._slots[self._n % len(self._slots)])
                self._n += 1

    def _resolve(self, i):
        if i not in self._done:
            self._schedule(i)
            fut = self._fut.pop(i)
            try:
                self._done[i] = ('ok', fut.result())
            except Exception as e:      # noqa: BLE001 -- handed to whoever asks for the frame
                self._done[i] = ('error', e)
        return self._done[i]

    def _hand_over(self, res):
        items, ev, dev = res
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            dev.record_stream(cur)
        return items

    def get(self, i):
        kind, res = self._resolve(i)
        if kind == 'error':
            raise res
        return self._hand_over(res)

    def peek(self, i):
        if i is None:
            return None
        kind, res = self._resolve(i)
        return None if kind == 'error' else self._hand_over(res)

    def drop(self, i):
        """Frame i is finished: let go of its block (the caching allocator hands it out again once the streams recorded on it have passed)."""
        self._done.pop(i, None)
        k = self.pos.get(i)
        if k is not None and k + 1 < len(self.order):
            self._schedule(self.order[k + 1])

    def close(self):
        self._pool.shutdown(wait=True, cancel_futures=True)
        self._fut.clear()
        self._done.clear()


class MeshWriter:
    """Finished frames to disk behind the loop.  `submit(tensors, write, tag)`: `tensors` name -> tensor (device or host); `write(arrays)` is called on a writer
    thread with name -> NumPy array (views of a pinned slot, valid only inside the call).  `close()` waits for everything and returns [(tag, 'Type: message')]
    of the writes that failed."""

    def __init__(self, device=None, slots: int = 4, threads: int = 3):
        self.device = torch.device(device) if device is not None else torch.device('cpu')
        self.on_gpu = self.device.type == 'cuda'
        self._free = queue.Queue()
        for _ in range(max(1, slots)):
            self._free.put(_Pinned())
        self._jobs = queue.Queue()
        self._copy = torch.cuda.Stream(self.device) if self.on_gpu else None
        self.failed = []
        self.d2h_bytes = 0
        self.submitted = 0
        self.waited_for_slot_s = 0.0
        self._threads = [threading.Thread(target=self._run, name=f'avc-writer-{k}', daemon=True) for k in range(max(1, threads))]
        for t in self._threads:
            t.start()

    def submit(self, tensors: dict, write, tag=None):
        import time
        tensors = {k: v for k, v in tensors.items() if v is not None}
        self.submitted += 1
        if not self.on_gpu:
            arrays = {k: (v.detach().cpu().contiguous().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in tensors.items()}
            self._jobs.put((None, None, arrays, write, tag))
            return
        t0 = time.perf_counter()
        slot = self._free.get()                          # back-pressure: at most `slots` frames between the device and the disk
        self.waited_for_slot_s += time.perf_counter() - t0
        cur = torch.cuda.current_stream(self.device)
        tensors = {k: v.contiguous() for k, v in tensors.items()}
        plan, off = [], 0
        for k, v in tensors.items():
            off = _align(off)
            nb = v.numel() * v.element_size()
            plan.append((k, v, off, nb))
            off += nb
        pinned = slot.ensure(max(off, _ALIGN))
        ready = torch.cuda.Event()
        ready.record(cur)
        arrays = {}
        with torch.cuda.stream(self._copy):
            self._copy.wait_event(ready)
            for k, v, o, nb in plan:
                if nb:
                    dst = pinned[o:o + nb].view(v.dtype).view(v.shape)
                    dst.copy_(v, non_blocking=True)
                    v.record_stream(self._copy)
                    arrays[k] = dst.numpy()
                else:
                    arrays[k] = np.empty(tuple(v.shape), dtype=torch.empty(0, dtype=v.dtype).numpy().dtype)
            done = torch.cuda.Event()
            done.record(self._copy)
        slot.event = done
        self.d2h_bytes += off
        self._jobs.put((slot, done, arrays, write, tag))

    def _run(self):
        while True:
            job = self._jobs.get()
            if job is None:
                return
            slot, done, arrays, write, tag = job
            try:
                if done is not None:
                    done.synchronize()
                write(arrays)
            except Exception as e:      # noqa: BLE001 -- reported by close(); the loop goes on
                self.failed.append((tag, f'{type(e).__name__}: {e}'))
            finally:
                if slot is not None:
                    self._free.put(slot)
                self._jobs.task_done()

    def drain(self):
        self._jobs.join()

    def close(self):
        self._jobs.join()
        for _ in self._threads:
            self._jobs.put(None)
        for t in self._threads:
            t.join()
        return list(self.failed)
