"""Host -> HBM feeder for `Model.fit` / `evaluate`: the step before the hot path when the caller holds NumPy arrays
(the reference's `model.fit(x, y)` / `fit(Sequence)`, README.md:241-297).

One worker thread gathers each minibatch straight into pinned staging buffers (a single host pass: the shuffled
row gather IS the staging copy), the caller's thread enqueues the H2D copies on a copy stream, and the compute
stream waits on an event only, so the PCIe transfer of batch i+1 runs under the training step of batch i.
Staging slots and device buffers are recycled through events; nothing here synchronises the device.
"""
import os
import queue
import threading
import time

import numpy as np
import torch

_DEPTH = 3      # pinned staging slots (host gather of i+2 | PCIe copy of i+1 | step i)
_DEV_SETS = 2   # device-side input sets


def _as_f32_array(a):
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    a = np.asarray(a)
    if a.dtype != np.float32:
        a = a.astype(np.float32)
    return np.ascontiguousarray(a)


class _Slot:
    def __init__(self):
        self.bufs = None      # pinned tensors, one per array of the batch
        self.host = None      # NumPy views of the same memory
        self.copied = None    # event: the H2D copies out of this slot have finished
        self.n = 0


class FeederBuffers:
    """Staging slots, device sets and their events; kept by the Model so that epochs reuse them (pinning
    hundreds of MB per epoch is slow, and the events carry the ordering from one epoch into the next)."""

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = [_Slot() for _ in range(_DEPTH)]
        self.dev_sets = [None] * _DEV_SETS
        self.dev_free = [None] * _DEV_SETS    # event: the step that read this set has finished


class HostFeeder:
    """Iterates device-resident `(x, [y...])` batches built from host data.

    `source` is either `("arrays", [x, y0, y1, ...], order, batch_size)` - row gather by `order` - or
    `("batches", iterable)` yielding `(x, y)` host batches (the Sequence protocol of the reference's
    `_Yolov3DataSequence`, yolov3/__init__.py:41-53). Yields `(x, [y0, ...])` views of recycled device buffers:
    valid until the next item is requested."""

    def __init__(self, source, buffers=None):
        self.source = source
        self.bufs = buffers if buffers is not None else FeederBuffers()
        self.device, self.copy_stream = self.bufs.device, self.bufs.copy_stream
        self.free = queue.Queue()
        for s in self.bufs.slots:
            self.free.put(s)
        self.ready = queue.Queue(maxsize=_DEPTH)
        self.dev_sets, self.dev_free = self.bufs.dev_sets, self.bufs.dev_free
        self.error = None
        self.stop = False
        self.stats = {"stage_s": 0.0, "slot_wait_s": 0.0, "ready_wait_s": 0.0, "batches": 0}
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()

    # ---- worker thread: host gather into pinned memory ----
    def _stage(self, slot, tensors, index):
        n = int(index.size) if index is not None else int(tensors[0].shape[0])
        shapes = [(n,) + tuple(t.shape[1:]) for t in tensors]
        if slot.bufs is None or any(b.shape[0] < n or tuple(b.shape[1:]) != s[1:] for b, s in zip(slot.bufs, shapes)) \
                or len(slot.bufs) != len(tensors):
            slot.bufs = [torch.empty(s, dtype=torch.float32).pin_memory() for s in shapes]
            slot.host = [b.numpy() for b in slot.bufs]
        t0 = time.perf_counter()
        if slot.copied is not None:
            slot.copied.synchronize()     # the previous H2D out of this slot must have drained
        t1 = time.perf_counter()
        self.stats["slot_wait_s"] += t1 - t0
        # one thread, GIL released inside NumPy: the caller's thread is busy enqueuing ~700 launches per step and
        # must not share its cores with a spinning intra-op pool (torch.index_select cost 25 % of the step rate)
        for h, t in zip(slot.host, tensors):
            if index is not None:
                np.take(t, index, axis=0, out=h[:n], mode="clip")   # "clip": unbuffered write into `out`
            else:
                np.copyto(h[:n], t)
        slot.n = n
        self.stats["stage_s"] += time.perf_counter() - t1
        self.stats["batches"] += 1

    def _work(self):
        try:
            if self.source[0] == "arrays":
                _, arrays, order, bs = self.source
                tensors = [_as_f32_array(a) for a in arrays]
                order_t = np.ascontiguousarray(order).astype(np.int64)
                for i in range(0, len(order), bs):
                    if self.stop:
                        break
                    slot = self.free.get()
                    self._stage(slot, tensors, order_t[i:i + bs])
                    self.ready.put(slot)
            else:
                for xb, yb in self.source[1]:
                    if self.stop:
                        break
                    ys = list(yb) if isinstance(yb, (list, tuple)) else [yb]
                    slot = self.free.get()
                    self._stage(slot, [_as_f32_array(a) for a in [xb] + ys], None)
                    self.ready.put(slot)
        except BaseException as e:   # surfaced on the consumer's thread
            self.error = e
        self.ready.put(None)

    # ---- consumer thread ----
    def __iter__(self):
        k = 0
        compute = torch.cuda.current_stream(self.device)
        while True:
            t0 = time.perf_counter()
            slot = self.ready.get()
            self.stats["ready_wait_s"] += time.perf_counter() - t0
            if slot is None:
                if os.environ.get("YOLO_FEED_STATS") == "1":
                    nb = max(self.stats["batches"], 1)
                    print("[feeder] per batch: host gather %.1f ms, staging-slot wait %.1f ms, consumer wait %.1f ms"
                          % tuple(self.stats[k] / nb * 1e3 for k in ("stage_s", "slot_wait_s", "ready_wait_s")), flush=True)
                if self.error is not None:
                    raise self.error
                return
            j = k % _DEV_SETS
            n = slot.n
            dset = self.dev_sets[j]
            if dset is None or len(dset) != len(slot.bufs) or any(
                    d.shape[0] < n or d.shape[1:] != b.shape[1:] for d, b in zip(dset, slot.bufs)):
                dset = self.dev_sets[j] = [torch.empty(b.shape, dtype=torch.float32, device=self.device)
                                           for b in slot.bufs]
                # fresh blocks may be recycled from tensors the compute stream is still using
                self.copy_stream.wait_stream(compute)
                self.dev_free[j] = None
            with torch.cuda.stream(self.copy_stream):
                if self.dev_free[j] is not None:
                    self.copy_stream.wait_event(self.dev_free[j])
                for d, b in zip(dset, slot.bufs):
                    d[:n].copy_(b[:n], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            slot.copied = ev
            self.free.put(slot)
            compute.wait_event(ev)
            views = [d[:n] for d in dset]
            yield views[0], views[1:]
            done = torch.cuda.Event()
            done.record(compute)
            self.dev_free[j] = done
            k += 1

    def close(self):
        self.stop = True
        try:
            while True:   # unblock a worker waiting for a slot / a full queue
                s = self.ready.get_nowait()
                if s is not None:
                    self.free.put(s)
        except queue.Empty:
            pass
        self.free.put(_Slot())   # never staged: only wakes a worker blocked on an empty pool
        self.thread.join(timeout=5)
        torch.cuda.current_stream(self.device).wait_stream(self.copy_stream)
