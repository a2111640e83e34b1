"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference has no distributed path at all; this is the MI355X-native
addition BASELINE.json asks for. All trainable gradients live in one flat buffer whose layout
follows graph construction order, and backward completes units in reverse construction order,
so gradient buckets are CONTIGUOUS slices taken from the END of the buffer: no gather/scatter
copies, a bucket is ready when backward has passed the unit that owns its lowest offset.
Each ready bucket is summed with one all-reduce on a side stream while backward keeps running;
the optimizer waits on the side stream and applies 1/world (the global-batch mean).

xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU); with ~248 MB of fp32 gradients
a handful of large buckets (default 48 MB) keeps every collective bandwidth- rather than
latency-bound, and the first bucket (heads + FPN) is launched within the first few ms of
backward. BatchNormalization statistics stay per-replica (what "bs=32/GPU" with the
reference's plain BN means); moving statistics are not reduced.
"""
import os

import torch
import torch.distributed as dist


def _forced():
    """YOLO_DP_FORCE=1: run the collectives even in a world of one rank (exercises the RCCL calls, the side
    stream and the bucket views on a single GPU; tests/test_gpu_dp.py)."""
    return os.environ.get("YOLO_DP_FORCE") == "1"


def plan_buckets(segments, total, bucket_elems):
    """segments: [(offset, size)] in BACKWARD-COMPLETION order (descending offsets, contiguous).
    Returns (buckets, closes): buckets = [(lo, hi)] in ready order, closes[i] = list of bucket
    indices that become ready once segment i is complete."""
    buckets, closes = [], [[] for _ in segments]
    hi = total
    acc = 0
    for i, (off, size) in enumerate(segments):
        acc = hi - off
        last = i == len(segments) - 1
        if acc >= bucket_elems or last:
            lo = 0 if last else off
            if hi > lo:
                buckets.append((lo, hi))
                closes[i].append(len(buckets) - 1)
            hi = lo
    return buckets, closes


class GradReducer:
    """Bucketed, overlapped all-reduce(sum) of a flat gradient tensor."""

    def __init__(self, flat_grads, segments, process_group=None, bucket_bytes=48 << 20):
        self.flat = flat_grads
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (dist.is_initialized() and _forced())
        self.buckets, self.closes = plan_buckets(segments, flat_grads.numel(), max(bucket_bytes // 4, 1))
        self.on_gpu = flat_grads.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.on_gpu else None
        self._works = []
        # callable -> list of further CUDA streams that produce gradients (the engine's filter-gradient
        # stream): a bucket is reduced only after the work enqueued on them so far has finished too
        self.extra_streams = None

    def segment_done(self, i):
        if not self.active:
            return
        for b in self.closes[i]:
            self.reduce_bucket(b, self.extra_streams() if self.extra_streams is not None else ())

    def reduce_bucket(self, b, extra_streams=()):
        """all-reduce bucket b on the communication stream, ordered after everything enqueued so far on the current
        stream (and on `extra_streams`). The captured step (capture.py) calls this between two graph replays: the
        replayed segment has joined the filter-gradient stream itself, so there are no extra streams."""
        lo, hi = self.buckets[b]
        view = self.flat[lo:hi]
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            for st in extra_streams:
                ev2 = torch.cuda.Event()
                ev2.record(st)
                self.comm_stream.wait_event(ev2)
            with torch.cuda.stream(self.comm_stream):
                self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self):
        """Make the compute stream wait for every outstanding bucket. Returns 1/world."""
        if not self.active:
            return 1.0
        for w in self._works:
            w.wait()           # on GPU: stream-level wait, does not block the host
        self._works = []
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        return 1.0 / self.world


def broadcast_parameters(tensors, src=0, process_group=None):
    if not dist.is_initialized() or (dist.get_world_size(process_group) == 1 and not _forced()):
        return
    for t in tensors:
        dist.broadcast(t, src=src, group=process_group)


def network_segments(net):
    """(offset, size) of each parameterised unit's slice of the flat buffer, in backward order."""
    segs, units = [], []
    for u in reversed(net.units):
        if u.kind not in ("conv", "head"):
            continue
        specs = [s for s in (getattr(u, "p_kernel", None), getattr(u, "p_bias", None),
                             getattr(u, "p_gamma", None), getattr(u, "p_beta", None)) if s is not None]
        lo = min(s.offset for s in specs)
        hi = max(s.offset + (s.size + 63) // 64 * 64 for s in specs)
        segs.append((lo, hi - lo))
        units.append(u)
    return segs, units
