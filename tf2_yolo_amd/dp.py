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


def plan_buckets(segments, total, bucket_elems, first_bucket_elems=None):
    """segments: [(offset, size)] in BACKWARD-COMPLETION order (descending offsets, contiguous).
    Returns (buckets, closes): buckets = [(lo, hi)] in ready order, closes[i] = list of bucket
    indices that become ready once segment i is complete. first_bucket_elems: a smaller threshold for the FIRST bucket,
    so that the first collective starts early in backward (the heads and the coarsest FPN level: 2.9 ms into backward
    with 12 MB against 4.0 ms with a whole 48 MB bucket, profiles/r04_b_dp_readiness.json)."""
    buckets, closes = [], [[] for _ in segments]
    hi = total
    acc = 0
    for i, (off, size) in enumerate(segments):
        acc = hi - off
        last = i == len(segments) - 1
        if acc >= (first_bucket_elems if (first_bucket_elems and not buckets) else bucket_elems) or last:
            lo = 0 if last else off
            if hi > lo:
                buckets.append((lo, hi))
                closes[i].append(len(buckets) - 1)
            hi = lo
    return buckets, closes


class GradReducer:
    """Bucketed, overlapped all-reduce(sum) of a flat gradient tensor."""

    def __init__(self, flat_grads, segments, process_group=None, bucket_bytes=48 << 20):
        if os.environ.get("YOLO_DP_BUCKET_MB"):          # experiments: profiles/r04_b_dp_readiness.json
            bucket_bytes = int(float(os.environ["YOLO_DP_BUCKET_MB"]) * (1 << 20))
        self._dry = os.environ.get("YOLO_DP_DRYRUN") == "1"   # experiments: every event and wait, no collective call
        self.flat = flat_grads
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (dist.is_initialized() and _forced())
        first = int(float(os.environ.get("YOLO_DP_FIRST_BUCKET_MB", "12")) * (1 << 20))
        self.buckets, self.closes = plan_buckets(segments, flat_grads.numel(), max(bucket_bytes // 4, 1),
                                                 max(min(first, bucket_bytes) // 4, 1))
        self.on_gpu = flat_grads.is_cuda
        if self.on_gpu:   # a stream that demonstrably runs beside the compute stream and the filter-gradient stream (ops.concurrent_stream)
            from . import ops
            self.comm_stream = ops.concurrent_stream("comm", device=flat_grads.device)
        else:
            self.comm_stream = None
        self._works = []
        # callable -> list of further CUDA streams that produce gradients (the engine's filter-gradient
        # stream): a bucket is reduced only after the work enqueued on them so far has finished too
        self.extra_streams = None
        # YOLO_DP_TRACE / start_trace(): per step, when (on the device) backward began, when each bucket was ready for its
        # all-reduce (every gradient of the bucket written on both gradient streams) and when the all-reduce had finished
        self.tracing = os.environ.get("YOLO_DP_TRACE") == "1"
        self._t0 = None
        self._trace = []

    def start_trace(self, on=True):
        self.tracing = bool(on)
        self._t0, self._trace = None, []

    def backward_begin(self):
        """called (through the launch tape too) when Network.backward starts: the time origin of the bucket trace"""
        if self.active and self.tracing and self.on_gpu:
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t0.record(torch.cuda.current_stream())
            self._trace = []

    def trace_ms(self):
        """after a device synchronisation: [(bucket megabytes, ms from backward start until the bucket was ready, until its
        all-reduce had finished)] of the last traced step"""
        if self._t0 is None:
            return []
        return [(round((hi - lo) * 4 / 1e6, 1), round(self._t0.elapsed_time(r), 3), round(self._t0.elapsed_time(d), 3))
                for (lo, hi), r, d in self._trace]

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
                if self.tracing and self._t0 is not None:
                    ready = torch.cuda.Event(enable_timing=True)
                    ready.record(self.comm_stream)
                work = None
                if not self._dry:
                    work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                    self._works.append(work)
                if self.tracing and self._t0 is not None:
                    # The collective does NOT run on this stream: RCCL runs it on the process group's own stream, gloo on
                    # host threads. `done` is therefore recorded behind work.wait() -- for RCCL a stream-level wait (the
                    # communication stream waits for the collective's stream, the host does not block), for gloo a host wait
                    # (traced steps only). Round 4 recorded `done` right behind the async call and measured nothing.
                    if work is not None:
                        work.wait()
                    done = torch.cuda.Event(enable_timing=True)
                    done.record(self.comm_stream)
                    self._trace.append(((lo, hi), ready, done))
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
