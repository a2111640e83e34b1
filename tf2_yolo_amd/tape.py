"""The training step as a replayed LAUNCH TAPE: the C-ABI calls of one step, recorded once with their marshalled
arguments, re-issued from a flat list.

Why not a hipGraph? `capture.py` builds one, and it is bit-identical -- but on this runtime (ROCm 7.2) the graph executor
runs the two captured streams of the step one after the other: 35.4 ms per step against 31.2 ms for the same launches
enqueued eagerly on their two streams (same box, profiles/r03_*). The tape keeps the eager execution model -- the same
launches on the same two HIP streams with the same events -- and only removes the Python that produced them: ~700
library calls through ops.py (argument checks, ctypes marshalling, tensor bookkeeping: ~20 ms of host time per step)
become ~900 replayed (function, argument tuple) pairs (~2 ms). One process per GPU must not depend on a free host core
to keep its GPU fed; with eight ranks on one host that is the point.

What is recorded (tape.ACTIVE is set): every libyolo_hip.so call that enqueues work (`_lib` wraps the library's
functions), the stream / event operations of the executor and of the gradient reducer (the helpers below), and the few
torch copies of the step as closures. Every tensor whose address went into a recorded argument is kept alive by the tape
(ops._p), so replays read and write exactly the buffers of the recorded step -- the role a graph's private memory pool
plays. Step-dependent scalars (Adam's bias-corrected rate, 1/world) live in device memory and are refreshed before each
replay (Adam.refresh_hyper), as for the graph.
"""
import torch

ACTIVE = None     # the Tape being recorded, or None


class Tape:
    def __init__(self):
        self.entries = []     # (c function | None, args | python closure)
        self.keep = []        # tensors whose addresses are inside recorded arguments

    def c(self, f, args, name):
        self.entries.append((f, args, name))

    def py(self, fn):
        self.entries.append((None, fn, None))

    def replay(self):
        for f, a, name in self.entries:
            if f is None:
                a()
            elif f(*a) != 0:
                from . import _lib
                msg = _lib.load().yolo_last_error()
                raise _lib.YoloHipError(f"{name} failed during a tape replay: {msg.decode() if msg else ''}")


# ---- stream / event operations of the step: executed now and, while a tape records, remembered ----
def wait_stream(waiter, other):
    """`waiter` waits for everything enqueued on `other` so far"""
    waiter.wait_stream(other)
    if ACTIVE is not None:
        ev = torch.cuda.Event()      # one event object per recorded wait, re-recorded at every replay (Stream.wait_stream
        rec, wait = ev.record, waiter.wait_event   # would create a fresh one each time: ~10 us of host time per wait)

        def again():
            rec(other)
            wait(ev)
        ACTIVE.py(again)


def record_event(stream):
    ev = torch.cuda.Event()
    ev.record(stream)
    if ACTIVE is not None:
        ACTIVE.py(lambda: ev.record(stream))
    return ev


def wait_event(stream, ev):
    stream.wait_event(ev)
    if ACTIVE is not None:
        ACTIVE.py(lambda: stream.wait_event(ev))


def torch_op(fn):
    """a torch operation of the step (a device-to-device copy): run on the current stream now, on the same stream at replay"""
    fn()
    if ACTIVE is not None:
        st = torch.cuda.current_stream()

        def again():
            with torch.cuda.stream(st):
                fn()
        ACTIVE.py(again)


def host_call(fn):
    """a host-side action that belongs to the step (the gradient reducer's bucket launch): run now, and again at replay"""
    r = fn()
    if ACTIVE is not None:
        ACTIVE.py(fn)
    return r
