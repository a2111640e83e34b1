"""The training step as replayed hipGraphs: forward + losses + backward + optimizer captured ONCE, then a handful of graph
launches per step instead of ~700 kernel launches enqueued from Python (20 ms of host time per 32 ms step; DESIGN.md
section 5). One process per GPU keeps that host time off the critical path only while a core is free for it; eight ranks on
one host should not depend on that.

Segments. In a data-parallel job the gradient buckets are all-reduced WHILE backward runs (dp.GradReducer). Collectives
are not captured (gloo cannot be, and RCCL captures are not worth the risk on a path nobody can rehearse here): the
step is cut into graph segments at the points where a bucket becomes complete, and the all-reduce of that bucket is
issued eagerly between two replays:

    [ forward, losses, backward down to bucket 0 ] -> all-reduce(bucket 0) on the communication stream
    [ backward down to bucket 1 ]                 -> all-reduce(bucket 1)
    ...
    [ rest of backward ]                          -> wait for the communication stream
    [ Adam ]

Without data parallelism the whole step is ONE graph. The executor's second stream (filter gradients) is forked and
joined inside every segment, so a cut costs one join of the two streams. The optimizer's step-dependent scalars (the
bias-corrected rate, 1/world) live in device memory and are refreshed before each replay (Adam.refresh_hyper), so the
captured launches never change. Results are bit-identical to the eager step (same kernels, same order, same operands;
the filter-gradient reductions are atomics-free: ops.ensure_wgrad_workspace).
"""
import torch

from ._lib import YoloHipError


_LEAKED = []   # graph objects of failed captures (see StepGraphs._capture)


class _StaticInputs:
    """the static input buffers of a recorded step: a caller's batch is copied in before a replay unless it is the very
    tensor object (same storage, same version counter: not modified in place since) that was copied last time"""

    def __init__(self, x, y_list):
        self.x = x.clone()
        self.ys = [y.clone() for y in y_list]
        self._last = [(x, x._version)] + [(y, y._version) for y in y_list]

    def load(self, x, y_list):
        for i, (dst, src) in enumerate(zip([self.x] + self.ys, [x] + list(y_list))):
            last, ver = self._last[i]
            if src is last and src._version == ver:
                continue
            dst.copy_(src, non_blocking=True)
            self._last[i] = (src, src._version)


class StepGraphs:
    """Captured training step of one Model for one (batch size, loss list, optimizer, reducer) configuration."""

    def __init__(self, model, x, y_list):
        self.model = model
        self.key = self.key_of(model, x)
        net = model.net
        opt = model.optimizer
        if not getattr(opt, "capturable", False):
            raise YoloHipError("this optimizer has no capturable form")
        # static inputs of the graphs: the caller's batch is copied into them before every replay
        self.inputs = _StaticInputs(x, y_list)
        self.x, self.ys = self.inputs.x, self.inputs.ys
        self.segments = []          # (graph, action): action = ("reduce", bucket) | ("finish",) | None
        opt._hyper_buffers()        # (pinned host + device scalars: allocated here, never inside a capture)
        torch.cuda.synchronize()
        self._capture(net, opt)

    @staticmethod
    def key_of(model, x):
        return (tuple(x.shape), getattr(model.net, "alloc_gen", 0), tuple(id(l) for l in model.loss), id(model.optimizer),
                id(model._reducer), model.net.anchors_trainable)

    def _capture(self, net, opt):
        model = self.model
        red = model._reducer if (model._reducer is not None and model._reducer.active) else None
        pool = torch.cuda.graph_pool_handle()
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        state = {"g": None}

        def begin():
            g = torch.cuda.CUDAGraph()
            # thread_local: other threads of the process keep their runtime calls (RCCL's watchdog polls events while this
            # thread captures; in the default "global" mode that poll is an error that kills the process group)
            g.capture_begin(pool=pool, capture_error_mode="thread_local")
            state["g"] = g

        def end(action):
            state["g"].capture_end()
            self.segments.append((state["g"], action))
            state["g"] = None

        # events recorded by the eager steps before (the dy-planes double buffer's) belong to uncaptured work: a captured
        # stream must not wait for them, and need not -- `stream` is ordered after everything enqueued so far, and the
        # second stream was joined at the end of the last backward
        net._dyp_events = [None, None]
        net._wp_event = net._wT_event = None
        old_hook = net.grad_ready_hook
        if red is not None:
            index = model._dp_unit_index

            def cut_hook(u):
                for b in red.closes[index[id(u)]]:
                    # a bucket is complete: join the filter-gradient stream, close this segment; the all-reduce of the
                    # bucket is issued between this replay and the next
                    net._join_wgrad()
                    net._dyp_events = [None, None]
                    end(("reduce", b))
                    begin()
            net.grad_ready_hook = cut_hook
        try:
            with torch.cuda.stream(stream):
                begin()
                outs = net.forward(self.x, training=True)
                for i, (o, yt) in enumerate(zip(outs, self.ys)):
                    model.loss[i].fwd_bwd(yt, o, grad_scale=1.0, dpred=model._dpred[i], loss_out=model._loss_bufs[i])
                net.backward(model._dpred)
                net._dyp_events = [None, None]
                if red is not None:
                    end(("finish",))
                    begin()
                opt.step_captured()
                end(None)
        except Exception:
            # end the broken capture as far as the runtime lets us and keep every graph object of this attempt alive for
            # the life of the process: destroying a graph whose capture was invalidated aborts inside the runtime
            _LEAKED.extend(g for g, _ in self.segments)
            if state["g"] is not None:
                _LEAKED.append(state["g"])
                try:
                    if net._wgrad_stream is not None:
                        torch.cuda.current_stream().wait_stream(net._wgrad_stream)
                    state["g"].capture_end()
                except Exception:
                    pass
            self.segments = []
            net._dyp_events = [None, None]
            net._wp_event = net._wT_event = None
            net._wgrad_pending = False
            raise
        finally:
            net.grad_ready_hook = old_hook
        torch.cuda.current_stream().wait_stream(stream)
        # the capture pass executed nothing, but it ran the host side of a step: leave the host flags as after a step
        net.mark_params_changed()

    def replay(self, x, y_list):
        model, net, opt = self.model, self.model.net, self.model.optimizer
        red = model._reducer if (model._reducer is not None and model._reducer.active) else None
        self.inputs.load(x, y_list)
        opt.refresh_hyper(grad_scale=(1.0 / red.world) if red is not None else 1.0)
        for g, action in self.segments:
            g.replay()
            if action is None:
                continue
            if action[0] == "reduce":
                red.reduce_bucket(action[1])
            else:
                if net.anchors_trainable and red.world > 1:
                    import torch.distributed as dist
                    dist.all_reduce(net.anchor_grads, group=red.pg)
                red.finish()
        net.mark_params_changed()
        return model._loss_bufs


class StepTape:
    """The recorded training step of one Model (tape.py): same key, same replay interface as StepGraphs. Recording
    EXECUTES the step (it is an ordinary eager step whose library calls are remembered), so `record` returns its losses."""

    def __init__(self, model, x, y_list):
        from . import tape as tape_mod
        self.model = model
        self.key = StepGraphs.key_of(model, x)
        opt, net = model.optimizer, model.net
        if not getattr(opt, "capturable", False):
            raise YoloHipError("this optimizer has no capturable form")
        self.inputs = _StaticInputs(x, y_list)
        self.x, self.ys = self.inputs.x, self.inputs.ys
        opt._hyper_buffers()
        red = model._reducer if (model._reducer is not None and model._reducer.active) else None
        # (events of the eager steps before: the recorded step must not wait for objects no replay will ever re-record;
        # the two streams were joined at the end of the last backward)
        net._dyp_events = [None, None]
        self.tape = tape_mod.Tape()
        opt.refresh_hyper(grad_scale=(1.0 / red.world) if red is not None else 1.0)
        tape_mod.ACTIVE = self.tape
        try:
            outs = net.forward(self.x, training=True)
            for i, (o, yt) in enumerate(zip(outs, self.ys)):
                model.loss[i].fwd_bwd(yt, o, grad_scale=1.0, dpred=model._dpred[i], loss_out=model._loss_bufs[i])
            net.backward(model._dpred)
            if red is not None:
                if net.anchors_trainable and red.world > 1:
                    import torch.distributed as dist
                    tape_mod.host_call(lambda: dist.all_reduce(net.anchor_grads, group=red.pg))
                tape_mod.host_call(red.finish)
            opt.step_captured()
        finally:
            tape_mod.ACTIVE = None
        net.mark_params_changed()

    def replay(self, x, y_list):
        model, net, opt = self.model, self.model.net, self.model.optimizer
        red = model._reducer if (model._reducer is not None and model._reducer.active) else None
        self.inputs.load(x, y_list)
        opt.refresh_hyper(grad_scale=(1.0 / red.world) if red is not None else 1.0)
        self.tape.replay()
        net.mark_params_changed()
        return model._loss_bufs
