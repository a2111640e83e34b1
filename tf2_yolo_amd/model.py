"""Keras-like Model shell over the HIP executor: the object `Yolo.create_model` hands back.

Mirrors the slice of tf.keras.Model the reference's README drives (README.md:241-336):
compile / fit / train_on_batch / evaluate / predict / __call__, load_weights / save_weights
(.npz keyed by Keras layer names; h5py is not available), get_layer(name).get_weights() /
.set_weights(), .output[i].shape, .layers, count_params().
"""
import time

import numpy as np
import torch

from . import dp as dp_mod
from . import optimizers as opt_mod
from ._lib import YoloHipError
from .engine import Network


def _to_input(x):
    if torch.is_tensor(x):
        return x.to(device="cuda", dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).cuda()


class LayerView:
    """get_layer(name) result: Keras-style weight access onto slices of the flat buffers."""

    def __init__(self, model, name, getter, setter, trainable=True):
        self.model, self.name, self._get, self._set, self.trainable = model, name, getter, setter, trainable

    def get_weights(self):
        return self._get()

    def set_weights(self, weights):
        self.model.net.before_param_write()
        self._set(weights)
        self.model.net.mark_params_changed()


class Model:
    def __init__(self, builder, version, seed=1234, unbiased_moving_var=True):
        self.net = Network(builder, seed=seed, unbiased_moving_var=unbiased_moving_var)
        self.version = version
        self.single_output = len(self.net.outputs) == 1 and version in (1, 2)
        self.optimizer = None
        self.loss = None
        self.metrics = None
        self._reducer = None
        self._loss_bufs = None
        self.stop_training = False
        self.history = None

    # ---- structure ----
    @property
    def input_shape(self):
        return (None, self.net.input.h, self.net.input.w, self.net.input.c)

    @property
    def output(self):
        outs = self.net.outputs
        return outs[0] if self.single_output else list(outs)

    @property
    def output_shape(self):
        outs = [o.shape for o in self.net.outputs]
        return outs[0] if self.single_output else outs

    def count_params(self):
        return self.net.num_params + self.net.num_state

    def trainable_count(self):
        return self.net.num_params

    @property
    def layers(self):
        return [self.get_layer(n) for n in self.layer_names()]

    def layer_names(self):
        names = []
        for u in self.net.units:
            if u.kind == "conv":
                names.append(f"{u.name}_conv")
                if u.bn:
                    names.append(f"{u.name}_bn")
            elif u.kind == "head":
                names += self._head_layer_names(u)
            else:
                names.append(u.name)
        return names

    def _head_layer_names(self, u):
        i = u.level + 1
        if u.version == 1:
            return [f"out{i}_xywhc_conv", f"out{i}_prob_conv"]
        names = []
        for j in range(u.A):
            names += [f"out{i}_box{j + 1}_{p}_conv" for p in ("xy", "wh", "conf", "prob")]
            if u.version == 4:
                names.append(f"out{i}_box{j + 1}_anchor")
        return names

    def _head_rows(self, u, lname):
        """row range of the fused head kernel that a Keras head layer owns"""
        D = 5 + u.C
        if u.version == 1:
            return (0, 5 * u.A) if lname.endswith("_xywhc_conv") else (5 * u.A, 5 * u.A + u.C)
        j = int(lname.split("_box")[1].split("_")[0]) - 1
        part = lname.split("_")[-2]
        lo, hi = {"xy": (0, 2), "wh": (2, 4), "conf": (4, 5), "prob": (5, D)}[part]
        return j * D + lo, j * D + hi

    def get_layer(self, name=None, index=None):
        if name is None:
            name = self.layer_names()[index]
        net = self.net
        for u in net.units:
            if u.kind == "conv" and name == f"{u.name}_conv":
                def get(u=u):
                    k = net.params.view(u.p_kernel.name).cpu().numpy().reshape(u.p_kernel.shape)
                    out = [np.transpose(k, (1, 2, 3, 0)).copy()]
                    if u.p_bias is not None:
                        out.append(net.params.view(u.p_bias.name).cpu().numpy().copy())
                    return out

                def set_(w, u=u):
                    k = np.transpose(np.asarray(w[0], dtype=np.float32), (3, 0, 1, 2))
                    net.params.view(u.p_kernel.name).copy_(torch.from_numpy(np.ascontiguousarray(k)).reshape(-1))
                    if u.p_bias is not None:
                        net.params.view(u.p_bias.name).copy_(torch.from_numpy(np.asarray(w[1], dtype=np.float32)))
                return LayerView(self, name, get, set_)
            if u.kind == "conv" and u.bn and name == f"{u.name}_bn":
                def get(u=u):
                    return [net.params.view(u.p_gamma.name).cpu().numpy().copy(),
                            net.params.view(u.p_beta.name).cpu().numpy().copy(),
                            net.state.view(u.s_mean.name).cpu().numpy().copy(),
                            net.state.view(u.s_var.name).cpu().numpy().copy()]

                def set_(w, u=u):
                    for view, a in zip((net.params.view(u.p_gamma.name), net.params.view(u.p_beta.name),
                                        net.state.view(u.s_mean.name), net.state.view(u.s_var.name)), w):
                        view.copy_(torch.from_numpy(np.asarray(a, dtype=np.float32)))
                return LayerView(self, name, get, set_)
            if u.kind == "head" and name in self._head_layer_names(u):
                if name.endswith("_anchor"):
                    j = int(name.split("_box")[1].split("_")[0]) - 1

                    def get(u=u, j=j):
                        a = net._anchors_dev[u.name].cpu().numpy().reshape(-1, 2)[j]
                        return [a.reshape(1, 1, 1, 2).copy()]

                    def set_(w, u=u, j=j):
                        a = np.asarray(w[0], dtype=np.float32).reshape(2)
                        net._anchors_dev[u.name][2 * j:2 * j + 2] = torch.from_numpy(a).cuda()
                        u.anchors[j] = (float(a[0]), float(a[1]))
                    return LayerView(self, name, get, set_, trainable=False)
                lo, hi = self._head_rows(u, name)
                cin = u.src.c

                def get(u=u, lo=lo, hi=hi, cin=cin):
                    k = net.params.view(u.p_kernel.name).cpu().numpy().reshape(-1, cin)[lo:hi]
                    b = net.params.view(u.p_bias.name).cpu().numpy()[lo:hi]
                    return [k.T.reshape(1, 1, cin, hi - lo).copy(), b.copy()]

                def set_(w, u=u, lo=lo, hi=hi, cin=cin):
                    k = np.asarray(w[0], dtype=np.float32).reshape(cin, hi - lo).T
                    net.params.view(u.p_kernel.name).reshape(-1, cin)[lo:hi] = torch.from_numpy(
                        np.ascontiguousarray(k)).cuda()
                    net.params.view(u.p_bias.name)[lo:hi] = torch.from_numpy(np.asarray(w[1], dtype=np.float32)).cuda()
                return LayerView(self, name, get, set_)
            if u.kind not in ("conv", "head") and name == u.name:
                return LayerView(self, name, lambda: [], lambda w: None, trainable=False)
        raise ValueError(f"No such layer: {name}")

    def get_weights(self):
        out = []
        for n in self.layer_names():
            out += self.get_layer(n).get_weights()
        return out

    def set_weights(self, weights):
        i = 0
        for n in self.layer_names():
            layer = self.get_layer(n)
            k = len(layer.get_weights())
            if k:
                layer.set_weights(weights[i:i + k])
                i += k

    def set_body_weights(self, other):
        """`model_body.set_weights(pretrained_body.get_weights())` (yolov3/__init__.py:170-171): copy every
        non-head layer from another Model of the same body, by layer name."""
        for n in other.layer_names():
            if n.startswith("out") and ("_box" in n or n.endswith(("_xywhc_conv", "_prob_conv"))):
                continue
            w = other.get_layer(n).get_weights()
            if w:
                self.get_layer(n).set_weights(w)

    def set_anchors_trainable(self, trainable):
        """yolov4/__init__.py:150-159: the Anchor layers' weights join the optimizer. d(loss)/d(anchor) comes from
        the head-activation backward kernel (yolo_head_act_bwd `danchors`); the loss closures keep the anchors they
        were created with, as in the reference."""
        if trainable and not self.net.has_anchors:
            raise YoloHipError("this model has no anchors")
        self.net.anchors_trainable = bool(trainable)
        self.net.anchor_grads.zero_()

    @staticmethod
    def _npz_path(path):
        p = str(path)
        if p.endswith((".h5", ".hdf5")):
            raise YoloHipError("HDF5 weights need h5py, which is not available here; use .npz "
                               "(keys '<layer name>/<index>')")
        return p if p.endswith(".npz") else p + ".npz"

    def save_weights(self, path):
        """.npz keyed '<keras layer name>/<index>' (h5py is unavailable: SURVEY.md section 5). The same path rule
        as load_weights: '.h5' / '.hdf5' are refused on both sides, anything else gets '.npz' appended once."""
        d = {}
        for n in self.layer_names():
            for i, a in enumerate(self.get_layer(n).get_weights()):
                d[f"{n}/{i}"] = a
        np.savez(self._npz_path(path), **d)

    def load_weights(self, path, by_name=False, skip_mismatch=False):
        """Keras semantics: every parameterised layer must be in the file with matching shapes (ValueError otherwise);
        by_name=True loads the layers the file has and returns the names it skipped, skip_mismatch=True (with by_name)
        also skips layers whose shapes differ."""
        d = np.load(self._npz_path(path))
        skipped = []
        for n in self.layer_names():
            layer = self.get_layer(n)
            cur = layer.get_weights()
            if not cur:
                continue
            keys = [f"{n}/{i}" for i in range(len(cur))]
            if not all(k in d for k in keys):
                if not by_name:
                    raise ValueError(f"load_weights: layer '{n}' is missing from {path} (pass by_name=True to load a "
                                     f"partial / renamed checkpoint)")
                skipped.append(n)
                continue
            new = [d[k] for k in keys]
            if any(a.shape != b.shape for a, b in zip(new, cur)):
                if not (by_name and skip_mismatch):
                    raise ValueError(f"load_weights: layer '{n}' has shapes {[a.shape for a in new]} in the file, "
                                     f"{[b.shape for b in cur]} in the model")
                skipped.append(n)
                continue
            layer.set_weights(new)
        return skipped

    # ---- inference ----
    def __call__(self, x, training=False):
        # Keras returns fresh tensors: clone the network's persistent head buffers (the zero-copy path stays
        # internal: train_step_device / infer), so that `a = model(x1); b = model(x2)` leaves `a` intact
        outs = [t.clone() for t in self.net.forward(_to_input(x), training=training)]
        return outs[0] if self.single_output else outs

    def predict(self, x, batch_size=32, verbose=0, **_):
        if hasattr(x, "__getitem__") and hasattr(x, "__len__") and not isinstance(x, (np.ndarray, torch.Tensor)):
            chunks = [x[i][0] for i in range(len(x))]     # Sequence of (img, label) batches
        else:
            n = len(x)
            chunks = [x[i:i + batch_size] for i in range(0, n, batch_size)]
        outs = None
        for c in chunks:
            o = self.net.infer(_to_input(c))
            o = [t.cpu().numpy() for t in o]
            outs = [[a] for a in o] if outs is None else [acc + [a] for acc, a in zip(outs, o)]
        res = [np.concatenate(a, axis=0) for a in outs]
        return res[0] if self.single_output else res

    # ---- training ----
    def compile(self, optimizer="adam", loss=None, metrics=None, **_):
        if isinstance(optimizer, str):
            optimizer = {"adam": opt_mod.Adam, "sgd": opt_mod.SGD}[optimizer.lower()]()
        self.optimizer = optimizer
        self.optimizer.bind(self.net)
        nout = len(self.net.outputs)
        self.loss = list(loss) if isinstance(loss, (list, tuple)) else [loss] * nout
        if len(self.loss) != nout:
            raise ValueError(f"need {nout} losses, got {len(self.loss)}")
        if metrics is None:
            metrics = [[] for _ in range(nout)]
        elif nout == 1 and (not metrics or not isinstance(metrics[0], (list, tuple))):
            metrics = [list(metrics)]
        self.metrics = [list(m) for m in metrics]
        self._loss_bufs = [torch.zeros(8, device="cuda", dtype=torch.float64) for _ in range(nout)]
        self._dpred = None
        self._drop_step_graphs()

    def enable_data_parallel(self, process_group=None, bucket_bytes=48 << 20):
        """Call after compile() in a torch.distributed job (one process per GPU)."""
        segs, units = dp_mod.network_segments(self.net)
        self._reducer = dp_mod.GradReducer(self.net.grads, segs, process_group, bucket_bytes)
        index = {id(u): i for i, u in enumerate(units)}
        self._dp_unit_index = index
        self._drop_step_graphs()
        self.net.grad_ready_hook = lambda u: self._reducer.segment_done(index[id(u)])
        self.net.backward_begin_hook = self._reducer.backward_begin
        net = self.net
        self._reducer.extra_streams = lambda: [net._wgrad_stream] if net._wgrad_stream is not None else []
        dp_mod.broadcast_parameters([self.net.params.data, self.net.state.data], 0, process_group)
        self.net.mark_params_changed()

    def _labels(self, y):
        ys = [y] if not isinstance(y, (list, tuple)) else list(y)
        out = []
        for a in ys:
            if torch.is_tensor(a):
                out.append(a.to(device="cuda", dtype=torch.float32).contiguous())
            else:
                out.append(torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).cuda())
        return out

    # ---- the step as replayed hipGraphs (capture.py) ----
    def _drop_step_graphs(self):
        self._step_graphs = None
        self._eager_steps = {}

    def _captured_step(self, x, y_list):
        """Replay the recorded step if there is one for this configuration; record it after two eager steps of the same
        configuration (allocations, lazy initialisation and filter preparation have then happened).
        YOLO_STEP_MODE = tape (default: the launch tape of tape.py -- same launches on the same two streams, none of the
        Python), graph (hipGraphs, capture.py: bit-identical too, but this runtime serialises the two captured streams:
        35.4 vs 31.2 ms per step), eager (every step through Python; YOLO_STEP_GRAPH=0 means the same). A failed
        recording is reported once and switches the feature off for this model."""
        import os
        mode = os.environ.get("YOLO_STEP_MODE", "tape")
        if os.environ.get("YOLO_STEP_GRAPH", "1") == "0" or mode == "eager" or getattr(self, "_graphs_failed", False):
            return None
        if not getattr(self.optimizer, "capturable", False):
            return None
        if self.net.grad_ready_hook is not None and self._reducer is None:
            return None          # a foreign gradient hook: its host side would not run during replays
        from .capture import StepGraphs, StepTape
        self.net.allocate(x.shape[0])     # (the key names the allocation the recording belongs to)
        key = StepGraphs.key_of(self, x)
        g = self._step_graphs
        if g is not None and g.key == key:
            return g.replay(x, y_list)
        n = self._eager_steps.get(key, 0)
        if os.environ.get("YOLO_STEP_DEBUG"):
            import sys
            print(f"[step] mode {mode} eager steps so far {n} key {key} known {list(self._eager_steps)}", file=sys.stderr)
        if n < 2 or self._dpred is None:
            self._eager_steps = {key: n + 1}
            self._step_graphs = None
            return None
        try:
            if mode == "graph":
                self._step_graphs = StepGraphs(self, x, y_list)
                return self._step_graphs.replay(x, y_list)
            self._step_graphs = StepTape(self, x, y_list)        # recording executes this step
            return self._loss_bufs
        except Exception as e:   # e.g. a runtime that cannot capture some call: stay eager, say so once
            import sys
            import traceback
            print(f"[tf2_yolo_amd] step recording ({mode}) failed ({e!r}); training continues with eager launches\n"
                  + "".join(traceback.format_exc().splitlines(True)[-12:]), file=sys.stderr)
            self._graphs_failed = True
            self._step_graphs = None
            if mode != "graph":
                from . import ops as _ops      # a recording that broke half-way has accumulated part of the gradients:
                _ops.zero_bytes(self.net.grads)   # the eager step that follows starts from zero again
            return None

    def train_step_device(self, x, y_list, with_metrics=False):
        """One optimizer step on device-resident float32 tensors. Returns the per-output loss
        buffers (device float64[8], element 0 = loss) without synchronising. From the third step of a configuration on
        the step is a replay of captured hipGraphs (capture.py) unless metrics are asked for or an ops.TIMER is active."""
        if self.optimizer is None:
            raise YoloHipError("compile() the model before training")
        from . import ops as _ops
        if not with_metrics and _ops.TIMER is None:
            bufs = self._captured_step(x, y_list)
            if bufs is not None:
                return bufs, None
        outs = self.net.forward(x, training=True)
        if self._dpred is None or self._dpred[0].shape != outs[0].shape:
            self._dpred = [torch.empty_like(o) for o in outs]
        for i, (o, yt) in enumerate(zip(outs, y_list)):
            self.loss[i].fwd_bwd(yt, o, grad_scale=1.0, dpred=self._dpred[i], loss_out=self._loss_bufs[i])
        mvals = None
        if with_metrics:
            mvals = [[m(yt, o) for m in ms] for ms, o, yt in zip(self.metrics, outs, y_list)]
        self.net.backward(self._dpred)
        if self._reducer is not None and self.net.anchors_trainable and self._reducer.world > 1:
            import torch.distributed as dist   # a few dozen floats: one plain all-reduce
            dist.all_reduce(self.net.anchor_grads, group=self._reducer.pg)
        scale = self._reducer.finish() if self._reducer is not None else 1.0
        self.optimizer.step(grad_scale=scale)
        return self._loss_bufs, mvals

    def train_on_batch(self, x, y, return_dict=False, **_):
        bufs, mvals = self.train_step_device(_to_input(x), self._labels(y), with_metrics=bool(self.metrics))
        losses = [float(b[0].item()) for b in bufs]
        total = sum(losses)
        flat = [float(v.item()) for ms in (mvals or []) for v in ms]
        res = [total] + (losses if len(losses) > 1 else []) + flat
        return res if len(res) > 1 else res[0]

    def test_on_batch(self, x, y):
        outs = self.net.forward(_to_input(x), training=False)
        ys = self._labels(y)
        losses = [float(l(yt, o).item()) for l, o, yt in zip(self.loss, outs, ys)]
        mvals = [float(m(yt, o).item()) for ms, o, yt in zip(self.metrics, outs, ys) for m in ms]
        return [sum(losses)] + (losses if len(losses) > 1 else []) + mvals

    def _iter_batches(self, x, y, batch_size, shuffle, rng):
        if y is None:   # Sequence-like: x[i] -> (img, labels)
            order = np.arange(len(x))
            for i in order:
                yield x[int(i)]
            if hasattr(x, "on_epoch_end"):
                x.on_epoch_end()
            return
        n = len(x)
        order = rng.permutation(n) if shuffle else np.arange(n)
        ys = y if isinstance(y, (list, tuple)) else [y]
        for i in range(0, n, batch_size):
            sel = order[i:i + batch_size]
            yb = [a[sel] for a in ys]
            yield x[sel], (yb if isinstance(y, (list, tuple)) else yb[0])

    def _host_source(self, x, y, batch_size, shuffle, rng):
        """Feeder source for host-resident data, or None when the data already lives on the device."""
        if y is None:
            def batches():
                for i in range(len(x)):
                    yield x[int(i)]
                if hasattr(x, "on_epoch_end"):
                    x.on_epoch_end()
            return ("batches", batches())
        ys = list(y) if isinstance(y, (list, tuple)) else [y]
        if any(torch.is_tensor(a) and a.is_cuda for a in [x] + ys):
            return None
        n = len(x)
        order = rng.permutation(n) if shuffle else np.arange(n)
        return ("arrays", [x] + ys, order, batch_size)

    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=1, validation_data=None, shuffle=True,
            initial_epoch=0, callbacks=None, **_):
        """Keras-style training loop (README.md:241-297). Host arrays / Sequences are streamed through
        `feeder.HostFeeder` (pinned staging, copy stream) and the per-batch losses stay on the device until
        the epoch ends, so the loop never waits for the GPU; YOLO_FIT_PIPELINE=0 keeps the plain
        batch-by-batch loop (`train_on_batch` per batch)."""
        import os
        from .feeder import FeederBuffers, HostFeeder
        batch_size = batch_size or 32
        rng = np.random.default_rng(0)
        hist = {"loss": []}
        pipelined = os.environ.get("YOLO_FIT_PIPELINE", "1") != "0"
        # Keras callbacks (README.md:262-275 passes ModelCheckpoint-style objects): the epoch-level hooks are called
        # with Keras' signatures; batch-level hooks would force a host sync per batch and are not called
        cbs = list(callbacks or [])

        def _cb(hook, *args):
            for c in cbs:
                f = getattr(c, hook, None)
                if f is not None:
                    f(*args)
        for c in cbs:
            if hasattr(c, "set_model"):
                c.set_model(self)
        self.stop_training = False
        _cb("on_train_begin", {})
        for ep in range(initial_epoch, epochs):
            t0 = time.time()
            _cb("on_epoch_begin", ep, {})
            tot, nb = 0.0, 0
            src = self._host_source(x, y, batch_size, shuffle, rng) if pipelined else None
            if src is not None:
                if getattr(self, "_feed_bufs", None) is None:
                    self._feed_bufs = FeederBuffers()
                feeder = HostFeeder(src, self._feed_bufs)
                tot_dev = torch.zeros((), dtype=torch.float64, device="cuda")
                try:
                    for xb, ys in feeder:
                        bufs, _m = self.train_step_device(xb, ys)
                        for b in bufs:
                            tot_dev += b[0]
                        nb += 1
                finally:
                    feeder.close()
                tot = float(tot_dev.item())
            else:
                for xb, yb in self._iter_batches(x, y, batch_size, shuffle, rng):
                    r = self.train_on_batch(xb, yb)
                    tot += r[0] if isinstance(r, list) else r
                    nb += 1
            hist["loss"].append(tot / max(nb, 1))
            msg = f"Epoch {ep + 1}/{epochs} - {time.time() - t0:.1f}s - loss: {hist['loss'][-1]:.4f}"
            if validation_data is not None:
                v = self.evaluate(validation_data[0], validation_data[1], batch_size=batch_size, verbose=0)
                hist.setdefault("val_loss", []).append(v[0] if isinstance(v, list) else v)
                msg += f" - val_loss: {hist['val_loss'][-1]:.4f}"
            if verbose:
                print(msg)
            _cb("on_epoch_end", ep, {k: v[-1] for k, v in hist.items()})
            if self.stop_training:
                break
        _cb("on_train_end", {})
        self.history = type("History", (), {"history": hist})()
        return self.history

    def evaluate(self, x=None, y=None, batch_size=None, verbose=1, **_):
        batch_size = batch_size or 32
        acc, nb = None, 0
        for xb, yb in self._iter_batches(x, y, batch_size, False, None):
            r = self.test_on_batch(xb, yb)
            acc = r if acc is None else [a + b for a, b in zip(acc, r)]
            nb += 1
        res = [a / max(nb, 1) for a in acc]
        if verbose:
            print("evaluate:", res)
        return res if len(res) > 1 else res[0]
