"""Optimizers over the flat parameter buffer (one fused HIP launch per step).

`Adam` mirrors tf.keras.optimizers.Adam (README.md:241: `Adam(lr=1e-4)`): defaults beta_1=0.9,
beta_2=0.999, epsilon=1e-7, bias-corrected, no weight decay."""
import torch

from . import ops


class Optimizer:
    def bind(self, net):
        self.net = net

    def step(self, grad_scale=1.0):
        raise NotImplementedError


class Adam(Optimizer):
    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, lr=None, **_):
        self.learning_rate = float(lr if lr is not None else learning_rate)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        self.iterations = 0
        self.m = self.v = None

    def bind(self, net):
        self.net = net
        self.m = torch.zeros_like(net.params.data)
        self.v = torch.zeros_like(net.params.data)
        self.am = torch.zeros_like(net.anchors_flat)   # trainable anchors (v4): one more small tensor
        self.av = torch.zeros_like(net.anchors_flat)

    def step(self, grad_scale=1.0):
        self.iterations += 1
        self.net.before_param_write()
        ops.adam_step(self.net.params.data, self.net.grads, self.m, self.v, self.learning_rate, self.iterations,
                      self.beta_1, self.beta_2, self.epsilon, grad_scale=grad_scale, zero_grad=True)
        if self.net.anchors_trainable and self.net.has_anchors:
            ops.adam_step(self.net.anchors_flat, self.net.anchor_grads, self.am, self.av, self.learning_rate,
                          self.iterations, self.beta_1, self.beta_2, self.epsilon, grad_scale=grad_scale, zero_grad=True)
        self.net.mark_params_changed()


    # ---- the form a captured hipGraph replays (capture.py): the five scalars live in device memory ----
    capturable = True

    _HYPER_SLOTS = 64

    def _hyper_buffers(self):
        if getattr(self, "_hyper_dev", None) is None:
            # A RING of pinned rows, one per step in flight: the upload is asynchronous and reads its row when the stream
            # gets to it, which can be several steps after the host wrote it (a loop that does not synchronise per step runs
            # ahead of the GPU by as many launches as the queue holds). One row re-written every step -- the form until round
            # 6 -- let step k's upload read the scalars of step k + 1 .. k + 3: a learning rate with the wrong bias
            # correction, a different one from run to run (scripts/step_repro.py with REPRO_SYNC=0: all parameters of
            # YOLOv2-416 differ after the third step by 4e-6 .. 1e-5, the loss after 11 steps takes one of four values).
            self._hyper_host = torch.zeros(self._HYPER_SLOTS, 5, dtype=torch.float32).pin_memory()
            self._hyper_events = [None] * self._HYPER_SLOTS
            self._hyper_dev = torch.zeros(5, dtype=torch.float32, device="cuda")
        return self._hyper_host, self._hyper_dev

    def refresh_hyper(self, grad_scale=1.0):
        """host side of one captured step: advance the step counter, upload {lr_t, beta1, beta2, eps, grad_scale}
        (lr_t computed exactly as yolo_adam_step computes it) on the current stream, ahead of the replay that reads it"""
        self.iterations += 1
        ring, dev = self._hyper_buffers()
        slot = self.iterations % self._HYPER_SLOTS
        if self._hyper_events[slot] is not None:
            self._hyper_events[slot].synchronize()   # (its upload of 64 steps ago has long run: returns at once)
        host = ring[slot]
        host[0] = ops.adam_lr_t(self.learning_rate, self.iterations, self.beta_1, self.beta_2)
        host[1], host[2], host[3], host[4] = self.beta_1, self.beta_2, self.epsilon, float(grad_scale)
        dev.copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._hyper_events[slot] = ev

    def step_captured(self):
        """enqueue (inside a stream capture) the update with the scalars read from the device"""
        _, dev = self._hyper_buffers()
        ops.adam_step_dev(self.net.params.data, self.net.grads, self.m, self.v, dev, zero_grad=True)
        if self.net.anchors_trainable and self.net.has_anchors:
            ops.adam_step_dev(self.net.anchors_flat, self.net.anchor_grads, self.am, self.av, dev, zero_grad=True)


class SGD(Optimizer):
    def __init__(self, learning_rate=0.01, lr=None, **_):
        self.learning_rate = float(lr if lr is not None else learning_rate)
        self.iterations = 0

    def step(self, grad_scale=1.0):
        self.iterations += 1
        self.net.before_param_write()
        ops.sgd_step(self.net.params.data, self.net.grads, self.learning_rate, grad_scale=grad_scale, zero_grad=True)
        if self.net.anchors_trainable and self.net.has_anchors:
            ops.sgd_step(self.net.anchors_flat, self.net.anchor_grads, self.learning_rate, grad_scale=grad_scale,
                         zero_grad=True)
        self.net.mark_params_changed()
