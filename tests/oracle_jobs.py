"""CPU oracle passes of the full-batch headline tests, run in WORKER PROCESSES beside the GPU tests (test infrastructure;
the GPU suite's wall time was dominated by the host: 116 + 66 s of float64 / float32 oracle forwards during which the GPU
idled). conftest.py starts the two jobs at the beginning of a GPU session; tests/test_gpu_fullsize.py collects the results.
Workers never touch the GPU (HIP_VISIBLE_DEVICES is emptied before torch is imported in them).
(Nothing here imports torch at module level: the spawned worker sets its environment first.)"""
import os


def worker_init():
    os.environ["HIP_VISIBLE_DEVICES"] = ""
    os.environ["CUDA_VISIBLE_DEVICES"] = ""


def oracle_fns(version, g0, unbiased=True, anchors9=None, class_num=3):
    """(forward, per-level oracle losses) of the YOLOv3 / YOLOv4 end-to-end cases: the CPU half of
    tests/test_gpu_model.py:_setup, usable without a device"""
    from oracle import losses as OL
    from oracle import models as OM
    A9 = anchors9
    if version == 3:
        fwd = lambda w, x, tr, m=None: OM.yolov3_forward(w, x, A9, training=tr, leaky_masks=m, unbiased_moving_var=unbiased)
        loss_o = [OL.wrap_yolo_loss_v3((g0 * 2 ** i, g0 * 2 ** i), 3, class_num, anchors=A9[3 * i:3 * i + 3],
                                       loss_weight=[1, 1, 5, 1]) for i in range(3)]
    elif version == 4:
        fwd = lambda w, x, tr, m=None: OM.yolov4_forward(w, x, A9, training=tr, leaky_masks=m, unbiased_moving_var=unbiased)
        loss_o = [OL.wrap_yolo_loss_v4((g0 * 2 ** i, g0 * 2 ** i), 3, class_num, anchors=A9[3 * i:3 * i + 3],
                                       loss_weight=[1, 5, 1]) for i in range(3)]
    else:
        raise ValueError(version)
    return fwd, loss_o


def headline_job(version, g0, anchors9, w, x, ys, threads, class_num=3):
    """float64 and float32 training-mode forward + losses of one headline configuration at its true batch, without autograd
    and without retaining activations. Returns numpy arrays only."""
    import time
    import torch
    from oracle import models as OM
    t0 = time.time()
    torch.set_num_threads(threads)
    fwd, loss_o = oracle_fns(version, g0, True, anchors9, class_num)
    keep = OM.KEEP_ACTS
    OM.KEEP_ACTS = False      # (restored below: this also runs in-process, tests/test_oracle_jobs_cpu.py and the single-test fallback)
    try:
        with torch.no_grad():
            ref, ctx = fwd({k: torch.tensor(v, dtype=torch.float64) for k, v in w.items()}, torch.tensor(x, dtype=torch.float64), True)
            ref_losses = [float(lf(torch.tensor(yt, dtype=torch.float64), o)) for lf, yt, o in zip(loss_o, ys, ref)]
            t1 = time.time()
            o32, _ = fwd({k: torch.tensor(v) for k, v in w.items()}, torch.tensor(x), True)
            l32 = [float(lf(torch.tensor(yt), o)) for lf, yt, o in zip(loss_o, ys, o32)]
    finally:
        OM.KEEP_ACTS = keep
    return {"ref": [o.numpy() for o in ref], "ref_losses": ref_losses, "o32": [o.numpy() for o in o32], "l32": l32,
            "moving": {k: (mm.numpy(), mv.numpy()) for k, (mm, mv) in ctx.moving.items()},
            "seconds": (t1 - t0, time.time() - t1), "threads": threads}
