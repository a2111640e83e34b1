#!/usr/bin/env python3
"""Headline benchmark: images/sec training YOLOv3 416x416 bs=32/GPU (BASELINE.json).

One "step" = forward (training-mode BN) + the three fused loss/gradient kernels + backward +
gradient all-reduce (N > 1, RCCL over xGMI, overlapped) + Adam, on a synthetic batch that is
already resident in HBM. Prints ONE JSON line (rank 0) with the `roofline` of the dominant kernel
(the conv kernel variant with the most device time: fp16x3 "planes" MFMA implicit GEMM) measured with
HIP events inside the timed region -- beside the spec-peak fraction it carries `held_clock_ceiling`, the raw
fp16 MFMA rate a bare MFMA loop sustains on random data on THIS box right after the timed region
(yolo_mfma_probe) -- and, at N = 1, a `cpu_baseline` (the torch-CPU restatement of the same training step on
the host cores; tf.keras is not installable here).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

# More hardware queues than the runtime's default of 4: a data-parallel rank runs four or five streams, and with 4 queues the two
# compute streams of the step shared one (34.2 instead of 30.1 ms per step; tf2_yolo_amd/__init__.py). Read at HIP initialisation.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 MFMA peak
SPLIT_PASSES = 6                # conv_split.hip: one fp32 product = 6 bf16 MFMA passes (exact 3-way split)
PLANES_PASSES = 3               # conv_planes.hip: one product = 3 fp16 MFMA passes (two scaled fp16 planes)


def kernel_peak(name):
    """Peak of ALGORITHMIC (fp32-equivalent) TFLOP/s for a kernel: the fp32-MFMA peak for the fp32-MFMA kernels,
    the 16-bit MFMA peak / 6 for the bf16x6 split kernels, / 3 for the fp16x3 planes kernels (each algorithmic
    FLOP costs that many 16-bit MFMA FLOPs)."""
    if "planes" in name:     # (also conv_win_planes_kernel: the 3x3 window kernel, same 3 passes)
        return BF16_MFMA_PEAK_TFLOPS / PLANES_PASSES
    return BF16_MFMA_PEAK_TFLOPS / SPLIT_PASSES if "split" in name else FP32_MFMA_PEAK_TFLOPS
BATCH = 32
HW = 416
CLASSES = 80


def hbm_traffic_from_profile(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC passes (bench.py cannot
    collect PMC itself; FETCH_SIZE and WRITE_SIZE need separate passes). None if no profile is committed."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    if not paths:
        return None
    prof = json.load(open(paths[-1]))["kernels"]
    if kernel in prof:           # split kernels: the timer name is the rocprof name without "void yolo::" and spaces
        return prof[kernel]["hbm_bytes_per_launch"]
    # the timer's name is the kernel's leading template arguments (tile shape); rocprof's name carries the trailing ones too
    # (ring depth, diagnostic flags, class / split instantiations): launch-weighted mean over the instantiations of the shape
    fam = [v for n, v in prof.items() if n.startswith(kernel[:-1] + ",") and kernel.endswith(">")]
    cnt = sum(v["launches"] for v in fam)
    if cnt:
        return int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in fam) / cnt)
    base, _, rest = kernel.partition("<")
    dims = rest.rstrip(">").split(",")
    flat = "true" if "flat" in dims else "false"
    if base == "wgrad_win_planes_kernel":   # timer name <128,288[,mfma16]> = rocprof wgrad_win_kernel<ring,..> / wgrad_win16_kernel<ring,..>
        pre = "wgrad_win16_kernel<" if "mfma16" in dims else "wgrad_win_kernel<"
        fam = [v for n, v in prof.items() if n.startswith(pre)]
        cnt = sum(v["launches"] for v in fam)
        return int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in fam) / cnt) if cnt else None
    if base == "conv_win_planes_kernel":   # timer name <rows,128> = rocprof conv_win_kernel<rows/64, window chunks, false>
        fam = [v for n, v in prof.items() if n.startswith(f"conv_win_kernel<{int(dims[0]) // 64},")]
        cnt = sum(v["launches"] for v in fam)
        return int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in fam) / cnt) if cnt else None
    if base == "gather_conv_kernel" and len(dims) >= 2:
        for name, v in prof.items():
            if name.startswith(f"{base}<{dims[0]},{dims[1]},") and name.endswith(f",{flat}>"):
                return v["hbm_bytes_per_launch"]
    if base == "wgrad_kernel":   # the timer aggregates the fp32 wgrad tile shapes; the profile has one entry per shape
        tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for n, v in prof.items() if n.startswith("wgrad_kernel"))
        cnt = sum(v["launches"] for n, v in prof.items() if n.startswith("wgrad_kernel"))
        return int(tot / cnt) if cnt else None
    return None


def pmc_from_profile(kernel):
    """MFMA-busy %, held clock and L2 hit rate of the kernel family from the newest committed PMC passes
    (profiles/r*_conv_pmc.json, scripts/pmc_round.sh: the standalone layer benchmark under rocprofv3 --pmc; bench.py
    cannot collect counters itself). Averages over the profiled layers of that family; None if there is no profile."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_conv_pmc.json")))
    if not paths:
        return None
    fam = ("wgradwin" if kernel.startswith("wgrad_win_planes_kernel") else
           "wgrad128x128" if kernel.startswith("wgrad_planes_kernel<128,128") else
           "wgrad64x128" if kernel.startswith("wgrad_planes_kernel<64,128") else
           "win128" if kernel.startswith("conv_win_planes_kernel<128") else
           "win256" if kernel.startswith("conv_win_planes_kernel<256") else
           "planes128x128" if kernel.startswith("gather_conv_planes_kernel<128,128") else
           "planes128x64" if kernel.startswith("gather_conv_planes_kernel<128,64") else
           "planes128x32" if kernel.startswith("gather_conv_planes_kernel<128,32") else None)
    if fam is None:
        return None
    rows = [k for k in json.load(open(paths[-1]))["kernels"] if k["label"].startswith(fam) and "mfma_busy_pct" in k["derived"]]
    if not rows:
        return None
    avg = lambda key: round(sum(k["derived"][key] for k in rows if key in k["derived"]) / len(rows), 3)
    return {"mfma_busy_pct": avg("mfma_busy_pct"), "effective_clock_ghz": avg("effective_clock_ghz"),
            "l2_hit_rate": avg("l2_hit_rate"), "lds_active_pct": avg("lds_active_pct"),
            "layers": [k["layer_H_Cin_Cout_k_s_N"] + " " + k["mode"] for k in rows],
            "source": os.path.basename(paths[-1]) + " (standalone launches under rocprofv3 --pmc, SQ_VALU_MFMA_BUSY_CYCLES / 1024 "
                                                    "SIMDs over GRBM_GUI_ACTIVE / 8 XCDs)"}


def cpu_baseline(threads, n=BATCH, timed=1, weights=None, batch=None):
    """Bounded sample of the SAME workload on the host CPU: training steps (forward, loss, autograd backward) of the torch-CPU
    restatement of the reference graph, fp32, at the benchmark's own batch of 32 (rounds 4-5 ran batch 16). One warm-up step
    OUTSIDE the timer (oneDNN primitive creation, allocator growth, page faults), then `timed` steps of ~17 s each;
    value = batch / median step time."""
    from oracle import losses as OL
    from oracle import models as OM
    from tf2_yolo_amd import graphs, labels
    torch.set_num_threads(threads)
    if batch is not None:     # the device run's own batch (rank 0) and, below, its initial weights: the first-step losses of the
        x, ys = batch         # two executions can then be compared (this CPU step is the oracle's, in fp32)
        n = x.shape[0]
    else:
        rng = np.random.default_rng(1234)
        x, ys = labels.synthetic_batch(rng, n, (HW, HW), CLASSES)
    if weights is not None:
        w = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in weights.items()}
    else:
        b = graphs.build_yolov3((HW, HW, 3), CLASSES)
        w = {k: torch.from_numpy(v) for k, v in labels.synthetic_keras_weights(b, 1234).items()}
    for v in w.values():
        v.requires_grad_(True)
    anchors = graphs.V3_DEFAULT_ANCHORS
    lossf = [OL.wrap_yolo_loss_v3((13 * 2 ** i, 13 * 2 ** i), 3, CLASSES, anchors=anchors[3 * i:3 * i + 3],
                                  loss_weight=[1, 1, 5, 1]) for i in range(3)]
    xt = torch.from_numpy(x)
    yts = [torch.from_numpy(y) for y in ys]

    losses = []

    def step():
        for v in w.values():
            v.grad = None
        t0 = time.perf_counter()
        outs, _ = OM.yolov3_forward(w, xt, anchors, training=True)
        total = sum(f(y, o) for f, y, o in zip(lossf, yts, outs))
        total.backward()
        losses.append(float(total.detach()))
        return time.perf_counter() - t0

    warm = step()
    times = sorted(step() for _ in range(timed))
    dt = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
    return {"value": round(n / dt, 4), "unit": "images/s", "cores": threads, "kind": "port", "batch": n,
            "loss": losses[0], "weights": "the device run's initial weights and batch" if weights is not None else "synthetic_keras_weights(1234)",
            "step_seconds": {"warmup_untimed": round(warm, 2), "timed": [round(t, 2) for t in times]},
            "sample": f"{timed} timed training step(s) after 1 untimed warm-up step (fwd+loss+autograd bwd, no optimizer) of the "
                      f"torch-CPU/oneDNN fp32 restatement of the reference YOLOv3 graph at the benchmark's batch {n}, 416x416 C=80, "
                      f"median {dt:.1f} s per step on {threads} threads; NOT tf.keras (TensorFlow is not installable in this "
                      f"pipeline)"}


def decode_nms_block():
    """BASELINE.json config 5 (outside the timed region): decode + the three NMS modes on BASELINE.md's uniform-noise
    levels, GPU milliseconds beside the reference's own CPU milliseconds (tests/golden/tools_timing.json, produced in
    the build container by tests/golden/make_timing.py -- the reference does not travel to the GPU box)."""
    from tf2_yolo_amd import tools
    path = os.path.join(ROOT, "tests", "golden", "tools_timing.json")
    cpu = json.load(open(path)) if os.path.exists(path) else None
    rng = np.random.default_rng(1234)
    lv = [torch.from_numpy(rng.random((g, g, 255), dtype=np.float32)).cuda() for g in (13, 26, 52)]

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, r

    cases = []
    for thr, n in ((0.9, 10), (0.5, 3)):
        t_dec, dec = timed(lambda: tools.decode_device(*lv, class_num=80, threshold=thr, version=3), n)
        c = {"conf_threshold": thr, "candidates": int(dec.shape[0]), "gpu_decode_ms": round(t_dec, 3)}
        for name, fn in (("nms", lambda: tools.nms(dec, class_num=80, nms_threshold=0.5)),
                         ("diou_nms", lambda: tools.nms(dec, class_num=80, nms_threshold=0.5, iou_mode=2)),
                         ("soft_nms", lambda: tools.soft_nms(dec, class_num=80, nms_threshold=0.5, conf_threshold=thr,
                                                             sigma=0.5))):
            t, out = timed(fn, n)
            c[f"gpu_{name}_ms"] = round(t, 3)
            c[f"{name}_kept"] = int(out.shape[0])
        if cpu is not None:
            ref = next((r for r in cpu["cases"] if r["conf_threshold"] == thr), None)
            if ref is not None:
                c["cpu_reference"] = {k: ref[k] for k in ref if k.endswith("_ms") or k.endswith("_kept") or k == "candidates"}
                c["same_counts_as_reference"] = bool(ref["candidates"] == c["candidates"] and all(
                    ref[f"{m}_kept"] == c[f"{m}_kept"] for m in ("nms", "diou_nms", "soft_nms")))
        cases.append(c)
    return {"what": "YOLOv3-416 shaped uniform-noise predictions (13,26,52 levels, C=80), decode + NMS; GPU = this library "
                    "(results bit-identical to the reference's on the golden vectors, tests/test_gpu_decode_nms.py), "
                    "CPU = the reference's utils.tools on one core of the build container",
            "cpu_host": None if cpu is None else cpu["host"], "cases": cases}


def strict_fp32_block(steps=5, warmup=3):
    """The reference computes in fp32 (yolov3/models/backbone.py:27-55, default float32 Keras layers). The headline value runs
    the convolutions as two scaled fp16 planes x 3 MFMA passes; this block is the SAME training step with every convolution on
    the fp32-input matrix instructions (YOLO_CONV_MODE=fp32: v_mfma_f32_32x32x2_f32, bit-exact fp32 FMA chains), so that the
    strict-fp32 rate is in the driver's record too -- and, beside it, the headline arithmetic run through the SAME child
    program on the same seeds for the same warmup + steps: the loss of the very first step (identical weights and batch: the
    two arithmetics must agree to 1e-5 relative there) and the loss after all of them (where a LeakyReLU network under Adam has
    started to diverge chaotically, between two exact arithmetics as much as between either and the planes).
    The switch is read when the package is imported: child processes."""
    import subprocess
    res = {}
    t0 = time.perf_counter()
    for key, mode in (("fp32", {"YOLO_CONV_MODE": "fp32"}), ("planes", {})):
        env = dict(os.environ, **mode)
        env.pop("YOLO_DP_FORCE", None)
        if not mode:
            env.pop("YOLO_CONV_MODE", None)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--plain", "--steps", str(steps), "--warmup", str(warmup)],
                             env=env, capture_output=True, text=True, timeout=420)
        if out.returncode != 0:
            return {"error": key + ": " + out.stderr[-400:]}
        res[key] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    j = dict(res["fp32"])
    n = j["warmup_steps_run"] + j["steps"]
    l1a, l1b = res["fp32"]["loss_first_step"], res["planes"]["loss_first_step"]
    j.update({"what": "the same training step, every convolution on the fp32-input MFMA (YOLO_CONV_MODE=fp32), child process, "
                      "outside the timed region", "peak_tflops_fp32_mfma": FP32_MFMA_PEAK_TFLOPS,
              "frac_of_fp32_mfma_peak_whole_step": round(STEP_TFLOP / (j["ms_per_step"] * 1e-3) / FP32_MFMA_PEAK_TFLOPS, 4),
              "same_steps_headline_arithmetic": {
                  "what": f"the headline arithmetic (fp16x2 planes, 3 MFMA passes) through the same child program: same seeds, same "
                          f"{n} steps",
                  "ms_per_step": res["planes"]["ms_per_step"], "images_per_s": res["planes"]["images_per_s"],
                  "loss_first_step": l1b, f"loss_after_{n}_steps": res["planes"]["loss"]},
              f"loss_after_{n}_steps": j["loss"],
              "first_step_loss_rel_diff_planes_vs_fp32": abs(l1a - l1b) / max(abs(l1a), 1e-30),
              "first_step_losses_agree_to_1e-5": bool(abs(l1a - l1b) <= 1e-5 * abs(l1a)),
              "wall_s": round(time.perf_counter() - t0, 1)})
    return j


STEP_TFLOP = 6.31   # 197.3 GFLOP per image and training step at C = 80, bs 32


def darknet53_block(model, x, ceiling):
    """BASELINE.json / north_star's second figure is defined on the Darknet-53 FORWARD (yolov3/models/backbone.py:74-82: conv1 +
    five residual stages, 52 convolutions, 49.032 GFLOP per image = 1.569 TFLOP at bs 32): the training-mode forward stopped
    behind block5's last unit. ms = wall over 5 passes; conv rate = algorithmic conv FLOPs / summed conv-launch time (HIP events)."""
    from tf2_yolo_amd import ops
    net, last = model.net, "block5_4_3x3"
    net.forward(x, training=True, stop_after=last)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        net.forward(x, training=True, stop_after=last)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    tm = ops.KernelTimer()
    ops.TIMER = tm
    for _ in range(2):
        net.forward(x, training=True, stop_after=last)
    torch.cuda.synchronize()
    ops.TIMER = None
    fa = tm.summary()
    fl = sum(v["flops"] for v in fa.values()) / 2
    kms = sum(v["ms"] for v in fa.values()) / 2
    pk = sum(v["flops"] / 2 / kernel_peak(k) for k, v in fa.items())
    raw = PLANES_PASSES * sum(v["flops"] for k, v in fa.items() if "planes" in k) / 2 / kms / 1e9
    out = {"what": "Darknet-53 backbone only (conv1 .. block5, 52 conv-BN-Leaky units), training-mode forward, bs %d" % x.shape[0],
           "ms": round(ms, 3), "conv_gflop": round(fl / 1e9, 1), "conv_kernel_ms": round(kms, 3),
           "conv_tflops": round(fl / kms / 1e9, 1), "conv_frac_of_peak": round(pk / 1e9 / kms, 4),
           "whole_tflops": round(fl / ms / 1e9, 1), "whole_frac_of_peak": round(pk / 1e9 / ms, 4),
           "ms_at_60pct_of_fp32_mfma_peak_SURVEY_8d": 16.6,
           "images_per_s_forward": round(x.shape[0] / ms * 1e3, 1)}
    if ceiling and "error" not in ceiling:
        out["conv_frac_of_held_clock_ceiling"] = round(raw / ceiling["raw_fp16_mfma_tflops"], 4)
    return out


def configs_block():
    """BASELINE.json configs[0], [1], [3], [4] (the headline configs[2] is the timed region), driver-observed: C1 / C2 / C4 one
    training step at their true batch, C5 = bs-1 Model.predict (hipGraph replay) + decode + the three NMS modes on the network's
    OWN prediction (README.md:296-334), beside the reference's CPU timings for the same network and image
    (tests/golden/tools_timing.json 'model_output', produced in the build container by tests/golden/make_timing.py)."""
    from tf2_yolo_amd import graphs, labels, optimizers, tools
    res = {}

    def train_ms(yolo, loss, batch, levels, stride, steps):
        m = yolo.model
        m.compile(optimizer=optimizers.Adam(learning_rate=1e-4), loss=loss)
        rng = np.random.default_rng(1234)
        H = yolo.input_shape[0]
        xh, yh = labels.synthetic_batch(rng, batch, (H, H), yolo.class_num, levels=levels, finest_stride=stride)
        xd, yd = torch.from_numpy(xh).cuda(), [torch.from_numpy(a).cuda() for a in yh]
        for _ in range(3):
            m.train_step_device(xd, yd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            bufs, _ = m.train_step_device(xd, yd)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        lv = float(sum(b[0].item() for b in bufs))
        return {"batch": batch, "ms_per_step": round(dt * 1e3, 3), "images_per_s": round(batch / dt, 1),
                "loss_finite": bool(np.isfinite(lv))}

    import yolov1_5
    y = yolov1_5.Yolo((224, 224, 3), ["raccoon"])
    y.create_model(bbox_num=2)
    res["c1_yolov1_5_224_bs4_train"] = train_ms(y, y.loss(binary_weight=0.5), 4, 1, 56, 8)
    del y
    import yolov2
    y = yolov2.Yolo((416, 416, 3), [f"c{i}" for i in range(20)])
    y.create_model()
    res["c2_yolov2_416_bs16_train"] = train_ms(y, y.loss(), 16, 1, 32, 8)
    del y
    torch.cuda.empty_cache()
    import yolov4
    y = yolov4.Yolo((608, 608, 3), [f"c{i}" for i in range(80)])
    y.create_model(anchors=graphs.V4_DEFAULT_ANCHORS, pretrained_body=None)
    res["c4_yolov4_608_bs16_train"] = train_ms(y, y.loss(), 16, 3, 8, 5)
    del y
    torch.cuda.empty_cache()

    import yolov3
    y = yolov3.Yolo((416, 416, 3), [f"c{i}" for i in range(80)])
    y.create_model(pretrained_body=None)
    m = y.model
    w = labels.synthetic_keras_weights(graphs.build_yolov3((416, 416, 3), 80), 1234, residual_gamma=0.1)
    for n in m.layer_names():
        lay = m.get_layer(n)
        k = len(lay.get_weights())
        if k:
            lay.set_weights([w[f"{n}/{i}"] for i in range(k)])
    x1 = torch.from_numpy(np.random.default_rng(1234).random((1, 416, 416, 3), dtype=np.float32)).cuda()

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, r

    t_fwd, outs = timed(lambda: m.net.infer(x1), 20)
    lv = [outs[2][0], outs[1][0], outs[0][0]]           # README.md:320-325: fine -> coarse
    t_dec, dec = timed(lambda: tools.decode_device(*lv, class_num=80, threshold=0.5, version=3), 10)
    c5 = {"input": "the network's own bs-1 prediction (synthetic_keras_weights seed 1234, rng(1234) pixels)",
          "predict_ms_hipgraph": round(t_fwd, 3), "conf_threshold": 0.5, "candidates": int(dec.shape[0]),
          "gpu_decode_ms": round(t_dec, 3)}
    for name, fn in (("nms", lambda: tools.nms(dec, class_num=80, nms_threshold=0.5)),
                     ("diou_nms", lambda: tools.nms(dec, class_num=80, nms_threshold=0.5, iou_mode=2)),
                     ("soft_nms", lambda: tools.soft_nms(dec, class_num=80, nms_threshold=0.5, conf_threshold=0.5, sigma=0.5))):
        t, out = timed(fn, 5)
        c5[f"gpu_{name}_ms"] = round(t, 3)
        c5[f"{name}_kept"] = int(out.shape[0])
    path = os.path.join(ROOT, "tests", "golden", "tools_timing.json")
    ref = json.load(open(path)).get("model_output") if os.path.exists(path) else None
    if ref is not None:
        c5["cpu_reference"] = {k: ref[k] for k in ref if k.endswith("_ms") or k.endswith("_kept") or k == "candidates"}
        c5["end_to_end_ms"] = {"gpu": round(t_fwd + t_dec + c5["gpu_nms_ms"], 3),
                               "cpu_reference_decode_plus_nms_only": round(ref["decode_ms"] + ref["nms_ms"], 1)}
    res["c5_yolov3_416_bs1_predict_decode_nms"] = c5
    return res


def launch_ranks(args):
    """`python bench.py --gpus N` without a torchrun environment: start N fresh child processes of this file, one rank per
    GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and GPU_MAX_HW_QUEUES in their environment), relay rank 0's JSON line and
    return the exit code (non-zero if any rank failed). The parent never touches the GPU (torch.cuda.device_count() does not
    initialise HIP on this image) and never re-execs itself. With fewer visible devices than ranks the job degrades to the
    visible count and says so; returns None when that leaves one rank (the caller then runs the N = 1 job itself).
    YOLO_BENCH_SINGLE_DEVICE=1 (rehearsal: every rank on device 0, YOLO_DIST_BACKEND=gloo) keeps N."""
    import socket
    import subprocess
    n = args.gpus
    single_dev = os.environ.get("YOLO_BENCH_SINGLE_DEVICE") == "1"
    visible = torch.cuda.device_count()
    if not single_dev and visible < n:
        print(f"[bench] --gpus {n} asked for, {visible} device(s) visible: running {max(visible, 1)} rank(s)",
              file=sys.stderr, flush=True)
        n = max(visible, 1)
    if n == 1:
        args.gpus = 1
        return None
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = [sys.executable, os.path.abspath(__file__), "--gpus", str(n), "--steps", str(args.steps), "--warmup", str(args.warmup),
            "--batch", str(args.batch)]
    for flag in ("no_cpu_baseline", "no_kernel_timer", "no_extra_blocks", "plain"):
        if getattr(args, flag):
            argv.append("--" + flag.replace("_", "-"))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.setdefault("GPU_MAX_HW_QUEUES", "8")
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's stdout is the job's stdout; a rank that dies takes the others down (they would wait in a collective for ever)
    import threading
    out0 = []
    rd = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    code = 0
    alive = list(procs)
    try:
        while alive:
            for p in list(alive):
                rc = p.poll()
                if rc is None:
                    continue
                alive.remove(p)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 1
                    print(f"[bench] rank {procs.index(p)} exited with {rc}: stopping the other ranks", file=sys.stderr, flush=True)
                    for q in alive:
                        q.terminate()          # (exactly the PIDs started here)
            time.sleep(0.2)
    finally:
        for q in alive:                        # the launcher itself is going away (signal, exception): no orphaned ranks
            if q.poll() is None:
                q.terminate()
    rd.join(timeout=10)
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    return code


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-extra-blocks", action="store_true", help="skip strict_fp32 / configs (profiling runs)")
    ap.add_argument("--plain", action="store_true", help="timed region only; prints {ms_per_step, images_per_s, loss}")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process (which has made NO GPU call) becomes the launcher of N ranks
        code = launch_ranks(args)
        if code is not None:
            raise SystemExit(code)
        # (one visible device and no single-device rehearsal: fall through as the N = 1 job, said on stderr)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node "
                         f"{args.gpus}, or unset WORLD_SIZE and let bench.py start its own ranks")
    # YOLO_BENCH_SINGLE_DEVICE=1 + YOLO_DIST_BACKEND=gloo: rehearse the N > 1 code path (buckets, side
    # stream, events, 1/world) with several ranks on ONE GPU; the real runs use RCCL, one rank per GPU
    single_dev = os.environ.get("YOLO_BENCH_SINGLE_DEVICE") == "1"
    backend = os.environ.get("YOLO_DIST_BACKEND", "nccl")
    dev_index = 0 if single_dev else local_rank
    torch.cuda.set_device(dev_index)
    from tf2_yolo_amd import ops as _ops_early
    _ops_early.create_side_streams()   # BEFORE RCCL creates its streams: the step's two compute streams get hardware queues of their own
    import torch.distributed as dist
    # YOLO_DP_FORCE=1: initialise RCCL and run the gradient exchange in a world of ONE rank too
    # (tests/test_gpu_dp.py: the RCCL calls, side stream and bucket views on a single GPU)
    force_dp = os.environ.get("YOLO_DP_FORCE") == "1"
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import yolov3
    from tf2_yolo_amd import labels, ops, optimizers

    yolo = yolov3.Yolo((HW, HW, 3), [f"c{i}" for i in range(CLASSES)])
    yolo.create_model(pretrained_body=None, seed=1234)            # same weights on every rank
    model = yolo.model
    model.compile(optimizer=optimizers.Adam(learning_rate=1e-4), loss=yolo.loss())   # README.md:241
    if world > 1 or force_dp:
        model.enable_data_parallel()

    rng = np.random.default_rng(1234 + rank)                     # seed = 1234 + rank
    x_h, ys_h = labels.synthetic_batch(rng, args.batch, (HW, HW), CLASSES)
    x = torch.from_numpy(x_h).cuda()
    ys = [torch.from_numpy(y).cuda() for y in ys_h]

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    log(f"model built: {model.trainable_count()} trainable params; data ready")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # W untimed steps (at least three: the step is captured into hipGraphs after two eager steps of a configuration,
    # tf2_yolo_amd/capture.py, and the capture itself must not fall into the timed region)
    warm_run = max(args.warmup, 3)
    loss_first = None
    # the initial weights, for the CPU oracle's step at the end (cpu_baseline): its loss on the same weights and batch is
    # compared with the device's first-step loss -- an oracle-anchored check inside the benchmark line itself
    w0 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.plain:
        w0 = {f"{n}/{i}": np.array(a, dtype=np.float32, copy=True) for n in model.layer_names()
              for i, a in enumerate(model.get_layer(n).get_weights())}
    for i in range(warm_run):
        b0, _ = model.train_step_device(x, ys)
        if i == 0 and (args.plain or w0 is not None):     # (the strict-fp32 block compares the two arithmetics on the very first step)
            loss_first = float(sum(b[0].item() for b in b0))
    barrier()
    captured = getattr(model, "_step_graphs", None) is not None
    launch_mode = ("eager: every launch enqueued from Python" if not captured else
                   "launch tape: the recorded C-ABI calls of one step re-issued on the same two HIP streams, none of the Python "
                   "(tf2_yolo_amd/tape.py)" if type(model._step_graphs).__name__ == "StepTape" else
                   "hipGraph replay: forward + losses + backward + Adam captured once, cut at the gradient buckets in "
                   "data-parallel jobs (tf2_yolo_amd/capture.py)")
    log(f"warmup done ({warm_run} steps; step launch mode: {launch_mode.split(':')[0]})")
    # ---- THE timed region: exactly K steps between barriers ----
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bufs, _ = model.train_step_device(x, ys)
    barrier()
    dt = time.perf_counter() - t0
    log(f"timed region done: {dt / args.steps * 1e3:.2f} ms/step")
    loss_val = float(sum(b[0].item() for b in bufs))
    if args.plain:
        if rank == 0:
            print(json.dumps({"ms_per_step": round(dt / args.steps * 1e3, 3), "images_per_s": round(world * args.batch * args.steps / dt, 2),
                              "loss": round(loss_val, 4), "loss_first_step": loss_first, "conv_mode": ops.CONV_MODE,
                              "planes": bool(ops.USE_PLANES),
                              "steps": args.steps, "warmup_steps_run": warm_run}), flush=True)
        if world > 1 or force_dp:
            dist.destroy_process_group()
        return
    # ---- the same K steps once more with every launch enqueued from Python and bracketed by HIP events on its stream
    # (ops.KernelTimer): per-launch events are not part of a replayed step, so the roofline block is measured
    # here, over K steps timed exactly like the region above (every rank runs it: the collectives must match) ----
    timer = None if args.no_kernel_timer else ops.KernelTimer()
    dt_eager = None
    if timer is not None:
        ops.TIMER = timer
        t1 = time.perf_counter()
        for _ in range(args.steps):
            model.train_step_device(x, ys)
        barrier()
        dt_eager = time.perf_counter() - t1
        ops.TIMER = None
        log(f"eager + per-launch events region done: {dt_eager / args.steps * 1e3:.2f} ms/step")

    # Outside the timed region (every rank: the collectives must match): three more steps with the gradient reducer's trace on --
    # per bucket, when it was ready for its all-reduce and when the all-reduce had finished, measured from the start of backward
    dp_trace = None
    red = getattr(model, "_reducer", None)
    if red is not None and red.active and os.environ.get("YOLO_STEP_MODE", "tape") != "graph":
        red.start_trace(True)
        for _ in range(3):
            model.train_step_device(x, ys)
        barrier()
        dp_trace = {"what": "last of 3 extra steps: per gradient bucket (MB, ms from the start of backward until every gradient of the "
                            "bucket was written = its all-reduce can start, ms until the all-reduce had finished on the communication stream)",
                    "buckets": red.trace_ms(), "backend": backend, "world": world}
        red.start_trace(False)

    # Outside the timed region, right behind it (the chip is as warm as it was inside): what a bare fp16 MFMA loop on
    # random register operands sustains on THIS box -- the ceiling any fp16 MFMA kernel has at the clock the chip holds
    ceiling = None
    if rank == 0 and world == 1:
        try:
            ceiling = ops.mfma_ceiling(0.15)
        except Exception as e:   # informational
            ceiling = {"error": repr(e)}

    # Outside the timed region (rank 0, informational): the same step with the filter-gradient stream
    # switched off, so that every conv kernel has the chip to itself. In the timed region the filter gradient
    # runs BESIDE the data gradient / BatchNorm backward; that shortens the step but lengthens each of the
    # overlapped launches, so per-launch rates measured there understate the kernels.
    iso = None
    if timer is not None and rank == 0 and world == 1:
        model.net._overlap_wgrad = False
        iso_timer = ops.KernelTimer()
        ops.TIMER = iso_timer
        for _ in range(2):
            model.train_step_device(x, ys)
        torch.cuda.synchronize()
        ops.TIMER = None
        model.net._overlap_wgrad = True
        iso = iso_timer.summary()

    # Outside the timed region (rank 0, informational): BASELINE.json's second figure, the MFMA rate of the
    # Darknet-53 training-mode FORWARD alone (convs + batch-stat BN + activations + heads, no loss / backward).
    # conv rate = algorithmic conv FLOPs / summed conv-launch time (HIP events); whole = same FLOPs / wall time.
    fwd = None
    if timer is not None and rank == 0 and world == 1:
        nf = 5
        model.net.forward(x, training=True)
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        for _ in range(nf):
            model.net.forward(x, training=True)
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - tf0) / nf * 1e3
        f_timer = ops.KernelTimer()
        ops.TIMER = f_timer
        for _ in range(2):
            model.net.forward(x, training=True)
        torch.cuda.synchronize()
        ops.TIMER = None
        fa = f_timer.summary()
        fl = sum(v["flops"] for v in fa.values()) / 2
        kms = sum(v["ms"] for v in fa.values()) / 2
        pk = sum(v["flops"] / 2 / kernel_peak(k) for k, v in fa.items())   # time at peak, summed per kernel
        fwd = {"what": "training-mode forward only (batch-stat BN), bs %d, after the timed region" % args.batch,
               "ms": round(fwd_ms, 3), "conv_gflop": round(fl / 1e9, 1), "conv_kernel_ms": round(kms, 3),
               "conv_tflops": round(fl / kms / 1e9, 1), "conv_frac_of_peak": round(pk / 1e9 / kms, 4),
               "whole_forward_tflops": round(fl / fwd_ms / 1e9, 1),
               "whole_forward_frac_of_peak": round(pk / 1e9 / fwd_ms, 4),
               "raw_fp16_mfma_tflops_conv": round(PLANES_PASSES * sum(v["flops"] for k, v in fa.items() if "planes" in k)
                                                  / 2 / kms / 1e9, 1)}

        try:
            fwd["darknet53"] = darknet53_block(model, x, ceiling)
        except Exception as e:   # informational
            fwd["darknet53"] = {"error": repr(e)}

    t = torch.tensor([dt, dt_eager if dt_eager is not None else 0.0], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t[0].item())
    dt_eager = float(t[1].item()) if dt_eager is not None else None

    dp_in_sync = None
    if world > 1 or force_dp:   # outside the timed region: every replica must hold bit-identical weights after K steps
        # element-wise: min over ranks == max over ranks for EVERY parameter (not one checksum)
        lo, hi = model.net.params.data.clone(), model.net.params.data.clone()
        if backend != "nccl":
            lo, hi = lo.cpu(), hi.cpu()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dp_in_sync = bool(torch.equal(lo, hi))

    if rank == 0:
        roof = None
        if timer is not None:
            agg = timer.summary()
            # the dominant kernel = the family with the most device time in the timed steps; families within 10 % of the top
            # are ranked by their algorithmic FLOPs (the window forward / data gradient and the x-window filter gradient sit
            # within 5 % of each other since round 6 launches the latter -- a leaf on the second stream -- on one workgroup per
            # CU: box-to-box noise must not decide which of the two the line describes)
            top_ms = max(v["ms"] for v in agg.values())
            name, a = max(((k, v) for k, v in agg.items() if v["ms"] >= 0.9 * top_ms), key=lambda kv: kv[1]["flops"])
            achieved = a["flops"] / (a["ms"] * 1e-3) / 1e12
            peak = kernel_peak(name)
            roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": round(peak, 1),
                    "dominance_rule": "most device time over the timed steps; families within 10 % of the top ranked by algorithmic FLOPs",
                    "most_device_time_kernel": max(agg.items(), key=lambda kv: kv[1]["ms"])[0],
                    "peak_basis": ("fp16 MFMA dense peak 2500 TFLOP/s / 3 passes per product (two scaled fp16 planes)"
                                   if "planes" in name else
                                   "bf16 MFMA dense peak 2500 TFLOP/s / 6 passes per fp32 product (exact 3-way split)"
                                   if "split" in name else "fp32-input MFMA peak (v_mfma_f32_32x32x2_f32)"),
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                    "held_clock_ceiling": None if not ceiling or "error" in ceiling else dict(
                        ceiling, algorithmic_tflops=round(ceiling["raw_fp16_mfma_tflops"] / PLANES_PASSES, 1),
                        what="bare v_mfma_f32_32x32x16_f16 loops on random fp16 register operands, 2 x 8 waves per CU, no "
                             "memory traffic, ONE launch right behind the timed region (csrc/probe.hip); algorithmic = raw / 3 "
                             "passes per product"),
                    "frac_of_held_clock_ceiling": (None if not ceiling or "error" in ceiling or "planes" not in name else
                                                   round(achieved / (ceiling["raw_fp16_mfma_tflops"] / PLANES_PASSES), 4)),
                    "traffic": hbm_traffic_from_profile(name),
                    "pmc": pmc_from_profile(name),
                    "ratio_to_fp32_input_mfma_peak_NOT_a_roofline_fraction": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                    "traffic_source": "newest profiles/r*_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                                      "passes, gfx950 x2 FETCH correction), bytes per launch",
                    "launches": a["launches"], "avg_launch_us": round(a["ms"] * 1e3 / a["launches"], 2),
                    "isolated": (None if iso is None or name not in iso else
                                 {"note": "same kernel, same step, filter-gradient stream off (no concurrent kernels); "
                                          "2 extra steps outside the timed region",
                                  "achieved": round(iso[name]["flops"] / (iso[name]["ms"] * 1e-3) / 1e12, 2),
                                  "frac": round(iso[name]["flops"] / (iso[name]["ms"] * 1e-3) / 1e12 / peak, 4),
                                  "frac_of_held_clock_ceiling": (
                                      None if not ceiling or "error" in ceiling or "planes" not in name else
                                      round(iso[name]["flops"] / (iso[name]["ms"] * 1e-3) / 1e12
                                            / (ceiling["raw_fp16_mfma_tflops"] / PLANES_PASSES), 4)),
                                  "avg_launch_us": round(iso[name]["ms"] * 1e3 / iso[name]["launches"], 2)}),
                    "flops_per_launch": a["flops"] / a["launches"],
                    "all_conv_kernels": {k: {"launches": v["launches"], "ms_per_step": round(v["ms"] / args.steps, 3),
                                             "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                             "frac_of_peak": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / kernel_peak(k), 4)}
                                         for k, v in sorted(agg.items())}}
        out = {"metric": "images/sec training YOLOv3 416x416 bs=32/GPU", "value": round(world * args.batch * args.steps / dt, 2),
               "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32-storage/fp16x2-mfma", "data": "synthetic",
               "arithmetic": "fp32 storage and fp32 accumulation everywhere; conv operands enter the matrix cores as two scaled fp16 planes (22-23 significant bits, fp32 has 24), one product = 3 fp16 MFMA passes exact in the fp32 accumulator: measured 1.5e-6 / 2.4e-6 / 6e-7 (fwd / dgrad / wgrad) against the float64 oracle at the 13x13x512->1024 3x3 layer, bs 32, K=4608 (tests/test_gpu_conv.py), parity bar 1e-4; the first conv (Cin=3) and Cout<=32 layers use the exact bf16x6 / fp32-input MFMA kernels (stride-2 filter gradients run on the planes kernels too); YOLO_CONV_PLANES=0 selects the exact bf16x6 kernels everywhere (474 img/s), YOLO_CONV_MODE=fp32 the fp32-input MFMA kernels",
               "config": {"workload": "YOLOv3 Darknet-53 416x416, 9 anchors / 3 FPN scales, C=80: training step = "
                                      "forward (batch-stat BN) + 3 fused loss/grad kernels + backward + "
                                      "gradient all-reduce + Adam",
                          "global_batch": world * args.batch, "per_gpu_batch": args.batch,
                          "parallelism": f"dp{world}", "loss": round(loss_val, 4), "replicas_in_sync": dp_in_sync,
                          "step_launch_mode": launch_mode,
                          "warmup_steps_run": warm_run},
               "eager_region": (None if dt_eager is None else
                                {"what": "the same K steps again, every launch enqueued from Python and bracketed by HIP events on "
                                         "its stream (the roofline block's measurements come from here: a replayed step "
                                         "carries no per-launch events)",
                                 "ms_per_step": round(dt_eager / args.steps * 1e3, 3),
                                 "images_per_s": round(world * args.batch * args.steps / dt_eager, 2)}),
               "roofline": roof, "forward": fwd}
        if dp_trace is not None:
            out["dp_trace"] = dp_trace
        if fwd is not None and ceiling and "error" not in ceiling:
            fwd["conv_frac_of_held_clock_ceiling"] = round(fwd["raw_fp16_mfma_tflops_conv"] / ceiling["raw_fp16_mfma_tflops"], 4)
        if world == 1:
            try:
                out["decode_nms"] = decode_nms_block()
            except Exception as e:   # informational block: never lose the headline line over it
                out["decode_nms"] = {"error": repr(e)}
        if world == 1 and not args.no_extra_blocks:
            log("strict fp32 step (child process) ...")
            try:
                out["strict_fp32"] = strict_fp32_block()
            except Exception as e:   # informational blocks: never lose the headline line over them
                out["strict_fp32"] = {"error": repr(e)}
            log("other BASELINE.json configs ...")
            try:
                out["configs"] = configs_block()
            except Exception as e:
                out["configs"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (bounded sample) ...")
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            cb = cpu_baseline(min(cores, 16), weights=w0, batch=(x_h, ys_h) if w0 is not None else None)   # the GPU box's CPU share is 16 cores
            if w0 is not None and loss_first is not None:
                rel = abs(cb["loss"] - loss_first) / max(abs(cb["loss"]), 1.0)
                cb["first_step_loss_check"] = {
                    "what": "loss of the device's FIRST training step against the CPU oracle's fp32 step on the same initial "
                            "weights and the same batch (training-mode BatchNorm, the three losses summed)",
                    "device": loss_first, "cpu_oracle_fp32": cb["loss"], "rel_diff": rel, "agree_to_1e-4": bool(rel < 1e-4)}
            out["cpu_baseline"] = cb
        # LAST key: the figures of the informational blocks once more, flat and short (a record that keeps only the tail
        # of this line still holds them)
        try:
            sm = {"images_per_s": out["value"], "ms_per_step": out["ms_per_step"]}
            if roof is not None:
                sm["dominant_kernel_frac_in_step"] = roof["frac"]
                sm["dominant_kernel_frac_isolated"] = (roof.get("isolated") or {}).get("frac")
                roof["dominant_kernel_frac_isolated"] = sm["dominant_kernel_frac_isolated"]
            if fwd is not None:
                sm.update({"forward_ms": fwd["ms"], "forward_conv_frac_of_peak": fwd["conv_frac_of_peak"]})
                dk = fwd.get("darknet53") or {}
                sm.update({"darknet53_forward_ms": dk.get("ms"), "darknet53_conv_frac_of_peak": dk.get("conv_frac_of_peak"),
                           "darknet53_whole_frac_of_peak": dk.get("whole_frac_of_peak")})
                if roof is not None:   # (scalars inside `roofline` survive a parser that drops nested blocks)
                    roof.update({"forward_conv_frac_of_peak": fwd["conv_frac_of_peak"], "darknet53_forward_ms": dk.get("ms"),
                                 "darknet53_conv_frac_of_peak": dk.get("conv_frac_of_peak")})
            sf = out.get("strict_fp32") or {}
            sm.update({"strict_fp32_images_per_s": sf.get("images_per_s"), "strict_fp32_ms_per_step": sf.get("ms_per_step")})
            cf = out.get("configs") or {}
            for key, short in (("c1_yolov1_5_224_bs4_train", "c1"), ("c2_yolov2_416_bs16_train", "c2"),
                               ("c4_yolov4_608_bs16_train", "c4")):
                sm[short + "_ms_per_step"] = (cf.get(key) or {}).get("ms_per_step")
            c5 = cf.get("c5_yolov3_416_bs1_predict_decode_nms") or {}
            sm["c5"] = {k: c5.get(k) for k in ("predict_ms_hipgraph", "candidates", "gpu_decode_ms", "gpu_nms_ms", "gpu_diou_nms_ms",
                                              "gpu_soft_nms_ms")}
            ref5 = c5.get("cpu_reference") or {}
            sm["c5_cpu_reference_ms"] = {k: ref5.get(k) for k in ("candidates", "decode_ms", "nms_ms", "diou_nms_ms", "soft_nms_ms")}
            chk = (out.get("cpu_baseline") or {}).get("first_step_loss_check")
            if chk:
                sm["first_step_loss_rel_diff_vs_cpu_oracle"] = chk["rel_diff"]
            out["summary"] = sm
        except Exception as e:
            out["summary"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1 or force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
