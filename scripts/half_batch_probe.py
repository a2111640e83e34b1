"""VERDICT r04 next #2, the forward half: would a 2-way half-batch pipeline shorten the training forward? In the forward
pass conv(L+1) needs BatchNorm-apply(L), which needs the statistics of ALL of conv(L)'s output: a strict chain. Split into
halves A and B of the batch the only overlap the dependencies allow is conv_A(L+1) beside apply_B(L) (the statistics of
L+1 need conv_B(L+1) too, so apply_A(L+1) cannot start before it). This probe runs a chain of residual-block units
(1x1 C -> C/2, 3x3 C/2 -> C, bs 32) at one resolution in both schedules on the real kernels:
  full:     one stream:  conv(32) -> apply(32) -> conv(32) -> ...
  pipeline: two streams: s1: conv_A(L) -> [conv_B(L) done] -> apply_A(L) -> conv_A(L+1) ...
                         s2: [conv_A(L-1) done] apply_B(L-1) -> conv_B(L) -> ...
(finalize kernels left out of both: identical in both; BN coefficients fixed.) usage: half_batch_probe.py out.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from tf2_yolo_amd import ops
ops.ensure_conv_workspace()
g = torch.Generator(device="cuda").manual_seed(2)
UNITS, REP = 16, 6


def make(hw, C, n):
    """buffers of a chain of UNITS conv units at batch n: 1x1 C -> C/2 and 3x3 C/2 -> C alternating"""
    P = n * hw * hw
    units = []
    for u in range(UNITS):
        cin, cout, k = (C, C // 2, 1) if u % 2 == 0 else (C // 2, C, 3)
        d = ops.conv_desc((n, hw, hw, cin), cout, k, k, 1, "same")
        w = torch.randn(cout, k * k * cin, device="cuda", generator=g) * (1.0 / (k * k * cin) ** 0.5)
        units.append(dict(d=d, cin=cin, cout=cout, wp=ops.split_planes(w, cout, k * k * cin),
                          y=torch.empty(P, cout, device="cuda"),
                          pl=torch.zeros(ops.planes_bytes(P, cout), device="cuda", dtype=torch.uint8),
                          stats=torch.zeros(ops.BN_STAT_SLOTS * 2 * cout, device="cuda", dtype=torch.float64),
                          scale=torch.full((cout,), 0.7, device="cuda"), shift=torch.zeros(cout, device="cuda"),
                          bound=torch.full((1,), 8.0, device="cuda").view(torch.int32), outb=torch.zeros(1, device="cuda")))
    x0 = ops.split_planes(torch.randn(P, C, device="cuda", generator=g), P, C)
    return x0, units


def conv(u, xp):
    ops.conv2d_fwd_planes(u["d"], xp, u["wp"], None, out=u["y"].view(u["d"].N, u["d"].Ho, u["d"].Wo, u["cout"]), stats=u["stats"])


def apply(u):
    ops.bn_act_fwd(u["y"], u["cout"], u["scale"], u["shift"], 1, None, out=None, planes=u["pl"], want_out=False,
                   bn_bound=u["bound"], out_bound=u["outb"])


def run_full(x0, units):
    xp = x0
    for u in units:
        conv(u, xp)
        apply(u)
        xp = u["pl"]


def run_pipeline(xa, ua, xb, ub, s1, s2):
    ev_a = [torch.cuda.Event() for _ in ua]     # conv_A(L) done
    ev_b = [torch.cuda.Event() for _ in ub]     # conv_B(L) done
    pa, pb = xa, xb
    for L in range(len(ua)):
        with torch.cuda.stream(s1):
            conv(ua[L], pa)
            ev_a[L].record(s1)
        with torch.cuda.stream(s2):
            conv(ub[L], pb)
            ev_b[L].record(s2)
            s2.wait_event(ev_a[L])      # statistics of layer L complete: both halves' convs are done
            apply(ub[L])
        with torch.cuda.stream(s1):
            s1.wait_event(ev_b[L])
            apply(ua[L])
        pa, pb = ua[L]["pl"], ub[L]["pl"]


def timed(fn, sync_streams=()):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REP):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REP * 1e3


res = {"what": __doc__.split("usage")[0].strip(), "units_per_chain": UNITS, "cases": []}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for hw, C in ((52, 256), (26, 512)):
    x0, units = make(hw, C, 32)
    xa, ua = make(hw, C, 16)
    xb, ub = make(hw, C, 16)
    t_full = timed(lambda: run_full(x0, units))
    t_half_serial = timed(lambda: (run_full(xa, ua), run_full(xb, ub)))
    t_pipe = timed(lambda: run_pipeline(xa, ua, xb, ub, s1, s2))
    r = {"layer": f"{hw}x{hw}, {C} <-> {C // 2} channels, bs 32", "full_batch_one_stream_ms": round(t_full, 3),
         "two_half_batches_one_after_the_other_ms": round(t_half_serial, 3), "half_batch_pipeline_two_streams_ms": round(t_pipe, 3),
         "pipeline_over_full": round(t_pipe / t_full, 3)}
    res["cases"].append(r)
    print(json.dumps(r), flush=True)
    del x0, units, xa, ua, xb, ub
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
