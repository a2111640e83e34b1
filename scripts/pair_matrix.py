"""what does one kernel of the backward pass cost BESIDE another? For the filter gradient of a 3x3 layer
(the side stream's work) against each kind of launch the compute stream makes meanwhile -- window data gradient, BatchNorm
backward reduce / apply, 1x1 data gradient (accumulate), BatchNorm forward apply -- at the benchmark's layer sizes (bs 32):
time of N launches of each ALONE, of both CONCURRENTLY on two streams (started together, until both are done), and the
ratio concurrent / (sum of alone): 1.0 = nothing gained by overlapping, 0.5 = the shorter one is free. Also the same pairs
with the filter-gradient stream at low priority. usage: pair_matrix.py out.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from tf2_yolo_amd import ops
out_path = sys.argv[1] if len(sys.argv) > 1 else None
ops.ensure_conv_workspace(); ops.ensure_wgrad_workspace()
g = torch.Generator(device="cuda").manual_seed(1)
N = 32
REP = 12


def layer(hw, cin, cout):
    """operands of one residual block's 3x3 unit (cin -> cout at hw x hw) and of the 1x1 in front of the next block"""
    d3 = ops.conv_desc((N, hw, hw, cin), cout, 3, 3, 1, "same")
    d1 = ops.conv_desc((N, hw, hw, cout), cin, 1, 1, 1, "same")
    P = N * hw * hw
    x = torch.randn(P, cin, device="cuda", generator=g)
    y = torch.randn(P, cout, device="cuda", generator=g)
    dout = torch.randn(P, cout, device="cuda", generator=g) * 1e-2
    xp = ops.split_planes(x, P, cin)
    dyp = ops.split_planes(dout, P, cout)
    w3 = torch.randn(cout, 9 * cin, device="cuda", generator=g) * 0.03
    wT3 = ops.split_planes(ops.filter_transpose(w3.reshape(-1), cout, 9, cin), cin, 9 * cout)
    w1 = torch.randn(cin, cout, device="cuda", generator=g) * 0.05
    wT1 = ops.split_planes(ops.filter_transpose(w1.reshape(-1), cin, 1, cout), cout, cin)
    dy1p = ops.split_planes(torch.randn(P, cin, device="cuda", generator=g) * 1e-2, P, cin)
    dw = torch.zeros(cout * 9 * cin, device="cuda")
    dx3 = torch.empty((N, hw, hw, cin), device="cuda")
    dx1 = torch.randn((N, hw, hw, cout), device="cuda", generator=g)
    gamma = torch.rand(cout, device="cuda", generator=g) + 0.5
    scale, shift = gamma.clone(), torch.zeros(cout, device="cuda")
    smean, sinv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
    red = torch.zeros(513 * 2 * cout, device="cuda", dtype=torch.float64)
    aux = torch.zeros(68, device="cuda", dtype=torch.int32)
    dg, db = torch.zeros(cout, device="cuda"), torch.zeros(cout, device="cuda")
    pl = torch.zeros(ops.planes_bytes(P, cout), device="cuda", dtype=torch.uint8)
    bound = torch.ones(1, device="cuda", dtype=torch.float32).view(torch.int32)
    outb = torch.zeros(1, device="cuda")
    lib = ops._lib.load()
    from tf2_yolo_amd.ops import _p, _stream
    def bn_reduce():
        ops.check(lib.yolo_bn_act_bwd_reduce_bound(_p(y), _p(dout), P, cout, _p(scale), _p(shift), _p(smean), _p(sinv), 1, _p(red),
                                                   _p(aux), _stream()), "reduce")
    def bn_apply():
        ops.check(lib.yolo_bn_act_bwd_apply_planes(_p(y), _p(dout), P, cout, _p(gamma), _p(scale), _p(shift), _p(smean), _p(sinv), 1,
                                                   _p(red), _p(dg), _p(db), _p(None), _p(pl), _p(aux), _stream()), "apply")
    return {
        "wgrad 3x3 (x-window)": lambda: ops.conv2d_wgrad_planes(d3, xp, dyp, dw),
        "dgrad 3x3 (window)": lambda: ops.conv2d_dgrad_planes(d3, dyp, wT3, dx=dx3),
        "bn bwd reduce": bn_reduce,
        "bn bwd apply (planes out)": bn_apply,
        "dgrad 1x1 (accumulate)": lambda: ops.conv2d_dgrad_planes(d1, dy1p, wT1, dx=dx1, accumulate=True),
        "bn fwd apply (planes out)": lambda: ops.bn_act_fwd(y, cout, scale, shift, 1, None, out=None, planes=pl, want_out=False,
                                                            bn_bound=bound, out_bound=outb),
    }


def timed_alone(fn, stream):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(REP):
            fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REP * 1e6


def timed_pair(fa, sa, fb, sb, na, nb):
    """na launches of fa on sa and nb of fb on sb, enqueued interleaved, both streams released together"""
    torch.cuda.synchronize()
    gate = torch.cuda.Event()
    gate.record(torch.cuda.current_stream())
    sa.wait_event(gate); sb.wait_event(gate)
    t0 = time.perf_counter()
    ia = ib = 0
    while ia < na or ib < nb:
        if ia < na:
            with torch.cuda.stream(sa):
                fa()
            ia += 1
        if ib < nb:
            with torch.cuda.stream(sb):
                fb()
            ib += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


res = {"what": __doc__.split("usage")[0].strip(), "batch": N, "launches_per_measurement": REP, "layers": {}}
lo, hi = torch.cuda.Stream.priority_range()
main = torch.cuda.Stream()
side = torch.cuda.Stream()
side_low = torch.cuda.Stream(priority=lo)      # (priority_range(): (lowest, highest) as torch reports it)
main_high = torch.cuda.Stream(priority=hi)
res["stream_priority_range"] = [lo, hi]
for hw, cin, cout in ((52, 128, 256), (26, 256, 512)):
    fns = layer(hw, cin, cout)
    alone = {k: timed_alone(f, main) for k, f in fns.items()}
    rows = {}
    wname = "wgrad 3x3 (x-window)"
    for k, f in fns.items():
        if k == wname:
            continue
        # equal total durations: n_w launches of the filter gradient beside n_k launches of the other kernel
        n_w = REP
        n_k = max(1, round(REP * alone[wname] / alone[k]))
        for tag, sm, ss in (("two streams", main, side), ("filter-gradient stream low priority", main, side_low),
                            ("compute stream high priority", main_high, side)):
            timed_pair(f, sm, fns[wname], ss, 2, 2)
            both = min(timed_pair(f, sm, fns[wname], ss, n_k, n_w) for _ in range(3))
            serial = n_k * alone[k] + n_w * alone[wname]
            rows.setdefault(k, {"alone_us": round(alone[k], 1), "launches_beside_%d_filter_gradients" % n_w: n_k})[tag] = {
                "concurrent_us": round(both, 1), "sum_of_alone_us": round(serial, 1), "ratio": round(both / serial, 3)}
    res["layers"][f"{hw}x{hw} {cin}->{cout} (bs {N})"] = {"filter_gradient_alone_us": round(alone[wname], 1), "beside": rows}
    print(json.dumps(res["layers"][f"{hw}x{hw} {cin}->{cout} (bs {N})"], indent=1), flush=True)
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
