"""GPU parity of BN(+act, +residual), glue ops, head activations and Adam against the torch-CPU
float64 oracle (oracle/layers.py). fp32 tolerance 1e-4 relative to the tensor scale."""
import numpy as np
import pytest
import torch

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import layers as L  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rel(a, b):
    return (a.double().cpu() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


@pytest.mark.parametrize("C,act,use_res", [(32, 1, False), (64, 1, True), (96, 2, False), (128, 2, True),
                                           (1024, 1, False), (2048, 1, False)])
def test_bn_act_train_fwd_bwd(C, act, use_res):
    from tf2_yolo_amd import ops
    g = torch.Generator().manual_seed(C + act)
    N, H, W = 3, 7, 5
    x = (torch.randn(N, H, W, C, generator=g, dtype=torch.float64) * 2 + 0.7).requires_grad_(True)
    gamma = (torch.rand(C, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    beta = torch.randn(C, generator=g, dtype=torch.float64).requires_grad_(True)
    res = torch.randn(N, H, W, C, generator=g, dtype=torch.float64) if use_res else None
    z, mean, var = L.batchnorm_train(x, gamma, beta)
    a = L.leaky(z) if act == 1 else L.mish(z)
    out = a + res if use_res else a
    dout = torch.randn(out.shape, generator=g, dtype=torch.float64)
    out.backward(dout)

    dev = "cuda"
    xd = x.detach().float().to(dev)
    P = N * H * W
    stats = torch.zeros(64 * 2 * C, device=dev, dtype=torch.float64)
    red = torch.zeros(513 * 2 * C, device=dev, dtype=torch.float64)
    f = lambda: torch.empty(C, device=dev)
    scale, shift, smean, sinv = f(), f(), f(), f()
    mm = torch.zeros(C, device=dev)
    mv = torch.ones(C, device=dev)
    gd, bd = gamma.detach().float().to(dev), beta.detach().float().to(dev)
    ops.bn_stats(xd, C, stats)
    ops.bn_finalize(stats, P, C, gd, bd, mm, mv, scale, shift, smean, sinv)
    o = ops.bn_act_fwd(xd, C, scale, shift, act, None if res is None else res.float().to(dev))
    assert _rel(o, out.detach()) < TOL
    assert _rel(smean, mean.detach()) < TOL
    assert _rel(mm, 0.01 * mean.detach()) < TOL
    assert _rel(mv, 0.99 + 0.01 * var.detach()) < TOL
    dg = torch.zeros(C, device=dev)
    db = torch.zeros(C, device=dev)
    dx = ops.bn_act_bwd(xd, dout.float().to(dev), C, gd, scale, shift, smean, sinv, act, red, dg, db)
    assert _rel(dx, x.grad) < TOL
    assert _rel(dg, gamma.grad) < TOL
    assert _rel(db, beta.grad) < TOL
    # unbiased moving-variance switch
    mm2, mv2 = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    stats.zero_()
    ops.bn_stats(xd, C, stats)
    ops.bn_finalize(stats, P, C, gd, bd, mm2, mv2, scale, shift, smean, sinv, unbiased=True)
    assert _rel(mv2, 0.99 + 0.01 * var.detach() * P / (P - 1)) < TOL


@pytest.mark.parametrize("C,act", [(32, 1), (64, 2), (24, 1)])
def test_bn_act_bwd_reads_a_channel_slice_in_place(C, act):
    """yolo_bn_act_bwd_reduce_bound_ld / _apply_planes_ld: dout given as channels [off, off + C) of a wider tensor (the
    gradient of a Concatenate, ops.ChannelSlice) -- the same arithmetic on the same values in the same order as on the dense
    copy, so dx, its planes, dgamma, dbeta and the bound words must be BIT-identical to the dense call; pixel counts that are
    not multiples of 16, slices at the start / middle / end of the wide tensor."""
    from tf2_yolo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(40 + C)
    N, H, W = 2, 9, 7
    P = N * H * W
    Cw = C + 48 + 16
    x = torch.randn(P, C, device="cuda", generator=g) * 2 + 0.3
    wide = torch.randn(P, Cw, device="cuda", generator=g)
    gamma = torch.rand(C, device="cuda", generator=g) + 0.5
    scale = torch.rand(C, device="cuda", generator=g) + 0.5
    shift = torch.randn(C, device="cuda", generator=g) * 0.1
    smean = torch.randn(C, device="cuda", generator=g) * 0.1
    sinv = torch.rand(C, device="cuda", generator=g) + 0.5
    for off in (0, 48, Cw - C):
        dense = wide[:, off:off + C].contiguous()
        res = []
        for dout in (dense, ops.ChannelSlice(wide, Cw, off, C)):
            red = torch.zeros(513 * 2 * C, device="cuda", dtype=torch.float64)
            dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
            aux = torch.zeros(68, device="cuda", dtype=torch.int32)
            pl = torch.zeros(ops.planes_bytes(P, C), device="cuda", dtype=torch.uint8) if C % 16 == 0 else None
            dx = ops.bn_act_bwd(x, dout, C, gamma, scale, shift, smean, sinv, act, red, dg, db, planes=pl,
                                bound_aux=aux if pl is not None else None)
            torch.cuda.synchronize()
            res.append((dx, dg, db, pl, aux))
        for a, b in zip(*res):
            assert (a is None and b is None) or torch.equal(a, b)
    with pytest.raises(Exception):
        ops.bn_act_bwd(x, ops.ChannelSlice(wide, Cw, 2, C), C, gamma, scale, shift, smean, sinv, act,
                       torch.zeros(513 * 2 * C, device="cuda", dtype=torch.float64), torch.zeros(C, device="cuda"),
                       torch.zeros(C, device="cuda"))


def test_bn_inference_fold():
    from tf2_yolo_amd import ops
    C = 64
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 5, C, generator=g, dtype=torch.float64)
    gamma, beta = torch.rand(C, generator=g, dtype=torch.float64) + .5, torch.randn(C, generator=g, dtype=torch.float64)
    mm, mv = torch.randn(C, generator=g, dtype=torch.float64), torch.rand(C, generator=g, dtype=torch.float64) + .1
    ref = L.leaky(L.batchnorm_infer(x, gamma, beta, mm, mv))
    scale, shift = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_fold_inference(C, gamma.float().cuda(), beta.float().cuda(), mm.float().cuda(), mv.float().cuda(), scale, shift)
    o = ops.bn_act_fwd(x.float().cuda(), C, scale, shift, 1)
    assert _rel(o, ref) < TOL


def test_upsample_concat_s2d_maxpool():
    from tf2_yolo_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 6, 4, 8, generator=g, dtype=torch.float64).requires_grad_(True)
    y = torch.randn(2, 12, 8, 12, generator=g, dtype=torch.float64).requires_grad_(True)
    up = L.upsample2x(x)
    cat = torch.cat([up, y], dim=-1)
    dcat = torch.randn(cat.shape, generator=g, dtype=torch.float64)
    cat.backward(dcat)
    xd, yd = x.detach().float().cuda(), y.detach().float().cuda()
    upd = torch.empty(2, 12, 8, 8, device="cuda")
    ops.upsample2x_fwd(xd, upd, 8, 0)
    catd = torch.empty(2, 12, 8, 20, device="cuda")
    ops.copy_channels_in(upd, 8, catd, 20, 0)
    ops.copy_channels_in(yd, 12, catd, 20, 8)
    assert torch.equal(catd.double().cpu(), cat.detach().float().double())
    dcd = dcat.float().cuda()
    dup = torch.empty(2, 12, 8, 8, device="cuda")
    ops.copy_channels_out(dcd, 20, 0, dup, 8)
    dy = torch.empty(2, 12, 8, 12, device="cuda")
    ops.copy_channels_out(dcd, 20, 8, dy, 12)
    dx = torch.empty(2, 6, 4, 8, device="cuda")
    ops.upsample2x_bwd(dup, 8, 0, dx)
    assert _rel(dx, x.grad) < 1e-6 and _rel(dy, y.grad) < 1e-6
    ops.upsample2x_bwd(dup, 8, 0, dx, accumulate=True)
    assert _rel(dx, 2 * x.grad) < 1e-6
    # space_to_depth
    s = torch.randn(2, 6, 4, 5, generator=g, dtype=torch.float64).requires_grad_(True)
    sd = L.space_to_depth2(s)
    ds = torch.randn(sd.shape, generator=g, dtype=torch.float64)
    sd.backward(ds)
    o = torch.empty(2, 3, 2, 20, device="cuda")
    ops.space_to_depth2_fwd(s.detach().float().cuda(), o, 20, 0)
    assert torch.equal(o.double().cpu(), sd.detach().float().double())
    dsx = torch.empty(2, 6, 4, 5, device="cuda")
    ops.space_to_depth2_bwd(ds.float().cuda(), 20, 0, dsx)
    assert _rel(dsx, s.grad) < 1e-6
    # max-pool: 2x2 s2 valid (odd size floors), SPP 5/9/13 s1 same, 2x2 s1 same
    for (k, st, pad, hh, ww) in [(2, 2, "valid", 7, 6), (5, 1, "same", 7, 6), (13, 1, "same", 7, 6),
                                 (2, 1, "same", 5, 5), (2, 2, "same", 7, 7)]:
        m = torch.randn(2, hh, ww, 6, generator=g, dtype=torch.float64).requires_grad_(True)
        ref = L.maxpool(m, k, st, pad)
        dref = torch.randn(ref.shape, generator=g, dtype=torch.float64)
        ref.backward(dref)
        Ho, Wo = ref.shape[1], ref.shape[2]
        pt = L.same_pad(hh, k, st)[1] if pad == "same" else 0
        pl = L.same_pad(ww, k, st)[1] if pad == "same" else 0
        md = m.detach().float().cuda()
        out = torch.empty(2, Ho, Wo, 6, device="cuda")
        arg = torch.empty(2, Ho, Wo, 6, device="cuda", dtype=torch.int32)
        ops.maxpool_fwd(md, k, st, pt, pl, Ho, Wo, out, 6, 0, arg)
        assert torch.equal(out.double().cpu(), ref.detach().float().double())
        dm = torch.zeros(2, hh, ww, 6, device="cuda")
        ops.maxpool_bwd(dref.float().cuda(), 2, Ho, Wo, 6, 6, 0, arg, dm)
        assert _rel(dm, m.grad) < 1e-5


@pytest.mark.parametrize("k,hh,ww,c", [(5, 19, 19, 16), (9, 19, 19, 16), (13, 19, 19, 24), (13, 13, 13, 8), (5, 7, 30, 8),
                                       (9, 32, 32, 8)])
def test_spp_pools_plane_kernels(k, hh, ww, c):
    """stride-1 'same' pools with C % 8 == 0 and H W <= 1024 run on the plane kernels (csrc/elementwise.hip): values and
    winners as the fp64 oracle's (first maximum in row-major window order; the data has ties), the gather backward equals the
    oracle's gradient and is the same bit for bit from run to run"""
    from tf2_yolo_amd import ops
    g = torch.Generator().manual_seed(100 * k + hh)
    m = torch.randint(-6, 7, (3, hh, ww, c), generator=g).double()          # small integers: many ties inside a window
    m = (m + 0.25 * torch.randn(3, hh, ww, c, generator=g, dtype=torch.float64).round()).requires_grad_(True)
    ref = L.maxpool(m, k, 1, "same")
    dref = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dref)
    pt, pl = L.same_pad(hh, k, 1)[1], L.same_pad(ww, k, 1)[1]
    md = m.detach().float().cuda()
    out = torch.empty(3, hh, ww, c + 8, device="cuda")                       # written into a channel slice
    arg = torch.empty(3, hh, ww, c, device="cuda", dtype=torch.int32)
    ops.maxpool_fwd(md, k, 1, pt, pl, hh, ww, out, c + 8, 4, arg)
    assert torch.equal(out[..., 4:4 + c].double().cpu(), ref.detach().float().double())
    # winners: the value at the saved offset is the output, and it is the FIRST such value of its window
    flat = md.reshape(-1)
    assert torch.equal(flat[arg.reshape(-1).long()].reshape(3, hh, ww, c), out[..., 4:4 + c])
    mc, argc = m.detach().float(), arg.cpu()
    for _ in range(300):
        b, ho, wo, ch = (int(torch.randint(0, n_, (1,), generator=g)) for n_ in (3, hh, ww, c))
        first = None
        for r in range(k):
            for q in range(k):
                h, w = ho + r - pt, wo + q - pl
                if 0 <= h < hh and 0 <= w < ww and (first is None or mc[b, h, w, ch] > mc[first]):
                    first = (b, h, w, ch)
        assert int(argc[b, ho, wo, ch]) == ((first[0] * hh + first[1]) * ww + first[2]) * c + first[3]
    dyd = torch.zeros(3, hh, ww, c + 8, device="cuda")
    dyd[..., 4:4 + c] = dref.float().cuda()
    runs = []
    for _ in range(2):
        dm = torch.ones(3, hh, ww, c, device="cuda")                         # dx += ...
        ops.maxpool_bwd_same(dyd, 3, hh, ww, c, c + 8, 4, arg, k, pt, pl, dm)
        runs.append(dm.clone())
    assert torch.equal(runs[0], runs[1])
    # the scatter form on the same winners is the reference for WHICH input receives each gradient
    dscat = torch.ones(3, hh, ww, c, device="cuda")
    ops.maxpool_bwd(dyd[..., 4:4 + c].contiguous(), 3, hh, ww, c, c, 0, arg, dscat)
    assert _rel(runs[0], dscat.double().cpu()) < 1e-5


@pytest.mark.parametrize("version,A,C", [(3, 3, 80), (4, 3, 7), (2, 5, 20), (1, 2, 1), (1, 2, 4)])
def test_head_act(version, A, C):
    from tf2_yolo_amd import ops
    g = torch.Generator().manual_seed(version * 10 + C)
    P = 2 * 5 * 5
    D = 5 * A + C if version == 1 else A * (5 + C)
    t = torch.randn(P, D, generator=g, dtype=torch.float64).requires_grad_(True)
    anchors = torch.rand(A, 2, generator=g, dtype=torch.float64) + 0.1
    if version == 1:
        y = torch.cat([torch.sigmoid(t[:, :5 * A]), torch.softmax(t[:, 5 * A:], dim=-1)], dim=-1)
    else:
        tt = t.reshape(P, A, 5 + C)
        cls = torch.softmax(tt[..., 5:], dim=-1) if version == 2 else torch.sigmoid(tt[..., 5:])
        y = torch.cat([torch.sigmoid(tt[..., 0:2]), torch.exp(tt[..., 2:4]) * anchors, torch.sigmoid(tt[..., 4:5]),
                       cls], dim=-1).reshape(P, D)
    dy = torch.randn(P, D, generator=g, dtype=torch.float64)
    y.backward(dy)
    anc = anchors.float().reshape(-1).cuda() if version != 1 else None
    td = t.detach().float().cuda()
    yd = ops.head_act_fwd(td, A, C, version, anc)
    assert _rel(yd, y.detach()) < TOL
    dt = ops.head_act_bwd(yd, dy.float().cuda(), A, C, version, anc)
    assert _rel(dt, t.grad) < TOL
    if version == 4:
        da = torch.zeros(2 * A, device="cuda")
        ops.head_act_bwd(yd, dy.float().cuda(), A, C, version, anc, danchors=da)
        tt = t.detach().reshape(P, A, 5 + C)
        ref = (dy.reshape(P, A, 5 + C)[..., 2:4] * torch.exp(tt[..., 2:4])).sum(0).reshape(-1)
        assert _rel(da, ref) < TOL


def test_adam_matches_keras_formula():
    from tf2_yolo_amd import ops
    n = 1000 * 4 + 3
    rng = np.random.default_rng(0)
    p, gsum = rng.standard_normal(n), None
    pd = torch.tensor(p, dtype=torch.float32, device="cuda")
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    pm, mm, vv = p.astype(np.float64).copy(), np.zeros(n), np.zeros(n)
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-7
    for step in range(1, 4):
        g = rng.standard_normal(n)
        gd = torch.tensor(g, dtype=torch.float32, device="cuda")
        ops.adam_step(pd, gd, m, v, lr, step, grad_scale=0.5)
        assert float(gd.abs().max()) == 0.0       # zero_grad
        gg = g * 0.5
        mm = b1 * mm + (1 - b1) * gg
        vv = b2 * vv + (1 - b2) * gg * gg
        lr_t = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        pm -= lr_t * mm / (np.sqrt(vv) + eps)
    assert np.abs(pd.cpu().numpy() - pm).max() < 1e-5


@pytest.mark.parametrize("C,act,use_res", [(32, 1, False), (64, 1, True), (48, 2, False), (128, 2, True)])
def test_bn_kernels_emit_planes(C, act, use_res):
    """fused planes outputs of the BN/activation kernels: the planes decode to the fp32 result within the
    format's accuracy, and the statistics-derived bound really bounds the data (no fp16 overflow)"""
    from planes_util import planes_to_dense
    from tf2_yolo_amd import ops
    g = torch.Generator().manual_seed(7 * C + act)
    N, H, W = 3, 7, 5
    P = N * H * W
    dev = "cuda"
    x = (torch.randn(N, H, W, C, generator=g) * 2 + 0.7).to(dev)
    x[0, 0, 0, :] *= 30.0                                  # an outlier pixel
    res = torch.randn(N, H, W, C, generator=g).to(dev) * 3 if use_res else None
    gd = (torch.rand(C, generator=g) + 0.5).to(dev)
    bd = torch.randn(C, generator=g).to(dev)
    stats = torch.zeros(64 * 2 * C, device=dev, dtype=torch.float64)
    red = torch.zeros(513 * 2 * C, device=dev, dtype=torch.float64)
    f = lambda: torch.empty(C, device=dev)
    scale, shift, smean, sinv = f(), f(), f(), f()
    aux = torch.zeros(72, device=dev, dtype=torch.int32)
    rb = torch.tensor([float(res.abs().max()) if use_res else 0.0], device=dev)
    ob = torch.zeros(1, device=dev)
    ops.bn_stats(x, C, stats)
    ops.bn_finalize(stats, P, C, gd, bd, None, None, scale, shift, smean, sinv, bound=aux[0:1])
    loose = float(aux[0:1].view(torch.float32))
    # with the producer's per-channel max|x| the bound is tight (within the |mean| slack), without it only valid
    aux[0] = 0
    amax = x.reshape(P, C).abs().max(0).values.view(torch.int32).contiguous()
    ops.bn_finalize(stats, P, C, gd, bd, None, None, scale, shift, smean, sinv, bound=aux[0:1], absmax=amax)
    assert float(aux[0:1].view(torch.float32)) <= loose * 1.01
    pl = torch.zeros(ops.planes_bytes(P, C), device=dev, dtype=torch.uint8)
    o = ops.bn_act_fwd(x, C, scale, shift, act, res, planes=pl, bn_bound=aux[0:1], residual_bound=rb if use_res else None,
                       out_bound=ob)
    o_plain = ops.bn_act_fwd(x, C, scale, shift, act, res)
    assert torch.equal(o, o_plain)
    dense, bound, s, tail0 = planes_to_dense(pl.cpu(), P, C)
    amax = float(o.abs().max())
    assert tail0 and bound >= amax and float(ob) == bound and s * bound <= 2.0 ** 15
    ref = o.double().cpu().reshape(P, C)
    assert ((dense - ref).abs() <= torch.maximum(2.0 ** -22 * ref.abs(), torch.tensor(2.0 ** -25 / s, dtype=torch.float64))).all()
    # backward
    dout = torch.randn(N, H, W, C, generator=g).to(dev) * 1e-3
    dgamma, dbeta = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dx_plain = ops.bn_act_bwd(x, dout, C, gd, scale, shift, smean, sinv, act, red, dgamma, dbeta)
    red.zero_()
    pl2 = torch.zeros(ops.planes_bytes(P, C), device=dev, dtype=torch.uint8)
    dx = ops.bn_act_bwd(x, dout, C, gd, scale, shift, smean, sinv, act, red, None, None, planes=pl2, bound_aux=aux[1:69])
    assert torch.equal(dx, dx_plain)
    dense2, bound2, s2, tail02 = planes_to_dense(pl2.cpu(), P, C)
    assert tail02 and bound2 >= float(dx.abs().max()) and s2 * bound2 <= 2.0 ** 15
    ref2 = dx.double().cpu().reshape(P, C)
    assert ((dense2 - ref2).abs() <= torch.maximum(2.0 ** -22 * ref2.abs(), torch.tensor(2.0 ** -25 / s2, dtype=torch.float64))).all()
    red.zero_(); aux.zero_()
    ops.bn_stats(x, C, stats.zero_())
    only = ops.bn_act_bwd(x, dout, C, gd, scale, shift, smean, sinv, act, red, None, None, planes=pl2, want_dx=False,
                          bound_aux=aux[1:69])
    assert only is None


@pytest.mark.parametrize("C,act", [(32, 1), (64, 2), (128, 1), (48, 1)])
def test_bn_act_fwd_residual_from_planes(C, act):
    """yolo_bn_act_fwd_res_planes: the residual read from its PLANES (h + l = the value to 22-23 bits, bound from the planes
    header) instead of an fp32 copy -- what the engine does for residual blocks in training, so that the fp32 copy of a
    block's input is never written. Against the fp32-residual kernel: identical up to the planes' rounding of the
    residual (2^-22 of the value or 2^-25 / scale), the output bound is bn bound + residual bound."""
    from planes_util import planes_to_dense
    from tf2_yolo_amd import ops
    g = torch.Generator().manual_seed(100 + C + act)
    N, H, W = 2, 9, 7
    P = N * H * W
    x = (torch.randn(N, H, W, C, generator=g) * 2 + 0.3).cuda()
    res = (torch.randn(N, H, W, C, generator=g) * 3).cuda()
    scale = (torch.rand(C, generator=g) + 0.5).cuda()
    shift = torch.randn(C, generator=g).cuda()
    rpl = ops.split_planes(res, P, C)
    rdense, rbound, rs, _ = planes_to_dense(rpl.cpu(), P, C)
    bnb = torch.tensor([float(ops.bn_act_fwd(x, C, scale, shift, act).abs().max()) * 1.01], device="cuda").view(torch.int32)
    ob, ob2 = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    pl = torch.zeros(ops.planes_bytes(P, C), device="cuda", dtype=torch.uint8)
    pl2 = torch.zeros(ops.planes_bytes(P, C), device="cuda", dtype=torch.uint8)
    o = ops.bn_act_fwd(x, C, scale, shift, act, None, planes=pl, bn_bound=bnb, out_bound=ob, residual_planes=rpl)
    o_ref = ops.bn_act_fwd(x, C, scale, shift, act, res, planes=pl2, bn_bound=bnb,
                           residual_bound=torch.tensor([rbound], device="cuda"), out_bound=ob2)
    torch.cuda.synchronize()
    assert float(ob) == float(ob2)
    # exactly act(bn(x)) + the DECODED residual
    plain = ops.bn_act_fwd(x, C, scale, shift, act)
    assert torch.equal(o, plain + rdense.float().reshape(N, H, W, C).cuda())
    err = (o.double() - o_ref.double()).abs().cpu().reshape(P, C)
    lim = torch.maximum(2.0 ** -22 * res.double().abs().cpu().reshape(P, C), torch.tensor(2.0 ** -24 / rs, dtype=torch.float64))
    assert (err <= lim + 2.0 ** -22 * o_ref.double().abs().cpu().reshape(P, C)).all()   # (+ one fp32 rounding of each sum)
    # planes-only form (no fp32 output at all)
    pl3 = torch.zeros(ops.planes_bytes(P, C), device="cuda", dtype=torch.uint8)
    assert ops.bn_act_fwd(x, C, scale, shift, act, None, planes=pl3, bn_bound=bnb, want_out=False, residual_planes=rpl) is None
    torch.cuda.synchronize()
    assert torch.equal(pl3, pl)


@pytest.mark.parametrize("chans", [(256, 128), (512, 512, 512, 512), (64,), (24, 40)])
def test_split_planes_concat(chans):
    """yolo_split_planes_concat: up to four fp32 sources -> the planes of their channel concatenation in one pass, the scale
    from the sources' RECORDED bounds (loose on purpose here: 3x the true maximum). A 1x1 convolution on those planes equals
    the convolution of the concatenated tensor to fp32 summation accuracy; the optional fp32 output is the exact
    concatenation; the result's bound is the largest source bound."""
    from tf2_yolo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(4)
    n, h, w = 2, 13, 19
    rows = n * h * w
    srcs = [torch.randn(rows, c, device="cuda", generator=g) * (1.0 + 3.0 * i) for i, c in enumerate(chans)]
    bounds = [(t.abs().max() * 3.0).reshape(1) for t in srcs]
    C = sum(chans)
    pl = torch.zeros(ops.planes_bytes(rows, C), device="cuda", dtype=torch.uint8)
    d32 = torch.empty(rows, C, device="cuda")
    ob = torch.zeros(1, device="cuda")
    ops.split_planes_concat(srcs, list(chans), bounds, rows, pl, dst32=d32, out_bound=ob)
    cat = torch.cat(srcs, dim=1)
    assert torch.equal(d32, cat)
    assert abs(float(ob) - float(max(b.item() for b in bounds)) * 1.001) <= 1e-5 * float(ob)
    cout = 64
    wk = torch.randn(cout, C, device="cuda", generator=g) / C ** 0.5
    d = ops.conv_desc((n, h, w, C), cout, 1, 1, 1, "same")
    y = ops.conv2d_fwd_planes(d, pl, ops.split_planes(wk, cout, C))
    ref = cat.double() @ wk.double().t()
    assert (y.reshape(rows, cout).double() - ref).abs().max().item() / ref.abs().max().item() < 2e-6
    # without the fp32 output, planes only
    pl2 = torch.zeros_like(pl)
    ops.split_planes_concat(srcs, list(chans), bounds, rows, pl2)
    assert torch.equal(pl, pl2)


@pytest.mark.parametrize("n,h,w,chans,up", [(1, 26, 26, (256, 512), (True, False)), (2, 52, 52, (128, 256), (True, False)),
                                            (3, 6, 10, (24, 40, 16), (False, True, True)), (1, 14, 14, (64,), (True,))])
def test_split_planes_concat_reads_through_upsampling(n, h, w, chans, up):
    """yolo_split_planes_concat_ex (round 6): sources behind UpSampling2D(2) are read at half size, bounds come as WORDS
    (their maximum is the bound). Planes, fp32 output and bound bit-identical to yolo_split_planes_concat on the explicitly
    upsampled tensors (yolo_upsample2x_fwd) with the folded bounds (yolov3/models/darknet.py:87-93)."""
    from tf2_yolo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(14)
    rows = n * h * w
    half = [torch.randn(n, h // 2, w // 2, c, device="cuda", generator=g) * (1.0 + i) if f else None
            for i, (c, f) in enumerate(zip(chans, up))]
    def _up(t, c):
        y = torch.empty(n, h, w, c, device="cuda")
        ops.upsample2x_fwd(t, y, c, 0)
        return y.reshape(rows, c)
    full = [_up(t, c) if f else torch.randn(rows, c, device="cuda", generator=g) * 2.5 for t, c, f in zip(half, chans, up)]
    srcs = [t if f else x for t, x, f in zip(half, full, up)]
    # bounds: words (many, the maximum somewhere in the middle; the others smaller or zero), or one float
    words, floats = [], []
    for i, x in enumerate(full):
        m = x.abs().max() * 1.5
        if i % 2 == 0:
            wds = torch.zeros(300 + 77 * i, device="cuda")
            wds[torch.randint(0, wds.numel(), (50,), device="cuda", generator=g)] = m * 0.25
            wds[123] = m
            words.append(wds.view(torch.int32))
        else:
            words.append(m.reshape(1))
        floats.append(m.reshape(1).clone())
    C = sum(chans)
    pl = torch.zeros(ops.planes_bytes(rows, C), device="cuda", dtype=torch.uint8)
    d32 = torch.empty(rows, C, device="cuda")
    ob = torch.zeros(1, device="cuda")
    ops.split_planes_concat_ex(srcs, list(chans), words, rows, pl, upsample=list(up), hw=(h, w), dst32=d32, out_bound=ob)
    pl_ref = torch.zeros_like(pl)
    d32_ref = torch.empty_like(d32)
    ob_ref = torch.zeros(1, device="cuda")
    ops.split_planes_concat(full, list(chans), floats, rows, pl_ref, dst32=d32_ref, out_bound=ob_ref)
    torch.cuda.synchronize()
    assert torch.equal(d32, d32_ref) and torch.equal(d32, torch.cat(full, dim=1))
    assert float(ob) == float(ob_ref)
    body = ops.planes_bytes(rows, C) - 256
    assert torch.equal(pl[:body + 12], pl_ref[:body + 12])
    # planes only
    pl2 = torch.zeros_like(pl)
    ops.split_planes_concat_ex(srcs, list(chans), words, rows, pl2, upsample=list(up), hw=(h, w))
    assert torch.equal(pl2[:body + 12], pl[:body + 12])


@pytest.mark.parametrize("P,C,act", [(32 * 52 * 52, 256, 1), (32 * 13 * 13, 1024, 1), (4 * 19 * 19, 512, 2), (2 * 7 * 5, 32, 1),
                                     (16 * 26 * 26 + 3, 48, 1), (32 * 104 * 104, 64, 1)])
def test_bn_backward_reduction_finished_by_its_own_launch(P, C, act):
    """yolo_bn_act_bwd_reduce_fold_ld (round 6): the BatchNormalization-backward reduction in ONE launch -- the workgroups that
    arrive last fold the per-workgroup slots, two levels of ticket words -- against the two-launch form
    (bn_bwd_reduce + bn_bwd_sum): the final fp64 sums agree to 1e-13 of their magnitude (another fixed summation order), the
    bound words agree, dx / its planes are equal to float rounding, the ticket words are back at zero, and two runs are
    BIT-identical (who arrives last changes nothing in the arithmetic). Sizes: the benchmark's 52x52x256 and 13x13x1024
    layers at bs 32 (512 workgroups: 32 full groups), a 208x208-sized tensor, and small / ragged ones (one group, a partial
    last group, fewer workgroups than a group holds)."""
    from tf2_yolo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(P % 1000 + C + act)
    dev = "cuda"
    x = torch.randn(P, C, device=dev, generator=g) * 2 + 0.3
    dout = torch.randn(P, C, device=dev, generator=g) * 1e-3
    gd = torch.rand(C, device=dev, generator=g) + 0.5
    bd = torch.randn(C, device=dev, generator=g)
    stats = torch.zeros(64 * 2 * C, device=dev, dtype=torch.float64)
    f = lambda: torch.empty(C, device=dev)
    scale, shift, smean, sinv = f(), f(), f(), f()
    ops.bn_stats(x, C, stats)
    ops.bn_finalize(stats, P, C, gd, bd, None, None, scale, shift, smean, sinv)

    def run(fold):
        red = torch.full(((ops.BN_RED_SLOTS + 1) * 2 * C,), float("nan"), device=dev, dtype=torch.float64)   # no slot is read unwritten
        aux = torch.zeros(68 + ops.BN_FOLD_TICKET_WORDS, device=dev, dtype=torch.int32)
        pl = torch.zeros(ops.planes_bytes(P, C), device=dev, dtype=torch.uint8) if C % 16 == 0 else None
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dx = ops.bn_act_bwd(x, dout, C, gd, scale, shift, smean, sinv, act, red, dg, db, planes=pl, bound_aux=aux[:68],
                            tickets=aux[68:] if fold else None)
        torch.cuda.synchronize()
        return red[ops.BN_RED_SLOTS * 2 * C:].clone(), aux.clone(), dx, pl, dg, db

    two = run(False)
    one = run(True)
    again = run(True)
    assert torch.isfinite(one[0]).all()
    mag = two[0].abs().max().item()
    assert (one[0] - two[0]).abs().max().item() <= 1e-13 * max(mag, 1e-300)
    assert int(one[1][68:].abs().max()) == 0                       # tickets back at zero
    a1, a2 = one[1][:3].view(torch.float32), two[1][:3].view(torch.float32)
    assert a1[0] == a2[0] and torch.allclose(a1[1:], a2[1:], rtol=1e-6, atol=0)
    assert torch.allclose(one[2], two[2], rtol=1e-5, atol=1e-9) and torch.allclose(one[4], two[4], rtol=1e-6, atol=1e-9)
    for a, b in zip(one, again):                                   # run-to-run bit identity
        if a is not None:
            assert torch.equal(a, b)
    # against a float64 evaluation of the two sums
    xd, dd = x.double(), dout.double()
    z = xd * scale.double() + shift.double()
    if act == 1:
        dz = dd * torch.where(z > 0, 1.0, 0.1)
        ref0, ref1 = dz.sum(0), (dz * ((xd - smean.double()) * sinv.double())).sum(0)
        assert (one[0][:C] - ref0).abs().max().item() <= 1e-5 * ref0.abs().max().item()
        assert (one[0][C:] - ref1).abs().max().item() <= 1e-5 * ref1.abs().max().item()


@pytest.mark.parametrize("N,hh,ww,c", [(2, 8, 6, 8), (3, 26, 26, 64), (1, 2, 2, 4), (2, 104, 104, 32)])
def test_maxpool_2x2_stride2_tiling_kernels(N, hh, ww, c):
    """2x2 / stride-2 pools whose windows tile the input (C % 4 == 0) run on the 16-byte-per-lane kernels (round 6;
    csrc/elementwise.hip: maxpool2x2_fwd_kernel / maxpool2x2_bwd_kernel): values, recorded winners (first maximum in row-major
    window order, ties included) and the gradient against the oracle and against the general scatter kernel; the backward in
    its assign form overwrites a poisoned tensor completely (no zero fill needed), in its accumulate form it adds."""
    from oracle import layers as L
    from tf2_yolo_amd import ops
    g = torch.Generator().manual_seed(N * 1000 + hh + c)
    m = torch.randn(N, hh, ww, c, generator=g, dtype=torch.float64)
    m[:, ::2, ::2, : c // 2] = m[:, 1::2, 1::2, : c // 2]      # exact ties between the first and the last element of a window
    m = m.float().double().requires_grad_(True)
    ref = L.maxpool(m, 2, 2, "same")
    dref = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dref)
    Ho, Wo = hh // 2, ww // 2
    md = m.detach().float().cuda()
    out = torch.empty(N, Ho, Wo, c, device="cuda")
    arg = torch.empty(N, Ho, Wo, c, device="cuda", dtype=torch.int32)
    ops.maxpool_fwd(md, 2, 2, 0, 0, Ho, Wo, out, c, 0, arg)
    assert torch.equal(out.double().cpu(), ref.detach().float().double())
    # the general kernel (selected here by an output channel slice: Cy > C) records the same winners
    wide = torch.empty(N, Ho, Wo, c + 4, device="cuda")
    arg2 = torch.empty_like(arg)
    ops.maxpool_fwd(md, 2, 2, 0, 0, Ho, Wo, wide, c + 4, 4, arg2)
    assert torch.equal(arg, arg2) and torch.equal(wide[..., 4:], out)
    dyd = dref.float().cuda()
    dm = torch.full((N, hh, ww, c), float("nan"), device="cuda")
    ops.maxpool2x2_bwd(dyd, N, Ho, Wo, c, arg, dm, False)
    scat = torch.zeros(N, hh, ww, c, device="cuda")
    ops.maxpool_bwd(dyd, N, Ho, Wo, c, c, 0, arg, scat)
    assert torch.equal(dm, scat)
    assert _rel(dm, m.grad) < 1e-6
    base = torch.randn(N, hh, ww, c, device="cuda")
    acc = base.clone()
    ops.maxpool2x2_bwd(dyd, N, Ho, Wo, c, arg, acc, True)
    assert torch.equal(acc, base + scat)


@pytest.mark.parametrize("N,hh,ww,C,act", [(2, 8, 6, 16, 1), (3, 26, 26, 64, 1), (1, 4, 4, 32, 2), (2, 52, 52, 32, 1), (1, 6, 10, 8, 1)])
def test_bn_act_maxpool2x2_in_one_pass(N, hh, ww, C, act):
    """yolo_bn_act_maxpool2x2_fwd (round 6): BatchNorm apply + activation + MaxPooling2D(2, 2) + the planes of the pooled
    tensor in one launch against the three-launch path (bn_act_fwd -> maxpool_fwd -> split_planes): the pooled values and the
    recorded winners are bit-identical, the planes decode to the pooled tensor within the format's accuracy under the bound of
    the UNPOOLED activation, their tail rows and zero block are zero; C = 8 (no planes) gives the fp32 tensor alone."""
    from planes_util import planes_to_dense
    from tf2_yolo_amd import ops
    g = torch.Generator().manual_seed(N + hh + C + act)
    dev = "cuda"
    y = (torch.randn(N, hh, ww, C, generator=g) * 2 + 0.3).to(dev)
    y[:, ::2, ::2] = y[:, 1::2, 1::2]                      # exact ties inside every window (first element wins)
    P = N * hh * ww
    gd = (torch.rand(C, generator=g) + 0.5).to(dev)
    bd = torch.randn(C, generator=g).to(dev)
    stats = torch.zeros(64 * 2 * C, device=dev, dtype=torch.float64)
    f = lambda: torch.empty(C, device=dev)
    scale, shift, smean, sinv = f(), f(), f(), f()
    aux = torch.zeros(4, device=dev, dtype=torch.int32)
    ops.bn_stats(y, C, stats)
    ops.bn_finalize(stats, P, C, gd, bd, None, None, scale, shift, smean, sinv, bound=aux[0:1])
    a = ops.bn_act_fwd(y, C, scale, shift, act)
    Ho, Wo = hh // 2, ww // 2
    ref = torch.empty(N, Ho, Wo, C, device=dev)
    arg_ref = torch.empty(N, Ho, Wo, C, device=dev, dtype=torch.int32)
    ops.maxpool_fwd(a, 2, 2, 0, 0, Ho, Wo, ref, C, 0, arg_ref)
    out = torch.full((N, Ho, Wo, C), float("nan"), device=dev)
    arg = torch.full((N, Ho, Wo, C), -5, device=dev, dtype=torch.int32)
    pl = torch.full((ops.planes_bytes(N * Ho * Wo, C),), 0x55, device=dev, dtype=torch.uint8) if C % 16 == 0 else None
    ob = torch.zeros(1, device=dev)
    ops.bn_act_maxpool2x2_fwd(y, C, scale, shift, act, arg, out=out, planes=pl, bn_bound=aux[0:1], out_bound=ob)
    assert torch.equal(out, ref) and torch.equal(arg, arg_ref)
    if pl is not None:
        dense, bound, s, tail0 = planes_to_dense(pl.cpu(), N * Ho * Wo, C)
        assert tail0 and bound == float(aux[0:1].view(torch.float32)) == float(ob) and bound >= float(a.abs().max())
        r = ref.double().cpu().reshape(-1, C)
        assert ((dense - r).abs() <= torch.maximum(2.0 ** -22 * r.abs(), torch.tensor(2.0 ** -25 / s, dtype=torch.float64))).all()
        only = torch.full_like(pl, 0x33)
        arg2 = torch.empty_like(arg)
        ops.bn_act_maxpool2x2_fwd(y, C, scale, shift, act, arg2, out=None, planes=only, bn_bound=aux[0:1])
        used = pl.numel() - 256 + 12          # body + the three header words (the rest of the 256-byte header is never written)
        assert torch.equal(only[:used], pl[:used]) and torch.equal(arg2, arg)
