"""GPU parity of the HBM-bound kernels (BN, pooling, resizes, blur, losses, EDT, epilogue backward) through
the C ABI against torch fp32 CPU ops / the oracle.  fp16-storage kernels: 2e-3 of the tensor's max magnitude;
fp32 kernels: 1e-5."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _eng():
    from csbsr_amd.engine import Engine
    return Engine("cuda:0")


def to_fm(x):
    from csbsr_amd.engine import FM, pad8
    N, Cc, H, W = x.shape
    t = torch.zeros(N, H, W, pad8(Cc), dtype=torch.float16)
    t[..., :Cc] = x.permute(0, 2, 3, 1).half()
    return FM(t.cuda(), Cc)


def from_fm(fm):
    return fm.t[..., :fm.c].float().cpu().permute(0, 3, 1, 2)


def relmax(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-20))


def r16(x):
    return x.half().float()


_KEEP = []


def P(t):
    """device pointer of a tensor; the tensor is kept alive until the test module is torn down (a temporary such as
    ``x.cuda()`` would otherwise be freed -- and its block reused -- before the asynchronous kernel runs)."""
    from csbsr_amd.engine import _ptr
    if t is not None:
        _KEEP.append(t)
        if len(_KEEP) > 256:
            torch.cuda.synchronize()
            del _KEEP[:128]
    return _ptr(t)


@pytest.mark.parametrize("act,res_mode", [("relu", 0), ("prelu", 1), ("prelu", 2), ("lrelu", 0), ("sigmoid", 0), ("none", 4)])
def test_epilogue_backward(act, res_mode):
    from csbsr_amd import _lib as L
    eng = _eng()
    torch.manual_seed(1)
    N, Cc, H, W = 2, 20, 7, 9
    pre = torch.randn(N, Cc, H, W).requires_grad_(True)
    res = r16(torch.randn(N, Cc, H, W)).requires_grad_(True)
    res2 = r16(torch.randn(N, Cc, H, W)).requires_grad_(True)
    slope = torch.tensor([0.2], requires_grad=True)
    bias = torch.zeros(Cc, requires_grad=True)
    z = pre + bias.view(1, -1, 1, 1)
    y = {"relu": F.relu, "prelu": lambda v: F.prelu(v, slope), "lrelu": lambda v: F.leaky_relu(v, 0.1), "sigmoid": torch.sigmoid,
         "none": lambda v: v}[act](z)
    out = {0: y, 1: y + res, 2: y - res, 4: y + res * res2}[res_mode]
    out16 = r16(out.detach())
    dout = r16(torch.randn_like(out))
    # reference on the fp16-rounded saved output (what the kernel sees)
    out.backward(dout)
    actc = {"relu": L.ACT_RELU, "prelu": L.ACT_PRELU, "lrelu": L.ACT_LRELU, "sigmoid": L.ACT_SIGMOID, "none": L.ACT_NONE}[act]
    d = to_fm(dout)
    dres, dres2 = eng.new(N, H, W, Cc), eng.new(N, H, W, Cc)
    dbias = torch.zeros(24, device="cuda")
    dpr = torch.zeros(1, device="cuda")
    sl = slope.detach().cuda()
    eng.epilogue_bwd(d, out=to_fm(out16), act=actc, slope=0.1, prelu=sl if act == "prelu" else None,
                     res=to_fm(res.detach()) if res_mode else None, res2=to_fm(res2.detach()) if res_mode == 4 else None, res_mode=res_mode,
                     dpre=d, dres=dres if res_mode else None, dres2=dres2 if res_mode == 4 else None, dbias=dbias,
                     dprelu=dpr if act == "prelu" else None)
    torch.cuda.synchronize()
    assert relmax(from_fm(d), pre.grad) < 4e-3
    assert relmax(dbias[:Cc].cpu(), bias.grad) < 4e-3
    if res_mode:
        assert relmax(from_fm(dres), res.grad) < 2e-3
    if res_mode == 4:
        assert relmax(from_fm(dres2), res2.grad) < 2e-3
    if act == "prelu":
        assert abs(float(dpr) - float(slope.grad)) < 5e-3 * float(pre.grad.abs().sum())


@pytest.mark.parametrize("act,with_res,with_drop", [("relu", False, False), ("relu", True, False), ("prelu", False, True), ("none", False, False)])
def test_batchnorm_train_fwd_bwd(act, with_res, with_drop):
    from csbsr_amd import _lib as L
    from csbsr_amd.engine import BatchNorm, Conv
    eng = _eng()
    torch.manual_seed(2)
    N, Cc, H, W = 3, 16, 9, 10
    x = r16(torch.randn(N, 8, H, W))
    w = r16(torch.randn(Cc, 8, 3, 3) / 8)
    gamma = (1 + 0.2 * torch.randn(Cc)).requires_grad_(True)
    beta = (0.1 * torch.randn(Cc)).requires_grad_(True)
    res = r16(torch.randn(N, Cc, H, W)).requires_grad_(True)
    slope = torch.tensor([0.25], requires_grad=True)
    drop = ((torch.rand(N, Cc) > 0.3).float() / 0.7) if with_drop else None
    params = {"c.weight": w.cuda(), "b.weight": gamma.detach().cuda(), "b.bias": beta.detach().cuda(),
              "b.running_mean": torch.zeros(Cc).cuda(), "b.running_var": torch.ones(Cc).cuda(),
              "b.num_batches_tracked": torch.zeros((), dtype=torch.long).cuda()}
    conv = Conv(eng, "c", params, 3, 1, 1, bias=False)
    bn = BatchNorm(eng, "b", params, Cc)
    stat = bn.new_stat()
    raw = conv.fwd(to_fm(x), stat=stat, stat_mode=L.STAT_BN)
    mean, invstd = bn.finalize(stat, raw.npix)
    actc = {"relu": L.ACT_RELU, "prelu": L.ACT_PRELU, "none": L.ACT_NONE}[act]
    sl = slope.detach().cuda()
    dr = drop.cuda() if with_drop else None
    y = bn.apply(raw, mean, invstd, act=actc, prelu=sl if act == "prelu" else None, res=to_fm(res.detach()) if with_res else None, drop=dr)
    # reference
    rawr = F.conv2d(x, w, None, 1, 1).requires_grad_(True)
    rm, rv = torch.zeros(Cc), torch.ones(Cc)
    z = F.batch_norm(rawr, rm, rv, gamma, beta, True, 0.1, 1e-5)
    if with_res:
        z = z + res
    yr = {"relu": F.relu, "prelu": lambda v: F.prelu(v, slope), "none": lambda v: v}[act](z)
    if with_drop:
        yr = yr * drop[:, :, None, None]
    torch.cuda.synchronize()
    assert relmax(from_fm(y), yr) < 3e-3
    assert relmax(params["b.running_mean"].cpu(), rm) < 1e-3 and relmax(params["b.running_var"].cpu(), rv) < 1e-3
    dy = r16(torch.randn_like(yr))
    yr.backward(dy)
    dres = eng.new(N, H, W, Cc)
    dpr = torch.zeros(1, device="cuda")
    dx = bn.backward(to_fm(dy), raw, mean, invstd, act=actc, prelu=sl if act == "prelu" else None, res=to_fm(res.detach()) if with_res else None,
                     drop=dr, dres=dres if with_res else None, dprelu=dpr if act == "prelu" else None)
    torch.cuda.synchronize()
    assert relmax(from_fm(dx), rawr.grad) < 5e-3
    assert relmax(params["b.weight"].gacc.cpu(), gamma.grad) < 5e-3
    assert relmax(params["b.bias"].gacc.cpu(), beta.grad) < 5e-3
    if with_res:
        assert relmax(from_fm(dres), res.grad) < 3e-3
    if act == "prelu":
        assert abs(float(dpr) - float(slope.grad)) < 1e-2 * abs(float(slope.grad)) + 1e-2


def test_maxpool_and_adaptive_pool():
    from csbsr_amd import _lib as L
    eng = _eng()
    torch.manual_seed(3)
    x = r16(torch.relu(torch.randn(2, 16, 11, 14))).requires_grad_(True)     # relu: ties at 0 exercise the argmax rule
    y = F.max_pool2d(x, 3, 2, 1)
    dy = r16(torch.randn_like(y))
    y.backward(dy)
    fx = to_fm(x.detach())
    fy = eng.new(2, y.shape[2], y.shape[3], 16)
    L.call("csbsr_maxpool3x3s2_fwd", P(fx.t), P(fy.t), 2, 11, 14, 16, eng.stream)
    fdx = eng.new(2, 11, 14, 16)
    L.call("csbsr_maxpool3x3s2_bwd", P(fx.t), P(fy.t), P(to_fm(dy).t), P(fdx.t), 2, 11, 14, 16, eng.stream)
    torch.cuda.synchronize()
    assert relmax(from_fm(fy), y.detach()) == 0.0
    assert relmax(from_fm(fdx), x.grad) < 2e-3
    for (n_, c_, h_, w_) in ((1, 8, 7, 9), (2, 24, 16, 12), (1, 8, 1, 5)):          # odd / even / degenerate sizes, many ties (values on a coarse grid)
        xq = r16((torch.randn(n_, c_, h_, w_) * 2).round() / 2).requires_grad_(True)
        yq = F.max_pool2d(xq, 3, 2, 1)
        dq = r16(torch.randn_like(yq))
        yq.backward(dq)
        fq = to_fm(xq.detach())
        fo = eng.new(n_, yq.shape[2], yq.shape[3], c_)
        L.call("csbsr_maxpool3x3s2_fwd", P(fq.t), P(fo.t), n_, h_, w_, c_, eng.stream)
        fd = eng.new(n_, h_, w_, c_)
        L.call("csbsr_maxpool3x3s2_bwd", P(fq.t), P(fo.t), P(to_fm(dq).t), P(fd.t), n_, h_, w_, c_, eng.stream)
        torch.cuda.synchronize()
        assert relmax(from_fm(fo), yq.detach()) == 0.0
        assert relmax(from_fm(fd), xq.grad) < 2e-3, (n_, c_, h_, w_)
    for size in (1, 2, 3, 6):
        x2 = r16(torch.randn(2, 24, 8, 8)).requires_grad_(True)
        y2 = F.adaptive_avg_pool2d(x2, size)
        dy2 = r16(torch.randn_like(y2))
        y2.backward(dy2)
        f2 = to_fm(x2.detach())
        o2 = eng.new(2, size, size, 24)
        L.call("csbsr_adaptive_avgpool_fwd", P(f2.t), f2.ld, P(o2.t), 2, 8, 8, 24, size, size, eng.stream)
        d2 = eng.new(2, 8, 8, 24, zero=True)
        L.call("csbsr_adaptive_avgpool_bwd", P(to_fm(dy2).t), P(d2.t), d2.ld, 1, 2, 8, 8, 24, size, size, eng.stream)
        torch.cuda.synchronize()
        assert relmax(from_fm(o2), y2.detach()) < 2e-3
        assert relmax(from_fm(d2), x2.grad) < 2e-3


@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("shape,c", [((96, 128, 192, 256), 64), ((100, 75, 230, 160), 40), ((64, 64, 64, 64), 24), ((50, 120, 51, 300), 16)])
def test_bilinear_column_kernels(align, shape, c):
    """the column-owning forward (bilinear_fwd_cols_kernel) and the tabulated backward (bil_tab_kernel + bilinear_bwd_cols_kernel) at sizes
    that reach them (>= 64 workgroups; ratios <= 2.5 for the backward): against torch, with the dropout scale row and in accumulate mode"""
    eng = _eng()
    H, W, OH, OW = shape
    torch.manual_seed(H + c)
    N = 2
    x = r16(torch.randn(N, c, H, W)).requires_grad_(True)
    drop = (torch.rand(N, c) > 0.3).float() * 1.25
    y = F.interpolate(x, size=(OH, OW), mode="bilinear", align_corners=align) * drop[:, :, None, None]
    dy = r16(torch.randn_like(y))
    y.backward(dy)
    from csbsr_amd.engine import pad8
    dd = torch.zeros(N, pad8(c)); dd[:, :c] = drop
    dd = dd.cuda()
    fy = eng.bilinear(to_fm(x.detach()), OH, OW, align, drop=dd)
    base = r16(torch.randn(N, c, H, W))
    dx = to_fm(base)
    eng.bilinear_bwd(to_fm(dy), dx, True, align, drop=dd)
    torch.cuda.synchronize()
    assert relmax(from_fm(fy), y.detach()) < 2e-3
    assert relmax(from_fm(dx), base + x.grad) < 2e-3


@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("shape", [(8, 8, 16, 16), (1, 1, 8, 8), (3, 3, 8, 8), (6, 6, 8, 8), (8, 8, 64, 64), (5, 7, 20, 21)])
def test_bilinear_fwd_bwd(align, shape):
    from csbsr_amd import _lib as L
    eng = _eng()
    H, W, OH, OW = shape
    torch.manual_seed(4)
    x = r16(torch.randn(2, 8, H, W)).requires_grad_(True)
    y = F.interpolate(x, size=(OH, OW), mode="bilinear", align_corners=align)
    dy = r16(torch.randn_like(y))
    y.backward(dy)
    fy = eng.bilinear(to_fm(x.detach()), OH, OW, align)
    dx = eng.new(2, H, W, 8)
    eng.bilinear_bwd(to_fm(dy), dx, False, align)
    torch.cuda.synchronize()
    assert relmax(from_fm(fy), y.detach()) < 2e-3
    assert relmax(from_fm(dx), x.grad) < 2e-3
    # fp32 single-plane variant
    x1 = torch.randn(2, 1, H, W).requires_grad_(True)
    y1 = F.interpolate(x1, size=(OH, OW), mode="bilinear", align_corners=align)
    d1 = torch.randn_like(y1)
    y1.backward(d1)
    o = torch.empty(2, 1, OH, OW, device="cuda")
    L.call("csbsr_bilinear32_fwd", P(x1.detach().cuda()), P(o), 2, H, W, OH, OW, int(align), eng.stream)
    g = torch.empty(2, 1, H, W, device="cuda")
    L.call("csbsr_bilinear32_bwd", P(d1.cuda()), P(g), 2, H, W, OH, OW, int(align), eng.stream)
    torch.cuda.synchronize()
    assert relmax(o.cpu(), y1.detach()) < 1e-5
    assert relmax(g.cpu(), x1.grad) < 1e-5


@pytest.mark.parametrize("stride", [4, 1, 8])
def test_blur_fwd_bwd(stride):
    from csbsr_amd import _lib as L
    from oracle import csbsr_oracle as O
    eng = _eng()
    torch.manual_seed(5)
    N, H, W, K = 2, 36, 40, 21
    x = torch.rand(N, 3, H, W).requires_grad_(True)
    k = torch.rand(N, K * K)
    k = (k / k.sum(1, keepdim=True)).requires_grad_(True)
    y = O.blur_down(x, k, K, stride)
    sub = torch.rand_like(y)
    dy = torch.randn_like(y)
    (y - sub).backward(dy)
    xc, kc = x.detach().cuda(), k.detach().cuda()
    o = torch.empty_like(y, device="cuda")
    L.call("csbsr_blur_fwd", P(xc), P(kc), N, 3, H, W, K, stride, P(sub.cuda()), P(o), None, 0, eng.stream)
    dx = torch.empty(N, 3, H, W, device="cuda")
    L.call("csbsr_blur_bwd_input", P(dy.cuda()), P(kc), P(dx), 0, N, 3, H, W, K, stride, eng.stream)
    dk = torch.zeros(N, K * K, device="cuda")
    L.call("csbsr_blur_bwd_kernel", P(dy.cuda()), P(xc), P(dk), N, 3, H, W, K, stride, eng.stream)
    torch.cuda.synchronize()
    assert relmax(o.cpu(), (y - sub).detach()) < 1e-5
    assert relmax(dx.cpu(), x.grad) < 1e-5
    assert relmax(dk.cpu(), k.grad) < 1e-4
    # accumulate into an existing gradient (the tiled stride-4 kernel has its own read-modify-write path)
    dx2 = torch.ones(N, 3, H, W, device="cuda")
    L.call("csbsr_blur_bwd_input", P(dy.cuda()), P(kc), P(dx2), 1, N, 3, H, W, K, stride, eng.stream)
    torch.cuda.synchronize()
    assert relmax(dx2.cpu() - 1, x.grad) < 1e-5


def test_blur_kernel_gradient_stride1_multi_tile():
    """csbsr_blur_bwd_kernel, stride 1 (blur_bwd_kernel_s1_kernel: four taps per thread on a sliding register window, the two half
    workgroups on alternate tile rows): several ragged 32 x 32 tiles per plane against autograd of the oracle's blur."""
    from csbsr_amd import _lib as L
    from oracle import csbsr_oracle as O
    eng = _eng()
    torch.manual_seed(15)
    N, H, W, K = 2, 70, 97, 21
    x = torch.rand(N, 3, H, W)
    k = torch.rand(N, K * K)
    k = (k / k.sum(1, keepdim=True)).requires_grad_(True)
    y = O.blur_down(x, k, K, 1)
    dy = torch.randn_like(y)
    y.backward(dy)
    dk = torch.zeros(N, K * K, device="cuda")
    L.call("csbsr_blur_bwd_kernel", P(dy.cuda()), P(x.cuda()), P(dk), N, 3, H, W, K, 1, eng.stream)
    torch.cuda.synchronize()
    assert relmax(dk.cpu(), k.grad) < 1e-4
    dk2 = torch.zeros(N, K * K, device="cuda")
    L.call("csbsr_blur_bwd_kernel", P(dy.cuda()), P(x.cuda()), P(dk2), N, 3, H, W, K, 1, eng.stream)
    torch.cuda.synchronize()
    assert torch.equal(dk, dk2)          # order-fixed sums: bit-identical twice


@pytest.mark.parametrize("aa", [1, 0])
def test_bicubic_resizes(aa):
    from csbsr_amd import _lib as L
    eng = _eng()
    torch.manual_seed(6)
    x = torch.rand(2, 3, 32, 40).requires_grad_(True)
    y = F.interpolate(x, size=(8, 10), mode="bicubic", align_corners=False, antialias=bool(aa))
    dy = torch.randn_like(y)
    y.backward(dy)
    o = torch.empty(2, 3, 8, 10, device="cuda")
    L.call("csbsr_aa_bicubic_down_fwd", P(x.detach().cuda()), P(o), 6, 32, 40, 4, aa, eng.stream)
    dx = torch.empty(2, 3, 32, 40, device="cuda")
    L.call("csbsr_aa_bicubic_down_bwd", P(dy.cuda()), P(dx), 0, 6, 32, 40, 4, aa, eng.stream)
    torch.cuda.synchronize()
    assert relmax(o.cpu(), y.detach()) < 1e-5
    assert relmax(dx.cpu(), x.grad) < 1e-5
    xl = torch.rand(2, 3, 9, 7)
    up = F.interpolate(xl, scale_factor=4, mode="bicubic", align_corners=False)
    acc = torch.ones(2, 3, 36, 28, device="cuda")
    L.call("csbsr_bicubic_up_add", P(xl.cuda()), P(acc), 6, 9, 7, 4, eng.stream)
    torch.cuda.synchronize()
    assert relmax(acc.cpu() - 1, up) < 1e-5


def test_sdf_matches_reference_fixture_and_oracle():
    from csbsr_amd import _lib as L
    from golden_utils import load_golden
    from oracle import csbsr_oracle as O
    from csbsr_amd.data.synthetic import crack_masks
    eng = _eng()
    g = load_golden("sdf_handdrawn")
    m = torch.from_numpy(g["mask"].astype(np.float32)).cuda()
    N, _, H, W = m.shape
    sdf = torch.empty(N, 1, H, W, device="cuda")
    scratch = torch.empty(3 * N * H * W + 2 * N, device="cuda")
    L.call("csbsr_sdf", P(m), P(sdf), P(scratch), N, H, W, eng.stream)
    torch.cuda.synchronize()
    assert np.abs(sdf.cpu().numpy() - g["sdf"]).max() < 2e-6
    m2 = crack_masks(2, 160, 192, torch.Generator().manual_seed(3))
    m2 = m2[:, :, :160, :192].contiguous()
    ref = O.compute_sdf(m2.numpy())
    s2 = torch.empty(2, 1, 160, 192, device="cuda")
    sc2 = torch.empty(3 * 2 * 160 * 192 + 4, device="cuda")
    L.call("csbsr_sdf", P(m2.cuda()), P(s2), P(sc2), 2, 160, 192, eng.stream)
    torch.cuda.synchronize()
    assert np.abs(s2.cpu().numpy() - ref).max() < 2e-6
    # sparse masks on a wide image (the full-size bench data: nearest crack hundreds of columns away, many 32-column blocks without any
    # feature -- the block-skipping row pass of round 6), a ragged width, a single pixel in a corner, an empty and a full sample
    Hs, Ws = 96, 1000
    m3 = torch.zeros(5, 1, Hs, Ws)
    m3[0, 0, 40:43, 700:960] = 1.0
    m3[0, 0, 5:90, 17] = 1.0
    m3[1, 0, Hs - 1, Ws - 1] = 1.0
    m3[2, 0, 10:12, :] = 1.0
    m3[4] = 1.0
    ref3 = O.compute_sdf(m3.numpy())
    s3 = torch.empty(5, 1, Hs, Ws, device="cuda")
    sc3 = torch.empty(3 * 5 * Hs * Ws + 10, device="cuda")
    L.call("csbsr_sdf", P(m3.cuda()), P(s3), P(sc3), 5, Hs, Ws, eng.stream)
    torch.cuda.synchronize()
    got3 = s3.cpu().numpy()
    assert np.abs(got3[:4] - ref3[:4]).max() < 2e-6          # (sample 4, all foreground: the reference divides 0 / 0 there, the kernel writes 0)
    assert np.isfinite(got3).all() and np.abs(got3[3]).max() == 0.0


def test_segloss_and_l1():
    from csbsr_amd import _lib as L
    from oracle import csbsr_oracle as O
    eng = _eng()
    torch.manual_seed(7)
    N, H, W = 2, 24, 20
    p = torch.rand(N, 1, H, W)
    p[0, 0, 0, :4] = 0.0                       # exercises clamp(min=1e-8)
    p = p.requires_grad_(True)
    t = (torch.rand(N, 1, H, W) > 0.8).float()
    sdf = torch.from_numpy(O.compute_sdf(t.numpy())).float()
    cfg = O.PathCfg(bce_w=(20.0, 1.0), wbd_w=(1.0, 2.0))
    alpha = 0.6
    loss = O.boundary_combo_loss(p, t, alpha, cfg, sdf)
    gsc = torch.tensor([0.5, 2.0])
    (loss * gsc * 0.4).sum().backward()
    pc, tc, sc = p.detach().cuda(), t.cuda(), sdf.cuda()
    sums = torch.zeros(N, 8, device="cuda")
    lo = torch.zeros(N, device="cuda")
    dp = torch.empty(N, 1, H, W, device="cuda")
    L.call("csbsr_segloss_reduce", P(pc), P(tc), P(sc), N, H * W, P(sums), 20.0, 1.0, eng.stream)
    L.call("csbsr_segloss_finish", P(pc), P(tc), P(sc), N, H * W, P(sums), alpha, 20.0, 1.0, 1.0, 2.0, 0.4, P(gsc.cuda()), P(lo), P(dp), 0,
           eng.stream)
    torch.cuda.synchronize()
    assert relmax(lo.cpu(), 0.4 * loss.detach()) < 1e-5
    assert relmax(dp.cpu(), p.grad) < 1e-4
    a = torch.rand(N, 3, H, W).requires_grad_(True)
    b = torch.rand(N, 3, H, W)
    wm = torch.rand(N, 1, H, W) + 0.5
    l1 = (wm * (a - b).abs()).sum((1, 2, 3))
    (l1 * gsc).sum().backward()
    s = torch.zeros(N, device="cuda")
    da = torch.empty(N, 3, H, W, device="cuda")
    L.call("csbsr_l1_fwd_bwd", P(a.detach().cuda()), P(b.cuda()), P(wm.cuda()), N, 3, H * W, P(s), 1.0, P(gsc.cuda()), P(da), 0, eng.stream)
    torch.cuda.synchronize()
    assert relmax(s.cpu(), l1.detach()) < 1e-5
    assert relmax(da.cpu(), a.grad) < 1e-6


def test_instance_norm_fwd_bwd():
    from csbsr_amd import _lib as L
    eng = _eng()
    torch.manual_seed(8)
    N, H, W = 2, 20, 24
    x = torch.rand(N, 3, H, W).requires_grad_(True)
    y = F.instance_norm(x, eps=1e-5)
    dy = r16(torch.randn(N, 3, H, W))
    y.backward(dy)
    xc = x.detach().cuda()
    red = torch.zeros(N * 3, 2, device="cuda")
    L.call("csbsr_plane_reduce", P(xc), None, N * 3, H * W, P(red), eng.stream)
    mean = (red[:, 0] / (H * W)).contiguous()
    invstd = torch.rsqrt(red[:, 1] / (H * W) - mean * mean + 1e-5).contiguous()
    fm = eng.nchw32_to_fm(xc, mean=mean, invstd=invstd)
    dx = torch.empty(N, 3, H, W, device="cuda")
    red2 = torch.zeros(N * 3, 2, device="cuda")
    dfm = to_fm(dy)
    L.call("csbsr_instnorm_bwd", P(dfm.t), dfm.ld, P(xc), P(mean), P(invstd), P(dx), 0, N, 3, H * W, P(red2), eng.stream)
    torch.cuda.synchronize()
    assert relmax(from_fm(fm), y.detach()) < 2e-3
    assert relmax(dx.cpu(), x.grad) < 1e-4


@pytest.mark.parametrize("shape", [(2, 9, 11, 24), (1, 70, 66, 56), (2, 64, 64, 136)])
def test_border_class_fill_and_sums(shape):
    """constant-operand folding helpers: class fill and its adjoint (small-image path and the large-image path)."""
    from csbsr_amd import _lib as L
    eng = _eng()
    N, H, W, Cc = shape
    torch.manual_seed(9)
    V = torch.randn(N, 16, Cc)
    out = eng.new(N, H, W, Cc)
    L.call("csbsr_border_class_fill", P(V.cuda()), P(out.t), out.ld, N, H, W, Cc, eng.stream)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    cls = (yy == 0) * 8 + (yy == H - 1) * 4 + (xx == 0) * 2 + (xx == W - 1) * 1
    ref = V[:, cls.reshape(-1)].reshape(N, H, W, Cc).permute(0, 3, 1, 2)
    x = r16(torch.randn(N, Cc, H, W))
    sums = torch.zeros(N, 16, Cc, device="cuda")
    fx = to_fm(x)
    L.call("csbsr_border_class_sums", P(fx.t), fx.ld, P(sums), N, H, W, Cc, eng.stream)
    torch.cuda.synchronize()
    assert relmax(from_fm(out), r16(ref)) < 1e-6
    onehot = torch.nn.functional.one_hot(cls.reshape(-1), 16).float()          # [HW, 16]
    ref_s = torch.einsum("nchw,hwk->nkc", x, onehot.reshape(H, W, 16))
    assert relmax(sums.cpu(), ref_s) < 2e-4


@pytest.mark.parametrize("shape", [(1, 70, 66, 32), (3, 64, 64, 136), (2, 130, 40, 8)])
def test_border_class_sums_with_the_prelu_slope_term(shape):
    """csbsr_border_class_sums_prelu: the class sums of dPre plus, per channel, the sum of dPre * out over the pixels with out <= 0 --
    sum / slope^2 is the PReLU-slope gradient of the layer whose activation derivative rode on the dgrad above as a mask"""
    from csbsr_amd import _lib as L
    eng = _eng()
    N, H, W, Cc = shape
    torch.manual_seed(10)
    x, t = r16(torch.randn(N, Cc, H, W)), r16(torch.randn(N, Cc, H, W))
    fx, ft = to_fm(x), to_fm(t)
    sums = torch.zeros(N, 16, fx.cp, device="cuda")
    neg = torch.zeros(N, fx.cp, device="cuda")
    L.call("csbsr_border_class_sums_prelu", P(fx.t), fx.ld, P(ft.t), ft.ld, P(sums), P(neg), N, H, W, fx.cp, eng.stream)
    plain = torch.zeros(N, 16, fx.cp, device="cuda")
    L.call("csbsr_border_class_sums", P(fx.t), fx.ld, P(plain), N, H, W, fx.cp, eng.stream)
    torch.cuda.synchronize()
    assert torch.equal(sums, plain)
    ref = (x * t * (t <= 0)).double().sum((2, 3)).float()
    assert relmax(neg[:, :Cc].cpu(), ref) < 2e-4


@pytest.mark.parametrize("shape", [(2, 5, 5, 8), (2, 9, 11, 24), (1, 70, 66, 32), (3, 64, 64, 136)])
def test_ring_class_sums(shape):
    """the 25 two-ring class sums (adjoint of a per-class bias table with conv desc cbias_mode 1) vs a one-hot einsum, and twice: the
    reduction is order-fixed, so the two results must be bit-identical."""
    from csbsr_amd import _lib as L
    eng = _eng()
    N, H, W, Cc = shape
    torch.manual_seed(11)
    x = r16(torch.randn(N, Cc, H, W))
    fx = to_fm(x)
    res = []
    for _ in range(2):
        sums = torch.zeros(N, 25, Cc, device="cuda")
        L.call("csbsr_ring_class_sums", P(fx.t), fx.ld, P(sums), N, H, W, Cc, eng.stream)
        torch.cuda.synchronize()
        res.append(sums.cpu())
    typ = lambda v, n: torch.where(v < 2, v, torch.where(v >= n - 2, v - n + 5, torch.full_like(v, 2)))
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    cls = typ(yy, H) * 5 + typ(xx, W)
    onehot = torch.nn.functional.one_hot(cls.reshape(-1), 25).float().reshape(H, W, 25)
    ref = torch.einsum("nchw,hwk->nkc", x, onehot)
    assert torch.equal(res[0], res[1])
    assert relmax(res[0], ref) < 2e-4


@pytest.mark.parametrize("n,relu", [(1, True), (3, True), (4, False)])
def test_sum_act(n, relu):
    """n-ary fuse sum of an HRNet module (csbsr_sum_act) vs torch."""
    from csbsr_amd import _lib as L
    eng = _eng()
    torch.manual_seed(3)
    xs = [r16(torch.randn(2, 48, 9, 11)) for _ in range(n)]
    fms = [to_fm(x) for x in xs]
    out = eng.new(2, 9, 11, 48)
    ptrs = (C.c_void_p * 4)(*[P(f.t) for f in fms], *([None] * (4 - n)))
    lds = (C.c_int64 * 4)(*[f.ld for f in fms], *([0] * (4 - n)))
    L.call("csbsr_sum_act", out.npix, out.cp, n, ptrs, lds, P(out.t), out.ld, int(relu), eng.stream)
    ref = sum(xs)
    ref = F.relu(ref) if relu else ref
    assert relmax(from_fm(out), ref) < 2e-3


def test_weighted_pool_fwd_bwd():
    """soft object-region pooling of the OCR head (SpatialGather_Module for one class) and its adjoint vs torch autograd."""
    from csbsr_amd import _lib as L
    eng = _eng()
    torch.manual_seed(4)
    N, Cc, H, W = 3, 512, 37, 41                  # hw = 1517: two 1024-pixel chunks, ragged
    x = r16(torch.randn(N, Cc, H, W)).requires_grad_(True)
    logits = torch.randn(N, H * W)
    w = torch.softmax(logits, 1).requires_grad_(True)
    ref = torch.einsum("np,ncp->nc", w, x.reshape(N, Cc, -1))
    g = torch.randn(N, Cc)
    ref.backward(g)
    fx = to_fm(x.detach())
    out = torch.zeros(N, Cc, device="cuda")
    wd = w.detach().cuda().contiguous()
    L.call("csbsr_weighted_pool_fwd", P(fx.t), fx.ld, P(wd), P(out), N, H * W, Cc, eng.stream)
    assert relmax(out.cpu(), ref.detach()) < 1e-4
    base = r16(torch.randn(N, Cc, H, W))          # dx is accumulated into
    dx = to_fm(base)
    dw = torch.empty(N, H * W, device="cuda")
    gd = g.cuda().contiguous()
    L.call("csbsr_weighted_pool_bwd", P(fx.t), fx.ld, P(wd), P(gd), P(dx.t), dx.ld, P(dw), N, H * W, Cc, eng.stream)
    assert relmax(dw.cpu(), w.grad) < 1e-4
    assert relmax(from_fm(dx), base + x.grad) < 2e-3


@pytest.mark.parametrize("N,H,W,c,step", [(2, 64, 48, 128, 8), (3, 17, 23, 40, 4), (1, 9, 9, 8, 1), (4, 96, 96, 512, 8), (2, 520, 260, 64, 2)])
def test_subsampled_channel_mean(N, H, W, c, step):
    """csbsr_channel_mean_sub: per-sample, per-channel mean over every step-th row / column (the input statistic of the weight-rounding
    compensation, engine.Conv._dc_bias), on a channel SLICE of a wider buffer; two calls give the same bits."""
    import ctypes as C
    from csbsr_amd import _lib as L
    eng = _eng()
    g = torch.Generator().manual_seed(N * 100 + c)
    full = (torch.randn(N, H, W, c + 16, generator=g) + 0.5).half().cuda()
    x = full[..., 8:8 + c]
    outs = []
    for _ in range(2):
        m = torch.zeros(N, c, dtype=torch.float32, device="cuda")
        L.call("csbsr_channel_mean_sub", C.c_void_p(x.data_ptr()), x.stride(0), x.stride(1), x.stride(2), N, H, W, c, step, C.c_void_p(m.data_ptr()), eng.stream)
        torch.cuda.synchronize()
        outs.append(m.clone())
    ref = x[:, ::step, ::step].float().mean((1, 2))
    assert torch.equal(outs[0], outs[1])
    assert float((outs[0] - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("shape,mode", [((64, 32, 3, 3), 1), ((16, 24, 8, 8), 4), ((16, 24, 8, 8), 1), ((8, 8, 12, 12), 8), ((40, 24, 1, 1), 1),
                                        ((5, 7, 3, 3), 0), ((128, 128, 8, 8), 4), ((8, 8, 12, 12), 1), ((3, 128, 8, 8), 4), ((128, 128, 8, 8), 1)])
def test_tap_sum_preserving_weight_rounding(shape, mode):
    """csbsr_round_weights (engine.Conv._wq): every value is an fp16 number at most one fp16 step from the weight; per (d0, d1) and tap
    group (all taps, or the taps of one output phase of a stride-``mode`` transposed layer) the summed rounding residual is under half an
    ulp of the group; S = the tap sums of what is left; mode 0 = round to nearest.  Against the study's CPU form of the same greedy rule
    (tests/study_kbpn_precision.py::tapsum_round: ties may resolve differently, so a small mismatch fraction is allowed)."""
    import ctypes as C
    from csbsr_amd import _lib as L
    from study_kbpn_precision import tapsum_round
    eng = _eng()
    g = torch.Generator().manual_seed(sum(shape) + mode)
    w = torch.randn(shape, generator=g) / (shape[1] * shape[2] * shape[3]) ** 0.5
    w.view(-1)[::97] *= 30.0                      # a few large and a few tiny values: other binades, fp16 subnormals
    w.view(-1)[5::89] *= 1e-4
    wd = w.cuda()
    q = torch.empty_like(wd)
    S = torch.empty(shape[0], shape[1], device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    L.call("csbsr_round_weights", P(wd), P(q), P(S), shape[0], shape[1], shape[2], shape[3], mode, eng.stream)
    torch.cuda.synchronize()
    q, S = q.cpu(), S.cpu()
    assert torch.equal(q, q.half().float())
    d = w - q
    ulp = torch.exp2(torch.floor(torch.log2(q.abs().clamp_min(2.0 ** -14))) - 10)
    assert bool((d.abs() <= ulp * 1.0001 + 1e-12).all())          # (ulp of the value it landed on: moves are to the NEIGHBOUR, also across a binade)
    assert float((S - d.sum((2, 3))).abs().max()) < 1e-7
    if mode == 0:
        assert torch.equal(q, w.half().float())
        return
    K0, K1 = shape[2], shape[3]
    ph = (torch.arange(K0)[:, None] % mode) * mode + (torch.arange(K1)[None, :] % mode)
    d0 = w - w.half().float()
    for gidx in range(mode * mode):
        m = (ph == gidx)
        r = d[:, :, m].sum(-1)
        assert bool((r.abs() <= 0.5 * ulp[:, :, m].amax(-1) * 1.0001 + 1e-12).all()), gidx
        if m.sum() >= 9:        # nearest rounding leaves a random walk: the preserved sums are several times smaller (judged on the filters without
            # one of the 30x enlarged taps: such a tap keeps its own residual, nothing smaller can cancel it)
            big = torch.zeros(w.numel(), dtype=torch.bool)
            big[::97] = True
            sel = ~big.view(shape)[:, :, m].any(-1)
            if int(sel.sum()) >= 20:          # (a 144-tap filter always holds one of them)
                assert float(r[sel].pow(2).mean().sqrt()) < 0.6 * float(d0[:, :, m].sum(-1)[sel].pow(2).mean().sqrt())
    assert float(d.pow(2).mean().sqrt()) < 1.35 * float(d0.pow(2).mean().sqrt())      # the price: a few taps one step further away
    ref = tapsum_round(w, transposed=mode > 1, stride=mode)
    assert float((ref == q).float().mean()) > 0.995


def test_adam_step_matches_torch():
    """csbsr_amd.optim.Adam (one multi-tensor HIP launch per group) against torch.optim.Adam with the reference's hyper-parameters
    (/root/reference/train.py:91) over six steps: odd sizes (scalar, a tail that is not a multiple of four, several 8192-element chunks), a
    parameter whose gradient is None in two of the steps (its moments and step count must not advance), a learning-rate change between
    steps (what LambdaLR does), and the state_dict interchange both ways."""
    from csbsr_amd.optim import Adam
    torch.manual_seed(21)
    shapes = [(1,), (7,), (3, 3, 3, 3), (64, 33, 3, 3), (20000,), (8192,), (128, 128, 8, 8)]
    P0 = [torch.randn(s) * 0.1 for s in shapes]
    pa = [torch.nn.Parameter(t.clone().cuda()) for t in P0]
    pb = [torch.nn.Parameter(t.clone().cuda()) for t in P0]
    oa = Adam(pa, lr=2e-5, betas=(0.9, 0.999), eps=1e-8)
    ob = torch.optim.Adam(pb, lr=2e-5, betas=(0.9, 0.999), eps=1e-8)
    for it in range(6):
        for k, (a, b) in enumerate(zip(pa, pb)):
            if k == 3 and it in (1, 4):
                a.grad = b.grad = None
                continue
            g = torch.randn_like(a) * (10.0 ** (-(k % 4)))
            a.grad, b.grad = g.clone(), g.clone()
        if it == 3:
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 5e-4
        oa.step(); ob.step()
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(pa, pb)):
        sa, sb = oa.state[a], ob.state[b]
        assert float(sa["step"]) == float(sb["step"]) == (4.0 if k == 3 else 6.0)
        for key in ("exp_avg", "exp_avg_sq"):
            assert float((sa[key] - sb[key]).abs().max()) <= 2e-6 * float(sb[key].abs().max()) + 1e-30, (k, key)
        assert float((a - b).abs().max()) <= 1e-6 * 5e-4 * 6 + 2e-7 * float(b.abs().max()), k      # (six steps of at most ~lr each)
    # state_dict interchange: torch -> ours -> one more identical step
    oa2 = Adam(pa, lr=5e-4, betas=(0.9, 0.999), eps=1e-8)
    oa2.load_state_dict(oa.state_dict())
    ob2 = torch.optim.Adam(pa, lr=5e-4, betas=(0.9, 0.999), eps=1e-8)
    ob2.load_state_dict(oa.state_dict())
    assert float(oa2.state[pa[0]]["step"]) == 6.0 and set(ob2.state_dict()["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
