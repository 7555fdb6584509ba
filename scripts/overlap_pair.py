"""CU-partitioned streams, measured: (A) which CUs a hipExtStreamCreateWithCUMask prefix mask selects,
(B) how each kernel class of the backward scales with the number of CUs it is confined to, (C) one MFMA-bound weight gradient
next to one HBM-bound link of the dgrad chain on disjoint CU sets against the same launches back to back on the whole chip.

    python scripts/overlap_pair.py [trace] [scale] [pair]        (default: all three)
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from csbsr_amd import _lib as L
from csbsr_amd.engine import Engine, Conv, FM, BatchNorm, pad8

NW = 10           # mask words: 320 bits (the device reports 256 CUs; bits past the last CU are ignored by the runtime)
_streams = {}


def masked_stream(lo, hi, budget=None):
    """torch ExternalStream confined to CU-mask bits [lo, hi)"""
    key = (lo, hi)
    if key not in _streams:
        words = (C.c_uint32 * NW)()
        for b in range(lo, hi):
            words[b // 32] |= 1 << (b % 32)
        h = C.c_void_p()
        L.call("csbsr_debug_stream_create_cu_mask", C.byref(h), words, NW)
        _streams[key] = (torch.cuda.ExternalStream(h.value, device="cuda:0"), h)
    st, h = _streams[key]
    L.call("csbsr_debug_stream_set_cu_budget", h, (hi - lo) if budget is None else budget)
    return st


def trace():
    print("== A. CU-mask bit -> (XCD, CU): distinct CUs a 4096-workgroup spinning grid ran on")
    out = torch.zeros(2 * 4096, dtype=torch.int32, device="cuda")
    res = {}
    for lo, hi in ((0, 256), (0, 8), (0, 64), (0, 128), (128, 256), (0, 192), (192, 256), (0, 320), (256, 320)):
        try:
            st = masked_stream(lo, hi)
        except L.CsbsrHipError as e:
            print(f"  bits [{lo},{hi}): {e}")
            continue
        out.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            L.call("csbsr_debug_cu_trace", C.c_void_p(out.data_ptr()), 4096, 400000, C.c_void_p(st.cuda_stream))
        st.synchronize()
        o = out.cpu().view(-1, 2)
        hw, xcc = o[:, 0] & 0xFFFF, o[:, 1] & 0xF
        cu = (hw >> 8) & 0xFF                      # cu 11:8, sh 12, se 15:13
        ids = set((int(x), int(c)) for x, c in zip(xcc, cu))
        per = [sum(1 for (x, c) in ids if x == k) for k in range(8)]
        print(f"  bits [{lo:3d},{hi:3d}): {len(ids):3d} distinct CUs, per XCD {per}")
        res[f"{lo}-{hi}"] = {"distinct": len(ids), "per_xcd": per}
    return res


def make_kernels(eng, N=4):
    """the backward's kernel classes at the shapes of one KBPN micro-batch of 4 (config 2) -- name: (callable, GFLOP, GB)"""
    dev = "cuda"
    ks = {}

    def conv(name, H, W, cin, cout, k, s, p, tr=False, act=L.ACT_LRELU):
        wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
        params = {"l.weight": torch.randn(wshape, device=dev) / (cin * k * k) ** 0.5, "l.bias": torch.zeros(cout, device=dev)}
        c = Conv(eng, "l", params, k, s, p, 1, transposed=tr, bias=not tr, act=act, slope=0.1)
        x = FM(torch.randn(N, H, W, pad8(cin), device=dev, dtype=torch.float16), cin)
        OH, OW = c.out_size(H, W)
        y = eng.new(N, OH, OW, cout)
        dy = FM(torch.randn(N, OH, OW, pad8(cout), device=dev, dtype=torch.float16), cout)
        dx = eng.new(N, H, W, cin)
        taps = k * k if not tr else ((k + s - 1) // s) ** 2
        gf = 2.0 * N * OH * OW * cout * cin * taps / 1e9
        gb = 2.0 * N * (H * W * cin + OH * OW * cout) / 1e9
        return c, x, y, dy, dx, gf, gb

    c, x, y, dy, dx, gf, gb = conv("sft", 448, 448, 825, 384, 3, 1, 1)
    ks["wgrad_sft825_384 <256,256>"] = (lambda c=c, dy=dy, x=x: c._bwd_weights_impl(dy, x), gf, gb)
    ks["x3_sft825_384 fwd"] = (lambda c=c, x=x, y=y: c.fwd(x, out=y), gf, gb)
    c2, x2, y2, dy2, dx2, gf2, gb2 = conv("c8s4", 1792, 1792, 128, 128, 8, 4, 2)
    ks["wgrad_conv8s4 <128,512>"] = (lambda c=c2, dy=dy2, x=x2: c._bwd_weights_impl(dy, x), gf2, gb2)
    ks["x3<2>_conv8s4 fwd"] = (lambda c=c2, x=x2, y=y2: c.fwd(x, out=y), gf2, gb2)
    c3, x3, y3, dy3, dx3, gf3, gb3 = conv("d8s4", 448, 448, 128, 128, 8, 4, 2, tr=True, act=L.ACT_NONE)
    ks["tp_deconv8s4 fwd"] = (lambda c=c3, x=x3, y=y3: c.fwd(x, out=y), gf3, gb3)
    c4, x4, y4, dy4, dx4, gf4, gb4 = conv("hr32", 1792, 1792, 32, 32, 3, 1, 1)
    ks["conv_hr 32->32 fwd"] = (lambda c=c4, x=x4, y=y4: c.fwd(x, out=y), gf4, gb4)
    ks["wgrad_hr 32->32"] = (lambda c=c4, dy=dy4, x=x4: c._bwd_weights_impl(dy, x), gf4, gb4)
    # epilogue backward of a residual PReLU layer at HR (up_conv3): reads dOut, out, res; writes dPre
    d = FM(torch.randn(N, 1792, 1792, 128, device=dev, dtype=torch.float16), 128)
    o = FM(torch.randn(N, 1792, 1792, 128, device=dev, dtype=torch.float16), 128)
    r = FM(torch.randn(N, 1792, 1792, 128, device=dev, dtype=torch.float16), 128)
    dp = eng.new(N, 1792, 1792, 128)
    pr = torch.tensor([0.1], device=dev)
    dpr = torch.zeros(1, device=dev)
    ks["epilogue_bwd HR128"] = (lambda: eng.epilogue_bwd(d, out=o, act=L.ACT_PRELU, prelu=pr, res=r, res_mode=L.RES_ADD, dpre=dp, dprelu=dpr, creal=128),
                                0.0, 4 * 2.0 * N * 1792 * 1792 * 128 / 1e9)
    # BatchNorm backward of a detector layer (64 channels at 896^2, B = 8)
    cbn = 64
    P = {"bn.weight": torch.ones(cbn, device=dev), "bn.bias": torch.zeros(cbn, device=dev), "bn.running_mean": torch.zeros(cbn, device=dev),
         "bn.running_var": torch.ones(cbn, device=dev), "bn.num_batches_tracked": torch.zeros((), dtype=torch.long, device=dev)}
    bn = BatchNorm(eng, "bn", P, cbn)
    xb = FM(torch.randn(8, 896, 896, cbn, device=dev, dtype=torch.float16), cbn)
    db = FM(torch.randn(8, 896, 896, cbn, device=dev, dtype=torch.float16), cbn)
    mean, iv = torch.zeros(cbn, device=dev), torch.ones(cbn, device=dev)
    ks["bn_backward 64ch 896^2 B8"] = (lambda: bn.backward(db, xb, mean, iv, act=L.ACT_RELU), 0.0, 5 * 2.0 * 8 * 896 * 896 * cbn / 1e9)
    # bilinear x2 up-sampling backward-sized pass (decoder): forward here
    xl = FM(torch.randn(8, 448, 448, 256, device=dev, dtype=torch.float16), 256)
    yl = eng.new(8, 896, 896, 256)
    ks["bilinear x2 256ch"] = (lambda: eng.bilinear(xl, 896, 896, True, out=yl), 0.0, 2.0 * 8 * (448 * 448 + 896 * 896) * 256 / 1e9)
    return ks


def time_on(st, fn, iters):
    with torch.cuda.stream(st):
        fn()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters):
            fn()
        e1.record(st)
        st.synchronize()
    return e0.elapsed_time(e1) / iters


def scale(eng, ks):
    print("== B. one kernel confined to n CUs (prefix mask, persistent grids sized to n): ms per launch, rate relative to the whole chip")
    cus = (256, 224, 192, 160, 128, 96, 64, 32)
    res = {}
    print(f"  {'kernel':30s}" + "".join(f"{n:>14d}" for n in cus))
    for name, (fn, gf, gb) in ks.items():
        row = []
        for n in cus:
            row.append(time_on(masked_stream(0, n), fn, 6))
        res[name] = dict(zip(map(str, cus), row))
        unit = (lambda ms: f"{gf / ms:5.0f}TF") if gf > 500 else (lambda ms: f"{gb / ms * 1e3 / 1e3:4.2f}TB")
        print(f"  {name:30s}" + "".join(f"{ms:7.2f} {unit(ms)}" for ms in row))
        print(f"  {'':30s}" + "".join(f"{row[0] / ms:13.2f}x" for ms in row))
    return res


def pair(eng, ks, ta):
    print("== C. a weight gradient on CU bits [0, a) NEXT TO an HBM-bound link on bits [a, 256): wall time of (nA x A, nB x B) concurrent vs back to back on the whole chip")
    full = masked_stream(0, 256)
    res = {}
    A_names = ("wgrad_sft825_384 <256,256>", "wgrad_conv8s4 <128,512>")
    B_names = ("epilogue_bwd HR128", "bn_backward 64ch 896^2 B8", "conv_hr 32->32 fwd", "bilinear x2 256ch", "tp_deconv8s4 fwd", "x3_sft825_384 fwd")
    for an in A_names:
        fa = ks[an][0]
        for bn_ in B_names:
            fb = ks[bn_][0]
            tA, tB = ta[an]["256"], ta[bn_]["256"]
            nA = 6
            nB = max(1, round(nA * tA / tB))             # equal time on the whole chip
            # back to back on the unmasked stream
            def serial():
                for _ in range(nA):
                    fa()
                for _ in range(nB):
                    fb()
            t_ser = time_on(full, serial, 2)
            line = f"  {an:28s} x{nA} | {bn_:28s} x{nB}: serial {t_ser:7.2f} ms;"
            rr = {"serial_ms": t_ser, "nA": nA, "nB": nB}
            for a in (128, 160, 192, 208, 224):
                sa, sb = masked_stream(0, a), masked_stream(a, 256)
                best = None
                for rep in range(2):
                    torch.cuda.synchronize()
                    cur = torch.cuda.current_stream()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(cur)
                    sa.wait_event(e0)
                    sb.wait_event(e0)
                    with torch.cuda.stream(sa):
                        for _ in range(nA):
                            fa()
                        ea.record(sa)
                    with torch.cuda.stream(sb):
                        for _ in range(nB):
                            fb()
                        eb.record(sb)
                    cur.wait_event(ea)
                    cur.wait_event(eb)
                    e1.record(cur)
                    torch.cuda.synchronize()
                    t = (e0.elapsed_time(e1), e0.elapsed_time(ea), e0.elapsed_time(eb))
                    if best is None or t[0] < best[0]:
                        best = t
                rr[str(a)] = {"wall_ms": best[0], "A_done_ms": best[1], "B_done_ms": best[2]}
                line += f"  a={a}: {best[0]:6.2f} ({best[0] / t_ser:4.2f}x; A {best[1]:5.1f} B {best[2]:5.1f})"
            print(line)
            res[f"{an} || {bn_}"] = rr
    return res


if __name__ == "__main__":
    what = sys.argv[1:] or ["trace", "scale", "pair"]
    eng = Engine()
        out = {}
    if "trace" in what:
        out["trace"] = trace()
    if "scale" in what or "pair" in what:
        ks = make_kernels(eng)
        out["scale"] = scale(eng, ks)
        if "pair" in what:
            out["pair"] = pair(eng, ks, out["scale"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r05_overlap_pair.json"), "w") as f:
        json.dump(out, f, indent=1)
