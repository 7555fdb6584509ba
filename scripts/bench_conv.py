"""Micro-benchmark of one convolution shape through the C ABI (GPU): TF/s of forward / dgrad / wgrad."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from csbsr_amd import _lib as L
from csbsr_amd.engine import Engine, Conv, FM, pad8

def run(name, N, H, W, cin, cout, k, s, p, d=1, tr=False, iters=10, what=("fwd", "fwd_noepi")):
    eng = Engine()
    eng._wg_on = False          # (the timing events below sit on the caller's stream)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    params = {"l.weight": (torch.randn(wshape, device="cuda") / (cin * k * k) ** 0.5) * (0.0 if os.environ.get("BENCH_ZERO") else 1.0), "l.bias": torch.zeros(cout, device="cuda")}
    conv = Conv(eng, "l", params, k, s, p, d, transposed=tr, bias=os.environ.get("BENCH_NOBIAS") is None, act=L.ACT_LRELU, slope=0.1)
    x = FM(torch.randn(N, H, W, pad8(cin), device="cuda", dtype=torch.float16) * (0.0 if os.environ.get("BENCH_ZERO") else 1.0), cin)
    OH, OW = conv.out_size(H, W)
    y = eng.new(N, OH, OW, cout)
    dy = FM(torch.randn(N, OH, OW, pad8(cout), device="cuda", dtype=torch.float16), cout)
    dx = eng.new(N, H, W, cin)
    taps = k * k if not tr else ((k + s - 1) // s) ** 2
    flops = 2.0 * N * OH * OW * cout * cin * taps
    if tr: flops = 2.0 * N * OH * OW * cout * cin * taps
    res = {}
    for w_ in what:
        fn = {"fwd": lambda: conv.fwd(x, out=y), "fwd_noepi": lambda: conv._launch((x,), conv._pack("fwd", 2 if tr else 0, cin, 0, 0, cout, s, p), tr, k, s, p, d, H, W, OH, OW, cout, y, None, None, 0, 0.0, None, None, None, 0, False, None, 0, -12345.0), "dgrad": lambda: conv.bwd_input(dy, out=dx, in_hw=(H, W)), "wgrad": lambda: conv.bwd_weights(dy, x)}[w_]
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        res[w_] = (ms, flops / ms / 1e9)
    print(f"{name:28s} " + "  ".join(f"{k_}: {v[0]:7.3f} ms {v[1]:7.1f} TF/s" for k_, v in res.items()))

if __name__ == "__main__":
    sel = sys.argv[1] if len(sys.argv) > 1 else "all"
    it = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    if len(sys.argv) > 3:
        L.load().csbsr_debug_set_conv_glds(int(sys.argv[3]))
    if os.environ.get("BENCH_NOX3"):          # the wide 3x3 layers on the LDS-ring kernel (as the detector's BatchNorm'd layers run)
        L.load().csbsr_debug_set_conv_x3(0)
    what = tuple(sys.argv[4].split(",")) if len(sys.argv) > 4 else ("fwd",)
    nb = int(sys.argv[5]) if len(sys.argv) > 5 else None
    shapes = {
        "sft825": (1, 448, 448, 825, 825, 3, 1, 1),
        "sft384_825": (1, 448, 448, 384, 825, 3, 1, 1),
        "sft825_384": (1, 448, 448, 825, 384, 3, 1, 1),
        "sft256_697": (1, 448, 448, 256, 697, 3, 1, 1),
        "sft569_128": (1, 448, 448, 569, 128, 3, 1, 1),
        "sft128_569": (1, 448, 448, 128, 569, 3, 1, 1),
        "conv8s4": (1, 1792, 1792, 128, 128, 8, 4, 2),
        "deconv8s4": (1, 448, 448, 128, 128, 8, 4, 2, 1, True),
        "res512": (8, 224, 224, 512, 512, 3, 1, 1),
        "hr32": (1, 1792, 1792, 32, 32, 3, 1, 1),
        "gemm1x1": (1, 1792, 1792, 128, 128, 1, 1, 0),
        "hr49": (1, 1792, 1792, 49, 49, 3, 1, 1),
        "hr32_49": (1, 1792, 1792, 32, 49, 3, 1, 1),
        "hr49_32": (1, 1792, 1792, 49, 32, 3, 1, 1),
        "hr1x1_32_49": (1, 1792, 1792, 32, 49, 1, 1, 0),
        "thin3_49": (1, 1792, 1792, 3, 49, 3, 1, 1),           # fe_SR.0 forward (bias-free: BENCH_NOBIAS=1 reaches the streaming kernel)
        "thin3_128": (1, 1792, 1792, 3, 128, 3, 1, 1),
        "thin3_512": (1, 1792, 1792, 3, 512, 3, 1, 1),
        "thin512_3": (1, 1792, 1792, 512, 3, 3, 1, 1),         # output_conv / kb.sr_reconst of the last stage: the 3-channel image heads (conv_thin_cout)
        "thin128_3": (1, 1792, 1792, 128, 3, 3, 1, 1),
        "conv8s4_small": (1, 448, 448, 128, 128, 8, 4, 2),
        "c128_small": (1, 112, 112, 128, 128, 3, 1, 1),
        "c128": (1, 448, 448, 128, 128, 3, 1, 1),
        "res256": (8, 224, 224, 256, 256, 3, 1, 1),
        "up1024": (1, 448, 448, 1024, 256, 3, 1, 1),
        "up3_64": (1, 1792, 1792, 64, 64, 3, 1, 1),            # detector decoder, 64-cout layers (the 64-cout LDS-DMA tile)
        "up3_128_64": (1, 1792, 1792, 128, 64, 3, 1, 1),
        "up2_256_64": (1, 896, 896, 256, 64, 3, 1, 1),
        "l1_64": (1, 448, 448, 64, 64, 3, 1, 1),
        "bs_scale1": (1, 1792, 1792, 505, 64, 3, 1, 1),        # PSPNet_BlurSkip conv_scale.1 (config 5)
        "bs_conv0": (1, 1792, 1792, 64, 505, 3, 1, 1),         # PSPNet_BlurSkip conv_scale.0 (config 5), folded: 64 feature channels -> 505
        "lz16": (1, 1792, 1792, 16, 128, 3, 1, 1),             # gathered thin dgrads into one stage's slice of the concatenated gradient
        "lz24": (1, 1792, 1792, 24, 128, 3, 1, 1),
        "lz32": (1, 1792, 1792, 32, 128, 3, 1, 1),
        "lz1x1_256": (1, 1792, 1792, 256, 128, 1, 1, 0),
        "lz1x1_384": (1, 1792, 1792, 384, 128, 1, 1, 0),
    }
    for n, sh in shapes.items():
        if sel in ("all", n):
            if nb:
                sh = (nb,) + tuple(sh[1:])
            run(n, *sh, iters=it, what=what)
