"""Streaming rate of the BatchNorm apply / backward passes and of the fp16 bilinear resize at the detector's largest maps (GPU)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from csbsr_amd import _lib as L
from csbsr_amd.engine import Engine, FM, BatchNorm
eng = Engine()
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for (N, H, W, c) in ((8, 896, 896, 64), (8, 448, 448, 256), (8, 224, 224, 1024)):
    P = {"b.weight": torch.ones(c, device="cuda"), "b.bias": torch.zeros(c, device="cuda"), "b.running_mean": torch.zeros(c, device="cuda"),
         "b.running_var": torch.ones(c, device="cuda"), "b.num_batches_tracked": torch.zeros((), dtype=torch.long, device="cuda")}
    bn = BatchNorm(eng, "b", P, c)
    x = FM(torch.randn(N, H, W, c, device="cuda", dtype=torch.float16), c)
    y = eng.new(N, H, W, c)
    mean, invstd = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    gb = x.t.numel() * 2 / 1e9
    ms = t(lambda: bn.apply(x, mean, invstd, act=L.ACT_RELU, out=y))
    print(f"bn_apply [{N},{H},{W},{c}] {ms:.3f} ms  {2 * gb / ms:.2f} TB/s (r+w)")
    xs, ys = eng.new(N, H, W, c, split=True), eng.new(N, H, W, c, split=True)      # the detector precision mode: hi + lo planes in and out
    xs.t.normal_()
    ms = t(lambda: bn.apply(xs, mean, invstd, act=L.ACT_RELU, out=ys))
    print(f"bn_apply split [{N},{H},{W},{c}] {ms:.3f} ms  {4 * gb / ms:.2f} TB/s (r+w, two planes each)")
    ms = t(lambda: bn.apply(xs, mean, invstd, act=L.ACT_RELU, res=ys, out=ys))
    print(f"bn_apply split + residual {ms:.3f} ms  {6 * gb / ms:.2f} TB/s")
    del xs, ys
    dy = FM(torch.randn(N, H, W, c, device="cuda", dtype=torch.float16), c)
    ms = t(lambda: bn.backward(dy, x, mean, invstd, act=L.ACT_RELU))
    print(f"bn_backward (reduce + apply) {ms:.3f} ms  {5 * gb / ms:.2f} TB/s (2 x (dy, x) reads + 1 write)")
for (N, H, W, c, OH, OW) in ((8, 896, 896, 64, 1792, 1792), (8, 448, 448, 256, 896, 896)):
    x = FM(torch.randn(N, H, W, c, device="cuda", dtype=torch.float16), c)
    y = eng.new(N, OH, OW, c)
    ms = t(lambda: eng.bilinear(x, OH, OW, True, out=y))
    print(f"bilinear [{N},{H},{W},{c}] -> {OH}: {ms:.3f} ms  {(x.t.numel() + y.t.numel()) * 2 / 1e9 / ms:.2f} TB/s")
