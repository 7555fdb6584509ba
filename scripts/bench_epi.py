"""Micro-benchmark: epilogue_bwd / BN backward with and without their per-channel reductions (atomics contention check)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from csbsr_amd import _lib as L
from csbsr_amd.engine import Engine, FM, BatchNorm, pad8

eng = Engine()
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for (N, H, W, c) in ((1, 448, 448, 128), (1, 1792, 1792, 128), (1, 1792, 1792, 32), (4, 448, 448, 48), (4, 56, 56, 384), (8, 224, 224, 512)):
    d = FM(torch.randn(N, H, W, pad8(c), device="cuda", dtype=torch.float16), c)
    o = FM(torch.randn(N, H, W, pad8(c), device="cuda", dtype=torch.float16), c)
    db = torch.zeros(c, device="cuda"); pr = torch.tensor([0.1], device="cuda"); dp = torch.zeros(1, device="cuda")
    a = t(lambda: eng.epilogue_bwd(d, out=o, act=L.ACT_PRELU, prelu=pr, dpre=d, dbias=db, dprelu=dp, creal=c))
    b = t(lambda: eng.epilogue_bwd(d, out=o, act=L.ACT_PRELU, prelu=pr, dpre=d, creal=c))
    P = {"bn.weight": torch.ones(pad8(c), device="cuda"), "bn.bias": torch.zeros(pad8(c), device="cuda"), "bn.running_mean": torch.zeros(pad8(c), device="cuda"),
         "bn.running_var": torch.ones(pad8(c), device="cuda"), "bn.num_batches_tracked": torch.zeros((), dtype=torch.long, device="cuda")}
    bn = BatchNorm(eng, "bn", P, c)
    mean, iv = torch.zeros(pad8(c), device="cuda"), torch.ones(pad8(c), device="cuda")
    cbn = t(lambda: bn.backward(d, o, mean, iv, act=L.ACT_RELU)) if c % 8 == 0 else float("nan")
    gb = N * H * W * c * 2 / 1e9
    print(f"[{N},{H},{W},{c}] {gb*1e3:7.1f} MB/tensor  epi_bwd with sums {a:7.1f} us  without {b:7.1f} us   bn_backward (reduce+apply) {cbn:7.1f} us")
