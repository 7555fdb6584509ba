"""Per-workgroup phase times of the wgrad kernel (needs a -DCSBSR_TS build copied over libcsbsr_hip.so)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
from csbsr_amd import _lib as L
from csbsr_amd.engine import Engine, Conv, FM, pad8
shape = sys.argv[1]; nb = int(sys.argv[2])
shapes = {"deconv8s4": (448, 448, 128, 128, 8, 4, 2, True), "conv8s4": (1792, 1792, 128, 128, 8, 4, 2, False), "c128": (448, 448, 128, 128, 3, 1, 1, False),
          "res512": (224, 224, 512, 512, 3, 1, 1, False), "sft825": (448, 448, 825, 825, 3, 1, 1, False)}
H, W, cin, cout, k, s, p, tr = shapes[shape]
eng = Engine()
wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
params = {"l.weight": torch.randn(wshape, device="cuda") * 0.01, "l.bias": torch.zeros(cout, device="cuda")}
conv = Conv(eng, "l", params, k, s, p, 1, transposed=tr, bias=True, act=L.ACT_LRELU, slope=0.1)
x = FM(torch.randn(nb, H, W, pad8(cin), device="cuda", dtype=torch.float16), cin)
OH, OW = conv.out_size(H, W)
dy = FM(torch.randn(nb, OH, OW, pad8(cout), device="cuda", dtype=torch.float16), cout)
for _ in range(3): conv.bwd_weights(dy, x)
torch.cuda.synchronize()
lib = L.load()
n = 65536
buf = np.zeros(n * 8, dtype=np.uint64)
lib.csbsr_debug_read_wts.argtypes = [ctypes.c_void_p, ctypes.c_long]
lib.csbsr_debug_read_wts(buf.ctypes.data, n * 8)
t = buf.reshape(n, 8).astype(np.int64)
t = t[t.sum(1) > 0] * 10.0
print(shape, "wgrad N", nb, "workgroups", len(t), "; ns per workgroup, summed over its steps")
names = ["prologue", "first issue", "barrier A (+MFMA tail of others)", "wait loads + LDS write", "barrier B", "issue next loads", "ds_read + MFMA", "epilogue"]
tot = t.sum(1).mean()
for i, nm in enumerate(names):
    print(f"  {nm:34s} mean {t[:, i].mean():9.0f} ns  {100 * t[:, i].mean() / tot:5.1f} %")
print(f"  total {tot:9.0f} ns")
