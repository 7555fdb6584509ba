"""Per-workgroup phase timestamps of the glds conv kernel (needs a -DCSBSR_TS build: CSBSR_LIB=<that library>)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
from csbsr_amd import _lib as L
from csbsr_amd.engine import Engine, Conv, FM, pad8
shape = sys.argv[1]; nb = int(sys.argv[2]); what = sys.argv[3]
shapes = {"deconv8s4": (448, 448, 128, 128, 8, 4, 2, True), "conv8s4": (1792, 1792, 128, 128, 8, 4, 2, False), "c128": (448, 448, 128, 128, 3, 1, 1, False),
          "gemm1x1": (1792, 1792, 128, 128, 1, 1, 0, False), "hr32": (1792, 1792, 32, 32, 3, 1, 1, False), "hr49": (1792, 1792, 49, 49, 3, 1, 1, False),
          "hr64": (1792, 1792, 64, 64, 3, 1, 1, False), "res512": (224, 224, 512, 512, 3, 1, 1, False), "up1024": (448, 448, 1024, 256, 3, 1, 1, False)}
H, W, cin, cout, k, s, p, tr = shapes[shape]
eng = Engine()
L.load().csbsr_debug_set_conv_x3(0)
wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
params = {"l.weight": torch.randn(wshape, device="cuda") * 0.01, "l.bias": torch.zeros(cout, device="cuda")}
conv = Conv(eng, "l", params, k, s, p, 1, transposed=tr, bias=True, act=L.ACT_LRELU, slope=0.1)
x = FM(torch.randn(nb, H, W, pad8(cin), device="cuda", dtype=torch.float16), cin)
OH, OW = conv.out_size(H, W)
y = eng.new(nb, OH, OW, cout)
dy = FM(torch.randn(nb, OH, OW, pad8(cout), device="cuda", dtype=torch.float16), cout)
dx = eng.new(nb, H, W, cin)
fn = (lambda: conv.fwd(x, out=y)) if what == "fwd" else (lambda: conv.bwd_input(dy, out=dx, in_hw=(H, W)))
for _ in range(3): fn()
torch.cuda.synchronize()
lib = L.load()
n = 262144
buf = np.zeros(n * 8, dtype=np.uint64)
rd = lib.csbsr_debug_read_its if (len(sys.argv) > 4 and sys.argv[4] == "generic") else lib.csbsr_debug_read_ts
rd.argtypes = [ctypes.c_void_p, ctypes.c_long]
rd(buf.ctypes.data, n * 8)
t = buf.reshape(n, 8).astype(np.int64)
live = t[:, 6] > t[:, 0]
t = t[live]
print(shape, what, "N", nb, "workgroups", len(t), "(wall_clock64 ticks are 100 MHz: 10 ns)")
names = ["prologue(decode)", "tapmask+first issue", "first stage landed", "K loop", "stage acc->LDS", "row epilogue"]
d = np.diff(t[:, :7], axis=1) * 10.0   # ns
for i, nm in enumerate(names):
    print(f"  {nm:22s} mean {d[:, i].mean():8.0f} ns   p50 {np.median(d[:, i]):8.0f}   p90 {np.percentile(d[:, i], 90):8.0f}")
print(f"  entry -> decode start (TS7-TS0): mean {(t[:, 7] - t[:, 0]).mean() * 10:8.0f} ns")
print(f"  total                  mean {(t[:, 6] - t[:, 0]).mean() * 10:8.0f} ns ; kernel span {(t[:, 6].max() - t[:, 0].min()) * 10 / 1e6:.3f} ms")
