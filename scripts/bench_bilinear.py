"""bilinear x2 forward of the decoder's three maps, standalone (GPU): ms and TB/s, split / plain planes, with / without the dropout scale."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from csbsr_amd.engine import Engine, FM
eng = Engine()
for (N, H, W, c) in ((8, 224, 224, 1024), (8, 448, 448, 256), (8, 896, 896, 64)):
    for split in (True, False):
        for drop in (True, False):
            x = eng.new(N, H, W, c, split=split)
            x.t.normal_()
            out = eng.new(N, 2 * H, 2 * W, c, split=split)
            d = torch.ones(N, c, device="cuda") if drop else None
            eng.bilinear(x, 2 * H, 2 * W, False, out=out, drop=d); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): eng.bilinear(x, 2 * H, 2 * W, False, out=out, drop=d)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            gb = N * H * W * c * 2 * 5 * (2 if split else 1) / 1e9
            print(f"{N}x{H}x{W}x{c} split={split} drop={drop}: {ms:.3f} ms  {gb / ms:.2f} TB/s", flush=True)
            del x, out
