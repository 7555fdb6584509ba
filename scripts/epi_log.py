"""Which stand-alone epilogue-backward passes does a step still run?  Wraps Engine.epilogue_bwd with HIP events and runs bench.py.
    python scripts/epi_log.py [bench flags]        (GPU)  -> per call site: launches, ms and tensor bytes per step"""
import collections
import os
import runpy
import sys
import traceback

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
from csbsr_amd import engine as E

LOG = []
orig = E.Engine.epilogue_bwd


def wrapped(self, dout, out=None, **kw):
    fr = traceback.extract_stack(limit=4)
    site = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr[:-1]))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = orig(self, dout, out, **kw)
    b.record()
    nt = 1 + sum(kw.get(k) is not None for k in ("res", "res2", "dpre", "dres", "dres2")) + (out is not None)
    nt += sum(bool(kw.get(k)) for k in ("dres_acc", "dres2_acc"))
    LOG.append((site, (dout.N, dout.H, dout.W, dout.cp), nt * dout.npix * dout.cp * 2, a, b))
    return r


E.Engine.epilogue_bwd = wrapped
steps, warm = 2, 1
sys.argv = ["bench.py", "--steps", str(steps), "--warmup", str(warm), "--no-other-precision-leg", "--no-h2d-leg", "--no-cpu-baseline",
            "--no-kernel-timing"] + sys.argv[1:]
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for site, shape, nbytes, a, b in LOG:
    v = agg[(site, shape)]
    v[0] += 1; v[1] += a.elapsed_time(b); v[2] += nbytes
n = steps + warm
tot = sum(v[1] for v in agg.values()) / n
print(f"epilogue_bwd: {len(LOG) / n:.0f} launches, {tot:.1f} ms, {sum(v[2] for v in agg.values()) / n / 1e9:.1f} GB per step")
for (site, shape), v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[1] / n:7.2f} ms {v[2] / n / 1e9:7.2f} GB n={v[0] / n:5.1f} {str(shape):24s} {site}")
