"""Streaming bandwidth reference points through the C ABI (fill = write only, axpby = 1 read + 1 write)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from csbsr_amd import _lib as L
from csbsr_amd.engine import Engine, FM, _ptr
eng = Engine()
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for N in (1, 4):
    a = eng.new(N, 1792, 1792, 128); b = eng.new(N, 1792, 1792, 128)
    gb = a.t.numel() * 2 / 1e9
    tf = t(lambda: L.call("csbsr_fill_f16", _ptr(a.t), a.npix, a.cp, a.ld, 1.0, eng.stream))
    ta = t(lambda: L.call("csbsr_axpby", a.npix, a.cp, _ptr(a.t), a.ld, 1.0, None, 0, 0.0, _ptr(b.t), b.ld, eng.stream))
    tt = t(lambda: b.t.copy_(a.t))
    tz = t(lambda: a.t.zero_())
    print(f"N={N} {gb:.2f} GB: fill {tf:.0f} us = {gb/tf*1e6/1e3:.2f} TB/s write; axpby copy {ta:.0f} us = {2*gb/ta*1e6/1e3:.2f} TB/s r+w; torch copy_ {tt:.0f} us = {2*gb/tt*1e6/1e3:.2f}; torch zero_ {tz:.0f} us = {gb/tz*1e6/1e3:.2f} TB/s")
