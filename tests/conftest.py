import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle runs inside many tests.  On a many-core host (the GPU box) torch's default intra-op pool -- one thread per core --
    # makes its small convolutions slower, not faster (bench.py measured 315 s against 1.3 s for one LR-32 step with 256 vs 16 threads)
    import torch
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
