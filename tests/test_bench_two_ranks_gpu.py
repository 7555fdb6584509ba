"""The N > 1 path of bench.py on GPU tensors: two ranks on ONE device over gloo (RCCL refuses two ranks per GPU; the collective
calls, the side-stream bucket launch, the barrier / max-over-ranks timing and the rank-0 JSON line are the same code as with nccl)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_device():
    env = dict(os.environ, CSBSR_DIST_BACKEND="gloo", CSBSR_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2", "--lr-size", "64", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["steps"] == 2 and "roofline" in d
    assert d["loss"] == d["loss"]                       # not NaN


def test_bench_rccl_backend_one_rank():
    """The same N > 1 code path with the REAL backend: `nccl` (= RCCL on ROCm) with world size 1 on the one GPU of the test box
    (CSBSR_FORCE_DIST=1): process-group init with device_id, parameter broadcast, flat-bucket all-reduces on the side stream launched
    from inside the backward, barrier, MAX all-reduce of the step time.  What a one-GPU box cannot show is only the transfer itself."""
    env = dict(os.environ, CSBSR_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CSBSR_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "2", "--lr-size", "64", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-h2d-leg"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["loss"] == d["loss"]
