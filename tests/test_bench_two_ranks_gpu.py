"""The N > 1 path of bench.py on GPU tensors: two ranks on ONE device over gloo (RCCL refuses two ranks per GPU; the collective
calls, the side-stream bucket launch, the barrier / max-over-ranks timing and the rank-0 JSON line are the same code as with nccl)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """a port nobody listens on right now (a fixed rendezvous port that a previous run left in TIME_WAIT hangs the store's bind)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_bench_two_ranks_one_device():
    env = dict(os.environ, CSBSR_DIST_BACKEND="gloo", CSBSR_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", CSBSR_BENCH_WATCHDOG="300")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2", "--lr-size", "64", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["steps"] == 2 and "roofline" in d
    assert d["loss"] == d["loss"]                       # not NaN
    # the KBPN residency schedule is agreed over the ranks (one MIN all-reduce at the first forward) and reported with every rank's peak
    sch, red = d["schedule"], d["reducer"]
    assert sch["same_on_every_rank"] is True and len(sch["peak_mem_gb_per_rank"]) == 2 and sch["n_resident"] >= 1
    assert len(sch["step_ms_per_rank"]) == 2 and max(sch["step_ms_per_rank"]) <= d["ms_per_step"] * 1.001      # every rank's own time; the line reports the slowest
    assert red["agreements"] == 1 and red["exposed_all_reduce_ms_per_step"] >= 0.0


def test_bench_plain_command_starts_its_own_ranks():
    """Round 6 (review item 2): `python bench.py --gpus 2` with NO launcher on the command line starts the two ranks itself (a
    torch.distributed.run child, before the parent touches the GPU) and relays rank 0's line -- one command drives all GPUs, like the
    reference's train.py:105-112.  Two ranks on one device over gloo, as above."""
    env = dict(os.environ, CSBSR_DIST_BACKEND="gloo", CSBSR_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", CSBSR_BENCH_WATCHDOG="300")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2", "--lr-size", "64", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2" and d["value"] > 0
    assert d["schedule"]["same_on_every_rank"] is True and len(d["schedule"]["step_ms_per_rank"]) == 2


def test_bench_rccl_backend_one_rank():
    """The same N > 1 code path with the REAL backend: `nccl` (= RCCL on ROCm) with world size 1 on the one GPU of the test box.
    CSBSR_FORCE_DIST=1 makes the reducer and the parameter broadcast ISSUE their collectives in a one-rank group (they return early
    otherwise): process-group init with device_id, one broadcast per (parameters | buffers, dtype) group, per step the six flat-bucket all-reduces
    (segmentation net, KBPN stages 4..1, KBPN head) on the side stream launched from inside the backward, barrier, MAX all-reduce of
    the step time.  What a one-GPU box cannot show is only the transfer between GPUs."""
    env = dict(os.environ, CSBSR_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", CSBSR_BENCH_WATCHDOG="300")
    env.pop("CSBSR_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "2", "--lr-size", "64", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-h2d-leg", "--no-other-precision-leg", "--no-kernel-timing"]      # (no extra legs: exactly 1 + 2 steps)
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["loss"] == d["loss"]
    r = d["reducer"]
    assert r["backend"] == "nccl" and r["steps"] == 3                       # 1 warmup + 2 timed steps
    assert r["all_reduces"] == 6 * r["steps"] and r["on_side_stream"] == r["all_reduces"], r
    assert r["bytes"] == r["steps"] * 4 * 89_249_704, r      # every trainable parameter of KBPN x4 + PSPNet, fp32, once per step
    assert 1 <= r["broadcasts"] <= 8, r                     # flat groups (fp32 weights, fp32 running statistics, int64 counters), not 410 tensors


def test_forced_rccl_all_reduce_leaves_gradients_bit_identical():
    """One-rank `nccl` group in this process: a step with the forced reducer (six in-place all-reduces of the flat gradient buckets on the
    side stream, launched from inside the backward) must give exactly the gradients of the undistributed step -- the path is
    bit-reproducible (tests/test_determinism_gpu.py), so ANY corruption by the exchange (a bucket reduced before its last writer, a
    missing stream wait, a wrong view) shows as a bit difference."""
    import torch
    import torch.distributed as dist
    from golden_utils import load_golden
    from test_joint_gpu import build_model
    from csbsr_amd.parallel import GradBucketReducer
    g = load_golden("e2e_pspnet_it40000")
    t = lambda k: torch.from_numpy(g[k])

    def step(m):
        for p in m.parameters():
            p.grad = None
        seg_l, sr_l, _, _, _ = m(40000, t("x"), sr_targets=t("hr"), segment_targets=t("mask"), kernel_targets=t("kernel"))
        (0.7 * sr_l.mean() + 0.3 * seg_l.mean()).backward()
        torch.cuda.synchronize()
        return {k: v.grad.detach().clone() for k, v in m._named_full() if isinstance(v, torch.nn.Parameter) and v.grad is not None}
    m, _ = build_model(g, micro_batch=1)        # two micro-batches: the stage buckets are launched under the LAST one's backward
    m.max_resident = 8
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    base = step(m)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        m.load_state_dict(sd0)
        m.reducer = GradBucketReducer(side_stream=torch.cuda.Stream(torch.device("cuda:0")), force=True)
        got = step(m)
        st = dict(m.reducer.stats)
    finally:
        m.reducer = None
        dist.destroy_process_group()
    assert st["all_reduces"] == 6 and st["on_side_stream"] == 6 and st["steps"] == 1, st
    assert set(got) == set(base) and len(base) == 290
    bad = [k for k in base if not torch.equal(base[k], got[k])]
    assert not bad, bad[:5]
