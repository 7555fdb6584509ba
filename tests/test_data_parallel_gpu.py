"""Data-parallel correctness on the REAL model (row e): two ranks on one device over gloo (RCCL refuses two ranks per GPU; the
reducer, the bucket layout, the side-stream launches and their ordering are the same code as with nccl).

  * iter 1 (SR pretrain: no BatchNorm in the gradient): 2 ranks x B=2 after the all-reduce == 1 rank x B=4, every KBPN gradient;
  * iter 40000 (joint phase, BatchNorm per replica as in the reference's multi-GPU behaviour): every rank's result == the average of
    the two shards' single-process gradients -- segmentation bucket and the per-stage KBPN buckets launched under the backward.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(micro_batch=2):
    from csbsr_amd.config import cfg as base_cfg
    from csbsr_amd.modeling.build_model import JointModelWithLoss
    from csbsr_amd.utils.detfill import deterministic_fill
    m = JointModelWithLoss(base_cfg.clone(), 1000, 0, None)
    deterministic_fill(m.state_dict())
    m.ss_loss_fn.alpha = 0.8
    m.dropout_enabled = False
    m.micro_batch, m.max_resident = micro_batch, 8
    m.train()
    return m


def _grads(m, it, batch):
    x, hr, mask, k = batch
    m.zero_grad(set_to_none=True)
    seg_l, sr_l, *_ = m(it, x, sr_targets=hr, segment_targets=mask, kernel_targets=k)
    loss = sr_l.mean() if it < 30001 else 0.7 * sr_l.mean() + 0.3 * seg_l.mean()
    loss.backward()
    torch.cuda.synchronize()
    return {n: (None if p.grad is None else p.grad.detach().cpu().clone()) for n, p in m._named_full() if isinstance(p, torch.nn.Parameter)}


def _worker(rank, world, port, it, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from csbsr_amd.parallel import GradBucketReducer
    from csbsr_amd.parallel.reducer import broadcast_parameters
    from csbsr_amd.data.synthetic import make_batch
    torch.cuda.set_device(0)
    m = _build(micro_batch=1)            # two micro-batches per rank: the bucket launches ride on the LAST one's backward
    m._runtime()
    broadcast_parameters(m)
    m.reducer = GradBucketReducer(side_stream=torch.cuda.Stream(torch.device("cuda:0")))
    full = make_batch(4, 16, seed=21)
    shard = tuple(t[rank * 2:(rank + 1) * 2] for t in full)
    out[rank] = _grads(m, it, shard)
    dist.destroy_process_group()


@pytest.mark.parametrize("it", [1, 40000])
def test_two_ranks_equal_the_global_batch(it):
    from csbsr_amd.data.synthetic import make_batch
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), it, out), nprocs=world, join=True)
    g0, g1 = out[0], out[1]
    full = make_batch(4, 16, seed=21)
    m = _build(micro_batch=2)
    if it == 1:          # no batch-coupled op in the gradient: the all-reduced result IS the global-batch gradient
        ref = _grads(m, it, full)
    else:                # per-replica BatchNorm: the all-reduced result is the average of the shards' own gradients
        ga = _grads(m, it, tuple(t[0:2] for t in full))
        sd = m.state_dict()
        from csbsr_amd.utils.detfill import deterministic_fill
        deterministic_fill(sd)          # (running statistics moved in the first pass; gradients do not depend on them in train mode)
        gb = _grads(m, it, tuple(t[2:4] for t in full))
        ref = {n: (None if ga[n] is None else 0.5 * (ga[n] + gb[n])) for n in ga}
    worst, n_checked, devs = 0.0, 0, []
    for n, r in ref.items():
        assert (r is None) == (g0[n] is None) == (g1[n] is None), n
        if r is None:
            continue
        assert torch.equal(g0[n], g1[n]), n                        # every replica holds the same reduced gradient
        den = float(r.norm())
        if den < 1e-12:
            continue
        e = float((g0[n] - r).norm()) / den
        worst = max(worst, e)
        n_checked += 1
        if r.numel() > 1:
            devs.append(e)
        if it == 1:
            # fp32 accumulation order differs (pixel splits of the wgrad slabs) and the loss scale differs by 2x between B=2 and B=4
            # (a power of two: fp16 roundings identical short of underflow): fp32-noise level
            assert e < (2e-3 if r.numel() > 1 else 5e-2), (n, e)
    devs = torch.tensor(devs)
    print(f"iter {it}: {n_checked} gradient tensors, relative L2 deviation median {float(devs.median()):.2e} worst {worst:.2e}")
    assert n_checked > (100 if it == 1 else 250)
    if it != 1:
        # joint phase: two runs of the SAME shard already differ at this level -- the order of the fp32 atomics behind the kernel
        # predictor's global-average-pool sums flips fp16 roundings of the SR image, and the random-weight detector amplifies that
        # ~100x (tests/test_wc_parity_gpu.py) -- so the check is that the exchange happened for every bucket (bit-equal replicas,
        # above) and that the result is the shards' average up to that run-to-run noise
        assert float(devs.median()) < 5e-2 and worst < 0.6
