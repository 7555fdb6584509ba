"""bench.py -- training imgs/s of the CSBSR joint SR+segmentation hot path on N MI355X GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 8] [--lr-size 448] [--micro-batch 8]
    python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...          (driver, N > 1)

`python bench.py --gpus N` with N > 1 and no launcher on the command line starts the N ranks itself (a `torch.distributed.run` CHILD
process, before this process touches the GPU -- one command drives all GPUs like the reference's train.py:105-112) and relays rank 0's
JSON line and the exit code; a WORLD_SIZE that disagrees with --gpus, or fewer visible devices than ranks, is an error, not a silent 1-GPU run.

One step = one pass of the hot path over one synthetic minibatch already resident in HBM: KBPN (x4) + PSPNet
forward, fused losses, explicit HIP backward, gradient all-reduce over RCCL (N > 1), Adam -- BASELINE.json config 2
(B=8 per GPU, LR 448 -> HR 1792, iter 40000 = joint phase, beta = 0.3); weak scaling (per-GPU batch fixed).
Prints ONE JSON line (rank 0).  See DESIGN.md section "Measurement" for the algorithmic FLOP / byte figures.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

# Algorithmic work per image, fwd+bwd, x4 / LR 448 / PSPNet.  SURVEY.md section 8(d): 108.3 TFLOP / 249 GB as executed by the
# reference.  This build folds the convolutions over spatially constant operands exactly (fe_kernel.0: 4 x 4.996 TFLOP forward;
# the 441 kernel-code input channels of the six SFT conv0's: 6.66 TFLOP forward; x3 for fwd + dgrad + wgrad = 35.0 TFLOP), so
# the reduced figures are the denominator (never divide folded run time into unfolded work).  Bytes: the folded
# fe_kernel.0 conv I/O (37.8 GB) is replaced by the class-filled map (2.5 GB); the SFT code channels were never read from HBM.
# Round 3 folded one layer deeper (the whole fe_kernel branch of a kernel predictor is a [B, 25, 32] bias table):
# fe_kernel.1 (49 -> 49 3x3 at HR, 4 x 0.1388 TFLOP forward) and the second half of fe_cat.0's input (49 -> 32 1x1, 4 x 0.0101) no
# longer run: -0.595 TFLOP forward, x3 = -1.79 TFLOP; bytes: the class-filled fe_kernel.0 map (2.5 GB) is gone again, with it
# fe_kernel.1's conv I/O (98 channel planes at HR x 4 stages x 3 passes = 7.55 GB) and the 49 constant-branch input planes of fe_cat.0
# (3.78 GB).
ALG_TFLOP_PER_IMG_448 = 108.3 - 35.0 - 1.79
ALG_GB_PER_IMG_448 = 249.0 - 37.8 - 7.55 - 3.78
MFMA_PEAK_TFLOPS = 2500.0     # dense fp16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def cpu_baseline(lr=64, seconds_budget=24.0, threads=0, min_steps=3):
    """The oracle (CPU restatement of the reference path, fp32 torch ops) timed on this host's cores on a bounded sample of the
    bench workload: B=2, LR 64 -> HR 256 by default (LR 448 needs ~300 GB of host RAM for the autograd tape;
    --cpu-baseline-lr 112 is BASELINE.md section 2's CPU-runnable size, ~24 s per step), forward + backward, after one tiny untimed step
    that pays the allocator's and the thread pool's warm-up; at least ``min_steps`` timed steps, more until ~the budget; the value is
    quoted from the MEDIAN step (min / median / max reported) and scaled to LR 448 by the pixel ratio (every term of the path is linear
    in pixels)."""
    from oracle import csbsr_oracle as O
    from csbsr_amd.utils.detfill import det_state_dict
    from csbsr_amd.modeling.shapes import joint_state_shapes
    from csbsr_amd.data.synthetic import make_batch
    # intra-op threads: measured on the GPU box's host (256 hardware threads) the oracle's convolutions do not scale past ~16 threads --
    # one LR-32 step took 315 s on the default 256-thread pool against 1.3 s with 16 (oversubscribed OpenMP barriers in the many small
    # grouped / thin convolutions) -- so the pool is capped at 16 unless --cpu-baseline-threads says otherwise; the count used is reported
    ncpu = os.cpu_count() or 1
    cores = min(ncpu, threads if threads > 0 else 16)
    torch.set_num_threads(cores)
    P = det_state_dict(joint_state_shapes())
    for k, v in P.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    cfg = O.PathCfg()
    B = 2

    def one(xb):
        out = O.joint_forward(P, cfg, 40000, *xb, alpha=0.9)
        O.calc_loss(out["segment_loss"], out["sr_loss"], 40000, cfg).backward()
    one(make_batch(B, 16, seed=2))          # untimed warm-up (HR 64)
    batch = make_batch(B, lr, seed=1)
    t0, per = time.time(), []
    while True:
        t1 = time.time()
        one(batch)
        per.append(time.time() - t1)
        if (len(per) >= min_steps and time.time() - t0 > seconds_budget) or len(per) >= 40:
            break
    dt = time.time() - t0
    srt = sorted(per)
    med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    ips = B / med
    scale = (448.0 / lr) ** 2
    return {"value": ips / scale, "unit": "imgs/s", "cores": cores, "kind": "port",
            "steps": len(per), "step_s": {"min": round(srt[0], 2), "median": round(med, 2), "max": round(srt[-1], 2)},
            "cores_note": f"{cores} intra-op threads of the host's {ncpu}: the fp32 torch oracle stops scaling there (a 256-thread pool ran the same step 240x slower); --cpu-baseline-threads overrides",
            "sample": f"oracle fwd+bwd, B={B}, LR {lr}->HR {lr * 4}, {len(per)} timed steps in {dt:.1f}s after a tiny warm-up step (min / median / max "
                      f"{srt[0]:.1f} / {med:.1f} / {srt[-1]:.1f} s per step) = {ips:.4f} img/s at LR {lr} from the median step; "
                      f"divided by (448/{lr})^2 = {scale:.2f} (conv work linear in pixels) to quote it at LR 448"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch")
    ap.add_argument("--lr-size", type=int, default=448)
    ap.add_argument("--micro-batch", type=int, default=8,
                    help="images per KBPN micro-batch (KBPN has no batch-coupled op; the detector always runs the whole batch).  8 = the batch "
                         "as one micro-batch since round 5: 243 GiB peak, 2 %% faster than two of 4 (the persistent tile kernels' last round of "
                         "tiles is fuller, half the launches); a smaller value trades that for memory")
    ap.add_argument("--max-resident", type=int, default=-1, help="micro-batches whose KBPN activations stay resident for backward")
    ap.add_argument("--workload", default="pspnet_x4", choices=("pspnet_x4", "blurskip_x8", "hrnet_x4"),
                    help="pspnet_x4 = BASELINE config 2 (the bench line); blurskip_x8 = config 5 (x8, PSPNet_BlurSkip, w^F; use --lr-size 224 "
                         "--batch 4); hrnet_x4 = config 4 (HRNet-W48 + OCR, beta 0.9; use --batch 4) -- coverage timings, not the headline")
    ap.add_argument("--detector-precision", default="split", choices=("fp16", "split"),
                    help="split (default, the bench line) = hi+lo fp16 detector forward: the mode for which the detector meets 1e-3 against the "
                         "reference (tests/test_wc_parity_gpu.py::test_detector_on_reference_sr); fp16 = plain fp16 storage in the detector too "
                         "(faster, 4e-2 on the random-weight fixtures / 3e-3 on the contractive ones): timed in an extra leg and reported "
                         "beside the headline as other_precision")
    ap.add_argument("--no-other-precision-leg", action="store_true", help="skip the extra (never `value`) leg that re-times the step in the other detector precision mode")
    ap.add_argument("--no-h2d-leg", action="store_true", help="skip the extra (untimed-for-value) leg that re-times the step with the PCIe copy of the batch inside")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-lr", type=int, default=64, help="LR size of the bounded CPU sample (64: three timed B=2 steps fit ~25 s; 112 = BASELINE.md section 2's size, ~24 s per step)")
    ap.add_argument("--cpu-baseline-threads", type=int, default=0, help="intra-op threads of the CPU sample (0 = min(host threads, 16), see cpu_baseline)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of csbsr_amd.optim.Adam (A/B timing)")
    ap.add_argument("--watchdog-s", type=int, default=int(os.environ.get("CSBSR_BENCH_WATCHDOG", "0")),
                    help="dump every thread's Python stack to stderr and exit if the run is still going after this many seconds (0 = off): a hung "
                         "collective in a multi-rank run then says where each rank stands instead of timing out silently")
    ap.add_argument("--dump-layers", default=None, help="write the per-launch conv / wgrad log of the timed region (layer, shape, kernel, ms) as JSON")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if launched and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    # one rank per GPU (the test hook CSBSR_FORCE_DEVICE puts every rank on one device over gloo).  device_count() does not initialise the GPU
    if "CSBSR_FORCE_DEVICE" not in os.environ and torch.cuda.device_count() < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} device(s) visible")
    if args.gpus > 1 and not launched:
        # self-launch: this process has not touched the GPU and never will -- the ranks are CHILDREN (never exec a process that has
        # initialised HIP), their stdout / stderr are inherited so rank 0's JSON line is this command's JSON line
        import socket
        import subprocess
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = str(s_.getsockname()[1])
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=env).returncode)
    if args.watchdog_s > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog_s, exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    # test hooks (one-GPU smoke of the N > 1 path): CSBSR_DIST_BACKEND=gloo CSBSR_FORCE_DEVICE=0 puts every rank on one device
    backend = os.environ.get("CSBSR_DIST_BACKEND", "nccl")
    if "CSBSR_FORCE_DEVICE" in os.environ:
        local = int(os.environ["CSBSR_FORCE_DEVICE"])
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    # test hook: CSBSR_FORCE_DIST=1 runs the N > 1 code path (process group, broadcast, bucket reducer on the side stream, barrier,
    # max-over-ranks timing) with world size 1 -- on a one-GPU box that is the only way to put the RCCL backend itself under the calls
    dist_on = world > 1 or os.environ.get("CSBSR_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from csbsr_amd.config import cfg as base_cfg
    from csbsr_amd.modeling.build_model import JointModelWithLoss
    from csbsr_amd.parallel import GradBucketReducer
    from csbsr_amd.parallel.reducer import broadcast_parameters

    cfg = base_cfg.clone()
    x8 = args.workload == "blurskip_x8"
    if x8:
        cfg.MODEL.SCALE_FACTOR, cfg.MODEL.DETECTOR_TYPE = 8, "PSPNet_BlurSkip"
        cfg.SOLVER.SEG_FAIL_ORIENTED_WEIGHT4SR_AMP, cfg.SOLVER.ORIENTED_WEIGHT_ITER = 1.0, 0
    if args.workload == "hrnet_x4":
        cfg.MODEL.DETECTOR_TYPE, cfg.SOLVER.TASK_LOSS_WEIGHT = "HRNet_OCR", 0.9
    other = args.workload != "pspnet_x4"
    scale = cfg.MODEL.SCALE_FACTOR
    torch.manual_seed(cfg.SEED + rank)      # Dropout2d masks come from torch's CUDA generator, whose initial seed is per process on this build: two runs print the same loss
    model = JointModelWithLoss(cfg, 9000, 40000, None, device=str(dev))
    model.micro_batch = args.micro_batch
    model.detector_precision = args.detector_precision
    model.max_resident = None if args.max_resident < 0 else args.max_resident     # None: as many as the free HBM allows
    model.train()
    rt = model._runtime()
    n_bcast = 0
    if dist_on:
        n_bcast = broadcast_parameters(model)
        model.reducer = GradBucketReducer(side_stream=torch.cuda.Stream(dev))
    # Adam with the reference's hyper-parameters (train.py:91) as one multi-tensor HIP launch (csbsr_amd/optim.py: a torch.optim.Optimizer
    # drop-in, same state keys, agreement with torch.optim.Adam to fp32 rounding); --torch-adam times torch's own foreach kernels instead
    from csbsr_amd.optim import Adam as HipAdam
    Opt = torch.optim.Adam if args.torch_adam else HipAdam
    opt = Opt([p for p in model.parameters() if p.requires_grad], lr=cfg.SOLVER.LR, betas=(0.9, 0.999), eps=1e-8)

    B, lr = args.batch, args.lr_size
    # synthetic minibatch (seed differs per rank: each GPU gets its own shard of the global batch), resident in HBM when the timed region
    # starts.  Round 6: HR textures + crack masks are generated at FULL size on the host (no tiling of a small batch: periodic data), and
    # the degradation -- per-sample anisotropic Gaussian blur + antialiased bicubic down-scaling, SURVEY 8 row f1 -- runs on the device
    # (csbsr_amd.data.degrade.DeviceDegradation), untimed
    from csbsr_amd.data.synthetic import make_hr_mask
    from csbsr_amd.data.degrade import DeviceDegradation
    gen = torch.Generator().manual_seed(1121 + rank)
    hr, mask = make_hr_mask(B, lr * scale, gen)
    deg = DeviceDegradation(scale, device=str(dev), seed=77 + rank)
    x, hr, mask, k, _ = deg(hr, mask, with_sdf=False)
    x = x.clamp_(0, 1)
    torch.cuda.synchronize()
    it = 40000
    beta = cfg.SOLVER.TASK_LOSS_WEIGHT

    host = None

    def step():
        opt.zero_grad(set_to_none=True)
        if host is not None:        # H2D leg only: the batch crosses PCIe inside the step, as the reference's loader hands it over
            xx, hh, mm, kk = (t.to(dev, non_blocking=True) for t in host)
        else:
            xx, hh, mm, kk = x, hr, mask, k
        seg_l, sr_l, seg, sr, kp = model(it, xx, sr_targets=hh, segment_targets=mm, kernel_targets=kk)
        loss = (1 - beta) * sr_l.mean() + beta * seg_l.mean()          # trainer.py:406-438, iter >= 30001
        loss.backward()
        opt.step()
        return float(loss.detach())

    for _ in range(args.warmup):
        step()
    eng = rt["eng"]
    eng.timing = None           # round 6: no per-launch HIP events inside the region `value` is timed over (they run in their own leg below)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = 0.0
    ovf0, bad_loss = model.overflow_steps, []
    for i_ in range(args.steps):
        last = step()
        if not math.isfinite(last):
            bad_loss.append((i_, last))
    torch.cuda.synchronize()
    dt_rank = time.perf_counter() - t0          # this rank alone, before it waits for the others (schedule.step_ms_per_rank)
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    if bad_loss or model.overflow_steps != ovf0:
        # a timed step that produced a non-finite loss, or whose backward overflowed (the optimiser then skips the update: less work than a
        # real step), is not a measurement -- fail loudly instead of printing a throughput
        sys.stderr.write(f"bench.py: INVALID timed region on rank {rank}: non-finite losses {bad_loss}, overflowed steps {model.overflow_steps - ovf0}\n")
        sys.exit(3)
    ms = dt / args.steps * 1e3
    imgs = B * world * args.steps / dt
    peak_main = torch.cuda.max_memory_allocated(dev)          # of the warm-up + timed steps only: the extra legs below have their own peaks
    # ---- roofline leg: the SAME steps once more with a HIP-event pair around every conv / wgrad launch on the engine's stream (the
    # per-kernel launch durations of the roofline block; rocprofv3's kernel statistics of this command must agree, profiles/).  Its wall
    # time is reported as `with_launch_events` and is never `value`
    timing_log, ev_leg = None, None
    if not args.no_kernel_timing:
        n1 = args.steps if args.steps <= 4 else max(4, args.steps // 4)
        eng.timing = []
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n1):
            step()
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        timing_log, eng.timing = eng.timing, None
        ev_leg = {"steps": n1, "ms_per_step": round(dt1 / n1 * 1e3, 1)}
    # the KBPN residency schedule the step ran with (agreed over the ranks: JointModelWithLoss._auto_resident) and every rank's peak
    n_mb = (B + max(1, min(args.micro_batch, B)) - 1) // max(1, min(args.micro_batch, B))
    schedule = {"micro_batches": n_mb, "n_resident": int(getattr(model, "_n_res", 0)), "lean_saves": bool(getattr(model, "_lean", False)),
                "peak_mem_gb_per_rank": [round(peak_main / 2 ** 30, 1)]}
    exposed = model.reducer.exposed_ms()[-args.steps:] if dist_on else []          # of the timed steps (the device is synchronised)
    if dist_on:
        pk = torch.tensor([peak_main / 2 ** 30], device=dev, dtype=torch.float64)
        allpk = [torch.zeros_like(pk) for _ in range(world)]
        dist.all_gather(allpk, pk)
        schedule["peak_mem_gb_per_rank"] = [round(float(t), 1) for t in allpk]
        # every rank's own wall time of the timed region (the reported time is the maximum): a straggler shows as a spread
        mine = torch.tensor([dt_rank / args.steps * 1e3], device=dev, dtype=torch.float64)
        allms = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allms, mine)
        schedule["step_ms_per_rank"] = [round(float(t), 1) for t in allms]
        sc = torch.tensor([schedule["n_resident"], int(schedule["lean_saves"])], device=dev, dtype=torch.int64)
        lo, hi = sc.clone(), sc.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        schedule["same_on_every_rank"] = bool((lo == hi).all())
    # ---- extra leg (never `value`): the same step with the host -> device copy of the batch inside it (SURVEY 8d counts it; the
    # bench contract wants inputs resident).  Pinned host buffers, a few steps.
    h2d = None
    if not args.no_h2d_leg and world == 1:
        host = tuple(t.cpu().pin_memory() for t in (x, hr, mask, k))
        n2 = max(2, min(5, args.steps))
        step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n2):
            step()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        h2d = {"imgs_per_s": round(B * n2 / dt2, 4), "ms_per_step": round(dt2 / n2 * 1e3, 1), "steps": n2,
               "batch_bytes": int(sum(t.numel() * t.element_size() for t in host))}
        host = None

    # ---- extra leg (never `value`): the same step in the OTHER detector precision mode, a few steps, so both numbers come from one run
    other_prec = None
    if not args.no_other_precision_leg and world == 1:
        alt = "fp16" if args.detector_precision == "split" else "split"
        model.detector_precision = alt
        torch.cuda.reset_peak_memory_stats(dev)
        n3 = max(2, min(4, args.steps))
        step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n3):
            step()
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t1
        other_prec = {"detector_precision": alt, "value": round(B * n3 / dt3, 4), "unit": "imgs/s", "ms_per_step": round(dt3 / n3 * 1e3, 1), "steps": n3,
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1)}
        model.detector_precision = args.detector_precision

    roof = None
    if timing_log and args.dump_layers and rank == 0:
        json.dump([{"kind": t[0], "flops": t[1], "bytes": t[2], "ms": t[3].elapsed_time(t[4]), "layer": t[5], "shape": list(t[6]),
                    "kernel": (t[7] if len(t) > 7 else -1), "executed": (t[8] if len(t) > 8 else 1)} for t in timing_log], open(args.dump_layers, "w"))
    if timing_log:
        # one roofline block = ONE kernel = one rocprofv3 ROW (template instance): the row with the largest share of the timed region.
        # csbsr_debug_last_conv_kernel / _wgrad_kernel tag every launch with the instance it dispatched to; csbsr_amd/utils/kernel_names.py
        # gives it the name scripts/summarise_profiles.py gives the same row of the rocprofv3 CSVs.  achieved = algorithmic FLOPs (or bytes)
        # of those launches / their HIP-event time on the engine's stream.
        from csbsr_amd.utils.kernel_names import conv_row, wgrad_row, family
        per = {}
        for t in timing_log:
            key = conv_row(t[7]) if t[0] == "conv" else wgrad_row(t[7] if len(t) > 7 else -1)
            a = per.setdefault(key, [0.0, 0.0, 0.0, 0, 0.0])
            a[0] += t[1]; a[1] += t[2]; a[2] += t[3].elapsed_time(t[4]) * 1e-3; a[3] += 1
            a[4] += t[1] * (t[8] if len(t) > 8 else 1)          # executed MFMA work: the split-precision launches run 2 or 3 K blocks per product
        fam = {}
        for k_, v in per.items():
            f_ = fam.setdefault(family(k_), [0.0, 0.0, 0])
            f_[0] += v[2]; f_[1] += v[0]; f_[2] += v[3]
        # the dominant kernel is chosen over EVERY MFMA kernel of the step, each wgrad variant on its own
        dt_ev = ev_leg["ms_per_step"] * 1e-3 * ev_leg["steps"]          # the wall time the logged launches belong to
        dom = max(per, key=lambda k: per[k][2])
        fl, by, tt, nl, flx = per[dom]
        ach = fl / tt / 1e12
        traffic = None
        # the newest committed PMC summaries (scripts/collect_profiles.sh + summarise_profiles.py; round tag in the file name)
        def newest(kind):
            import glob
            c = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r0[0-9]_pmc_{kind}.json")))
            return c[-1] if c else ""
        tpath = newest("traffic")
        if os.path.exists(tpath) and args.workload == "pspnet_x4":
            try:
                tj = json.load(open(tpath))
                ks = tj.get("kernels", {})
                traffic = ks.get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # MFMA-pipe utilisation in cycles and the clock the kernel ran at, from the committed SQ_VALU_MFMA_BUSY_CYCLES pass
        mfma_pmc = None
        mpath = newest("mfma")
        if os.path.exists(mpath) and args.workload == "pspnet_x4":
            try:
                mk = json.load(open(mpath)).get("kernels", {})
                var = [v for n_, v in mk.items() if n_ == dom]
                if var:
                    wsum = sum(v["launches"] * v["avg_launch_us"] for v in var)
                    mfma_pmc = {"mfma_pipe_busy": round(sum(v["mfma_util"] * v["launches"] * v["avg_launch_us"] for v in var) / wsum, 4),
                                "clock_ghz": round(sum(v["clock_ghz"] * v["launches"] * v["avg_launch_us"] for v in var) / wsum, 3),
                                "source": "profiles/" + os.path.basename(mpath) + " (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, same command at --batch 4)"}
            except Exception:
                mfma_pmc = None
        # which roof bounds the row: HBM when its algorithmic intensity (executed flop per algorithmic byte, fused epilogue operands
        # included) is below the machine balance (peak flop/s / peak byte/s), or when the committed PMC pass shows the MFMA pipe under
        # 35 % busy -- then `achieved` is bytes/s against the HBM peak
        balance = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        hbm_bound = (flx / max(by, 1.0)) < balance or (mfma_pmc is not None and mfma_pmc["mfma_pipe_busy"] < 0.35)
        if hbm_bound:
            ach_b = by / tt / 1e9
            head = {"bound": "hbm", "kernel": dom, "achieved": round(ach_b, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_b / HBM_PEAK_GBS, 4),
                    "mfma_tflops": round(ach, 1), "mfma_frac": round(ach / MFMA_PEAK_TFLOPS, 4)}
        else:
            head = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                    "hbm_gbs": round(by / tt / 1e9, 1)}
        roof = {**head, "traffic": traffic,
                "selection": "largest share of the timed region by rocprofv3 row (template instance); families = the same time summed per kernel source",
                "families": {k_: {"share_of_step_time": round(v[0] / dt_ev, 3), "achieved_tflops": round(v[1] / v[0] / 1e12, 1), "launches": v[2]}
                             for k_, v in sorted(fam.items(), key=lambda kv: -kv[1][0])},
                "traffic_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/" + os.path.basename(tpath) + "; "
                                "FETCH_SIZE doubled per MI355X_MICROARCH.md), same command at --batch 4 (the default run is one KBPN micro-batch of 8: the same per-image work in half the launches)" if traffic else None,
                "mfma_pmc": mfma_pmc, "launches": nl, "avg_launch_ms": round(tt * 1e3 / max(nl, 1), 4),
                "alg_flop_per_launch": round(fl / max(nl, 1) / 1e9, 2), "alg_flop_unit": "GFLOP",
                "executed": round(flx / tt / 1e12, 1), "executed_frac": round(flx / tt / 1e12 / MFMA_PEAK_TFLOPS, 4),
                "executed_note": "MFMA work the kernel actually ran / its time: the split-precision detector launches multiply hi+lo operand pairs "
                                 "(3 K blocks per algorithmic product forward, 2 in the dgrads); 'achieved' and 'frac' count the algorithmic product once",
                "alg_bytes_per_launch": round(by / max(nl, 1)), "share_of_step_time": round(tt / dt_ev, 3),
                "other_kernels": {k: {"achieved": round(v[0] / v[2] / 1e12, 1), "executed": round(v[4] / v[2] / 1e12, 1), "share_of_step_time": round(v[2] / dt_ev, 3), "launches": v[3],
                                      "avg_launch_ms": round(v[2] * 1e3 / v[3], 4)} for k, v in per.items() if k != dom}}
    out = None
    if rank == 0:
        pix = (lr / 448.0) ** 2
        per_img_s = dt / (B * args.steps)
        out = {"metric": "training imgs/s (448->1792 x4, PSPNet)" if not other else f"training imgs/s ({args.workload})", "value": round(imgs, 4), "unit": "imgs/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 1), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "fp16 storage / fp32 accumulate" + (" (detector forward: split fp16 hi+lo)" if args.detector_precision == "split" else ""),
               "data": "synthetic", "detector_precision": args.detector_precision, "other_precision": other_prec,
               "config": {"workload": f"CSBSR KBPN x{scale} + {cfg.MODEL.DETECTOR_TYPE}, beta={beta}, joint phase (iter 40000), per-GPU batch {B}, "
                                      f"LR {lr}x{lr} -> HR {lr * scale}x{lr * scale}, fwd+loss+bwd+Adam; batch resident in HBM (H2D outside `value`, "
                                      f"timed beside it as with_h2d_inside_step); full-size synthetic crack images degraded on the device", "global_batch": B * world,
                          "micro_batch": args.micro_batch, "parallelism": f"dp{world}"},
               "loss": round(last, 5),
               "step_roofline": {"hbm_frac": round(ALG_GB_PER_IMG_448 * pix / per_img_s / HBM_PEAK_GBS, 4),
                                 "mfma_frac": round(ALG_TFLOP_PER_IMG_448 * pix / per_img_s / MFMA_PEAK_TFLOPS, 4),
                                 "note": f"algorithmic work per image at LR 448 after exact constant-operand folding (the fe_kernel branch of every kernel predictor + the SFT code channels): {ALG_TFLOP_PER_IMG_448:.1f} TFLOP / {ALG_GB_PER_IMG_448:.1f} GB (SURVEY.md 8(d) as-executed: 108.3 TFLOP / 249 GB; rounds 1-3 divided by 73.3 TFLOP / 213.7 GB)"},
               "roofline": roof,
               "peak_mem_gb": round(peak_main / 2 ** 30, 1),
               "with_h2d_inside_step": h2d, "with_launch_events": ev_leg}
        out["schedule"] = schedule
        if dist_on:      # what the gradient exchange did (per rank): collectives, how many rode the side stream, payload, and how long the
            # compute stream waited for the exchange at the end of each backward (the part NOT hidden under it), rank 0's view
            out["reducer"] = dict(model.reducer.stats, broadcasts=n_bcast, backend=backend, world=world,
                                  exposed_all_reduce_ms_per_step=round(sum(exposed) / max(len(exposed), 1), 3),
                                  exposed_all_reduce_ms_max=max(exposed, default=0.0))
        if other:
            out["step_roofline"] = None          # the folded-work figures above are config 2's
        if not args.no_cpu_baseline and world == 1 and not other:
            out["cpu_baseline"] = cpu_baseline(args.cpu_baseline_lr, threads=args.cpu_baseline_threads)
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
