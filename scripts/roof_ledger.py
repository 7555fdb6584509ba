"""Per-row / per-layer roof ledger of one bench.py configuration (round-5 review item 1): for every rocprofv3 kernel row of the step --
and, under the convolution rows, every layer -- the measured time per step next to its time AT THE CALIBRATED ROOF of this part, sorted by
(measured - roof), and the sums, so that "what is the ceiling of this step with these kernels' work" is a table one can re-add.

    python scripts/roof_ledger.py <tag> [layers.json]        ->  profiles/<tag>_ledger.md

Inputs (all committed under profiles/ by scripts/collect_profiles.sh + summarise_profiles.py, same command, same box):
  <tag>_bench.json               the bench line (steps, warm-up, ms per step, batch)
  <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command: calls and total duration per kernel row
  <tag>_pmc_traffic.json         fabric bytes per launch and row (FETCH_SIZE x 2 + WRITE_SIZE, at --batch 4: x 2 per launch for the batch of 8)
  <tag>_layers.json              bench.py --dump-layers: every conv / wgrad launch of the event leg with its layer, shape, row, algorithmic FLOP and bytes

Calibrated roofs (profiles/r05_hipblaslt_calibration.txt): 1.2 PFLOP/s -- what hipBLASLt reaches on this part for large
fp16 GEMMs on random data (0.48 of the 2.5 PF/s dense peak; the clock is power-limited under the matrix pipe) -- and 4.7 TB/s -- what a mixed
read + write stream reaches (0.59 of the 8 TB/s peak).  A row's roof time is max(FLOP / 1.2 PF/s, bytes / 4.7 TB/s): FLOP = the MFMA work
the row EXECUTES (the split-precision detector forward runs three fp16 products per algorithmic product: that is the price of the 1e-3
parity mode, listed in its own column), bytes = the algorithmic bytes of its launches (inputs + outputs + every map a fused epilogue
reads); rows the engine does not log per launch (element-wise, BatchNorm, losses, packs, torch's own kernels) are priced at their measured
FABRIC bytes (an upper bound of their algorithmic bytes, so their gap is a lower bound).
"""
import csv
import json
import os
import re
import sys
from collections import OrderedDict, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from csbsr_amd.utils.kernel_names import canon, conv_row, wgrad_row  # noqa: E402

PF, TBS = 1.2e15, 4.7e12


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    P = os.path.join(ROOT, "profiles")
    bench = json.load(open(os.path.join(P, f"{tag}_bench.json")))
    layers = json.load(open(sys.argv[2] if len(sys.argv) > 2 else os.path.join(P, f"{tag}_layers.json")))
    nsteps_stats = bench["steps"] + bench["warmup"]           # the stats pass runs exactly these (no extra legs)
    ev_steps = (bench.get("with_launch_events") or {}).get("steps") or bench["steps"]
    B = bench["config"]["global_batch"]
    # ---- measured time per row and step
    rows = OrderedDict()
    with open(os.path.join(P, f"{tag}_bench_kernel_stats.csv")) as f:
        for r in csv.DictReader(f):
            k = canon(r["Name"])
            a = rows.setdefault(k, dict(calls=0, ns=0.0))
            a["calls"] += int(r["Calls"]); a["ns"] += float(r["TotalDurationNs"])
    traffic = {}
    tp = os.path.join(P, f"{tag}_pmc_traffic.json")
    if os.path.exists(tp):
        traffic = json.load(open(tp))["kernels"]
    # ---- algorithmic work per row and per (row, layer) from the launch log
    alg = defaultdict(lambda: [0.0, 0.0, 0.0, 0.0, 0])           # flop, executed flop, bytes, event ms, launches  (per step)
    lay = defaultdict(lambda: [0.0, 0.0, 0.0, 0.0, 0])
    for t in layers:
        row = conv_row(t["kernel"]) if t["kind"] == "conv" else wgrad_row(t["kernel"])
        ex = t.get("executed", 1)
        name = re.sub(r"back_projection_stages\.", "S", t["layer"]).replace("sr_model.", "").replace("segmentation_model.", "det.")
        for d, key in ((alg, row), (lay, (row, t["kind"], name, tuple(t["shape"])))):
            a = d[key]
            a[0] += t["flops"] / ev_steps; a[1] += t["flops"] * ex / ev_steps; a[2] += t["bytes"] / ev_steps
            a[3] += t["ms"] / ev_steps; a[4] += 1.0 / ev_steps
    out = []
    tot = dict(ms=0.0, roof=0.0, roof_alg=0.0)
    table = []
    for k, a in rows.items():
        ms = a["ns"] / nsteps_stats / 1e6
        n = a["calls"] / nsteps_stats
        fab = None
        if k in traffic:
            fab = traffic[k]["hbm_bytes_per_launch"] * (B / 4.0) * n       # PMC passes ran at --batch 4: twice the bytes per launch at 8
        if k in alg:
            fl, flx, by = alg[k][0], alg[k][1], alg[k][2]
            roof = max(flx / PF, by / TBS) * 1e3
            roof_alg = max(fl / PF, by / TBS) * 1e3
            src = "log"
        else:
            fl = flx = 0.0
            by = fab if fab is not None else 0.0
            roof = roof_alg = by / TBS * 1e3
            src = "fabric" if fab is not None else "-"
        roof, roof_alg = min(roof, ms), min(roof_alg, ms)            # (a row cannot owe negative time: tiny rows with no byte figure)
        tot["ms"] += ms; tot["roof"] += roof; tot["roof_alg"] += roof_alg
        table.append((ms - roof, k, n, ms, fl, flx, by, fab, roof, roof_alg, src))
    table.sort(key=lambda r: -r[0])
    step_ms = bench["ms_per_step"]
    out.append(f"# Roof ledger `{tag}` -- {bench['config']['workload']}\n")
    out.append(f"`value` {bench['value']} img/s, {step_ms} ms per step (B = {B}); kernel time in the rocprofv3 pass {tot['ms']:.1f} ms per step in "
               f"{sum(a['calls'] for a in rows.values()) / nsteps_stats:.0f} launches.  Roofs: {PF / 1e15:.1f} PFLOP/s (hipBLASLt's rate on this part) and "
               f"{TBS / 1e12:.1f} TB/s (mixed read + write stream), `scripts/roof_ledger.py`.\n")
    out.append("**Sums.** measured {:.1f} ms; at the calibrated roofs {:.1f} ms with the work as executed (split-precision products included) = "
               "**{:.2f} img/s**; {:.1f} ms = {:.2f} img/s if every detector forward product ran once (plain fp16, outside the 1e-3 parity mode).  "
               "The step is at {:.0%} of its executed-work ceiling.\n".format(
                   tot["ms"], tot["roof"], B / (tot["roof"] + (step_ms - tot["ms"])) * 1e3, tot["roof_alg"],
                   B / (tot["roof_alg"] + (step_ms - tot["ms"])) * 1e3, (tot["roof"] + step_ms - tot["ms"]) / step_ms))
    out.append("## Rows, sorted by measured - roof\n")
    out.append("| row | launches / step | measured ms | alg. TFLOP (executed) | alg. GB | fabric GB | roof ms | gap ms | roof / measured | bytes from |")
    out.append("|---|---|---|---|---|---|---|---|---|---|")
    for gap, k, n, ms, fl, flx, by, fab, roof, roof_alg, src in table:
        if ms < 0.3:
            continue
        fls = f"{fl / 1e12:.2f}" + (f" ({flx / 1e12:.2f})" if flx > fl * 1.01 else "") if fl else ""
        out.append(f"| `{k}` | {n:.0f} | {ms:.2f} | {fls} | {by / 1e9:.1f} | {'' if fab is None else f'{fab / 1e9:.1f}'} | {roof:.2f} | {gap:.2f} | {roof / ms:.2f} | {src} |")
    small = [r for r in table if r[3] < 0.3]
    out.append(f"| {len(small)} rows under 0.3 ms | {sum(r[2] for r in small):.0f} | {sum(r[3] for r in small):.2f} | | | | {sum(r[8] for r in small):.2f} | {sum(r[0] for r in small):.2f} | | |")
    out.append(f"| **total** | | **{tot['ms']:.1f}** | | | | **{tot['roof']:.1f}** | **{tot['ms'] - tot['roof']:.1f}** | {tot['roof'] / tot['ms']:.2f} | |\n")
    # ---- per layer
    out.append("## Convolution / weight-gradient launches by layer (HIP events of the bench's event leg), the 60 largest gaps\n")
    out.append("| layer | kind | N, H, W, Cin, Cout, k, s, T | row | launches / step | measured ms | TF/s (executed) | TB/s | roof ms | gap ms |")
    out.append("|---|---|---|---|---|---|---|---|---|---|")
    lt = []
    for (row, kind, name, shape), a in lay.items():
        roof = max(a[1] / PF, a[2] / TBS) * 1e3
        lt.append((a[3] - roof, name, kind, shape, row, a[4], a[3], a[0], a[1], a[2], roof))
    lt.sort(key=lambda r: -r[0])
    for gap, name, kind, shape, row, n, ms, fl, flx, by, roof in lt[:60]:
        tf = f"{fl / ms / 1e9:.0f}" + (f" ({flx / ms / 1e9:.0f})" if flx > fl * 1.01 else "")
        out.append(f"| {name} | {kind} | {', '.join(str(s) for s in shape)} | `{row}` | {n:.0f} | {ms:.2f} | {tf} | {by / ms / 1e9:.2f} | {roof:.2f} | {gap:.2f} |")
    rest = lt[60:]
    out.append(f"| {len(rest)} more | | | | {sum(r[5] for r in rest):.0f} | {sum(r[6] for r in rest):.2f} | | | {sum(r[10] for r in rest):.2f} | {sum(r[0] for r in rest):.2f} |")
    out.append(f"| **total (event leg)** | | | | | **{sum(r[6] for r in lt):.1f}** | | | **{sum(r[10] for r in lt):.1f}** | **{sum(r[0] for r in lt):.1f}** |\n")
    with open(os.path.join(P, f"{tag}_ledger.md"), "w") as f:
        f.write("\n".join(out) + "\n")
    print("\n".join(out[:12]))


if __name__ == "__main__":
    main()
