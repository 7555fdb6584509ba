"""Turn gpurun_out/<tag>/ (written by scripts/collect_profiles.sh on the GPU box) into the committed evidence under profiles/
(usage: summarise_profiles.py [dir under gpurun_out] [tag], default r02 r02):

  profiles/<tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `python bench.py` (default flags)
  profiles/<tag>_bench.json               the JSON line of the same command
  profiles/<tag>_pmc_traffic.json         per-kernel HBM bytes per launch from the two --pmc passes (FETCH_SIZE, WRITE_SIZE), at --batch 4 (one micro-batch of 4, as in the bench)
  profiles/<tag>_pmc_mfma.json            per-kernel MFMA-pipe utilisation from the SQ_VALU_MFMA_BUSY_CYCLES pass (same command)

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950
FETCH_SIZE counts 128-byte requests as 64 bytes for wide coalesced reads, so it is doubled; WRITE_SIZE is taken as is (uncalibrated).
"""
import csv
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r03")
DST = os.environ.get("CSBSR_PROFILES_DST") or os.path.join(ROOT, "profiles")      # (on the GPU box: a directory under gpurun_out/, the only one that travels back)
TAG = sys.argv[2] if len(sys.argv) > 2 else "r03"


sys.path.insert(0, ROOT)
from csbsr_amd.utils.kernel_names import canon      # one name per rocprofv3 row, the one bench.py prints  # noqa: E402


def pmc(path, counter):
    agg = defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = agg[canon(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return agg


os.makedirs(DST, exist_ok=True)
shutil.copy(os.path.join(SRC, "stats", "bench_kernel_stats.csv"), os.path.join(DST, f"{TAG}_bench_kernel_stats.csv"))
line = [l for l in open(os.path.join(SRC, "bench.json")) if l.startswith("{")][-1]
json.dump(json.loads(line), open(os.path.join(DST, f"{TAG}_bench.json"), "w"), indent=1)
if os.path.exists(os.path.join(SRC, "layers.json")):      # bench.py --dump-layers of the same run: scripts/roof_ledger.py reads it
    shutil.copy(os.path.join(SRC, "layers.json"), os.path.join(DST, f"{TAG}_layers.json"))
fetch = pmc(os.path.join(SRC, "pmc_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
write = pmc(os.path.join(SRC, "pmc_write", "write_counter_collection.csv"), "WRITE_SIZE")
out = {"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py --batch 4 --steps 1 --warmup 1 --no-cpu-baseline "
                  "--no-kernel-timing --no-h2d-leg --no-other-precision-leg (two separate passes; a third with TCC_HIT_sum TCC_MISS_sum)",
       "units": "bytes per launch; FETCH_SIZE(KiB) x 1024 x 2 (gfx950 correction), WRITE_SIZE(KiB) x 1024.  CAVEAT (MI355X_MICROARCH.md, HBM "
                "section): FETCH_SIZE derives from the L2's fabric-side read requests, so reads served by the 256 MB Infinity Cache are "
                "counted too -- it is an upper bound on HBM reads; l2_hit_rate = TCC_HIT / (TCC_HIT + TCC_MISS) of the same launches says how "
                "much of a kernel's traffic never left its XCD", "kernels": {}}
l2p = os.path.join(SRC, "pmc_l2", "l2_counter_collection.csv")
l2hit = pmc(l2p, "TCC_HIT_sum") if os.path.exists(l2p) else {}
l2miss = pmc(l2p, "TCC_MISS_sum") if os.path.exists(l2p) else {}
for k in sorted(fetch, key=lambda k: -fetch[k][0]):
    f, n = fetch[k]
    w, nw = write.get(k, (0.0, 0))
    if n < 2:
        continue
    fb, wb = f / n * 1024 * 2, (w / nw * 1024 if nw else 0.0)
    out["kernels"][k] = {"launches": n, "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                         "hbm_bytes_per_launch": round(fb + wb)}
    if k in l2hit and (l2hit[k][0] + l2miss.get(k, (0.0, 0))[0]) > 0:
        out["kernels"][k]["l2_hit_rate"] = round(l2hit[k][0] / (l2hit[k][0] + l2miss[k][0]), 4)
json.dump(out, open(os.path.join(DST, f"{TAG}_pmc_traffic.json"), "w"), indent=1)
# ---- MFMA utilisation: SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs (= 32 x the 32x32x16 MFMA instructions, checked
# against SQ_INSTS_MFMA); utilisation = busy / (1024 SIMDs x the kernel's cycles), the kernel's cycles taken from GRBM_GUI_ACTIVE
# (GRBM_GUI_ACTIVE is reported summed over the 8 XCDs, so / 8; per launch the effective clock under the counters comes out at
# 1.8-2.3 GHz, not the 2.4 GHz peak clock the roofline is priced at)
mp = os.path.join(SRC, "pmc_mfma", "mfma_counter_collection.csv")
if os.path.exists(mp):
    busy, gui, wavec, ninst = (pmc(mp, c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_INSTS_MFMA"))
    dur = defaultdict(lambda: [0.0, 0])
    with open(mp) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                a = dur[canon(r["Kernel_Name"])]
                a[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); a[1] += 1
    mo = {"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA -- python3 "
                     "bench.py --batch 4 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-h2d-leg",
          "units": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); clock_ghz = GRBM_GUI_ACTIVE / 8 / launch duration; "
                   "mfma_util_at_2p4ghz = busy cycles / (1024 x 2.4e9 x duration), i.e. against the peak-clock MFMA rate the roofline uses",
          "kernels": {}}
    for k in sorted(busy, key=lambda k: -busy[k][0]):
        b, n = busy[k]
        g = gui[k][0] / 8.0
        d_ns = dur[k][0]
        if n < 2 or g <= 0 or b <= 0:
            continue
        mo["kernels"][k] = {"launches": n, "avg_launch_us": round(d_ns / n / 1e3, 1), "mfma_util": round(b / (1024.0 * g), 4),
                            "clock_ghz": round(g / d_ns, 3), "mfma_util_at_2p4ghz": round(b / (1024.0 * 2.4 * d_ns), 4),
                            "mfma_insts_per_launch": round(ninst[k][0] / n)}
    json.dump(mo, open(os.path.join(DST, f"{TAG}_pmc_mfma.json"), "w"), indent=1)
    for k, v in list(mo["kernels"].items())[:12]:
        print(f"{k:44s} n={v['launches']:5d} {v['avg_launch_us']:9.1f} us  MFMA busy {100*v['mfma_util']:5.1f} %  clock {v['clock_ghz']:.2f} GHz")
tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in out["kernels"].values())
print("total HBM bytes over the profiled run: %.1f GB" % (tot / 1e9))
for k, v in list(out["kernels"].items())[:12]:
    print(f"{k:40s} n={v['launches']:5d} fetch/launch {v['fetch_bytes_per_launch']/1e6:9.1f} MB  write/launch {v['write_bytes_per_launch']/1e6:9.1f} MB")

# ---- one table of current numbers (DESIGN.md section 5 quotes it verbatim): profiles/<tag>_summary.md
bj = json.loads(line)
rows = list(csv.DictReader(open(os.path.join(DST, f"{TAG}_bench_kernel_stats.csv"))))
nsteps = bj["steps"] + bj["warmup"]
tot_ns = sum(float(r["TotalDurationNs"]) for r in rows)
roof = bj.get("roofline") or {}
alg = dict(roof.get("other_kernels", {}))
if roof:
    alg[roof["kernel"]] = {"achieved": roof.get("mfma_tflops", roof.get("achieved")), "executed": roof.get("executed")}
mfma = json.load(open(os.path.join(DST, f"{TAG}_pmc_mfma.json")))["kernels"] if os.path.exists(os.path.join(DST, f"{TAG}_pmc_mfma.json")) else {}
md = [f"<!-- generated by scripts/summarise_profiles.py {os.path.basename(SRC)} {TAG} -->",
      f"**{bj['config']['workload']}** — `{bj['value']}` img/s, {bj['ms_per_step']} ms/step ({bj['detector_precision']}); "
      f"other mode: {(bj.get('other_precision') or {}).get('value')} img/s; peak {bj.get('peak_mem_gb')} GiB; "
      f"kernel time {tot_ns / nsteps / 1e6:.0f} ms/step in {sum(int(r['Calls']) for r in rows) / nsteps:.0f} launches",
      ""]
if bj.get("step_roofline"):
    md.append(f"step roofline: HBM {bj['step_roofline']['hbm_frac']}, MFMA {bj['step_roofline']['mfma_frac']} of peak; "
              f"fabric traffic {tot / 1e9 / 8:.0f} GB per image-step (PMC passes at B = 4: warm-up + one step = 8 image-steps)")
    md.append("")
md += ["| rocprofv3 row | share | launches/step | avg µs | alg. TF/s (executed) | GB fetched+written / launch | L2 hit | MFMA busy @ GHz |", "|---|---|---|---|---|---|---|---|"]
for r in rows[:16]:
    k = canon(r["Name"])
    a = alg.get(k, {})
    t = out["kernels"].get(k, {})
    m = mfma.get(k, {})
    tf = f"{a.get('achieved', '')}" + (f" ({a['executed']})" if a.get("executed") not in (None, a.get("achieved")) else "") if a else ""
    md.append(f"| `{k}` | {float(r['Percentage']):.1f} % | {int(r['Calls']) / nsteps:.0f} | {float(r['AverageNs']) / 1e3:.0f} | {tf} | "
              f"{t.get('hbm_bytes_per_launch', 0) / 1e9:.2f} | {t.get('l2_hit_rate', '')} | "
              + (f"{100 * m['mfma_util']:.0f} % @ {m['clock_ghz']:.2f}" if m else "") + " |")
if roof.get("families"):
    md += ["", "families (same time, summed per kernel source): " + ", ".join(f"`{k}` {100 * v['share_of_step_time']:.1f} %" for k, v in list(roof["families"].items())[:8])]
open(os.path.join(DST, f"{TAG}_summary.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md))
