"""Turn gpurun_out/r01/ (written by scripts/collect_profiles.sh on the GPU box) into the committed evidence under profiles/:

  profiles/r01_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `python bench.py` (default flags)
  profiles/r01_bench.json               the JSON line of the same command
  profiles/r01_pmc_traffic.json         per-kernel HBM bytes per launch from the two --pmc passes (FETCH_SIZE, WRITE_SIZE), at --batch 4 (one micro-batch of 4, as in the bench)

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
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r01")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[2] if len(sys.argv) > 2 else "r01"


def canon(name):
    """kernel name as bench.py prints it"""
    m = re.search(r"conv_igemm_glds_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name) or re.search(r"conv_igemm_glds_kernel<(\d+), (\d+), (\d+), (\d+)>", name)
    if m:      # <BM, wave rows, stages, cout tiles>: the single-cout-tile variants keep their three-parameter name
        g = m.groups()
        return "conv_igemm_glds_kernel<%s,%s,%s>" % g[:3] if g[3] == "1" else "conv_igemm_glds_kernel<%s,%s,%s,%s>" % g
    m = re.search(r"conv_igemm_glds_kernelILi(\d+)ELi(\d+)ELi(\d+)E", name) or re.search(r"conv_igemm_glds_kernel<(\d+), (\d+), (\d+)>", name)
    if m:
        return "conv_igemm_glds_kernel<%s,%s,%s>" % m.groups()
    m = re.search(r"conv_igemm_kernelILi(\d+)ELi(\d+)ELi(\d+)E", name) or re.search(r"conv_igemm_kernel<(\d+), (\d+), (\d+)>", name)
    if m:
        return "conv_igemm_kernel<%s,%s,%s>" % m.groups()
    for k in ("conv_wgrad_thin_kernel", "conv_wgrad_kernel", "conv_thin_cout_kernel", "conv_thin_cin_kernel", "epilogue_bwd_kernel",
              "unpack_wgrad_kernel", "bn_bwd_apply_kernel", "bn_bwd_reduce_kernel", "bn_apply_kernel"):
        if k in name:
            return k
    return name[:60]


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
fetch = pmc(os.path.join(SRC, "pmc_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
write = pmc(os.path.join(SRC, "pmc_write", "write_counter_collection.csv"), "WRITE_SIZE")
out = {"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py --batch 4 --steps 1 --warmup 1 --no-cpu-baseline "
                  "--no-kernel-timing (two separate passes)",
       "units": "bytes per launch; FETCH_SIZE(KiB) x 1024 x 2 (gfx950 correction), WRITE_SIZE(KiB) x 1024", "kernels": {}}
for k in sorted(fetch, key=lambda k: -fetch[k][0]):
    f, n = fetch[k]
    w, nw = write.get(k, (0.0, 0))
    if n < 2:
        continue
    fb, wb = f / n * 1024 * 2, (w / nw * 1024 if nw else 0.0)
    out["kernels"][k] = {"launches": n, "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                         "hbm_bytes_per_launch": round(fb + wb)}
json.dump(out, open(os.path.join(DST, f"{TAG}_pmc_traffic.json"), "w"), indent=1)
tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in out["kernels"].values())
print("total HBM bytes over the profiled run: %.1f GB" % (tot / 1e9))
for k, v in list(out["kernels"].items())[:12]:
    print(f"{k:40s} n={v['launches']:5d} fetch/launch {v['fetch_bytes_per_launch']/1e6:9.1f} MB  write/launch {v['write_bytes_per_launch']/1e6:9.1f} MB")
