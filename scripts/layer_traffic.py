"""Per-launch HBM fetch vs algorithmic bytes: joins bench.py --dump-layers (launch order of the conv / wgrad calls of one step)
with the per-dispatch rows of the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command (scripts/pmc_only.sh)."""
import csv, json, sys, os, re
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
layers = json.load(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "layers.json")))
src = os.path.join(ROOT, "gpurun_out", "r01")

def rows(path, counter):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if r["Counter_Name"] == counter and re.search(r"conv_igemm|conv_thin|conv_wgrad", n):
                out.append((int(r["Dispatch_Id"]), n, float(r["Counter_Value"]) * 1024, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out.sort()
    return out
fe = rows(os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
wr = rows(os.path.join(src, "pmc_write", "write_counter_collection.csv"), "WRITE_SIZE")
n = len(layers)
print(len(fe), len(wr), n)
fe, wr = fe[-n:], wr[-n:]
agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0, ""])
for L, f, w in zip(layers, fe, wr):
    key = (L["kind"], L["layer"].split(".")[-1] if False else re.sub(r"\d+", "#", L["layer"]), tuple(L["shape"]))
    a = agg[key]
    a[0] += 1; a[1] += L["bytes"]; a[2] += 2 * f[2]; a[3] += w[2]; a[4] += L["ms"]; a[5] = f[1][:40]
tot = sorted(agg.items(), key=lambda kv: -(kv[1][2] + kv[1][3]))
print(f"{'kind':6s} {'layer':34s} {'shape':38s} {'n':>3s} {'alg GB':>8s} {'fetch GB':>9s} {'write GB':>9s} {'ratio':>6s} {'ms':>8s}")
for (kind, name, shape), a in tot[:45]:
    print(f"{kind:6s} {name[-34:]:34s} {str(shape):38s} {a[0]:3d} {a[1]/1e9:8.1f} {a[2]/1e9:9.1f} {a[3]/1e9:9.1f} {(a[2]+a[3])/max(a[1],1):6.2f} {a[4]:8.1f}")
print("total alg %.0f GB, fetch %.0f GB, write %.0f GB" % (sum(a[1] for a in agg.values())/1e9, sum(a[2] for a in agg.values())/1e9, sum(a[3] for a in agg.values())/1e9))
