"""Fold rocprofv3 --pmc counter_collection.csv files (one directory per pass, prefix given) into per-kernel counter means per launch.
    python scripts/summarise_pmc.py <dir> <pass-dir-prefix>  -> JSON on stdout
Only the library's kernels (names containing conv_ / _kernel from libcsbsr_hip.so) with >= 1 ms of wave cycles are kept."""
import csv, glob, json, os, re, sys
from collections import defaultdict

base, prefix = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sorted(glob.glob(os.path.join(base, prefix + "*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                n = r["Kernel_Name"]
                if "conv_" not in n:
                    continue
                n = re.sub(r"\(.*", "", n)
                a = agg[n][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
out = {}
for n, cs in agg.items():
    row = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
    row["launches_seen"] = max(v[1] for v in cs.values())
    wc = row.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        # fractions of a wave's resident cycles
        for c in ("SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM",
                  "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_VMEM"):
            if c in row:
                row[c + "/WAVE_CYCLES"] = round(row[c] / wc, 4)
    if row.get("SQ_INSTS_MFMA") and row.get("SQ_INSTS_VALU") is not None:
        row["valu_per_mfma"] = round((row["SQ_INSTS_VALU"] - row["SQ_INSTS_MFMA"]) / row["SQ_INSTS_MFMA"], 3)
    if row.get("SQ_INSTS_LDS") and row.get("SQ_LDS_BANK_CONFLICT") is not None and row.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_conflict_frac"] = round(row["SQ_LDS_BANK_CONFLICT"] / row["SQ_LDS_IDX_ACTIVE"], 4)
    out[n] = {k: (round(v, 1) if isinstance(v, float) and abs(v) > 10 else v) for k, v in row.items()}
json.dump(out, sys.stdout, indent=1, sort_keys=True)
