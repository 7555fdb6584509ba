#!/bin/bash
# FETCH_SIZE of the 8x8 stride-4 weight gradient alone (scripts/bench_conv.py conv8s4 wgrad, N = 4): one rocprofv3 --pmc pass
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/wg512; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- python3 $R/scripts/bench_conv.py conv8s4 4 2 wgrad 4 > $O/out.txt 2> $O/err.txt
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/f/*counter_collection.csv")[0]
a=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]=="FETCH_SIZE" and "wgrad" in r["Kernel_Name"]: a[r["Kernel_Name"][:60]].append(float(r["Counter_Value"])*1024*2/1e9)
for k,v in a.items(): print(k, len(v), "launches", round(sum(v)/len(v),2), "GB fetched per launch")
PY
grep conv8 $O/out.txt
