#!/bin/bash
# collect + summarise ON the box (the raw per-dispatch counter CSVs exceed what gpurun copies back); usage: _job.sh <tag> "<bench args>"
TAG=$1; EXTRA=$2
bash scripts/collect_profiles.sh $TAG "$EXTRA" > /tmp/collect_$TAG.log 2>&1
mkdir -p gpurun_out/profiles_$TAG
CSBSR_PROFILES_DST=gpurun_out/profiles_$TAG python3 scripts/summarise_profiles.py $TAG $TAG > /tmp/sum_$TAG.log 2>&1
cp /tmp/collect_$TAG.log /tmp/sum_$TAG.log gpurun_out/profiles_$TAG/
rm -rf gpurun_out/$TAG
ls gpurun_out/profiles_$TAG | head -12; head -3 gpurun_out/profiles_$TAG/${TAG}_summary.md | cut -c1-100; grep -o '"value": [0-9.]*, "unit": "imgs/s", "n_gpus"' gpurun_out/profiles_$TAG/${TAG}_bench.json
