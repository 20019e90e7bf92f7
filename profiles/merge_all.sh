#!/bin/bash
# Here, after profile_all.sh: copies the summaries into profiles/<round>/ and merges them into profiles/latest_pmc.json.
# usage: bash profiles/merge_all.sh <tag> <round dir, e.g. r03>
TAG=$1; R=$2
cd "$(dirname "$0")/.."
cp gpurun_out/prof_$TAG/summary.txt profiles/$R/c2k_summary.txt
python3 profiles/summarize.py gpurun_out/prof_$TAG --merge profiles/latest_pmc.json --workload c2k --source profiles/$R/c2k_summary.txt > /dev/null
for w in c2 c3 c5; do
  cp gpurun_out/prof_${TAG}_$w/summary.txt profiles/$R/${w}_summary.txt
  python3 profiles/summarize.py gpurun_out/prof_${TAG}_$w --merge profiles/latest_pmc.json --workload $w --source profiles/$R/${w}_summary.txt > /dev/null
done
for w in c2k c3; do
  cp gpurun_out/pmc_full_${TAG}_$w.log profiles/$R/distance_kernels_${w}_summary.txt
  python3 profiles/summarize.py gpurun_out/pmcfull_${TAG}_$w --kind distance --merge profiles/latest_pmc.json --workload $w --source profiles/$R/distance_kernels_${w}_summary.txt > /dev/null
done
python3 - <<'PY'
import json
j = json.load(open("profiles/latest_pmc.json"))
for k, v in sorted(j.items()):
    print(k, v.get("kernel_fingerprint"), v.get("source"))
PY
python3 profiles/fingerprint.py .
