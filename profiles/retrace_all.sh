#!/bin/bash
# Runs on the GPU box: only the two kernel-trace passes of run_profile.sh for the four flag-path workloads (the PMC passes
# of the same kernel fingerprint are kept).  usage: bash profiles/retrace_all.sh <tag>; afterwards, here, re-summarize:
# python3 profiles/summarize.py gpurun_out/prof_<tag>[_<workload>] > gpurun_out/prof_<tag>[_<workload>]/summary.txt, then merge_all.sh
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export MVK_PROFILE_TRACE_ONLY=1
bash $ROOT/profiles/run_profile.sh $TAG > $ROOT/gpurun_out/retrace_$TAG.log 2>&1 || exit 1
for w in c2 c3 c5; do bash $ROOT/profiles/run_profile.sh ${TAG}_$w --workload $w > $ROOT/gpurun_out/retrace_${TAG}_$w.log 2>&1 || exit 1; done
echo retraced
