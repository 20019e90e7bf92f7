#!/bin/bash
# Runs on the GPU box: the round's full set of trace + PMC passes (flag path of c2k / c2 / c3 / c5, distance kernels of
# c2k / c3).  usage: bash profiles/profile_all.sh <tag>; afterwards merge with summarize.py (see profiles/README.md).
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash $ROOT/profiles/run_profile.sh $TAG > $ROOT/gpurun_out/run_profile_$TAG.log 2>&1 || exit 1
for w in c2 c3 c5; do bash $ROOT/profiles/run_profile.sh ${TAG}_$w --workload $w > $ROOT/gpurun_out/run_profile_${TAG}_$w.log 2>&1 || exit 1; done
for w in c2k c3; do bash $ROOT/profiles/pmc_full.sh $TAG $w > $ROOT/gpurun_out/pmc_full_${TAG}_$w.log 2>&1 || exit 1; done
grep -h "^== .*valu_busy" $ROOT/gpurun_out/run_profile_$TAG*.log $ROOT/gpurun_out/pmc_full_${TAG}_*.log
