#!/bin/bash
# Per-kernel durations of the distance kernels (full path, avoidance rows): rocprofv3 --kernel-trace --stats around
# profiles/full_time.py.  usage: bash profiles/ktrace_full.sh <tag> <workload>
TAG=$1; W=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/ktfull_${TAG}_$W
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $ROOT/profiles/full_time.py $W > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
cat $OUT/log.txt | grep -E 'model_kernel|distance_kernel'
python3 $ROOT/profiles/kstats.py $OUT
