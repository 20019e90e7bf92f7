#!/bin/bash
# Per-kernel durations only (rocprofv3 --kernel-trace --stats). usage: bash profiles/ktrace.sh <tag> <workloads...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  OUT=$ROOT/gpurun_out/kt_${TAG}_$w
  mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $ROOT/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-planner --no-edges > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
  python3 $ROOT/profiles/kstats.py $OUT
done
