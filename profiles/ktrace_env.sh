#!/bin/bash
# like ktrace.sh, for one workload, with the environment passed through (diagnostic switches such as MVK_QUEUE_REC_CAP)
# usage: VAR=... bash profiles/ktrace_env.sh <tag> <workload>
TAG=$1; W=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/kt_${TAG}_$W
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $ROOT/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-planner --no-edges > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 $ROOT/profiles/kstats.py $OUT
