#!/bin/bash
# rocprofv3 kernel stats of any python probe. usage: bash profiles/ktrace_cmd.sh <tag> <script.py> [args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/kt_$TAG
mkdir -p $OUT
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $SCRIPT "$@" > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 $ROOT/profiles/kstats.py $OUT
