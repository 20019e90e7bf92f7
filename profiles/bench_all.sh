#!/bin/bash
# Runs on the GPU box: GPU parity tests, then one bench line per workload. usage: bash profiles/bench_all.sh <tag>
TAG=${1:-x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -4 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
for wl in c2 c2se3 c2k c3 c5; do
  timeout -k 10 120 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-planner --no-edges > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || { echo "bench $wl failed"; tail -3 $OUT/bench_$wl.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$wl.json").read().strip().splitlines()[-1])
print("$wl", "%.4g checks/s" % d["value"], "kernel_ms %.3f" % d["roofline"]["kernel_ms"])
PY
done
