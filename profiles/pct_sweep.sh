#!/bin/bash
# slack percentiles of the two distance grids against the full path's time. usage: bash profiles/pct_sweep.sh [workloads]
for p in ${PCTS:-10 20 30 40}; do echo "pct $p / ${PCT2:-95}"; MVK_DISTANCE_GRID_PCT=$p MVK_DISTANCE_GRID_PCT2=${PCT2:-95} timeout -k 10 100 python profiles/full_time.py ${@:-c2k c3} || exit 1; done
