#!/bin/bash
# avoidance rows / full path against the second distance grid's cell size (fraction of its radius)
for f in 0.5 0.35 0.25 0.18; do
  echo "== MVK_GRID_CELL_FRAC=$f"
  MVK_GRID_CELL_FRAC=$f MVK_DISTANCE_GRID_TRACE=1 timeout -k 10 200 python profiles/full_time.py c2k c3 c5 2>&1 | grep -E "grid 1|per 1M" || exit 1
done
