for p in 40 60 75 85 92; do echo "pct $p"; MVK_DISTANCE_GRID_PCT=$p MVK_DISTANCE_GRID_TRACE=1 timeout -k 10 100 python profiles/full_time.py c2k c3 || exit 1; done
