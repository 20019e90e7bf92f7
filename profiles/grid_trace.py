import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["MVK_DISTANCE_GRID_TRACE"] = "1"
import common, motion_planning_amd as mp
for name, f in [("c2k", common.config2k), ("c3", common.config3), ("c2", common.config2), ("c5", common.config5)]:
    w = f(n=64)
    print(name, flush=True)
    e = mp.Engine(w.model, w.boxes, w.params); e.close()
