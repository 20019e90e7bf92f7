"""mvk_set_scene builds its grids on up to four host threads (mvk_capi.cpp upload_distance_grids: the probe distances in
chunks, then the two distance grids and the far grid side by side; mvk_grid.cpp build_grid: the counting pass over box
ranges, the per-cell sorts over chunks of cells).  The builds are the same code on the same inputs,
so every output must carry the bits of the one-thread build (MVK_SCENE_THREADS=1).

Reference: MotionPlanning::setScene (motion_planning.cpp:167-175) replaces the obstacle set; what the library derives
from it is its own."""
import numpy as np
import pytest

import common
import motion_planning_amd as mp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,make", [("c2k", common.config2k), ("c3", common.config3)])
def test_threaded_scene_build_gives_the_bits_of_the_serial_one(monkeypatch, name, make):
    w = make(n=8192)
    out = {}
    for threads in ("4", "1"):
        monkeypatch.setenv("MVK_SCENE_THREADS", threads)
        eng = mp.Engine(w.model, w.boxes[:7], w.params)
        eng.set_scene(w.boxes)                      # (the path a caller with a changing world takes)
        full = eng.eval_batch(w.soa, w.layout)
        rows = eng.collision_avoidance_batch(w.soa[:, :2048], w.layout)
        flags = eng.is_valid_batch(w.soa, w.layout)
        eng.close()
        out[threads] = (full, rows, flags)
    for key in ("flags", "band", "min_dist", "margin"):
        assert np.array_equal(out["4"][0][key], out["1"][0][key], equal_nan=out["4"][0][key].dtype.kind == "f"), (name, key)
    for key in ("dist", "box", "p_env", "p_robot", "row", "lb", "ok"):
        a, b = out["4"][1][key], out["1"][1][key]
        assert np.array_equal(a, b, equal_nan=a.dtype.kind == "f"), (name, key)
    assert np.array_equal(out["4"][2]["flags"], out["1"][2]["flags"])
