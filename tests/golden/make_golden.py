#!/usr/bin/env python3
"""Generates tests/golden/*.npz: seeded inputs + the CPU oracle's outputs for each BASELINE config.

Run in the build container:  python tests/golden/make_golden.py
Before writing, every vector is cross-checked against the independent NumPy restatement
(oracle/np_vspec.py) on a subset.  The reference itself cannot be run (it needs ROS/OMPL/MoveIt/KDL and
the hydrus/dragon packages), so these vectors pin the build's own V-spec, not the third-party maths
("parity unpinned" for statics / collision; the edge arithmetic is pinned by ref_hydrus_planning_log.txt).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import common  # noqa: E402
import np_vspec as NP  # noqa: E402
import oracle_lib as O  # noqa: E402

KEYS = ("flags", "band", "stable", "collision", "min_dist", "margin", "fc_t_min", "fc_rp_min", "thrust",
        "rp_position_margin", "cog", "gimbal", "pivot_ratio")


def crosscheck(w, ref, count=24):
    for i in range(count):
        q = NP.model_update(w.model, w.soa[:, i])
        assert np.allclose(q["cog"], ref["cog"][:, i], rtol=1e-11, atol=1e-12)
        assert abs(q["fc_rp_min"] - ref["fc_rp_min"][i]) <= 1e-9 * max(1, abs(q["fc_rp_min"]))
        assert abs(q["fc_t_min"] - ref["fc_t_min"][i]) <= 1e-9 * max(1, abs(q["fc_t_min"]))
        assert abs(q["rp_position_margin"] - ref["rp_position_margin"][i]) <= 1e-9
        assert np.allclose(q["gimbal"], ref["gimbal"][:len(q["gimbal"]), i], atol=1e-10)
        if np.isfinite(ref["margin"][i]) and q["cond"] < 1e6:
            assert np.allclose(q["thrust"], ref["thrust"][:len(q["thrust"]), i], rtol=1e-7, atol=1e-8)
            assert abs(q["margin"] - ref["margin"][i]) <= 1e-7 * max(1, abs(q["margin"]))


def main():
    n = 1536
    cases = {
        "c2_hydrus6_se2": common.config2(n=n),
        "c2_hydrus6_se3": common.config2(n=n, orient="so3"),
        "c3_dragon4": common.config3(n=n),
        "c5_hydrus8": common.config5(n=n, n_boxes=200),
        "c1_hydrus4": common.config2(n=n, n_links=4, n_boxes=30),
    }
    for name, w in cases.items():
        ref = O.eval_batch(w.model, w.boxes, w.params, w.soa, w.layout, threads=8)
        crosscheck(w, ref)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), soa=w.soa, boxes=w.boxes,
                            **{k: ref[k] for k in KEYS})
        print(name, "valid", int(ref["flags"].sum()), "of", n, "band", int(ref["band"].sum()))
    # edges (c4 shape, fewer edges)
    model, params, boxes, s1, s2, nd = common.config4(n_edges=512)
    ok, first, band = O.check_motion_batch(model, boxes, params, s1, s2, nd)
    sp = O.Space(params, model.n_joint)
    nd_var = np.array([max(1, sp.valid_segment_count(s1[:, e], s2[:, e])) for e in range(s1.shape[1])], np.int32)
    ok_v, first_v, band_v = O.check_motion_batch(model, boxes, params, s1, s2, nd_var)
    np.savez_compressed(os.path.join(HERE, "c4_edges.npz"), s1=s1, s2=s2, boxes=boxes, nd=nd, ok=ok, first=first,
                        band=band, nd_var=nd_var, ok_var=ok_v, first_var=first_v, band_var=band_v)
    print("c4 edges ok", int(ok.sum()), "of", len(ok), "| variable nd: ok", int(ok_v.sum()), "mean nd", nd_var.mean())


if __name__ == "__main__":
    main()
