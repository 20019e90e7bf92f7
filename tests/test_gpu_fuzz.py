"""Randomised scenes through every kernel family against the oracle: robot kind and link count, box count (0 to a
thousand: no grid, one grid, all three), box shapes down to plates and needles, box and root orientations, states
inside, around and far outside the scene, a non-finite state.  Per draw: flag path (flags and packed mask), full
path in both kernel sizes (wave per state / lane per state), avoidance rows in both, edge checks.

Bar (BASELINE.json north_star): flags bit-exact outside the 1e-6 band, distances and margins within 1e-9 absolute /
1e-5 relative as in test_gpu_parity.py.  Reference: MotionPlanning::isStateValid (motion_planning.cpp:76-164),
DiscreteMotionValidator (:238-249), CollisionAvoidance::getConstraint (collision_avoidance.cpp:102-277)."""
import os

import numpy as np
import pytest
import torch

import common
import motion_planning_amd as mp
import oracle_lib as O
from motion_planning_amd import _abi as A

pytestmark = pytest.mark.gpu

DRAWS = [("hydrus", 4, 0), ("hydrus", 6, 1), ("hydrus", 8, 30), ("dragon", 4, 70), ("hydrus", 6, 300), ("dragon", 4, 1000),
         ("hydrus", 4, 1000), ("hydrus", 8, 600)]


@pytest.mark.parametrize("kind,n_links,n_boxes", DRAWS)
def test_random_scene_all_kernel_families(kind, n_links, n_boxes):
    # (MVK_FUZZ_SEED_OFFSET: other draws of the same families, profiles/soak_fuzz.py; the committed runs use 0)
    rng = np.random.default_rng(1000 * n_links + n_boxes + (7 if kind == "dragon" else 0) + 100003 * int(os.environ.get("MVK_FUZZ_SEED_OFFSET", "0")))
    model = mp.make_model(kind, n_links)
    se3 = kind == "dragon" or n_boxes % 2 == 0
    params = mp.default_params(model, A.MODE_JOINTS_AND_BASE, A.SE3 if se3 else A.SE2)
    J = model.n_joint
    boxes = common.random_boxes(rng, n_boxes, [-3, -3, -2], [3, 3, 2], 0.02, 0.3, "so3" if se3 else "yaw")
    if n_boxes >= 30:                       # plates, needles, a point
        boxes[0:10, 7] = 0.0
        boxes[10:20, 8:10] = 1e-3
        boxes[20, 7:10] = 0.0
    n = 2600
    st = np.zeros((7 + J, n))
    scale = np.where(np.arange(n) % 4 == 3, 3.0, 1.0)          # a quarter of the states around / far outside the scene
    st[0:3] = rng.uniform(-2.5, 2.5, (3, n)) * scale
    st[3:7] = common.random_unit_quats(rng, n) if se3 else common.yaw_quats(rng.uniform(-np.pi, np.pi, n))
    st[7:] = rng.uniform(-1.5, 1.5, (J, n))
    st[8 if J > 1 else 7, 17] = np.nan
    layout = A.LAYOUT_POSE7
    ref = O.eval_batch(model, boxes, params, st, layout, threads=8)
    eng = mp.Engine(model, boxes, params)
    dev = torch.device("cuda:0")
    d = torch.from_numpy(st).to(dev)

    # flag path: flags and packed mask
    flags = eng.is_valid_batch(st, layout)["flags"]
    mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
    eng.is_valid_mask_dev(d, layout, mask)
    torch.cuda.synchronize()
    bits = np.unpackbits(mask.cpu().numpy().view(np.uint8), bitorder="little")[:n]
    clear = ~ref["band"].astype(bool)
    assert np.array_equal(flags[clear], ref["flags"][clear]) and np.array_equal(bits, flags)
    assert flags[17] == 0

    # full path, lane per state (n > 4096 is not needed: MVK_SMALL_FULL_MAX is a host-side switch) and wave per state
    big = eng.eval_batch(np.concatenate([st, st], 1), layout)            # 5200 states: distance_kernel
    small = eng.eval_batch(st[:, :300], layout)                           # distance_small_kernel
    one = eng.eval_batch(st[:, 40:41], layout)                            # full_small_kernel, a single state
    fin = np.isfinite(ref["min_dist"])
    for got, sl in ((big, slice(0, n)), (small, slice(0, 300)), (one, slice(40, 41))):
        k = sl.stop - sl.start
        band = (ref["band"][sl] | got["band"][:k]).astype(bool)
        assert np.array_equal(got["flags"][:k][~band], ref["flags"][sl][~band])
        f = fin[sl]
        assert np.array_equal(np.isfinite(got["min_dist"][:k]), f)
        assert np.allclose(got["min_dist"][:k][f], ref["min_dist"][sl][f], rtol=0, atol=1e-9)
        fm = np.isfinite(ref["margin"][sl])
        assert np.all(common.margins_agree(model, st[:, sl], got["margin"][:k], ref["margin"][sl])[fm])   # (1e-5; near-singular thrust solves: see there)
    assert np.array_equal(big["min_dist"][:n], big["min_dist"][n:], equal_nan=True)
    assert np.array_equal(big["min_dist"][:300], small["min_dist"], equal_nan=True)     # the two kernel sizes: same bits
    assert np.array_equal(big["min_dist"][40:41], one["min_dist"], equal_nan=True)

    # avoidance rows, both kernel sizes
    ok_states = np.all(np.isfinite(st), axis=0)
    sa = np.ascontiguousarray(st[:, ok_states][:, :1200])
    ra = O.avoid_batch(model, boxes, params, sa)
    ga_small = eng.collision_avoidance_batch(sa, layout)
    ga_big = eng.collision_avoidance_batch(np.concatenate([sa] * 4, 1), layout)         # 4800 states: avoid_kernel
    assert np.allclose(ga_small["dist"], ra["dist"], rtol=1e-5, atol=1e-9)
    assert np.array_equal(ga_small["ok"], ra["ok"])
    for key in ("dist", "box", "lb", "ok"):
        a, b = ga_small[key], ga_big[key][..., :sa.shape[1]]
        assert np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" else np.array_equal(a, b), key
    if n_boxes:
        assert (ga_small["box"] == ra["box"]).mean() > 0.995

    # edge checks between consecutive states, ragged step counts
    sp = mp.Space(params, J)
    E = 400
    s1 = np.stack([_pack(st[:, i], se3) for i in range(100, 100 + E)], 1)
    s2 = np.stack([_pack(st[:, i] * 0.98 + 0.02 * st[:, i + 1], se3) for i in range(100, 100 + E)], 1)
    assert s1.shape[0] == sp.dim
    nd = rng.integers(1, 40, E).astype(np.int32)
    ok, first = eng.check_motion_batch(s1, s2, nd)
    g_ok, g_first, g_band = O.check_motion_batch(model, boxes, params, s1, s2, nd)
    diff = (ok != g_ok) | (first != g_first)
    assert not np.any(diff & ~g_band.astype(bool))
    eng.close()


def _pack(pose7j, se3):
    """POSE7 row -> the OMPL layout of mode 2: SE(3): x y z qx qy qz qw joints; SE(2): x y yaw joints"""
    if se3:
        return pose7j
    qz, qw = pose7j[5], pose7j[6]
    return np.concatenate([pose7j[0:2], [2.0 * np.arctan2(qz, qw)], pose7j[7:]])
