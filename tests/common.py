"""Seeded synthetic workloads of BASELINE.json's configs, shared by tests and bench.py.

Everything is derived from numpy's counter-based Philox generator with seed 0x5EED0000 + config id, so the
same inputs are produced on the CPU box (golden generation) and on the GPU box.
"""
import numpy as np

import motion_planning_amd as mp
from motion_planning_amd import _abi as A

SEED0 = 0x5EED0000


def rng_for(config_id, stream=0):
    return np.random.Generator(np.random.Philox(key=SEED0 + config_id, counter=[0, 0, 0, stream]))


def random_unit_quats(rng, n):
    q = rng.normal(size=(4, n))
    return q / np.linalg.norm(q, axis=0)


def yaw_quats(yaw):
    q = np.zeros((4, yaw.shape[0]))
    q[2] = np.sin(0.5 * yaw)
    q[3] = np.cos(0.5 * yaw)
    return q


def random_boxes(rng, m, lo, hi, half_lo, half_hi, orient):
    b = np.zeros((m, 10))
    b[:, 0:3] = rng.uniform(lo, hi, (m, 3))
    if orient == "yaw":
        b[:, 3:7] = yaw_quats(rng.uniform(-np.pi, np.pi, m)).T
    elif orient == "so3":
        b[:, 3:7] = random_unit_quats(rng, m).T
    else:
        b[:, 6] = 1.0
    b[:, 7:10] = 2.0 * rng.uniform(half_lo, half_hi, (m, 3))
    return b


class Workload:
    def __init__(self, name, model, params, boxes, soa, layout):
        self.name, self.model, self.params, self.boxes, self.soa, self.layout = name, model, params, boxes, soa, layout


def hydrus_states(rng, n, J, orient="yaw"):
    """pose xy in the planner bounds (motion_planning.cpp:471-476), z in the z bounds, joints in +-1.58"""
    st = np.zeros((7 + J, n))
    st[0] = rng.uniform(-2.0, 5.0, n)
    st[1] = rng.uniform(-2.2, 2.2, n)
    st[2] = rng.uniform(-2.2, 2.2, n)
    st[3:7] = yaw_quats(rng.uniform(-np.pi, np.pi, n)) if orient == "yaw" else random_unit_quats(rng, n)
    st[7:] = rng.uniform(-1.58, 1.58, (J, n))
    return st


def config2(n=1 << 20, orient="yaw", n_links=6, n_boxes=100):
    """c2: HYDRUS 6-link, joints+base, 1M random samples, 100 yawed boxes."""
    rng = rng_for(2)
    model = mp.make_model("hydrus", n_links)
    params = mp.default_params(model, A.MODE_JOINTS_AND_BASE, A.SE2 if orient == "yaw" else A.SE3)
    boxes = random_boxes(rng, n_boxes, [-2.0, -2.2, -2.2], [5.0, 2.2, 2.2], 0.05, 0.5, "yaw")
    soa = hydrus_states(rng, n, model.n_joint, orient)
    return Workload("c2", model, params, boxes, soa, A.LAYOUT_POSE7)


def config2k(n=1 << 20, n_boxes=1000):
    """north-star target workload: HYDRUS 6-link, joints+base SE(2), 1k obstacles.  The c2 volume with 1000 boxes of
    half extent 0.02-0.25 m (the c3 size law), yawed; boxes fill ~15 % of the volume."""
    rng = rng_for(6)
    model = mp.make_model("hydrus", 6)
    params = mp.default_params(model, A.MODE_JOINTS_AND_BASE, A.SE2)
    boxes = random_boxes(rng, n_boxes, [-2.0, -2.2, -2.2], [5.0, 2.2, 2.2], 0.02, 0.25, "yaw")
    soa = hydrus_states(rng, n, model.n_joint, "yaw")
    return Workload("c2k", model, params, boxes, soa, A.LAYOUT_POSE7)


def config3(n=1 << 20, n_boxes=1000):
    """c3: DRAGON 4-link (gimbals), SE(3), 1k randomly oriented boxes in [-3,3]^3."""
    rng = rng_for(3)
    model = mp.make_model("dragon", 4)
    params = mp.default_params(model, A.MODE_JOINTS_AND_BASE, A.SE3)
    params.pos_low[:] = [-1.0, -1.0, 0.0]   # dragon_quad.yaml:37-42
    params.pos_high[:] = [1.0, 1.0, 1.4]
    boxes = random_boxes(rng, n_boxes, [-3.0] * 3, [3.0] * 3, 0.02, 0.25, "so3")
    J = model.n_joint
    st = np.zeros((7 + J, n))
    st[0] = rng.uniform(-1.0, 1.0, n)
    st[1] = rng.uniform(-1.0, 1.0, n)
    st[2] = rng.uniform(0.0, 1.4, n)
    st[3:7] = random_unit_quats(rng, n)
    st[7:] = rng.uniform(-1.58, 1.58, (J, n))
    return Workload("c3", model, params, boxes, st, A.LAYOUT_POSE7)


def config5(n=1 << 20, n_boxes=1000):
    """c5 (one shard): HYDRUS 8-link joint-space sweep at a fixed pose, the c3 box scene."""
    rng = rng_for(5)
    model = mp.make_model("hydrus", 8)
    params = mp.default_params(model, A.MODE_JOINTS_AND_BASE, A.SE3)
    boxes = random_boxes(rng_for(3), n_boxes, [-3.0] * 3, [3.0] * 3, 0.02, 0.25, "so3")
    # the robot sweeps the z = 0 plane about the origin: keep a clear disc of 1.3 m there so that compact
    # configurations are collision free and extended ones are not (otherwise every sample collides)
    slab = (np.abs(boxes[:, 2]) < 0.75) & (np.hypot(boxes[:, 0], boxes[:, 1]) < 1.3)
    boxes = boxes[~slab]
    J = model.n_joint
    st = np.zeros((7 + J, n))
    st[6] = 1.0
    st[7:] = rng.uniform(-1.58, 1.58, (J, n))
    return Workload("c5", model, params, boxes, st, A.LAYOUT_POSE7)


def config1():
    """c1: HYDRUS quad in the default horizontal-gap world (sampling_based_method.cpp:84-112),
    start / goal of config/hydrus/quad/GapWithTransformation.yaml:1-13."""
    model = mp.make_model("hydrus", 4)
    params = mp.default_params(model, A.MODE_JOINTS_AND_BASE, A.SE2)
    boxes = mp.horizontal_gap_scene()
    start = np.array([-1.0, 0.0, -1.57, 1.57, 1.57, 1.57])
    goal = np.array([1.0, 1.0, -1.57, 1.57, 1.57, 1.57])
    return model, params, boxes, start, goal


def config4(n_edges=1 << 16, n_links=6, n_boxes=100):
    """c4: HYDRUS 6-link edges in the OMPL layout of mode 2 / SE2, nd = 64 steps per edge;
    s2 = s1 + a step of OMPL distance ~ 0.2 * maxExtent (the RRT range default)."""
    rng = rng_for(4)
    model = mp.make_model("hydrus", n_links)
    params = mp.default_params(model, A.MODE_JOINTS_AND_BASE, A.SE2)
    boxes = random_boxes(rng_for(2), n_boxes, [-2.0, -2.2, -2.2], [5.0, 2.2, 2.2], 0.05, 0.5, "yaw")
    J = model.n_joint
    s1 = np.zeros((3 + J, n_edges))
    s1[0] = rng.uniform(-2.0, 5.0, n_edges)
    s1[1] = rng.uniform(-2.2, 2.2, n_edges)
    s1[2] = rng.uniform(-np.pi, np.pi, n_edges)
    s1[3:] = rng.uniform(-1.58, 1.58, (J, n_edges))
    ext = np.hypot(7.0, 4.4) + 0.5 * np.pi + 0.001 * np.sqrt(J) * 3.16
    step = 0.2 * ext
    d = rng.normal(size=(3 + J, n_edges))
    d[2] *= 0.5
    d /= np.linalg.norm(d, axis=0)
    s2 = s1 + d * rng.uniform(0.3, 1.0, n_edges) * step
    s2[0] = np.clip(s2[0], -2.0, 5.0)
    s2[1] = np.clip(s2[1], -2.2, 2.2)
    s2[2] = (s2[2] + np.pi) % (2 * np.pi) - np.pi
    s2[3:] = np.clip(s2[3:], -1.58, 1.58)
    return model, params, boxes, s1, s2, 64


def margins_agree(model, states, got, ref):
    """Per state, [n] bool: the stability margins `got` and `ref` agree to the bar.  The bar is 1e-5 relative
    (BASELINE.json north_star).  The static thrust is the minimum-norm solution through the normal equations (V-spec 4, as
    the reference's FullPivLU of P P^T); near a singular rotor arrangement (all rotors nearly in line) that solve
    carries cond(A)^2 roundings, which two correct fp64 evaluations in different operation order (the GPU contracts to
    FMAs, the oracle is built -ffp-contract=off) do not share: measured 0.2 - 0.35 u cond(A)^2 on the two such states the
    fuzz soak found in 2.9 M (profiles/soak_fuzz.py, profiles/r03/soak_fuzz.txt: thrusts of 4e4 and 3.8e5 N, cond(A) 1e6
    and 2e6).  A state that misses the 1e-5 bar is therefore held to 8 u cond(A)^2 instead, cond(A) from the independent
    NumPy restatement (oracle/np_vspec.py).  Such states are unstable by orders of magnitude: no flag is in question."""
    import os, sys
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    with np.errstate(invalid="ignore"):   # (inf - inf where both report a non-finite margin)
        ok = np.abs(got - ref) <= 1e-9 + 1e-5 * np.abs(ref)
    ok |= ~np.isfinite(ref)
    if not ok.all():
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        import np_vspec as NP
        for i in np.flatnonzero(~ok):
            cond = NP.model_update(model, np.ascontiguousarray(states[:, i]))["cond"]
            ok[i] = abs(got[i] - ref[i]) <= 8.0 * 1.1102230246251565e-16 * cond * cond * abs(ref[i])
    return ok
