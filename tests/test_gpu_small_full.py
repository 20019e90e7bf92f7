"""Full path (verdict + exact minimum distance + margin + band + statistics) on small batches: up to 4096 states per
call go through distance_small_kernel (a wave per state, one lane per primitive and list slice) instead
of distance_kernel (one lane per state); MVK_NO_SMALL_FULL=1 keeps the latter.  Both take the minimum of the same
per-pair values, so every output must carry the same bits, and both must agree with the oracle.

Reference: the per-state call MotionPlanning::isStateValid (motion_planning.cpp:76-164) and the statistics of addState
(:546-642) that mvk_is_valid / mvk_eval_batch / mvk_reduce_stats stand for."""
import numpy as np
import pytest

import common
import motion_planning_amd as mp
import oracle_lib as O
from motion_planning_amd import _abi as A

pytestmark = pytest.mark.gpu


def _workloads():
    w = common.config2k(n=4096)
    yield "c2k", w.model, w.boxes, w.params, w.soa, w.layout
    w = common.config3(n=4096)
    yield "c3 dragon", w.model, w.boxes, w.params, w.soa, w.layout
    w = common.config2(n=4096)
    yield "c2 (100 boxes)", w.model, w.boxes, w.params, w.soa, w.layout
    w = common.config5(n=4096)
    yield "c5 hydrus-8", w.model, w.boxes, w.params, w.soa, w.layout
    # far from every box: neither distance grid can answer, the sweep over all boxes does
    w = common.config2k(n=2048)
    far = w.soa.copy()
    far[0] += 9.0
    far[2] += np.linspace(0.0, 6.0, far.shape[1])
    yield "c2k far away", w.model, w.boxes, w.params, far, w.layout
    # fewer than 64 boxes: no distance grid at all
    model, params, boxes, start, goal = common.config1()
    rng = np.random.default_rng(4)
    st = common.hydrus_states(rng, 1024, model.n_joint)
    st[0] = rng.uniform(-1.5, 1.5, 1024)
    st[1] = rng.uniform(-1.5, 1.5, 1024)
    st[2] = 0.0
    yield "c1 gap world", model, boxes, params, st, A.LAYOUT_POSE7


def _same(a, b, key):
    x, y = a[key], b[key]
    return np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)


@pytest.mark.parametrize("name,model,boxes,params,soa,layout", list(_workloads()), ids=lambda v: v if isinstance(v, str) else "")
def test_group_per_state_kernel_equals_lane_per_state_kernel(monkeypatch, name, model, boxes, params, soa, layout):
    soa = soa.copy()
    soa[7, 5] = np.nan          # a non-finite state: invalid, band, NaN distance, out of the statistics
    soa[0, 11] = np.inf
    sizes = (1, 3, 4, 5, 33, 300, min(4096, soa.shape[1]))
    out, stats = {}, {}
    for small in (True, False):
        if not small:
            monkeypatch.setenv("MVK_NO_SMALL_FULL", "1")
        eng = mp.Engine(model, boxes, params)
        out[small] = [eng.eval_batch(soa[:, :n], layout) for n in sizes]
        stats[small] = eng.reduce_stats()          # of the last call
        eng.close()
    for a, b, n in zip(out[True], out[False], sizes):
        for key in ("flags", "band", "min_dist", "margin", "fc_t_min", "thrust", "cog"):
            assert _same(a, b, key), (name, n, key, np.flatnonzero(a[key] != b[key])[:5])
    assert stats[True] == stats[False], (stats[True], stats[False])
    # and against the oracle (the tolerances of test_gpu_parity.py)
    n = sizes[-1]
    ref = O.eval_batch(model, boxes, params, soa[:, :n], layout, threads=8)
    got = out[True][-1]
    clear = ~ref["band"].astype(bool)
    assert np.array_equal(got["flags"][clear], ref["flags"][clear])
    fin = np.isfinite(ref["min_dist"])
    assert np.array_equal(np.isfinite(got["min_dist"]), fin)
    assert np.allclose(got["min_dist"][fin], ref["min_dist"][fin], rtol=0, atol=1e-9)
    assert not got["flags"][5] and got["band"][5] and np.isnan(got["min_dist"][5])


def test_single_state_call_matches_the_oracle():
    """mvk_is_valid (one state in the OMPL layout of mode 2 / SE2: x, y, yaw, joints; z = 0)"""
    w = common.config2k(n=64)
    soa = w.soa.copy()
    soa[2] = 0.0
    eng = mp.Engine(w.model, w.boxes, w.params)
    ref = O.eval_batch(w.model, w.boxes, w.params, soa, w.layout)
    for i in range(64):
        s = soa[:, i]
        yaw = 2.0 * np.arctan2(s[5], s[6])
        r = eng.is_valid(np.concatenate([s[0:2], [yaw], s[7:]]))
        assert abs(r.min_dist - ref["min_dist"][i]) < 1e-9 and abs(r.margin - ref["margin"][i]) < 1e-5 * max(1.0, abs(ref["margin"][i])), i
        if not ref["band"][i]:
            assert r.valid == ref["flags"][i] and r.collision == (ref["min_dist"][i] <= 0.0), i
    eng.close()


def test_many_small_full_calls_in_a_row_keep_their_answers_apart():
    """host calls of up to 1024 states read and write pinned memory and are reported done through a pinned word by the
    last block of the kernel: a few hundred calls of changing size on one context, interleaved with flag-only calls
    (which share the word and its sequence numbers), against one large call (lane-per-state kernel) on the same states"""
    w = common.config2k(n=8192)
    eng = mp.Engine(w.model, w.boxes, w.params)
    big = eng.is_valid_batch(w.soa, w.layout, want=("flags", "min_dist", "margin", "band"))
    rng = np.random.default_rng(3)
    for it in range(400):
        n = int(rng.integers(1, 1025 if it % 50 == 0 else 41))
        a = int(rng.integers(0, 8192 - n))
        got = eng.is_valid_batch(w.soa[:, a:a + n], w.layout, want=("flags", "min_dist", "margin", "band"))
        for key in ("flags", "min_dist", "margin", "band"):
            assert np.array_equal(got[key], big[key][a:a + n]), (it, n, a, key)
        if it % 3 == 0:
            f = eng.is_valid_batch(w.soa[:, a:a + n], w.layout)["flags"]
            assert np.array_equal(f, big["flags"][a:a + n]), (it, n, a)
    eng.close()


@pytest.mark.parametrize("name,model,boxes,params,soa,layout", list(_workloads()), ids=lambda v: v if isinstance(v, str) else "")
def test_avoidance_rows_wave_per_state_equal_lane_per_state(monkeypatch, name, model, boxes, params, soa, layout):
    """collision-avoidance rows (collision_avoidance.cpp:102-277): up to 4096 states per call take avoid_small_kernel
    (a wave per state; host calls of up to 256 states through pinned memory and the done word); every output must
    equal avoid_kernel's bit for bit, and the oracle's within the tolerances of test_gpu_avoid.py"""
    soa = soa[:, :1024]
    sizes = (1, 2, 5, 64, 256, 257, 1024)
    out = {}
    for small in (True, False):
        if not small:
            monkeypatch.setenv("MVK_NO_SMALL_FULL", "1")
        eng = mp.Engine(model, boxes, params)
        out[small] = [eng.collision_avoidance_batch(soa[:, :n], layout) for n in sizes]
        eng.close()
    for a, b, n in zip(out[True], out[False], sizes):
        for key in ("dist", "box", "p_env", "p_robot", "row", "lb", "ok"):
            assert np.array_equal(a[key], b[key], equal_nan=(a[key].dtype.kind == "f")), (name, n, key)
    ref = O.avoid_batch(model, boxes, params, soa[:, :256])
    got = out[True][4]
    assert np.allclose(got["dist"], ref["dist"], rtol=1e-5, atol=1e-9)
    tie = np.abs(got["dist"] - ref["dist"]) > 0
    assert np.all((got["box"] == ref["box"]) | tie) and np.array_equal(got["ok"], ref["ok"])


@pytest.mark.parametrize("name,model,boxes,params,soa,layout", list(_workloads()), ids=lambda v: v if isinstance(v, str) else "")
def test_flag_calls_of_a_few_states_equal_the_fused_path(monkeypatch, name, model, boxes, params, soa, layout):
    """OMPL's isValid (motion_planning.cpp:236: one state per call): host calls of up to 256 states give every state a
    block of two waves (flag_small_kernel: walk 1 and the collision test side by side); same verdicts as the
    lane-per-state kernels and as the oracle outside the band, over single calls and small batches"""
    soa = soa[:, :600].copy()
    soa[7, 5] = np.nan
    ref = O.eval_batch(model, boxes, params, soa, layout, threads=8)
    clear = ~ref["band"].astype(bool)
    got = {}
    for small in (True, False):
        if not small:
            monkeypatch.setenv("MVK_NO_SMALL_FULL", "1")
        eng = mp.Engine(model, boxes, params)
        singles = np.concatenate([eng.is_valid_batch(soa[:, i:i + 1], layout)["flags"] for i in range(300)])
        chunks = np.concatenate([eng.is_valid_batch(soa[:, a:a + 16], layout)["flags"] for a in range(300, 332, 16)]
                                + [eng.is_valid_batch(soa[:, 332:588], layout)["flags"], eng.is_valid_batch(soa[:, 588:600], layout)["flags"]])
        got[small] = np.concatenate([singles, chunks])
        eng.close()
    assert np.array_equal(got[True], got[False])
    assert np.array_equal(got[True][clear], ref["flags"][clear])
    assert got[True][5] == 0
    assert 0 < got[True].sum() < 600 or name.startswith("c2k far")


def test_edge_calls_of_a_few_steps_equal_the_fused_path(monkeypatch):
    """OMPL's checkMotion (motion_planning.cpp:238,249: one edge per call): edge calls of up to 256 steps in all give
    every step a block of two waves (flag_small_kernel, edge form); same first_invalid as the lane-per-step kernels
    and as the oracle outside the band -- single edges, a few edges, ragged step counts, begin/end"""
    model, params, boxes, s1, s2, nd64 = common.config4(n_edges=400)
    sp = O.Space(params, model.n_joint)
    nd_var = np.array([max(1, min(60, sp.valid_segment_count(s1[:, e], s2[:, e]))) for e in range(400)], np.int32)
    g_ok, g_first, g_band = O.check_motion_batch(model, boxes, params, s1, s2, nd_var)
    got = {}
    for small in (True, False):
        if not small:
            monkeypatch.setenv("MVK_NO_SMALL_FULL", "1")
        eng = mp.Engine(model, boxes, params)
        ok, first = [], []
        for e in range(200):                                   # one edge per call
            o, f = eng.check_motion_batch(s1[:, e:e + 1], s2[:, e:e + 1], nd_var[e:e + 1])
            ok.append(o); first.append(f)
        for a in range(200, 400, 4):                           # four edges per call (<= 240 steps), uniform and ragged
            o, f = eng.check_motion_batch(s1[:, a:a + 4], s2[:, a:a + 4], nd_var[a:a + 4])
            ok.append(o); first.append(f)
        eng.check_motion_batch_begin(s1[:, :3], s2[:, :3], 41)
        o41, f41 = eng.check_motion_batch_end()
        got[small] = (np.concatenate(ok), np.concatenate(first), o41, f41)
        eng.close()
    for a, b in zip(got[True], got[False]):
        assert np.array_equal(a, b)
    clear = ~g_band.astype(bool)
    assert np.array_equal(got[True][0][clear], g_ok[clear]) and np.array_equal(got[True][1][clear], g_first[clear])
    assert 0 < got[True][0].sum() < 400


def test_mixed_small_calls_share_the_done_word_across_the_sequence_wrap():
    """every polled call of a context (flag, full path, avoidance rows, edges; block-per-state kernels and the fused
    kernel alike) takes the next 16-bit sequence number and the same pinned done word, and the edge tags pack that
    number: 70 000 calls of all kinds in turn, across the wrap of the number, each checked against answers computed
    beforehand in large calls"""
    w = common.config2(n=2048)
    eng = mp.Engine(w.model, w.boxes, w.params)
    big = eng.is_valid_batch(w.soa, w.layout, want=("flags", "min_dist", "margin", "band"))
    av = eng.collision_avoidance_batch(w.soa[:, :64], w.layout)
    model, params, boxes, s1, s2, nd = common.config4(n_edges=600)
    eng_e = mp.Engine(model, boxes, params)
    ok_all, first_all = eng_e.check_motion_batch(s1, s2, 41)       # one large call (two launches)
    rng = np.random.default_rng(11)
    for it in range(70000):
        i = int(rng.integers(0, 2000))
        kind = it % 7
        e = i % 590
        ok, first = eng_e.check_motion_batch(s1[:, e:e + 1], s2[:, e:e + 1], 41)    # OMPL checkMotion (its own context: numbers and tags)
        assert ok[0] == ok_all[e] and first[0] == first_all[e], it
        if kind < 4:        # OMPL isValid
            assert eng.is_valid_batch(w.soa[:, i:i + 1], w.layout)["flags"][0] == big["flags"][i], it
            if kind == 1 and it % 49 == 1:   # the fused kernel (more than 256 steps) on the same tags
                ok, first = eng_e.check_motion_batch(s1[:, e:e + 9], s2[:, e:e + 9], 41)
                assert np.array_equal(ok, ok_all[e:e + 9]) and np.array_equal(first, first_all[e:e + 9]), it
        elif kind == 4:
            got = eng.is_valid_batch(w.soa[:, i:i + 3], w.layout, want=("flags", "min_dist"))
            assert np.array_equal(got["min_dist"], big["min_dist"][i:i + 3]) and np.array_equal(got["flags"], big["flags"][i:i + 3]), it
        elif kind == 5 and it % 35 == 5:
            j = i % 60
            got = eng.collision_avoidance_batch(w.soa[:, j:j + 2], w.layout)
            assert np.array_equal(got["dist"], av["dist"][:, j:j + 2]) and np.array_equal(got["box"], av["box"][:, j:j + 2]), it
        else:
            n = 300 + i % 50    # fused_kernel, state form
            assert np.array_equal(eng.is_valid_batch(w.soa[:, :n], w.layout)["flags"], big["flags"][:n]), it
    eng.close()
    eng_e.close()
