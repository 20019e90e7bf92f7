"""The fused (one-launch) flag path's recovery: when the blocks of a fused launch are not co-resident in time, a block
gives up at the grid barrier, says so in the done word, and the host repeats the call with two launches
(mvk_capi.cpp fused_wait / mvk_kernels.hip fused_tail).  Never seen in normal runs, so MVK_FUSED_SPIN_LIMIT=0 (read at
mvk_create) forces it: every block that is not the last to arrive gives up at its first poll.  Verdicts must equal
those of the two-launch path (MVK_NO_FUSED=1) for states and for edges, and the calls that follow on the same ctx
must stay correct (counter sets, first_tag, sequence numbers)."""
import os

import numpy as np
import pytest

import common
import motion_planning_amd as mp
from motion_planning_amd import _abi as A

pytestmark = pytest.mark.gpu


def engine_with(env, model, boxes, params):
    keep = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return mp.Engine(model, boxes, params)
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_fused_barrier_give_up_falls_back_to_two_launches():
    lib = A.load()
    n = 3000                                     # 12 blocks: the fused kernel's range (flag_small_kernel takes <= 256)
    w = common.config2(n=n)
    two = engine_with({"MVK_NO_FUSED": "1"}, w.model, w.boxes, w.params)
    lim = engine_with({"MVK_FUSED_SPIN_LIMIT": "0"}, w.model, w.boxes, w.params)
    plain = mp.Engine(w.model, w.boxes, w.params)
    model, params, boxes, s1, s2, _ = common.config4(n_edges=96)
    e_two = engine_with({"MVK_NO_FUSED": "1"}, model, boxes, params)
    e_lim = engine_with({"MVK_FUSED_SPIN_LIMIT": "0"}, model, boxes, params)
    try:
        ref = two.is_valid_batch(w.soa, w.layout)["flags"]
        assert 0 < ref.sum() < n
        assert np.array_equal(plain.is_valid_batch(w.soa, w.layout)["flags"], ref)
        assert lib.mvk_debug_fused_giveups(plain.ctx, 0) == 0           # the normal run never gives up
        got = lim.is_valid_batch(w.soa, w.layout)["flags"]
        assert lib.mvk_debug_fused_giveups(lim.ctx, 0) >= 1, "the switch did not force the give-up path"
        assert np.array_equal(got, ref)

        nd = 41                                    # 96 x 41 = 3936 steps = 16 blocks: the fused edge kernels' range
        ok_r, first_r = e_two.check_motion_batch(s1, s2, nd)
        assert 0 < ok_r.sum() < len(ok_r)
        for it in range(52):                       # give-ups, with 50+ calls following one another on the same ctx
            if it % 3 == 0:
                ok, first = e_lim.check_motion_batch(s1, s2, nd)
            elif it % 3 == 1:
                e_lim.check_motion_batch_begin(s1, s2, nd)
                ok, first = e_lim.check_motion_batch_end()
            else:                                  # a smaller call in between: the other counter set, a shorter tag range
                k = 24 + it
                ok, first = e_lim.check_motion_batch(s1[:, :k], s2[:, :k], nd)
                assert np.array_equal(ok, ok_r[:k]) and np.array_equal(first, first_r[:k]), it
                continue
            assert np.array_equal(ok, ok_r) and np.array_equal(first, first_r), it
        assert lib.mvk_debug_fused_giveups(e_lim.ctx, 1) >= 10
        assert lib.mvk_debug_fused_giveups(e_lim.ctx, 0) == 0            # reset
        # states again on the forced ctx after its give-up, at two sizes
        assert np.array_equal(lim.is_valid_batch(w.soa, w.layout)["flags"], ref)
        assert np.array_equal(lim.is_valid_batch(w.soa[:, :700], w.layout)["flags"], ref[:700])
    finally:
        for e in (two, lim, plain, e_two, e_lim):
            e.close()
