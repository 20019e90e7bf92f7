"""The C-ABI multi-GPU exchange (csrc/mvk_dist.cpp): shard rule and statistics merge on the host
(no GPU), and -- marked gpu -- a communicator of one rank over RCCL: the collectives must be the identity there and
the gathered count must equal mvk_mask_count_dev.  More ranks cannot share this box's single GPU (RCCL refuses two
ranks on one device); the N > 1 run is the driver's bench.py --gpus N, which issues its exchange through these calls."""
import ctypes as C

import numpy as np
import pytest

import common
import motion_planning_amd as mp
from motion_planning_amd import _abi as A
from motion_planning_amd import dist as D


def c_shard(n, rank, world):
    lib = A.load()
    lo, hi, per = C.c_size_t(), C.c_size_t(), C.c_size_t()
    assert lib.mvk_shard_range(n, rank, world, C.byref(lo), C.byref(hi), C.byref(per)) == 0
    return lo.value, hi.value, per.value


def test_shard_range_is_the_python_layers_rule():
    for n in (0, 1, 63, 64, 65, 1000, 1 << 20, (1 << 26) + 17):
        for world in (1, 2, 3, 8):
            cover = 0
            for r in range(world):
                lo, hi, per = c_shard(n, r, world)
                assert (lo, hi) == D.shard_range(n, r, world) and per == D.shard_words(n, world)
                assert (lo % 64 == 0 or lo == hi) and hi - lo <= per * 64
                cover += hi - lo
            assert cover == n
    assert A.load().mvk_shard_range(10, 2, 2, None, None, None) == A.ERR_INVALID_ARG


def stats(**kw):
    s = A.Stats()
    for k, _ in A.Stats._fields_:
        setattr(s, k, -1 if k.endswith("_index") else 0)
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def test_merge_stats_sums_counts_and_keeps_global_indices():
    lib = A.load()
    loc = (A.Stats * 3)(
        stats(n=100, valid_count=40, band_count=1, min_dist=0.5, min_dist_index=7, min_margin=-2.0, min_margin_index=3,
              max_thrust=17.0, max_thrust_index=9, min_thrust=4.0, min_thrust_index=1, min_rp_position_margin=0.3,
              min_rp_position_margin_index=0),
        stats(n=100, valid_count=10, band_count=0, min_dist=0.25, min_dist_index=5, min_margin=-2.0, min_margin_index=0,
              max_thrust=17.0, max_thrust_index=2, min_thrust=5.0, min_thrust_index=4, min_rp_position_margin=0.3,
              min_rp_position_margin_index=99),
        stats(n=0))                                        # a rank with an empty shard: every index -1
    off = (C.c_uint64 * 3)(0, 100, 200)
    g = A.Stats()
    assert lib.mvk_merge_stats(loc, off, 3, C.byref(g)) == 0
    assert (g.n, g.valid_count, g.band_count) == (200, 50, 1)
    assert (g.min_dist, g.min_dist_index) == (0.25, 105)
    assert (g.min_margin, g.min_margin_index) == (-2.0, 3)            # tie: the smallest global index
    assert (g.max_thrust, g.max_thrust_index) == (17.0, 9)
    assert (g.min_thrust, g.min_thrust_index) == (4.0, 1)
    assert (g.min_rp_position_margin, g.min_rp_position_margin_index) == (0.3, 0)
    empty = (A.Stats * 1)(stats(n=0))
    assert lib.mvk_merge_stats(empty, (C.c_uint64 * 1)(0), 1, C.byref(g)) == 0
    assert g.n == 0 and g.min_dist_index == -1 and g.max_thrust_index == -1


@pytest.mark.gpu
def test_communicator_of_one_rank_over_rccl():
    import torch
    lib = A.load()
    dev = torch.device("cuda", 0)
    ident = (C.c_ubyte * A.COMM_ID_BYTES)()
    assert lib.mvk_comm_unique_id(ident) == 0, lib.mvk_last_error()
    comm = C.c_void_p()
    assert lib.mvk_comm_create(ident, 0, 1, 0, C.byref(comm)) == 0, lib.mvk_last_error()
    assert lib.mvk_comm_rank(comm) == 0 and lib.mvk_comm_world(comm) == 1
    try:
        n = 50000 + 21
        w = common.config2k(n=n)
        eng = mp.Engine(w.model, w.boxes, w.params)
        lo, hi, per = c_shard(n, 0, 1)
        assert (lo, hi) == (0, n)
        d = torch.from_numpy(w.soa).to(dev)
        mask = torch.zeros(per, dtype=torch.int64, device=dev)
        eng.is_valid_mask_dev(d, w.layout, mask)
        full = torch.full((per,), -1, dtype=torch.int64, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        assert lib.mvk_gather_mask_dev(comm, mask.data_ptr(), per, full.data_ptr(), s) == 0, lib.mvk_last_error()
        cnt = torch.zeros(2, dtype=torch.int64, device=dev)
        assert lib.mvk_mask_count_dev(full.data_ptr(), n, cnt.data_ptr(), s) == 0
        assert lib.mvk_allreduce_count_dev(comm, cnt.data_ptr(), 1, s) == 0, lib.mvk_last_error()
        torch.cuda.synchronize()
        assert torch.equal(full, mask)
        flags = eng.is_valid_batch(w.soa, w.layout)["flags"]
        assert int(cnt[0]) == int(flags.sum()) and 0 < int(cnt[0]) < n
        # the one-call form of the step's exchange: event on the producer stream, all-gather + count on another
        side = torch.cuda.Stream(device=dev)
        for slot in (0, 3, 0):
            full.fill_(-1)
            cnt.zero_()
            eng.is_valid_mask_dev(d, w.layout, mask)
            assert lib.mvk_exchange_masks_dev(comm, slot, mask.data_ptr(), per, full.data_ptr(), cnt.data_ptr(), s, side.cuda_stream) == 0, lib.mvk_last_error()
            assert lib.mvk_exchange_wait(comm, slot) == 0
            assert torch.equal(full, mask) and int(cnt[0]) == int(flags.sum())
        assert lib.mvk_exchange_wait(comm, -1) == 0
        assert lib.mvk_exchange_masks_dev(comm, 8, mask.data_ptr(), per, full.data_ptr(), None, s, s) == A.ERR_INVALID_ARG
        # statistics of a full evaluation: the exchange of one rank returns them with the offset applied
        eng.eval_batch(w.soa[:, :4096], w.layout)
        loc = A.Stats()
        assert lib.mvk_reduce_stats(eng.ctx, C.byref(loc)) == 0
        g = A.Stats()
        assert lib.mvk_allreduce_stats(comm, C.byref(loc), 1000, C.byref(g), s) == 0, lib.mvk_last_error()
        for k, _ in A.Stats._fields_:
            a, b = getattr(loc, k), getattr(g, k)
            assert b == (a + 1000 if k.endswith("_index") and a >= 0 else a), k
        eng.close()
    finally:
        assert lib.mvk_comm_destroy(comm) == 0
    # argument errors do not reach RCCL
    assert lib.mvk_comm_create(ident, 1, 1, 0, C.byref(comm)) == A.ERR_INVALID_ARG
    assert lib.mvk_comm_create(ident, 0, 1, 99, C.byref(comm)) == A.ERR_NO_DEVICE


@pytest.mark.gpu
def test_a_communicator_per_pipe_lane_gathers_on_the_lanes_streams():
    """bench.py --gpus N: every rank runs a pipe (include/mvk.h mvk_pipe_*), a batch's all-gather follows it on its lane's
    own stream through the lane's own communicator -- two communicators alive side by side, collectives of either issued
    while the other lane computes.  With one rank the gathered words are the lane's mask; eight batches, in order."""
    import torch
    lib = A.load()
    dev = torch.device("cuda", 0)
    comms = []
    for _ in range(2):
        ident = (C.c_ubyte * A.COMM_ID_BYTES)()
        assert lib.mvk_comm_unique_id(ident) == 0, lib.mvk_last_error()
        c = C.c_void_p()
        assert lib.mvk_comm_create(ident, 0, 1, 0, C.byref(c)) == 0, lib.mvk_last_error()
        comms.append(c)
    try:
        n = 1 << 16
        w = common.config2k(n=n)
        words = n // 64
        rng = np.random.default_rng(9)
        batches = [torch.from_numpy(np.ascontiguousarray(w.soa[:, rng.permutation(n)])).to(dev) for _ in range(8)]
        eng = mp.Engine(w.model, w.boxes, w.params)
        want = []
        for b in batches:
            m = torch.zeros(words, dtype=torch.int64, device=dev)
            eng.is_valid_mask_dev(b, w.layout, m)
            torch.cuda.synchronize()
            want.append(m.clone())
        eng.close()
        pipe = mp.Pipe(w.model, w.boxes, w.params, depth=2)
        streams = [C.c_void_p(lib.mvk_pipe_stream(pipe.pipe, lane)) for lane in range(2)]
        assert streams[0].value and streams[1].value and streams[0].value != streams[1].value
        masks = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(4)]     # four buffer sets, reused
        alls = [torch.full((words,), -1, dtype=torch.int64, device=dev) for _ in range(8)]
        for k, b in enumerate(batches):
            lane = pipe.is_valid_mask(b, w.layout, masks[k % 4])
            assert lane == k % 2
            assert lib.mvk_gather_mask_dev(comms[lane], masks[k % 4].data_ptr(), words, alls[k].data_ptr(), streams[lane]) == 0, lib.mvk_last_error()
        pipe.drain()
        for k in range(8):
            assert torch.equal(alls[k], want[k]), k
        pipe.close()
    finally:
        for c in comms:
            assert lib.mvk_comm_destroy(c) == 0
