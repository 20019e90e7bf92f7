"""Pipelined batches (include/mvk.h mvk_pipe_*, csrc/mvk_pipe.cpp): several contexts with a stream each, consecutive
batches handed to them round-robin, so that the flag path's exact stage of one batch runs beside the screen stage of
the next.  Per batch the result is that of the ctx entry point: same bits as one ctx, whatever is in flight beside it.

Reference: MotionPlanning::isStateValid (motion_planning.cpp:76-164) per state; the pipe belongs to the batched entry
point the north star adds and has no counterpart there."""
import numpy as np
import pytest

import common
import motion_planning_amd as mp
import oracle_lib as O
from motion_planning_amd import _abi as A

pytestmark = pytest.mark.gpu


def _bits(mask, n):
    return np.unpackbits(mask.cpu().numpy().view(np.uint8), bitorder="little")[:n]


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_batches_in_flight_give_the_bits_of_one_ctx(depth):
    import torch
    w = common.config2k(n=1 << 16)
    n = w.soa.shape[1]
    rng = np.random.default_rng(3)
    # eight different batches: the same states in another order, and sizes that differ
    perms = [rng.permutation(n)[: n - 1000 * k] for k in range(8)]
    batches = [torch.from_numpy(np.ascontiguousarray(w.soa[:, p])).cuda() for p in perms]
    eng = mp.Engine(w.model, w.boxes, w.params)
    want = []
    for b in batches:
        m = torch.zeros((b.shape[1] + 63) // 64, dtype=torch.int64, device="cuda")
        eng.is_valid_mask_dev(b, w.layout, m)
        torch.cuda.synchronize()
        want.append(_bits(m, b.shape[1]))
    eng.close()
    pipe = mp.Pipe(w.model, w.boxes, w.params, depth=depth)
    assert pipe.depth == depth
    masks = [torch.zeros((b.shape[1] + 63) // 64, dtype=torch.int64, device="cuda") for b in batches]
    tickets = [pipe.is_valid_mask(b, w.layout, m) for b, m in zip(batches, masks)]     # all eight in flight
    assert tickets == [k % depth for k in range(8)]
    pipe.drain()
    for k, (m, b) in enumerate(zip(masks, batches)):
        assert np.array_equal(_bits(m, b.shape[1]), want[k]), k
    # against the oracle as well, on the first batch
    ref = O.eval_batch(w.model, w.boxes, w.params, np.ascontiguousarray(w.soa[:, perms[0][:4000]]), w.layout, threads=8)
    band = ref["band"].astype(bool)
    assert not np.any((_bits(masks[0], n)[:4000] != ref["flags"]) & ~band)
    pipe.close()


def test_producer_and_consumer_streams_are_ordered_without_host_waits():
    import torch
    w = common.config2(n=1 << 15)
    n = w.soa.shape[1]
    src = torch.from_numpy(w.soa).cuda()
    eng = mp.Engine(w.model, w.boxes, w.params)
    m_ref = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    eng.is_valid_mask_dev(src, w.layout, m_ref)
    torch.cuda.synchronize()
    eng.close()
    pipe = mp.Pipe(w.model, w.boxes, w.params, depth=2)
    producer, consumer = torch.cuda.Stream(), torch.cuda.Stream()
    staged = torch.zeros_like(src)
    out = torch.zeros_like(m_ref)
    for rep in range(6):
        with torch.cuda.stream(producer):
            staged.zero_()
            for _ in range(20):          # keep the producer busy: the batch must not start before the copy lands
                staged.add_(0.0)
            staged.copy_(src, non_blocking=True)
        m = torch.zeros_like(m_ref)
        t = pipe.is_valid_mask(staged, w.layout, m, producer_stream=producer)
        pipe.wait(t, consumer_stream=consumer)              # GPU-side: the consumer's copy runs behind the batch
        with torch.cuda.stream(consumer):
            out.copy_(m, non_blocking=True)
        consumer.synchronize()
        assert torch.equal(out, m_ref), rep
        producer.wait_stream(consumer)                      # (staged is rewritten next round)
    pipe.close()


def test_full_path_and_sweep_through_a_pipe_and_errors():
    import torch
    w = common.config5(n=8192)
    n = w.soa.shape[1]
    pose = np.array([0, 0, 0, 0, 0, 0, 1.0])
    eng = mp.Engine(w.model, w.boxes, w.params)
    m_ref = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    eng.sweep_mask_dev(pose, 0, n, common.SEED0 + 5, m_ref)
    d = torch.from_numpy(w.soa).cuda()
    ref = eng.eval_batch(w.soa, w.layout)
    torch.cuda.synchronize()
    eng.close()
    pipe = mp.Pipe(w.model, w.boxes, w.params, depth=2)
    m = torch.zeros_like(m_ref)
    md = torch.zeros(n, dtype=torch.float64, device="cuda")
    mg = torch.zeros(n, dtype=torch.float64, device="cuda")
    fl = torch.zeros(n, dtype=torch.uint8, device="cuda")
    t0 = pipe.sweep_mask(pose, 0, n, common.SEED0 + 5, m)
    t1 = pipe.eval_batch(d, w.layout, d_flags=fl, d_min_dist=md, d_margin=mg)
    assert (t0, t1) == (0, 1)
    pipe.wait(t0)
    pipe.wait(t1)
    assert torch.equal(m, m_ref)
    assert np.array_equal(md.cpu().numpy(), ref["min_dist"], equal_nan=True) and np.array_equal(mg.cpu().numpy(), ref["margin"])
    assert np.array_equal(fl.cpu().numpy(), ref["flags"])
    moved = w.boxes.copy()
    moved[:, 2] += 50.0                                    # the scene out of the way: every stable state is valid
    pipe.set_scene(moved)
    pipe.eval_batch(d, w.layout, d_flags=fl)
    pipe.drain()
    assert fl.cpu().numpy().sum() > ref["flags"].sum()
    with pytest.raises(mp.MvkError):
        pipe.wait(5)
    pipe.close()
    with pytest.raises(mp.MvkError):
        mp.Pipe(w.model, w.boxes, w.params, depth=0)
    with pytest.raises(mp.MvkError):
        mp.Pipe(w.model, w.boxes, w.params, depth=9)


def test_pipe_lanes_run_side_by_side():
    """mvk_pipe_create probes its lanes with a one-wave spin kernel and replaces a stream that shares a hardware queue
    with another lane (csrc/mvk_pipe.cpp lanes_overlap): after creation every pair of lanes runs side by side, also
    when the process holds other streams and earlier pipes have come and gone"""
    import torch
    lib = A.load()
    w = common.config2(n=64, n_boxes=10)
    others = [torch.cuda.Stream() for _ in range(5)]           # other users of the runtime's queues
    apart, shared = 0, []
    for cycle in range(4):
        for depth in (2, 3):
            pipe = mp.Pipe(w.model, w.boxes, w.params, depth=depth)
            assert lib.mvk_debug_pipe_restreamed(pipe.pipe) >= 0
            for a in range(depth):
                for b in range(a + 1, depth):
                    v = lib.mvk_debug_pipe_lanes_overlap(pipe.pipe, a, b)
                    assert v in (0, 1)
                    apart += v
                    if v == 0:
                        shared.append((cycle, depth, a, b, lib.mvk_debug_pipe_restreamed(pipe.pipe)))
            assert lib.mvk_debug_pipe_lanes_overlap(pipe.pipe, 0, 0) == -1 and lib.mvk_debug_pipe_lanes_overlap(pipe.pipe, 0, depth) == -1
            pipe.close()
    del others
    assert lib.mvk_debug_pipe_restreamed(None) == -1
    assert apart > 0
    if shared:   # (a timing probe on a shared machine: reported, not a failure of the library's results)
        pytest.xfail(f"lanes still behind one hardware queue after the replacements: {shared}")
