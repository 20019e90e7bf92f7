"""BASELINE.json's configurations at their FULL sizes, through size-independent properties (the oracle does not reach them):

config 5 -- 2^26 samples of the HYDRUS-8 joint space generated in the kernel, 2^23 per GPU on eight GPUs: every shard in
one launch equals its eight 2^20-sample launches word for word, a shard through a two-context pipe equals the shard
through one ctx, the valid counts of the shards add up to the count of the gathered mask, and a slice of every shard
agrees with the oracle on the materialised states;
config 2 / 3 -- 2^20 samples: a batch equals the concatenation of its halves, the mask equals the flags;
config 4 -- 65,536 edges x 64 steps: the batch equals its quarters, a sample of the edges against the oracle.

Reference: MotionPlanning::isStateValid (motion_planning.cpp:76-164); the sweep mirrors
sampling_based_method/scripts/state_validation_l4.py:46-68."""
import numpy as np
import pytest
import torch

import common
import motion_planning_amd as mp
import oracle_lib as O
from motion_planning_amd import _abi as A

pytestmark = pytest.mark.gpu


def test_config5_full_sweep_in_shards():
    w = common.config5(n=8)
    eng = mp.Engine(w.model, w.boxes, w.params)
    pipe = mp.Pipe(w.model, w.boxes, w.params, depth=2)
    pose = np.array([0, 0, 0, 0, 0, 0, 1.0])
    seed = common.SEED0 + 5
    shard, part, world = 1 << 23, 1 << 20, 8
    big = torch.zeros(shard // 64, dtype=torch.int64, device="cuda")
    small = torch.zeros(shard // 64, dtype=torch.int64, device="cuda")
    piped = torch.zeros(shard // 64, dtype=torch.int64, device="cuda")
    gathered = torch.zeros(world * shard // 64, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    total = 0
    for r in range(world):
        first = r * shard
        eng.sweep_mask_dev(pose, first, shard, seed, big)
        for k in range(shard // part):
            eng.sweep_mask_dev(pose, first + k * part, part, seed, small[k * part // 64:(k + 1) * part // 64])
        for k in range(shard // part):   # (alternating lanes)
            pipe.sweep_mask(pose, first + k * part, part, seed, piped[k * part // 64:(k + 1) * part // 64])
        pipe.drain()
        torch.cuda.synchronize()
        assert torch.equal(big, small), r
        assert torch.equal(big, piped), r
        eng.mask_count_dev(big, shard, cnt)
        total += int(cnt.item())
        gathered[r * shard // 64:(r + 1) * shard // 64] = big
        # a slice of the shard against the oracle on the materialised states
        k = 2048
        st = eng.sweep_states(pose, first + 12345, k, seed)
        o = O.eval_batch(w.model, w.boxes, w.params, st, A.LAYOUT_POSE7, threads=8)
        m = torch.zeros(k // 64, dtype=torch.int64, device="cuda")
        eng.sweep_mask_dev(pose, first + 12345, k, seed, m)
        torch.cuda.synchronize()
        bits = np.unpackbits(m.cpu().numpy().view(np.uint8), bitorder="little")[:k]
        assert not np.any((bits != o["flags"]) & ~o["band"].astype(bool)), r
    eng.mask_count_dev(gathered, world * shard, cnt)
    assert int(cnt.item()) == total and 0 < total < world * shard
    print(f"config 5, 2^26 samples: {total} valid ({total / (world * shard):.4f})")
    pipe.close()
    eng.close()


@pytest.mark.parametrize("make", [common.config2, common.config3], ids=["c2", "c3"])
def test_full_batch_equals_its_halves(make):
    n = 1 << 20
    w = make(n=n)
    eng = mp.Engine(w.model, w.boxes, w.params)
    d = torch.from_numpy(w.soa).cuda()
    whole = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
    halves = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
    flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
    eng.is_valid_mask_dev(d, w.layout, whole)
    h = n // 2
    eng.is_valid_mask_dev(d[:, :h], w.layout, halves[:h // 64])
    eng.is_valid_mask_dev(d[:, h:], w.layout, halves[h // 64:])
    eng.is_valid_batch_dev(d, w.layout, d_flags=flags)
    torch.cuda.synchronize()
    assert torch.equal(whole, halves)
    bits = np.unpackbits(whole.cpu().numpy().view(np.uint8), bitorder="little")
    assert np.array_equal(bits, flags.cpu().numpy())
    eng.close()


def test_config4_full_edge_batch_equals_its_parts():
    """65,536 edges x 64 steps: the batch equals its four quarters edge for edge (verdict and first invalid step), and a
    sample of the edges agrees with the oracle's discrete motion validator"""
    model, params, boxes, s1, s2, nd = common.config4(n_edges=1 << 16)
    E = s1.shape[1]
    eng = mp.Engine(model, boxes, params)
    d1, d2 = torch.from_numpy(s1).cuda(), torch.from_numpy(s2).cuda()
    ok = torch.zeros(E, dtype=torch.uint8, device="cuda")
    first = torch.zeros(E, dtype=torch.int32, device="cuda")
    okq = torch.zeros(E, dtype=torch.uint8, device="cuda")
    firstq = torch.zeros(E, dtype=torch.int32, device="cuda")
    eng.check_motion_batch_dev(d1, d2, nd, ok, first)
    q = E // 4
    for k in range(4):
        eng.check_motion_batch_dev(d1[:, k * q:(k + 1) * q], d2[:, k * q:(k + 1) * q], nd, okq[k * q:(k + 1) * q], firstq[k * q:(k + 1) * q])
    torch.cuda.synchronize()
    assert torch.equal(ok, okq) and torch.equal(first, firstq)
    idx = np.linspace(0, E - 1, 256).astype(int)
    nds = np.full(len(idx), nd, np.int32)
    g_ok, g_first, g_band = O.check_motion_batch(model, boxes, params, np.ascontiguousarray(s1[:, idx]), np.ascontiguousarray(s2[:, idx]), nds)
    h_ok, h_first = ok.cpu().numpy()[idx], first.cpu().numpy()[idx]
    diff = (h_ok != g_ok) | (h_first != g_first)
    assert not np.any(diff & ~g_band.astype(bool))
    assert 0 < int(ok.sum().item()) < E
    eng.close()
