"""The no-allocation entry point mvk_is_valid_mask_noalloc_dev against mvk_is_valid_mask_dev (same kernels, hand-off
buffers in the library's static __device__ array instead of ctx-owned hipMalloc memory): same mask bits, on two calls
in a row -- the second with a smaller input, so it meets the first call's leftovers in the shared array -- and on a
second context of another robot; sizes beyond the stated maximum are refused.  Collected last (file name): it must pass
alone in a fresh process and after the other tests."""
import numpy as np
import pytest

import common
import motion_planning_amd as mp
import oracle_lib as O
from motion_planning_amd import _abi as A

pytestmark = pytest.mark.gpu


def masks(eng, w, n, dev):
    import torch
    d = torch.from_numpy(np.ascontiguousarray(w.soa[:, :n])).to(dev)
    words = (n + 63) // 64
    ref = torch.zeros(words, dtype=torch.int64, device=dev)
    got = torch.full((words,), -1, dtype=torch.int64, device=dev)
    eng.is_valid_mask_dev(d, w.layout, ref)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    eng.is_valid_mask_noalloc_dev(d, w.layout, got)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0          # the device's free memory did not move
    return ref.cpu().numpy(), got.cpu().numpy()


def test_noalloc_entry_point_equals_the_allocating_one():
    import torch
    dev = torch.device("cuda", 0)
    w = common.config2k(n=A.NOALLOC_MAX_STATES)
    eng = mp.Engine(w.model, w.boxes, w.params)
    # first call: the largest batch the entry point takes; second call: a smaller, ragged one
    for n in (A.NOALLOC_MAX_STATES, 9000 + 37):
        ref, got = masks(eng, w, n, dev)
        assert np.array_equal(ref, got), n
        bits = np.unpackbits(got.view(np.uint8), bitorder="little")[:n]
        assert 0.02 < bits.mean() < 0.98
        if n < 20000:   # and against the oracle
            o = O.eval_batch(w.model, w.boxes, w.params, w.soa[:, :n], w.layout, threads=8)
            assert not np.any((bits != o["flags"]) & ~o["band"].astype(bool))
    # another context, another robot, same GPU: it shares the array and finds the first one's leftovers in it
    w3 = common.config3(n=5000)
    eng3 = mp.Engine(w3.model, w3.boxes, w3.params)
    ref, got = masks(eng3, w3, 5000, dev)
    assert np.array_equal(ref, got)
    ref, got = masks(eng, w, 777, dev)
    assert np.array_equal(ref, got)
    # beyond the maximum: refused with a clear error, nothing launched
    big = common.config2k(n=A.NOALLOC_MAX_STATES + 1)
    d = torch.from_numpy(big.soa).to(dev)
    m = torch.zeros((big.soa.shape[1] + 63) // 64, dtype=torch.int64, device=dev)
    with pytest.raises(mp.MvkError) as e:
        eng.is_valid_mask_noalloc_dev(d, big.layout, m)
    assert "at most 65536" in str(e.value)
    assert int(m.abs().sum()) == 0
    eng.close()
    eng3.close()
