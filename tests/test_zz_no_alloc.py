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


def test_noalloc_call_captured_into_a_graph_replays():
    """the entry point is documented for code under HIP graph capture.  A host call (the ctx's own stream)
    comes first, then the noalloc call is captured on another stream and the graph is launched twice with different
    inputs in the same device buffers: same masks as the allocating entry point, and the ctx still works afterwards
    (the ordering state must not end up pointing into the capture)."""
    import torch
    dev = torch.device("cuda", 0)
    n = 20000 + 13
    w = common.config2k(n=2 * n)
    eng = mp.Engine(w.model, w.boxes, w.params)
    words = (n + 63) // 64
    d_in = torch.from_numpy(np.ascontiguousarray(w.soa[:, :n])).to(dev)
    d_mask = torch.zeros(words, dtype=torch.int64, device=dev)
    ref = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(2)]
    for k in range(2):
        eng.is_valid_mask_dev(torch.from_numpy(np.ascontiguousarray(w.soa[:, k * n:(k + 1) * n])).to(dev), w.layout, ref[k])
    side = torch.cuda.Stream(device=dev)
    # an asynchronous call still "in flight" for all the ctx knows (another stream, nobody waited): the capture of the
    # next call must be refused cleanly -- before the capture begins here, so the refusal costs no capture
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        g0 = torch.cuda.CUDAGraph()
        g0.capture_begin(capture_error_mode="thread_local")
        try:
            with pytest.raises(mp.MvkError) as e:
                eng.is_valid_mask_noalloc_dev(d_in, w.layout, d_mask, stream=side)
            assert "graph capture" in str(e.value)
        finally:
            g0.capture_end()
    del g0
    host_flags = eng.is_valid_batch(w.soa[:, :512], w.layout)["flags"]      # a blocking host call: the ctx's own stream
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        g.capture_begin(capture_error_mode="thread_local")
        eng.is_valid_mask_noalloc_dev(d_in, w.layout, d_mask, stream=side)
        g.capture_end()
    assert int(d_mask.abs().sum()) == 0                # captured, not run
    for k in range(2):
        d_in.copy_(torch.from_numpy(np.ascontiguousarray(w.soa[:, k * n:(k + 1) * n])))
        d_mask.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(d_mask, ref[k]), k
    # the ctx is not wedged: host and device calls on other streams still work and agree
    assert np.array_equal(eng.is_valid_batch(w.soa[:, :512], w.layout)["flags"], host_flags)
    again = torch.zeros(words, dtype=torch.int64, device=dev)
    eng.is_valid_mask_dev(d_in, w.layout, again)
    torch.cuda.synchronize()
    assert torch.equal(again, ref[1])
    eng.close()


def test_ctx_survives_a_destroyed_caller_stream():
    """the previous call's stream belongs to the caller and may be gone when the next call arrives; the
    ordering logic must look before it records anything there, and the ctx must keep working."""
    import torch
    dev = torch.device("cuda", 0)
    n = 4096
    w = common.config2(n=n)
    eng = mp.Engine(w.model, w.boxes, w.params)
    d = torch.from_numpy(w.soa).to(dev)
    words = n // 64
    ref = torch.zeros(words, dtype=torch.int64, device=dev)
    eng.is_valid_mask_dev(d, w.layout, ref)
    torch.cuda.synchronize()
    hip = A.hip_runtime()
    s = A.C.c_void_p()
    assert hip.hipStreamCreate(A.C.byref(s)) == 0
    got = torch.zeros(words, dtype=torch.int64, device=dev)

    class _S:   # the shape Engine._stream expects
        cuda_stream = s.value
    eng.is_valid_mask_dev(d, w.layout, got, stream=_S)
    assert hip.hipStreamSynchronize(s) == 0
    assert hip.hipStreamDestroy(s) == 0
    assert torch.equal(got, ref)
    for _ in range(3):   # next calls: another stream, then the host path
        got.zero_()
        eng.is_valid_mask_dev(d, w.layout, got)
        torch.cuda.synchronize()
        assert torch.equal(got, ref)
        bits = np.unpackbits(ref.cpu().numpy().view(np.uint8), bitorder="little")[:256]
        assert np.array_equal(eng.is_valid_batch(w.soa[:, :256], w.layout)["flags"], bits)
    eng.close()
