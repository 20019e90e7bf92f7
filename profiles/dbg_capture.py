import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import common, motion_planning_amd as mp
from motion_planning_amd import _abi as A
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
torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "g0"
if mode == "g0":
    with torch.cuda.stream(side):
        g0 = torch.cuda.CUDAGraph()
        g0.capture_begin(capture_error_mode="thread_local")
        try:
            eng.is_valid_mask_noalloc_dev(d_in, w.layout, d_mask, stream=side)
        except mp.MvkError as e:
            print("refused:", str(e)[:80])
        g0.capture_end()
    del g0
host_flags = eng.is_valid_batch(w.soa[:, :512], w.layout)["flags"]
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    g.capture_begin(capture_error_mode="thread_local")
    eng.is_valid_mask_noalloc_dev(d_in, w.layout, d_mask, stream=side)
    g.capture_end()
print("after capture mask sum", int(d_mask.abs().sum()))
for rep in range(6):
    k = rep % 2
    d_in.copy_(torch.from_numpy(np.ascontiguousarray(w.soa[:, k * n:(k + 1) * n])))
    d_mask.zero_()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    x = (d_mask ^ ref[k]).cpu().numpy()
    bad = np.nonzero(x)[0]
    print("replay", rep, "k", k, "differing words", len(bad), bad[:8], [bin(int(v) & (2**64-1)).count("1") for v in x[bad[:8]]])
    # direct call for comparison
    d_mask.zero_()
    eng.is_valid_mask_noalloc_dev(d_in, w.layout, d_mask)
    torch.cuda.synchronize()
    print("   direct noalloc equal:", bool(torch.equal(d_mask, ref[k])))
eng.close()
