#!/usr/bin/env python3
"""host cost per call of the pieces of bench.py's per-step exchange (a communicator of one rank)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import common, motion_planning_amd as mp
from motion_planning_amd import _abi as A
lib = A.load()
dev = torch.device("cuda", 0)
ident = (C.c_ubyte * 128)(); lib.mvk_comm_unique_id(ident)
comm = C.c_void_p(); assert lib.mvk_comm_create(ident, 0, 1, 0, C.byref(comm)) == 0
n = 1 << 20; words = n // 64
w = common.config2k(n=n); eng = mp.Engine(w.model, w.boxes, w.params)
d = torch.from_numpy(w.soa).to(dev)
mask = torch.zeros(words, dtype=torch.int64, device=dev); full = torch.zeros(words, dtype=torch.int64, device=dev); cnt = torch.zeros(1, dtype=torch.int64, device=dev)
side = torch.cuda.Stream(device=dev); s0 = torch.cuda.current_stream().cuda_stream
def t(fn, reps=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    dt = (time.perf_counter() - t0) / reps * 1e6; torch.cuda.synchronize(); return dt
print("is_valid_mask_dev issue      %.1f us" % t(lambda: eng.is_valid_mask_dev(d, w.layout, mask)))
print("mvk_gather_mask_dev issue    %.1f us" % t(lambda: lib.mvk_gather_mask_dev(comm, mask.data_ptr(), words, full.data_ptr(), side.cuda_stream)))
print("mvk_mask_count_dev issue     %.1f us" % t(lambda: lib.mvk_mask_count_dev(full.data_ptr(), n, cnt.data_ptr(), side.cuda_stream)))
print("mvk_exchange_masks_dev issue %.1f us" % t(lambda: lib.mvk_exchange_masks_dev(comm, 0, mask.data_ptr(), words, full.data_ptr(), cnt.data_ptr(), s0, side.cuda_stream)))
print("mvk_exchange_wait            %.1f us" % t(lambda: lib.mvk_exchange_wait(comm, 0)))
lib.mvk_comm_destroy(comm)
