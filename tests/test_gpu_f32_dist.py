"""The fp32 variant of the distance and margin outputs (mvk_eval_batch_f32_dev, include/mvk.h; BASELINE.json north_star:
"distances and margins must agree within 1e-5 relative in fp64 (fp32 variant reported with its own tolerance)").

Distances: the narrowphase of distance_kernel in fp32 (separating-axis gaps, edge-to-box and segment-to-box distances
on the fp32 copy of the boxes).  TOLERANCE, stated in mvk.h, per state: |d32 - d64| <= mvk_f32_distance_tolerance(ctx, X) =
64 * 2^-24 * (X + reach + 1 m), X the largest |coordinate| of the root position; the largest error seen is printed and
asserted to stay within a quarter of it.
Margins: an enclosure [m - e, m + e] of the fp64 margin from the fp32 walk with running error bounds: the tolerance is
per state (e), and the fp64 margin -- the kernels' and the oracle's -- must lie inside every finite one.
Flags: equal to the fp64 flags outside the variant's own band.

Reference: MotionPlanning::isStateValid (motion_planning.cpp:76-164, all fp64) and the minimum distance / margins of
addState's statistics (motion_planning.cpp:546-642)."""
import numpy as np
import pytest

import common
import motion_planning_amd as mp
import oracle_lib as O

pytestmark = pytest.mark.gpu


def run_both(w, n):
    import torch
    eng = mp.Engine(w.model, w.boxes, w.params)
    soa = np.ascontiguousarray(w.soa[:, :n])
    ref = eng.eval_batch(soa, w.layout)
    d = torch.from_numpy(soa).cuda()
    f = dict(flags=torch.zeros(n, dtype=torch.uint8, device="cuda"), band=torch.zeros(n, dtype=torch.uint8, device="cuda"),
             min_dist=torch.zeros(n, dtype=torch.float32, device="cuda"), margin=torch.zeros(n, dtype=torch.float32, device="cuda"),
             margin_err=torch.zeros(n, dtype=torch.float32, device="cuda"))
    eng.eval_batch_f32_dev(d, w.layout, d_flags=f["flags"], d_min_dist=f["min_dist"], d_margin=f["margin"],
                           d_margin_err=f["margin_err"], d_band=f["band"])
    torch.cuda.synchronize()
    tol = np.array([eng.f32_distance_tolerance(x) for x in np.abs(soa[0:3]).max(0)])
    eng.close()
    return ref, {k: v.cpu().numpy() for k, v in f.items()}, tol


@pytest.mark.parametrize("make,n", [(common.config2k, 20000), (common.config3, 12000), (common.config2, 12000), (common.config5, 8000),
                                    (lambda n: common.config2(n=n, orient="so3"), 8000), (common.config2k, 300),
                                    (common.config3, 1 << 18)],   # (large: contacts between nearly parallel edges, whose depth fp32 cannot resolve, turn up)
                         ids=["c2k", "c3", "c2", "c5", "c2se3", "c2k_small", "c3_262k"])
def test_f32_distances_and_margins_within_their_tolerance(make, n):
    w = make(n=n)
    ref, f32, tol = run_both(w, n)
    fin = np.isfinite(ref["min_dist"])
    assert fin.mean() > 0.99
    err = np.abs(f32["min_dist"][fin].astype(np.float64) - ref["min_dist"][fin])
    print(f"fp32 distances: max |d32 - d64| = {err.max():.3g} m, mean {err.mean():.3g}; tolerance {tol[fin].min():.3g} .. {tol[fin].max():.3g} m; "
          f"largest error / tolerance {np.max(err / tol[fin]):.3f}")
    assert np.all(err <= tol[fin])
    assert np.max(err / tol[fin]) < 0.25          # (observed: a few fp32 roundings, well inside the stated 64)
    # the enclosure of the margin holds the fp64 margin wherever it is finite
    e = f32["margin_err"].astype(np.float64)
    m = f32["margin"].astype(np.float64)
    ok = np.isfinite(e) & np.isfinite(ref["margin"])
    assert ok.mean() > 0.6, ok.mean()
    slack = 4e-7 * np.maximum(np.abs(ref["margin"][ok]), 1.0)       # the midpoint and half width are themselves floats
    assert np.all(np.abs(m[ok] - ref["margin"][ok]) <= e[ok] + slack), np.max(np.abs(m[ok] - ref["margin"][ok]) - e[ok])
    print(f"fp32 margins: enclosure finite for {ok.mean():.3f} of the states, median half width {np.median(e[ok]):.3g}, "
          f"max |m32 - m64| {np.abs(m[ok] - ref['margin'][ok]).max():.3g}")
    # verdicts: the fp64 flags outside the variant's band
    band = f32["band"].astype(bool)
    assert not np.any((f32["flags"] != ref["flags"]) & ~band), int(np.sum((f32["flags"] != ref["flags"]) & ~band))
    assert band.mean() < 0.5


def test_f32_distances_against_the_oracle():
    w = common.config2k(n=3000)
    ref = O.eval_batch(w.model, w.boxes, w.params, w.soa, w.layout, threads=8)
    _, f32, tol = run_both(w, 3000)
    fin = np.isfinite(ref["min_dist"])
    assert np.all(np.abs(f32["min_dist"][fin].astype(np.float64) - ref["min_dist"][fin]) <= tol[fin] + 1e-9)
    e, m = f32["margin_err"].astype(np.float64), f32["margin"].astype(np.float64)
    ok = np.isfinite(e) & np.isfinite(ref["margin"])
    assert np.all(np.abs(m[ok] - ref["margin"][ok]) <= e[ok] + 4e-7 * np.maximum(np.abs(ref["margin"][ok]), 1.0))
    band = f32["band"].astype(bool) | ref["band"].astype(bool)
    assert not np.any((f32["flags"] != ref["flags"]) & ~band)


def test_f32_variant_edge_cases():
    import torch
    w = common.config2(n=512, n_boxes=40)          # no distance grid (< 64 boxes): the fp64 narrowphase, stored as floats
    ref, f32, tol = run_both(w, 512)
    fin = np.isfinite(ref["min_dist"])
    assert np.all(np.abs(f32["min_dist"][fin].astype(np.float64) - ref["min_dist"][fin]) <= 1e-6 * np.maximum(1.0, np.abs(ref["min_dist"][fin])))
    w = common.config2k(n=256)
    soa = w.soa.copy()
    soa[2, 5] = np.nan                              # a non-finite state: invalid, in the band, as in the fp64 path
    eng = mp.Engine(w.model, w.boxes, w.params)
    d = torch.from_numpy(soa).cuda()
    fl = torch.ones(256, dtype=torch.uint8, device="cuda")
    bd = torch.zeros(256, dtype=torch.uint8, device="cuda")
    md = torch.zeros(256, dtype=torch.float32, device="cuda")
    eng.eval_batch_f32_dev(d, w.layout, d_flags=fl, d_min_dist=md, d_band=bd)     # (margin outputs left out)
    eng.eval_batch_f32_dev(d[:, :0], w.layout)                                     # empty batch
    torch.cuda.synchronize()
    eng.close()
    assert fl[5].item() == 0 and bd[5].item() == 1 and np.isnan(md[5].item())
