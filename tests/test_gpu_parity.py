"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): validity flags bit-exact except samples inside the 1e-6 boundary band
(reported separately); min distance and stability margin within 1e-5 relative in fp64.
"""
import numpy as np
import pytest

import common
import motion_planning_amd as mp
import oracle_lib as O
from motion_planning_amd import _abi as A

pytestmark = pytest.mark.gpu

REL = 1e-5   # north_star tolerance on distances / margins (fp64)
ABS = 1e-9   # absolute floor for values that are themselves ~0


def _close(a, b, rel=REL, abs_=ABS):
    return np.abs(a - b) <= abs_ + rel * np.maximum(np.abs(a), np.abs(b))


def check_against_oracle(w, n, threads=8):
    soa = w.soa[:, :n]
    ref = O.eval_batch(w.model, w.boxes, w.params, soa, w.layout, threads=threads)
    eng = mp.Engine(w.model, w.boxes, w.params)
    fast = eng.is_valid_batch(soa, w.layout, want=("flags",))["flags"]
    full = eng.eval_batch(soa, w.layout)
    stats = eng.reduce_stats()
    eng.close()

    band = (ref["band"] | full["band"]).astype(bool)
    # 1. flags: bit exact outside the band, for both the flag-only and the full kernel
    for name, flags in (("fast", fast), ("full", full["flags"])):
        diff = flags != ref["flags"]
        assert not np.any(diff & ~band), f"{w.name}/{name}: {np.sum(diff & ~band)} flag mismatches outside the band"
    # 2. distances and margins
    fin = np.isfinite(ref["min_dist"])
    assert np.array_equal(fin, np.isfinite(full["min_dist"]))
    assert np.all(_close(full["min_dist"][fin], ref["min_dist"][fin])), "min_dist"
    well = ref["pivot_ratio"] > 1e-9   # static-thrust normal matrix not near singular
    assert np.all(_close(full["margin"][well], ref["margin"][well])), "margin"
    assert np.all(_close(full["fc_t_min"], ref["fc_t_min"])), "fc_t_min"
    assert np.all(_close(full["fc_rp_min"], ref["fc_rp_min"])), "fc_rp_min"
    assert np.all(_close(full["thrust"][:, well], ref["thrust"][:, well], abs_=1e-7)), "thrust"
    assert np.all(_close(full["rp_position_margin"], ref["rp_position_margin"], abs_=1e-7)), "rp_position_margin"
    assert np.all(_close(full["cog"], ref["cog"])), "cog"
    if w.model.has_gimbal:
        assert np.all(_close(full["gimbal"], ref["gimbal"], abs_=1e-7)), "gimbal"
    # 3. device-side reduction == reduction of the per-sample outputs
    assert stats["n"] == n and stats["valid_count"] == int(full["flags"].sum())
    assert stats["band_count"] == int(full["band"].sum())
    assert stats["min_dist"] == full["min_dist"][fin].min() and stats["min_dist_index"] == int(np.nanargmin(np.where(fin, full["min_dist"], np.inf)))
    return dict(n=n, valid=int(ref["flags"].sum()), band=int(band.sum()), mismatch_in_band=int((fast != ref["flags"]).sum()))


def test_c2_hydrus6_se2_vs_oracle():
    r = check_against_oracle(common.config2(n=20000), 20000)
    print("c2:", r)
    assert 0.05 < r["valid"] / r["n"] < 0.95


def test_c2_hydrus6_se3_vs_oracle():
    r = check_against_oracle(common.config2(n=10000, orient="so3"), 10000)
    print("c2/se3:", r)


def test_c3_dragon4_vs_oracle():
    r = check_against_oracle(common.config3(n=4000), 4000)
    print("c3:", r)


def test_c5_hydrus8_vs_oracle():
    r = check_against_oracle(common.config5(n=3000), 3000)
    print("c5:", r)


def test_hydrus4_vs_oracle():
    w = common.config2(n=8000, n_links=4, n_boxes=40)
    r = check_against_oracle(w, 8000)
    print("hydrus4:", r)
