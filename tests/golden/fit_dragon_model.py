#!/usr/bin/env python3
"""Fits the synthetic DRAGON description to the per-state statics columns of the reference's planner logs.

Data: tests/golden/ref_dragon__*.txt and ref_dragon_log_tilt_45deg_gap_0p8.txt -- verbatim copies of files under
aerial_motion/sampling_based_method/path_log/dragon/.  Nine of them print, per state, what
motion_planning.cpp:615-631,639 logs: pose 7 + 6 link joints [+ 8 gimbal angles] + getRollPitchPositionMargin()
+ getStaticThrust().maxCoeff() -- 2,761 states of outputs of the absent aerial_robot_model / hydrus / dragon code.

    python tests/golden/fit_dragon_model.py            # geometry + margin (seconds)
    python tests/golden/fit_dragon_model.py --thrust   # + the static-thrust formulations tried (minutes)

RESULT 1 (closes to print precision; frozen in csrc/mvk_model.cpp, tests/test_pathlog_statics.py):
  rp_position_margin = sqrt(lambda_min(Cov_xy(p_i))) / |p_2 - p_1|   (p_i = rotor origins in the world-aligned CoG
  frame; the normaliser is the distance between the first two ROTORS, not the link length), with the chain
  link 0.424 -> pitch axis -> 0.0515 -> yaw axis, gimbal roll axis origin at mid-link (0.212), thrust-frame
  origin 0.037 up the roll frame's z axis.  Three numbers (rotor fraction a, pitch->yaw offset d1, roll-frame
  offset h) fitted on ONE log (498 states) come out as 0.5000 / 0.0515 / 0.0370 and reproduce the margin column of
  all nine logs with rms 4-6e-7 (the logs print 6 digits).  The same formula gives HYDRUS' minimum_var
  0.452857 @ state 200 (path_log/hydrus/hydrus_planning_log.txt) with the rotor at mid-link.

RESULT 2 (does NOT close): the max-static-thrust column.  Formulations tried, each with a 12-parameter mass
  model (relative link masses, inter-joint mass, link CoM of link 1 / links 2-4, gimbal-module offset, total weight)
  fitted on one log and validated on the others -- see thrust_table().  None generalises: held-out rms 0.3-1.3 N
  against a column that spans 18.4-24.2 N and is printed to 1e-4 N.  A 60-parameter model (weight + CoM of every
  link / inter-joint / gimbal roll / gimbal pitch frame) does not fit even its training log (0.08 N).  The thrust
  is the minimum-norm solution of a 3- or 4-row wrench balance about the CoG, and 1 mm of CoG moves it by 0.16 N:
  without the URDF's inertial tags the column cannot be reproduced; one log (vertial_gap_0p6_ceil_1p2) was
  written with a different mass revision altogether (its level square prints 20.7189 N, the others 20.5815 N).
"""
import glob
import os
import sys

import numpy as np
from scipy.optimize import least_squares

HERE = os.path.dirname(os.path.abspath(__file__))
L = 0.424


def load(path):
    rows = []
    for line in open(path):
        if line.startswith("state") and not line.startswith("states"):
            rows.append([float(x) for x in line.split()[1:]])
    return np.array(rows)


def statics_logs():
    """name -> [n, 15]: pose 7, joints 6, margin, max thrust (logs that carry the two trailers)"""
    out = {}
    for f in sorted(glob.glob(os.path.join(HERE, "ref_dragon_*.txt"))):
        S = load(f)
        if S.ndim == 2 and S.shape[1] in (15, 23):
            out[os.path.basename(f)] = S[:, list(range(13)) + [-2, -1]]
    return out


def quat_R(q):
    x, y, z, w = (q[:, i] for i in range(4))
    s = 2.0 / (x * x + y * y + z * z + w * w)
    R = np.empty((len(q), 3, 3))
    R[:, 0, 0] = 1 - s * (y * y + z * z); R[:, 0, 1] = s * (x * y - w * z); R[:, 0, 2] = s * (x * z + w * y)
    R[:, 1, 0] = s * (x * y + w * z); R[:, 1, 1] = 1 - s * (x * x + z * z); R[:, 1, 2] = s * (y * z - w * x)
    R[:, 2, 0] = s * (x * z - w * y); R[:, 2, 1] = s * (y * z + w * x); R[:, 2, 2] = 1 - s * (x * x + y * y)
    return R


def rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    R = np.zeros((len(a), 3, 3))
    i, j, k = axis, (axis + 1) % 3, (axis + 2) % 3
    R[:, i, i] = 1; R[:, j, j] = c; R[:, j, k] = -s; R[:, k, j] = s; R[:, k, k] = c
    return R


def frames(S, d1):
    """link / inter-joint / gimbal-roll frames of the chain: link L along x, pitch about y, d1 along x, yaw about z"""
    R = quat_R(S[:, 3:7]); o = S[:, 0:3].copy()
    links, inters, rolls = [], [], []
    for i in range(4):
        links.append((o.copy(), R.copy()))
        v = R[:, 2, :]                                   # world z seen from the link
        rolls.append((o + 0.5 * L * R[:, :, 0], R @ rot(0, np.arctan2(-v[:, 1], v[:, 2]))))
        if i < 3:
            o = o + L * R[:, :, 0]; R = R @ rot(1, S[:, 7 + 2 * i]); inters.append((o.copy(), R.copy()))
            o = o + d1 * R[:, :, 0]; R = R @ rot(2, S[:, 8 + 2 * i])
    return links, inters, rolls


def rotor_origins(S, a, d1, h):
    links, _, rolls = frames(S, d1)
    return np.stack([links[i][0] + a * L * links[i][1][:, :, 0] + h * rolls[i][1][:, :, 2] for i in range(4)], 1)


def margin(S, a, d1, h):
    P = rotor_origins(S, a, d1, h)
    xy = P[:, :, :2] - P[:, :, :2].mean(1, keepdims=True)
    C = np.einsum("nki,nkj->nij", xy, xy) / 4
    tr = C[:, 0, 0] + C[:, 1, 1]; det = C[:, 0, 0] * C[:, 1, 1] - C[:, 0, 1] ** 2
    lmin = tr / 2 - np.sqrt(np.maximum(tr * tr / 4 - det, 0))
    return np.sqrt(np.maximum(lmin, 0)) / np.linalg.norm(P[:, 1] - P[:, 0], axis=1)


def geometry_table(train="ref_dragon__dragon_planning_log_vertial_gap_0p5_ceil_1p4.txt"):
    logs = statics_logs()
    St = logs[train]
    r = least_squares(lambda x: margin(St, *x) - St[:, -2], [0.4, 0.02, 0.0], xtol=1e-14, ftol=1e-14)
    print(f"fitted on {train} ({len(St)} states): rotor fraction a = {r.x[0]:.4f}, pitch->yaw offset d1 = {r.x[1]:.4f}, "
          f"roll-frame z offset h = {r.x[2]:.4f}")
    print(f"{'log':72s} {'n':>4s} {'rms':>9s} {'max':>9s}   (frozen 0.5 / 0.0515 / 0.037)")
    worst = 0.0
    for k, S in logs.items():
        e = margin(S, 0.5, 0.0515, 0.037) - S[:, -2]
        worst = max(worst, np.abs(e).max())
        print(f"{k:72s} {len(S):4d} {np.sqrt(np.mean(e ** 2)):9.2e} {np.abs(e).max():9.2e}")
    return r.x, worst


# ---- static thrust: the formulations tried ----------------------------------------------------------------------
def thrust_model(S, variant):
    links, inters, rolls = frames(S, 0.0515)
    P = np.stack([rolls[i][0] + 0.037 * rolls[i][1][:, :, 2] for i in range(4)], 1)
    n = len(S); E3 = np.eye(3)[None]; sig = np.array([1.0, -1.0, 1.0, -1.0])

    def lam(x):
        # x: w2 w3 w4 (link masses, w1 = 1), wj (inter-joint), c1[3] (CoM link 1), c[3] (links 2-4), cr (gimbal
        # module offset up the roll frame), kappa, W
        w = np.array([1.0, x[0], x[1], x[2]]); wj = x[3]; c1 = x[4:7]; c = x[7:10]; cr = x[10]; kap = x[11]; W = x[12]
        pts = []
        for i in range(4):
            o, Rk = links[i]
            pts.append((w[i], o + np.einsum("nij,j->ni", Rk, c1 if i == 0 else c) + cr * rolls[i][1][:, :, 2]))
        for i in range(3):
            pts.append((wj, inters[i][0]))
        M = sum(m for m, _ in pts)
        cog = sum(m * q for m, q in pts) / M
        p = P - cog[:, None, :]
        if variant == "root_frame":                       # positions in the root link's axes, thrust along its z
            p = np.einsum("nji,nkj->nki", links[0][1], p)
        rows = [np.ones((n, 4)), p[:, :, 1], -p[:, :, 0]]
        if variant == "four_rows":                        # + yaw balance sum(sigma_i lambda_i) = 0
            rows.append(np.tile(sig, (n, 1)))
        if variant == "inertia_mixed":                    # rows (I^-1 [y; -x; kappa sigma])_{x,y}: LQI three-axis mode
            I = np.zeros((n, 3, 3))
            for m, q in pts:
                r = q - cog
                I += m * ((r * r).sum(1)[:, None, None] * E3 - r[:, :, None] * r[:, None, :])
            T = np.linalg.solve(I, np.stack([p[:, :, 1], -p[:, :, 0], np.tile(kap * sig, (n, 1))], 1))
            rows = [rows[0], T[:, 0], T[:, 1]]
        if variant == "forces_3d":                        # minimum-norm 3-D module forces, |f_i|
            A = np.zeros((n, 6, 12))
            for i in range(4):
                A[:, 0, 3 * i] = A[:, 1, 3 * i + 1] = A[:, 2, 3 * i + 2] = 1
                px, py, pz = p[:, i, 0], p[:, i, 1], p[:, i, 2]
                A[:, 3, 3 * i + 1] = -pz; A[:, 3, 3 * i + 2] = py
                A[:, 4, 3 * i] = pz; A[:, 4, 3 * i + 2] = -px
                A[:, 5, 3 * i] = -py; A[:, 5, 3 * i + 1] = px
            b = np.zeros((n, 6)); b[:, 2] = W
            f = (A.transpose(0, 2, 1) @ np.linalg.solve(A @ A.transpose(0, 2, 1), b[..., None]))[..., 0]
            return np.linalg.norm(f.reshape(n, 4, 3), axis=2)
        A = np.stack(rows, 1)
        A[:, 1:] /= np.linalg.norm(A[:, 1:], axis=2, keepdims=True)
        b = np.zeros((n, A.shape[1])); b[:, 0] = W
        return (A.transpose(0, 2, 1) @ np.linalg.solve(A @ A.transpose(0, 2, 1), b[..., None]))[..., 0]
    return lam


def thrust_table(train="ref_dragon__dragon_planning_log_vertial_gap_0p5_ceil_1p4.txt", starts=4):
    logs = {k: v for k, v in statics_logs().items() if "0p6_ceil_1p2" not in k}   # (another mass revision)
    St = np.unique(logs[train], axis=0)
    print(f"\nstatic thrust, fitted on {train}; rms [N] of max_i lambda_i against the logged column")
    for variant in ("vertical_min_norm", "four_rows", "inertia_mixed", "forces_3d", "root_frame"):
        lam = thrust_model(St, variant)
        rng = np.random.default_rng(0); best = None
        for _ in range(starts):
            x0 = np.r_[1, 1, 1, 0.1, 0.212, 0, 0, 0.212, 0, 0, 0.0, 0.02, rng.uniform(68, 82)]
            x0[4:11] += rng.normal(0, 0.01, 7)
            try:
                r = least_squares(lambda x: lam(x).max(1) - St[:, -1], x0, xtol=1e-14, ftol=1e-14, max_nfev=300)
            except (np.linalg.LinAlgError, ValueError):
                continue
            if best is None or r.cost < best.cost:
                best = r
        held = []
        for k, S in logs.items():
            if k != train:
                e = thrust_model(S, variant)(best.x).max(1) - S[:, -1]
                held.append(np.sqrt(np.mean(e ** 2)))
        print(f"  {variant:20s} train rms {np.sqrt(np.mean(best.fun ** 2)):.4f}   held-out rms "
              f"{min(held):.3f} .. {max(held):.3f}   weight {best.x[-1]:.2f} N")


if __name__ == "__main__":
    x, worst = geometry_table()
    assert np.allclose(x, [0.5, 0.0515, 0.037], atol=2e-4) and worst < 5e-6
    if "--thrust" in sys.argv:
        thrust_table()
