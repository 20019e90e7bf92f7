"""np_vspec.py -- TEST INFRASTRUCTURE: an independent NumPy/SciPy restatement of V-spec.

Written from DESIGN.md section 3 / SURVEY.md section 7.2 without looking at
vspec_oracle.c's formulation: homogeneous 4x4 transforms, numpy.linalg.pinv for
the static thrust, explicit zonotope supports for the feasible-control distances
and SciPy optimisers for the narrowphase distances.  It is slow and is only used
(a) to cross-check the C oracle in tests/test_oracle_crosscheck.py and (b) when
tests/golden/make_golden.py generates the committed golden vectors.  Parity of
the statics / collision maths with the (absent) third-party reference packages
stays UNPINNED; this file only guards the oracle against its own typos.
"""
import itertools

import numpy as np
from scipy.optimize import minimize, minimize_scalar


def quat_to_T(q, p):
    x, y, z, w = q / np.linalg.norm(q)
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = p
    return T


def rot_T(axis, th):
    c, s = np.cos(th), np.sin(th)
    T = np.eye(4)
    if axis == 0:
        T[1:3, 1:3] = [[c, -s], [s, c]]
    elif axis == 1:
        T[0, 0], T[0, 2], T[2, 0], T[2, 2] = c, s, -s, c
    elif axis == 2:
        T[0:2, 0:2] = [[c, -s], [s, c]]
    return T


def trans_T(t):
    T = np.eye(4)
    T[:3, 3] = t
    return T


def model_update(m, st):
    """One state (px py pz qx qy qz qw joints).  Returns dict of model quantities."""
    N, ns = m.n_rotor, m.n_seg
    F = [quat_to_T(np.array(st[3:7]), np.array(st[0:3]))]
    for k in range(1, ns):
        T = F[-1] @ trans_T(np.array(m.seg_t[k][:]))
        if m.seg_axis[k] >= 0:
            T = T @ rot_T(m.seg_axis[k], st[7 + m.seg_joint[k]] if m.seg_joint[k] >= 0 else 0.0)
        F.append(T)
    masses = np.array(m.seg_mass[:ns])
    coms = np.array([(F[k] @ np.r_[np.array(m.seg_com[k][:]), 1.0])[:3] for k in range(ns)])
    mass = masses.sum()
    cog = (masses[:, None] * coms).sum(0) / mass
    P, U, GR, GP, O, gim = [], [], [], [], [], []
    for i in range(N):
        Fs = F[m.rotor_seg[i]]
        o = (Fs @ np.r_[np.array(m.rotor_pos[i][:]), 1.0])[:3]
        if m.has_gimbal:
            zl = Fs[:3, :3].T @ np.array([0.0, 0.0, 1.0])
            roll = np.arctan2(-zl[1], zl[2])
            pitch = np.arctan2(zl[0], -zl[1] * np.sin(roll) + zl[2] * np.cos(roll))
            Tr = Fs @ trans_T(np.array(m.rotor_pos[i][:])) @ rot_T(0, roll)
            Tp = Tr @ rot_T(1, pitch)
            u = Tp[:3, 2]
            GR.append(Tr)
            GP.append(Tp)
            gim += [roll, pitch]
            thr = (Tr @ np.array([0.0, 0.0, m.rotor_roll_z, 1.0]))[:3]   # thrust-frame origin, up the roll frame's z
        else:
            u = Fs[:3, 2]
            thr = o
        O.append(thr)
        P.append(thr - cog)
        U.append(u)
    P, U = np.array(P), np.array(U)
    sig = np.array(m.rotor_sigma[:N])
    V = np.cross(P, U) + m.m_f_rate * sig[:, None] * U
    rows = [U[:, 2], V[:, 0], V[:, 1], V[:, 2]][: m.wrench_dof]
    Aw = np.array(rows)
    b = np.zeros(m.wrench_dof)
    b[0] = mass * m.gravity
    lam = np.linalg.pinv(Aw) @ b
    fmax = m.thrust_max
    # torque zonotope facets: normals +-(v_i x v_j)
    fc_t = np.inf
    for i, j in itertools.permutations(range(N), 2):
        n = np.cross(V[i], V[j])
        nn = np.linalg.norm(n)
        tot = 0.0
        for k in range(N):
            if k in (i, j):
                continue
            tp = 0.0 if nn < 1e-5 else n @ V[k] / nn
            tot += max(0.0, tp * fmax)
        fc_t = min(fc_t, tot)
    fc_rp = np.inf
    for i in range(N):
        a = V[i, :2]
        na = np.linalg.norm(a)
        tot = 0.0
        for j in range(N):
            if j == i:
                continue
            tp = 0.0 if na < 1e-5 else (a[1] * V[j, 0] - a[0] * V[j, 1]) / na
            tot += max(0.0, tp * fmax)
        fc_rp = min(fc_rp, tot)
    cov = np.cov(P[:, :2].T, bias=True)
    rp_margin = np.sqrt(max(np.linalg.eigvalsh(cov)[0], 0.0)) / np.linalg.norm(P[1] - P[0])   # normaliser pinned by the logs
    slacks = []
    if m.fc_rp_min_thre > 0:
        slacks.append(fc_rp - m.fc_rp_min_thre)
    if m.fc_t_min_thre > 0:
        slacks.append(fc_t - m.fc_t_min_thre)
    slacks += [lam.min() - m.thrust_min, m.thrust_max - lam.max()]
    if m.has_gimbal:
        E = []
        for i in range(N):
            for sgn in (+1, -1):
                E.append(O[i] + GP[i][:3, :3] @ np.array([0.0, sgn * m.edf_offset_y, 0.0]))
        for a_, b_ in itertools.combinations(range(2 * N), 2):
            if a_ // 2 == b_ // 2:
                continue
            dlt = E[a_] - E[b_]
            slacks.append(np.hypot(dlt[0], dlt[1]) - (2 * m.edf_radius + abs(dlt[2]) * np.tan(m.edf_max_tilt)))
    prims = []
    for q in range(m.n_prim):
        if m.prim_frame[q] == 0:
            base = F[m.prim_seg[q]]
        elif m.prim_frame[q] == 1:
            base = GR[m.prim_rotor[q]]
        else:
            base = GP[m.prim_rotor[q]]
        L = np.eye(4)
        L[:3, :3] = np.array(m.prim_rot[q][:]).reshape(3, 3)
        L[:3, 3] = np.array(m.prim_pos[q][:])
        prims.append(base @ L)
    return dict(cog=cog, thrust=lam, fc_t_min=fc_t, fc_rp_min=fc_rp, rp_position_margin=rp_margin,
                margin=min(slacks), gimbal=np.array(gim), prims=prims, cond=np.linalg.cond(Aw))


# ---- narrowphase by optimisation (independent of the closed forms) -------------
def point_box_dist(y, h):
    return np.linalg.norm(np.maximum(np.abs(y) - h, 0.0))


def seg_box_dist_opt(p0, d, h):
    f = lambda t: point_box_dist(p0 + t * d, h) ** 2
    best = min(f(0.0), f(1.0))
    r = minimize_scalar(f, bounds=(0.0, 1.0), method="bounded", options=dict(xatol=1e-14))
    return np.sqrt(min(best, r.fun))


def box_box_dist_opt(cA, RA, a, cB, RB, b):
    """distance between two boxes by minimising |x_A - x_B| over both boxes (convex QP, SLSQP multi-start)."""
    def obj(z):
        xa = cA + RA @ z[:3]
        xb = cB + RB @ z[3:]
        dlt = xa - xb
        return dlt @ dlt

    def grad(z):
        xa = cA + RA @ z[:3]
        xb = cB + RB @ z[3:]
        dlt = xa - xb
        return np.r_[2 * RA.T @ dlt, -2 * RB.T @ dlt]
    bounds = [(-a[k], a[k]) for k in range(3)] + [(-b[k], b[k]) for k in range(3)]
    best = np.inf
    for start in (np.zeros(6), np.r_[RA.T @ (cB - cA), RB.T @ (cA - cB)]):
        z0 = np.clip(start, [lo for lo, _ in bounds], [hi for _, hi in bounds])
        r = minimize(obj, z0, jac=grad, bounds=bounds, method="L-BFGS-B", options=dict(ftol=1e-18, gtol=1e-14, maxiter=2000))
        best = min(best, r.fun)
    return np.sqrt(max(best, 0.0))


def prim_box_dist_opt(ptype, T, dim, box_row):
    """separation (>= 0; 0 when the shapes overlap) of a posed primitive and a world box row
    [cx cy cz qx qy qz qw sx sy sz], through generic optimisers."""
    dim = np.asarray(dim, float)
    oT = quat_to_T(np.array(box_row[3:7]), np.array(box_row[0:3]))
    oh = 0.5 * np.array(box_row[7:10])
    Rel = np.linalg.inv(oT) @ T
    c = Rel[:3, 3]
    if ptype == 0:
        return max(point_box_dist(c, oh) - dim[0], 0.0)
    if ptype == 1:
        ax = Rel[:3, 2]
        return max(seg_box_dist_opt(c - dim[1] * ax, 2 * dim[1] * ax, oh) - dim[0], 0.0)
    return box_box_dist_opt(c, Rel[:3, :3], dim, np.zeros(3), np.eye(3), oh)
