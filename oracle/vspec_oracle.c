/*
 * vspec_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU fp64 restatement ("oracle") of the validity hot path of
 * tongtybj/motion_planning (package sampling_based_method).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file;
 * the shipped library (motion_planning_amd/csrc) never links or calls it.
 *
 * PARITY STATUS
 *   - Edge densification (OMPL validSegmentCount + interpolate) is PINNED by the
 *     reference's own path log path_log/hydrus/hydrus_planning_log.txt
 *     (tests/test_oracle_golden.py reproduces its 235 states and 9 segment
 *     counts).
 *   - The gap worlds are pinned by the literal numbers of
 *     squeeze_navigation/src/planner/sampling_based_method.cpp:76-198.
 *   - Statics / feasible-control / collision arithmetic lives in third-party
 *     packages that are absent from the reference tree (jsk_aerial_robot 1.3.1
 *     aerial_robot_model / hydrus / dragon, OMPL, MoveIt + FCL, KDL) and the
 *     reference holds no test or golden vector for it: PARITY UNPINNED for those
 *     parts.  They restate the published algorithms as written down in
 *     SURVEY.md section 7.2 / Appendix C ("V-spec"), anchored on the reference's
 *     call sites cited at each function.
 *
 * Style: straight-line, readable, no cleverness, no FMA contraction
 * (compile with -O2 -ffp-contract=off).  All paths are relative to the
 * reference root.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mvk.h" /* POD struct definitions only */

#define ORC_PI 3.14159265358979323846

/* ---------------------------------------------------------------------- */
/* small vector / matrix helpers (row-major 3x3)                            */
/* ---------------------------------------------------------------------- */
static void mat_mul(const double A[9], const double B[9], double C[9]) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += A[3 * r + k] * B[3 * k + c];
      C[3 * r + c] = s;
    }
}
static void mat_vec(const double A[9], const double x[3], double y[3]) {
  for (int r = 0; r < 3; ++r) y[r] = A[3 * r] * x[0] + A[3 * r + 1] * x[1] + A[3 * r + 2] * x[2];
}
static void mat_t_vec(const double A[9], const double x[3], double y[3]) {
  for (int c = 0; c < 3; ++c) y[c] = A[c] * x[0] + A[3 + c] * x[1] + A[6 + c] * x[2];
}
static void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double a[3], const double b[3]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static double norm3(const double a[3]) { return sqrt(dot3(a, a)); }

/* tf::Matrix3x3::setRotation(q): the conversion used when the reference turns
 * root_pose.orientation into a transform (multilink_state.cpp:104-105,
 * motion_planning.cpp:607-609).  Divides by |q|^2, so non-unit input is
 * normalised and an all-zero quaternion yields NaN. */
static void quat_to_mat(const double q[4], double R[9]) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double d = x * x + y * y + z * z + w * w;
  double s = 2.0 / d;
  double xs = x * s, ys = y * s, zs = z * s;
  double wx = w * xs, wy = w * ys, wz = w * zs;
  double xx = x * xs, xy = x * ys, xz = x * zs;
  double yy = y * ys, yz = y * zs, zz = z * zs;
  R[0] = 1.0 - (yy + zz); R[1] = xy - wz;         R[2] = xz + wy;
  R[3] = xy + wz;         R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
  R[6] = xz - wy;         R[7] = yz + wx;         R[8] = 1.0 - (xx + yy);
}

static void axis_rot(int axis, double th, double R[9]) {
  double c = cos(th), s = sin(th);
  for (int i = 0; i < 9; ++i) R[i] = 0.0;
  R[0] = R[4] = R[8] = 1.0;
  if (axis == 0) { R[4] = c; R[5] = -s; R[7] = s; R[8] = c; }
  else if (axis == 1) { R[0] = c; R[2] = s; R[6] = -s; R[8] = c; }
  else if (axis == 2) { R[0] = c; R[1] = -s; R[3] = s; R[4] = c; }
}

/* ---------------------------------------------------------------------- */
/* state unpack: motion_planning.cpp:81-129                                 */
/* ---------------------------------------------------------------------- */
static int base_dim(const mvk_params* p) { return p->motion_type == MVK_SE2 ? 3 : 7; }

int orc_space_dim(const mvk_params* p, int J) {
  if (p->planning_mode == MVK_MODE_ONLY_JOINTS) return J;
  if (p->planning_mode == MVK_MODE_ONLY_BASE) return base_dim(p);
  return base_dim(p) + J;
}

/* OMPL state -> (px py pz qx qy qz qw j1..jJ).
 * ONLY_JOINTS: pose stays default.  The reference leaves a default-constructed
 * geometry_msgs::Pose (all-zero quaternion, motion_planning.cpp:78) which tf
 * turns into NaNs; V-spec uses the identity pose instead (documented quirk).
 * ONLY_BASE: joints stay 0 (quirk Q3, motion_planning.cpp:79,87-106).
 * SE2: tf::createQuaternionMsgFromYaw(yaw) = (0,0,sin(yaw/2),cos(yaw/2)). */
void orc_unpack(const mvk_params* p, int J, const double* s, double* o) {
  o[0] = o[1] = o[2] = 0.0; o[3] = o[4] = o[5] = 0.0; o[6] = 1.0;
  for (int j = 0; j < J; ++j) o[7 + j] = 0.0;
  int off = 0;
  if (p->planning_mode != MVK_MODE_ONLY_JOINTS) {
    if (p->motion_type == MVK_SE2) {
      o[0] = s[0]; o[1] = s[1];
      o[5] = sin(s[2] * 0.5); o[6] = cos(s[2] * 0.5);
      off = 3;
    } else {
      for (int k = 0; k < 7; ++k) o[k] = s[k];
      off = 7;
    }
  }
  if (p->planning_mode != MVK_MODE_ONLY_BASE)
    for (int j = 0; j < J; ++j) o[7 + j] = s[off + j];
}

/* ---------------------------------------------------------------------- */
/* primitive-vs-box narrowphase                                             */
/* ---------------------------------------------------------------------- */
static double excess(double x, double h) { /* signed overshoot outside [-h,h] */
  if (x > h) return x - h;
  if (x < -h) return x + h;
  return 0.0;
}

/* distance from a point (box frame) to the solid box [-h,h]^3 */
static double point_box_dist(const double y[3], const double h[3]) {
  double e0 = excess(y[0], h[0]), e1 = excess(y[1], h[1]), e2 = excess(y[2], h[2]);
  return sqrt(e0 * e0 + e1 * e1 + e2 * e2);
}

static double seg_dfun(const double p0[3], const double d[3], const double h[3], double t) {
  double s = 0.0;
  for (int k = 0; k < 3; ++k) s += d[k] * excess(p0[k] + t * d[k], h[k]);
  return 2.0 * s; /* derivative of squared distance along the segment */
}
static int cmp_double(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}
/* Exact distance between the segment p0 + t*d, t in [0,1], and the solid box,
 * both given in the box frame.  f(t) = dist^2 is convex, C1 and piecewise
 * quadratic with breakpoints where a coordinate crosses +-h, so f' is
 * nondecreasing piecewise linear: locate its zero between two consecutive
 * breakpoints and interpolate. */
static double seg_box_dist(const double p0[3], const double d[3], const double h[3]) {
  double T[8];
  int nt = 0;
  T[nt++] = 0.0;
  T[nt++] = 1.0;
  for (int k = 0; k < 3; ++k) {
    if (d[k] != 0.0) {
      double ta = (h[k] - p0[k]) / d[k], tb = (-h[k] - p0[k]) / d[k];
      if (ta > 0.0 && ta < 1.0) T[nt++] = ta;
      if (tb > 0.0 && tb < 1.0) T[nt++] = tb;
    }
  }
  qsort(T, nt, sizeof(double), cmp_double);
  double tstar;
  if (seg_dfun(p0, d, h, 0.0) >= 0.0) tstar = 0.0;
  else if (seg_dfun(p0, d, h, 1.0) <= 0.0) tstar = 1.0;
  else {
    tstar = 1.0;
    for (int i = 1; i < nt; ++i) {
      double fb = seg_dfun(p0, d, h, T[i]);
      if (fb >= 0.0) {
        double fa = seg_dfun(p0, d, h, T[i - 1]);
        tstar = (fb > fa) ? T[i - 1] + (T[i] - T[i - 1]) * (-fa) / (fb - fa) : T[i - 1];
        break;
      }
    }
  }
  double y[3] = {p0[0] + tstar * d[0], p0[1] + tstar * d[1], p0[2] + tstar * d[2]};
  return point_box_dist(y, h);
}

/* Separating-axis test between box A (centre cA, axes = columns of RA, half
 * extents a) and box B.  Returns the largest metric gap over the 15 axes:
 * > 0 separated (a lower bound of the distance), <= 0 overlapping, in which
 * case -gap is the penetration depth (minimum translation distance). */
static double box_box_sat(const double cA[3], const double RA[9], const double a[3],
                          const double cB[3], const double RB[9], const double b[3]) {
  double R[9], AbsR[9], t[3], dw[3] = {cB[0] - cA[0], cB[1] - cA[1], cB[2] - cA[2]};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += RA[3 * k + i] * RB[3 * k + j];
      R[3 * i + j] = s;
      AbsR[3 * i + j] = fabs(s);
    }
  mat_t_vec(RA, dw, t);
  double best = -INFINITY;
  for (int i = 0; i < 3; ++i) { /* A's face normals */
    double g = fabs(t[i]) - (a[i] + b[0] * AbsR[3 * i] + b[1] * AbsR[3 * i + 1] + b[2] * AbsR[3 * i + 2]);
    if (g > best) best = g;
  }
  for (int j = 0; j < 3; ++j) { /* B's face normals */
    double g = fabs(t[0] * R[j] + t[1] * R[3 + j] + t[2] * R[6 + j]) -
               (b[j] + a[0] * AbsR[j] + a[1] * AbsR[3 + j] + a[2] * AbsR[6 + j]);
    if (g > best) best = g;
  }
  for (int i = 0; i < 3; ++i) /* edge x edge axes  A_i x B_j */
    for (int j = 0; j < 3; ++j) {
      int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      double len2 = 1.0 - R[3 * i + j] * R[3 * i + j]; /* |A_i x B_j|^2 */
      if (len2 < 1e-12) continue;                     /* parallel edges: axis degenerate */
      double ra = a[i1] * AbsR[3 * i2 + j] + a[i2] * AbsR[3 * i1 + j];
      double rb = b[j1] * AbsR[3 * i + j2] + b[j2] * AbsR[3 * i + j1];
      double g = (fabs(t[i2] * R[3 * i1 + j] - t[i1] * R[3 * i2 + j]) - (ra + rb)) / sqrt(len2);
      if (g > best) best = g;
    }
  return best;
}

/* min over the 12 edges of box A (given in B's frame: centre c, axes = columns
 * of R, half extents a) of the distance to the solid box B = [-b,b]^3 */
static double box_edges_to_box(const double c[3], const double R[9], const double a[3],
                               const double b[3]) {
  double best = INFINITY;
  for (int ax = 0; ax < 3; ++ax) {
    int u = (ax + 1) % 3, v = (ax + 2) % 3;
    for (int su = -1; su <= 1; su += 2)
      for (int sv = -1; sv <= 1; sv += 2) {
        double p0[3], d[3];
        for (int k = 0; k < 3; ++k) {
          double mid = c[k] + su * a[u] * R[3 * k + u] + sv * a[v] * R[3 * k + v];
          p0[k] = mid - a[ax] * R[3 * k + ax];
          d[k] = 2.0 * a[ax] * R[3 * k + ax];
        }
        double dist = seg_box_dist(p0, d, b);
        if (dist < best) best = dist;
      }
  }
  return best;
}

/* Signed separation of two boxes: exact Euclidean distance when disjoint
 * (closest features of two convex polytopes always include a point on an
 * edge of one of them), -penetration depth when overlapping. */
static double box_box_signed_dist(const double cA[3], const double RA[9], const double a[3],
                                  const double cB[3], const double RB[9], const double b[3]) {
  double sep = box_box_sat(cA, RA, a, cB, RB, b);
  if (!(sep > 0.0)) return sep;
  double R[9], c[3], dw[3];
  /* A in B's frame */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += RB[3 * k + i] * RA[3 * k + j];
      R[3 * i + j] = s;
    }
  for (int k = 0; k < 3; ++k) dw[k] = cA[k] - cB[k];
  mat_t_vec(RB, dw, c);
  double d1 = box_edges_to_box(c, R, a, b);
  /* B in A's frame */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += RA[3 * k + i] * RB[3 * k + j];
      R[3 * i + j] = s;
    }
  for (int k = 0; k < 3; ++k) dw[k] = cB[k] - cA[k];
  mat_t_vec(RA, dw, c);
  double d2 = box_edges_to_box(c, R, b, a);
  return d1 < d2 ? d1 : d2;
}

/* Signed separation between one posed robot primitive and one world box.
 * Role of fcl::distance in collision_avoidance.cpp:148-153 (world pose of the
 * primitive per :123-132); "<= 0" means contact, as :214-218 treats it.
 * SPHERE / CAPSULE saturate at -radius once their core (point / segment)
 * touches the box. */
static double prim_box_signed_dist(int type, const double pc[3], const double pR[9],
                                   const double dim[3], const double oc[3], const double oR[9],
                                   const double oh[3]) {
  if (type == MVK_PRIM_SPHERE) {
    double dw[3] = {pc[0] - oc[0], pc[1] - oc[1], pc[2] - oc[2]}, y[3];
    mat_t_vec(oR, dw, y);
    return point_box_dist(y, oh) - dim[0];
  }
  if (type == MVK_PRIM_CAPSULE) {
    double ax[3] = {pR[2], pR[5], pR[8]}; /* local z in world */
    double aw[3], dirw[3], p0[3], d[3];
    for (int k = 0; k < 3; ++k) {
      aw[k] = pc[k] - dim[1] * ax[k] - oc[k];
      dirw[k] = 2.0 * dim[1] * ax[k];
    }
    mat_t_vec(oR, aw, p0);
    mat_t_vec(oR, dirw, d);
    return seg_box_dist(p0, d, oh) - dim[0];
  }
  return box_box_signed_dist(pc, pR, dim, oc, oR, oh);
}

/* radius of the sphere about the primitive origin that contains it */
static double prim_bound_radius(int type, const double dim[3]) {
  if (type == MVK_PRIM_SPHERE) return dim[0];
  if (type == MVK_PRIM_CAPSULE) return dim[0] + dim[1];
  return sqrt(dim[0] * dim[0] + dim[1] * dim[1] + dim[2] * dim[2]);
}

/* ---------------------------------------------------------------------- */
/* the model update + predicate for one unpacked state                       */
/* ---------------------------------------------------------------------- */
typedef struct orc_out {
  int valid, stable, collision, band;
  double min_dist, margin;
  double fc_t_min, fc_rp_min, rp_position_margin;
  double thrust[MVK_MAX_ROTOR];
  double cog[3];
  double gimbal[2 * MVK_MAX_ROTOR];
  double prim_c[MVK_MAX_PRIM][3];
  double prim_R[MVK_MAX_PRIM][9];
  double pivot_ratio; /* conditioning of the static-thrust normal matrix */
} orc_out;

static int stage_on(int knob, int reference_gate) { return knob < 0 ? reference_gate : knob != 0; }

/* brute != 0 disables the (result-preserving) bounding-sphere pruning */
void orc_eval_state(const mvk_model* m, const mvk_box* boxes, size_t n_boxes, const mvk_params* prm,
                    const double* st, int brute, orc_out* o) {
  const int N = m->n_rotor;
  memset(o, 0, sizeof(*o));
  o->min_dist = INFINITY;
  o->margin = INFINITY;

  /* non-finite input: invalid + band (ABI error convention) */
  int finite = 1;
  for (int k = 0; k < 7 + m->n_joint; ++k)
    if (!isfinite(st[k])) finite = 0;
  double Rw[9];
  quat_to_mat(st + 3, Rw);
  for (int k = 0; k < 9; ++k)
    if (!isfinite(Rw[k])) finite = 0;
  if (!finite) {
    o->band = 1;
    o->margin = -INFINITY;
    o->min_dist = NAN;
    return;
  }

  /* 1. forward kinematics of every segment, in the world frame
   *    (RobotModel::fullForwardKinematics: f_child = f_parent * segment.pose(q),
   *    SURVEY App. C; root = link1, motion_planning.cpp:188) */
  double segR[MVK_MAX_SEG][9], segp[MVK_MAX_SEG][3];
  memcpy(segR[0], Rw, sizeof(Rw));
  segp[0][0] = st[0]; segp[0][1] = st[1]; segp[0][2] = st[2];
  for (int k = 1; k < m->n_seg; ++k) {
    double off[3], J[9];
    mat_vec(segR[k - 1], m->seg_t[k], off);
    for (int c = 0; c < 3; ++c) segp[k][c] = segp[k - 1][c] + off[c];
    if (m->seg_axis[k] >= 0) {
      double th = m->seg_joint[k] >= 0 ? st[7 + m->seg_joint[k]] : 0.0;
      axis_rot(m->seg_axis[k], th, J);
      mat_mul(segR[k - 1], J, segR[k]);
    } else {
      memcpy(segR[k], segR[k - 1], sizeof(Rw));
    }
  }

  /* 2. centre of gravity (KDL::RigidBodyInertia sum -> getCOG, App. C);
   *    world position = root_tf * p(root->cog), multilink_state.cpp:115-117 */
  double mass = 0.0, cog[3] = {0, 0, 0};
  for (int k = 0; k < m->n_seg; ++k) {
    double cw[3];
    mat_vec(segR[k], m->seg_com[k], cw);
    for (int c = 0; c < 3; ++c) cog[c] += m->seg_mass[k] * (segp[k][c] + cw[c]);
    mass += m->seg_mass[k];
  }
  for (int c = 0; c < 3; ++c) { cog[c] /= mass; o->cog[c] = cog[c]; }

  /* 3. rotor origins p_i / normals u_i in the CoG frame.  The reference sets
   *    cog_desire = q_root * R(root->baselink) with baselink == root
   *    (multilink_state.cpp:106-112, motion_planning.cpp:188), so the CoG frame
   *    is world-aligned: p_i = o_i - cog, u_i in world axes.
   *    Gimballed rotors (Dragon::HydrusLikeRobotModel, App. C): roll/pitch are
   *    solved so the thrust axis is the CoG z axis, then FK is re-evaluated. */
  double P[MVK_MAX_ROTOR][3], U[MVK_MAX_ROTOR][3];
  double gimR_roll[MVK_MAX_ROTOR][9], gimR_pitch[MVK_MAX_ROTOR][9], rot_o[MVK_MAX_ROTOR][3], thr_o[MVK_MAX_ROTOR][3];
  for (int i = 0; i < N; ++i) {
    int s = m->rotor_seg[i];
    double off[3];
    mat_vec(segR[s], m->rotor_pos[i], off);
    for (int c = 0; c < 3; ++c) { rot_o[i][c] = thr_o[i][c] = segp[s][c] + off[c]; P[i][c] = rot_o[i][c] - cog[c]; }
    if (m->has_gimbal) {
      double zl[3] = {segR[s][6], segR[s][7], segR[s][8]}; /* CoG z axis seen from the link */
      double roll = atan2(-zl[1], zl[2]);
      double pitch = atan2(zl[0], -zl[1] * sin(roll) + zl[2] * cos(roll));
      double Rx[9], Ry[9];
      axis_rot(0, roll, Rx);
      axis_rot(1, pitch, Ry);
      mat_mul(segR[s], Rx, gimR_roll[i]);
      mat_mul(gimR_roll[i], Ry, gimR_pitch[i]);
      o->gimbal[2 * i] = roll;
      o->gimbal[2 * i + 1] = pitch;
      U[i][0] = gimR_pitch[i][2]; U[i][1] = gimR_pitch[i][5]; U[i][2] = gimR_pitch[i][8];
      /* the thrust frame's origin sits rotor_roll_z up the roll frame's z axis (pinned by the
       * rp_position_margin column of the reference's DRAGON logs, tests/test_pathlog_statics.py) */
      for (int c = 0; c < 3; ++c) {
        thr_o[i][c] = rot_o[i][c] + m->rotor_roll_z * gimR_roll[i][3 * c + 2];
        P[i][c] = thr_o[i][c] - cog[c];
      }
    } else {
      U[i][0] = segR[s][2]; U[i][1] = segR[s][5]; U[i][2] = segR[s][8];
    }
  }

  /* 4. wrench columns v_i = p_i x u_i + kappa*sigma_i*u_i (calcWrenchMatrixOnCoG) */
  double V[MVK_MAX_ROTOR][3];
  for (int i = 0; i < N; ++i) {
    cross3(P[i], U[i], V[i]);
    for (int c = 0; c < 3; ++c) V[i][c] += m->m_f_rate * m->rotor_sigma[i] * U[i][c];
  }

  /* static thrust: lambda = A^T (A A^T)^-1 b on the wrench_dof rows
   * (HydrusRobotModel::calcStaticThrust; getStaticThrust() is consumed at
   * motion_planning.cpp:623-631, static_thrust.cpp:69) */
  const int D = m->wrench_dof;
  double A[4][MVK_MAX_ROTOR], b[4] = {mass * m->gravity, 0, 0, 0};
  for (int i = 0; i < N; ++i) {
    A[0][i] = U[i][2]; A[1][i] = V[i][0]; A[2][i] = V[i][1]; A[3][i] = V[i][2];
  }
  double G[4][4], L[4][4], y[4];
  for (int r = 0; r < D; ++r)
    for (int c = 0; c < D; ++c) {
      double s = 0.0;
      for (int i = 0; i < N; ++i) s += A[r][i] * A[c][i];
      G[r][c] = s;
    }
  double trace = 0.0, minpiv = INFINITY;
  for (int r = 0; r < D; ++r) trace += G[r][r];
  int singular = 0;
  memset(L, 0, sizeof(L));
  for (int r = 0; r < D && !singular; ++r) { /* Cholesky G = L L^T */
    for (int c = 0; c <= r; ++c) {
      double s = G[r][c];
      for (int k = 0; k < c; ++k) s -= L[r][k] * L[c][k];
      if (r == c) {
        if (s < minpiv) minpiv = s;
        if (!(s > 1e-14 * trace)) { singular = 1; break; }
        L[r][r] = sqrt(s);
      } else {
        L[r][c] = s / L[c][c];
      }
    }
  }
  o->pivot_ratio = singular ? 0.0 : minpiv / trace;
  double lam_min = INFINITY, lam_max = -INFINITY;
  if (!singular) {
    double z[4];
    for (int r = 0; r < D; ++r) { /* forward */
      double s = b[r];
      for (int k = 0; k < r; ++k) s -= L[r][k] * z[k];
      z[r] = s / L[r][r];
    }
    for (int r = D - 1; r >= 0; --r) { /* backward */
      double s = z[r];
      for (int k = r + 1; k < D; ++k) s -= L[k][r] * y[k];
      y[r] = s / L[r][r];
    }
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
      for (int r = 0; r < D; ++r) s += A[r][i] * y[r];
      o->thrust[i] = s;
      if (s < lam_min) lam_min = s;
      if (s > lam_max) lam_max = s;
    }
  } else {
    for (int i = 0; i < N; ++i) o->thrust[i] = NAN;
  }

  /* 5. feasible-control torque distances (calcFeasibleControlTDists, App. C):
   *    d_ij = sum_{k != i,j} max(0, ((v_i x v_j)/|v_i x v_j|) . v_k * thrust_max),
   *    triple product := 0 when |v_i x v_j| < 1e-5 */
  double fc_t_min = INFINITY;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      if (i == j) continue;
      double n[3], d = 0.0;
      cross3(V[i], V[j], n);
      double nn = norm3(n);
      for (int k = 0; k < N; ++k) {
        if (k == i || k == j) continue;
        double tp = nn < 1e-5 ? 0.0 : dot3(n, V[k]) / nn;
        double term = tp * m->thrust_max;
        d += term > 0.0 ? term : 0.0;
      }
      if (d < fc_t_min) fc_t_min = d;
    }
  o->fc_t_min = fc_t_min;

  /* roll/pitch version (HydrusRobotModel::calcFeasibleControlRollPitchDists):
   * the same construction with v projected on (x,y); N distances */
  double fc_rp_min = INFINITY;
  for (int i = 0; i < N; ++i) {
    double na = sqrt(V[i][0] * V[i][0] + V[i][1] * V[i][1]), d = 0.0;
    for (int j = 0; j < N; ++j) {
      if (j == i) continue;
      double tp = na < 1e-5 ? 0.0 : (V[i][1] * V[j][0] - V[i][0] * V[j][1]) / na;
      double term = tp * m->thrust_max;
      d += term > 0.0 ? term : 0.0;
    }
    if (d < fc_rp_min) fc_rp_min = d;
  }
  o->fc_rp_min = fc_rp_min;

  /* legacy roll/pitch position margin, logged as minimum_var (motion_planning.cpp:615-620):
   * sqrt(lambda_min(Cov_xy(p_i))) / |p_2 - p_1|, the normaliser being the distance between the first two
   * rotor origins in the CoG frame.  PINNED: with it the DRAGON logs' per-state margins (2,761 states) and the
   * HYDRUS log's minimum_var 0.452857 @ state 200 come out to the printed digits; link_length as the
   * normaliser (this oracle before round 3) gives 0.354 where the logs print 0.5. */
  {
    double ax = 0, ay = 0, sxx = 0, sxy = 0, syy = 0;
    for (int i = 0; i < N; ++i) { ax += P[i][0]; ay += P[i][1]; }
    ax /= N; ay /= N;
    for (int i = 0; i < N; ++i) {
      double dx = P[i][0] - ax, dy = P[i][1] - ay;
      sxx += dx * dx; sxy += dx * dy; syy += dy * dy;
    }
    sxx /= N; sxy /= N; syy /= N;
    double hd = 0.5 * (sxx - syy);
    double lmin = 0.5 * (sxx + syy) - sqrt(hd * hd + sxy * sxy);
    double d01[3] = {P[1][0] - P[0][0], P[1][1] - P[0][1], P[1][2] - P[0][2]};
    o->rp_position_margin = sqrt(lmin > 0.0 ? lmin : 0.0) / norm3(d01);
  }

  /* 6. stabilityCheck() (RobotModel / HydrusRobotModel / Dragon overlap, App. C) */
  double margin = INFINITY;
  {
    double s;
    /* a threshold <= 0 switches that test off (the way the Hydrus ctor passes
     * fc_f_min_thre = 0; DRAGON's contra-rotating fans make fc_t identically 0) */
    if (m->fc_rp_min_thre > 0.0) { s = fc_rp_min - m->fc_rp_min_thre; if (s < margin) margin = s; }
    if (m->fc_t_min_thre > 0.0) { s = fc_t_min - m->fc_t_min_thre; if (s < margin) margin = s; }
    if (singular) margin = -INFINITY;
    else {
      s = lam_min - m->thrust_min; if (s < margin) margin = s;
      s = m->thrust_max - lam_max; if (s < margin) margin = s;
    }
    if (m->has_gimbal) { /* EDF overlap: projected distance vs 2 r + |dz| tan(max tilt) */
      double E[2 * MVK_MAX_ROTOR][3];
      for (int i = 0; i < N; ++i)
        for (int sgn = 0; sgn < 2; ++sgn) {
          double loc[3] = {0.0, sgn ? -m->edf_offset_y : m->edf_offset_y, 0.0}, w[3];
          mat_vec(gimR_pitch[i], loc, w);
          for (int c = 0; c < 3; ++c) E[2 * i + sgn][c] = thr_o[i][c] + w[c] - cog[c];
        }
      for (int a = 0; a < 2 * N; ++a)
        for (int c2 = a + 1; c2 < 2 * N; ++c2) {
          if (a / 2 == c2 / 2) continue; /* the two fans of one module */
          double dx = E[a][0] - E[c2][0], dy = E[a][1] - E[c2][1], dz = E[a][2] - E[c2][2];
          double proj = sqrt(dx * dx + dy * dy);
          double thre = m->edf_radius + fabs(dz) * tan(m->edf_max_tilt) + m->edf_radius;
          s = proj - thre; if (s < margin) margin = s;
        }
    }
  }
  o->margin = margin;
  o->stable = margin >= 0.0;

  /* 7. robot collision primitives vs world boxes (PlanningScene::checkCollision
   *    with every self pair allowed: config/hydrus/quad/moveit.srdf:5-32) */
  for (int q = 0; q < m->n_prim; ++q) {
    double base_R[9], base_p[3];
    int s = m->prim_seg[q];
    if (m->prim_frame[q] == MVK_FRAME_SEG) {
      memcpy(base_R, segR[s], sizeof(base_R));
      memcpy(base_p, segp[s], sizeof(base_p));
    } else {
      int i = m->prim_rotor[q];
      memcpy(base_R, m->prim_frame[q] == MVK_FRAME_GIMBAL_ROLL ? gimR_roll[i] : gimR_pitch[i],
             sizeof(base_R));
      memcpy(base_p, rot_o[i], sizeof(base_p));
    }
    double off[3];
    mat_vec(base_R, m->prim_pos[q], off);
    for (int c = 0; c < 3; ++c) o->prim_c[q][c] = base_p[c] + off[c];
    mat_mul(base_R, m->prim_rot[q], o->prim_R[q]);
  }

  int do_stab = stage_on(prm->check_stability, prm->planning_mode != MVK_MODE_ONLY_BASE ||
                                                   prm->motion_type == MVK_SE3);
  int do_coll = stage_on(prm->check_collision, prm->planning_mode != MVK_MODE_ONLY_JOINTS);

  double min_dist = INFINITY;
  if (do_coll) {
    for (size_t ob = 0; ob < n_boxes; ++ob) {
      double oR[9], oh[3] = {0.5 * boxes[ob].size[0], 0.5 * boxes[ob].size[1], 0.5 * boxes[ob].size[2]};
      quat_to_mat(boxes[ob].quat, oR);
      for (int q = 0; q < m->n_prim; ++q) {
        if (!brute) { /* pruning that cannot change the minimum */
          double dw[3] = {o->prim_c[q][0] - boxes[ob].center[0], o->prim_c[q][1] - boxes[ob].center[1],
                          o->prim_c[q][2] - boxes[ob].center[2]}, yb[3];
          mat_t_vec(oR, dw, yb);
          double lb = point_box_dist(yb, oh) - prim_bound_radius(m->prim_type[q], m->prim_dim[q]);
          if (lb > 0.0 && lb > min_dist) continue; /* separated and farther than the current minimum */
        }
        double d = prim_box_signed_dist(m->prim_type[q], o->prim_c[q], o->prim_R[q], m->prim_dim[q],
                                        boxes[ob].center, oR, oh);
        if (d < min_dist) min_dist = d;
      }
    }
  }
  o->min_dist = min_dist;
  o->collision = do_coll && min_dist <= 0.0;

  /* 8. verdict (motion_planning.cpp:138-163) + boundary band */
  o->valid = (!do_stab || o->stable) && !o->collision;
  o->band = (do_stab && fabs(margin) <= prm->eps_margin) || (do_coll && fabs(min_dist) <= prm->eps_dist);
}

/* ---------------------------------------------------------------------- */
/* batch driver (the "CPU checker" timed as cpu_baseline)                    */
/* ---------------------------------------------------------------------- */
typedef struct orc_job {
  const mvk_model* m; const mvk_box* boxes; size_t n_boxes; const mvk_params* prm;
  const double* soa; size_t n, ld; int layout, brute;
  uint8_t* flags; double* min_dist; double* margin; uint8_t* band;
  double* fc_t_min; double* fc_rp_min; double* thrust; double* rp_margin; double* cog; double* gimbal;
  uint8_t* stable; uint8_t* collision; double* pivot_ratio;
  size_t begin, end;
} orc_job;

static void* orc_worker(void* arg) {
  orc_job* j = (orc_job*)arg;
  const int J = j->m->n_joint, N = j->m->n_rotor;
  int dim = j->layout == MVK_LAYOUT_POSE7 ? 7 + J : orc_space_dim(j->prm, J);
  double s[MVK_MAX_STATE_DIM], st[MVK_MAX_STATE_DIM];
  orc_out o;
  for (size_t i = j->begin; i < j->end; ++i) {
    for (int r = 0; r < dim; ++r) s[r] = j->soa[(size_t)r * j->ld + i];
    if (j->layout == MVK_LAYOUT_POSE7) {
      /* same gating as the unpack: mode 0 ignores the pose, mode 1 the joints */
      memcpy(st, s, sizeof(double) * (7 + J));
      if (j->prm->planning_mode == MVK_MODE_ONLY_JOINTS) {
        st[0] = st[1] = st[2] = st[3] = st[4] = st[5] = 0.0; st[6] = 1.0;
      }
      if (j->prm->planning_mode == MVK_MODE_ONLY_BASE)
        for (int q = 0; q < J; ++q) st[7 + q] = 0.0;
    } else {
      orc_unpack(j->prm, J, s, st);
    }
    orc_eval_state(j->m, j->boxes, j->n_boxes, j->prm, st, j->brute, &o);
    if (j->flags) j->flags[i] = (uint8_t)o.valid;
    if (j->min_dist) j->min_dist[i] = o.min_dist;
    if (j->margin) j->margin[i] = o.margin;
    if (j->band) j->band[i] = (uint8_t)o.band;
    if (j->stable) j->stable[i] = (uint8_t)o.stable;
    if (j->collision) j->collision[i] = (uint8_t)o.collision;
    if (j->fc_t_min) j->fc_t_min[i] = o.fc_t_min;
    if (j->fc_rp_min) j->fc_rp_min[i] = o.fc_rp_min;
    if (j->rp_margin) j->rp_margin[i] = o.rp_position_margin;
    if (j->pivot_ratio) j->pivot_ratio[i] = o.pivot_ratio;
    if (j->thrust) for (int r = 0; r < N; ++r) j->thrust[(size_t)r * j->ld + i] = o.thrust[r];
    if (j->cog) for (int r = 0; r < 3; ++r) j->cog[(size_t)r * j->ld + i] = o.cog[r];
    if (j->gimbal) for (int r = 0; r < 2 * N; ++r) j->gimbal[(size_t)r * j->ld + i] = o.gimbal[r];
  }
  return NULL;
}

typedef struct orc_batch_out {
  uint8_t* flags; double* min_dist; double* margin; uint8_t* band;
  uint8_t* stable; uint8_t* collision;
  double* fc_t_min; double* fc_rp_min; double* thrust; double* rp_margin; double* cog; double* gimbal;
  double* pivot_ratio;
} orc_batch_out;

int orc_eval_batch(const mvk_model* m, const mvk_box* boxes, size_t n_boxes, const mvk_params* prm,
                   const double* soa, size_t n, size_t ld, int layout, int brute, int n_threads,
                   const orc_batch_out* out) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  pthread_t th[256];
  orc_job jobs[256];
  size_t chunk = (n + n_threads - 1) / n_threads;
  int used = 0;
  for (int t = 0; t < n_threads; ++t) {
    orc_job* j = &jobs[t];
    memset(j, 0, sizeof(*j));
    j->m = m; j->boxes = boxes; j->n_boxes = n_boxes; j->prm = prm;
    j->soa = soa; j->n = n; j->ld = ld; j->layout = layout; j->brute = brute;
    j->flags = out->flags; j->min_dist = out->min_dist; j->margin = out->margin; j->band = out->band;
    j->stable = out->stable; j->collision = out->collision;
    j->fc_t_min = out->fc_t_min; j->fc_rp_min = out->fc_rp_min; j->thrust = out->thrust;
    j->rp_margin = out->rp_margin; j->cog = out->cog; j->gimbal = out->gimbal;
    j->pivot_ratio = out->pivot_ratio;
    j->begin = (size_t)t * chunk;
    j->end = j->begin + chunk < n ? j->begin + chunk : n;
    if (j->begin >= j->end) break;
    used++;
  }
  if (used == 1) { orc_worker(&jobs[0]); return 0; }
  for (int t = 0; t < used; ++t) pthread_create(&th[t], NULL, orc_worker, &jobs[t]);
  for (int t = 0; t < used; ++t) pthread_join(th[t], NULL);
  return 0;
}

/* posed primitives of one state (for geometry tests) */
void orc_prim_poses(const mvk_model* m, const mvk_params* prm, const double* pose7j, double* centers,
                    double* rots) {
  orc_out o;
  mvk_params p = *prm;
  p.check_collision = 0;
  orc_eval_state(m, NULL, 0, &p, pose7j, 1, &o);
  for (int q = 0; q < m->n_prim; ++q) {
    memcpy(centers + 3 * q, o.prim_c[q], 3 * sizeof(double));
    memcpy(rots + 9 * q, o.prim_R[q], 9 * sizeof(double));
  }
}

double orc_prim_box_signed_dist(int type, const double* pc, const double* pR, const double* dim,
                                const mvk_box* box) {
  double oR[9], oh[3] = {0.5 * box->size[0], 0.5 * box->size[1], 0.5 * box->size[2]};
  quat_to_mat(box->quat, oR);
  return prim_box_signed_dist(type, pc, pR, dim, box->center, oR, oh);
}

/* ---------------------------------------------------------------------- */
/* OMPL state-space arithmetic (third-party, absent; published algorithm of   */
/* ompl::base::{RealVector,SO2,SO3,SE2,SE3,Compound}StateSpace as summarised  */
/* in SURVEY.md section 3.3 / App. C; configured by motion_planning.cpp:192-258) */
/* ---------------------------------------------------------------------- */
static double so2_distance(double a, double b) {
  double d = fabs(a - b);
  return d > ORC_PI ? 2.0 * ORC_PI - d : d;
}
static double so2_interpolate(double from, double to, double t) {
  double diff = to - from;
  if (fabs(diff) <= ORC_PI) return from + diff * t;
  if (diff > 0.0) diff = 2.0 * ORC_PI - diff;
  else diff = -2.0 * ORC_PI - diff;
  double v = from - diff * t;
  if (v > ORC_PI) v -= 2.0 * ORC_PI;
  else if (v < -ORC_PI) v += 2.0 * ORC_PI;
  return v;
}
static double so3_arc(const double* a, const double* b) {
  double dq = fabs(a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]);
  if (dq > 1.0 - 1e-9) return 0.0;
  return acos(dq);
}
static void so3_interpolate(const double* from, const double* to, double t, double* out) {
  double theta = so3_arc(from, to);
  if (theta > 2.220446049250313e-16) {
    double d = 1.0 / sin(theta);
    double s0 = sin((1.0 - t) * theta);
    double s1 = sin(t * theta);
    double dq = from[0] * to[0] + from[1] * to[1] + from[2] * to[2] + from[3] * to[3];
    if (dq < 0) s1 = -s1; /* long-angle case */
    for (int k = 0; k < 4; ++k) out[k] = (from[k] * s0 + to[k] * s1) * d;
  } else {
    for (int k = 0; k < 4; ++k) out[k] = from[k];
  }
}
static double rv_distance(const double* a, const double* b, int n) {
  double s = 0.0;
  for (int k = 0; k < n; ++k) s += (a[k] - b[k]) * (a[k] - b[k]);
  return sqrt(s);
}
static double rv_extent(const double* lo, const double* hi, int n) {
  double s = 0.0;
  for (int k = 0; k < n; ++k) s += (hi[k] - lo[k]) * (hi[k] - lo[k]);
  return sqrt(s);
}

typedef struct leaf_info { double dist, extent; } leaf_info;

/* distances / extents of the leaf spaces: [0] translation, [1] rotation, [2] joints */
static void leaves(const mvk_params* p, int J, const double* a, const double* b, leaf_info L[3],
                   int present[3]) {
  int off = 0;
  present[0] = present[1] = present[2] = 0;
  if (p->planning_mode != MVK_MODE_ONLY_JOINTS) {
    present[0] = present[1] = 1;
    if (p->motion_type == MVK_SE2) {
      L[0].dist = rv_distance(a, b, 2);      L[0].extent = rv_extent(p->pos_low, p->pos_high, 2);
      L[1].dist = so2_distance(a[2], b[2]);  L[1].extent = ORC_PI;
      off = 3;
    } else {
      L[0].dist = rv_distance(a, b, 3);      L[0].extent = rv_extent(p->pos_low, p->pos_high, 3);
      L[1].dist = so3_arc(a + 3, b + 3);     L[1].extent = 0.5 * ORC_PI;
      off = 7;
    }
  }
  if (p->planning_mode != MVK_MODE_ONLY_BASE) {
    present[2] = 1;
    L[2].dist = rv_distance(a + off, b + off, J);
    L[2].extent = rv_extent(p->joint_low, p->joint_high, J);
  }
}

static double rot_weight(const mvk_params* p) { return p->motion_type == MVK_SE2 ? 0.5 : 1.0; }

double orc_space_distance(const mvk_params* p, int J, const double* a, const double* b) {
  leaf_info L[3]; int pr[3];
  leaves(p, J, a, b, L, pr);
  double base = pr[0] ? L[0].dist + rot_weight(p) * L[1].dist : 0.0;
  if (p->planning_mode == MVK_MODE_ONLY_JOINTS) return L[2].dist;
  if (p->planning_mode == MVK_MODE_ONLY_BASE) return base;
  return 1.0 * base + p->joint_subspace_weight * L[2].dist;
}

double orc_space_max_extent(const mvk_params* p, int J) {
  double z[MVK_MAX_STATE_DIM] = {0};
  z[6] = 1.0;
  leaf_info L[3]; int pr[3];
  leaves(p, J, z, z, L, pr);
  double base = pr[0] ? L[0].extent + rot_weight(p) * L[1].extent : 0.0;
  if (p->planning_mode == MVK_MODE_ONLY_JOINTS) return L[2].extent;
  if (p->planning_mode == MVK_MODE_ONLY_BASE) return base;
  return 1.0 * base + p->joint_subspace_weight * L[2].extent;
}

/* StateSpace::validSegmentCount = factor * ceil(distance / (maxExtent * fraction))
 * per leaf; a compound takes the max over its components.  In mode 2 the
 * reference installs neither resolution nor factor (quirk Q4), so OMPL's
 * defaults 0.01 / 1 apply when mode2_ompl_defaults is set.  In mode 1 the
 * reference sets the factor on the SE2/SE3 compound (motion_planning.cpp:244),
 * which never reads it, while the resolution (:248) reaches the leaves: pinned
 * by path_log/hydrus/planning_log.txt (tests/test_pathlog_pin_more.py). */
int orc_space_valid_segment_count(const mvk_params* p, int J, const double* a, const double* b) {
  double frac = p->validity_resolution;
  int factor = p->valid_segment_count_factor;
  if (p->planning_mode == MVK_MODE_JOINTS_AND_BASE && p->mode2_ompl_defaults) { frac = 0.01; factor = 1; }
  if (p->planning_mode == MVK_MODE_ONLY_BASE) factor = 1;
  leaf_info L[3]; int pr[3];
  leaves(p, J, a, b, L, pr);
  int best = 0;
  for (int k = 0; k < 3; ++k) {
    if (!pr[k]) continue;
    int c = factor * (int)ceil(L[k].dist / (L[k].extent * frac));
    if (c > best) best = c;
  }
  return best;
}

void orc_space_interpolate(const mvk_params* p, int J, const double* a, const double* b, double t,
                           double* out) {
  int off = 0;
  if (p->planning_mode != MVK_MODE_ONLY_JOINTS) {
    if (p->motion_type == MVK_SE2) {
      out[0] = a[0] + (b[0] - a[0]) * t;
      out[1] = a[1] + (b[1] - a[1]) * t;
      out[2] = so2_interpolate(a[2], b[2], t);
      off = 3;
    } else {
      for (int k = 0; k < 3; ++k) out[k] = a[k] + (b[k] - a[k]) * t;
      so3_interpolate(a + 3, b + 3, t, out + 3);
      off = 7;
    }
  }
  if (p->planning_mode != MVK_MODE_ONLY_BASE)
    for (int k = 0; k < J; ++k) out[off + k] = a[off + k] + (b[off + k] - a[off + k]) * t;
}

/* DiscreteMotionValidator::checkMotion (bound at motion_planning.cpp:238,249):
 * s1 assumed valid; states j/nd, j = 1..nd (j = nd is s2).  Returns ok and the
 * smallest invalid j (nd+1 when none).  nd <= 0 is treated as 1. */
int orc_check_motion(const mvk_model* m, const mvk_box* boxes, size_t n_boxes, const mvk_params* prm,
                     const double* s1, const double* s2, int nd, int* first_invalid, int* band_any) {
  const int J = m->n_joint;
  double s[MVK_MAX_STATE_DIM], st[MVK_MAX_STATE_DIM];
  orc_out o;
  if (nd < 1) nd = 1;
  int first = nd + 1, band = 0;
  for (int j = 1; j <= nd; ++j) {
    if (j == nd) memcpy(s, s2, sizeof(double) * orc_space_dim(prm, J));
    else orc_space_interpolate(prm, J, s1, s2, (double)j / (double)nd, s);
    orc_unpack(prm, J, s, st);
    orc_eval_state(m, boxes, n_boxes, prm, st, 0, &o);
    if (o.band) band = 1;
    if (!o.valid && j < first) first = j;
  }
  if (first_invalid) *first_invalid = first;
  if (band_any) *band_any = band;
  return first == nd + 1;
}

int orc_check_motion_batch(const mvk_model* m, const mvk_box* boxes, size_t n_boxes,
                           const mvk_params* prm, const double* s1, const double* s2, size_t n_edges,
                           size_t ld, const int32_t* nd, int nd_uniform, uint8_t* ok,
                           int32_t* first_invalid, uint8_t* band_any) {
  int dim = orc_space_dim(prm, m->n_joint);
  double a[MVK_MAX_STATE_DIM], b[MVK_MAX_STATE_DIM];
  for (size_t e = 0; e < n_edges; ++e) {
    for (int r = 0; r < dim; ++r) { a[r] = s1[(size_t)r * ld + e]; b[r] = s2[(size_t)r * ld + e]; }
    int fi, bd;
    int r = orc_check_motion(m, boxes, n_boxes, prm, a, b, nd ? nd[e] : nd_uniform, &fi, &bd);
    if (ok) ok[e] = (uint8_t)r;
    if (first_invalid) first_invalid[e] = fi;
    if (band_any) band_any[e] = (uint8_t)bd;
  }
  return 0;
}

/* ---------------------------------------------------------------------- */
/* gap worlds: SamplingBasedMethod::setCollisionWorld                        */
/* (squeeze_navigation/src/planner/sampling_based_method.cpp:58-202)          */
/* ---------------------------------------------------------------------- */
static void put_box(mvk_box* b, double x, double y, double z, double sx, double sy, double sz) {
  b->center[0] = x; b->center[1] = y; b->center[2] = z;
  b->quat[0] = b->quat[1] = b->quat[2] = 0.0; b->quat[3] = 1.0; /* wall_pose.orientation.w = 1 (:67) */
  b->size[0] = sx; b->size[1] = sy; b->size[2] = sz;
}

/* HORIZONTAL_GAP (:76-113).  Quirk Q2: the second wall keeps the first wall's
 * x dimension (gap_left_width) although it is positioned with gap_right_width. */
int orc_scene_horizontal_gap(double gap_left_x, double gap_y_offset, double gap_left_width,
                             double gap_right_width, mvk_box* out) {
  put_box(&out[0], gap_left_x + gap_left_width / 2, (2.5 + gap_y_offset) / 2, 0.0, gap_left_width, 2.5, 10);
  put_box(&out[1], gap_left_x + gap_right_width / 2, -(2.5 + gap_y_offset) / 2, 0.0, gap_left_width, 2.5, 10);
  put_box(&out[2], 1.0, 2.5, 0.0, 8, 0.6, 10);
  put_box(&out[3], 1.0, -2.5, 0.0, 8, 0.6, 10);
  return 4;
}

/* VERTICAL_GAP (:114-198); the six planning-only walls when planning_walls != 0 */
int orc_scene_vertical_gap(double gx, double gy, double gh, int planning_walls, double ceiling_offset,
                           double side_wall_width, mvk_box* out) {
  const double wl = 10;
  int n = 0;
  put_box(&out[n++], gx / 2 + wl / 2, 0, gh, wl, wl, 0.05);
  put_box(&out[n++], 0, gy / 2 + wl / 2, gh, wl, wl, 0.05);
  put_box(&out[n++], -gx / 2 - wl / 2, 0, gh, wl, wl, 0.05);
  put_box(&out[n++], 0, -gy / 2 - wl / 2, gh, wl, wl, 0.05);
  if (planning_walls) {
    put_box(&out[n++], 0, 0, ceiling_offset + gh, wl, wl, 0.05); /* ceiling */
    put_box(&out[n++], 0, 0, 0, wl, wl, 0.05);                   /* ground */
    put_box(&out[n++], side_wall_width / 2, 0, gh / 2, 0.05, side_wall_width, gh);
    put_box(&out[n++], -side_wall_width / 2, 0, gh / 2, 0.05, side_wall_width, gh);
    put_box(&out[n++], 0, side_wall_width / 2, gh / 2, side_wall_width, 0.05, gh);
    put_box(&out[n++], 0, -side_wall_width / 2, gh / 2, side_wall_width, 0.05, gh);
  }
  return n;
}

/* ---------------------------------------------------------------------- */
/* collision-avoidance constraint rows                     */
/* Restates differential_kinematics CollisionAvoidance::getConstraint           */
/* (src/constraint/collision_avoidance.cpp:102-277): per robot collision        */
/* primitive the minimum distance to the obstacle set with nearest points       */
/* (:147-153), the unit normal from the environment to the robot (:255), the    */
/* row n^T J_lin (:259-260) and the damped lower bound (:262-265,                 */
/* constraint/base_plugin.h:108-113).  fcl::distance and                         */
/* RobotModel::getJacobian are third-party and absent: the closed forms below    */
/* and the column convention [root v (world), root w (world), planned joints]    */
/* are this build's V-spec -- parity unpinned.                                   */
/* ---------------------------------------------------------------------- */
/* segment p0 + t d vs box: parameter, point on the segment, point on the box */
static double seg_box_closest(const double p0[3], const double d[3], const double h[3], double y[3],
                              double z[3]) {
  double T[8];
  int nt = 0;
  T[nt++] = 0.0;
  T[nt++] = 1.0;
  for (int k = 0; k < 3; ++k)
    if (d[k] != 0.0) {
      double ta = (h[k] - p0[k]) / d[k], tb = (-h[k] - p0[k]) / d[k];
      if (ta > 0.0 && ta < 1.0) T[nt++] = ta;
      if (tb > 0.0 && tb < 1.0) T[nt++] = tb;
    }
  qsort(T, nt, sizeof(double), cmp_double);
  double tstar;
  if (seg_dfun(p0, d, h, 0.0) >= 0.0) tstar = 0.0;
  else if (seg_dfun(p0, d, h, 1.0) <= 0.0) tstar = 1.0;
  else {
    tstar = 1.0;
    for (int i = 1; i < nt; ++i) {
      double fb = seg_dfun(p0, d, h, T[i]);
      if (fb >= 0.0) {
        double fa = seg_dfun(p0, d, h, T[i - 1]);
        tstar = (fb > fa) ? T[i - 1] + (T[i] - T[i - 1]) * (-fa) / (fb - fa) : T[i - 1];
        break;
      }
    }
  }
  for (int k = 0; k < 3; ++k) {
    y[k] = p0[k] + tstar * d[k];
    z[k] = y[k] > h[k] ? h[k] : (y[k] < -h[k] ? -h[k] : y[k]);
  }
  return point_box_dist(y, h);
}

/* closest pair between the 12 edges of box A (in B's frame) and solid B */
static double box_edges_to_box_cp(const double c[3], const double R[9], const double a[3], const double b[3],
                                  double pa[3], double pb[3]) {
  double best = INFINITY;
  for (int ax = 0; ax < 3; ++ax) {
    int u = (ax + 1) % 3, v = (ax + 2) % 3;
    for (int su = -1; su <= 1; su += 2)
      for (int sv = -1; sv <= 1; sv += 2) {
        double p0[3], d[3], y[3], z[3];
        for (int k = 0; k < 3; ++k) {
          double mid = c[k] + su * a[u] * R[3 * k + u] + sv * a[v] * R[3 * k + v];
          p0[k] = mid - a[ax] * R[3 * k + ax];
          d[k] = 2.0 * a[ax] * R[3 * k + ax];
        }
        double dist = seg_box_closest(p0, d, b, y, z);
        if (dist < best) {
          best = dist;
          memcpy(pa, y, sizeof(y));
          memcpy(pb, z, sizeof(z));
        }
      }
  }
  return best;
}

/* signed distance + nearest points in world coordinates (NaN when in contact) */
static double prim_box_closest(int type, const double pc[3], const double pR[9], const double dim[3],
                               const double oc[3], const double oR[9], const double oh[3], double p_robot[3],
                               double p_env[3]) {
  double dist;
  p_robot[0] = p_robot[1] = p_robot[2] = p_env[0] = p_env[1] = p_env[2] = NAN;
  if (type == MVK_PRIM_SPHERE || type == MVK_PRIM_CAPSULE) {
    const double hl = type == MVK_PRIM_CAPSULE ? dim[1] : 0.0;
    double ax[3] = {pR[2], pR[5], pR[8]}, aw[3], dirw[3], p0[3], d[3], y[3], z[3];
    for (int k = 0; k < 3; ++k) {
      aw[k] = pc[k] - hl * ax[k] - oc[k];
      dirw[k] = 2.0 * hl * ax[k];
    }
    mat_t_vec(oR, aw, p0);
    mat_t_vec(oR, dirw, d);
    double core = seg_box_closest(p0, d, oh, y, z);
    dist = core - dim[0];
    if (dist > 0.0) { /* robot point: from the core towards the box by the radius */
      double yw[3], zw[3];
      mat_vec(oR, y, yw);
      mat_vec(oR, z, zw);
      for (int k = 0; k < 3; ++k) {
        p_env[k] = oc[k] + zw[k];
        p_robot[k] = oc[k] + yw[k] + dim[0] * (zw[k] - yw[k]) / core;
      }
    }
    return dist;
  }
  dist = box_box_sat(pc, pR, dim, oc, oR, oh);
  if (!(dist > 0.0)) return dist;
  double R[9], c[3], dw[3], pa1[3], pb1[3], pa2[3], pb2[3];
  for (int i = 0; i < 3; ++i) /* primitive in the obstacle frame */
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += oR[3 * k + i] * pR[3 * k + j];
      R[3 * i + j] = s;
    }
  for (int k = 0; k < 3; ++k) dw[k] = pc[k] - oc[k];
  mat_t_vec(oR, dw, c);
  double d1 = box_edges_to_box_cp(c, R, dim, oh, pa1, pb1); /* pa1 on the robot, pb1 on the box (box frame) */
  for (int i = 0; i < 3; ++i) /* obstacle in the primitive frame */
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += pR[3 * k + i] * oR[3 * k + j];
      R[3 * i + j] = s;
    }
  for (int k = 0; k < 3; ++k) dw[k] = oc[k] - pc[k];
  mat_t_vec(pR, dw, c);
  double d2 = box_edges_to_box_cp(c, R, oh, dim, pa2, pb2); /* pa2 on the box, pb2 on the robot (primitive frame) */
  double w[3];
  if (d1 <= d2) {
    mat_vec(oR, pa1, w); for (int k = 0; k < 3; ++k) p_robot[k] = oc[k] + w[k];
    mat_vec(oR, pb1, w); for (int k = 0; k < 3; ++k) p_env[k] = oc[k] + w[k];
    return d1;
  }
  mat_vec(pR, pa2, w); for (int k = 0; k < 3; ++k) p_env[k] = pc[k] + w[k];
  mat_vec(pR, pb2, w); for (int k = 0; k < 3; ++k) p_robot[k] = pc[k] + w[k];
  return d2;
}

/* One state (unpacked px..qw j1..jJ).  Arrays are per primitive q: dist[q], box[q], p_env[3q..],
 * p_robot[3q..], row[(6+J)q..], lb[q].  Returns getConstraint's verdict: 0 when a distance <= 0. */
int orc_avoid_state(const mvk_model* m, const mvk_box* boxes, size_t n_boxes, const mvk_params* prm,
                    const double* st, double damping_gain, double constraint_range, double forbidden_range,
                    int full_body, double* dist, int32_t* box, double* p_env, double* p_robot, double* row,
                    double* lb) {
  const int J = m->n_joint, W = 6 + J;
  orc_out o;
  mvk_params p = *prm;
  p.check_collision = 0;
  orc_eval_state(m, NULL, 0, &p, st, 1, &o); /* posed primitives (incl. gimbal frames) */
  /* joint origins and axes in the world frame */
  double R[9], pos[3] = {st[0], st[1], st[2]}, jo[MVK_MAX_SEG][3], ja[MVK_MAX_SEG][3];
  quat_to_mat(st + 3, R);
  for (int k = 1; k < m->n_seg; ++k) {
    double off[3], Jm[9], Rn[9];
    mat_vec(R, m->seg_t[k], off);
    for (int c = 0; c < 3; ++c) pos[c] += off[c];
    for (int c = 0; c < 3; ++c) { jo[k][c] = pos[c]; ja[k][c] = m->seg_axis[k] >= 0 ? R[3 * c + m->seg_axis[k]] : 0.0; }
    if (m->seg_axis[k] >= 0) {
      axis_rot(m->seg_axis[k], m->seg_joint[k] >= 0 ? st[7 + m->seg_joint[k]] : 0.0, Jm);
      mat_mul(R, Jm, Rn);
      memcpy(R, Rn, sizeof(R));
    }
  }
  int ok = 1;
  for (int q = 0; q < m->n_prim; ++q) {
    double best = INFINITY, be[3] = {NAN, NAN, NAN}, br[3] = {NAN, NAN, NAN};
    int bi = -1;
    for (size_t ob = 0; ob < n_boxes; ++ob) {
      double oR[9], oh[3] = {0.5 * boxes[ob].size[0], 0.5 * boxes[ob].size[1], 0.5 * boxes[ob].size[2]}, pr[3], pe[3];
      quat_to_mat(boxes[ob].quat, oR);
      double d = prim_box_closest(m->prim_type[q], o.prim_c[q], o.prim_R[q], m->prim_dim[q], boxes[ob].center, oR, oh, pr, pe);
      if (d < best) { best = d; bi = (int)ob; memcpy(be, pe, sizeof(pe)); memcpy(br, pr, sizeof(pr)); }
    }
    dist[q] = best;
    box[q] = bi;
    for (int c = 0; c < 3; ++c) { p_env[3 * q + c] = be[c]; p_robot[3 * q + c] = br[c]; }
    double* a = row + (size_t)W * q;
    for (int c = 0; c < W; ++c) a[c] = 0.0;
    lb[q] = -1e6;
    if (bi < 0) continue;                 /* empty world: unconstrained */
    if (!(best > 0.0)) { ok = 0; continue; } /* collision_avoidance.cpp:214-218 */
    double n[3] = {br[0] - be[0], br[1] - be[1], br[2] - be[2]}, nn = norm3(n);
    for (int c = 0; c < 3; ++c) n[c] /= nn;
    if (full_body) { /* :267 zeroes the root columns otherwise */
      double r[3] = {br[0] - st[0], br[1] - st[1], br[2] - st[2]}, rxn[3];
      cross3(r, n, rxn);
      for (int c = 0; c < 3; ++c) { a[c] = n[c]; a[3 + c] = rxn[c]; }
    }
    for (int k = 1; k <= m->prim_seg[q]; ++k) {
      if (m->seg_joint[k] < 0) continue;
      double r[3] = {br[0] - jo[k][0], br[1] - jo[k][1], br[2] - jo[k][2]}, v[3];
      cross3(ja[k], r, v);
      a[6 + m->seg_joint[k]] = dot3(n, v);
    }
    if (best < constraint_range) /* damplingBound(min_dist, -gain, range, forbidden) */
      lb[q] = -damping_gain * ((best - forbidden_range) / (constraint_range - forbidden_range));
  }
  return ok;
}

/* SoA batch of unpacked states; outputs with row pitch ld */
int orc_avoid_batch(const mvk_model* m, const mvk_box* boxes, size_t n_boxes, const mvk_params* prm,
                    const double* soa, size_t n, size_t ld, double damping_gain, double constraint_range,
                    double forbidden_range, int full_body, double* dist, int32_t* box, double* p_env,
                    double* p_robot, double* row, double* lb, uint8_t* ok) {
  const int J = m->n_joint, P = m->n_prim, W = 6 + J;
  double st[MVK_MAX_STATE_DIM], d[MVK_MAX_PRIM], pe[3 * MVK_MAX_PRIM], pr[3 * MVK_MAX_PRIM],
      rw[(6 + MVK_MAX_JOINT) * MVK_MAX_PRIM], l[MVK_MAX_PRIM];
  int32_t b[MVK_MAX_PRIM];
  for (size_t i = 0; i < n; ++i) {
    for (int r = 0; r < 7 + J; ++r) st[r] = soa[(size_t)r * ld + i];
    ok[i] = (uint8_t)orc_avoid_state(m, boxes, n_boxes, prm, st, damping_gain, constraint_range, forbidden_range,
                                     full_body, d, b, pe, pr, rw, l);
    for (int q = 0; q < P; ++q) {
      dist[(size_t)q * ld + i] = d[q];
      box[(size_t)q * ld + i] = b[q];
      lb[(size_t)q * ld + i] = l[q];
      for (int c = 0; c < 3; ++c) {
        p_env[(size_t)(3 * q + c) * ld + i] = pe[3 * q + c];
        p_robot[(size_t)(3 * q + c) * ld + i] = pr[3 * q + c];
      }
      for (int c = 0; c < W; ++c) row[(size_t)(W * q + c) * ld + i] = rw[W * q + c];
    }
  }
  return 0;
}
