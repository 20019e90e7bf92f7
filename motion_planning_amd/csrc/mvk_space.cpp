// mvk_space.cpp -- host-side arithmetic of the OMPL state spaces that
// MotionPlanning::planInit builds (sampling_based_method/src/motion_planning.cpp:192-258):
//   mode 0  RealVector[J]
//   mode 1  SE2 = R^2 (w 1) x SO2 (w 0.5)   or   SE3 = R^3 (w 1) x SO3 (w 1)
//   mode 2  base (w 1) + RealVector[J] (w 0.001, :254-255)
// OMPL itself is a third-party dependency that is absent from the reference
// tree; these are its published StateSpace::distance / interpolate /
// validSegmentCount / getMaximumExtent rules.  They are
// used by the OMPL-shaped adapter and by the RRT-Connect planner; the kernels
// carry their own device copy of interpolate for in-kernel edge densification.
#include <cmath>
#include <cstring>

#include "mvk_internal.h"

namespace mvk {
namespace {

constexpr double kPi = 3.14159265358979323846;

struct Leaf {
  bool present = false;
  double dist = 0.0, extent = 0.0;
};

struct View {  // where the parts of an OMPL-layout state live
  int n_base, joint_off, n_joint;
  bool se2, has_base, has_joints;
};

View view_of(const mvk_params* p, int J) {
  View v{};
  v.has_base = p->planning_mode != MVK_MODE_ONLY_JOINTS;
  v.has_joints = p->planning_mode != MVK_MODE_ONLY_BASE;
  v.se2 = p->motion_type == MVK_SE2;
  v.n_base = v.has_base ? (v.se2 ? 3 : 7) : 0;
  v.joint_off = v.n_base;
  v.n_joint = v.has_joints ? J : 0;
  return v;
}

double l2(const double* a, const double* b, int n) {
  double acc = 0.0;
  for (int i = 0; i < n; ++i) {
    const double d = a[i] - b[i];
    acc += d * d;
  }
  return std::sqrt(acc);
}

double angle_gap(double a, double b) {  // SO2StateSpace::distance
  const double d = std::fabs(a - b);
  return d > kPi ? 2.0 * kPi - d : d;
}

double quat_arc(const double* q1, const double* q2) {  // SO3StateSpace::distance
  const double dq = std::fabs(q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2] + q1[3] * q2[3]);
  return dq > 1.0 - 1e-9 ? 0.0 : std::acos(dq);
}

void measure(const mvk_params* p, int J, const double* a, const double* b, Leaf& trans, Leaf& rot,
             Leaf& joints) {
  const View v = view_of(p, J);
  if (v.has_base) {
    const int nt = v.se2 ? 2 : 3;
    trans.present = rot.present = true;
    trans.dist = l2(a, b, nt);
    trans.extent = l2(p->pos_low, p->pos_high, nt);
    if (v.se2) {
      rot.dist = angle_gap(a[2], b[2]);
      rot.extent = kPi;
    } else {
      rot.dist = quat_arc(a + 3, b + 3);
      rot.extent = 0.5 * kPi;
    }
  }
  if (v.has_joints) {
    joints.present = true;
    joints.dist = l2(a + v.joint_off, b + v.joint_off, J);
    joints.extent = l2(p->joint_low, p->joint_high, J);
  }
}

double combine(const mvk_params* p, double trans, double rot, double joints) {
  const double w_rot = p->motion_type == MVK_SE2 ? 0.5 : 1.0;
  switch (p->planning_mode) {
    case MVK_MODE_ONLY_JOINTS: return joints;
    case MVK_MODE_ONLY_BASE: return trans + w_rot * rot;
    default: return (trans + w_rot * rot) + p->joint_subspace_weight * joints;
  }
}

}  // namespace

int space_dim(const mvk_params* p, int J) {
  const View v = view_of(p, J);
  return v.n_base + v.n_joint;
}

}  // namespace mvk

using namespace mvk;

extern "C" int mvk_space_dim(const mvk_params* p, int n_joint) {
  if (!p || n_joint < 0 || n_joint > MVK_MAX_JOINT) return fail(MVK_ERR_INVALID_ARG, "mvk_space_dim: bad argument");
  return space_dim(p, n_joint);
}

extern "C" double mvk_space_distance(const mvk_params* p, int J, const double* s1, const double* s2) {
  Leaf t, r, j;
  measure(p, J, s1, s2, t, r, j);
  return combine(p, t.dist, r.dist, j.dist);
}

extern "C" double mvk_space_max_extent(const mvk_params* p, int J) {
  double zero[MVK_MAX_STATE_DIM] = {0};
  zero[6] = 1.0;
  Leaf t, r, j;
  measure(p, J, zero, zero, t, r, j);
  return combine(p, t.extent, r.extent, j.extent);
}

// validSegmentCount: a leaf space counts factor * ceil(d / (extent * fraction)); a compound space keeps the
// maximum over its leaves and never reads a factor of its own.  The reference sets the factor on config_space_
// itself (motion_planning.cpp:233,244): in mode 0 that is the RealVector leaf, so the factor counts; in mode 1 it
// is the SE2/SE3 compound, whose leaves keep factor 1, while setStateValidityCheckingResolution (:237,248) does
// reach the leaves (quirk Q13, pinned by the reference's mode-1 log path_log/hydrus/planning_log.txt: its edges
// have ceil(d / (extent * 0.005)) states, not 20 times that).  Mode 2 is never given a resolution or a factor
// (quirk Q4) and so runs on OMPL's defaults 1 % / 1.
extern "C" int mvk_space_valid_segment_count(const mvk_params* p, int J, const double* s1,
                                             const double* s2) {
  double fraction = p->validity_resolution;
  int factor = p->valid_segment_count_factor;
  if (p->planning_mode == MVK_MODE_JOINTS_AND_BASE && p->mode2_ompl_defaults) {
    fraction = 0.01;
    factor = 1;
  }
  if (p->planning_mode == MVK_MODE_ONLY_BASE) factor = 1;  // quirk Q13
  Leaf leaf[3];
  measure(p, J, s1, s2, leaf[0], leaf[1], leaf[2]);
  int count = 0;
  for (const Leaf& l : leaf) {
    if (!l.present) continue;
    const int c = factor * static_cast<int>(std::ceil(l.dist / (l.extent * fraction)));
    if (c > count) count = c;
  }
  return count;
}

extern "C" int mvk_space_interpolate(const mvk_params* p, int J, const double* from, const double* to,
                                     double t, double* out) {
  if (!p || !from || !to || !out) return fail(MVK_ERR_INVALID_ARG, "mvk_space_interpolate: null argument");
  const View v = view_of(p, J);
  if (v.has_base) {
    const int nt = v.se2 ? 2 : 3;
    for (int i = 0; i < nt; ++i) out[i] = from[i] + (to[i] - from[i]) * t;
    if (v.se2) {
      // SO2StateSpace::interpolate: along the shorter arc, result wrapped to [-pi, pi]
      double diff = to[2] - from[2];
      if (std::fabs(diff) <= kPi) {
        out[2] = from[2] + diff * t;
      } else {
        diff = diff > 0.0 ? 2.0 * kPi - diff : -2.0 * kPi - diff;
        double yaw = from[2] - diff * t;
        if (yaw > kPi) yaw -= 2.0 * kPi;
        else if (yaw < -kPi) yaw += 2.0 * kPi;
        out[2] = yaw;
      }
    } else {
      // SO3StateSpace::interpolate: slerp, sign flipped on the long way round
      const double* qa = from + 3;
      const double* qb = to + 3;
      const double theta = quat_arc(qa, qb);
      if (theta > 2.220446049250313e-16) {
        const double inv = 1.0 / std::sin(theta);
        const double wa = std::sin((1.0 - t) * theta);
        double wb = std::sin(t * theta);
        if (qa[0] * qb[0] + qa[1] * qb[1] + qa[2] * qb[2] + qa[3] * qb[3] < 0.0) wb = -wb;
        for (int i = 0; i < 4; ++i) out[3 + i] = (qa[i] * wa + qb[i] * wb) * inv;
      } else {
        for (int i = 0; i < 4; ++i) out[3 + i] = qa[i];
      }
    }
  }
  for (int i = 0; i < v.n_joint; ++i) {
    const int k = v.joint_off + i;
    out[k] = from[k] + (to[k] - from[k]) * t;
  }
  return MVK_OK;
}

// MotionPlanning::isStateValid's unpack (motion_planning.cpp:81-129).
extern "C" int mvk_space_unpack(const mvk_params* p, int J, const double* s, double* o) {
  if (!p || !s || !o) return fail(MVK_ERR_INVALID_ARG, "mvk_space_unpack: null argument");
  const View v = view_of(p, J);
  for (int i = 0; i < 7 + J; ++i) o[i] = 0.0;
  o[6] = 1.0;  // identity pose when the mode carries no base (V-spec; the reference leaves an all-zero quaternion)
  if (v.has_base) {
    if (v.se2) {
      o[0] = s[0];
      o[1] = s[1];
      o[5] = std::sin(0.5 * s[2]);  // tf::createQuaternionMsgFromYaw (:93,:114)
      o[6] = std::cos(0.5 * s[2]);
    } else {
      std::memcpy(o, s, 7 * sizeof(double));
    }
  }
  for (int i = 0; i < v.n_joint; ++i) o[7 + i] = s[v.joint_off + i];  // quirk Q3: mode 1 keeps zeros
  return MVK_OK;
}
