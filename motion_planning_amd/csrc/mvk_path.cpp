// mvk_path.cpp -- host side of the path post-processing that follows the planner
// (SURVEY.md 8(f)-4): MultilinkState's root <-> CoG conversions, discrete-path
// resampling and the clamped uniform B-spline the reference fits through the result.
//
// Reference:
//   MultilinkState::convertRootPose2CogPose / convertCogPose2RootPose
//       aerial_motion_planning_msgs/src/multilink_state.cpp:42-61, 98-118
//   ContinuousPathGenerator::discretePathResampling / calcContinuousPath
//       aerial_motion_planning_msgs/src/continuous_path_generator.cpp:119-230, 232-282
//   BsplineRos::initialize / evaluate          bspline_ros/src/bspline_ros.cpp:45-110
//   generateContinousEulerAngle                multilink_state.cpp:179-215
// tf (Quaternion::angleShortestPath / slerp, Matrix3x3::getRPY) and tinyspline (clamped
// uniform knots, de Boor evaluation, derive) are third-party and absent: their published
// algorithms are restated.  discretePathSmoothing (:57-117) is restated from its call
// parameters (mvk_path_smooth below); its FirFilter / FirFilterQuaternion live in the
// external kalman_filter package and nothing in the reference pins their coefficients:
// PARITY UNPINNED for that one function.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "mvk_internal.h"

namespace mvk {
namespace {

struct Q { double x, y, z, w; };
double qdot(const Q& a, const Q& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
// tf::Quaternion::angleShortestPath
double angle_shortest_path(const Q& a, const Q& b) {
  const double s = std::sqrt(qdot(a, a) * qdot(b, b));
  double c = qdot(a, b) / s;
  if (c < 0.0) c = -c;
  if (c > 1.0) c = 1.0;
  return std::acos(c) * 2.0;
}
// tf::Quaternion::slerp
Q slerp(const Q& a, const Q& b, double t) {
  const double theta = angle_shortest_path(a, b) / 2.0;
  if (theta != 0.0) {
    const double d = 1.0 / std::sin(theta), s0 = std::sin((1.0 - t) * theta), s1 = std::sin(t * theta);
    const double sg = qdot(a, b) < 0.0 ? -1.0 : 1.0;
    return Q{(a.x * s0 + sg * b.x * s1) * d, (a.y * s0 + sg * b.y * s1) * d, (a.z * s0 + sg * b.z * s1) * d,
             (a.w * s0 + sg * b.w * s1) * d};
  }
  return a;
}
void quat_rows(const double q[4], double R[9]) {  // tf::Matrix3x3::setRotation
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double s = 2.0 / (x * x + y * y + z * z + w * w);
  R[0] = 1.0 - s * (y * y + z * z); R[1] = s * (x * y - w * z);       R[2] = s * (x * z + w * y);
  R[3] = s * (x * y + w * z);       R[4] = 1.0 - s * (x * x + z * z); R[5] = s * (y * z - w * x);
  R[6] = s * (x * z - w * y);       R[7] = s * (y * z + w * x);       R[8] = 1.0 - s * (x * x + y * y);
}
// tf::Matrix3x3::getRPY (getEulerYPR, solution 1)
void get_rpy(const double R[9], double& roll, double& pitch, double& yaw) {
  if (std::fabs(R[6]) >= 1.0) {  // gimbal lock
    yaw = 0.0;
    const double delta = std::atan2(R[7], R[8]);
    if (R[6] < 0.0) { pitch = M_PI / 2.0; roll = delta; }
    else { pitch = -M_PI / 2.0; roll = delta; }
    return;
  }
  pitch = -std::asin(R[6]);
  const double c = std::cos(pitch);
  roll = std::atan2(R[7] / c, R[8] / c);
  yaw = std::atan2(R[3] / c, R[0] / c);
}
// multilink_state.cpp:179-215
double continuous_euler(double ang, double prev) {
  if (std::fabs(ang - prev) > M_PI) {
    if (ang > prev) {
      while (std::fabs(ang - prev) > M_PI) {
        ang -= 2 * M_PI;
        if (ang < prev - 2 * M_PI) { ang += 2 * M_PI; break; }
      }
    } else {
      while (std::fabs(ang - prev) > M_PI) {
        ang += 2 * M_PI;
        if (ang > prev + 2 * M_PI) { ang -= 2 * M_PI; break; }
      }
    }
  }
  return ang;
}

// CoG of the chain in the root frame (RobotModel::getCog().p; lumped segment masses)
void cog_in_root(const mvk_model* m, const double* joints, double cog[3]) {
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0}, mass = 0.0;
  cog[0] = cog[1] = cog[2] = 0.0;
  for (int k = 0; k < m->n_seg; ++k) {
    if (k > 0) {
      for (int r = 0; r < 3; ++r) p[r] += R[3 * r] * m->seg_t[k][0] + R[3 * r + 1] * m->seg_t[k][1] + R[3 * r + 2] * m->seg_t[k][2];
      if (m->seg_axis[k] >= 0) {
        const double th = m->seg_joint[k] >= 0 ? joints[m->seg_joint[k]] : 0.0, c = std::cos(th), s = std::sin(th);
        const int a = (m->seg_axis[k] + 1) % 3, b = (m->seg_axis[k] + 2) % 3;
        for (int r = 0; r < 3; ++r) {
          const double ua = R[3 * r + a], ub = R[3 * r + b];
          R[3 * r + a] = c * ua + s * ub;
          R[3 * r + b] = c * ub - s * ua;
        }
      }
    }
    for (int r = 0; r < 3; ++r)
      cog[r] += m->seg_mass[k] * (p[r] + R[3 * r] * m->seg_com[k][0] + R[3 * r + 1] * m->seg_com[k][1] + R[3 * r + 2] * m->seg_com[k][2]);
    mass += m->seg_mass[k];
  }
  for (int r = 0; r < 3; ++r) cog[r] /= mass;
}

}  // namespace
}  // namespace mvk

using namespace mvk;

// state rows: px py pz qx qy qz qw j1..jJ.  "cog state": position = CoG, quaternion = desired
// baselink attitude (= root attitude, baselink is the root link: motion_planning.cpp:188).
extern "C" int mvk_root_to_cog(const mvk_model* m, const double* root_state, double* cog_state) {
  if (!m || !root_state || !cog_state) return fail(MVK_ERR_INVALID_ARG, "null argument");
  double R[9], c[3];
  quat_rows(root_state + 3, R);
  cog_in_root(m, root_state + 7, c);
  std::memmove(cog_state, root_state, sizeof(double) * (7 + m->n_joint));
  const double p[3] = {root_state[0], root_state[1], root_state[2]};
  for (int r = 0; r < 3; ++r) cog_state[r] = p[r] + R[3 * r] * c[0] + R[3 * r + 1] * c[1] + R[3 * r + 2] * c[2];
  return MVK_OK;
}

extern "C" int mvk_cog_to_root(const mvk_model* m, const double* cog_state, double* root_state) {
  if (!m || !cog_state || !root_state) return fail(MVK_ERR_INVALID_ARG, "null argument");
  double R[9], c[3];
  quat_rows(cog_state + 3, R);
  cog_in_root(m, cog_state + 7, c);
  const double p[3] = {cog_state[0], cog_state[1], cog_state[2]};
  std::memmove(root_state, cog_state, sizeof(double) * (7 + m->n_joint));
  for (int r = 0; r < 3; ++r) root_state[r] = p[r] - (R[3 * r] * c[0] + R[3 * r + 1] * c[1] + R[3 * r + 2] * c[2]);
  return MVK_OK;
}

// ContinuousPathGenerator::discretePathResampling (continuous_path_generator.cpp:119-230) on cog
// states [n][7+J] (row major).  Returns the number of states written, negative on error.
// ContinuousPathGenerator::discretePathSmoothing (continuous_path_generator.cpp:57-117).  PARITY UNPINNED: the
// filters are kalman_filter's FirFilter(filter_rate, dim) / FirFilterQuaternion(filter_rate) (lpf_filter.h, package
// absent from the reference tree and from this image).  What the call site fixes, and what is restated:
//   * one filter over (root xyz, planned joints), one over the root orientation, both of factor filter_rate (0.1, :61)
//   * both start filled with the first state (setInitValues, :76,:80)
//   * the loop feeds the n states and then repeats the last one until index < n + 1 / filter_rate (:83-86): the
//     comparison is in floating point, so the output has n + 1 / filter_rate states when that is an integer and
//     n + ceil(1 / filter_rate) otherwise; it ends ON the last state once the window holds nothing else
//   * [UPSTREAM-RECOLLECTION] the filter is a moving average over the last round(1 / filter_rate) inputs; the
//     quaternion one averages the sign-aligned quaternions of the window and renormalises
// states: row major [n][7 + J] root states (what MultilinkState holds); returns the number of states written.
extern "C" int mvk_path_smooth(const double* states, int n, int n_joint, double filter_rate, double* out, int cap) {
  if (!states || n < 1 || n_joint < 0 || n_joint > MVK_MAX_JOINT) return fail(MVK_ERR_INVALID_ARG, "mvk_path_smooth: bad argument");
  if (!(filter_rate > 0.0 && filter_rate <= 1.0)) return fail(MVK_ERR_INVALID_ARG, "mvk_path_smooth: filter_rate must be in (0, 1]");
  const int w = 7 + n_joint;
  const int window = std::max(1, static_cast<int>(std::lround(1.0 / filter_rate)));
  // indices with  index < n + 1 / filter_rate  in floating point (:83): n + 1/rate states when 1/rate is an integer
  // (the default 0.1: n + 10), n + ceil(1/rate) otherwise (0.3: n + 4)
  const int total = static_cast<int>(std::ceil(static_cast<double>(n) + 1.0 / filter_rate - 1e-9));
  if (!out) return total;
  if (cap < total) return fail(MVK_ERR_CAPACITY, "mvk_path_smooth: output holds %d states, %d needed", cap, total);
  // ring of the last `window` inputs, all equal to the first state at the start
  std::vector<double> ring(static_cast<size_t>(window) * w);
  for (int k = 0; k < window; ++k) std::memcpy(&ring[static_cast<size_t>(k) * w], states, sizeof(double) * w);
  for (int index = 0; index < total; ++index) {
    const double* in = states + static_cast<size_t>(std::min(index, n - 1)) * w;
    std::memcpy(&ring[static_cast<size_t>(index % window) * w], in, sizeof(double) * w);
    double* o = out + static_cast<size_t>(index) * w;
    for (int c = 0; c < w; ++c) o[c] = 0.0;
    for (int k = 0; k < window; ++k) {
      const double* r = &ring[static_cast<size_t>(k) * w];
      // q and -q are the same attitude: align every quaternion of the window with the newest input
      const double sgn = (r[3] * in[3] + r[4] * in[4] + r[5] * in[5] + r[6] * in[6]) < 0.0 ? -1.0 : 1.0;
      for (int c = 0; c < 3; ++c) o[c] += r[c];
      for (int c = 3; c < 7; ++c) o[c] += sgn * r[c];
      for (int c = 7; c < w; ++c) o[c] += r[c];
    }
    const double inv = 1.0 / window;
    for (int c = 0; c < 3; ++c) o[c] *= inv;
    for (int c = 7; c < w; ++c) o[c] *= inv;
    const double qn = std::sqrt(o[3] * o[3] + o[4] * o[4] + o[5] * o[5] + o[6] * o[6]);
    if (!(qn > 0.0)) return fail(MVK_ERR_INVALID_ARG, "mvk_path_smooth: quaternions of state %d cancel out", index);
    for (int c = 3; c < 7; ++c) o[c] /= qn;
  }
  return total;
}

extern "C" int mvk_path_resample(const double* cog_states, int n, int n_joint, double angular_rate,
                                 double seg_diff_thresh, double* out, int cap) {
  if (!cog_states || !out || n < 1 || n_joint < 0 || n_joint > MVK_MAX_JOINT) return fail(MVK_ERR_INVALID_ARG, "mvk_path_resample: bad argument");
  const int D = 7 + n_joint;
  auto P = [&](const double* s, int c) { return s[c]; };
  auto quat = [&](const double* s) { return Q{s[3], s[4], s[5], s[6]}; };
  auto metric = [&](const double* a, const double* b) {
    const double dx = P(a, 0) - P(b, 0), dy = P(a, 1) - P(b, 1), dz = P(a, 2) - P(b, 2);
    return std::sqrt(dx * dx + dy * dy + dz * dz) + angular_rate * std::fabs(angle_shortest_path(quat(a), quat(b)));
  };
  double path_length = 0.0;
  for (int i = 0; i + 1 < n; ++i) path_length += metric(cog_states + D * i, cog_states + D * (i + 1));  // :128-140
  const double ave = path_length / n;                                                                     // :142
  std::vector<double> res(cog_states, cog_states + D);                                                    // :149
  auto back = [&]() { return res.data() + res.size() - D; };
  double delta = 0.0;
  bool not_enough = false;
  for (int i = 1; i < n;) {
    const double* prev = cog_states + D * (i - 1);
    const double* cur = cog_states + D * i;
    if (delta == 0.0) delta = metric(back(), cur);                 // :163
    if (not_enough) { delta += metric(prev, cur); not_enough = false; }  // :164-168
    if (std::fabs(delta - ave) < seg_diff_thresh * ave) {          // :178-188
      res.insert(res.end(), cur, cur + D);
      delta = 0.0;
      ++i;
    } else if (delta > ave) {                                      // :189-220
      const double rate = 1.0 - (delta - ave) / metric(prev, cur);
      std::vector<double> s(D);
      for (int c = 0; c < 3; ++c) s[c] = prev[c] * (1.0 - rate) + cur[c] * rate;
      const Q q = slerp(quat(prev), quat(cur), rate);
      s[3] = q.x; s[4] = q.y; s[5] = q.z; s[6] = q.w;
      for (int j = 0; j < n_joint; ++j) s[7 + j] = prev[7 + j] * (1.0 - rate) + cur[7 + j] * rate;
      res.insert(res.end(), s.begin(), s.end());
      delta = metric(back(), cur);
      if (static_cast<int>(res.size() / D) > 64 * n + 1024) return fail(MVK_ERR_CAPACITY, "mvk_path_resample: did not converge");
    } else {                                                       // :221-226
      not_enough = true;
      ++i;
    }
  }
  const int m = static_cast<int>(res.size() / D);
  if (m > cap) return fail(MVK_ERR_CAPACITY, "mvk_path_resample: %d states, room for %d", m, cap);
  std::memcpy(out, res.data(), sizeof(double) * res.size());
  return m;
}

// ---- B-spline ---------------------------------------------------------------------
struct mvk_spline {
  int degree = 0, dim = 0, n_ctrl = 0, n_joint = 0;
  double t0 = 0.0, t1 = 1.0;
  // spline k = k-th derivative of the curve: control points [n_ctrl - k][dim], degree - k, its own knots
  std::vector<std::vector<double>> ctrl, knots;
  void* d_blob = nullptr;  // device copy (mvk_spline.hip), created on first device evaluation
  int d_device = -1;
};

namespace mvk {
const std::vector<double>& spline_ctrl(const mvk_spline* s, int k) { return s->ctrl[k]; }
const std::vector<double>& spline_knots(const mvk_spline* s, int k) { return s->knots[k]; }
void spline_shape(const mvk_spline* s, int* degree, int* dim, int* n_ctrl, double* t0, double* t1) {
  *degree = s->degree; *dim = s->dim; *n_ctrl = s->n_ctrl; *t0 = s->t0; *t1 = s->t1;
}
void** spline_device_slot(mvk_spline* s, int** device) { *device = &s->d_device; return &s->d_blob; }
void spline_device_release(mvk_spline* s);  // mvk_spline.hip
}  // namespace mvk

// calcContinuousPath (:232-282): control points = cog xyz, rpy of the baselink attitude with a
// continuous yaw, joints; first and last state doubled; clamped uniform B-spline over [0, period].
extern "C" mvk_spline* mvk_spline_create(const double* cog_states, int n, int n_joint, double period, int degree) {
  if (!cog_states || n < 1 || n_joint < 0 || n_joint > MVK_MAX_JOINT || degree < 1 || !(period > 0.0)) {
    fail(MVK_ERR_INVALID_ARG, "mvk_spline_create: bad argument");
    return nullptr;
  }
  const int D = 7 + n_joint, dim = 6 + n_joint, nc = n + 2;
  if (nc <= degree) {  // bspline_ros.cpp:47-51
    fail(MVK_ERR_INVALID_ARG, "mvk_spline_create: %d control points for degree %d", nc, degree);
    return nullptr;
  }
  mvk_spline* s = new (std::nothrow) mvk_spline();
  if (!s) { fail(MVK_ERR_CAPACITY, "out of memory"); return nullptr; }
  s->degree = degree; s->dim = dim; s->n_ctrl = nc; s->n_joint = n_joint; s->t0 = 0.0; s->t1 = period;
  std::vector<double> c(static_cast<size_t>(nc) * dim);
  double prev_yaw = 0.0;
  for (int i = -1; i < n + 1; ++i) {
    const int id = std::min(std::max(i, 0), n - 1);
    const double* st = cog_states + static_cast<size_t>(D) * id;
    double R[9], r, p, y;
    quat_rows(st + 3, R);
    get_rpy(R, r, p, y);
    if (id == 0) prev_yaw = y;
    y = continuous_euler(y, prev_yaw);
    prev_yaw = y;
    double* o = c.data() + static_cast<size_t>(i + 1) * dim;
    o[0] = st[0]; o[1] = st[1]; o[2] = st[2]; o[3] = r; o[4] = p; o[5] = y;
    for (int j = 0; j < n_joint; ++j) o[6 + j] = st[7 + j];
  }
  // tinyspline TS_CLAMPED knots on [0, 1]: degree+1 zeros, uniform interior, degree+1 ones
  std::vector<double> u(nc + degree + 1);
  for (int i = 0; i < static_cast<int>(u.size()); ++i)
    u[i] = i <= degree ? 0.0 : (i >= nc ? 1.0 : static_cast<double>(i - degree) / static_cast<double>(nc - degree));
  s->ctrl.push_back(c);
  s->knots.push_back(u);
  // derive(): Q_i = p (P_{i+1} - P_i) / (u_{i+p+1} - u_{i+1}), knots without the two end knots
  for (int k = 1; k < degree; ++k) {  // bspline_ros.cpp:83 keeps degree-1 derivatives
    const std::vector<double>& pc = s->ctrl.back();
    const std::vector<double>& pu = s->knots.back();
    const int pdeg = degree - (k - 1), pn = nc - (k - 1);
    std::vector<double> qc(static_cast<size_t>(pn - 1) * dim), qu(pu.begin() + 1, pu.end() - 1);
    for (int i = 0; i < pn - 1; ++i) {
      const double den = pu[i + pdeg + 1] - pu[i + 1];
      for (int d = 0; d < dim; ++d)
        qc[static_cast<size_t>(i) * dim + d] = den > 0.0 ? pdeg * (pc[static_cast<size_t>(i + 1) * dim + d] - pc[static_cast<size_t>(i) * dim + d]) / den : 0.0;
    }
    s->ctrl.push_back(qc);
    s->knots.push_back(qu);
  }
  return s;
}

extern "C" void mvk_spline_destroy(mvk_spline* s) {
  if (!s) return;
  spline_device_release(s);
  delete s;
}

extern "C" int mvk_spline_dim(const mvk_spline* s) { return s ? s->dim : fail(MVK_ERR_INVALID_ARG, "null spline"); }

// BsplineRos::evaluate (bspline_ros.cpp:88-110): clamp t to [0, period], map to [0,1], de Boor on
// the derive-th derivative spline, scale by period^-derive.  out: [nt][dim] row major.
extern "C" int mvk_spline_eval(const mvk_spline* s, const double* t, int nt, int derive, double* out) {
  if (!s || (nt > 0 && (!t || !out)) || nt < 0) return fail(MVK_ERR_INVALID_ARG, "mvk_spline_eval: bad argument");
  if (derive < 0 || derive >= static_cast<int>(s->ctrl.size())) return fail(MVK_ERR_INVALID_ARG, "derivative %d exceeds the %d kept", derive, static_cast<int>(s->ctrl.size()) - 1);
  const std::vector<double>& c = s->ctrl[derive];
  const std::vector<double>& u = s->knots[derive];
  const int p = s->degree - derive, n = s->n_ctrl - derive, dim = s->dim;
  const double span = s->t1 - s->t0, scale = 1.0 / std::pow(span, derive);
  std::vector<double> w(static_cast<size_t>(p + 1) * dim);
  for (int q = 0; q < nt; ++q) {
    double x = t[q];
    if (x > s->t1) x = s->t1; else if (x < s->t0) x = s->t0;
    x = (x - s->t0) / span;
    int k = p;  // knot span: u[k] <= x < u[k+1], last span for x == 1
    while (k < n - 1 && x >= u[k + 1]) ++k;
    for (int j = 0; j <= p; ++j) std::memcpy(&w[static_cast<size_t>(j) * dim], &c[static_cast<size_t>(k - p + j) * dim], sizeof(double) * dim);
    for (int r = 1; r <= p; ++r)
      for (int j = p; j >= r; --j) {
        const double den = u[j + 1 + k - r] - u[j + k - p];
        const double a = den > 0.0 ? (x - u[j + k - p]) / den : 0.0;
        for (int d = 0; d < dim; ++d) w[static_cast<size_t>(j) * dim + d] = (1.0 - a) * w[static_cast<size_t>(j - 1) * dim + d] + a * w[static_cast<size_t>(j) * dim + d];
      }
    for (int d = 0; d < dim; ++d) out[static_cast<size_t>(q) * dim + d] = w[static_cast<size_t>(p) * dim + d] * scale;
  }
  return MVK_OK;
}
