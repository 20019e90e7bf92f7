// mvk_ompl_adapter.hpp -- OMPL-shaped front end of the C ABI (include/mvk.h).
//
// Drop-in for what sampling_base::MotionPlanning::planInit installs
// (sampling_based_method/src/motion_planning.cpp):
//   :236,:247,:257  si->setStateValidityChecker(boost::bind(&MotionPlanning::isStateValid, this, _1))
//   :238,:249       si->setMotionValidator(MotionValidatorPtr(new DiscreteMotionValidator(si)))
// with the same virtual signatures
//   bool isValid(const ompl::base::State*) const
//   bool checkMotion(const State* s1, const State* s2) const
//   bool checkMotion(const State* s1, const State* s2, std::pair<State*, double>& lastValid) const
// plus the batched entry point the north star adds:
//   void isValid(const State* const* states, std::size_t n, std::uint8_t* out) const
//
// The header is written against the OMPL class names.  OMPL is not installed in
// the build image, so when MVK_OMPL_STUB is defined a minimal look-alike of the
// handful of OMPL types that are touched is provided (same member names as
// ompl::base::{State, CompoundState, SE2StateSpace::StateType,
// SE3StateSpace::StateType, SO3StateSpace::StateType,
// RealVectorStateSpace::StateType}); with real OMPL simply include the OMPL
// headers first and leave MVK_OMPL_STUB undefined.
#ifndef MVK_OMPL_ADAPTER_HPP_
#define MVK_OMPL_ADAPTER_HPP_

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "mvk.h"

#ifdef MVK_OMPL_STUB
namespace ompl {
namespace base {
class State {
 public:
  virtual ~State() = default;
  template <class T> const T* as() const { return static_cast<const T*>(this); }
  template <class T> T* as() { return static_cast<T*>(this); }
};
class CompoundState : public State {
 public:
  State** components = nullptr;
  template <class T> const T* as(unsigned i) const { return static_cast<const T*>(components[i]); }
  template <class T> T* as(unsigned i) { return static_cast<T*>(components[i]); }
};
struct RealVectorStateSpace {
  class StateType : public State {
   public:
    double* values = nullptr;
  };
};
struct SO3StateSpace {
  class StateType : public State {
   public:
    double x = 0, y = 0, z = 0, w = 1;
  };
};
struct SE2StateSpace {
  class StateType : public State {
   public:
    double x_ = 0, y_ = 0, yaw_ = 0;
    double getX() const { return x_; }
    double getY() const { return y_; }
    double getYaw() const { return yaw_; }
    void setXY(double x, double y) { x_ = x; y_ = y; }
    void setYaw(double yaw) { yaw_ = yaw; }
  };
};
struct SE3StateSpace {
  class StateType : public State {
   public:
    double x_ = 0, y_ = 0, z_ = 0;
    SO3StateSpace::StateType rot_;
    double getX() const { return x_; }
    double getY() const { return y_; }
    double getZ() const { return z_; }
    void setXYZ(double x, double y, double z) { x_ = x; y_ = y; z_ = z; }
    const SO3StateSpace::StateType& rotation() const { return rot_; }
    SO3StateSpace::StateType& rotation() { return rot_; }
  };
};
}  // namespace base
}  // namespace ompl
#endif  // MVK_OMPL_STUB

namespace mvk {

// Flattens an OMPL state of the planner's space into MVK_LAYOUT_OMPL, reading
// exactly the members MotionPlanning::isStateValid reads (motion_planning.cpp:81-129).
inline void flatten_state(const mvk_params& p, int n_joint, const ompl::base::State* state, double* out) {
  namespace ob = ompl::base;
  int off = 0;
  auto put_base = [&](const ob::State* b) {
    if (p.motion_type == MVK_SE2) {
      const auto* s = b->as<ob::SE2StateSpace::StateType>();
      out[0] = s->getX(); out[1] = s->getY(); out[2] = s->getYaw();
      off = 3;
    } else {
      const auto* s = b->as<ob::SE3StateSpace::StateType>();
      out[0] = s->getX(); out[1] = s->getY(); out[2] = s->getZ();
      out[3] = s->rotation().x; out[4] = s->rotation().y; out[5] = s->rotation().z; out[6] = s->rotation().w;
      off = 7;
    }
  };
  if (p.planning_mode == MVK_MODE_ONLY_JOINTS) {
    const double* v = state->as<ob::RealVectorStateSpace::StateType>()->values;
    for (int j = 0; j < n_joint; ++j) out[j] = v[j];
  } else if (p.planning_mode == MVK_MODE_ONLY_BASE) {
    put_base(state);
  } else {
    const auto* c = static_cast<const ob::CompoundState*>(state);
    put_base(c->as<ob::State>(0));
    const double* v = c->as<ob::RealVectorStateSpace::StateType>(1)->values;
    for (int j = 0; j < n_joint; ++j) out[off + j] = v[j];
  }
}

// Writes a flat MVK_LAYOUT_OMPL state back into an OMPL state (for lastValid).
inline void unflatten_state(const mvk_params& p, int n_joint, const double* in, ompl::base::State* state) {
  namespace ob = ompl::base;
  int off = 0;
  auto get_base = [&](ob::State* b) {
    if (p.motion_type == MVK_SE2) {
      auto* s = b->as<ob::SE2StateSpace::StateType>();
      s->setXY(in[0], in[1]);
      s->setYaw(in[2]);
      off = 3;
    } else {
      auto* s = b->as<ob::SE3StateSpace::StateType>();
      s->setXYZ(in[0], in[1], in[2]);
      s->rotation().x = in[3]; s->rotation().y = in[4]; s->rotation().z = in[5]; s->rotation().w = in[6];
      off = 7;
    }
  };
  if (p.planning_mode == MVK_MODE_ONLY_JOINTS) {
    double* v = state->as<ob::RealVectorStateSpace::StateType>()->values;
    for (int j = 0; j < n_joint; ++j) v[j] = in[j];
  } else if (p.planning_mode == MVK_MODE_ONLY_BASE) {
    get_base(state);
  } else {
    auto* c = static_cast<ob::CompoundState*>(state);
    get_base(c->as<ob::State>(0));
    double* v = c->as<ob::RealVectorStateSpace::StateType>(1)->values;
    for (int j = 0; j < n_joint; ++j) v[j] = in[off + j];
  }
}

// ompl::base::StateValidityChecker look-alike (derive from the real class when
// OMPL headers are present: the member signatures are identical).
class StateValidityChecker
#ifndef MVK_OMPL_STUB
    : public ompl::base::StateValidityChecker
#endif
{
 public:
#ifndef MVK_OMPL_STUB
  StateValidityChecker(const ompl::base::SpaceInformationPtr& si, mvk_ctx* ctx, int n_joint)
      : ompl::base::StateValidityChecker(si), ctx_(ctx), n_joint_(n_joint) { init(); }
#else
  StateValidityChecker(mvk_ctx* ctx, int n_joint) : ctx_(ctx), n_joint_(n_joint) { init(); }
  virtual ~StateValidityChecker() = default;
#endif

  // replaces MotionPlanning::isStateValid (motion_planning.cpp:76-164)
  virtual bool isValid(const ompl::base::State* state) const {
    double s[MVK_MAX_STATE_DIM];
    flatten_state(params_, n_joint_, state, s);
    // the flag path (a batch of one): the verdict alone, without the distances mvk_is_valid also reports
    std::uint8_t flag = 0;
    check(mvk_is_valid_batch(ctx_, s, 1, 1, MVK_LAYOUT_OMPL, &flag, nullptr, nullptr, nullptr));
    return flag != 0;
  }

  // batched isValid(states[N]): one kernel launch for all N states
  virtual void isValid(const ompl::base::State* const* states, std::size_t n, std::uint8_t* out) const {
    if (n == 0) return;
    const int dim = mvk_space_dim(&params_, n_joint_);
    std::vector<double> soa(static_cast<std::size_t>(dim) * n);
    double s[MVK_MAX_STATE_DIM];
    for (std::size_t i = 0; i < n; ++i) {
      flatten_state(params_, n_joint_, states[i], s);
      for (int r = 0; r < dim; ++r) soa[static_cast<std::size_t>(r) * n + i] = s[r];
    }
    check(mvk_is_valid_batch(ctx_, soa.data(), n, n, MVK_LAYOUT_OMPL, out, nullptr, nullptr, nullptr));
  }

  const mvk_params& params() const { return params_; }
  mvk_ctx* ctx() const { return ctx_; }
  int jointCount() const { return n_joint_; }

 protected:
  static void check(int rc) {
    if (rc != MVK_OK) throw std::runtime_error(std::string("mvk: ") + mvk_last_error());
  }
  void init() { check(mvk_get_params(ctx_, &params_)); }
  mvk_ctx* ctx_;
  int n_joint_;
  mvk_params params_{};
};

// ompl::base::DiscreteMotionValidator look-alike: the whole edge is densified and
// checked by ONE kernel launch (order of evaluation does not change the Boolean).
class MotionValidator
#ifndef MVK_OMPL_STUB
    : public ompl::base::MotionValidator
#endif
{
 public:
#ifndef MVK_OMPL_STUB
  MotionValidator(const ompl::base::SpaceInformationPtr& si, mvk_ctx* ctx, int n_joint)
      : ompl::base::MotionValidator(si), ctx_(ctx), n_joint_(n_joint) { init(); }
#else
  MotionValidator(mvk_ctx* ctx, int n_joint) : ctx_(ctx), n_joint_(n_joint) { init(); }
  virtual ~MotionValidator() = default;
#endif

  virtual bool checkMotion(const ompl::base::State* s1, const ompl::base::State* s2) const {
    int first = 0;
    return run(s1, s2, first) != 0;
  }

  // lastValid.first (if non-null) receives interpolate(s1, s2, lastValid.second), as OMPL does
  virtual bool checkMotion(const ompl::base::State* s1, const ompl::base::State* s2,
                           std::pair<ompl::base::State*, double>& lastValid) const {
    int first = 0;
    double a[MVK_MAX_STATE_DIM], b[MVK_MAX_STATE_DIM], m[MVK_MAX_STATE_DIM];
    const int nd = run(s1, s2, first, a, b);
    if (first == nd + 1) return true;
    lastValid.second = static_cast<double>(first - 1) / static_cast<double>(nd);
    if (lastValid.first) {
      mvk_space_interpolate(&params_, n_joint_, a, b, lastValid.second, m);
      unflatten_state(params_, n_joint_, m, lastValid.first);
    }
    return false;
  }

  // batched form used by planners that grow many edges per iteration
  void checkMotions(const ompl::base::State* const* from, const ompl::base::State* const* to, std::size_t n,
                    std::uint8_t* ok, std::int32_t* first_invalid /*nullable*/) const {
    if (n == 0) return;
    const int dim = mvk_space_dim(&params_, n_joint_);
    std::vector<double> a(static_cast<std::size_t>(dim) * n), b(a.size());
    std::vector<std::int32_t> nd(n);
    double x[MVK_MAX_STATE_DIM], y[MVK_MAX_STATE_DIM];
    for (std::size_t e = 0; e < n; ++e) {
      flatten_state(params_, n_joint_, from[e], x);
      flatten_state(params_, n_joint_, to[e], y);
      nd[e] = mvk_space_valid_segment_count(&params_, n_joint_, x, y);
      for (int r = 0; r < dim; ++r) {
        a[static_cast<std::size_t>(r) * n + e] = x[r];
        b[static_cast<std::size_t>(r) * n + e] = y[r];
      }
    }
    check(mvk_check_motion_batch(ctx_, a.data(), b.data(), n, n, nd.data(), 0, ok, first_invalid));
  }

 protected:
  // returns nd (>= 1); first = smallest invalid step or nd + 1
  int run(const ompl::base::State* s1, const ompl::base::State* s2, int& first, double* a_out = nullptr,
          double* b_out = nullptr) const {
    double a[MVK_MAX_STATE_DIM], b[MVK_MAX_STATE_DIM];
    flatten_state(params_, n_joint_, s1, a);
    flatten_state(params_, n_joint_, s2, b);
    int nd = mvk_space_valid_segment_count(&params_, n_joint_, a, b);
    if (nd < 1) nd = 1;
    std::uint8_t ok = 0;
    std::int32_t fi = 0;
    check(mvk_check_motion_batch(ctx_, a, b, 1, 1, nullptr, nd, &ok, &fi));
    first = fi;
    const int dim = mvk_space_dim(&params_, n_joint_);
    if (a_out) for (int r = 0; r < dim; ++r) { a_out[r] = a[r]; b_out[r] = b[r]; }
    return ok ? (first = nd + 1, nd) : nd;
  }
  static void check(int rc) {
    if (rc != MVK_OK) throw std::runtime_error(std::string("mvk: ") + mvk_last_error());
  }
  void init() { check(mvk_get_params(ctx_, &params_)); }
  mvk_ctx* ctx_;
  int n_joint_;
  mvk_params params_{};
};

}  // namespace mvk

#endif  // MVK_OMPL_ADAPTER_HPP_
