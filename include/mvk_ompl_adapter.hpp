// mvk_ompl_adapter.hpp -- OMPL-shaped front end of the C ABI (include/mvk.h).
//
// Drop-in for what sampling_base::MotionPlanning::planInit installs
// (sampling_based_method/src/motion_planning.cpp):
//   :236,:247,:257  si->setStateValidityChecker(boost::bind(&MotionPlanning::isStateValid, this, _1))
//   :238,:249       si->setMotionValidator(MotionValidatorPtr(new DiscreteMotionValidator(si)))
// as classes derived from ompl::base::StateValidityChecker / ompl::base::MotionValidator, overriding
//   bool isValid(const ompl::base::State*) const
//   bool checkMotion(const State* s1, const State* s2) const
//   bool checkMotion(const State* s1, const State* s2, std::pair<State*, double>& lastValid) const
// plus the batched entry points the north star adds:
//   void isValid(const State* const* states, std::size_t n, std::uint8_t* out) const
//   void checkMotions(const State* const* from, const State* const* to, std::size_t n, ok, first_invalid) const
//
// The header is written against the OMPL class names.  OMPL is not installed in the build image, so when
// MVK_OMPL_STUB is defined a minimal look-alike of the handful of OMPL types that are touched is provided (same
// member names as ompl::base::{State, CompoundState, SE2StateSpace::StateType, SE3StateSpace::StateType,
// SO3StateSpace::StateType, RealVectorStateSpace::StateType, SpaceInformationPtr, StateValidityChecker,
// MotionValidator}); the adapter classes below are ONE body of code for both cases.  With real OMPL include
//   <ompl/base/SpaceInformation.h> <ompl/base/StateValidityChecker.h> <ompl/base/MotionValidator.h>
//   <ompl/base/spaces/SE2StateSpace.h> <ompl/base/spaces/SE3StateSpace.h> <ompl/base/spaces/RealVectorStateSpace.h>
// first and leave MVK_OMPL_STUB undefined.  (That build has never been compiled here -- OMPL is absent from the
// image; the stub build is compiled and RUN by tests/test_gpu_adapter.py through csrc/mvk_adapter_check.cpp.)
//
// Errors: OMPL's planner loops do not expect exceptions from a validity checker, so a failing C-ABI call makes
// the state / motion INVALID (the safe verdict), bumps errorCount() and leaves the text in lastError() (and in
// mvk_last_error() of the calling thread).  Only the constructors throw (a null or foreign ctx is a programming error).
#ifndef MVK_OMPL_ADAPTER_HPP_
#define MVK_OMPL_ADAPTER_HPP_

#include <cstddef>
#include <cstdint>
#include <limits>
#include <memory>
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
class SpaceInformation {};
typedef std::shared_ptr<SpaceInformation> SpaceInformationPtr;
// ompl/base/StateValidityChecker.h: the virtuals an override must not hide
class StateValidityChecker {
 public:
  explicit StateValidityChecker(const SpaceInformationPtr& si) : si_(si.get()) {}
  virtual ~StateValidityChecker() = default;
  virtual bool isValid(const State* state) const = 0;
  virtual bool isValid(const State* state, double& dist) const {
    dist = clearance(state);
    return isValid(state);
  }
  virtual double clearance(const State*) const { return 0.0; }

 protected:
  SpaceInformation* si_;
};
// ompl/base/MotionValidator.h
class MotionValidator {
 public:
  explicit MotionValidator(const SpaceInformationPtr& si) : si_(si.get()) {}
  virtual ~MotionValidator() = default;
  virtual bool checkMotion(const State* s1, const State* s2) const = 0;
  virtual bool checkMotion(const State* s1, const State* s2, std::pair<State*, double>& lastValid) const = 0;
  unsigned int getValidMotionCount() const { return valid_; }
  unsigned int getInvalidMotionCount() const { return invalid_; }
  void resetMotionCounter() { valid_ = invalid_ = 0; }

 protected:
  SpaceInformation* si_;
  mutable unsigned int valid_ = 0;
  mutable unsigned int invalid_ = 0;
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

namespace detail {
// error bookkeeping shared by both adapters: nothing is thrown into OMPL's loops
class ErrorLog {
 public:
  unsigned errorCount() const { return errors_; }
  const std::string& lastError() const { return last_error_; }

 protected:
  bool ok(int rc) const {
    if (rc == MVK_OK) return true;
    ++errors_;
    last_error_ = mvk_last_error();
    return false;
  }
  static mvk_params params_of(mvk_ctx* ctx) {
    mvk_params p{};
    if (mvk_get_params(ctx, &p) != MVK_OK) throw std::invalid_argument(std::string("mvk adapter: ") + mvk_last_error());
    return p;
  }
  mutable unsigned errors_ = 0;
  mutable std::string last_error_;
};
}  // namespace detail

// replaces the functor bound at motion_planning.cpp:236,247,257
class StateValidityChecker : public ompl::base::StateValidityChecker, public detail::ErrorLog {
 public:
  StateValidityChecker(const ompl::base::SpaceInformationPtr& si, mvk_ctx* ctx, int n_joint)
      : ompl::base::StateValidityChecker(si), ctx_(ctx), n_joint_(n_joint), params_(params_of(ctx)) {}


  // MotionPlanning::isStateValid (motion_planning.cpp:76-164)
  bool isValid(const ompl::base::State* state) const override {
    double s[MVK_MAX_STATE_DIM];
    flatten_state(params_, n_joint_, state, s);
    // the flag path (a batch of one): the verdict alone, without the distances mvk_is_valid also reports
    std::uint8_t flag = 0;
    if (!ok(mvk_is_valid_batch(ctx_, s, 1, 1, MVK_LAYOUT_OMPL, &flag, nullptr, nullptr, nullptr))) return false;
    return flag != 0;
  }

  // ompl::base::StateValidityChecker::clearance: "the distance to the nearest invalid state ... if the distance is
  // negative, the value of clearance is the penetration depth".  Here: the signed separation between the robot's
  // collision primitives and the world boxes (mvk_result::min_dist, the distance semantics of
  // collision_avoidance.cpp:211-226), +infinity when the collision stage is off (planning mode 0) or the world is
  // empty, NaN after a library error.  The reference installs an ObstacleBasedValidStateSampler allocator
  // (motion_planning.h:168-171); samplers of that family (MaximizeClearanceValidStateSampler) call this.
  double clearance(const ompl::base::State* state) const override {
    double dist = 0.0;
    (void)isValid(state, dist);
    return dist;
  }

  // verdict and clearance from ONE full evaluation (mvk_is_valid); the base class would evaluate twice
  bool isValid(const ompl::base::State* state, double& dist) const override {
    double s[MVK_MAX_STATE_DIM];
    flatten_state(params_, n_joint_, state, s);
    mvk_result r{};
    if (!ok(mvk_is_valid(ctx_, s, &r))) {
      dist = std::numeric_limits<double>::quiet_NaN();
      return false;
    }
    dist = r.min_dist;
    return r.valid != 0;
  }

  // batched isValid(states[N]): one kernel launch for all N states; on an error every out[i] = 0
  virtual void isValid(const ompl::base::State* const* states, std::size_t n, std::uint8_t* out) const {
    if (n == 0) return;
    const int dim = mvk_space_dim(&params_, n_joint_);
    std::vector<double> soa(static_cast<std::size_t>(dim) * n);
    double s[MVK_MAX_STATE_DIM];
    for (std::size_t i = 0; i < n; ++i) {
      flatten_state(params_, n_joint_, states[i], s);
      for (int r = 0; r < dim; ++r) soa[static_cast<std::size_t>(r) * n + i] = s[r];
    }
    if (!ok(mvk_is_valid_batch(ctx_, soa.data(), n, n, MVK_LAYOUT_OMPL, out, nullptr, nullptr, nullptr)))
      for (std::size_t i = 0; i < n; ++i) out[i] = 0;
  }

  const mvk_params& params() const { return params_; }
  mvk_ctx* ctx() const { return ctx_; }
  int jointCount() const { return n_joint_; }

 protected:
  mvk_ctx* ctx_;
  int n_joint_;
  mvk_params params_;
};

// replaces ompl::base::DiscreteMotionValidator installed at motion_planning.cpp:238,249 (and OMPL's default in
// mode 2): the whole edge is densified and checked by ONE kernel launch (the order of evaluation does not
// change the Boolean; lastValid comes from the smallest invalid step).
class MotionValidator : public ompl::base::MotionValidator, public detail::ErrorLog {
 public:
  MotionValidator(const ompl::base::SpaceInformationPtr& si, mvk_ctx* ctx, int n_joint)
      : ompl::base::MotionValidator(si), ctx_(ctx), n_joint_(n_joint), params_(params_of(ctx)) {}

  bool checkMotion(const ompl::base::State* s1, const ompl::base::State* s2) const override {
    Edge e;
    const bool valid = run(s1, s2, e) && e.first == e.nd + 1;
    count(valid);
    return valid;
  }

  // DiscreteMotionValidator semantics: on failure lastValid.second = (first invalid step - 1) / nd and
  // lastValid.first (if non-null) receives interpolate(s1, s2, lastValid.second)
  bool checkMotion(const ompl::base::State* s1, const ompl::base::State* s2,
                   std::pair<ompl::base::State*, double>& lastValid) const override {
    Edge e;
    const bool ran = run(s1, s2, e);
    const bool valid = ran && e.first == e.nd + 1;
    if (!valid) {
      lastValid.second = ran ? static_cast<double>(e.first - 1) / static_cast<double>(e.nd) : 0.0;
      if (lastValid.first) {
        double m[MVK_MAX_STATE_DIM];
        mvk_space_interpolate(&params_, n_joint_, e.a, e.b, lastValid.second, m);
        unflatten_state(params_, n_joint_, m, lastValid.first);
      }
    }
    count(valid);
    return valid;
  }

  // batched form used by planners that grow many edges per iteration; on an error every ok[e] = 0
  void checkMotions(const ompl::base::State* const* from, const ompl::base::State* const* to, std::size_t n,
                    std::uint8_t* ok_out, std::int32_t* first_invalid /*nullable*/) const {
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
    if (!ok(mvk_check_motion_batch(ctx_, a.data(), b.data(), n, n, nd.data(), 0, ok_out, first_invalid))) {
      for (std::size_t e = 0; e < n; ++e) {
        ok_out[e] = 0;
        if (first_invalid) first_invalid[e] = 1;
      }
    }
    for (std::size_t e = 0; e < n; ++e) count(ok_out[e] != 0);
  }

  const mvk_params& params() const { return params_; }

 protected:
  struct Edge {
    double a[MVK_MAX_STATE_DIM], b[MVK_MAX_STATE_DIM];
    int nd = 1;     // validSegmentCount(s1, s2), at least 1
    int first = 1;  // smallest invalid step j in 1..nd, or nd + 1 when every step is valid
  };
  // false when the C-ABI call failed (the edge then counts as invalid from its first step)
  bool run(const ompl::base::State* s1, const ompl::base::State* s2, Edge& e) const {
    flatten_state(params_, n_joint_, s1, e.a);
    flatten_state(params_, n_joint_, s2, e.b);
    e.nd = mvk_space_valid_segment_count(&params_, n_joint_, e.a, e.b);
    if (e.nd < 1) e.nd = 1;
    std::int32_t fi = 1;
    if (!ok(mvk_check_motion_batch(ctx_, e.a, e.b, 1, 1, nullptr, e.nd, nullptr, &fi))) return false;
    e.first = fi;
    return true;
  }
  void count(bool valid) const {
    if (valid) ++valid_;
    else ++invalid_;
  }
  mvk_ctx* ctx_;
  int n_joint_;
  mvk_params params_;
};

}  // namespace mvk

#endif  // MVK_OMPL_ADAPTER_HPP_
