// mvk_adapter_check.cpp -- runs include/mvk_ompl_adapter.hpp (MVK_OMPL_STUB build) the way an OMPL planner would:
// through pointers to the ompl::base base classes, on states built as OMPL state objects.  Built into
// lib/libmvk_adapter_check.so next to libmvk.so; tests/test_gpu_adapter.py loads it with ctypes and compares what
// the adapter returns with the C ABI called directly and with the oracle.  Test infrastructure for the C++ veneer
// that replaces motion_planning.cpp:236-249; not part of the product library.
#define MVK_OMPL_STUB
#include "../../include/mvk_ompl_adapter.hpp"

#include <memory>
#include <vector>

namespace {

namespace ob = ompl::base;

// one OMPL state of the planner's space, owning its parts (what si->allocState() would hand out)
struct Owned {
  ob::SE2StateSpace::StateType se2;
  ob::SE3StateSpace::StateType se3;
  ob::RealVectorStateSpace::StateType rv;
  std::vector<double> values;
  ob::CompoundState compound;
  ob::State* parts[2] = {nullptr, nullptr};
  ob::State* top = nullptr;

  Owned(const mvk_params& p, int n_joint) : values(static_cast<std::size_t>(n_joint > 0 ? n_joint : 1), 0.0) {
    rv.values = values.data();
    ob::State* base = p.motion_type == MVK_SE2 ? static_cast<ob::State*>(&se2) : static_cast<ob::State*>(&se3);
    if (p.planning_mode == MVK_MODE_ONLY_JOINTS) top = &rv;
    else if (p.planning_mode == MVK_MODE_ONLY_BASE) top = base;
    else {
      parts[0] = base;
      parts[1] = &rv;
      compound.components = parts;
      top = &compound;
    }
  }
  Owned(const Owned&) = delete;
  Owned& operator=(const Owned&) = delete;
};

}  // namespace

// s1, s2: [n][dim] row-major states in MVK_LAYOUT_OMPL.  Per pair i:
//   valid_single[2i], [2i+1]  StateValidityChecker::isValid(state) of s1_i, s2_i (virtual call through the base class)
//   valid_batch[2i], [2i+1]   the batched isValid(states[2n]) overload
//   motion2[i]                MotionValidator::checkMotion(s1, s2)
//   motion3[i], last_frac[i], last_state[i][dim]   checkMotion(s1, s2, lastValid) (fraction -1 / state untouched when valid)
//   motions_ok[i], motions_first[i]                the batched checkMotions
//   counters[4]               getValidMotionCount, getInvalidMotionCount, svc.errorCount, mv.errorCount
extern "C" int mvk_adapter_check_run(mvk_ctx* ctx, int n_joint, const double* s1, const double* s2, int n,
                                     std::uint8_t* valid_single, std::uint8_t* valid_batch, std::uint8_t* motion2,
                                     std::uint8_t* motion3, double* last_frac, double* last_state,
                                     std::uint8_t* motions_ok, std::int32_t* motions_first, unsigned* counters) {
  try {
    auto si = std::make_shared<ob::SpaceInformation>();
    mvk::StateValidityChecker svc_impl(si, ctx, n_joint);
    mvk::MotionValidator mv_impl(si, ctx, n_joint);
    const ob::StateValidityChecker* svc = &svc_impl;  // what SpaceInformation holds
    const ob::MotionValidator* mv = &mv_impl;
    const mvk_params prm = svc_impl.params();
    const int dim = mvk_space_dim(&prm, n_joint);
    std::vector<std::unique_ptr<Owned>> a, b;
    std::vector<const ob::State*> all, from, to;
    for (int i = 0; i < n; ++i) {
      a.emplace_back(new Owned(prm, n_joint));
      b.emplace_back(new Owned(prm, n_joint));
      mvk::unflatten_state(prm, n_joint, s1 + static_cast<std::size_t>(i) * dim, a.back()->top);
      mvk::unflatten_state(prm, n_joint, s2 + static_cast<std::size_t>(i) * dim, b.back()->top);
      all.push_back(a.back()->top);
      all.push_back(b.back()->top);
      from.push_back(a.back()->top);
      to.push_back(b.back()->top);
    }
    for (int i = 0; i < 2 * n; ++i) valid_single[i] = svc->isValid(all[i]) ? 1 : 0;
    svc_impl.isValid(all.data(), all.size(), valid_batch);
    Owned scratch(prm, n_joint);
    for (int i = 0; i < n; ++i) {
      motion2[i] = mv->checkMotion(from[i], to[i]) ? 1 : 0;
      std::pair<ob::State*, double> last(scratch.top, -1.0);
      mvk::unflatten_state(prm, n_joint, s1 + static_cast<std::size_t>(i) * dim, scratch.top);
      motion3[i] = mv->checkMotion(from[i], to[i], last) ? 1 : 0;
      last_frac[i] = last.second;
      mvk::flatten_state(prm, n_joint, scratch.top, last_state + static_cast<std::size_t>(i) * dim);
    }
    counters[0] = mv->getValidMotionCount();
    counters[1] = mv->getInvalidMotionCount();
    mv_impl.checkMotions(from.data(), to.data(), static_cast<std::size_t>(n), motions_ok, motions_first);
    counters[2] = svc_impl.errorCount();
    counters[3] = mv_impl.errorCount();
    return 0;
  } catch (const std::exception&) {
    return -101;
  }
}

// a ctx whose calls fail (NULL): the adapters must answer "invalid" and count the error, never throw into the caller
extern "C" int mvk_adapter_check_errors(mvk_ctx* good_ctx, int n_joint, const double* s1, const double* s2, unsigned* out) {
  try {
    auto si = std::make_shared<ob::SpaceInformation>();
    mvk::StateValidityChecker svc(si, good_ctx, n_joint);
    mvk::MotionValidator mv(si, good_ctx, n_joint);
    const mvk_params prm = svc.params();
    const int dim = mvk_space_dim(&prm, n_joint);
    (void)dim;
    Owned a(prm, n_joint), b(prm, n_joint), m(prm, n_joint);
    mvk::unflatten_state(prm, n_joint, s1, a.top);
    mvk::unflatten_state(prm, n_joint, s2, b.top);
    // an edge check left "in flight" makes every other call on the ctx fail with MVK_ERR_INVALID_ARG
    if (mvk_check_motion_batch_begin(good_ctx, s1, s2, 1, 1, nullptr, 4) != MVK_OK) return -102;
    std::pair<ob::State*, double> last(m.top, -1.0);
    out[0] = svc.isValid(a.top) ? 1 : 0;
    out[1] = mv.checkMotion(a.top, b.top) ? 1 : 0;
    out[2] = mv.checkMotion(a.top, b.top, last) ? 1 : 0;
    out[3] = svc.errorCount();
    out[4] = mv.errorCount();
    out[5] = mv.getInvalidMotionCount();
    out[6] = last.second == 0.0 ? 1 : 0;
    out[7] = svc.lastError().empty() ? 0 : 1;
    std::int32_t fi = 0;
    if (mvk_check_motion_batch_end(good_ctx, nullptr, &fi) != MVK_OK) return -103;
    return 0;
  } catch (const std::exception&) {
    return -101;
  }
}

// clearance() and isValid(state, dist) through the base-class pointer (the virtuals a valid-state sampler calls):
// states: [n][dim] row-major in MVK_LAYOUT_OMPL -> clearance[n], valid_with_dist[n], dist[n]
extern "C" int mvk_adapter_check_clearance(mvk_ctx* ctx, int n_joint, const double* states, int n, double* clearance,
                                           std::uint8_t* valid_with_dist, double* dist, unsigned* errors) {
  try {
    auto si = std::make_shared<ob::SpaceInformation>();
    mvk::StateValidityChecker svc_impl(si, ctx, n_joint);
    const ob::StateValidityChecker* svc = &svc_impl;
    const mvk_params prm = svc_impl.params();
    const int dim = mvk_space_dim(&prm, n_joint);
    Owned st(prm, n_joint);
    for (int i = 0; i < n; ++i) {
      mvk::unflatten_state(prm, n_joint, states + static_cast<std::size_t>(i) * dim, st.top);
      clearance[i] = svc->clearance(st.top);
      double d = -12345.0;
      valid_with_dist[i] = svc->isValid(st.top, d) ? 1 : 0;
      dist[i] = d;
    }
    *errors = svc_impl.errorCount();
    return 0;
  } catch (const std::exception&) {
    return -101;
  }
}
