// mvk_planner.cpp -- host-side RRT-Connect over the batched validity checker,
// solution densification and the reference's path-log text format.
//
// Reference anchors (paths relative to the reference root):
//   plan / densify   aerial_motion/sampling_based_method/src/motion_planning.cpp:377-463
//   savePath         :644-688        loadPath  :690-800
// The reference instantiates OMPL's RRT* (:396-400); RRT-Connect is this build's own
// planner following the structure of OMPL's RRTConnect:
// two trees rooted at start and goal, EXTEND toward a uniform sample by at most
// `range`, then CONNECT the other tree greedily toward the new node, swap.
// What is MI355X-shaped about it: validity is never asked one state at a time.
// Each iteration proposes `batch` extensions and sends them, TOGETHER with the
// connect edge of every proposal, to ONE edge-check launch; the first_invalid
// output of a connect edge says how far the straight connection got (the greedy
// repeat-extend of RRT-Connect collapses into "advance to the last valid
// interpolation step").
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "mvk_internal.h"

namespace mvk {
namespace {

constexpr double kPi = 3.14159265358979323846;

// splitmix64 / xoshiro256** : small, seedable, reproducible across platforms
struct Rng {
  uint64_t s[4];
  explicit Rng(uint64_t seed) {
    for (auto& v : s) {
      seed += 0x9E3779B97F4A7C15ull;
      uint64_t z = seed;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      v = z ^ (z >> 31);
    }
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  double uniform() { return static_cast<double>(next() >> 11) * (1.0 / 9007199254740992.0); }
  double uniform(double lo, double hi) { return lo + (hi - lo) * uniform(); }
  double normal() {  // Box-Muller
    const double u1 = 1.0 - uniform(), u2 = uniform();
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * kPi * u2);
  }
};

struct Tree {
  int dim;
  std::vector<double> states;  // row-major [n][dim]
  std::vector<int> parent;
  std::vector<std::vector<double>> col;  // the same states column-wise, for the vectorised nearest search
  std::vector<std::vector<float>> colf;  // ... and in fp32, for its first pass
  int size() const { return static_cast<int>(parent.size()); }
  const double* at(int i) const { return states.data() + static_cast<size_t>(i) * dim; }
  int add(const double* s, int par) {
    states.insert(states.end(), s, s + dim);
    parent.push_back(par);
    if (col.empty()) { col.resize(dim); colf.resize(dim); }
    for (int c = 0; c < dim; ++c) { col[c].push_back(s[c]); colf[c].push_back(static_cast<float>(s[c])); }
    return size() - 1;
  }
};

// A few persistent worker threads for the nearest-neighbour searches of a batch: with the validity
// checks on the GPU, the linear scans over the trees are what is left on the host's critical path.
// A search is issued every few tens of microseconds during a solve, and a condition-variable wake-up
// costs about as much as the search itself, so the workers spin for a short while after each job
// (and the caller spins for the completion) before they go back to sleep.
class Workers {
 public:
  static Workers& get() {
    static Workers w;
    return w;
  }
  int count() const { return static_cast<int>(th_.size()) + 1; }
  // runs fn(k) for k in [0, n), the caller taking part
  void parallel_for(int n, const std::function<void(int)>& fn) {
    // one job at a time: a second planner (another thread, another ctx) that finds the pool busy runs its loop itself
    std::unique_lock<std::mutex> busy(busy_, std::try_to_lock);
    if (n <= 1 || th_.empty() || !busy.owns_lock()) {
      for (int k = 0; k < n; ++k) fn(k);
      return;
    }
    // The job description lives in the slot of the epoch's parity: a worker that is one epoch late
    // still reads its own epoch's description, and its compare-exchange on the ticket then fails.
    const uint64_t epoch = (ticket_.load(std::memory_order_relaxed) >> 32) + 1;
    fn_[epoch & 1].store(&fn, std::memory_order_relaxed);
    n_[epoch & 1].store(n, std::memory_order_relaxed);
    done_.store(0, std::memory_order_relaxed);
    ticket_.store(epoch << 32, std::memory_order_release);  // job counter in the low half
    if (sleepers_.load(std::memory_order_acquire) > 0) {
      { std::lock_guard<std::mutex> lk(m_); }
      cv_.notify_all();
    }
    work(epoch, &fn, n);
    while (done_.load(std::memory_order_acquire) != n) cpu_relax();
  }

 private:
  static void cpu_relax() { __builtin_ia32_pause(); }
  std::mutex busy_;
  // take jobs of `epoch` until none is left (or the epoch has moved on: a late worker does nothing)
  void work(uint64_t epoch, const std::function<void(int)>* fn, int n) {
    for (;;) {
      uint64_t t = ticket_.load(std::memory_order_acquire);
      if ((t >> 32) != epoch || static_cast<int>(t & 0xffffffffu) >= n) return;
      if (!ticket_.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel)) continue;
      (*fn)(static_cast<int>(t & 0xffffffffu));
      done_.fetch_add(1, std::memory_order_release);
    }
  }
  Workers() {
    // threads taking part, the caller included: three quarters of the CPUs this process may use (OMP_NUM_THREADS when
    // set: a container's share of a bigger machine), at most 16; MVK_PLAN_THREADS overrides
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("OMP_NUM_THREADS")) {
      const int v = std::atoi(e);
      if (v >= 1 && static_cast<unsigned>(v) < hw) hw = static_cast<unsigned>(v);
    }
    int n = hw > 2 ? static_cast<int>(hw > 21 ? 16 : (3 * hw) / 4) - 1 : 0;
    if (const char* e = std::getenv("MVK_PLAN_THREADS")) {
      const int want = std::atoi(e);
      if (want >= 1 && want <= 64) n = want - 1;
    }
    for (int i = 0; i < n; ++i) th_.emplace_back([this] { loop(); });
  }
  ~Workers() {
    stop_.store(true, std::memory_order_release);
    { std::lock_guard<std::mutex> lk(m_); }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      // wait for a new epoch: spin for about 200 us, then sleep
      int spins = 0;
      while ((ticket_.load(std::memory_order_acquire) >> 32) == seen && !stop_.load(std::memory_order_acquire)) {
        if (++spins < 20000) { cpu_relax(); continue; }
        std::unique_lock<std::mutex> lk(m_);
        sleepers_.fetch_add(1, std::memory_order_acq_rel);
        cv_.wait(lk, [&] { return (ticket_.load(std::memory_order_acquire) >> 32) != seen || stop_.load(std::memory_order_acquire); });
        sleepers_.fetch_sub(1, std::memory_order_acq_rel);
        spins = 0;
      }
      if (stop_.load(std::memory_order_acquire)) return;
      seen = ticket_.load(std::memory_order_acquire) >> 32;
      work(seen, fn_[seen & 1].load(std::memory_order_relaxed), n_[seen & 1].load(std::memory_order_relaxed));
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_;
  std::atomic<const std::function<void(int)>*> fn_[2] = {{nullptr}, {nullptr}};
  std::atomic<int> n_[2] = {{0}, {0}};
  std::atomic<int> done_{0}, sleepers_{0};
  std::atomic<uint64_t> ticket_{0};  // epoch << 32 | next job
  std::atomic<bool> stop_{false};
};

// uniform sample of the planner's space (OMPL's default samplers: box-uniform
// RealVector, uniform SO2 angle, uniform SO3 quaternion)
void sample_uniform(const mvk_params& p, int J, Rng& rng, double* s) {
  int off = 0;
  if (p.planning_mode != MVK_MODE_ONLY_JOINTS) {
    if (p.motion_type == MVK_SE2) {
      s[0] = rng.uniform(p.pos_low[0], p.pos_high[0]);
      s[1] = rng.uniform(p.pos_low[1], p.pos_high[1]);
      s[2] = rng.uniform(-kPi, kPi);
      off = 3;
    } else {
      for (int k = 0; k < 3; ++k) s[k] = rng.uniform(p.pos_low[k], p.pos_high[k]);
      double q[4], n = 0.0;
      for (double& v : q) { v = rng.normal(); n += v * v; }
      n = 1.0 / std::sqrt(n);
      for (int k = 0; k < 4; ++k) s[3 + k] = q[k] * n;
      off = 7;
    }
  }
  if (p.planning_mode != MVK_MODE_ONLY_BASE)
    for (int j = 0; j < J; ++j) s[off + j] = rng.uniform(p.joint_low[j], p.joint_high[j]);
}

// nearest tree node under the space's distance (mvk_space_distance), the tree read column-wise so
// that the per-node arithmetic vectorises; `scratch` holds one distance per node
// SE(2) base + J joints in one pass over the nodes: the whole compound distance per node, then the argmin
template <int JT>
inline __attribute__((always_inline)) int nearest_se2_fused(const mvk_params& p, const Tree& t, int b0, const double* s, double* __restrict__ acc) {
  const int n = t.size() - b0;  // nodes b0 .. size-1
  const double* __restrict__ cx = t.col[0].data() + b0;
  const double* __restrict__ cy = t.col[1].data() + b0;
  const double* __restrict__ cw = t.col[2].data() + b0;
  const double* __restrict__ cj[JT > 0 ? JT : 1];
  double qj[JT > 0 ? JT : 1];
  for (int j = 0; j < JT; ++j) { cj[j] = t.col[3 + j].data() + b0; qj[j] = s[3 + j]; }
  const double qx = s[0], qy = s[1], qw = s[2], w = p.joint_subspace_weight;
#pragma clang loop vectorize(enable) interleave(enable)
  for (int i = 0; i < n; ++i) {
    double sj = 0.0;
#pragma unroll
    for (int j = 0; j < JT; ++j) { const double d = cj[j][i] - qj[j]; sj += d * d; }
    const double dx = cx[i] - qx, dy = cy[i] - qy;
    double g = std::fabs(cw[i] - qw);
    g = g > kPi ? 2.0 * kPi - g : g;
    acc[i] = (std::sqrt(dx * dx + dy * dy) + 0.5 * g) + w * std::sqrt(sj);
  }
  double bd = INFINITY;  // the minimum (vectorises), then its first index
  for (int i = 0; i < n; ++i) bd = acc[i] < bd ? acc[i] : bd;
  int best = 0;
  while (best < n - 1 && acc[best] != bd) ++best;
  return b0 + best;
}

// Two passes: the compound distance of every node in fp32 (eight nodes per AVX2 operation, and fp32 square roots
// cost a quarter of the fp64 ones -- the two per node are most of the scan), then the fp64 distance of the few nodes
// within a slack of the fp32 minimum, with the first-index rule of the one-pass scan.  The slack follows from the
// space's bounds: a coordinate of magnitude <= M rounds to fp32 within M 2^-24, a difference of two within 2 M 2^-24,
// the planar norm within 3 M 2^-24, the sum's roundings add ~2 M 2^-24 -- about 6 (M_pos + pi + w sqrt(J) M_joint)
// 2^-24 per node, twice that between the minimum and a candidate; 32 x is taken.  So the fp64 minimum is always among
// the candidates and the result is the one-pass scan's, whatever the bounds (large bounds only lengthen the list).
inline float near_slack(const mvk_params& p, int J) {
  double mpos = 0.0, mj = 0.0;
  for (int k = 0; k < 3; ++k) mpos = std::fmax(mpos, std::fmax(std::fabs(p.pos_low[k]), std::fabs(p.pos_high[k])));
  for (int j = 0; j < J; ++j) mj = std::fmax(mj, std::fmax(std::fabs(p.joint_low[j]), std::fabs(p.joint_high[j])));
  return static_cast<float>(32.0 * 5.9604645e-8 * (mpos + kPi + p.joint_subspace_weight * std::sqrt(static_cast<double>(J)) * mj));
}
template <int JT>
inline __attribute__((always_inline)) int nearest_se2_two_pass(const mvk_params& p, const Tree& t, int b0, const double* s, float* __restrict__ accf) {
  const int n = t.size() - b0;
  const float* __restrict__ fx = t.colf[0].data() + b0;
  const float* __restrict__ fy = t.colf[1].data() + b0;
  const float* __restrict__ fw = t.colf[2].data() + b0;
  const float* __restrict__ fj[JT > 0 ? JT : 1];
  float qjf[JT > 0 ? JT : 1];
  for (int j = 0; j < JT; ++j) { fj[j] = t.colf[3 + j].data() + b0; qjf[j] = static_cast<float>(s[3 + j]); }
  const float qxf = static_cast<float>(s[0]), qyf = static_cast<float>(s[1]), qwf = static_cast<float>(s[2]);
  const float wf = static_cast<float>(p.joint_subspace_weight), pif = static_cast<float>(kPi);
#pragma clang loop vectorize(enable) interleave(enable)
  for (int i = 0; i < n; ++i) {
    float sj = 0.0f;
#pragma unroll
    for (int j = 0; j < JT; ++j) { const float d = fj[j][i] - qjf[j]; sj += d * d; }
    const float dx = fx[i] - qxf, dy = fy[i] - qyf;
    float g = std::fabs(fw[i] - qwf);
    g = g > pif ? 2.0f * pif - g : g;
    accf[i] = (std::sqrt(dx * dx + dy * dy) + 0.5f * g) + wf * std::sqrt(sj);
  }
  float bf = INFINITY;
  for (int i = 0; i < n; ++i) bf = accf[i] < bf ? accf[i] : bf;
  const float thr = bf + near_slack(p, JT) + 4e-7f * bf;
  // the candidates in fp64, exactly as nearest_se2_fused computes a node
  const double qx = s[0], qy = s[1], qw = s[2], w = p.joint_subspace_weight;
  double bd = INFINITY;
  int best = 0;
  for (int i = 0; i < n; ++i) {
    if (!(accf[i] <= thr)) continue;
    double sj = 0.0;
    for (int j = 0; j < JT; ++j) { const double d = t.col[3 + j][b0 + i] - s[3 + j]; sj += d * d; }
    const double dx = t.col[0][b0 + i] - qx, dy = t.col[1][b0 + i] - qy;
    double g = std::fabs(t.col[2][b0 + i] - qw);
    g = g > kPi ? 2.0 * kPi - g : g;
    const double d = (std::sqrt(dx * dx + dy * dy) + 0.5 * g) + w * std::sqrt(sj);
    if (d < bd) { bd = d; best = i; }
  }
  return b0 + best;
}
__attribute__((target("avx2,fma"))) int nearest_se2_avx2_two_pass(const mvk_params& p, int J, const Tree& t, int b0, const double* s, float* accf) {
  return J == 3 ? nearest_se2_two_pass<3>(p, t, b0, s, accf) : J == 5 ? nearest_se2_two_pass<5>(p, t, b0, s, accf) : nearest_se2_two_pass<7>(p, t, b0, s, accf);
}

// the same loop compiled for AVX2 + FMA (picked at run time) and for the baseline ISA
__attribute__((target("avx2,fma"))) int nearest_se2_avx2(const mvk_params& p, int J, const Tree& t, int b0, const double* s, double* acc) {
  return J == 3 ? nearest_se2_fused<3>(p, t, b0, s, acc) : J == 5 ? nearest_se2_fused<5>(p, t, b0, s, acc) : nearest_se2_fused<7>(p, t, b0, s, acc);
}
int nearest_se2_base(const mvk_params& p, int J, const Tree& t, int b0, const double* s, double* acc) {
  return J == 3 ? nearest_se2_fused<3>(p, t, b0, s, acc) : J == 5 ? nearest_se2_fused<5>(p, t, b0, s, acc) : nearest_se2_fused<7>(p, t, b0, s, acc);
}

int nearest(const mvk_params& p, int J, const Tree& t, const double* s, double* dist_out, std::vector<double>& scratch, int b0 = 0) {
  const int n = t.size();
  if (p.planning_mode == MVK_MODE_JOINTS_AND_BASE && p.motion_type == MVK_SE2 && (J == 3 || J == 5 || J == 7)) {
    static const bool avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    static const bool one_pass = std::getenv("MVK_PLAN_ONE_PASS") != nullptr;  // diagnostics: the fp64 scan alone
    if (static_cast<int>(scratch.size()) < n) scratch.resize(static_cast<size_t>(n) + 256);
    const int best = (avx2 && !one_pass) ? nearest_se2_avx2_two_pass(p, J, t, b0, s, reinterpret_cast<float*>(scratch.data()))
                     : avx2 ? nearest_se2_avx2(p, J, t, b0, s, scratch.data()) : nearest_se2_base(p, J, t, b0, s, scratch.data());
    *dist_out = mvk_space_distance(&p, J, t.at(best), s);
    return best;
  }
  if (b0 > 0) {  // general spaces: plain scan of the nodes from b0 on
    int best = b0;
    double bd = INFINITY;
    for (int i = b0; i < n; ++i) {
      const double d = mvk_space_distance(&p, J, t.at(i), s);
      if (d < bd) { bd = d; best = i; }
    }
    *dist_out = bd;
    return best;
  }
  scratch.assign(n, 0.0);
  double* acc = scratch.data();
  const bool has_base = p.planning_mode != MVK_MODE_ONLY_JOINTS, has_joints = p.planning_mode != MVK_MODE_ONLY_BASE;
  const bool se2 = p.motion_type == MVK_SE2;
  const int nb = has_base ? (se2 ? 3 : 7) : 0, nt = se2 ? 2 : 3;
  if (has_joints && J > 0) {
    for (int j = 0; j < J; ++j) {
      const double* c = t.col[nb + j].data();
      const double q = s[nb + j];
      for (int i = 0; i < n; ++i) { const double d = c[i] - q; acc[i] += d * d; }
    }
    const double w = has_base ? p.joint_subspace_weight : 1.0;
    for (int i = 0; i < n; ++i) acc[i] = w * std::sqrt(acc[i]);
  }
  if (has_base) {
    std::vector<double> tr(n, 0.0);
    for (int k = 0; k < nt; ++k) {
      const double* c = t.col[k].data();
      const double q = s[k];
      for (int i = 0; i < n; ++i) { const double d = c[i] - q; tr[i] += d * d; }
    }
    if (se2) {
      const double* c = t.col[2].data();
      const double q = s[2];
      for (int i = 0; i < n; ++i) {
        double g = std::fabs(c[i] - q);
        g = g > kPi ? 2.0 * kPi - g : g;
        acc[i] = (std::sqrt(tr[i]) + 0.5 * g) + acc[i];
      }
    } else {
      const double *c0 = t.col[3].data(), *c1 = t.col[4].data(), *c2 = t.col[5].data(), *c3 = t.col[6].data();
      for (int i = 0; i < n; ++i) {
        const double dq = std::fabs(c0[i] * s[3] + c1[i] * s[4] + c2[i] * s[5] + c3[i] * s[6]);
        const double arc = dq > 1.0 - 1e-9 ? 0.0 : std::acos(dq);
        acc[i] = (std::sqrt(tr[i]) + arc) + acc[i];
      }
    }
  }
  int best = 0;
  double bd = INFINITY;
  for (int i = 0; i < n; ++i)
    if (acc[i] < bd) { bd = acc[i]; best = i; }
  // report the distance exactly as the space computes it (the scan may differ in the last bit)
  *dist_out = mvk_space_distance(&p, J, t.at(best), s);
  return best;
}

struct EdgeBatch {  // SoA staging for one check_motions call
  int dim;
  std::vector<double> a, b;
  std::vector<int32_t> nd, first;
  std::vector<uint8_t> ok;
  void pack(const std::vector<const double*>& from, const std::vector<const double*>& to, const mvk_params& p, int J) {
    const size_t n = from.size();
    a.assign(n * dim, 0.0);
    b.assign(n * dim, 0.0);
    nd.resize(n);
    first.assign(n, 0);
    ok.assign(n, 0);
    for (size_t e = 0; e < n; ++e) {
      for (int r = 0; r < dim; ++r) {
        a[static_cast<size_t>(r) * n + e] = from[e][r];
        b[static_cast<size_t>(r) * n + e] = to[e][r];
      }
      int c = mvk_space_valid_segment_count(&p, J, from[e], to[e]);
      nd[e] = c < 1 ? 1 : c;
    }
  }
};

}  // namespace
}  // namespace mvk

using namespace mvk;

extern "C" int mvk_plan_rrt_connect_cb(const mvk_params* params, int J, const mvk_checker* ck,
                                       const mvk_plan_request* req, double* waypoints, int cap,
                                       mvk_plan_result* out) {
  if (!params || !ck || !ck->check_states || !ck->check_motions || !req || !req->start || !req->goal || !out)
    return fail(MVK_ERR_INVALID_ARG, "mvk_plan_rrt_connect: null argument");
  if ((ck->check_motions_begin == nullptr) != (ck->check_motions_end == nullptr))
    return fail(MVK_ERR_INVALID_ARG, "mvk_plan_rrt_connect: check_motions_begin and check_motions_end go together (both or neither)");
  const mvk_params& p = *params;
  const int dim = space_dim(&p, J);
  std::memset(out, 0, sizeof(*out));
  const auto t0 = std::chrono::steady_clock::now();
  auto elapsed_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };

  // start / goal validity (planInit aborts on a colliding start: motion_planning.cpp:364-372)
  {
    std::vector<double> soa(2 * dim);
    for (int r = 0; r < dim; ++r) { soa[2 * r] = req->start[r]; soa[2 * r + 1] = req->goal[r]; }
    uint8_t ok[2] = {0, 0};
    if (int rc = ck->check_states(ck->user, soa.data(), 2, 2, ok)) return rc;
    out->state_checks += 2;
    if (!ok[0] || !ok[1]) {
      out->solve_ms = elapsed_ms();
      return fail(MVK_ERR_NO_SOLUTION, "%s state is invalid, stop planning", !ok[0] ? "start" : "goal");
    }
  }
  const double range = req->range > 0.0 ? req->range : 0.2 * mvk_space_max_extent(&p, J);
  const int batch = req->batch < 1 ? 1 : req->batch;
  Rng rng(req->seed);
  Tree tree[2] = {Tree{dim, {}, {}, {}}, Tree{dim, {}, {}, {}}};
  tree[0].add(req->start, -1);
  tree[1].add(req->goal, -1);
  EdgeBatch eb;
  eb.dim = dim;
  // MVK_PLAN_TRACE=1: where the wall time of the search goes (stderr)
  const bool trace = std::getenv("MVK_PLAN_TRACE") != nullptr;
  double t_sample = 0, t_near = 0, t_pack = 0, t_check = 0, t_tree = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  int link[2] = {-1, -1};  // indices of the two nodes that meet (same state in both trees)
  bool solved = false;
  int cur = 0;

  std::vector<const double*> from, to;
  std::vector<int> from_idx, near_a, near_idx, near_conn, fresh;
  std::vector<double> samples, targets, dist_a, dist_conn;
  bool have_next = false;  // samples / near_a / dist_a already hold the coming iteration's batch
  const bool merged = ck->check_motions_begin != nullptr && ck->check_motions_end != nullptr;
  std::vector<uint8_t> conn_ok, ext_ok;
  std::vector<int32_t> conn_first, conn_nd;
  // nearest node of `t` among nodes b0.. for each of n query states (row-major); `merge` keeps an
  // earlier result when it is at least as close (ties go to the lower index, as one scan would)
  auto search_all = [&](const Tree& t, int b0, const double* q, int n, std::vector<int>& idx, std::vector<double>& dst, bool merge) {
    if (!merge) { idx.resize(n); dst.resize(n); }
    const bool wide = static_cast<int64_t>(n) * (t.size() - b0) > 20000;
    auto job = [&](int k) {
      thread_local std::vector<double> scratch;
      double d;
      const int i = nearest(p, J, t, q + static_cast<size_t>(k) * dim, &d, scratch, b0);
      if (!merge || d < dst[k]) { idx[k] = i; dst[k] = d; }
    };
    if (wide) {
      // a ticket per query is a contended compare-exchange per ~0.2 us of work: hand the queries out in chunks
      const int chunk = std::max(1, std::atoi(std::getenv("MVK_PLAN_CHUNK") ? std::getenv("MVK_PLAN_CHUNK") : "16"));
      Workers::get().parallel_for((n + chunk - 1) / chunk, [&](int c) {
        for (int k = c * chunk; k < std::min(n, (c + 1) * chunk); ++k) job(k);
      });
    } else {
      for (int k = 0; k < n; ++k) job(k);
    }
  };
  // the edge check in two halves, so that host work can run while the GPU checks (a checker
  // without the split interface does all its work in the second half)
  auto check_begin = [&](size_t n) -> int {
    return merged ? ck->check_motions_begin(ck->user, eb.a.data(), eb.b.data(), n, n, eb.nd.data()) : 0;
  };
  auto check_end = [&](size_t n) -> int {
    return merged ? ck->check_motions_end(ck->user, eb.ok.data(), eb.first.data())
                  : ck->check_motions(ck->user, eb.a.data(), eb.b.data(), n, n, eb.nd.data(), eb.ok.data(), eb.first.data());
  };
  while (!solved) {
    if (req->max_iterations > 0 && out->iterations >= static_cast<uint64_t>(req->max_iterations)) break;
    if (req->time_limit_s > 0.0 && elapsed_ms() > req->time_limit_s * 1e3) break;
    ++out->iterations;
    Tree& ta = tree[cur];
    Tree& tb = tree[1 - cur];
    // ---- EXTEND and CONNECT in one launch -------------------------------------------------
    // Proposal e is the extension ta[from_idx[e]] -> target_e.  Its connect edge, the straight shot
    // from the other tree's nearest node to target_e, only matters if the extension is valid, but it
    // does not depend on the verdict: both edges go into the SAME launch (edges e and n_prop + e), and
    // the verdicts of connect edges whose extension failed are ignored.  A call costs the same up
    // to about a thousand edges, so this halves the GPU round trips of an iteration.
    from.clear();
    to.clear();
    from_idx.clear();
    auto tA = now();
    targets.resize(static_cast<size_t>(batch) * dim);
    if (!have_next) {  // (otherwise drawn and searched while the last check was in flight)
      samples.resize(static_cast<size_t>(batch) * dim);
      for (int k = 0; k < batch; ++k) sample_uniform(p, J, rng, samples.data() + static_cast<size_t>(k) * dim);
      t_sample += since(tA); tA = now();
      search_all(ta, 0, samples.data(), batch, near_a, dist_a, false);
      t_near += since(tA); tA = now();
    }
    have_next = false;
    for (int k = 0; k < batch; ++k) {
      const double d = dist_a[k];
      const int ni = near_a[k];
      if (!(d > 0.0)) continue;
      const double* smp = samples.data() + static_cast<size_t>(k) * dim;
      double* tgt = targets.data() + to.size() * dim;
      if (d > range) mvk_space_interpolate(&p, J, ta.at(ni), smp, range / d, tgt);
      else std::memcpy(tgt, smp, sizeof(double) * dim);
      from_idx.push_back(ni);
      to.push_back(tgt);
    }
    if (to.empty()) continue;
    const size_t n_prop = to.size();
    t_pack += since(tA); tA = now();
    search_all(tb, 0, targets.data(), static_cast<int>(n_prop), near_conn, dist_conn, false);
    t_near += since(tA); tA = now();
    for (int ni : from_idx) from.push_back(ta.at(ni));
    const int tb_before = tb.size();
    if (merged) {
      for (size_t e = 0; e < n_prop; ++e) {
        const double* tgt = to[e];
        from.push_back(tb.at(near_conn[e]));
        to.push_back(tgt);
      }
      eb.pack(from, to, p, J);
      t_pack += since(tA); tA = now();
      if (int rc = check_begin(2 * n_prop)) return rc;
      // while the check is in flight: the next iteration extends tb -- draw its samples and search
      // tb as it stands; the nodes this iteration adds to it are searched afterwards
      samples.resize(static_cast<size_t>(batch) * dim);
      for (int k = 0; k < batch; ++k) sample_uniform(p, J, rng, samples.data() + static_cast<size_t>(k) * dim);
      t_sample += since(tA); tA = now();
      search_all(tb, 0, samples.data(), batch, near_a, dist_a, false);
      t_near += since(tA); tA = now();
      if (int rc = check_end(2 * n_prop)) return rc;
      t_check += since(tA); tA = now();
      out->edge_checks += 2 * n_prop;
      for (int32_t c : eb.nd) out->edge_steps += static_cast<uint64_t>(c);
      conn_ok.assign(eb.ok.begin() + n_prop, eb.ok.end());
      conn_first.assign(eb.first.begin() + n_prop, eb.first.end());
      conn_nd.assign(eb.nd.begin() + n_prop, eb.nd.end());
    } else {
      // a blocking checker (the CPU one): two calls, the connect edges of the valid extensions only
      eb.pack(from, to, p, J);
      t_pack += since(tA); tA = now();
      if (int rc = check_end(n_prop)) return rc;
      t_check += since(tA); tA = now();
      out->edge_checks += n_prop;
      for (int32_t c : eb.nd) out->edge_steps += static_cast<uint64_t>(c);
      ext_ok.assign(eb.ok.begin(), eb.ok.end());
      from.clear();
      std::vector<const double*> cto;
      std::vector<size_t> which;
      for (size_t e = 0; e < n_prop; ++e)
        if (ext_ok[e]) { from.push_back(tb.at(near_conn[e])); cto.push_back(to[e]); which.push_back(e); }
      conn_ok.assign(n_prop, 0);
      conn_first.assign(n_prop, 0);
      conn_nd.assign(n_prop, 1);
      if (!which.empty()) {
        eb.pack(from, cto, p, J);
        t_pack += since(tA); tA = now();
        if (int rc = check_end(which.size())) return rc;
        t_check += since(tA); tA = now();
        out->edge_checks += which.size();
        for (int32_t c : eb.nd) out->edge_steps += static_cast<uint64_t>(c);
        for (size_t i = 0; i < which.size(); ++i) { conn_ok[which[i]] = eb.ok[i]; conn_first[which[i]] = eb.first[i]; conn_nd[which[i]] = eb.nd[i]; }
      }
      eb.ok.assign(ext_ok.begin(), ext_ok.end());
    }
    for (size_t e = 0; e < n_prop && !solved; ++e) {
      if (!eb.ok[e]) continue;  // the extension is invalid (merged: its connect edge was checked for nothing)
      const int fi = ta.add(to[e], from_idx[e]);
      if (conn_ok[e]) {
        link[cur] = fi;
        link[1 - cur] = tb.add(ta.at(fi), near_conn[e]);
        solved = true;
      } else if (conn_first[e] > 1) {
        // trapped part-way: keep the last valid interpolation step as a new node
        std::vector<double> adv(dim);
        const double t = static_cast<double>(conn_first[e] - 1) / static_cast<double>(conn_nd[e]);
        const std::vector<double> nb(tb.at(near_conn[e]), tb.at(near_conn[e]) + dim);
        mvk_space_interpolate(&p, J, nb.data(), ta.at(fi), t, adv.data());
        tb.add(adv.data(), near_conn[e]);
      }
    }
    t_tree += since(tA); tA = now();
    if (merged) {
      if (!solved && tb.size() > tb_before) search_all(tb, tb_before, samples.data(), batch, near_a, dist_a, true);
      have_next = true;
    }
    t_near += since(tA); tA = now();
    cur = 1 - cur;
  }
  out->solve_ms = elapsed_ms();
  if (trace)
    std::fprintf(stderr, "[mvk plan] %.2f ms, %llu iterations: sample %.2f, nearest %.2f, interpolate+pack %.2f, check_motions %.2f, tree %.2f\n",
                 out->solve_ms, static_cast<unsigned long long>(out->iterations), t_sample, t_near, t_pack, t_check, t_tree);
  out->tree_sizes[0] = tree[0].size();
  out->tree_sizes[1] = tree[1].size();
  if (!solved) return fail(MVK_ERR_NO_SOLUTION, "No solution found");
  // start -> meeting node (reverse of the parent walk), then meeting node -> goal
  std::vector<int> head;
  for (int i = link[0]; i >= 0; i = tree[0].parent[i]) head.push_back(i);
  std::vector<const double*> path;
  for (auto it = head.rbegin(); it != head.rend(); ++it) path.push_back(tree[0].at(*it));
  for (int i = tree[1].parent[link[1]]; i >= 0; i = tree[1].parent[i]) path.push_back(tree[1].at(i));
  out->solved = 1;
  out->n_waypoints = static_cast<int>(path.size());
  for (size_t i = 1; i < path.size(); ++i) out->path_length += mvk_space_distance(&p, J, path[i - 1], path[i]);
  if (!waypoints || cap < out->n_waypoints) return fail(MVK_ERR_CAPACITY, "way-point buffer too small (%d needed)", out->n_waypoints);
  for (size_t i = 0; i < path.size(); ++i) std::memcpy(waypoints + i * dim, path[i], sizeof(double) * dim);
  return MVK_OK;
}

namespace {
int gpu_check_states(void* user, const double* soa, size_t n, size_t ld, uint8_t* ok) {
  return mvk_is_valid_batch(static_cast<mvk_ctx*>(user), soa, n, ld, MVK_LAYOUT_OMPL, ok, nullptr, nullptr, nullptr);
}
int gpu_check_motions(void* user, const double* s1, const double* s2, size_t n, size_t ld, const int32_t* nd,
                      uint8_t* ok, int32_t* first) {
  return mvk_check_motion_batch(static_cast<mvk_ctx*>(user), s1, s2, n, ld, nd, 0, ok, first);
}
int gpu_check_motions_begin(void* user, const double* s1, const double* s2, size_t n, size_t ld, const int32_t* nd) {
  return mvk_check_motion_batch_begin(static_cast<mvk_ctx*>(user), s1, s2, n, ld, nd, 0);
}
int gpu_check_motions_end(void* user, uint8_t* ok, int32_t* first) {
  return mvk_check_motion_batch_end(static_cast<mvk_ctx*>(user), ok, first);
}
}  // namespace

extern "C" int mvk_plan_rrt_connect(mvk_ctx* ctx, const mvk_plan_request* req, double* waypoints, int cap,
                                    mvk_plan_result* out) {
  if (!ctx) return fail(MVK_ERR_INVALID_ARG, "null ctx");
  mvk_params p;
  if (int rc = mvk_get_params(ctx, &p)) return rc;
  const int J = mvk_state_dim(ctx, MVK_LAYOUT_POSE7) - 7;
  mvk_checker ck{gpu_check_states, gpu_check_motions, ctx, gpu_check_motions_begin, gpu_check_motions_end};
  return mvk_plan_rrt_connect_cb(&p, J, &ck, req, waypoints, cap, out);
}

// motion_planning.cpp:424-446
extern "C" int mvk_path_densify(const mvk_params* p, int J, const double* wp, int n, double* states, int cap) {
  if (!p || !wp || n < 1) return fail(MVK_ERR_INVALID_ARG, "mvk_path_densify: bad argument");
  const int dim = space_dim(p, J);
  int count = 0;
  auto emit = [&](const double* s) {
    if (states && count < cap) std::memcpy(states + static_cast<size_t>(count) * dim, s, sizeof(double) * dim);
    ++count;
  };
  std::vector<double> tmp(dim);
  for (int i = 1; i < n; ++i) {
    const double* s1 = wp + static_cast<size_t>(i - 1) * dim;
    const double* s2 = wp + static_cast<size_t>(i) * dim;
    emit(s1);                                                     // addState(state1)   :433
    const int nd = mvk_space_valid_segment_count(p, J, s1, s2);   // :434
    for (int j = 1; j < nd; ++j) {                                // :438-443
      mvk_space_interpolate(p, J, s1, s2, static_cast<double>(j) / static_cast<double>(nd), tmp.data());
      emit(tmp.data());
    }
    emit(s2);                                                     // addState(state2)   :445
  }
  if (states && count > cap) return fail(MVK_ERR_CAPACITY, "densified path needs %d states", count);
  return count;
}

// savePath (motion_planning.cpp:644-688).  Layout of the reference's committed logs
// (path_log/hydrus/hydrus_planning_log.txt); the J stray newlines of quirk Q6 are not emitted,
// so the file round-trips through loadPath.
extern "C" int mvk_path_log_save(const char* file, const mvk_path_log* h, const double* states) {
  if (!file || !h || (h->n_states > 0 && !states)) return fail(MVK_ERR_INVALID_ARG, "mvk_path_log_save: null argument");
  std::ofstream ofs(file);
  if (!ofs) return fail(MVK_ERR_IO, "cannot open %s for writing", file);
  const int w = 7 + h->n_joint;
  auto row = [&](const double* s) {
    for (int i = 0; i < w; ++i) ofs << " " << s[i];
    ofs << std::endl;
  };
  ofs << "start_state_: "; row(h->start);
  ofs << "goal_state_: "; row(h->goal);
  ofs << "states: " << h->n_states << std::endl;
  ofs << "planning_mode: " << h->planning_mode << std::endl;
  ofs << "planning_time: " << h->planning_time << std::endl;
  ofs << "motion_cost: " << h->motion_cost << std::endl;
  ofs << "minimum_var: " << h->minimum_var << std::endl;
  ofs << "minimum_var_state_entry: " << h->minimum_var_state_entry << std::endl;
  for (int k = 0; k < h->n_states; ++k) {
    ofs << "state" << k << ": ";
    row(states + static_cast<size_t>(k) * w);
  }
  ofs << "end" << std::endl;
  return ofs ? MVK_OK : fail(MVK_ERR_IO, "write to %s failed", file);
}

// loadPath (motion_planning.cpp:690-800).  Quirk Q7 (goal parsed from the exhausted start
// stream) is NOT reproduced: the goal line is parsed from its own line.
extern "C" int mvk_path_log_load(const char* file, int J, mvk_path_log* h, double* states, int cap) {
  if (!file || !h || J < 0 || J > MVK_MAX_JOINT) return fail(MVK_ERR_INVALID_ARG, "mvk_path_log_load: bad argument");
  std::ifstream ifs(file);
  if (!ifs) return fail(MVK_ERR_IO, "File do not exist: %s", file);
  std::memset(h, 0, sizeof(*h));
  h->n_joint = J;
  const int w = 7 + J;
  std::string line, header;
  // the reference's savePath leaves J - 1 blank lines after every state (quirk Q6, motion_planning.cpp:682-684):
  // lines without a token are skipped, so its files load as well as the ones mvk_path_log_save writes
  auto next_line = [&]() -> bool {
    while (std::getline(ifs, line))
      if (line.find_first_not_of(" \t\r") != std::string::npos) return true;
    return false;
  };
  auto read_row = [&](double* dst) -> bool {
    if (!next_line()) return false;
    std::stringstream ss(line);
    ss >> header;
    for (int i = 0; i < w; ++i)
      if (!(ss >> dst[i])) return false;
    return true;
  };
  if (!read_row(h->start) || !read_row(h->goal)) return fail(MVK_ERR_IO, "%s: malformed start/goal line", file);
  auto read_kv = [&](auto& dst) -> bool {
    if (!next_line()) return false;
    std::stringstream ss(line);
    ss >> header >> dst;
    return !ss.fail();
  };
  if (!read_kv(h->n_states) || !read_kv(h->planning_mode) || !read_kv(h->planning_time) || !read_kv(h->motion_cost) ||
      !read_kv(h->minimum_var) || !read_kv(h->minimum_var_state_entry))
    return fail(MVK_ERR_IO, "%s: malformed header", file);
  std::vector<double> tmp(w);
  for (int k = 0; k < h->n_states; ++k) {
    if (!read_row(tmp.data())) return fail(MVK_ERR_IO, "%s: malformed state line %d", file, k);
    if (states && k < cap) std::memcpy(states + static_cast<size_t>(k) * w, tmp.data(), sizeof(double) * w);
  }
  return h->n_states;
}
