// mvk_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the batched validity engine.
//
// One lane = one state (a sample of a SoA batch, or one interpolation step of
// an edge).  The path is fp64 VALU bound: a state costs
// kiloflops of arithmetic against ~100 B of HBM traffic.  Shape of the flag path
// (the headline): screen_kernel = model update + statics in fp64, then an fp32
// screen of the few boxes the broadphase grid lists near each chain segment,
// shared out over the wave through an LDS ring; the states it cannot decide are
// handed (state index + a compact record) to exact_kernel, which runs the fp64
// narrowphase one (primitive, box) pair per lane.  Wave-uniform data (model
// constants) comes through scalar loads.  The distance kernels (full_kernel:
// exact minimum distance, margins, statistics; avoid_kernel: nearest box and
// points per primitive) take per-lane candidate lists from distance grids keyed
// by primitive centres, serve the lanes a grid cannot answer with a
// wave-cooperative sweep over all boxes, and compute the expensive exact
// box-box distances as (pair, edge) items over the whole wave.  DESIGN.md 4-5.
//
// What one state evaluation computes (V-spec), replacing
//   MotionPlanning::isStateValid            sampling_based_method/src/motion_planning.cpp:76-164
//   MultilinkState::setStatesFromRoot       aerial_motion_planning_msgs/src/multilink_state.cpp:98-137
//   RobotModel::updateRobotModel/stabilityCheck   (third-party; call sites multilink_state.cpp:113, motion_planning.cpp:150)
//   PlanningScene::checkCollision           (third-party; call site motion_planning.cpp:156-161)
//   DiscreteMotionValidator::checkMotion    (third-party; bound at motion_planning.cpp:238,249)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "mvk_internal.h"

namespace mvk {

// ------------------------------------------------------------------------
// scratch of the no-allocation entry point (mvk_is_valid_mask_noalloc_dev)
// ------------------------------------------------------------------------
// Callers that cannot allocate (code under HIP graph capture, tight serving loops) get the flag path with its
// hand-off queues, records and counters in this file-scope array instead of ctx-owned hipMalloc memory.  Each
// GPU the library runs on holds its own copy (22.1 MiB) at its own address until the process ends.  In host code
// the NAME `scratch` is only a handle for the runtime, not a device address: the address of the current device's
// copy comes from hipGetSymbolAddress, once per device, and only that pointer is ever handed to kernels and
// memsets.
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

int static_scratch_for_current_device(int device, void** ptr) {
  constexpr int kMaxDevices = 64;
  static void* resolved[kMaxDevices] = {};
  static std::mutex guard;
  *ptr = nullptr;
  if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lk(guard);
  if (!resolved[device]) {
    int current = -1;
    hipError_t e = hipGetDevice(&current);
    if (e != hipSuccess) return e;
    if (current != device) return hipErrorInvalidDevice;  // the caller makes `device` current first
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    resolved[device] = p;
  }
  *ptr = resolved[device];
  return hipSuccess;
}

namespace {

// ------------------------------------------------------------------------
// scalar helpers
// ------------------------------------------------------------------------
#define DEV __device__ __forceinline__

// Diagnostic builds only (profiles/diag.py, -DMVK_PROF): event counters, one atomic per wave.
#ifdef MVK_PROFS  // stage clocks of the screen kernel instead of the exact kernel's: same records, same host side
#define MVK_PROFT
#endif
#if defined(MVK_PROF) || defined(MVK_PROFT)
__device__ unsigned long long g_prof[48];
#endif
#ifdef MVK_PROF
#define PROF_LANES(idx, pred) do { const unsigned long long b__ = __ballot(pred); if ((threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) atomicAdd(&g_prof[idx], (unsigned long long)__popcll(b__)); } while (0)
#define PROF_WAVE(idx) do { if ((threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) atomicAdd(&g_prof[idx], 1ull); } while (0)
#define PROF_SUM(idx, val) do { unsigned long long v__ = (unsigned long long)(val); for (int o__ = 32; o__; o__ >>= 1) v__ += __shfl_xor(v__, o__); if ((threadIdx.x & 63) == 0) atomicAdd(&g_prof[idx], v__); } while (0)
// stage clocks: accumulated per wave in registers (PROF_ACC_DECL in the function), one atomic per stage at PROF_ACC_FLUSH
#define PROF_ACC_DECL unsigned long long prof_acc__[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROF_T0(var) const unsigned long long var = wall_clock64()
#define PROF_CLK(slot, var) do { prof_acc__[slot] += wall_clock64() - (var); } while (0)
#define PROF_ACC_FLUSH(base) do { if ((threadIdx.x & 63) == 0) { for (int i__ = 0; i__ < 8; ++i__) if (prof_acc__[i__]) atomicAdd(&g_prof[(base) + i__], prof_acc__[i__]); } } while (0)
#else
#define PROF_ACC_DECL do { } while (0)
#define PROF_T0(var) do { } while (0)
#define PROF_CLK(slot, var) do { } while (0)
#define PROF_ACC_FLUSH(base) do { } while (0)
#define PROF_LANES(idx, pred) do { } while (0)
#define PROF_SUM(idx, val) do { } while (0)
#define PROF_WAVE(idx) do { } while (0)
#endif
#ifdef MVK_PROFT
// stage clocks of the exact kernel (100 MHz wall clock): every wave writes its own record (no
// atomics: same-address atomics from thousands of waves would be the thing measured); the host
// reduces them in mvk_debug_prof
constexpr int kProfWaves = 1 << 15;
__device__ unsigned long long g_rec[kProfWaves * 8];
#define PROF_T(var) const unsigned long long var = wall_clock64()
#define PROF_STAGE_DECL unsigned long long prof_d[4] = {0, 0, 0, 0}
#define PROF_STAGE(slot, t0, t1) prof_d[slot] += (t1) - (t0)
#define PROF_STAGE_FLUSH do { const unsigned w__ = ((blockIdx.y * gridDim.x + blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6)); \
    if ((threadIdx.x & 63) == 0 && w__ < kProfWaves) { unsigned long long* r__ = g_rec + 8 * w__; r__[0] = 1; r__[1] = t_entry; r__[2] = wall_clock64(); \
      for (int i__ = 0; i__ < 4; ++i__) r__[3 + i__] = prof_d[i__]; } } while (0)
#else
#define PROF_T(var) do { } while (0)
#define PROF_STAGE_DECL do { } while (0)
#define PROF_STAGE(slot, t0, t1) do { } while (0)
#define PROF_STAGE_FLUSH do { } while (0)
#endif

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
DEV void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

constexpr double kPi = 3.14159265358979323846;

// sin/cos for bounded arguments (joint angles, yaw, slerp arcs): two-constant
// Cody-Waite reduction by pi/2 with FMAs, then the fdlibm minimax kernels on
// |r| <= pi/4.  ~1 ulp for |x| < 1e5; avoids the Payne-Hanek slow path (and its
// scratch usage) of the library sincos.
DEV void sincos_cw(double x, double& s, double& c) {
  const double fn = rint(x * 6.36619772367581382433e-01);
  double r = fma(-fn, 1.57079632679489655800e+00, x);
  r = fma(-fn, 6.12323399573676603587e-17, r);
  const int q = static_cast<int>(fn);
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  const double sr = fma(r * z, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
  const double a = (q & 1) ? cr : sr;
  const double b = (q & 1) ? sr : cr;
  s = (q & 2) ? -a : a;
  c = ((q + 1) & 2) ? -b : b;
}

// 1/sqrt(x) for x > 0: hardware estimate + two Newton steps (relative error ~1e-16), without the
// division and the square root of the IEEE sequence.  x = 0 / inf / NaN give non-finite results,
// callers select around them.
DEV double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = fma(y, fma(-hx * y, y, 0.5), y);
  y = fma(y, fma(-hx * y, y, 0.5), y);
  return y;
}

// reciprocal by two Newton steps from v_rcp_f64: full precision up to rounding, a third of the IEEE divide.
// x == 0 (or inf) gives NaN, which the callers' range tests discard like the +-inf of a true division.
DEV double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(y, fma(-x, y, 1.0), y);
  y = fma(y, fma(-x, y, 1.0), y);
  return y;
}

// one Newton step: relative error ~1e-14, for scale factors of distances compared at 1e-5 / 1e-6
DEV double rsqrt_nr1(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  return fma(y, fma(-0.5 * x * y, y, 0.5), y);
}

DEV double excess(double x, double h) {  // signed overshoot outside [-h, h]
  return copysign(fmax(fabs(x) - h, 0.0), x);
}
DEV double excess_abs(double x, double h) { return fmax(fabs(x) - h, 0.0); }

// tf::Matrix3x3::setRotation.  Products and sums as written (no fused multiply-add): which of the two products of
// `xy - wz` the compiler fuses depends on the surrounding kernel, and every kernel must give a state the same matrix.
DEV void quat_to_mat(double x, double y, double z, double w, double R[9]) {
#pragma clang fp contract(off)
  const double d = x * x + y * y + z * z + w * w;
  double rd = __builtin_amdgcn_rcp(d);  // 1/d by Newton iteration (two steps: full precision) instead of a division
  rd = rd * fma(-d, rd, 2.0);
  rd = rd * fma(-d, rd, 2.0);
  const double s = 2.0 * rd;
  const double xs = x * s, ys = y * s, zs = z * s;
  const double wx = w * xs, wy = w * ys, wz = w * zs;
  const double xx = x * xs, xy = x * ys, xz = x * zs;
  const double yy = y * ys, yz = y * zs, zz = z * zs;
  R[0] = 1.0 - (yy + zz); R[1] = xy - wz;         R[2] = xz + wy;
  R[3] = xy + wz;         R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
  R[6] = xz - wy;         R[7] = yz + wx;         R[8] = 1.0 - (xx + yy);
}

// R <- R * Rot(AXIS, angle) : mixes columns a = AXIS+1, b = AXIS+2 (mod 3)
template <int AXIS>
DEV void rot_cols(double R[9], double c, double s) {
  constexpr int a = (AXIS + 1) % 3, b = (AXIS + 2) % 3;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double ua = R[3 * r + a], ub = R[3 * r + b];
    R[3 * r + a] = fma(c, ua, s * ub);
    R[3 * r + b] = fma(c, ub, -s * ua);
  }
}
DEV void rot_cols_dyn(double R[9], int axis, double c, double s) {  // axis is wave-uniform
  if (axis == 0) rot_cols<0>(R, c, s);
  else if (axis == 1) rot_cols<1>(R, c, s);
  else rot_cols<2>(R, c, s);
}
// y = p + R * t
DEV void xform(const double R[9], const double p[3], double tx, double ty, double tz, double y[3]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) y[r] = fma(R[3 * r], tx, fma(R[3 * r + 1], ty, fma(R[3 * r + 2], tz, p[r])));
}
// C = A * B (row-major)
DEV void mat_mul(const double A[9], const double B[9], double C[9]) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      C[3 * r + c] = fma(A[3 * r], B[c], fma(A[3 * r + 1], B[3 + c], A[3 * r + 2] * B[6 + c]));
}

// ------------------------------------------------------------------------
// narrowphase (all in the obstacle's frame; exact, closed form)
// ------------------------------------------------------------------------
DEV double seg_grad(const double p0[3], const double d[3], const double h[3], double t) {
  double g = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) g = fma(d[k], excess(fma(t, d[k], p0[k]), h[k]), g);
  return g;  // half the derivative of dist^2 along the segment
}

// squared distance between the segment p0 + t d (t in [0,1]) and the solid box
// [-h,h]^3.  dist^2(t) is convex piecewise quadratic, so its derivative is
// nondecreasing piecewise linear with breakpoints where a coordinate crosses
// +-h: bracket the zero between the two neighbouring breakpoints, interpolate.
DEV double seg_box_dist2(const double p0[3], const double d[3], const double h[3]) {
  const double g0 = seg_grad(p0, d, h, 0.0), g1 = seg_grad(p0, d, h, 1.0);
  double tlo = 0.0, glo = g0, thi = 1.0, ghi = g1;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double inv = rcp_nr(d[k]);  // NaN for d[k] == 0: no breakpoint on that axis, as with 1 / 0
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const double t = ((sg ? -h[k] : h[k]) - p0[k]) * inv;
      if (t > 0.0 && t < 1.0) {  // false for NaN / inf (d[k] == 0)
        const double g = seg_grad(p0, d, h, t);
        if (g < 0.0) {
          if (t > tlo) { tlo = t; glo = g; }
        } else if (t < thi) { thi = t; ghi = g; }
      }
    }
  }
  double ts = (ghi > glo) ? fma(thi - tlo, -glo / (ghi - glo), tlo) : tlo;
  ts = (g1 <= 0.0) ? 1.0 : ts;
  ts = (g0 >= 0.0) ? 0.0 : ts;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double e = excess_abs(fma(ts, d[k], p0[k]), h[k]);
    s = fma(e, e, s);
  }
  return s;
}

// Separating-axis sweep for box A (centre t, axes = columns of R, half extents a;
// all expressed in box B's frame) against B = [-b,b]^3.
// METRIC: gaps of the 9 edge-edge axes are divided by the axis length so the
// result is the largest metric gap (> 0 separated, <= 0: -penetration depth).
// Otherwise only its sign is meaningful.
// `cutoff` (METRIC): once the six face axes alone reach it the nine edge axes are skipped -- the caller only asks
// whether the gap undercuts cutoff.
template <bool METRIC>
DEV double box_box_sat(const double t[3], const double R[9], const double a[3], const double b[3], double cutoff = INFINITY) {
  double Q[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Q[i] = fabs(R[i]);
  double best = -INFINITY;
#pragma unroll
  for (int i = 0; i < 3; ++i) {  // B's face normals (frame axes)
    const double g = fabs(t[i]) - (b[i] + fma(a[0], Q[3 * i], fma(a[1], Q[3 * i + 1], a[2] * Q[3 * i + 2])));
    best = fmax(best, g);
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {  // A's face normals (columns of R)
    const double proj = fma(t[0], R[j], fma(t[1], R[3 + j], t[2] * R[6 + j]));
    const double g = fabs(proj) - (a[j] + fma(b[0], Q[j], fma(b[1], Q[3 + j], b[2] * Q[6 + j])));
    best = fmax(best, g);
  }
  if (METRIC && best >= cutoff) return best;
#pragma unroll
  for (int i = 0; i < 3; ++i)  // axes  B_i x A_j
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      constexpr int dummy = 0;
      (void)dummy;
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const double len2 = fma(-R[3 * i + j], R[3 * i + j], 1.0);
      const double rb = fma(b[i1], Q[3 * i2 + j], b[i2] * Q[3 * i1 + j]);
      const double ra = fma(a[j1], Q[3 * i + j2], a[j2] * Q[3 * i + j1]);
      double g = fabs(fma(t[i2], R[3 * i1 + j], -t[i1] * R[3 * i2 + j])) - (ra + rb);
      if (METRIC) g = g * rsqrt_nr(len2);  // 2 Newton steps: full precision up to rounding, a quarter of sqrt + divide
      g = (len2 < 1e-12) ? -INFINITY : g;  // parallel edges: axis degenerate, skip
      best = fmax(best, g);
    }
  return best;
}

// squared distance from edge e (0..11: axis e >> 2, signs from bits 0 and 1) of box A (centre c, axes = columns of
// R, half extents a; in B's frame) to solid B = [-b,b]^3.  Columns are selected by comparisons on e, no dynamic
// register indexing, whether e is wave-uniform or per lane.
DEV double box_edge_to_box2(int e, const double c[3], const double R[9], const double a[3], const double b[3]) {
  const int ax = e >> 2;
  const double su = (e & 1) ? 1.0 : -1.0, sv = (e & 2) ? 1.0 : -1.0;
  double p0[3], d[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double cx = R[3 * k], cy = R[3 * k + 1], cz = R[3 * k + 2];
    const double Aax = ax == 0 ? cx : (ax == 1 ? cy : cz);
    const double Au = ax == 0 ? cy : (ax == 1 ? cz : cx);
    const double Av = ax == 0 ? cz : (ax == 1 ? cx : cy);
    const double hax = ax == 0 ? a[0] : (ax == 1 ? a[1] : a[2]);
    const double hu = ax == 0 ? a[1] : (ax == 1 ? a[2] : a[0]);
    const double hv = ax == 0 ? a[2] : (ax == 1 ? a[0] : a[1]);
    const double mid = fma(su * hu, Au, fma(sv * hv, Av, c[k]));
    p0[k] = fma(-hax, Aax, mid);
    d[k] = 2.0 * hax * Aax;
  }
  return seg_box_dist2(p0, d, b);
}

// min over the 12 edges of box A (in B's frame) of the squared distance to solid B
DEV double box_edges_to_box2(const double c[3], const double R[9], const double a[3], const double b[3]) {
  double best = INFINITY;
#pragma unroll 1
  for (int e = 0; e < 12; ++e)  // rolled on purpose: 12 copies of seg_box_dist2 would bloat the kernel
    best = fmin(best, box_edge_to_box2(e, c, R, a, b));
  return best;
}


// ---- the same narrowphase in fp32 (the fp32 variant of the distance outputs, mvk_eval_batch_f32_dev): one rounding
// of 2^-24 relative per operation on box-frame coordinates of a few metres, i.e. errors of ~1e-6 m; the tolerance the
// variant is reported with is stated in include/mvk.h and measured in tests/test_gpu_f32_dist.py ----
DEV float excess_f(float x, float h) { return x > h ? x - h : (x < -h ? x + h : 0.0f); }
DEV float excess_abs_f(float x, float h) { return fmaxf(fabsf(x) - h, 0.0f); }
DEV float seg_grad_f(const float p0[3], const float d[3], const float h[3], float t) {
  float g = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) g = fmaf(d[k], excess_f(fmaf(t, d[k], p0[k]), h[k]), g);
  return g;
}
DEV float seg_box_dist2_f(const float p0[3], const float d[3], const float h[3]) {
  const float g0 = seg_grad_f(p0, d, h, 0.0f), g1 = seg_grad_f(p0, d, h, 1.0f);
  float tlo = 0.0f, glo = g0, thi = 1.0f, ghi = g1;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float inv = __frcp_rn(d[k]);  // inf for d[k] == 0: no breakpoint on that axis
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const float t = ((sg ? -h[k] : h[k]) - p0[k]) * inv;
      if (t > 0.0f && t < 1.0f) {  // false for NaN / inf
        const float g = seg_grad_f(p0, d, h, t);
        if (g < 0.0f) {
          if (t > tlo) { tlo = t; glo = g; }
        } else if (t < thi) { thi = t; ghi = g; }
      }
    }
  }
  float ts = (ghi > glo) ? fmaf(thi - tlo, -glo / (ghi - glo), tlo) : tlo;
  ts = (g1 <= 0.0f) ? 1.0f : ts;
  ts = (g0 >= 0.0f) ? 0.0f : ts;
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float e = excess_abs_f(fmaf(ts, d[k], p0[k]), h[k]);
    s = fmaf(e, e, s);
  }
  return s;
}
// largest metric separating-axis gap (box_box_sat<true>), the nine edge axes skipped once the face axes reach cutoff
// recheck: the pair is in contact by the axes taken and an ill-conditioned edge axis was left out -- the penetration
// depth may hang on it, the caller repeats the pair in fp64
DEV float box_box_sat_f(const float t[3], const float R[9], const float a[3], const float b[3], float cutoff, bool& recheck) {
  float Q[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Q[i] = fabsf(R[i]);
  float best = -INFINITY;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float g = fabsf(t[i]) - (b[i] + fmaf(a[0], Q[3 * i], fmaf(a[1], Q[3 * i + 1], a[2] * Q[3 * i + 2])));
    best = fmaxf(best, g);
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float proj = fmaf(t[0], R[j], fmaf(t[1], R[3 + j], t[2] * R[6 + j]));
    const float g = fabsf(proj) - (a[j] + fmaf(b[0], Q[j], fmaf(b[1], Q[3 + j], b[2] * Q[6 + j])));
    best = fmaxf(best, g);
  }
  recheck = false;
  if (best >= cutoff) return best;
  bool left_out = false;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      // |b_i x a_j|^2 from the two other components of a_j (1 - R_ij^2 would cancel)
      const float len2 = fmaf(R[3 * i1 + j], R[3 * i1 + j], R[3 * i2 + j] * R[3 * i2 + j]);
      const float rb = fmaf(b[i1], Q[3 * i2 + j], b[i2] * Q[3 * i1 + j]);
      const float ra = fmaf(a[j1], Q[3 * i + j2], a[j2] * Q[3 * i + j1]);
      float g = fabsf(fmaf(t[i2], R[3 * i1 + j], -t[i1] * R[3 * i2 + j])) - (ra + rb);
      g = g * __frsqrt_rn(len2);
      // (nearly) parallel edges: 1 / |axis| amplifies the roundings of the numerator (1e-7 m) -- beyond ten times the
      // axis is left out, which only ever lowers the gap (a lower bound stays one) and sends the pair to the edge distances
      left_out = left_out || (len2 < 1e-2f && len2 >= 1e-12f);
      g = (len2 < 1e-2f) ? -INFINITY : g;
      best = fmaxf(best, g);
    }
  recheck = left_out && best <= 0.0f;
  return best;
}
DEV float box_edge_to_box2_f(int e, const float c[3], const float R[9], const float a[3], const float b[3]) {
  const int ax = e >> 2;
  const float su = (e & 1) ? 1.0f : -1.0f, sv = (e & 2) ? 1.0f : -1.0f;
  float p0[3], d[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float cx = R[3 * k], cy = R[3 * k + 1], cz = R[3 * k + 2];
    const float Aax = ax == 0 ? cx : (ax == 1 ? cy : cz);
    const float Au = ax == 0 ? cy : (ax == 1 ? cz : cx);
    const float Av = ax == 0 ? cz : (ax == 1 ? cx : cy);
    const float hax = ax == 0 ? a[0] : (ax == 1 ? a[1] : a[2]);
    const float hu = ax == 0 ? a[1] : (ax == 1 ? a[2] : a[0]);
    const float hv = ax == 0 ? a[2] : (ax == 1 ? a[0] : a[1]);
    const float mid = fmaf(su * hu, Au, fmaf(sv * hv, Av, c[k]));
    p0[k] = fmaf(-hax, Aax, mid);
    d[k] = 2.0f * hax * Aax;
  }
  return seg_box_dist2_f(p0, d, b);
}

// ---- closest points (collision-avoidance rows) -----------------
// as seg_box_dist2, also returning the point on the segment (y) and on the box (z), box frame
DEV double seg_box_closest2(const double p0[3], const double d[3], const double h[3], double y[3], double z[3]) {
  const double g0 = seg_grad(p0, d, h, 0.0), g1 = seg_grad(p0, d, h, 1.0);
  double tlo = 0.0, glo = g0, thi = 1.0, ghi = g1;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double inv = rcp_nr(d[k]);  // NaN for d[k] == 0: no breakpoint on that axis, as with 1 / 0
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const double t = ((sg ? -h[k] : h[k]) - p0[k]) * inv;
      if (t > 0.0 && t < 1.0) {  // false for NaN / inf (d[k] == 0)
        const double g = seg_grad(p0, d, h, t);
        if (g < 0.0) {
          if (t > tlo) { tlo = t; glo = g; }
        } else if (t < thi) { thi = t; ghi = g; }
      }
    }
  }
  double ts = (ghi > glo) ? fma(thi - tlo, -glo / (ghi - glo), tlo) : tlo;
  ts = (g1 <= 0.0) ? 1.0 : ts;
  ts = (g0 >= 0.0) ? 0.0 : ts;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    y[k] = fma(ts, d[k], p0[k]);
    z[k] = fmin(fmax(y[k], -h[k]), h[k]);
    const double e = y[k] - z[k];
    s = fma(e, e, s);
  }
  return s;
}

// closest pair between edge e of box A (in B's frame) and solid B: squared distance, y on the edge, z on B.
// Edge order of the oracle: axis e >> 2, then the sign of u (bit 1), then the sign of v (bit 0).
DEV double box_edge_to_box_cp2(int e, const double c[3], const double R[9], const double a[3], const double b[3], double y[3], double z[3]) {
  const int ax = e >> 2;
  const double su = (e & 2) ? 1.0 : -1.0, sv = (e & 1) ? 1.0 : -1.0;
  double p0[3], d[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double cx = R[3 * k], cy = R[3 * k + 1], cz = R[3 * k + 2];
    const double Aax = ax == 0 ? cx : (ax == 1 ? cy : cz);
    const double Au = ax == 0 ? cy : (ax == 1 ? cz : cx);
    const double Av = ax == 0 ? cz : (ax == 1 ? cx : cy);
    const double hax = ax == 0 ? a[0] : (ax == 1 ? a[1] : a[2]);
    const double hu = ax == 0 ? a[1] : (ax == 1 ? a[2] : a[0]);
    const double hv = ax == 0 ? a[2] : (ax == 1 ? a[0] : a[1]);
    const double mid = fma(su * hu, Au, fma(sv * hv, Av, c[k]));
    p0[k] = fma(-hax, Aax, mid);
    d[k] = 2.0 * hax * Aax;
  }
  return seg_box_closest2(p0, d, b, y, z);
}

// closest pair between the 12 edges of box A (in B's frame) and solid B: squared distance,
// pa on A's edge, pb on B (both in B's frame)
DEV double box_edges_to_box_cp2(const double c[3], const double R[9], const double a[3], const double b[3],
                                double pa[3], double pb[3]) {
  double best = INFINITY;
#pragma unroll 1
  for (int e = 0; e < 12; ++e) {
    double y[3], z[3];
    const double s2 = box_edge_to_box_cp2(e, c, R, a, b, y, z);
    if (s2 < best) {
      best = s2;
#pragma unroll
      for (int k = 0; k < 3; ++k) { pa[k] = y[k]; pb[k] = z[k]; }
    }
  }
  return best;
}

// ------------------------------------------------------------------------
// per-state evaluation
// ------------------------------------------------------------------------
struct StateOut {
  bool valid, band, stable, collision;
  double min_dist, margin, fc_t_min, fc_rp_min, rp_margin, lam_min, lam_max;
};

template <int ROBOT, int NL>
struct Shape {
  static constexpr bool kGimbal = ROBOT == MVK_ROBOT_DRAGON;
  static constexpr int kSeg = kGimbal ? 2 * NL - 1 : NL;
  static constexpr int kJoint = kGimbal ? 2 * (NL - 1) : NL - 1;
  static constexpr int kDof = kGimbal ? 3 : 4;
  // doubles of a state record of the flag path: root position and quaternion, a pad, (cos, sin) of every joint
  static constexpr int kRecPad = 8 + 2 * kJoint;
  // joint axis of segment k (compile time): HYDRUS yaw chain; DRAGON pitch (odd) / yaw (even)
  static constexpr int axis(int k) { return kGimbal ? ((k & 1) ? 1 : 2) : 2; }
  static constexpr int joint(int k) { return k - 1; }
  static constexpr int rotor_seg(int i) { return kGimbal ? 2 * i : i; }
  // waves per SIMD the screen kernel is compiled for (caps its VGPRs at 512 / waves): the walk-1
  // arithmetic offers the scheduler more independent chains than pay off in registers
#ifndef MVK_SW_H4
#define MVK_SW_H4 3
#endif
#ifndef MVK_SW_H8
#define MVK_SW_H8 3
#endif
#ifndef MVK_SW_D
#define MVK_SW_D 3
#endif
  static constexpr int kScreenWaves = kGimbal ? MVK_SW_D : NL <= 4 ? MVK_SW_H4 : NL >= 8 ? MVK_SW_H8 : 3;  // (NL = 8: see the LDS budget in screen_kernel)
};

#ifndef MVK_LIST_CAP
#define MVK_LIST_CAP 32
#endif
constexpr int kListCap = MVK_LIST_CAP;     // full path: per-lane narrowphase candidates per primitive
constexpr int kItemCap = 4 * kBlock;  // flag path: fp64 (state, box) pairs per block and primitive slot

// LDS views.  The pointers carry the LDS address space explicitly so that every access is a
// ds_read / ds_write (a generic pointer would go through flat instructions and aperture checks).
typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(3))) unsigned short lds_ushort;
typedef __attribute__((address_space(3))) unsigned lds_uint;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) float lds_float;
struct Smem {
  // [2*J][kBlock] cos/sin of the planned joints, lane-contiguous
  lds_double* cs;
  // [list length][kBlock] candidate obstacle ids
  lds_ushort* list;
  // full path: this wave's batch of box pairs awaiting their exact distance, [kExactBatch][kExactItem] and the
  // running minimum of each pair's 24 edge-to-box squared distances (bits of a non-negative double), [kExactBatch]
  lds_double* ex_item;
  lds_u64* ex_res;
  // avoidance rows: the squared distance of every (pair, edge) item of the batch, [24 * kCpBatch], and the lowest
  // edge index among each pair's minima, [kCpBatch]
  lds_double* ex_s2;
  lds_uint* ex_win;
  // full path: per primitive (first kUpperSlots of the chain order) the first walk's upper bound, [kUpperSlots][kBlock]
  lds_float* upper;
  // full path, distance kernel: the joints' cos/sin rows left in HBM by the model kernel, [2*J][gcs_ld] (row pitch in
  // doubles), this lane's column gcs_col; nullptr: they are in `cs` (LDS)
  const double* gcs;
  uint64_t gcs_ld, gcs_col;
  // full path, distance kernel: this wave's broadphase ring [kDRingWords][kRing] and its lanes' candidate counts [64]
  lds_uint* dring;
  lds_uint* dcnt;
  // full path, flat drain: per lane of this wave the running minimum as an ordered key [64], and the smallest pending
  // separating-axis gap of the round [64]
  lds_u64* mres;
  lds_u64* gmin;
};
#ifndef MVK_UPPER_SLOTS
#define MVK_UPPER_SLOTS 16
#endif
#ifndef MVK_EXACT_BATCH
#define MVK_EXACT_BATCH 16
#endif
constexpr int kUpperSlots = MVK_UPPER_SLOTS;
constexpr int kExactBatch = MVK_EXACT_BATCH;  // box pairs whose exact distance one wave computes together
constexpr int kCpBatch = 8;      // the same for the avoidance rows (closest points as well)
constexpr int kExactItem = 16;   // doubles per pair: relative rotation 9, centre 3, box half extents 3, pad

// gimbal (roll, pitch) that turn the thrust axis onto the CoG z axis
// (Dragon::HydrusLikeRobotModel).  zl = CoG z axis in the link frame.
DEV void gimbal_cs(double zlx, double zly, double zlz, double& cr, double& sr, double& cp, double& sp) {
  const double rho = sqrt(fma(zly, zly, zlz * zlz));
  const bool deg = !(rho > 0.0);
  const double ir = 1.0 / rho;
  cr = deg ? 1.0 : zlz * ir;   // roll  = atan2(-zly, zlz)
  sr = deg ? 0.0 : -zly * ir;
  const double w = fma(-zly, sr, zlz * cr);  // = rho
  const double n = 1.0 / sqrt(fma(zlx, zlx, w * w));
  cp = w * n;                  // pitch = atan2(zlx, w)
  sp = zlx * n;
}

// the same with Newton reciprocal square roots instead of sqrt and two divisions (collision walks)
DEV void gimbal_cs_fast(double zlx, double zly, double zlz, double& cr, double& sr, double& cp, double& sp) {
  const double rho2 = fma(zly, zly, zlz * zlz);
  const bool deg = !(rho2 > 0.0);
  const double ir = rsqrt_nr(rho2);
  cr = deg ? 1.0 : zlz * ir;
  sr = deg ? 0.0 : -zly * ir;
  const double w = fma(-zly, sr, zlz * cr);
  const double n = rsqrt_nr(fma(zlx, zlx, w * w));
  cp = w * n;
  sp = zlx * n;
}

// ------------------------------------------------------------------------
// walk 1: FK, centre of gravity, rotor geometry, statics, stability margin
// ------------------------------------------------------------------------
// Replaces MultilinkState::setStatesFromRoot -> RobotModel::updateRobotModel
// (multilink_state.cpp:98-137) and stabilityCheck() (motion_planning.cpp:150).
// Leaves cos/sin of the planned joints in LDS slot `tid` for the collision walks.
template <int ROBOT, int NL, bool FULL>
DEV void model_update(const EvalArgs& A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
                      const double st[7 + Shape<ROBOT, NL>::kJoint], const Smem& sm, uint64_t gi, StateOut& out,
                      double Rw[9], bool& finite, double* thrust_out /*[NL] or nullptr*/) {
  using S = Shape<ROBOT, NL>;
  constexpr int NS = S::kSeg, NJ = S::kJoint, D = S::kDof;
  const int tid = threadIdx.x;

  out.valid = false; out.band = false; out.stable = false; out.collision = false;
  out.min_dist = INFINITY; out.margin = INFINITY;
  out.fc_t_min = NAN; out.fc_rp_min = NAN; out.rp_margin = NAN; out.lam_min = NAN; out.lam_max = NAN;

  quat_to_mat(st[3], st[4], st[5], st[6], Rw);
  finite = true;
#pragma unroll
  for (int k = 0; k < 7 + NJ; ++k) finite = finite && isfinite(st[k]);
#pragma unroll
  for (int k = 0; k < 9; ++k) finite = finite && isfinite(Rw[k]);

  // joint cos/sin -> LDS (read back by the rolled collision walks)
  double jc[NJ], js[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    sincos_cw(st[7 + j], js[j], jc[j]);
    sm.cs[(2 * j) * kBlock + tid] = jc[j];
    sm.cs[(2 * j + 1) * kBlock + tid] = js[j];
  }

  double R[9], p[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = Rw[k];
  p[0] = st[0]; p[1] = st[1]; p[2] = st[2];
  double cog[3] = {0.0, 0.0, 0.0}, mass = 0.0;
  double P[NL][3], U[NL][3];
  double E[S::kGimbal ? 2 * NL : 1][3];
  static_for<0, NS>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k > 0) {
      double pn[3];
      xform(R, p, M->seg_t[k][0], M->seg_t[k][1], M->seg_t[k][2], pn);
      p[0] = pn[0]; p[1] = pn[1]; p[2] = pn[2];
      rot_cols<S::axis(k)>(R, jc[S::joint(k)], js[S::joint(k)]);
    }
    const double mk = M->seg_mass[k];
    double cw[3];
    xform(R, p, M->seg_com[k][0], M->seg_com[k][1], M->seg_com[k][2], cw);
#pragma unroll
    for (int c = 0; c < 3; ++c) cog[c] = fma(mk, cw[c], cog[c]);
    mass += mk;
    const bool is_rotor_seg = S::kGimbal ? ((k & 1) == 0) : true;
    if (is_rotor_seg) {
      const int i = S::kGimbal ? k / 2 : k;
      xform(R, p, M->rotor_pos[i][0], M->rotor_pos[i][1], M->rotor_pos[i][2], P[i]);
      if (S::kGimbal) {
        double cr, sr, cp, sp;
        if (FULL) gimbal_cs(R[6], R[7], R[8], cr, sr, cp, sp);
        else gimbal_cs_fast(R[6], R[7], R[8], cr, sr, cp, sp);
        if (A.gimbal) {  // getGimbalProcessedJoint (multilink_state.cpp:136)
          A.gimbal[(2 * i) * A.ld + gi] = atan2(-R[7], R[8]);
          A.gimbal[(2 * i + 1) * A.ld + gi] = atan2(R[6], fma(-R[7], sr, R[8] * cr));
        }
        // thrust axis = R * Rx(roll) * Ry(pitch) * z ;  EDFs at +-y of the pitch frame
        const double ux = sp, uy = -sr * cp, uz = cr * cp;
        const double ey = cr, ez = sr;  // Rx(roll) * y = (0, cr, sr)
        const double hz = M->rotor_roll_z;  // thrust-frame origin: up the roll frame's z axis, Rx(roll) * z = (0, -sr, cr)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          U[i][r] = fma(R[3 * r], ux, fma(R[3 * r + 1], uy, R[3 * r + 2] * uz));
          P[i][r] = fma(hz, fma(R[3 * r + 2], cr, -R[3 * r + 1] * sr), P[i][r]);
          const double yw = fma(R[3 * r + 1], ey, R[3 * r + 2] * ez) * M->edf_offset_y;
          E[2 * i][r] = P[i][r] + yw;
          E[2 * i + 1][r] = P[i][r] - yw;
        }
      } else {
        U[i][0] = R[2]; U[i][1] = R[5]; U[i][2] = R[8];
      }
    }
  });
  const double inv_mass = X->inv_mass;  // = 1 / mass, from the host
#pragma unroll
  for (int c = 0; c < 3; ++c) cog[c] *= inv_mass;
  if (A.cog) {
#pragma unroll
    for (int c = 0; c < 3; ++c) A.cog[c * A.ld + gi] = cog[c];
  }
  // CoG frame is world aligned (cog_desire = q_root * R(root->baselink), baselink == root)
#pragma unroll
  for (int i = 0; i < NL; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) P[i][c] -= cog[c];

  // ---- statics ------------------------------------------------------------
  const double fmaxT = M->thrust_max;
  double V[NL][3];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const double ks = M->m_f_rate * M->rotor_sigma[i];
    V[i][0] = fma(P[i][1], U[i][2], fma(-P[i][2], U[i][1], ks * U[i][0]));
    V[i][1] = fma(P[i][2], U[i][0], fma(-P[i][0], U[i][2], ks * U[i][1]));
    V[i][2] = fma(P[i][0], U[i][1], fma(-P[i][1], U[i][0], ks * U[i][2]));
  }
  // static thrust  lambda = A^T (A A^T)^-1 b  on rows {fz, tx, ty, (tz)}
  double G[D][D];
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const double ar = r == 0 ? U[i][2] : V[i][r - 1];
        const double ac = c == 0 ? U[i][2] : V[i][c - 1];
        s = fma(ar, ac, s);
      }
      G[r][c] = s;
    }
  double trace = 0.0;
#pragma unroll
  for (int r = 0; r < D; ++r) trace += G[r][r];
  double Lm[D][D], Linv[D];
  bool singular = false;
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      double s = G[r][c];
#pragma unroll
      for (int k = 0; k < c; ++k) s = fma(-Lm[r][k], Lm[c][k], s);
      if (r == c) {
        singular = singular || !(s > 1e-14 * trace);
        Linv[r] = rsqrt_nr(s);
        Lm[r][r] = s * Linv[r];
      } else {
        Lm[r][c] = s * Linv[c];
      }
    }
  double zf[D], yb[D];
#pragma unroll
  for (int r = 0; r < D; ++r) {
    double s = r == 0 ? mass * M->gravity : 0.0;
#pragma unroll
    for (int k = 0; k < r; ++k) s = fma(-Lm[r][k], zf[k], s);
    zf[r] = s * Linv[r];
  }
#pragma unroll
  for (int r = D - 1; r >= 0; --r) {
    double s = zf[r];
#pragma unroll
    for (int k = r + 1; k < D; ++k) s = fma(-Lm[k][r], yb[k], s);
    yb[r] = s * Linv[r];
  }
  double lam_min = INFINITY, lam_max = -INFINITY;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    double s = U[i][2] * yb[0];
#pragma unroll
    for (int r = 1; r < D; ++r) s = fma(V[i][r - 1], yb[r], s);
    s = singular ? NAN : s;
    if (thrust_out) thrust_out[i] = s;
    lam_min = fmin(lam_min, s);
    lam_max = fmax(lam_max, s);
  }
  if (singular) { lam_min = NAN; lam_max = NAN; }

  // feasible-control torque distances.  For the pair (i, j) with n = V_i x V_j the two one-sided
  // sums over the other rotors are  dp = sum max(t_k, 0), dm = sum max(-t_k, 0), t_k = n . V_k, so
  //   min(dp, dm) = (sum |t_k| - |sum t_k|) / 2   and   sum t_k = n . sum_k V_k
  // (n is orthogonal to V_i and V_j): one dot product and one |.|-accumulate per rotor, and the
  // scale fmax / |n| applied once.  The flag path skips the whole block when the test is off.
  double Vs[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < NL; ++i) { Vs[0] += V[i][0]; Vs[1] += V[i][1]; Vs[2] += V[i][2]; }
  double fc_t = INFINITY;
  bool fc_t_ok = true, fc_rp_ok = true;
  const double fc_t_thre2 = M->fc_t_min_thre * M->fc_t_min_thre, fc_rp_thre2 = M->fc_rp_min_thre * M->fc_rp_min_thre;
  // (not for the gimbal models: the branch costs their flag kernel more in spills than the block costs)
  if (FULL || S::kGimbal || M->fc_t_min_thre > 0.0) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
      for (int j = i + 1; j < NL; ++j) {
        const double nx = fma(V[i][1], V[j][2], -V[i][2] * V[j][1]);
        const double ny = fma(V[i][2], V[j][0], -V[i][0] * V[j][2]);
        const double nz = fma(V[i][0], V[j][1], -V[i][1] * V[j][0]);
        const double n2 = fma(nx, nx, fma(ny, ny, nz * nz));
        double sabs = 0.0;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
          if (k == i || k == j) continue;
          sabs += fabs(fma(nx, V[k][0], fma(ny, V[k][1], nz * V[k][2])));
        }
        const double ssum = fma(nx, Vs[0], fma(ny, Vs[1], nz * Vs[2]));
        const double dist_n = fmax(sabs - fabs(ssum), 0.0) * (0.5 * fmaxT);  // the distance times |n|
        if (FULL) {
          const double rn = rsqrt_nr1(n2);
          fc_t = fmin(fc_t, !(n2 * rn >= 1e-5) ? 0.0 : dist_n * rn);  // |n| < 1e-5 (n2 = 0 gives NaN): distance 0
        } else {
          // flag path: only "distance >= threshold" is observable; squared, without the normalisation
          fc_t_ok = fc_t_ok && (n2 >= 1e-10) && (dist_n * dist_n >= fc_t_thre2 * n2);
        }
      }
  }
  // roll/pitch version: d_i = sum_j max(c_ij, 0) = (sum c_ij + sum |c_ij|) / 2, c_ij = (V_i x V_j)_z, c_ii = 0
  double fc_rp = INFINITY;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const double na2 = fma(V[i][0], V[i][0], V[i][1] * V[i][1]);
    double sabs = 0.0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      if (j == i) continue;
      sabs += fabs(fma(V[i][1], V[j][0], -V[i][0] * V[j][1]));
    }
    const double ssum = fma(V[i][1], Vs[0], -V[i][0] * Vs[1]);
    const double dist_n = fmax(sabs + ssum, 0.0) * (0.5 * fmaxT);
    if (FULL) {
      const double ra = rsqrt_nr1(na2);
      fc_rp = fmin(fc_rp, !(na2 * ra >= 1e-5) ? 0.0 : dist_n * ra);
    } else {
      fc_rp_ok = fc_rp_ok && (na2 >= 1e-10) && (dist_n * dist_n >= fc_rp_thre2 * na2);
    }
  }
  out.fc_t_min = fc_t;
  out.fc_rp_min = fc_rp;
  out.lam_min = lam_min;
  out.lam_max = lam_max;

  if (FULL) {  // legacy roll/pitch position margin ("minimum_var")
    double ax = 0.0, ay = 0.0;
#pragma unroll
    for (int i = 0; i < NL; ++i) { ax += P[i][0]; ay += P[i][1]; }
    ax *= 1.0 / NL; ay *= 1.0 / NL;
    double sxx = 0.0, sxy = 0.0, syy = 0.0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const double dx = P[i][0] - ax, dy = P[i][1] - ay;
      sxx = fma(dx, dx, sxx); sxy = fma(dx, dy, sxy); syy = fma(dy, dy, syy);
    }
    sxx *= 1.0 / NL; sxy *= 1.0 / NL; syy *= 1.0 / NL;
    const double hd = 0.5 * (sxx - syy);
    const double lmin = 0.5 * (sxx + syy) - sqrt(fma(hd, hd, sxy * sxy));
    // normaliser |p_2 - p_1| (pinned by the reference's logs)
    const double d0 = P[1][0] - P[0][0], d1 = P[1][1] - P[0][1], d2 = P[1][2] - P[0][2];
    out.rp_margin = sqrt(fmax(lmin, 0.0)) / sqrt(fma(d0, d0, fma(d1, d1, d2 * d2)));
  }

  // stabilityCheck(): min slack over the active tests
  double margin = INFINITY;
  if (FULL) {
    if (M->fc_rp_min_thre > 0.0) margin = fmin(margin, fc_rp - M->fc_rp_min_thre);
    if (M->fc_t_min_thre > 0.0) margin = fmin(margin, fc_t - M->fc_t_min_thre);
  } else {  // (the flag path sees only the sign of the margin)
    if (M->fc_rp_min_thre > 0.0 && !fc_rp_ok) margin = -1.0;
    if (M->fc_t_min_thre > 0.0 && !fc_t_ok) margin = -1.0;
  }
  margin = fmin(margin, fmin(lam_min - M->thrust_min, M->thrust_max - lam_max));
  if (S::kGimbal) {
    const double tt = X->tan_edf_max_tilt, r2 = 2.0 * M->edf_radius;
    bool clear = true;  // flag path: only the sign of the margin is observable, and h >= c <=> h^2 >= c^2 for c > 0
#pragma unroll
    for (int a = 0; a < 2 * NL; ++a)
#pragma unroll
      for (int b = a + 1; b < 2 * NL; ++b) {
        if (a / 2 == b / 2) continue;
        const double dx = E[a][0] - E[b][0], dy = E[a][1] - E[b][1], dz = E[a][2] - E[b][2];
        const double h2 = fma(dx, dx, dy * dy), c = fma(fabs(dz), tt, r2);
        if (FULL) margin = fmin(margin, sqrt(h2) - c);
        else clear = clear && (h2 >= c * c);
      }
    if (!FULL && !clear) margin = -1.0;
  }
  if (singular) margin = -INFINITY;
  out.margin = margin;
  out.stable = margin >= 0.0;
}

// ------------------------------------------------------------------------
// fp32 variant of the flag path: walk 1 in fp32 with error bounds
// ------------------------------------------------------------------------
// The stability verdict of a state is the sign of the smallest of a handful of slacks.  Here every slack s_t is
// evaluated in fp32 together with a bound e_t of its fp32 evaluation error, built by first-order propagation from
// two model constants (ModelF32::err_p / err_u: position and rotation-entry error of the fp32 frame walk, the
// joints' fp32 cos/sin included) and doubled.  The verdict is taken in fp32 only when it cannot be wrong:
//   every s_t - e_t > 0  -> certainly stable        some s_t + e_t < 0  -> certainly unstable
// anything else (and every state whose normal matrix or gimbal is close to singular) is "uncertain" and is
// evaluated again by the fp64 screen kernel, so the flags remain those of the fp64 semantics.
DEV void sincos_f32(float x, float& s, float& c) {  // |x| up to a few hundred; ~1e-7 absolute
  const float fn = rintf(x * 0.636619772f);
  float r = fmaf(-fn, 1.57079637f, x);
  r = fmaf(-fn, -4.37113883e-8f, r);
  const int q = static_cast<int>(fn);
  const float z = r * r;
  float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(z, ps, -1.6666654611e-1f);
  const float sr = fmaf(r * z, ps, r);
  float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(z, pc, 4.166664568298827e-2f);
  const float cr = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
  const float a = (q & 1) ? cr : sr;
  const float b = (q & 1) ? sr : cr;
  s = (q & 2) ? -a : a;
  c = ((q + 1) & 2) ? -b : b;
}

DEV float rsqrt_f32(float x) {  // hardware estimate + one Newton step
  const float y = __builtin_amdgcn_rsqf(x);
  return fmaf(y, fmaf(-0.5f * x * y, y, 0.5f), y);
}

template <int AXIS>
DEV void rot_cols_f(float R[9], float c, float s) {
  constexpr int a = (AXIS + 1) % 3, b = (AXIS + 2) % 3;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float ua = R[3 * r + a], ub = R[3 * r + b];
    R[3 * r + a] = fmaf(c, ua, s * ub);
    R[3 * r + b] = fmaf(c, ub, -s * ua);
  }
}
DEV void rot_cols_dyn_f(float R[9], int axis, float c, float s) {
  if (axis == 0) rot_cols_f<0>(R, c, s);
  else if (axis == 1) rot_cols_f<1>(R, c, s);
  else rot_cols_f<2>(R, c, s);
}
DEV void xform_f(const float R[9], const float p[3], float tx, float ty, float tz, float y[3]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) y[r] = fmaf(R[3 * r], tx, fmaf(R[3 * r + 1], ty, fmaf(R[3 * r + 2], tz, p[r])));
}

constexpr float kF32U = 5.9604645e-8f;  // 2^-24

// returns 1: certainly stable, 0: certainly unstable, -1: uncertain.  Leaves the joints' fp32 cos/sin in
// csf[(2j) * kBlock + tid], csf[(2j+1) * kBlock + tid] for walk 2.  Rf = float(Rw).
// dbg (nullable): [0] smallest s - e, [1] smallest s + e over the tests that could be bounded, [2] 1 when a test could not
// mdbg (nullable): the same two for the metric slacks -- an enclosure of the full path's fp64 margin when dbg[2] == 0
template <int ROBOT, int NL>
DEV int model_update_f32(const EvalArgs& A, const DevExtras* __restrict__ X, const double* st, const float Rf[9],
                         lds_float* csf, float* dbg, float* mdbg = nullptr) {
  using S = Shape<ROBOT, NL>;
  constexpr int NS = S::kSeg, NJ = S::kJoint, D = S::kDof;
  const ModelF32& F = X->f32;
  const int tid = threadIdx.x;
  float jc[NJ], js[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    sincos_f32(static_cast<float>(st[7 + j]), js[j], jc[j]);
    csf[(2 * j) * kBlock + tid] = jc[j];
    csf[(2 * j + 1) * kBlock + tid] = js[j];
  }
  const float bs = 2.0f * A.f32_bound_scale;
  // frame walk relative to the root origin (stability does not depend on the root position)
  float R[9], p[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = Rf[k];
  float cog[3] = {0.0f, 0.0f, 0.0f};
  float P[NL][3], U[NL][3];
  float E[S::kGimbal ? 2 * NL : 1][3];
  float err_u = F.err_u;  // (gimbal models: grows with the conditioning of the gimbal solve)
  bool unc = false;
  static_for<0, NS>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k > 0) {
      float pn[3];
      xform_f(R, p, F.seg_t[k][0], F.seg_t[k][1], F.seg_t[k][2], pn);
      p[0] = pn[0]; p[1] = pn[1]; p[2] = pn[2];
      rot_cols_f<S::axis(k)>(R, jc[S::joint(k)], js[S::joint(k)]);
    }
    float cw[3];
    const float z3[3] = {0.0f, 0.0f, 0.0f};
    xform_f(R, z3, F.seg_mcom[k][0], F.seg_mcom[k][1], F.seg_mcom[k][2], cw);  // mass * R * com
#pragma unroll
    for (int c = 0; c < 3; ++c) cog[c] += fmaf(F.seg_mass[k], p[c], cw[c]);
    const bool is_rotor_seg = S::kGimbal ? ((k & 1) == 0) : true;
    if (is_rotor_seg) {
      const int i = S::kGimbal ? k / 2 : k;
      xform_f(R, p, F.rotor_pos[i][0], F.rotor_pos[i][1], F.rotor_pos[i][2], P[i]);
      if (S::kGimbal) {
        const float zlx = R[6], zly = R[7], zlz = R[8];
        const float rho2 = fmaf(zly, zly, zlz * zlz);
        unc = unc || !(rho2 > 1e-2f);  // gimbal solve ill conditioned: the fp64 kernel decides
        const float ir = rsqrt_f32(rho2);
        const float cr = zlz * ir, sr = -zly * ir;
        const float w = fmaf(-zly, sr, zlz * cr);
        const float nn = rsqrt_f32(fmaf(zlx, zlx, w * w));
        const float cp = w * nn, sp = zlx * nn;
        err_u = fmaxf(err_u, 6.0f * F.err_u * ir);
        const float ux = sp, uy = -sr * cp, uz = cr * cp;
        const float ey = cr, ez = sr;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          U[i][r] = fmaf(R[3 * r], ux, fmaf(R[3 * r + 1], uy, R[3 * r + 2] * uz));
          P[i][r] = fmaf(F.rotor_roll_z, fmaf(R[3 * r + 2], cr, -R[3 * r + 1] * sr), P[i][r]);  // thrust-frame origin
          const float yw = fmaf(R[3 * r + 1], ey, R[3 * r + 2] * ez) * F.edf_offset_y;
          E[2 * i][r] = P[i][r] + yw;
          E[2 * i + 1][r] = P[i][r] - yw;
        }
      } else {
        U[i][0] = R[2]; U[i][1] = R[5]; U[i][2] = R[8];
      }
    }
  });
#pragma unroll
  for (int c = 0; c < 3; ++c) cog[c] *= F.inv_mass;
  float Pm = 0.0f;
#pragma unroll
  for (int i = 0; i < NL; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      P[i][c] -= cog[c];
      Pm = fmaxf(Pm, fabsf(P[i][c]));
    }
  // V = P x U + ks U and a bound on the error of every component: P (relative to the CoG) carries 2 err_p per
  // component, U carries err_ua (planar chains: every thrust axis IS the root's z axis, only its conversion to
  // fp32 counts; gimbal models: err_u grown by the conditioning of the gimbal solve), plus the rounding of the
  // three products.  Component-wise, because the rows of the wrench matrix differ in scale by orders of
  // magnitude (planar HYDRUS: the yaw-torque row is +-kappa).
  const float err_ua = bs * (S::kGimbal ? err_u : 2.0f * kF32U);
  // (gimbal models: the thrust-frame origin's offset up the roll frame's z axis rides on the solved roll)
  const float err_pc = bs * (2.0f * F.err_p + (S::kGimbal ? 2.0f * fabsf(F.rotor_roll_z) * err_u : 0.0f));
  float V[NL][3], dVc[NL][3], Vm = 1.0f, dV = 0.0f;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const float ks = F.rotor_ks[i];
    V[i][0] = fmaf(P[i][1], U[i][2], fmaf(-P[i][2], U[i][1], ks * U[i][0]));
    V[i][1] = fmaf(P[i][2], U[i][0], fmaf(-P[i][0], U[i][2], ks * U[i][1]));
    V[i][2] = fmaf(P[i][0], U[i][1], fmaf(-P[i][1], U[i][0], ks * U[i][2]));
    const float au[3] = {fabsf(U[i][0]), fabsf(U[i][1]), fabsf(U[i][2])};
    const float ap[3] = {fabsf(P[i][0]), fabsf(P[i][1]), fabsf(P[i][2])};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
      const float mag = fmaf(ap[c1], au[c2], fmaf(ap[c2], au[c1], fabsf(ks) * au[c]));
      dVc[i][c] = fmaf(err_pc, au[c1] + au[c2], fmaf(err_ua, ap[c1] + ap[c2] + fabsf(ks), 4.0f * kF32U * mag));
      dV = fmaxf(dV, dVc[i][c]);
    }
    Vm = fmaxf(Vm, fmaxf(fabsf(V[i][0]), fmaxf(fabsf(V[i][1]), fabsf(V[i][2]))));
  }
  float lo = INFINITY, hi = INFINITY;  // smallest s - e, smallest s + e
  // ... and the same enclosure of the METRIC slacks (the margin of the full path: the feasible-control tests there are
  // distance - threshold, here distance * |n| - threshold * |n|): numerator sl +- e over denominator den +- e_den
  float lo_m = INFINITY, hi_m = INFINITY;
  auto test = [&](float sl, float e, float den = 1.0f, float e_den = 0.0f) {
    lo = fminf(lo, sl - e); hi = fminf(hi, sl + e);
    const float a = sl - e, b = sl + e, dl = fmaxf(den - e_den, 1e-30f), dh = den + e_den;
    lo_m = fminf(lo_m, (a >= 0.0f ? a / dh : a / dl) - 4.0f * kF32U * fabsf(a / dl));
    hi_m = fminf(hi_m, (b >= 0.0f ? b / dl : b / dh) + 4.0f * kF32U * fabsf(b / dl));
  };

  // ---- static thrust: lambda = A^T (A A^T)^-1 b on rows {fz, tx, ty, (tz)} ----------------
  // First-order perturbation of the minimum-norm solution for A -> A + E, |E_ri| <= dA_ri:
  //   d lambda = -A^+ E lambda + (I - A^+ A) E^T y,  y = G^-1 b,  A^+ = A^T G^-1,  G = A A^T
  //   |d lambda_i| <= sum_r |a_ri| (|G^-1| |E| |lambda|)_r + sum_r dA_ri |y_r|
  // with |G^-1| taken entry-wise from the inverse of the Cholesky factor; the rounding of the fp32 solve itself is
  // a relative perturbation of A of a few u (folded into dA).
  {
    float G[D][D];
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          const float ar = r == 0 ? U[i][2] : V[i][r - 1];
          const float ac = c == 0 ? U[i][2] : V[i][c - 1];
          acc = fmaf(ar, ac, acc);
        }
        G[r][c] = acc;
      }
    float Lm[D][D], Linv[D];
    bool weak = false;
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        float acc = G[r][c];
#pragma unroll
        for (int k = 0; k < c; ++k) acc = fmaf(-Lm[r][k], Lm[c][k], acc);
        if (r == c) {
          weak = weak || !(acc > 1e-3f * G[r][r]);  // the row is (almost) a combination of the rows before it
          Linv[r] = rsqrt_f32(acc);
          Lm[r][r] = acc * Linv[r];
        } else {
          Lm[r][c] = acc * Linv[c];
        }
      }
    float Li[D][D];  // inverse of the Cholesky factor (lower triangular)
#pragma unroll
    for (int c = 0; c < D; ++c)
#pragma unroll
      for (int r = c; r < D; ++r) {
        float acc = r == c ? 1.0f : 0.0f;
#pragma unroll
        for (int k = c; k < r; ++k) acc = fmaf(-Lm[r][k], Li[k][c], acc);
        Li[r][c] = acc * Linv[r];
      }
    // y = G^-1 b with b = weight * e0, and |G^-1| <= |Li|^T |Li| entry-wise
    float yb[D], Gi[D][D];
#pragma unroll
    for (int r = 0; r < D; ++r) {
      float acc = 0.0f;
#pragma unroll
      for (int k = r; k < D; ++k) acc = fmaf(Li[k][r], Li[k][0], acc);
      yb[r] = acc * F.weight;
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        float g = 0.0f;
#pragma unroll
        for (int k = r; k < D; ++k) g = fmaf(fabsf(Li[k][r]), fabsf(Li[k][c]), g);
        Gi[r][c] = g;
        Gi[c][r] = g;
      }
    }
    float lam[NL], lam_min = INFINITY, lam_max = -INFINITY;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float acc = U[i][2] * yb[0];
#pragma unroll
      for (int r = 1; r < D; ++r) acc = fmaf(V[i][r - 1], yb[r], acc);
      lam[i] = acc;
      lam_min = fminf(lam_min, acc);
      lam_max = fmaxf(lam_max, acc);
    }
    float El[D];  // (|E| |lambda|)_r
#pragma unroll
    for (int r = 0; r < D; ++r) {
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const float a_ri = r == 0 ? U[i][2] : V[i][r - 1];
        const float d_ri = (r == 0 ? err_ua : dVc[i][r - 1]) + 16.0f * kF32U * fabsf(a_ri);
        acc = fmaf(d_ri, fabsf(lam[i]), acc);
      }
      El[r] = acc;
    }
    float w[D];
#pragma unroll
    for (int r = 0; r < D; ++r) {
      float acc = 0.0f;
#pragma unroll
      for (int c = 0; c < D; ++c) acc = fmaf(Gi[r][c], El[c], acc);
      w[r] = acc;
    }
    float e_lam = 0.0f;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float acc = 0.0f;
#pragma unroll
      for (int r = 0; r < D; ++r) {
        const float a_ri = r == 0 ? U[i][2] : V[i][r - 1];
        const float d_ri = (r == 0 ? err_ua : dVc[i][r - 1]) + 16.0f * kF32U * fabsf(a_ri);
        acc = fmaf(fabsf(a_ri), w[r], fmaf(d_ri, fabsf(yb[r]), acc));
      }
      e_lam = fmaxf(e_lam, acc);
    }
    e_lam = fmaf(8.0f * kF32U, fmaxf(fabsf(lam_min), fabsf(lam_max)), e_lam);
    if (weak || !(e_lam < INFINITY)) unc = true;
    else {
      test(lam_min - F.thrust_min, e_lam);
      test(F.thrust_max - lam_max, e_lam);
    }
  }

  float Vs[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < NL; ++i) { Vs[0] += V[i][0]; Vs[1] += V[i][1]; Vs[2] += V[i][2]; }
  const float half_fmax = 0.5f * F.thrust_max;
  const float e_c = 4.0f * Vm * dV + 8.0f * kF32U * Vm * Vm;  // a component of V_i x V_j (norm-wise)
  // ---- feasible-control torque distances (as model_update: distance * |n| against threshold * |n|) ----
  // Error bounds component by component: the triple products n . V_k are small differences of large terms
  // whenever the thrust axes are (nearly) parallel -- every planar chain -- and a norm-wise bound would swamp them.
  if (F.fc_t_thre > 0.0f) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
      for (int j = i + 1; j < NL; ++j) {
        float nv[3], dn[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
          nv[c] = fmaf(V[i][c1], V[j][c2], -V[i][c2] * V[j][c1]);
          const float mag = fmaf(fabsf(V[i][c1]), fabsf(V[j][c2]), fabsf(V[i][c2] * V[j][c1]));
          dn[c] = fmaf(fabsf(V[i][c1]), dVc[j][c2], fmaf(fabsf(V[j][c2]), dVc[i][c1],
                  fmaf(fabsf(V[i][c2]), dVc[j][c1], fmaf(fabsf(V[j][c1]), dVc[i][c2], 4.0f * kF32U * mag))));
        }
        const float n2 = fmaf(nv[0], nv[0], fmaf(nv[1], nv[1], nv[2] * nv[2]));
        const float an[3] = {fabsf(nv[0]), fabsf(nv[1]), fabsf(nv[2])};
        float sabs = 0.0f, e_sum = 0.0f;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
          // (k = i, j: exactly 0 in exact arithmetic; they enter ssum below with their rounding, bounded the same way)
          const float t = fmaf(nv[0], V[k][0], fmaf(nv[1], V[k][1], nv[2] * V[k][2]));
          const float mag = fmaf(an[0], fabsf(V[k][0]), fmaf(an[1], fabsf(V[k][1]), an[2] * fabsf(V[k][2])));
          e_sum += fmaf(an[0], dVc[k][0], fmaf(an[1], dVc[k][1], fmaf(an[2], dVc[k][2],
                   fmaf(fabsf(V[k][0]), dn[0], fmaf(fabsf(V[k][1]), dn[1], fmaf(fabsf(V[k][2]), dn[2], 4.0f * kF32U * mag))))));
          if (k != i && k != j) sabs += fabsf(t);
        }
        const float ssum = fmaf(nv[0], Vs[0], fmaf(nv[1], Vs[1], nv[2] * Vs[2]));
        const float dist_n = fmaxf(sabs - fabsf(ssum), 0.0f) * half_fmax;
        const float nn = sqrtf(n2);
        const float e_nn = dn[0] + dn[1] + dn[2];
        const float e = fmaf(half_fmax, 2.0f * e_sum + 8.0f * kF32U * (sabs + fabsf(ssum)), F.fc_t_thre * e_nn);
        // the fp64 rule "|n| < 1e-5 -> distance 0" must be decidable as well
        if (!(nn > 1e-5f + 2.0f * e_nn)) unc = true;
        else test(dist_n - F.fc_t_thre * nn, e, nn, e_nn);
      }
  }
  // ---- roll / pitch distances ----
  if (F.fc_rp_thre > 0.0f) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const float na2 = fmaf(V[i][0], V[i][0], V[i][1] * V[i][1]);
      float sabs = 0.0f;
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        if (j == i) continue;
        sabs += fabsf(fmaf(V[i][1], V[j][0], -V[i][0] * V[j][1]));
      }
      const float ssum = fmaf(V[i][1], Vs[0], -V[i][0] * Vs[1]);
      const float dist_n = fmaxf(sabs + ssum, 0.0f) * half_fmax;
      const float na = sqrtf(na2);
      const float e = half_fmax * (2.0f * NL) * e_c + F.fc_rp_thre * 2.0f * dV;
      if (!(na > 1e-5f + 4.0f * dV)) unc = true;
      else test(dist_n - F.fc_rp_thre * na, e, na, 4.0f * dV);
    }
  }
  // ---- EDF overlap (gimbal models) ----
  if (S::kGimbal) {
    const float dE = bs * (4.0f * F.err_p + 4.0f * (F.edf_offset_y + fabsf(F.rotor_roll_z)) * err_u + 8.0f * kF32U * (Pm + 1.0f));
#pragma unroll
    for (int a = 0; a < 2 * NL; ++a)
#pragma unroll
      for (int b = a + 1; b < 2 * NL; ++b) {
        if (a / 2 == b / 2) continue;
        const float dx = E[a][0] - E[b][0], dy = E[a][1] - E[b][1], dz = E[a][2] - E[b][2];
        const float h = sqrtf(fmaf(dx, dx, dy * dy)), c = fmaf(fabsf(dz), F.tan_edf_max_tilt, F.edf_two_r);
        test(h - c, (2.0f + F.tan_edf_max_tilt) * dE);
      }
  }
  if (dbg) { dbg[0] = lo; dbg[1] = hi; dbg[2] = unc ? 1.0f : 0.0f; }
  if (mdbg) { mdbg[0] = lo_m; mdbg[1] = hi_m; }
  if (hi < 0.0f) return 0;            // some test certainly fails (NaN compares false: uncertain)
  if (!unc && lo > 0.0f) return 1;    // every test certainly passes
  return -1;
}

// ------------------------------------------------------------------------
// walk 2 building blocks: the chain frame, primitive poses, box-frame tests
// ------------------------------------------------------------------------
// frame of segment k from the frame of segment k-1; joint cos/sin from LDS slot `src`
DEV void advance_frame(const mvk_model* __restrict__ M, const Smem& sm, int src, int k, double R[9], double p[3]) {
  double pn[3];
  xform(R, p, M->seg_t[k][0], M->seg_t[k][1], M->seg_t[k][2], pn);
  p[0] = pn[0]; p[1] = pn[1]; p[2] = pn[2];
  const int jn = M->seg_joint[k];
  double c, s;
  if (sm.gcs) {  // (wave-uniform) coalesced row reads: lane-contiguous columns
    c = sm.gcs[static_cast<uint64_t>(2 * jn) * sm.gcs_ld + sm.gcs_col];
    s = sm.gcs[static_cast<uint64_t>(2 * jn + 1) * sm.gcs_ld + sm.gcs_col];
  } else {
    c = sm.cs[(2 * jn) * kBlock + src];
    s = sm.cs[(2 * jn + 1) * kBlock + src];
  }
  rot_cols_dyn(R, M->seg_axis[k], c, s);
}

// pose of primitive q (wave-uniform) riding on the segment frame (R, p): world
// centre pc and, when WANT_ROT, world rotation PR (collision_avoidance.cpp:123-132:
// root * link * collision origin)
template <bool GIMBAL, bool WANT_ROT>
DEV void prim_pose(const mvk_model* __restrict__ M, int q, const double R[9], const double p[3], double pc[3], double PR[9]) {
  double B[9], bp[3];
#pragma unroll
  for (int e = 0; e < 9; ++e) B[e] = R[e];
  bp[0] = p[0]; bp[1] = p[1]; bp[2] = p[2];
  if (GIMBAL && M->prim_frame[q] != MVK_FRAME_SEG) {
    const int i = M->prim_rotor[q];
    double o[3];
    xform(R, p, M->rotor_pos[i][0], M->rotor_pos[i][1], M->rotor_pos[i][2], o);
    bp[0] = o[0]; bp[1] = o[1]; bp[2] = o[2];
    double cr, sr, cp, sp;
    gimbal_cs(R[6], R[7], R[8], cr, sr, cp, sp);
    rot_cols<0>(B, cr, sr);
    if (M->prim_frame[q] == MVK_FRAME_GIMBAL_PITCH) rot_cols<1>(B, cp, sp);
  }
  xform(B, bp, M->prim_pos[q][0], M->prim_pos[q][1], M->prim_pos[q][2], pc);
  if (WANT_ROT) {
    double Lr[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) Lr[e] = M->prim_rot[q][e];
    mat_mul(B, Lr, PR);
  }
}

DEV void to_box_frame(const DevObs& O, const double pc[3], double t[3]) {
  const double dx = pc[0] - O.c[0], dy = pc[1] - O.c[1], dz = pc[2] - O.c[2];
  t[0] = fma(O.R[0], dx, fma(O.R[3], dy, O.R[6] * dz));
  t[1] = fma(O.R[1], dx, fma(O.R[4], dy, O.R[7] * dz));
  t[2] = fma(O.R[2], dx, fma(O.R[5], dy, O.R[8] * dz));
}

// squared distance from a point (already in the box frame) to the box
DEV double point_box_dist2(const double t[3], const double h[3]) {
  const double e0 = excess_abs(t[0], h[0]), e1 = excess_abs(t[1], h[1]), e2 = excess_abs(t[2], h[2]);
  return fma(e0, e0, fma(e1, e1, e2 * e2));
}

// box primitive axes in the obstacle frame  Rr = O.R^T * PR
DEV void box_rel(const DevObs& O, const double PR[9], double Rr[9]) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      Rr[3 * r + c] = fma(O.R[r], PR[c], fma(O.R[3 + r], PR[3 + c], O.R[6 + r] * PR[6 + c]));
}

// exact separation of two disjoint boxes (A: centre t, axes Rr, half extents a, in the frame of B = [-b,b]^3):
// the closest features always include a point on an edge of one of them
DEV double box_box_exact(const double t[3], const double Rr[9], const double a[3], const double b[3]) {
  double s2 = box_edges_to_box2(t, Rr, a, b);
  double Rt[9], tb[3];  // B in A's frame
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Rt[3 * r + c] = Rr[3 * c + r];
#pragma unroll
  for (int r = 0; r < 3; ++r) tb[r] = -fma(Rr[r], t[0], fma(Rr[3 + r], t[1], Rr[6 + r] * t[2]));
  s2 = fmin(s2, box_edges_to_box2(tb, Rt, b, a));
  return sqrt(s2);
}

// The separating-axis gap bounds a pair's distance from below only up to rounding: for two pairs at the same
// distance (the flat bodies of a planar robot under one box face) the gap of the second can come out an ulp above the
// exact distance of the first, and whether the second is then evaluated would depend on the order of the pairs.  A
// pair is therefore skipped only when its gap exceeds the running minimum by kSatSlack, which makes the minimum
// independent of the order (distance_kernel and distance_small_kernel visit the pairs in different orders).
constexpr double kSatSlack = 1e-9;
constexpr double kSatSlackF32 = 2e-6;  // the fp32 variant's: a few fp32 roundings of a gap of a metre

// exact test of one primitive against one box (primitive kind is wave-uniform); t = primitive
// centre in the box frame.  SIGNED: signed distance; otherwise only the sign is meaningful.
// `cutoff` (SIGNED only): the caller keeps a running minimum; when the separating-axis bound of a box
// pair, a lower bound of its distance, already reaches it, the exact edge-edge distance is skipped and
// that bound is returned (>= cutoff, so the minimum is not affected).
template <bool SIGNED>
DEV double narrow(int type, const double PR[9], double d0, double d1, double d2, const DevObs& O, const double t[3],
                  double cutoff = INFINITY) {
  double d;
  if (type == MVK_PRIM_CAPSULE) {
    // axis = local z in world = third column of PR, into the box frame
    double a[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) a[c] = fma(O.R[c], PR[2], fma(O.R[3 + c], PR[5], O.R[6 + c] * PR[8]));
    double p0[3], dd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { p0[c] = fma(-d1, a[c], t[c]); dd[c] = 2.0 * d1 * a[c]; }
    const double s2 = seg_box_dist2(p0, dd, O.h);
    if (SIGNED) d = sqrt(s2) - d0;
    else d = (s2 <= d0 * d0) ? 0.0 : 1.0;
  } else {
    double Rr[9];
    box_rel(O, PR, Rr);
    const double a[3] = {d0, d1, d2};
    if (SIGNED) {
      d = box_box_sat<true>(t, Rr, a, O.h, cutoff + kSatSlack);
      if (d > 0.0 && d < cutoff + kSatSlack) d = box_box_exact(t, Rr, a, O.h);
    } else {
      d = box_box_sat<false>(t, Rr, a, O.h);
    }
  }
  return d;
}

// level 0 of the flag path: the broadphase grid (mvk_grid.cpp).  The cell holding c lists
// every box within the largest segment radius of any point of the cell; [gb, ge) are its
// groups of kGridGroup ids.  Points outside the grid volume have no box within that radius.
DEV void grid_cell(const GridDev& G, const double c[3], bool on, unsigned& gb, unsigned& ge) {
  const double gx = (c[0] - G.origin[0]) * G.inv_h, gy = (c[1] - G.origin[1]) * G.inv_h, gz = (c[2] - G.origin[2]) * G.inv_h;
  const bool inside = on && gx >= 0.0 && gy >= 0.0 && gz >= 0.0 && gx < static_cast<double>(G.n[0]) &&
                      gy < static_cast<double>(G.n[1]) && gz < static_cast<double>(G.n[2]);
  gb = 0; ge = 0;
  if (inside) {
    const unsigned cell = (static_cast<unsigned>(static_cast<int>(gz)) * G.n[1] + static_cast<unsigned>(static_cast<int>(gy))) * G.n[0] +
                          static_cast<unsigned>(static_cast<int>(gx));
    gb = G.start[cell];
    ge = G.start[cell + 1];
  }
}

// the same, and whether the point lies inside the grid's volume (the far grid answers only there)
DEV bool grid_cell_in(const GridDev& G, const double c[3], bool on, unsigned& gb, unsigned& ge) {
  const double gx = (c[0] - G.origin[0]) * G.inv_h, gy = (c[1] - G.origin[1]) * G.inv_h, gz = (c[2] - G.origin[2]) * G.inv_h;
  const bool inside = on && G.start != nullptr && gx >= 0.0 && gy >= 0.0 && gz >= 0.0 && gx < static_cast<double>(G.n[0]) &&
                      gy < static_cast<double>(G.n[1]) && gz < static_cast<double>(G.n[2]);
  gb = 0; ge = 0;
  if (inside) {
    const unsigned cell = (static_cast<unsigned>(static_cast<int>(gz)) * G.n[1] + static_cast<unsigned>(static_cast<int>(gy))) * G.n[0] +
                          static_cast<unsigned>(static_cast<int>(gx));
    gb = G.start[cell];
    ge = G.start[cell + 1];
  }
  return inside;
}

// fp32 squared distance from a point to a world box
DEV float point_box_dist2_f32(const DevObsF& F, float x, float y, float z) {
  const float dx = x - F.c[0], dy = y - F.c[1], dz = z - F.c[2];
  const float e0 = fmaxf(fabsf(fmaf(F.R[0], dx, fmaf(F.R[3], dy, F.R[6] * dz))) - F.h[0], 0.0f);
  const float e1 = fmaxf(fabsf(fmaf(F.R[1], dx, fmaf(F.R[4], dy, F.R[7] * dz))) - F.h[1], 0.0f);
  const float e2 = fmaxf(fabsf(fmaf(F.R[2], dx, fmaf(F.R[5], dy, F.R[8] * dz))) - F.h[2], 0.0f);
  return fmaf(e0, e0, fmaf(e1, e1, e2 * e2));
}

// The fp32 classification of a (primitive, box) pair by the primitive's centre distance.
// f32_delta bounds the fp32 evaluation error of that distance for coordinates within
// f32_range, so
//   l2 <= hit2  =>  the ball inscribed in the primitive truly touches the box   (collision)
//   l2 >  may2  =>  the ball circumscribing the primitive truly misses the box  (no contact)
// and only the pairs in between need fp64.  Outside f32_range nothing is decided in fp32.
struct PrimF32 {
  float x[kSegPrimMax], y[kSegPrimMax], z[kSegPrimMax], hit2[kSegPrimMax], may2[kSegPrimMax];
};

template <bool GIMBAL>
DEV void prims_f32(const EvalArgs& A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X, int pb, int np,
                   const double R[9], const double p[3], bool f32_safe, PrimF32& F) {
#pragma unroll
  for (int j = 0; j < kSegPrimMax; ++j) {
    F.x[j] = F.y[j] = F.z[j] = 0.0f;
    F.hit2[j] = -1.0f;
    F.may2[j] = -1.0f;
    if (j < np) {
      const int q = X->prim_order[pb + j];
      double pc[3];
      prim_pose<GIMBAL, false>(M, q, R, p, pc, nullptr);
      F.x[j] = static_cast<float>(pc[0]); F.y[j] = static_cast<float>(pc[1]); F.z[j] = static_cast<float>(pc[2]);
      const float rin = static_cast<float>(X->prim_inner[q]) * 0.999999f - A.f32_delta;
      const float rb = (static_cast<float>(X->prim_bound[q]) + A.f32_delta) * 1.000001f;
      F.hit2[j] = (f32_safe && rin > 0.0f) ? rin * rin : -1.0f;
      F.may2[j] = f32_safe ? rb * rb : INFINITY;
    }
  }
}

// ------------------------------------------------------------------------
// flag path, phase 1: fp32 screen of every (primitive, box) pair the grid lists
// ------------------------------------------------------------------------
// hit     : some pair is a certain collision                 -> the state is decided (invalid)
// segmask : bit k = a pair on segment k needs the fp64 test  -> undecided unless hit
// neither                                                   -> no contact, decided
//
// Most (state, segment) pairs have an empty candidate list, so a loop over a lane's own list
// would run with a quarter of the wave's lanes.  Instead every lane PUSHES its non-empty
// (segment, list) as a work item into the wave's ring in LDS, and whenever a full wave of items is
// waiting the wave POPS them one per lane: every lane tests one group of kGridGroup boxes of some
// lane's segment, ORs the verdict into that lane's result word and pushes the rest of the list
// back.  Items of a state already known to collide are dropped.
constexpr int kRing = 128;       // items per wave: < 64 waiting + at most 64 pushed by one segment
constexpr int kRingWords = 3 * kSegPrimMax + 2;  // primitive centres; first group + tag; end group + primitive count
constexpr unsigned kResHit = 0x80000000u;

struct ScreenLds {
  lds_uint* ring;      // this wave's items, [kRingWords][kRing]
  lds_uint* res;       // this wave's result words, [64]: kResHit | maybe bits per segment
  lds_float* thr;      // [segment][slot][2]: squared centre distances (certain hit, possible contact)
};

// screen thresholds of the primitive in slot j of segment k (see PrimF32 above)
DEV void screen_thresholds(const EvalArgs& A, const DevExtras* __restrict__ X, int k, int j, float delta, float& hit2, float& may2) {
  const int pb = X->seg_prim_begin[k], np = X->seg_prim_begin[k + 1] - pb;
  hit2 = -1.0f;
  may2 = -1.0f;
  if (j < np) {
    const int q = X->prim_order[pb + j];
    const float rin = static_cast<float>(X->prim_inner[q]) * 0.999999f - delta;
    const float rb = (static_cast<float>(X->prim_bound[q]) + delta) * 1.000001f;
    hit2 = rin > 0.0f ? rin * rin : -1.0f;
    may2 = rb * rb;
  }
}

// one pop: lane i takes item head+i, tests one group, pushes the remainder
DEV void screen_pop(const EvalArgs& A, const DevObsF* __restrict__ obsf, const ScreenLds& L, unsigned& head, unsigned& qn) {
  const int lane = threadIdx.x & 63;
  const unsigned m = qn < 64u ? qn : 64u;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  PROF_WAVE(5);
  const bool active = static_cast<unsigned>(lane) < m;
  const unsigned at = (head + lane) & (kRing - 1);
  unsigned wa = 0, wb = 0;
  float c[3 * kSegPrimMax];
  if (active) {
    wa = L.ring[(3 * kSegPrimMax) * kRing + at];
    wb = L.ring[(3 * kSegPrimMax + 1) * kRing + at];
#pragma unroll
    for (int e = 0; e < 3 * kSegPrimMax; ++e) c[e] = __uint_as_float(L.ring[e * kRing + at]);
  }
  // word a: first group (21 bits: kGridMaxItems / kGridGroup), source lane (6), segment (4), fp32-safe (1)
  // word b: end group (22 bits: it may equal the group count), primitives on the segment (2)
  const unsigned g = wa & 0x1FFFFFu, src = (wa >> 21) & 63u, k = (wa >> 27) & 15u;
  const bool safe = wa >> 31;
  const unsigned gend = wb & 0x3FFFFFu, np = wb >> 22;
  bool work = active && !(__hip_atomic_load(L.res + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) & kResHit);
  PROF_LANES(6, work);
  bool more = false;
  if (work) {
    const uint2 w = *reinterpret_cast<const uint2*>(A.grid.items + static_cast<size_t>(g) * kGridGroup);
    const unsigned ids[kGridGroup] = {w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16};
    DevObsF F[kGridGroup];
#pragma unroll
    for (int e = 0; e < kGridGroup; ++e) F[e] = obsf[ids[e] == kGridNone ? static_cast<unsigned>(A.n_obs) : ids[e]];
    bool hit = false, maybe = false;
#pragma unroll
    for (int j = 0; j < kSegPrimMax; ++j) {
      if (!__any(static_cast<unsigned>(j) < np)) break;
      float hit2 = L.thr[(k * kSegPrimMax + j) * 2], may2 = L.thr[(k * kSegPrimMax + j) * 2 + 1];
      if (!safe) { hit2 = -1.0f; may2 = may2 < 0.0f ? -1.0f : INFINITY; }  // outside the fp32 range nothing is decided here
#pragma unroll
      for (int e = 0; e < kGridGroup; ++e) {
        const float l2 = point_box_dist2_f32(F[e], c[3 * j], c[3 * j + 1], c[3 * j + 2]);
        const bool real = ids[e] != kGridNone;
        hit = hit || (real && l2 <= hit2);
        maybe = maybe || (real && l2 <= may2);
      }
    }
    if (hit) __hip_atomic_fetch_or(L.res + src, kResHit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    else if (maybe) __hip_atomic_fetch_or(L.res + src, 1u << k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    more = !hit && g + 1 < gend;
  }
  head = (head + m) & (kRing - 1);
  qn -= m;
  const unsigned long long bal = __ballot(more);
  if (bal) {
    const unsigned to = (head + qn + static_cast<unsigned>(__popcll(bal & ((1ull << lane) - 1ull)))) & (kRing - 1);
    if (more) {
#pragma unroll
      for (int e = 0; e < 3 * kSegPrimMax; ++e) L.ring[e * kRing + to] = __float_as_uint(c[e]);
      L.ring[(3 * kSegPrimMax) * kRing + to] = wa + 1u;  // next group, same tag (g + 1 < gend: no carry into the tag)
      L.ring[(3 * kSegPrimMax + 1) * kRing + to] = wb;
    }
    qn += static_cast<unsigned>(__popcll(bal));
  }
}

template <int ROBOT, int NL>
DEV void collide_screen(const EvalArgs& A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
                        const DevObsF* __restrict__ obsf, const double pos[3], const double Rw[9], const Smem& sm,
                        const ScreenLds& L, bool live, bool& hit, unsigned& segmask, unsigned long long* prof_pop) {
  using S = Shape<ROBOT, NL>;
  const int tid = threadIdx.x, lane = threadIdx.x & 63;
  const bool f32_safe = fmax(fabs(pos[0]), fmax(fabs(pos[1]), fabs(pos[2]))) + X->reach <= A.f32_range;
  double R[9], p[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = Rw[k];
  p[0] = pos[0]; p[1] = pos[1]; p[2] = pos[2];
  L.res[lane] = 0u;
  unsigned head = 0, qn = 0;  // wave-uniform
  // push the item of one segment (centres c, list [gb, ge), tag bits) unless the list is empty or
  // the state is already known to collide
  auto push_item = [&](const float* c, unsigned gb, unsigned ge, unsigned tag_a, unsigned tag_b) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const bool push = gb < ge && !(__hip_atomic_load(L.res + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) & kResHit);
    const unsigned long long bal = __ballot(push);
    if (bal) {
      const unsigned to = (head + qn + static_cast<unsigned>(__popcll(bal & ((1ull << lane) - 1ull)))) & (kRing - 1);
      if (push) {
#pragma unroll
        for (int e = 0; e < 3 * kSegPrimMax; ++e) L.ring[e * kRing + to] = __float_as_uint(c[e]);
        L.ring[(3 * kSegPrimMax) * kRing + to] = gb | tag_a;
        L.ring[(3 * kSegPrimMax + 1) * kRing + to] = ge | tag_b;
      }
      qn += static_cast<unsigned>(__popcll(bal));
    }
  };
  // The lookup of a segment's grid cell is a dependent global load; its result is consumed one
  // segment later (software pipeline of depth one), behind the next segment's frame arithmetic.
  float pc_[3 * kSegPrimMax];
  unsigned p_gb = 0, p_ge = 0, p_ta = 0, p_tb = 0;
  bool pending = false;  // wave-uniform
#pragma unroll 1
  for (int k = 0; k < S::kSeg; ++k) {
    const SegTab& T = X->seg_tab[k];
    if (k > 0) {
      double pn[3];
      xform(R, p, T.t[0], T.t[1], T.t[2], pn);
      p[0] = pn[0]; p[1] = pn[1]; p[2] = pn[2];
      rot_cols_dyn(R, T.axis, sm.cs[(2 * T.joint) * kBlock + tid], sm.cs[(2 * T.joint + 1) * kBlock + tid]);
    }
    const int np = T.np;
    if (np == 0) continue;
    double sc[3];
    xform(R, p, T.sph[0], T.sph[1], T.sph[2], sc);
    unsigned gb, ge;
    grid_cell(A.grid, sc, live, gb, ge);
    double gcr = 1.0, gsr = 0.0, gcp = 1.0, gsp = 0.0;
    if (S::kGimbal && T.pad) gimbal_cs_fast(R[6], R[7], R[8], gcr, gsr, gcp, gsp);  // (T.pad: some primitive here rides on a gimbal frame)
    float c[3 * kSegPrimMax];
#pragma unroll
    for (int j = 0; j < kSegPrimMax; ++j) {
      c[3 * j] = c[3 * j + 1] = c[3 * j + 2] = 0.0f;
      if (j < np) {
        double pc[3];
        if (S::kGimbal && T.prim_frame[j] != MVK_FRAME_SEG) {  // on the gimbal's roll (or roll + pitch) frame
          double B[9], o[3];
#pragma unroll
          for (int e = 0; e < 9; ++e) B[e] = R[e];
          xform(R, p, T.gimbal_pos[j][0], T.gimbal_pos[j][1], T.gimbal_pos[j][2], o);
          rot_cols<0>(B, gcr, gsr);
          if (T.prim_frame[j] == MVK_FRAME_GIMBAL_PITCH) rot_cols<1>(B, gcp, gsp);
          xform(B, o, T.prim_pos[j][0], T.prim_pos[j][1], T.prim_pos[j][2], pc);
        } else {
          xform(R, p, T.prim_pos[j][0], T.prim_pos[j][1], T.prim_pos[j][2], pc);
        }
        c[3 * j] = static_cast<float>(pc[0]); c[3 * j + 1] = static_cast<float>(pc[1]); c[3 * j + 2] = static_cast<float>(pc[2]);
      }
    }
    if (pending) push_item(pc_, p_gb, p_ge, p_ta, p_tb);
#pragma unroll
    for (int e = 0; e < 3 * kSegPrimMax; ++e) pc_[e] = c[e];
    p_gb = gb; p_ge = ge;
    p_ta = (static_cast<unsigned>(lane) << 21) | (static_cast<unsigned>(k) << 27) | (f32_safe ? 1u << 31 : 0u);
    p_tb = static_cast<unsigned>(np) << 22;
    pending = true;
#ifdef MVK_PROFS
    const unsigned long long tp0 = wall_clock64();
#endif
#pragma unroll 1
    while (qn >= 64u) screen_pop(A, obsf, L, head, qn);
#ifdef MVK_PROFS
    *prof_pop += wall_clock64() - tp0;
#endif
  }
  if (pending) push_item(pc_, p_gb, p_ge, p_ta, p_tb);
#ifdef MVK_PROFS
  const unsigned long long tp1 = wall_clock64();
#endif
#pragma unroll 1
  while (qn > 0u) screen_pop(A, obsf, L, head, qn);
#ifdef MVK_PROFS
  *prof_pop += wall_clock64() - tp1;
#endif
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const unsigned r = __hip_atomic_load(L.res + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  hit = live && (r & kResHit);
  segmask = live ? (r & ~kResHit) : 0u;
}

// level 0 for the fp32 frame walk: the grid's radius is padded by the walk's position error bound (mvk_capi.cpp,
// upload_scene), so a centre that is off by that much still finds every box within the segment radius in its cell
DEV void grid_cell_f(const GridDev& G, const float c[3], bool on, unsigned& gb, unsigned& ge) {
  const float ih = static_cast<float>(G.inv_h);
  const float gx = (c[0] - static_cast<float>(G.origin[0])) * ih, gy = (c[1] - static_cast<float>(G.origin[1])) * ih,
              gz = (c[2] - static_cast<float>(G.origin[2])) * ih;
  const bool inside = on && gx >= 0.0f && gy >= 0.0f && gz >= 0.0f && gx < static_cast<float>(G.n[0]) &&
                      gy < static_cast<float>(G.n[1]) && gz < static_cast<float>(G.n[2]);
  gb = 0; ge = 0;
  if (inside) {
    const unsigned cell = (static_cast<unsigned>(static_cast<int>(gz)) * G.n[1] + static_cast<unsigned>(static_cast<int>(gy))) * G.n[0] +
                          static_cast<unsigned>(static_cast<int>(gx));
    gb = G.start[cell];
    ge = G.start[cell + 1];
  }
}

// collide_screen with the chain frames in fp32 (fp32 variant of the flag path): same ring, same pops; the
// thresholds in L.thr carry f32_delta_fk, which covers the frame walk's position error as well.  Only lanes
// inside the fp32 coordinate range come here (the others are refined in fp64).
template <int ROBOT, int NL>
DEV void collide_screen_f32(const EvalArgs& A, const DevExtras* __restrict__ X, const DevObsF* __restrict__ obsf,
                            const double pos[3], const float Rf[9], const lds_float* csf, const ScreenLds& L, bool live,
                            bool& hit, unsigned& segmask) {
  using S = Shape<ROBOT, NL>;
  const ModelF32& F = X->f32;
  const int tid = threadIdx.x, lane = threadIdx.x & 63;
  float R[9], p[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = Rf[k];
  p[0] = static_cast<float>(pos[0]); p[1] = static_cast<float>(pos[1]); p[2] = static_cast<float>(pos[2]);
  L.res[lane] = 0u;
  unsigned head = 0, qn = 0;  // wave-uniform
  auto push_item = [&](const float* c, unsigned gb, unsigned ge, unsigned tag_a, unsigned tag_b) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const bool push = gb < ge && !(__hip_atomic_load(L.res + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) & kResHit);
    const unsigned long long bal = __ballot(push);
    if (bal) {
      const unsigned to = (head + qn + static_cast<unsigned>(__popcll(bal & ((1ull << lane) - 1ull)))) & (kRing - 1);
      if (push) {
#pragma unroll
        for (int e = 0; e < 3 * kSegPrimMax; ++e) L.ring[e * kRing + to] = __float_as_uint(c[e]);
        L.ring[(3 * kSegPrimMax) * kRing + to] = gb | tag_a;
        L.ring[(3 * kSegPrimMax + 1) * kRing + to] = ge | tag_b;
      }
      qn += static_cast<unsigned>(__popcll(bal));
    }
  };
  float pc_[3 * kSegPrimMax];
  unsigned p_gb = 0, p_ge = 0, p_ta = 0, p_tb = 0;
  bool pending = false;  // wave-uniform
#pragma unroll 1
  for (int k = 0; k < S::kSeg; ++k) {
    const SegTab& T = X->seg_tab[k];
    if (k > 0) {
      float pn[3];
      xform_f(R, p, F.seg_t[k][0], F.seg_t[k][1], F.seg_t[k][2], pn);
      p[0] = pn[0]; p[1] = pn[1]; p[2] = pn[2];
      rot_cols_dyn_f(R, T.axis, csf[(2 * T.joint) * kBlock + tid], csf[(2 * T.joint + 1) * kBlock + tid]);
    }
    const int np = T.np;
    if (np == 0) continue;
    float sc[3];
    xform_f(R, p, X->seg_sph_f[k][0], X->seg_sph_f[k][1], X->seg_sph_f[k][2], sc);
    unsigned gb, ge;
    grid_cell_f(A.grid, sc, live, gb, ge);
    float gcr = 1.0f, gsr = 0.0f, gcp = 1.0f, gsp = 0.0f;
    if (S::kGimbal && T.pad) {
      const float zlx = R[6], zly = R[7], zlz = R[8];
      const float rho2 = fmaf(zly, zly, zlz * zlz);
      const bool deg = !(rho2 > 0.0f);
      const float ir = rsqrt_f32(rho2);
      gcr = deg ? 1.0f : zlz * ir;
      gsr = deg ? 0.0f : -zly * ir;
      const float w = fmaf(-zly, gsr, zlz * gcr);
      const float nn = rsqrt_f32(fmaf(zlx, zlx, w * w));
      gcp = w * nn;
      gsp = zlx * nn;
    }
    float c[3 * kSegPrimMax];
#pragma unroll
    for (int j = 0; j < kSegPrimMax; ++j) {
      c[3 * j] = c[3 * j + 1] = c[3 * j + 2] = 0.0f;
      if (j < np) {
        float pc[3];
        if (S::kGimbal && T.prim_frame[j] != MVK_FRAME_SEG) {
          float B[9], o[3];
#pragma unroll
          for (int e = 0; e < 9; ++e) B[e] = R[e];
          xform_f(R, p, X->seg_gimbal_pos_f[k][j][0], X->seg_gimbal_pos_f[k][j][1], X->seg_gimbal_pos_f[k][j][2], o);
          rot_cols_f<0>(B, gcr, gsr);
          if (T.prim_frame[j] == MVK_FRAME_GIMBAL_PITCH) rot_cols_f<1>(B, gcp, gsp);
          xform_f(B, o, X->seg_prim_pos_f[k][j][0], X->seg_prim_pos_f[k][j][1], X->seg_prim_pos_f[k][j][2], pc);
        } else {
          xform_f(R, p, X->seg_prim_pos_f[k][j][0], X->seg_prim_pos_f[k][j][1], X->seg_prim_pos_f[k][j][2], pc);
        }
        c[3 * j] = pc[0]; c[3 * j + 1] = pc[1]; c[3 * j + 2] = pc[2];
      }
    }
    if (pending) push_item(pc_, p_gb, p_ge, p_ta, p_tb);
#pragma unroll
    for (int e = 0; e < 3 * kSegPrimMax; ++e) pc_[e] = c[e];
    p_gb = gb; p_ge = ge;
    p_ta = (static_cast<unsigned>(lane) << 21) | (static_cast<unsigned>(k) << 27) | (1u << 31);
    p_tb = static_cast<unsigned>(np) << 22;
    pending = true;
#pragma unroll 1
    while (qn >= 64u) screen_pop(A, obsf, L, head, qn);
  }
  if (pending) push_item(pc_, p_gb, p_ge, p_ta, p_tb);
#pragma unroll 1
  while (qn > 0u) screen_pop(A, obsf, L, head, qn);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const unsigned r = __hip_atomic_load(L.res + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  hit = live && (r & kResHit);
  segmask = live ? (r & ~kResHit) : 0u;
}

// ------------------------------------------------------------------------
// flag path, phase 2: exact fp64 tests for the pairs phase 1 left undecided
// ------------------------------------------------------------------------
// exact contact test of primitive q riding on the segment frame (R, p) against one box
template <bool GIMBAL>
DEV bool exact_pair(const mvk_model* __restrict__ M, const DevExtras* __restrict__ X, int q, const double R[9],
                    const double p[3], const DevObs& O) {
  const int type = M->prim_type[q];
  double pc[3], PR[9];
  prim_pose<GIMBAL, true>(M, q, R, p, pc, PR);
  const double rb = X->prim_bound[q], rin = X->prim_inner[q];
  double t[3];
  to_box_frame(O, pc, t);
  const double l2 = point_box_dist2(t, O.h);
  bool h = l2 <= rin * rin;  // a ball inside the primitive already touches the box
#ifdef MVK_PROF
  { const bool nw = !h && type != MVK_PRIM_SPHERE && l2 <= rb * rb; const unsigned long long b = __ballot(nw);
    if (b && (threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) { atomicAdd(&g_prof[11], 1ull); atomicAdd(&g_prof[12], (unsigned long long)__popcll(b)); } }
#endif
  if (!h && type != MVK_PRIM_SPHERE && l2 <= rb * rb)
    h = narrow<false>(type, PR, M->prim_dim[q][0], M->prim_dim[q][1], M->prim_dim[q][2], O, t) <= 0.0;
  return h;
}

// ------------------------------------------------------------------------
// candidate boxes of one primitive for the distance kernels (full_kernel, avoid_kernel)
// ------------------------------------------------------------------------
// Distance grid (mvk_capi.cpp, upload_distance_grid): the cell of a primitive's centre lists every box within
// A.dgrid_radius of it.
// grid_collect_nearest: both steps in one pass over the cell's list, for the avoidance rows (no bound is known
// beforehand there: the primitive's own nearest box sets it).  The threshold follows the running minimum, so the
// list also keeps a few boxes met before the nearest one -- the drain's separating-axis gaps discard them.
DEV bool grid_collect_nearest(const EvalArgs& A, const GridDev& G, double radius, const Smem& sm, bool on, const double pc[3], double rb,
                              int& cnt, unsigned& gb, unsigned& ge, int* e_near = nullptr) {
  const int tid = threadIdx.x;
  const DevObsF* __restrict__ obsf = A.obsf;
  const bool inside = grid_cell_in(G, pc, on, gb, ge);
  const float x = static_cast<float>(pc[0]), y = static_cast<float>(pc[1]), z = static_cast<float>(pc[2]);
  const float rbf = static_cast<float>(rb) * 1.000001f + 2.0f * A.f32_delta;
  float best2f = INFINITY, thr2f = INFINITY;
  int n = 0, n_best = 0;
#pragma unroll 1
  for (unsigned g = gb; g < ge; ++g) {
    if (G.glb) {  // sorted by the box-to-cell lower bound: this group and every later one lie beyond the threshold
      const float lb = G.glb[g];
      if (lb * lb > thr2f) break;
    }
    const ushort4 v = *reinterpret_cast<const ushort4*>(G.items + static_cast<size_t>(g) * kGridGroup);
    const unsigned short ids[kGridGroup] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < kGridGroup; ++e) {
      const float l2 = point_box_dist2_f32(obsf[ids[e] == kGridNone ? static_cast<unsigned>(A.n_obs) : ids[e]], x, y, z);
      if (ids[e] != kGridNone) {
        if (l2 < best2f) {
          best2f = l2;
          n_best = n;  // (it joins the list below: l2 <= its own threshold)
          const float t = (rbf + sqrtf(l2) * 1.000001f) * 1.000001f;
          thr2f = t * t;
        }
        if (l2 <= thr2f) {
          if (n < kListCap) sm.list[n * kBlock + tid] = ids[e];
          ++n;
        }
      }
    }
  }
  const double up = static_cast<double>(sqrtf(best2f)) * 1.000001 + static_cast<double>(A.f32_delta);
  if (!(inside && best2f < INFINITY && rb + up + 2.0 * static_cast<double>(A.f32_delta) <= radius)) return false;  // (+inf: nothing listed)
  cnt = n;
  if (e_near) *e_near = n_best;
  return true;
}

// The sweep over all boxes for the lanes the grid could not answer (`scan`; typically primitives that stick out
// of the obstacles' volume), in fp32 and by the whole wave: every lane holds one box per round and tests it for
// each such lane in turn (centre broadcast from that lane), so a box is loaded once per round however many lanes
// are served, and no load waits on another.  Pass 1: wave minimum of the centre distances -> upper bound of the
// lane's distance; pass 2: the boxes that can undercut min(bound, that) join the lane's candidate list, in
// ascending box order.  The fp32 error bound pads both, as in grid_candidates.
DEV float bcast_lane(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

constexpr int kSweepChunk = 8;  // lanes served per pass over the boxes

DEV void sweep_for_lanes_f32(const EvalArgs& A, const Smem& sm, bool scan, const double pc[3], double rb, double bound, int& cnt) {
  const int tid = threadIdx.x, lane = tid & 63, wave_tid = tid - lane;
  const unsigned long long todo = __ballot(scan);
  if (todo == 0ull) return;
  const DevObsF* __restrict__ obsf = A.obsf;
  const float x = static_cast<float>(pc[0]), y = static_cast<float>(pc[1]), z = static_cast<float>(pc[2]);
  // pass 1, kSweepChunk served lanes per pass over the boxes: their running minima are independent chains, and
  // the wave reductions come once per chunk, not once per box round
  float up2 = INFINITY;
#pragma unroll 1
  for (unsigned long long rem = todo; rem;) {
    int src[kSweepChunk];
    float cx[kSweepChunk], cy[kSweepChunk], cz[kSweepChunk], acc[kSweepChunk];
    int ns = 0;
#pragma unroll
    for (int j = 0; j < kSweepChunk; ++j) {
      if (rem) { src[j] = __ffsll(static_cast<long long>(rem)) - 1; rem &= rem - 1ull; ns = j + 1; }
      else src[j] = src[j > 0 ? j - 1 : 0];  // unused slot (j >= ns, wave-uniform)
      cx[j] = bcast_lane(x, src[j]); cy[j] = bcast_lane(y, src[j]); cz[j] = bcast_lane(z, src[j]);
      acc[j] = INFINITY;
    }
#pragma unroll 1
    for (int b = 0; b < A.n_obs; b += 64) {
      const DevObsF F = obsf[b + lane < A.n_obs ? b + lane : A.n_obs];  // past the end: the padding id's far box
#pragma unroll
      for (int j = 0; j < kSweepChunk; ++j)
        if (j < ns) acc[j] = fminf(acc[j], point_box_dist2_f32(F, cx[j], cy[j], cz[j]));
    }
#pragma unroll
    for (int j = 0; j < kSweepChunk; ++j)
      if (j < ns) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[j] = fminf(acc[j], __shfl_xor(acc[j], o));
        if (lane == src[j]) up2 = acc[j];
      }
  }
  const double up = static_cast<double>(sqrtf(up2)) * 1.000001 + static_cast<double>(A.f32_delta);
  const double thr = rb + fmax(fmin(bound, up), 0.0);
  const float tf = (static_cast<float>(thr) + A.f32_delta) * 1.000001f;
  const float thr2f = tf * tf;
  // pass 2: candidates of each served lane, ascending box order
  int n = 0;
#pragma unroll 1
  for (unsigned long long rem = todo; rem;) {
    int src[kSweepChunk], cn[kSweepChunk];
    float cx[kSweepChunk], cy[kSweepChunk], cz[kSweepChunk], ct[kSweepChunk];
    int ns = 0;
#pragma unroll
    for (int j = 0; j < kSweepChunk; ++j) {
      if (rem) { src[j] = __ffsll(static_cast<long long>(rem)) - 1; rem &= rem - 1ull; ns = j + 1; }
      else src[j] = src[j > 0 ? j - 1 : 0];
      cx[j] = bcast_lane(x, src[j]); cy[j] = bcast_lane(y, src[j]); cz[j] = bcast_lane(z, src[j]);
      ct[j] = bcast_lane(thr2f, src[j]);
      cn[j] = 0;
    }
#pragma unroll 1
    for (int b = 0; b < A.n_obs; b += 64) {
      const bool real = b + lane < A.n_obs;
      const DevObsF F = obsf[real ? b + lane : A.n_obs];
#pragma unroll
      for (int j = 0; j < kSweepChunk; ++j) {
        if (j < ns) {
          const bool pass = real && point_box_dist2_f32(F, cx[j], cy[j], cz[j]) <= ct[j];
          const unsigned long long pm = __ballot(pass);
          if (pass) {
            const int slot = cn[j] + __popcll(pm & ((1ull << lane) - 1ull));
            if (slot < kListCap) sm.list[slot * kBlock + wave_tid + src[j]] = static_cast<unsigned short>(b + lane);
          }
          cn[j] += __popcll(pm);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kSweepChunk; ++j) if (j < ns && lane == src[j]) n = cn[j];
  }
  if (scan) cnt = n;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// The far grid for the lanes the distance grid could not answer (`scan`: those inside its volume, cell groups
// gb .. ge): the same two passes over the cell's list instead of all boxes.  A list is a few dozen boxes, so the wave
// serves one lane at a time, every lane taking one entry per round: the nearest box's centre distance (wave
// minimum), then the entries within the threshold it sets, appended to the served lane's list in list order
// (ascending ids).
DEV void far_for_lanes_f32(const EvalArgs& A, const Smem& sm, bool scan, const double pc[3], double rb, unsigned gb, unsigned ge, int& cnt) {
  const int tid = threadIdx.x, lane = tid & 63, wave_tid = tid - lane;
  unsigned long long todo = __ballot(scan);
  if (todo == 0ull) return;
  const DevObsF* __restrict__ obsf = A.obsf;
  const uint16_t* __restrict__ items = A.fgrid.items;
  const float x = static_cast<float>(pc[0]), y = static_cast<float>(pc[1]), z = static_cast<float>(pc[2]);
  const float rbf = static_cast<float>(rb) * 1.000001f + 2.0f * A.f32_delta;
  int n = 0;
#pragma unroll 1
  while (todo) {
    const int src = __ffsll(static_cast<long long>(todo)) - 1;
    todo &= todo - 1ull;
    const float cx = bcast_lane(x, src), cy = bcast_lane(y, src), cz = bcast_lane(z, src), crb = bcast_lane(rbf, src);
    const unsigned b0 = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(gb), src)) * kGridGroup;
    const unsigned b1 = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(ge), src)) * kGridGroup;
    float acc = INFINITY;
#pragma unroll 1
    for (unsigned e = b0 + static_cast<unsigned>(lane); e < b1; e += 64u) {
      const unsigned id = items[e];
      if (id != static_cast<unsigned>(kGridNone)) acc = fminf(acc, point_box_dist2_f32(obsf[id], cx, cy, cz));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc = fminf(acc, __shfl_xor(acc, o));
    const float t = (crb + sqrtf(acc) * 1.000001f) * 1.000001f;
    const float thr2 = t * t;
    int cn = 0;
#pragma unroll 1
    for (unsigned base = b0; base < b1; base += 64u) {
      const unsigned e = base + static_cast<unsigned>(lane);
      const unsigned id = e < b1 ? items[e] : static_cast<unsigned>(kGridNone);
      const bool pass = id != static_cast<unsigned>(kGridNone) && point_box_dist2_f32(obsf[id], cx, cy, cz) <= thr2;
      const unsigned long long pm = __ballot(pass);
      if (pass) {
        const int slot = cn + __popcll(pm & ((1ull << lane) - 1ull));
        if (slot < kListCap) sm.list[slot * kBlock + wave_tid + src] = static_cast<unsigned short>(id);
      }
      cn += __popcll(pm);
    }
    if (lane == src) n = cn;
  }
  if (scan) cnt = n;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// The same sweep in fp64 for lanes beyond the coordinate range the fp32 bounds are trusted for, one such lane at a time: lane i takes boxes i, i + 64, ...; a wave minimum gives the centre's distance to the
// nearest box (best2, exact fp64), and the boxes that can still undercut min(bound, that distance) are
// appended to the source lane's candidate list in ascending order.  ~2 * n_obs / 64 rounds per lane served
// instead of 2 * n_obs for the whole wave.
DEV void sweep_for_lanes(const EvalArgs& A, const DevObs* __restrict__ obs, const Smem& sm, bool scan, const double pc[3],
                         double rb, double bound, double& best2, int& cnt) {
  const int tid = threadIdx.x, lane = tid & 63, wave_tid = tid - lane;
  unsigned long long todo = __ballot(scan);
  if (todo == 0ull) return;
  auto centre_dist2 = [&](int ob, const double c[3]) {
    double t[3];
    to_box_frame(obs[ob], c, t);
    return point_box_dist2(t, obs[ob].h);
  };
#pragma unroll 1
  while (todo) {
    const int src = __ffsll(static_cast<long long>(todo)) - 1;
    todo &= todo - 1ull;
    const double c[3] = {__shfl(pc[0], src), __shfl(pc[1], src), __shfl(pc[2], src)};
    const double bnd = __shfl(bound, src);
    double b2 = INFINITY;
#pragma unroll 1
    for (int ob = lane; ob < A.n_obs; ob += 64) b2 = fmin(b2, centre_dist2(ob, c));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b2 = fmin(b2, __shfl_xor(b2, o));
    const double thr = rb + fmax(fmin(bnd, sqrt(b2)), 0.0);
    const double thr2 = thr * thr * (1.0 + 1e-12);
    int n = 0;
#pragma unroll 1
    for (int b = 0; b < A.n_obs; b += 64) {
      const int ob = b + lane;
      const bool pass = ob < A.n_obs && centre_dist2(ob, c) <= thr2;
      const unsigned long long pm = __ballot(pass);
      if (pass) {
        const int slot = n + __popcll(pm & ((1ull << lane) - 1ull));
        if (slot < kListCap) sm.list[slot * kBlock + wave_tid + src] = static_cast<unsigned short>(ob);
      }
      n += __popcll(pm);
    }
    if (lane == src) { best2 = b2; cnt = n; }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------
// full path: broadphase shared out over the wave
// ------------------------------------------------------------------------
// A lane's scan of its primitive's cell list runs as long as the longest list among the 64 lanes, and a single lane
// that needs the sweep over all boxes holds the whole wave for it.  As in the flag path's screen, the lanes therefore
// PUSH what they need scanned -- (primitive centre, range of groups of kGridGroup boxes) -- into the wave's ring, and
// the wave POPS the items one per lane: every lane tests one group of boxes for some lane's primitive, delivers the
// result to that lane through LDS and pushes the rest of the range back.  A range is either a cell list of the
// distance grid or ALL boxes of the scene in order (implicit ids), which is how the lanes the grid cannot answer are
// served.  Two kinds of pop: the minimum squared centre distance (first walk: upper bounds), or the boxes within a
// threshold appended to the source lane's candidate list (second walk).
constexpr int kDRingWords = 6;  // centre xyz; first group | source lane | upper slot | all-boxes flag; end group; threshold

template <bool COLLECT>
DEV void dist_pop(const EvalArgs& A, const DevObsF* __restrict__ obsf, const Smem& sm, unsigned& head, unsigned& qn) {
  const int lane = threadIdx.x & 63, wave_tid = threadIdx.x - lane;
  const unsigned m = qn < 64u ? qn : 64u;
  // fewer items than lanes: every item is served by `rep` lanes, each taking another group of its range, so that a
  // lone long range (a lane that needs ALL boxes) is walked 64 groups at a time instead of one
  const unsigned rep = 64u / m;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const unsigned it = static_cast<unsigned>(lane) % m, sub = static_cast<unsigned>(lane) / m;
  const bool active = sub < rep;
  const unsigned at = (head + it) & (kRing - 1);
  unsigned wa = 0, wb = 0, wt = 0;
  float cx = 0.0f, cy = 0.0f, cz = 0.0f;
  if (active) {
    cx = __uint_as_float(sm.dring[0 * kRing + at]);
    cy = __uint_as_float(sm.dring[1 * kRing + at]);
    cz = __uint_as_float(sm.dring[2 * kRing + at]);
    wa = sm.dring[3 * kRing + at];
    wb = sm.dring[4 * kRing + at];
    wt = sm.dring[5 * kRing + at];
  }
  const unsigned g0 = wa & 0x1FFFFFu, src = (wa >> 21) & 63u, slot = (wa >> 27) & 15u;
  // the range's kind: all boxes in order / a cell of the wider second grid / a cell of the far grid / a cell of the grid
  const bool all = wa >> 31, wide = wb >> 31, farg = (wb >> 30) & 1u;
  const unsigned ge = wb & 0x3FFFFFFFu;
  const unsigned g = g0 + sub;
  // a grid with bounds (GridDev::glb): the walk stops at the first group that lies beyond the item's threshold -- the
  // lists are sorted by the box-to-cell lower bound, so every later box is beyond it too
  bool more = active && sub == 0u && g0 + rep < ge;  // the item's owner lane puts the rest of the range back
  float lb_next = 0.0f;
  if (more) {
    const float* __restrict__ glb = all ? nullptr : wide ? A.dgrid2.glb : farg ? nullptr : A.dgrid.glb;
    if (glb) lb_next = glb[g0 + rep];
  }
  if (active && g < ge) {
    unsigned ids[kGridGroup];
    if (all) {
#pragma unroll
      for (int e = 0; e < kGridGroup; ++e) {
        const unsigned id = g * kGridGroup + e;
        ids[e] = id < static_cast<unsigned>(A.n_obs) ? id : static_cast<unsigned>(kGridNone);
      }
    } else {
      const uint2 w = *reinterpret_cast<const uint2*>((wide ? A.dgrid2.items : farg ? A.fgrid.items : A.dgrid.items) + static_cast<size_t>(g) * kGridGroup);
      ids[0] = w.x & 0xFFFFu; ids[1] = w.x >> 16; ids[2] = w.y & 0xFFFFu; ids[3] = w.y >> 16;
    }
    DevObsF F[kGridGroup];
#pragma unroll
    for (int e = 0; e < kGridGroup; ++e) F[e] = obsf[ids[e] == kGridNone ? static_cast<unsigned>(A.n_obs) : ids[e]];
    float l2[kGridGroup];
#pragma unroll
    for (int e = 0; e < kGridGroup; ++e) l2[e] = point_box_dist2_f32(F[e], cx, cy, cz);
    if (COLLECT) {
      const float thr2 = __uint_as_float(wt);
#pragma unroll
      for (int e = 0; e < kGridGroup; ++e)
        if (ids[e] != kGridNone && l2[e] <= thr2) {
          const unsigned at_list = __hip_atomic_fetch_add(sm.dcnt + src, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          if (at_list < static_cast<unsigned>(kListCap)) sm.list[at_list * kBlock + wave_tid + src] = static_cast<unsigned short>(ids[e]);
        }
    } else {
      float mn = INFINITY;
#pragma unroll
      for (int e = 0; e < kGridGroup; ++e) mn = ids[e] != kGridNone ? fminf(mn, l2[e]) : mn;
      // (non-negative floats order like their bit patterns)
      __hip_atomic_fetch_min(reinterpret_cast<lds_uint*>(sm.upper) + slot * kBlock + wave_tid + src, __float_as_uint(mn), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
  if (more) {
    const float lim2 = COLLECT ? __uint_as_float(wt)
                               : __uint_as_float(__hip_atomic_load(reinterpret_cast<lds_uint*>(sm.upper) + slot * kBlock + wave_tid + src, __ATOMIC_RELAXED,
                                                                   __HIP_MEMORY_SCOPE_WAVEFRONT));
    more = lb_next * lb_next <= lim2;
  }
  head = (head + m) & (kRing - 1);
  qn -= m;
  const unsigned long long bal = __ballot(more);
  if (bal) {
    const unsigned to = (head + qn + static_cast<unsigned>(__popcll(bal & ((1ull << lane) - 1ull)))) & (kRing - 1);
    if (more) {
      sm.dring[0 * kRing + to] = __float_as_uint(cx);
      sm.dring[1 * kRing + to] = __float_as_uint(cy);
      sm.dring[2 * kRing + to] = __float_as_uint(cz);
      sm.dring[3 * kRing + to] = wa + rep;  // (first group + rep < end: no carry into the tags)
      sm.dring[4 * kRing + to] = wb;
      sm.dring[5 * kRing + to] = wt;
    }
    qn += static_cast<unsigned>(__popcll(bal));
  }
}

// one item per lane that wants one (wave-uniform head / qn)
// kind of a range: a cell of the distance grid, a cell of the wider second grid, all boxes in order, a cell of the far grid
constexpr unsigned kRangeCell = 0u, kRangeWide = 1u, kRangeAll = 2u, kRangeFar = 3u;
DEV void dist_push(const Smem& sm, bool want, const double pc[3], unsigned gb, unsigned ge, unsigned slot, unsigned kind, float thr2,
                   unsigned head, unsigned& qn) {
  const int lane = threadIdx.x & 63;
  const bool push = want && gb < ge;
  const unsigned long long bal = __ballot(push);
  if (bal) {
    const unsigned to = (head + qn + static_cast<unsigned>(__popcll(bal & ((1ull << lane) - 1ull)))) & (kRing - 1);
    if (push) {
      sm.dring[0 * kRing + to] = __float_as_uint(static_cast<float>(pc[0]));
      sm.dring[1 * kRing + to] = __float_as_uint(static_cast<float>(pc[1]));
      sm.dring[2 * kRing + to] = __float_as_uint(static_cast<float>(pc[2]));
      sm.dring[3 * kRing + to] = gb | (static_cast<unsigned>(lane) << 21) | (slot << 27) | (kind == kRangeAll ? 1u << 31 : 0u);
      sm.dring[4 * kRing + to] = ge | (kind == kRangeWide ? 1u << 31 : 0u) | (kind == kRangeFar ? 1u << 30 : 0u);
      sm.dring[5 * kRing + to] = __float_as_uint(thr2);
    }
    qn += static_cast<unsigned>(__popcll(bal));
  }
}

// doubles as unsigned keys that order like the values (negative ones included), for LDS atomic minima
DEV unsigned long long ord_key(double x) {
  const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(x));
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
DEV double ord_val(unsigned long long k) {
  return __longlong_as_double(static_cast<long long>((k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k));
}

// ------------------------------------------------------------------------
// full path: exact minimum distance over every (primitive, box) pair
// ------------------------------------------------------------------------
// F32N: the fp32 variant of the distance output -- the flat drains (every pair of a scene with a distance grid, bar the
// lanes whose list overflowed) take the box from the fp32 copy and run the separating-axis gap, the edge-to-box and the
// segment-to-box distances in fp32; frames, broadphase bounds and the merge of the minima are as in the fp64 form.
template <int ROBOT, int NL, bool F32N = false>
DEV double collide_full(const EvalArgs& A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
                        const DevObs* __restrict__ obs, const double pos[3], const double Rw[9], const Smem& sm,
                        bool live) {
  using S = Shape<ROBOT, NL>;
  const int tid = threadIdx.x, lane = tid & 63;
  double R[9], p[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = Rw[k];
  p[0] = pos[0]; p[1] = pos[1]; p[2] = pos[2];
  double mind = INFINITY;
  const bool use_grid = A.dgrid.start != nullptr;
  const bool grouped = use_grid && X->n_grp > 0;  // (the sweep over all boxes of small scenes keeps one primitive at a time)
  const bool f32_safe = fmax(fabs(pos[0]), fmax(fabs(pos[1]), fabs(pos[2]))) + X->reach <= A.f32_range;
  // with the distance grid, a first walk over the chain bounds the state's distance from above: the smallest
  // centre-to-listed-box distance over all primitives.  The second walk then collects, per primitive, only the
  // boxes that can undercut that bound, and a lane needs the sweep over all boxes only when the whole robot
  // is farther from every box than the grid's slack.
  double upper = INFINITY;
  PROF_ACC_DECL;
  PROF_T0(t_w1);
  const DevObsF* __restrict__ obsf = A.obsf;
  const double delta = static_cast<double>(A.f32_delta), dd = 2.0 * delta;
  const unsigned groups_all = static_cast<unsigned>((A.n_obs + kGridGroup - 1) / kGridGroup);
  const bool have_wide = A.dgrid2.start != nullptr && A.dgrid2_radius > A.dgrid_radius;
  lds_uint* up_bits = reinterpret_cast<lds_uint*>(sm.upper);
  unsigned head = 0, qn = 0;  // the wave's ring (wave-uniform)
  if (use_grid) {
    const bool on = live && f32_safe;
#pragma unroll
    for (int sl = 0; sl < kUpperSlots; ++sl) up_bits[sl * kBlock + tid] = 0x7F800000u;  // +inf
    // first walk: per primitive the smallest centre distance over its cell's list (fp32), through the ring
    auto upper_walk = [&](bool all_boxes, bool who) {  // (all_boxes: kept for experiments; the walk below uses the lists)
#pragma unroll
      for (int k = 0; k < 9; ++k) R[k] = Rw[k];
      p[0] = pos[0]; p[1] = pos[1]; p[2] = pos[2];
#pragma unroll 1
      for (int k = 0; k < S::kSeg; ++k) {
        if (k > 0) advance_frame(M, sm, tid, k, R, p);
        const int gfirst = grouped ? X->seg_grp_begin[k] : X->seg_prim_begin[k], glast = grouped ? X->seg_grp_begin[k + 1] : X->seg_prim_begin[k + 1];
#pragma unroll 1
        for (int g = gfirst; g < glast; ++g) {  // the group's leader stands for its members (its centre lies inside it)
          const int q = X->prim_order[grouped ? X->grp_lead[g] : g];
          double pc[3];
          prim_pose<S::kGimbal, false>(M, q, R, p, pc, nullptr);
          unsigned gb = 0, ge = groups_all;
          if (!all_boxes) grid_cell(A.dgrid, pc, who, gb, ge);
          dist_push(sm, who, pc, gb, ge, static_cast<unsigned>(g < kUpperSlots ? g : kUpperSlots - 1), all_boxes ? kRangeAll : kRangeCell, 0.0f, head, qn);
#pragma unroll 1
          while (qn >= 64u) dist_pop<false>(A, obsf, sm, head, qn);
        }
      }
#pragma unroll 1
      while (qn > 0u) dist_pop<false>(A, obsf, sm, head, qn);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float u2 = INFINITY;
#pragma unroll
      for (int sl = 0; sl < kUpperSlots; ++sl) u2 = fminf(u2, __uint_as_float(up_bits[sl * kBlock + tid]));
      return static_cast<double>(sqrtf(u2)) * 1.000001 + delta;  // an upper bound of the state's distance (+inf: nothing listed)
    };
    upper = upper_walk(false, on);
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = Rw[k];
    p[0] = pos[0]; p[1] = pos[1]; p[2] = pos[2];
  }
  PROF_CLK(0, t_w1);
#pragma unroll 1
  for (int k = 0; k < S::kSeg; ++k) {
    if (k > 0) advance_frame(M, sm, tid, k, R, p);
    const int gfirst = grouped ? X->seg_grp_begin[k] : X->seg_prim_begin[k], glast = grouped ? X->seg_grp_begin[k + 1] : X->seg_prim_begin[k + 1];
#pragma unroll 1
    for (int g = gfirst; g < glast; ++g) {
      // One broadphase per GROUP of co-located primitives (a link's bar and its propeller protector; a link's pipe and
      // its two gimbal modules): the candidates are collected about the leader's centre with a radius that contains
      // every member (DevExtras::grp_rb), then every member runs its narrowphase over that one list.
      const int qb = grouped ? X->grp_begin[g] : g, qe = grouped ? X->grp_begin[g + 1] : g + 1;
      const bool single = qe - qb == 1;
      double pc[3];
      prim_pose<S::kGimbal, false>(M, X->prim_order[grouped ? X->grp_lead[g] : g], R, p, pc, nullptr);
      const double rb = grouped ? X->grp_rb[g] : X->prim_bound[X->prim_order[qb]];
      const int qi = g;  // (the slot of the first walk's bound)

      // -- broadphase --------------------------------------------------------
      PROF_T0(t_bp);
      const bool sphere = single && M->prim_type[X->prim_order[qb]] == MVK_PRIM_SPHERE;
      const double d0_lead = M->prim_dim[X->prim_order[qb]][0];
      int cnt = 0;
      bool scan = live;  // lanes whose candidates come from the sweep over all boxes
      bool swept = false;
      unsigned cell_b = 0, cell_e = 0;  // the grid cell's groups, when the candidates came from it
      if (use_grid) {
        const double bound = fmin(mind, upper);
        // the first walk's minimum for this primitive: when even its nearest box (of a list that is complete for the
        // threshold) is too far to undercut the bound -- lower bound: centre distance - bounding radius -- there is
        // nothing to collect: most primitives of a state, since only the few nearest matter
        const bool on = live && f32_safe;
        double thr0 = rb + fmax(bound, 0.0);
        lds_uint* up_slot = up_bits + (qi < kUpperSlots ? qi : kUpperSlots - 1) * kBlock + tid;
        double up_q = static_cast<double>(sqrtf(__uint_as_float(*up_slot))) * 0.999999;
        // the cell's list holds every box within thr0 only when thr0 stays inside the grid's radius; a lane beyond
        // that (the bound still loose, the robot far from every box, or outside the grid's volume) takes ALL boxes as
        // its range: first their smallest centre distance, which tightens the bound, then the collection
        const bool far = on && !(thr0 + dd <= A.dgrid_radius);
        // ... unless the wider second grid (the avoidance rows' one) can answer: its cell's list is a few dozen boxes
        unsigned far_b = 0u, far_e = groups_all, far_kind = kRangeAll;
        if (__any(far)) {
          if (far && have_wide && thr0 + dd <= A.dgrid2_radius) {
            grid_cell(A.dgrid2, pc, true, far_b, far_e);
            far_kind = kRangeWide;
          } else if (far) {  // ... or the far grid, whose cell lists the boxes that can be nearest at any distance
            unsigned fb, fe;
            if (grid_cell_in(A.fgrid, pc, true, fb, fe)) { far_b = fb; far_e = fe; far_kind = kRangeFar; }
          }
          if (far) *up_slot = 0x7F800000u;
          dist_push(sm, far, pc, far_b, far_e, static_cast<unsigned>(qi < kUpperSlots ? qi : kUpperSlots - 1), far_kind, 0.0f, head, qn);
#pragma unroll 1
          while (qn > 0u) dist_pop<false>(A, obsf, sm, head, qn);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (far) {
            const double est = static_cast<double>(sqrtf(__uint_as_float(*up_slot)));
            up_q = est * 0.999999;
            upper = fmin(upper, est * 1.000001 + delta);
            thr0 = rb + fmax(fmin(mind, upper), 0.0);
          }
        }
        const bool none = (qi < kUpperSlots || far) && on && up_q - delta > thr0 + dd;
        const bool want = on && !none;
        const float tf = (static_cast<float>(thr0) + A.f32_delta) * 1.000001f;
        sm.dcnt[lane] = 0u;
        if (!far) grid_cell(A.dgrid, pc, want, cell_b, cell_e);
        dist_push(sm, want, pc, far ? far_b : cell_b, far ? far_e : cell_e, 0u, far ? far_kind : kRangeCell, tf * tf, head, qn);
#pragma unroll 1
        while (qn > 0u) dist_pop<true>(A, obsf, sm, head, qn);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        cnt = want ? static_cast<int>(sm.dcnt[lane]) : 0;
        scan = on && far;  // (a list that overflows is then replaced by all boxes, not by the cell's list)
        const bool slow = live && !f32_safe;  // beyond the fp32 coordinate range: the same in fp64, one lane at a time
        double b2 = INFINITY;
        int c2 = 0;
        sweep_for_lanes(A, obs, sm, slow, pc, rb, mind, b2, c2);
        if (slow) {
          scan = true;
          if (sphere) mind = fmin(mind, sqrt(b2) - d0_lead);  // exact for a sphere (a group of one)
          else cnt = c2;
        }
      } else if (__any(scan)) {
        swept = true;
        // sweep over all boxes: centre-to-box distance vs bounding radius; obstacle boxes are read at
        // wave-uniform addresses (scalar loads -> SGPRs)
        auto centre_dist2 = [&](const DevObs& O) {
          double t[3];
          to_box_frame(O, pc, t);
          return point_box_dist2(t, O.h);
        };
        // pass 1: the primitive's origin lies inside it, so min_ob dist(origin, box)
        // bounds this primitive's separation from above
        double best2 = INFINITY;
#pragma unroll 2
        for (int ob = 0; ob < A.n_obs; ++ob) best2 = fmin(best2, centre_dist2(obs[ob]));
        if (sphere) {
          mind = fmin(mind, sqrt(best2) - d0_lead);  // exact for a sphere
        } else {
          // pass 2: keep pairs whose lower bound (centre distance - bounding radius)
          // can still undercut the running minimum, and every possibly-touching pair
          const double thr = rb + fmax(fmin(mind, sqrt(best2)), 0.0);
          const double thr2 = thr * thr;
#pragma unroll 2
          for (int ob = 0; ob < A.n_obs; ++ob) {
            if (scan && centre_dist2(obs[ob]) <= thr2) {
              if (cnt < kListCap) sm.list[cnt * kBlock + tid] = static_cast<unsigned short>(ob);
              ++cnt;
            }
          }
        }
      }
      PROF_CLK(1, t_bp);
      const int cnt_g = cnt;
      const bool scan_g = scan;
#pragma unroll 1
      for (int qm = qb; qm < qe; ++qm) {
      const int q = X->prim_order[qm];
      const int type = M->prim_type[q];
      double PR[9];
      prim_pose<S::kGimbal, true>(M, q, R, p, pc, PR);  // (pc: this member's own centre from here on)
      const double d0 = M->prim_dim[q][0], d1 = M->prim_dim[q][1], d2 = M->prim_dim[q][2];
      cnt = cnt_g;
      scan = scan_g;
      // a member of a group of several walks the group's list, which reaches farther than its own would: a candidate
      // whose centre distance minus the member's own bounding radius cannot undercut the running minimum is dropped
      // before the separating-axis bound (it could not lower the minimum, so the result keeps its bits)
      const double own = single ? INFINITY : X->prim_bound[q];
      // -- narrowphase drain: kind is wave-uniform, list length is per lane ---
      // a lane with more candidates than its list holds walks its grid cell's whole list instead (complete by
      // construction), or every box of the scene when its candidates came from a sweep
      const bool overflow = cnt > kListCap;
      const bool from_cell = overflow && use_grid && !scan;
      int todo = live ? (overflow ? (from_cell ? static_cast<int>(cell_e - cell_b) * kGridGroup : A.n_obs) : cnt) : 0;
      if (sphere && swept) todo = 0;
      auto cand = [&](int e) -> int {  // kGridNone: padding, skip
        return !overflow ? sm.list[e * kBlock + tid] : (from_cell ? A.dgrid.items[static_cast<size_t>(cell_b) * kGridGroup + e] : e);
      };
#ifdef MVK_PROF_COUNTS
      PROF_LANES(30, live); PROF_LANES(31, live && scan); PROF_LANES(32, live && overflow); PROF_SUM(33, live ? cnt : 0);
#endif
      PROF_T0(t_dr);
      if (type == MVK_PRIM_BOX) {
        // The exact box-box separation (24 edge-to-box distances) costs ~30x the separating-axis bound, and
        // only the few candidates whose bound undercuts the running minimum need it.  Lanes therefore scan
        // their lists with the bound alone, each stopping at its next candidate that does need the exact
        // distance; the wave then computes those together (below).
        const double a[3] = {d0, d1, d2};
        // Flat drain (no lane's list overflowed, the common case): the lists of a primitive step have very different
        // lengths -- two candidates on average, ten in the longest of 64 -- so instead of every lane scanning its
        // own, the wave numbers the (lane, candidate) pairs through and takes them 64 at a time: a lane fetches the
        // owner's pose and running minimum (cross-lane reads), computes the gap, and merges contacts and exact
        // distances into the owner's minimum with LDS atomics on ordered keys.  Per round the pair with the smallest
        // gap of each owner gets its exact distance first; the others are re-tested against the result.
        const bool flat = use_grid && !__any(live && overflow);
        if (flat) {
          const int wave_tid = tid - lane;
          unsigned incl = static_cast<unsigned>(todo);
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
          }
          const unsigned T = __shfl(incl, 63);
          sm.dcnt[lane] = incl;
          sm.mres[lane] = ord_key(mind);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          // the exact distances of the pairs `mine`, kExactBatch at a time over the whole wave (see below), merged
          // into the owner's minimum
          auto run_exact = [&](bool want, int src, const double Rr[9], const double t[3], const double h[3]) {
            unsigned long long pend = __ballot(want);
            PROF_T0(t_ex);
#ifdef MVK_PROF_COUNTS
            PROF_SUM(34, want ? 1 : 0);
#endif
#pragma unroll 1
            while (pend) {
#ifdef MVK_PROF_COUNTS
              PROF_WAVE(35);
#endif
              const int rank = __popcll(pend & ((1ull << lane) - 1ull));
              const bool mine = want && ((pend >> lane) & 1ull) && rank < kExactBatch;
              const int E = min(static_cast<int>(__popcll(pend)), kExactBatch);
              if (mine) {
#pragma unroll
                for (int c = 0; c < 9; ++c) sm.ex_item[rank * kExactItem + c] = Rr[c];
#pragma unroll
                for (int c = 0; c < 3; ++c) { sm.ex_item[rank * kExactItem + 9 + c] = t[c]; sm.ex_item[rank * kExactItem + 12 + c] = h[c]; }
                sm.ex_res[rank] = 0x7FF0000000000000ull;  // +inf
              }
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
              __builtin_amdgcn_wave_barrier();
#pragma unroll 1
              for (int it = lane; it < 24 * E; it += 64) {
                const int pr = it / 24, ed = it - 24 * pr;
                double iR[9], it_[3], ih[3];
#pragma unroll
                for (int c = 0; c < 9; ++c) iR[c] = sm.ex_item[pr * kExactItem + c];
#pragma unroll
                for (int c = 0; c < 3; ++c) { it_[c] = sm.ex_item[pr * kExactItem + 9 + c]; ih[c] = sm.ex_item[pr * kExactItem + 12 + c]; }
                const bool second = ed >= 12;
                double s2;
                if constexpr (F32N) {
                  float fR[9], ft[3], fh[3], fa[3], cR[9], cc[3], ca[3], cb[3];
#pragma unroll
                  for (int c = 0; c < 9; ++c) fR[c] = static_cast<float>(iR[c]);
#pragma unroll
                  for (int c = 0; c < 3; ++c) { ft[c] = static_cast<float>(it_[c]); fh[c] = static_cast<float>(ih[c]); fa[c] = static_cast<float>(a[c]); }
#pragma unroll
                  for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) cR[3 * r + c] = second ? fR[3 * c + r] : fR[3 * r + c];
                    const float tb = -fmaf(fR[r], ft[0], fmaf(fR[3 + r], ft[1], fR[6 + r] * ft[2]));
                    cc[r] = second ? tb : ft[r];
                    ca[r] = second ? fh[r] : fa[r];
                    cb[r] = second ? fa[r] : fh[r];
                  }
                  s2 = static_cast<double>(box_edge_to_box2_f(second ? ed - 12 : ed, cc, cR, ca, cb));
                } else {
                  double cR[9], cc[3], ca[3], cb[3];
#pragma unroll
                  for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) cR[3 * r + c] = second ? iR[3 * c + r] : iR[3 * r + c];
                    const double tb = -fma(iR[r], it_[0], fma(iR[3 + r], it_[1], iR[6 + r] * it_[2]));
                    cc[r] = second ? tb : it_[r];
                    ca[r] = second ? ih[r] : a[r];
                    cb[r] = second ? a[r] : ih[r];
                  }
                  s2 = box_edge_to_box2(second ? ed - 12 : ed, cc, cR, ca, cb);
                }
                __hip_atomic_fetch_min(&sm.ex_res[pr], static_cast<unsigned long long>(__double_as_longlong(s2)), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WAVEFRONT);
              }
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
              __builtin_amdgcn_wave_barrier();
              if (mine)
                __hip_atomic_fetch_min(&sm.mres[src], ord_key(F32N ? static_cast<double>(sqrtf(static_cast<float>(__longlong_as_double(static_cast<long long>(sm.ex_res[rank])))))
                                                                    : sqrt(__longlong_as_double(static_cast<long long>(sm.ex_res[rank])))),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
              pend &= ~__ballot(mine);
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
              __builtin_amdgcn_wave_barrier();
            }
            PROF_CLK(3, t_ex);
          };
#pragma unroll 1
          for (unsigned base = 0; base < T; base += 64u) {
            const unsigned i = base + static_cast<unsigned>(lane);
            const bool act = i < T;
#ifdef MVK_PROF_COUNTS
            PROF_WAVE(26);
#endif
            unsigned lo = 0u, hi = 63u;  // the owner: the first lane whose inclusive count exceeds i
#pragma unroll
            for (int st = 0; st < 6; ++st) {
              const unsigned mid = (lo + hi) >> 1;
              if (sm.dcnt[mid] > i) hi = mid; else lo = mid + 1u;
            }
            const int src = act ? static_cast<int>(lo) : lane;
            const unsigned e = i - (src > 0 ? sm.dcnt[src - 1] : 0u);
            double spc[3], sPR[9];
#pragma unroll
            for (int c = 0; c < 3; ++c) spc[c] = __shfl(pc[c], src);
#pragma unroll
            for (int c = 0; c < 9; ++c) sPR[c] = __shfl(PR[c], src);
            bool need = false;
            double Rr[9], t[3], h[3] = {0.0, 0.0, 0.0}, g = INFINITY;
            sm.gmin[lane] = ~0ull;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (act) {
              const int ob = sm.list[e * kBlock + wave_tid + src];
              const double cut = ord_val(sm.mres[src]) + (F32N ? kSatSlackF32 : kSatSlack);
              const double reach_m = own + fmax(cut, 0.0);
              if constexpr (F32N) {
                const DevObsF Of = A.obsf[ob];
                const float dx = static_cast<float>(spc[0]) - Of.c[0], dy = static_cast<float>(spc[1]) - Of.c[1], dz = static_cast<float>(spc[2]) - Of.c[2];
                float tf[3], Rf[9], fP[9];
#pragma unroll
                for (int c = 0; c < 3; ++c) tf[c] = fmaf(Of.R[c], dx, fmaf(Of.R[3 + c], dy, Of.R[6 + c] * dz));
                const float rm = static_cast<float>(reach_m) * 1.00001f + 1e-6f;
                const float c2 = fmaf(excess_abs_f(tf[0], Of.h[0]), excess_abs_f(tf[0], Of.h[0]),
                                      fmaf(excess_abs_f(tf[1], Of.h[1]), excess_abs_f(tf[1], Of.h[1]), excess_abs_f(tf[2], Of.h[2]) * excess_abs_f(tf[2], Of.h[2])));
                if (single || c2 <= rm * rm) {
#pragma unroll
                  for (int c = 0; c < 9; ++c) fP[c] = static_cast<float>(sPR[c]);
#pragma unroll
                  for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) Rf[3 * r + c] = fmaf(Of.R[r], fP[c], fmaf(Of.R[3 + r], fP[3 + c], Of.R[6 + r] * fP[6 + c]));
                  const float af[3] = {static_cast<float>(a[0]), static_cast<float>(a[1]), static_cast<float>(a[2])};
                  bool recheck;
                  g = static_cast<double>(box_box_sat_f(tf, Rf, af, Of.h, static_cast<float>(cut), recheck));
#pragma unroll
                  for (int c = 0; c < 9; ++c) Rr[c] = static_cast<double>(Rf[c]);
#pragma unroll
                  for (int c = 0; c < 3; ++c) { t[c] = static_cast<double>(tf[c]); h[c] = static_cast<double>(Of.h[c]); }
                  if (recheck) {  // (a contact between nearly parallel edges: rare, and its depth needs the axis fp32 cannot resolve)
                    const DevObs& O = obs[ob];
                    double t64[3], R64[9];
                    to_box_frame(O, spc, t64);
                    box_rel(O, sPR, R64);
                    g = box_box_sat<true>(t64, R64, a, O.h, cut);
                  }
                  if (g <= 0.0) __hip_atomic_fetch_min(&sm.mres[src], ord_key(g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                  else need = g < cut;
                }
              } else {
              const DevObs& O = obs[ob];
              to_box_frame(O, spc, t);
              h[0] = O.h[0]; h[1] = O.h[1]; h[2] = O.h[2];
              if (single || point_box_dist2(t, h) <= reach_m * reach_m) {
                box_rel(O, sPR, Rr);
                g = box_box_sat<true>(t, Rr, a, h, cut);
#ifdef MVK_PROF_COUNTS
                atomicAdd(&g_prof[27], 1ull);
#endif
                if (g <= 0.0) __hip_atomic_fetch_min(&sm.mres[src], ord_key(g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                else need = g < cut;
              }
              }
              if (need) __hip_atomic_fetch_min(&sm.gmin[src], static_cast<unsigned long long>(__double_as_longlong(g)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool first = need && static_cast<unsigned long long>(__double_as_longlong(g)) == sm.gmin[src];
            run_exact(first, src, Rr, t, h);
            const bool again = need && !first && g < ord_val(sm.mres[src]) + (F32N ? kSatSlackF32 : kSatSlack);
            run_exact(again, src, Rr, t, h);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          mind = ord_val(sm.mres[lane]);
        }
        int e = flat ? todo : 0;
#pragma unroll 1
        while (__any(e < todo)) {
          bool need = false;
          double Rr[9], t[3], h[3];
#pragma unroll 1
          while (e < todo && !need) {
            const int ob = cand(e);
            if (ob != kGridNone) {
              const DevObs& O = obs[ob];
              to_box_frame(O, pc, t);
              h[0] = O.h[0]; h[1] = O.h[1]; h[2] = O.h[2];
              const double reach_m = own + fmax(mind + kSatSlack, 0.0);
              if (single || point_box_dist2(t, h) <= reach_m * reach_m) {
                box_rel(O, PR, Rr);
                const double g = box_box_sat<true>(t, Rr, a, h, mind + kSatSlack);
                if (g <= 0.0) mind = fmin(mind, g);
                else need = g < mind + kSatSlack;
              }
            }
            ++e;
          }
          // the exact distances, kExactBatch pairs at a time over the whole wave: the pairs go to LDS and every
          // lane takes (pair, edge) items -- 24 edge-to-box distances per pair, 12 edges of either box against
          // the other -- merging into the pair's minimum with an LDS atomic.  A wave with E pairs pending runs
          // ceil(24 E / 64) edge distances per lane instead of 24.
          unsigned long long pend = __ballot(need);
          PROF_T0(t_ex);
#ifdef MVK_PROF_COUNTS
          PROF_SUM(34, need ? 1 : 0);
          if (pend) PROF_WAVE(35);
#endif
#pragma unroll 1
          while (pend) {
            const int rank = __popcll(pend & ((1ull << lane) - 1ull));
            const bool mine = need && ((pend >> lane) & 1ull) && rank < kExactBatch;
            const int E = min(static_cast<int>(__popcll(pend)), kExactBatch);
            if (mine) {
#pragma unroll
              for (int c = 0; c < 9; ++c) sm.ex_item[rank * kExactItem + c] = Rr[c];
#pragma unroll
              for (int c = 0; c < 3; ++c) { sm.ex_item[rank * kExactItem + 9 + c] = t[c]; sm.ex_item[rank * kExactItem + 12 + c] = h[c]; }
              sm.ex_res[rank] = 0x7FF0000000000000ull;  // +inf
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll 1
            for (int it = lane; it < 24 * E; it += 64) {
              const int pr = it / 24, ed = it - 24 * pr;
              double iR[9], it_[3], ih[3];
#pragma unroll
              for (int c = 0; c < 9; ++c) iR[c] = sm.ex_item[pr * kExactItem + c];
#pragma unroll
              for (int c = 0; c < 3; ++c) { it_[c] = sm.ex_item[pr * kExactItem + 9 + c]; ih[c] = sm.ex_item[pr * kExactItem + 12 + c]; }
              // edges 0..11: the primitive's edges against the box (box frame); 12..23: the box's edges against
              // the primitive (primitive frame: rotation transposed, centre -R^T t), as box_box_exact
              const bool second = ed >= 12;
              double cR[9], cc[3], ca[3], cb[3];
#pragma unroll
              for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) cR[3 * r + c] = second ? iR[3 * c + r] : iR[3 * r + c];
                const double tb = -fma(iR[r], it_[0], fma(iR[3 + r], it_[1], iR[6 + r] * it_[2]));
                cc[r] = second ? tb : it_[r];
                ca[r] = second ? ih[r] : a[r];
                cb[r] = second ? a[r] : ih[r];
              }
              const double s2 = box_edge_to_box2(second ? ed - 12 : ed, cc, cR, ca, cb);
              __hip_atomic_fetch_min(&sm.ex_res[pr], static_cast<unsigned long long>(__double_as_longlong(s2)), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (mine) mind = fmin(mind, sqrt(__longlong_as_double(static_cast<long long>(sm.ex_res[rank]))));
            pend &= ~__ballot(mine);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
          PROF_CLK(3, t_ex);
        }
      } else {
        // Closed-form primitives (sphere, capsule): the same flat drain without the exact stage -- the wave numbers
        // the (lane, candidate) pairs through and takes them 64 at a time instead of every lane walking its own list
        // for as many rounds as the longest of 64 has entries; a distance merges into its owner's minimum with an LDS
        // atomic on the ordered key (the minimum of the same values: the same bits).
        const bool flat = use_grid && !__any(live && overflow);
        if (flat) {
          const int wave_tid = tid - lane;
          unsigned incl = static_cast<unsigned>(todo);
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
          }
          const unsigned T = __shfl(incl, 63);
          sm.dcnt[lane] = incl;
          sm.mres[lane] = ord_key(mind);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll 1
          for (unsigned base = 0; base < T; base += 64u) {
            const unsigned i = base + static_cast<unsigned>(lane);
            const bool act = i < T;
            unsigned lo = 0u, hi = 63u;  // the owner: the first lane whose inclusive count exceeds i
#pragma unroll
            for (int st = 0; st < 6; ++st) {
              const unsigned mid = (lo + hi) >> 1;
              if (sm.dcnt[mid] > i) hi = mid; else lo = mid + 1u;
            }
            const int src = act ? static_cast<int>(lo) : lane;
            const unsigned e = i - (src > 0 ? sm.dcnt[src - 1] : 0u);
            double spc[3], sPR[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int c = 0; c < 3; ++c) spc[c] = __shfl(pc[c], src);
#pragma unroll
            for (int c = 0; c < 3; ++c) sPR[3 * c + 2] = __shfl(PR[3 * c + 2], src);  // (a capsule's axis: all narrow() reads)
            if (act) {
              const int ob = sm.list[e * kBlock + wave_tid + src];
              double d = INFINITY;
              if constexpr (F32N) {
                const DevObsF Of = A.obsf[ob];
                const float dx = static_cast<float>(spc[0]) - Of.c[0], dy = static_cast<float>(spc[1]) - Of.c[1], dz = static_cast<float>(spc[2]) - Of.c[2];
                float tf[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) tf[c] = fmaf(Of.R[c], dx, fmaf(Of.R[3 + c], dy, Of.R[6 + c] * dz));
                const float c2 = fmaf(excess_abs_f(tf[0], Of.h[0]), excess_abs_f(tf[0], Of.h[0]),
                                      fmaf(excess_abs_f(tf[1], Of.h[1]), excess_abs_f(tf[1], Of.h[1]), excess_abs_f(tf[2], Of.h[2]) * excess_abs_f(tf[2], Of.h[2])));
                if (type == MVK_PRIM_SPHERE) {
                  d = static_cast<double>(sqrtf(c2) - static_cast<float>(d0));
                } else {
                  const float rm = static_cast<float>(own + fmax(ord_val(sm.mres[src]) + kSatSlackF32, 0.0)) * 1.00001f + 1e-6f;
                  if (single || c2 <= rm * rm) {  // a capsule: its axis (third column of the pose) into the box frame
                    const float ax[3] = {static_cast<float>(sPR[2]), static_cast<float>(sPR[5]), static_cast<float>(sPR[8])};
                    const float hl = static_cast<float>(d1);
                    float av[3], p0[3], dd[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) av[c] = fmaf(Of.R[c], ax[0], fmaf(Of.R[3 + c], ax[1], Of.R[6 + c] * ax[2]));
#pragma unroll
                    for (int c = 0; c < 3; ++c) { p0[c] = fmaf(-hl, av[c], tf[c]); dd[c] = 2.0f * hl * av[c]; }
                    d = static_cast<double>(sqrtf(seg_box_dist2_f(p0, dd, Of.h)) - static_cast<float>(d0));
                  }
                }
              } else {
              const DevObs& O = obs[ob];
              double t[3];
              to_box_frame(O, spc, t);
              const double h[3] = {O.h[0], O.h[1], O.h[2]};
              const double c2 = point_box_dist2(t, h);
              if (type == MVK_PRIM_SPHERE) {
                d = sqrt(c2) - d0;
              } else {
                const double reach_m = own + fmax(ord_val(sm.mres[src]) + kSatSlack, 0.0);
                if (single || c2 <= reach_m * reach_m) d = narrow<true>(type, sPR, d0, d1, d2, O, t, INFINITY);
              }
              }
              __hip_atomic_fetch_min(&sm.mres[src], ord_key(d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          mind = ord_val(sm.mres[lane]);
        }
#pragma unroll 1
        for (int e = flat ? todo : 0; __any(e < todo); ++e) {
          const int ob = e < todo ? cand(e) : kGridNone;
          if (ob != kGridNone) {
            const DevObs O = obs[ob];
            double t[3];
            to_box_frame(O, pc, t);
            if (type == MVK_PRIM_SPHERE) {
              mind = fmin(mind, sqrt(point_box_dist2(t, O.h)) - d0);
            } else {
              const double reach_m = own + fmax(mind + kSatSlack, 0.0);
              if (single || point_box_dist2(t, O.h) <= reach_m * reach_m) mind = fmin(mind, narrow<true>(type, PR, d0, d1, d2, O, t, mind));
            }
          }
        }
      }
      PROF_CLK(2, t_dr);
      }  // members
    }
  }
  PROF_ACC_FLUSH(20);
  return mind;
}


// exact separation of two disjoint boxes with the closest points (A: centre t, axes Rr, half extents a, in the
// frame of B = [-h,h]^3): pr on A, pe on B, both in B's frame
DEV double box_box_exact_cp(const double t[3], const double Rr[9], const double a[3], const double h[3], double pr[3], double pe[3]) {
  double pa1[3], pb1[3], pa2[3], pb2[3];
  const double s1 = box_edges_to_box_cp2(t, Rr, a, h, pa1, pb1);  // pa1 on the robot, pb1 on the box
  double Rt[9], tb[3];  // the box in the primitive's frame
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Rt[3 * r + c] = Rr[3 * c + r];
#pragma unroll
  for (int r = 0; r < 3; ++r) tb[r] = -fma(Rr[r], t[0], fma(Rr[3 + r], t[1], Rr[6 + r] * t[2]));
  const double s2 = box_edges_to_box_cp2(tb, Rt, h, a, pa2, pb2);  // pa2 on the box, pb2 on the robot (primitive frame)
  if (s1 <= s2) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { pr[c] = pa1[c]; pe[c] = pb1[c]; }
    return sqrt(s1);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {  // primitive frame -> box frame: x_B = Rr x_A + t
    pe[c] = fma(Rr[3 * c], pa2[0], fma(Rr[3 * c + 1], pa2[1], fma(Rr[3 * c + 2], pa2[2], t[c])));
    pr[c] = fma(Rr[3 * c], pb2[0], fma(Rr[3 * c + 1], pb2[1], fma(Rr[3 * c + 2], pb2[2], t[c])));
  }
  return sqrt(s2);
}

// signed distance of one primitive to one box with the nearest points (box frame): pr on the robot
// primitive, pe on the box.  Points are NaN when the pair is in contact (distance <= 0).
DEV double narrow_cp(int type, const double PR[9], double d0, double d1, double d2, const DevObs& O, const double t[3],
                     double pr[3], double pe[3], double cutoff = INFINITY) {
  pr[0] = pr[1] = pr[2] = pe[0] = pe[1] = pe[2] = NAN;
  if (type != MVK_PRIM_BOX) {
    const double hl = type == MVK_PRIM_CAPSULE ? d1 : 0.0;
    double a[3], p0[3], dd[3], y[3], z[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) a[c] = fma(O.R[c], PR[2], fma(O.R[3 + c], PR[5], O.R[6 + c] * PR[8]));
#pragma unroll
    for (int c = 0; c < 3; ++c) { p0[c] = fma(-hl, a[c], t[c]); dd[c] = 2.0 * hl * a[c]; }
    const double core = sqrt(seg_box_closest2(p0, dd, O.h, y, z));
    const double d = core - d0;
    if (d > 0.0) {
      const double sc = d0 / core;
#pragma unroll
      for (int c = 0; c < 3; ++c) { pe[c] = z[c]; pr[c] = fma(sc, z[c] - y[c], y[c]); }
    }
    return d;
  }
  double Rr[9];  // primitive axes in the box frame
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      Rr[3 * r + c] = fma(O.R[r], PR[c], fma(O.R[3 + r], PR[3 + c], O.R[6 + r] * PR[6 + c]));
  const double a[3] = {d0, d1, d2};
  double d = box_box_sat<true>(t, Rr, a, O.h, cutoff);
  if (!(d > 0.0) || d >= cutoff) return d;  // in contact, or the separating-axis bound cannot beat the caller's minimum
  return box_box_exact_cp(t, Rr, a, O.h, pr, pe);
}

// ------------------------------------------------------------------------
// state fetch: SoA batch rows, or in-kernel edge densification
// ------------------------------------------------------------------------
DEV double so2_interp(double from, double to, double t) {
  double diff = to - from;
  if (fabs(diff) <= kPi) return fma(diff, t, from);
  diff = (diff > 0.0) ? 2.0 * kPi - diff : -2.0 * kPi - diff;
  double v = fma(-diff, t, from);
  if (v > kPi) v -= 2.0 * kPi;
  else if (v < -kPi) v += 2.0 * kPi;
  return v;
}
DEV void so3_interp(const double* a, const double* b, double t, double* o) {
  const double dq = fma(a[0], b[0], fma(a[1], b[1], fma(a[2], b[2], a[3] * b[3])));
  const double adq = fabs(dq);
  const double theta = (adq > 1.0 - 1e-9) ? 0.0 : acos(adq);
  if (theta > 2.220446049250313e-16) {
    double s0, s1, st, dummy;
    sincos_cw(theta, st, dummy);
    sincos_cw((1.0 - t) * theta, s0, dummy);
    sincos_cw(t * theta, s1, dummy);
    const double d = 1.0 / st;
    if (dq < 0.0) s1 = -s1;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = fma(a[k], s0, b[k] * s1) * d;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = a[k];
  }
}

// unpack of motion_planning.cpp:81-129 from an OMPL-layout state held in `s`
template <int NJ>
DEV void unpack_ompl(const EvalArgs& A, const double* s, double* st) {
  st[0] = st[1] = st[2] = 0.0; st[3] = st[4] = st[5] = 0.0; st[6] = 1.0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) st[7 + j] = 0.0;
  if (A.planning_mode != MVK_MODE_ONLY_JOINTS) {
    if (A.motion_type == MVK_SE2) {
      st[0] = s[0]; st[1] = s[1];
      sincos_cw(0.5 * s[2], st[5], st[6]);  // createQuaternionMsgFromYaw
    } else {
#pragma unroll
      for (int k = 0; k < 7; ++k) st[k] = s[k];
    }
  }
}

// lane -> (edge, step) of an edge batch: step j in 1..nd tests interp(s1, s2, j/nd)
DEV void edge_step(const EvalArgs& A, uint64_t li, uint64_t& edge, int& step, int& nd_e) {
  if (A.nd == nullptr) {
    nd_e = A.nd_uniform;
    edge = li / static_cast<uint64_t>(nd_e);
    step = static_cast<int>(li - edge * nd_e) + 1;
  } else {
    uint64_t lo = 0, hi = A.n_edges;  // largest e with edge_off[e] <= li
    if (A.edge_block) {
      // the edge holding the first lane of this block of lanes, then a few steps forward: an edge of
      // nd steps spans nd lanes, so the answer is usually within a handful of edges (two dependent
      // loads instead of log2(n_edges))
      lo = A.edge_block[li / kBlock];
#pragma unroll 1
      for (int it = 0; it < 8; ++it) {
        if (lo + 1 >= A.n_edges || static_cast<uint64_t>(A.edge_off[lo + 1]) > li) { hi = lo + 1; break; }
        ++lo;
      }
    }
    while (hi - lo > 1) {
      const uint64_t mid = (lo + hi) >> 1;
      if (static_cast<uint64_t>(A.edge_off[mid]) <= li) lo = mid; else hi = mid;
    }
    edge = lo;
    nd_e = A.nd[edge] < 1 ? 1 : A.nd[edge];
    step = static_cast<int>(li - static_cast<uint64_t>(A.edge_off[edge])) + 1;
  }
}

// State of global lane index li: a SoA batch row set, or step j of edge e densified in place
// (lane -> (edge, step): step j in 1..nd tests interp(s1, s2, j/nd); j = nd is s2 itself).
// WANT_JOINTS = false skips the joint rows (phase 2 of the flag path only needs the root pose).
// INLINE: the kernel also serves host calls of one state whose rows travel in the kernel arguments (A.inline_on).
template <int NJ, bool EDGE, bool WANT_JOINTS, bool INLINE = false>
DEV void fetch_state(const EvalArgs& A, uint64_t li, double st[7 + NJ], uint64_t& edge, int& step) {
  edge = 0;
  step = 0;
  if (!EDGE && A.sweep_on) {
    // C-space sweep: joint j of sample i = low_j + span_j * unit(philox(counter = {i, j / 4}, key = seed)[j % 4])
#pragma unroll
    for (int r = 0; r < 7; ++r) st[r] = A.sweep_pose[r];
    const uint64_t i = A.sweep_first + li;
#pragma unroll
    for (int b = 0; b < (NJ + 3) / 4; ++b) {
      uint32_t u[4] = {0, 0, 0, 0};
      if (WANT_JOINTS)
        philox4x32_10(static_cast<uint32_t>(i), static_cast<uint32_t>(i >> 32), static_cast<uint32_t>(b), 0u,
                      static_cast<uint32_t>(A.sweep_seed), static_cast<uint32_t>(A.sweep_seed >> 32), u);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (4 * b + k < NJ) st[7 + 4 * b + k] = WANT_JOINTS ? fma(A.sweep_span[4 * b + k], sweep_unit(u[k]), A.sweep_lo[4 * b + k]) : 0.0;
    }
    return;
  }
  if (!EDGE) {
    const double* __restrict__ src = A.soa;
    const bool inl = INLINE && A.inline_on;
    auto row = [&](int r) -> double { return inl ? A.inline_state[r] : src[r * A.ld + li]; };
    if (A.layout == MVK_LAYOUT_POSE7) {
#pragma unroll
      for (int r = 0; r < 7; ++r) st[r] = row(r);
#pragma unroll
      for (int j = 0; j < NJ; ++j) st[7 + j] = WANT_JOINTS ? row(7 + j) : 0.0;
      if (A.planning_mode == MVK_MODE_ONLY_JOINTS) {
        st[0] = st[1] = st[2] = st[3] = st[4] = st[5] = 0.0; st[6] = 1.0;
      }
      if (A.planning_mode == MVK_MODE_ONLY_BASE) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) st[7 + j] = 0.0;
      }
    } else {
      double s[7];
      const int nb = (A.planning_mode == MVK_MODE_ONLY_JOINTS) ? 0 : (A.motion_type == MVK_SE2 ? 3 : 7);
#pragma unroll
      for (int r = 0; r < 7; ++r) s[r] = (r < nb) ? row(r) : 0.0;
      unpack_ompl<NJ>(A, s, st);
      if (WANT_JOINTS && A.planning_mode != MVK_MODE_ONLY_BASE) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) st[7 + j] = row(nb + j);
      }
    }
  } else {
    int nd_e;
    edge_step(A, li, edge, step, nd_e);
    const double t = static_cast<double>(step) / static_cast<double>(nd_e);
    const bool last = step == nd_e;
    const int nb = (A.planning_mode == MVK_MODE_ONLY_JOINTS) ? 0 : (A.motion_type == MVK_SE2 ? 3 : 7);
    const bool inl = INLINE && A.inline_on;  // (one edge: edge == 0)
    auto row1 = [&](int r) -> double { return inl ? A.inline_state[r] : A.s1[r * A.ld + edge]; };
    auto row2 = [&](int r) -> double { return inl ? A.inline_state2[r] : A.s2[r * A.ld + edge]; };
    double a[7], b[7], s[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      a[r] = (r < nb) ? row1(r) : 0.0;
      b[r] = (r < nb) ? row2(r) : 0.0;
      s[r] = b[r];
    }
    if (!last && nb > 0) {
      if (A.motion_type == MVK_SE2) {
        s[0] = fma(b[0] - a[0], t, a[0]);
        s[1] = fma(b[1] - a[1], t, a[1]);
        s[2] = so2_interp(a[2], b[2], t);
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] = fma(b[k] - a[k], t, a[k]);
        so3_interp(a + 3, b + 3, t, s + 3);
      }
    }
    unpack_ompl<NJ>(A, s, st);
    if (WANT_JOINTS && A.planning_mode != MVK_MODE_ONLY_BASE) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const double ja = row1(nb + j), jb = row2(nb + j);
        st[7 + j] = last ? jb : fma(jb - ja, t, ja);
      }
    }
  }
}

// ------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------
// Flag path = two launches on one stream.
//   screen_kernel : one lane per state.  Walk 1 (model update, stability) and the fp32 screen of
//                   the collision pairs.  Writes every output; a state the screen cannot decide
//                   is written as "valid so far" and its index is appended to the queue of every
//                   chain segment that holds an undecided pair.
//   exact_kernel  : every block takes chunks of one segment's queue, one lane per entry (dense
//                   waves, the segment block-uniform, and a lane's serial work is one segment
//                   instead of the whole chain).  Exact fp64 tests; a collision clears the state's flag / mask
//                   bit (or lowers first_invalid of its edge).
// Full path = full_kernel: exact minimum distance, margins, details and statistics, no early exit.
// edge mode: step `step` of `edge` is invalid.  Standard path: atomicMin into first_invalid (initialised to nd + 1).
// Fused path: atomicMax of (call sequence << 16 | 0xFFFF - step) into first_tag, which needs no initialisation --
// whatever an earlier call left there carries an older sequence number and loses.
DEV void mark_invalid_step(const EvalArgs& A, uint64_t edge, int step) {
  if (A.first_tag) atomicMax(A.first_tag + edge, (A.fused_seq << 16) | (0xFFFFu - static_cast<unsigned>(step)));
  else atomicMin(&A.first_invalid[edge], step);
}

constexpr int kRefineSlot = MVK_MAX_SEG * kQueueShards - 1;  // counter of the refine list, in each counter set
constexpr int kBarrierSlot = MVK_MAX_SEG * kQueueShards - 2, kDoneSlot = MVK_MAX_SEG * kQueueShards - 3;  // fused_kernel
static_assert(MVK_MAX_SEG * kQueueShards - 3 >= 64, "the queues' counters use the first 64 slots at most");

// Hand the undecided (state, segment) pairs of a wave to the exact stage.  Wave by wave and without a block barrier:
// lane k of the wave reserves the wave's share of segment k's queue with one atomic (the queues are split
// kQueueShards ways by wave index, so that no counter sees more than a few thousand atomics per launch), the lanes
// write their state index behind it.  SCATTERED (the lanes' states come from a list, not from one wave of the batch):
// one atomic per lane and segment on the shard of the lane's own state, whose quota nobody else uses.
template <int ROBOT, int NL, bool SCATTERED>
DEV void enqueue_undecided(const EvalArgs& A, uint64_t gi, unsigned todo) {
  using S = Shape<ROBOT, NL>;
  const int lane = threadIdx.x & 63;
  const unsigned shard = static_cast<unsigned>(gi >> 6) & (kQueueShards - 1);
  if (SCATTERED) {
#pragma unroll
    for (int k = 0; k < S::kSeg; ++k)
      if ((todo >> k) & 1u) {
        const unsigned at = atomicAdd(A.queue_count + (k * kQueueShards + shard) * kQueueCounterPitch, 1u);
        A.queue[static_cast<uint64_t>(k * kQueueShards + shard) * A.queue_stride + at] = static_cast<uint32_t>(gi);
      }
    return;
  }
  unsigned long long bal[S::kSeg];
  unsigned my_cnt = 0;
#pragma unroll
  for (int k = 0; k < S::kSeg; ++k) {
    bal[k] = __ballot((todo >> k) & 1u);
    my_cnt = (lane == k) ? static_cast<unsigned>(__popcll(bal[k])) : my_cnt;
  }
  unsigned my_base = 0;
  if (lane < S::kSeg && my_cnt) my_base = atomicAdd(A.queue_count + (lane * kQueueShards + shard) * kQueueCounterPitch, my_cnt);
#pragma unroll
  for (int k = 0; k < S::kSeg; ++k) {
    const unsigned base_k = __builtin_amdgcn_readlane(my_base, k);
    if ((todo >> k) & 1u)
      A.queue[static_cast<uint64_t>(k * kQueueShards + shard) * A.queue_stride + base_k + __popcll(bal[k] & ((1ull << lane) - 1ull))] =
          static_cast<uint32_t>(gi);
  }
}

// REFINE: the states come from the list the fp32 screen left (fp32 variant of the flag path): a fixed grid strides
// over it; a state was written "valid so far", so only a negative verdict is written (as the exact stage does).
// one state per lane through the fp64 screen stage: gi = the state's index in the batch (li = the index it reads:
// out-of-range lanes redo the last state), in_range = the lane owns a state
template <int ROBOT, int NL, bool EDGE, bool REFINE>
DEV void screen_body(const EvalArgs& A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
                     const DevObsF* __restrict__ obsf, const Smem& sm, const ScreenLds& L, uint64_t gi, uint64_t li,
                     bool in_range, unsigned long long t_entry) {
  using S = Shape<ROBOT, NL>;
  constexpr int NJ = S::kJoint;
  (void)t_entry;
  // the counters the NEXT call on this ctx appends to
  if (!REFINE && (gi < static_cast<uint64_t>(S::kSeg * kQueueShards) || (gi >= static_cast<uint64_t>(kDoneSlot) && gi <= static_cast<uint64_t>(kRefineSlot))))
    A.queue_count_next[gi * kQueueCounterPitch] = 0u;

  double st[7 + NJ];
  uint64_t edge;
  int step;
  fetch_state<NJ, EDGE, true>(A, li, st, edge, step);

  StateOut o;
  double Rw[9];
  bool finite;
  model_update<ROBOT, NL, false>(A, M, X, st, sm, li, o, Rw, finite, nullptr);

  // lanes whose verdict cannot change any more (unstable, or not finite) generate no work
  const bool live = finite && A.do_coll && A.n_obs > 0 && (!A.do_stab || o.stable);
  bool hit = false;
  unsigned segmask = 0u;
#ifdef MVK_PROFS
  PROF_STAGE_DECL;
  const unsigned long long t_w1 = wall_clock64();
  PROF_STAGE(0, t_entry, t_w1);
  unsigned long long prof_pop = 0;
  if (__any(live)) collide_screen<ROBOT, NL>(A, M, X, obsf, st, Rw, sm, L, live, hit, segmask, &prof_pop);
  const unsigned long long t_w2 = wall_clock64();
  prof_d[1] += t_w2 - t_w1 - prof_pop;
  prof_d[2] += prof_pop;
#else
  if (__any(live)) collide_screen<ROBOT, NL>(A, M, X, obsf, st, Rw, sm, L, live, hit, segmask, nullptr);
#endif
  const bool undecided = in_range && live && !hit && segmask != 0u;
  PROF_LANES(0, true); PROF_LANES(1, live); PROF_LANES(2, hit); PROF_LANES(3, undecided); PROF_LANES(4, live && !hit && segmask == 0u);

  // hand-off to the exact stage: queue entries, and an undecided lane leaves its record (root pose, joint
  // cos/sin) in the slot of its state
  {
    const unsigned todo = undecided ? segmask : 0u;
    if (__any(todo != 0u)) {
      enqueue_undecided<ROBOT, NL, REFINE>(A, gi, todo);
      if (todo != 0u) {
        double2* __restrict__ dst = reinterpret_cast<double2*>(A.queue_rec + gi * S::kRecPad);
        dst[0] = make_double2(st[0], st[1]);
        dst[1] = make_double2(st[2], st[3]);
        dst[2] = make_double2(st[4], st[5]);
        dst[3] = make_double2(st[6], 0.0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) dst[4 + j] = make_double2(sm.cs[(2 * j) * kBlock + threadIdx.x], sm.cs[(2 * j + 1) * kBlock + threadIdx.x]);
      }
    }
  }

  const bool valid = finite && (!A.do_stab || o.stable) && !hit;
  const bool band = !finite || (A.do_stab && fabs(o.margin) <= A.eps_margin);
  if (REFINE) {
    if (in_range && !valid) {  // the fp32 screen wrote this state as valid-so-far
      if (!EDGE) {
        if (A.flags) A.flags[gi] = 0;
        if (A.mask) atomicAnd(reinterpret_cast<unsigned long long*>(A.mask) + (gi >> 6), ~(1ull << (gi & 63)));
      } else {
        mark_invalid_step(A, edge, step);
      }
    }
  } else if (!EDGE) {
    if (in_range) {
      if (A.flags) A.flags[gi] = valid ? 1 : 0;
      if (A.band) A.band[gi] = band ? 1 : 0;
    }
    if (A.mask) {
      const unsigned long long bal = __ballot(in_range && valid);
      if ((threadIdx.x & 63) == 0 && (gi - (gi & 63)) < A.n) A.mask[gi >> 6] = bal;
    }
  } else {
    if (in_range && !valid) mark_invalid_step(A, edge, step);
    if (A.band && in_range && band) A.band[edge] = 1;
  }
#ifdef MVK_PROFS
  { const unsigned long long t_end = wall_clock64(); prof_d[3] += t_end - t_w2; }
  PROF_STAGE_FLUSH;
#endif
}

template <int ROBOT, int NL, bool EDGE, bool REFINE>
__global__ void __launch_bounds__(kBlock, (Shape<ROBOT, NL>::kScreenWaves))
screen_kernel(const EvalArgs A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
              const DevObsF* __restrict__ obsf) {
  using S = Shape<ROBOT, NL>;
  constexpr int NJ = S::kJoint;
  __shared__ double sm_cs[2 * NJ * kBlock];
  __shared__ unsigned sm_ring[(kBlock / 64) * kRingWords * kRing];
  __shared__ unsigned sm_res[kBlock];
  __shared__ float sm_thr[S::kSeg * kSegPrimMax * 2];
  Smem sm{(lds_double*)sm_cs, nullptr};
#ifdef MVK_PROFS
  const unsigned long long t_entry = wall_clock64();
#else
  const unsigned long long t_entry = 0;
#endif
  const ScreenLds L{(lds_uint*)sm_ring + (threadIdx.x >> 6) * (kRingWords * kRing), (lds_uint*)sm_res + (threadIdx.x & ~63), (lds_float*)sm_thr};
  if (threadIdx.x < S::kSeg * kSegPrimMax) {
    float h2, m2;
    screen_thresholds(A, X, threadIdx.x / kSegPrimMax, threadIdx.x % kSegPrimMax, A.f32_delta, h2, m2);
    L.thr[2 * threadIdx.x] = h2;
    L.thr[2 * threadIdx.x + 1] = m2;
  }
  __syncthreads();

  if constexpr (!REFINE) {
    const uint64_t gi = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const bool in_range = gi < A.n;
    screen_body<ROBOT, NL, EDGE, false>(A, M, X, obsf, sm, L, gi, in_range ? gi : (A.n - 1), in_range, t_entry);
  } else {
    // (no block-level synchronisation in the body: every wave works on its own slots of the shared arrays)
    const unsigned list_count = *A.refine_count;
#pragma unroll 1
    for (uint64_t base = static_cast<uint64_t>(blockIdx.x) * kBlock; base < list_count; base += static_cast<uint64_t>(gridDim.x) * kBlock) {
      const uint64_t slot = base + threadIdx.x;
      const bool in_range = slot < list_count;
      const uint64_t gi = A.refine_list[in_range ? slot : list_count - 1];
      screen_body<ROBOT, NL, EDGE, true>(A, M, X, obsf, sm, L, gi, gi, in_range, t_entry);
    }
  }
}

// fp32 variant of the flag path, screen stage: walk 1 and the frames of walk 2 in fp32.  Stability verdicts are
// taken only where the fp32 error bounds allow it (model_update_f32); the other states go to the refine list for
// screen_kernel<..., REFINE>.  The collision screen is the same fp32 classification with thresholds padded by the
// frame walk's error bound; its undecided pairs go to the exact stage with a record that holds the joint ANGLES
// (tagged), of which the exact stage takes the fp64 cos/sin itself.
#ifndef MVK_SCREEN32_WAVES
#define MVK_SCREEN32_WAVES 3
#endif
template <int ROBOT, int NL, bool EDGE>
__global__ void __launch_bounds__(kBlock, MVK_SCREEN32_WAVES)
screen32_kernel(const EvalArgs A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
                const DevObsF* __restrict__ obsf) {
  using S = Shape<ROBOT, NL>;
  constexpr int NJ = S::kJoint;
  __shared__ float sm_csf[2 * NJ * kBlock];
  __shared__ unsigned sm_ring[(kBlock / 64) * kRingWords * kRing];
  __shared__ unsigned sm_res[kBlock];
  __shared__ float sm_thr[S::kSeg * kSegPrimMax * 2];
  lds_float* csf = (lds_float*)sm_csf;
  const ScreenLds L{(lds_uint*)sm_ring + (threadIdx.x >> 6) * (kRingWords * kRing), (lds_uint*)sm_res + (threadIdx.x & ~63), (lds_float*)sm_thr};
  if (threadIdx.x < S::kSeg * kSegPrimMax) {
    float h2, m2;
    screen_thresholds(A, X, threadIdx.x / kSegPrimMax, threadIdx.x % kSegPrimMax, A.f32_delta_fk, h2, m2);
    L.thr[2 * threadIdx.x] = h2;
    L.thr[2 * threadIdx.x + 1] = m2;
  }
  __syncthreads();

  const uint64_t gi = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const bool in_range = gi < A.n;
  const uint64_t li = in_range ? gi : (A.n - 1);
  if (gi < static_cast<uint64_t>(S::kSeg * kQueueShards) || (gi >= static_cast<uint64_t>(kDoneSlot) && gi <= static_cast<uint64_t>(kRefineSlot)))
    A.queue_count_next[gi * kQueueCounterPitch] = 0u;

  double st[7 + NJ];
  uint64_t edge;
  int step;
  fetch_state<NJ, EDGE, true>(A, li, st, edge, step);
  double Rw[9];
  quat_to_mat(st[3], st[4], st[5], st[6], Rw);
  bool finite = true;
#pragma unroll
  for (int k = 0; k < 7 + NJ; ++k) finite = finite && isfinite(st[k]);
#pragma unroll
  for (int k = 0; k < 9; ++k) finite = finite && isfinite(Rw[k]);
  float Rf[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) Rf[k] = static_cast<float>(Rw[k]);

  int cls = 1;  // 1 stable (or not asked), 0 unstable, -1 uncertain
  if (A.do_stab) {
    float dbg[3];
    cls = model_update_f32<ROBOT, NL>(A, X, st, Rf, csf, A.fc_t_min ? dbg : nullptr);
    if (A.fc_t_min && in_range) {  // diagnostics (mvk_debug_f32_bounds): smallest s - e, smallest s + e, class
      A.fc_t_min[gi] = dbg[0];
      A.fc_rp_min[gi] = dbg[1];
      A.rp_margin[gi] = finite ? static_cast<double>(cls) : -2.0;
    }
  } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float sj, cj;
      sincos_f32(static_cast<float>(st[7 + j]), sj, cj);
      csf[(2 * j) * kBlock + threadIdx.x] = cj;
      csf[(2 * j + 1) * kBlock + threadIdx.x] = sj;
    }
  }
  const bool need_coll = A.do_coll && A.n_obs > 0;
  const bool f32_safe = fmax(fabs(st[0]), fmax(fabs(st[1]), fabs(st[2]))) + X->reach <= A.f32_range;
  // refined in fp64: stability the bounds cannot decide; states whose collision screen fp32 cannot be trusted with
  const bool refine = in_range && finite && cls != 0 && (cls < 0 || (need_coll && !f32_safe));
  const bool live = finite && need_coll && cls == 1 && f32_safe;
  bool hit = false;
  unsigned segmask = 0u;
  if (__any(live)) collide_screen_f32<ROBOT, NL>(A, X, obsf, st, Rf, csf, L, live, hit, segmask);
  const bool undecided = in_range && live && !hit && segmask != 0u;
  {
    const unsigned todo = undecided ? segmask : 0u;
    if (__any(todo != 0u)) {
      enqueue_undecided<ROBOT, NL, false>(A, gi, todo);
      if (todo != 0u) {  // record with the joint angles (tag 1 in the pad slot)
        double2* __restrict__ dst = reinterpret_cast<double2*>(A.queue_rec + gi * S::kRecPad);
        dst[0] = make_double2(st[0], st[1]);
        dst[1] = make_double2(st[2], st[3]);
        dst[2] = make_double2(st[4], st[5]);
        dst[3] = make_double2(st[6], 1.0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) dst[4 + j] = make_double2(st[7 + j], 0.0);
      }
    }
  }
  {
    const unsigned long long bal = __ballot(refine);
    if (bal) {
      const int lane = threadIdx.x & 63;
      const int leader = __ffsll(static_cast<long long>(bal)) - 1;
      unsigned at = 0;
      if (lane == leader) at = atomicAdd(A.refine_count, static_cast<unsigned>(__popcll(bal)));
      at = __shfl(at, leader) + static_cast<unsigned>(__popcll(bal & ((1ull << lane) - 1ull)));
      if (refine) A.refine_list[at] = static_cast<uint32_t>(gi);
    }
  }
  const bool valid = finite && cls != 0 && !hit;  // (a refined state: valid so far)
  if (!EDGE) {
    if (in_range) {
      if (A.flags) A.flags[gi] = valid ? 1 : 0;
      if (A.band) A.band[gi] = finite ? 0 : 1;
    }
    if (A.mask) {
      const unsigned long long bal = __ballot(in_range && valid);
      if ((threadIdx.x & 63) == 0 && (gi - (gi & 63)) < A.n) A.mask[gi >> 6] = bal;
    }
  } else {
    if (in_range && !valid) mark_invalid_step(A, edge, step);
    if (A.band && in_range && !finite) A.band[edge] = 1;
  }
}

// LDS of the exact stage: the joint cos/sin of the frame walk and, once every lane has its frame, the block's pair
// list share one region (a); then the frames [12][kBlock], the list lengths, the verdict bytes, the chunk table
template <int ROBOT, int NL>
struct ExactLds {
  using S = Shape<ROBOT, NL>;
  static constexpr size_t kCsBytes = sizeof(double) * 2 * S::kJoint * kBlock, kItemBytes = sizeof(unsigned) * kSegPrimMax * kItemCap;
  static constexpr size_t kABytes = kCsBytes > kItemBytes ? kCsBytes : kItemBytes;
  static constexpr size_t kFrameOff = kABytes, kNitemsOff = kFrameOff + sizeof(double) * 12 * kBlock,
                          kHitOff = kNitemsOff + 16, kFirstOff = kHitOff + kBlock,
                          kBytes = kFirstOff + sizeof(unsigned) * (S::kSeg * kQueueShards + 1 + 3);
};

template <int ROBOT, int NL, bool EDGE>
DEV void exact_body(const EvalArgs& A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
                    const DevObs* __restrict__ obs, const DevObsF* __restrict__ obsf, unsigned char* lds) {
  using S = Shape<ROBOT, NL>;
  using E = ExactLds<ROBOT, NL>;
  constexpr int NJ = S::kJoint;
  unsigned char* sm_a = lds;
  double* sm_frame_ = reinterpret_cast<double*>(lds + E::kFrameOff);
  unsigned* sm_nitems = reinterpret_cast<unsigned*>(lds + E::kNitemsOff);
  unsigned char* sm_hit = lds + E::kHitOff;
  unsigned* sm_first = reinterpret_cast<unsigned*>(lds + E::kFirstOff);
  Smem sm{(lds_double*)sm_a, nullptr};
  lds_uint* items = (lds_uint*)sm_a;
  lds_double* frames = (lds_double*)sm_frame_;
  const int tid = threadIdx.x, lane = threadIdx.x & 63;

#ifndef MVK_PROFS
  PROF_T(t_entry);
  PROF_STAGE_DECL;
#endif
  // The queues (segment-major, kQueueShards per segment) laid end to end in chunks of one block;
  // the grid strides over the chunks.  sm_first[q] = first chunk of queue q.
  constexpr int kQueues = S::kSeg * kQueueShards;
  static_assert(kQueues <= 64, "one wave scans the queue lengths");
  if (tid < 64) {  // counters written by screen_kernel, the launch before on this stream
    const unsigned mine_chunks = tid < kQueues ? (A.queue_count[tid * kQueueCounterPitch] + kBlock - 1) / kBlock : 0u;
    unsigned incl = mine_chunks;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = __shfl_up(incl, off);
      if (tid >= off) incl += o;
    }
    if (tid == 0) sm_first[0] = 0u;
    if (tid < kQueues) sm_first[tid + 1] = incl;
  }
  __syncthreads();
  const unsigned n_chunks = sm_first[kQueues];
  for (unsigned chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    unsigned lo = 0, hi = kQueues;  // the queue holding this chunk: the last q with first[q] <= chunk (block-uniform)
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const unsigned mid = (lo + hi) >> 1;
      if (hi - lo > 1) { if (sm_first[mid] <= chunk) lo = mid; else hi = mid; }
    }
    const int ks = static_cast<int>(lo / kQueueShards);  // the chain segment
    const unsigned count = A.queue_count[lo * kQueueCounterPitch];
    const uint32_t* __restrict__ queue = A.queue + static_cast<uint64_t>(lo) * A.queue_stride;
    const int pb = X->seg_prim_begin[ks], np = X->seg_prim_begin[ks + 1] - pb;
    const unsigned base = (chunk - sm_first[lo]) * kBlock;
  {
    const unsigned qi = base + tid;
    const bool mine = qi < count;
    const uint64_t gi = queue[mine ? qi : (count - 1)];
#ifndef MVK_PROFS
    PROF_T(t_q);
    PROF_STAGE(0, t_entry, t_q);
#endif
    double st[7], R[9], p[3];
    if (tid < kSegPrimMax) sm_nitems[tid] = 0u;
    sm_hit[tid] = 0;
    {
      const double2* __restrict__ rec = reinterpret_cast<const double2*>(A.queue_rec + gi * S::kRecPad);
      const double2 v0 = rec[0], v1 = rec[1], v2 = rec[2], v3 = rec[3];
      st[0] = v0.x; st[1] = v0.y; st[2] = v1.x; st[3] = v1.y; st[4] = v2.x; st[5] = v2.y; st[6] = v3.x;
      const bool angles = v3.y != 0.0;  // record of the fp32 screen: joint angles, the cos/sin are taken here in fp64
      if (__any(angles)) {  // (rolled: one copy of the sincos code, and the frame of segment ks turns on joints 0 .. ks-1 only)
#pragma unroll 1
        for (int j = 0; j < ks && j < NJ; ++j) {
          double2 v = rec[4 + j];
          if (angles) sincos_cw(v.x, v.y, v.x);
          sm.cs[(2 * j) * kBlock + tid] = v.x;
          sm.cs[(2 * j + 1) * kBlock + tid] = v.y;
        }
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (j < ks) {
            const double2 v = rec[4 + j];
            sm.cs[(2 * j) * kBlock + tid] = v.x;
            sm.cs[(2 * j + 1) * kBlock + tid] = v.y;
          }
        }
      }
    }
    quat_to_mat(st[3], st[4], st[5], st[6], R);
    p[0] = st[0]; p[1] = st[1]; p[2] = st[2];
    PROF_WAVE(15);
    // ---- the frame of segment ks ------------------------------------------------------------
#pragma unroll 1
    for (int k = 1; k <= ks; ++k) advance_frame(M, sm, tid, k, R, p);
#pragma unroll
    for (int e = 0; e < 9; ++e) frames[e * kBlock + tid] = R[e];
#pragma unroll
    for (int e = 0; e < 3; ++e) frames[(9 + e) * kBlock + tid] = p[e];
    __syncthreads();  // the cos/sin region is dead from here on: it becomes the pair list
#ifndef MVK_PROFS
    PROF_T(t_fk);
    PROF_STAGE(1, t_q, t_fk);
#endif

    // ---- level 1: the fp32 screen again; the pairs it cannot clear go to the BLOCK's list --------
    // (a lane's serial work would be its own longest list; shared out, it is the block's mean)
    const bool f32_safe = fmax(fabs(st[0]), fmax(fabs(st[1]), fabs(st[2]))) + X->reach <= A.f32_range;
    PrimF32 Pf;
    prims_f32<S::kGimbal>(A, M, X, pb, np, R, p, f32_safe, Pf);
    double sc[3];
    xform(R, p, X->seg_sph[ks][0], X->seg_sph[ks][1], X->seg_sph[ks][2], sc);
    unsigned gb, ge;
    grid_cell(A.grid, sc, mine, gb, ge);
    bool spilled = false;  // some pair of this lane did not fit the block's list
#pragma unroll 1
    for (unsigned g = gb; __any(g < ge); ++g) {
      PROF_WAVE(8);
      if (g < ge) {
        const uint2 w = *reinterpret_cast<const uint2*>(A.grid.items + static_cast<size_t>(g) * kGridGroup);
        const unsigned ids[kGridGroup] = {w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16};
        DevObsF F[kGridGroup];
#pragma unroll
        for (int c = 0; c < kGridGroup; ++c) F[c] = obsf[ids[c] == kGridNone ? static_cast<unsigned>(A.n_obs) : ids[c]];
#pragma unroll
        for (int c = 0; c < kGridGroup; ++c) {
#pragma unroll
          for (int j = 0; j < kSegPrimMax; ++j) {
            if (j < np) {
              const float l2 = point_box_dist2_f32(F[c], Pf.x[j], Pf.y[j], Pf.z[j]);
              const bool pass = ids[c] != kGridNone && l2 <= Pf.may2[j];
              const unsigned long long bal = __ballot(pass);
              if (bal) {  // one LDS atomic per wave and (candidate, primitive)
                const int leader = __ffsll(static_cast<long long>(bal)) - 1;
                unsigned at = 0;
                if (lane == leader) at = atomicAdd(&sm_nitems[j], static_cast<unsigned>(__popcll(bal)));
                at = __shfl(at, leader) + static_cast<unsigned>(__popcll(bal & ((1ull << lane) - 1ull)));
                if (pass) {
                  if (at < static_cast<unsigned>(kItemCap)) items[j * kItemCap + at] = (static_cast<unsigned>(tid) << 16) | ids[c];
                  else spilled = true;
                }
              }
            }
          }
        }
      }
    }
    __syncthreads();
#ifndef MVK_PROFS
    PROF_T(t_l1);
    PROF_STAGE(2, t_fk, t_l1);
#endif

    // ---- level 2: exact tests, one pair per lane, primitive kind wave-uniform ---------------------
#pragma unroll
    for (int j = 0; j < kSegPrimMax; ++j) {
      if (j >= np) continue;
      const int q = X->prim_order[pb + j];
      const unsigned total = min(sm_nitems[j], static_cast<unsigned>(kItemCap));
#pragma unroll 1
      for (unsigned i = tid; i < total; i += kBlock) {
        PROF_WAVE(9);
        PROF_LANES(10, true);
        const unsigned it = items[j * kItemCap + i];
        const unsigned src = it >> 16;
        double Rs[9], ps[3];
#pragma unroll
        for (int e = 0; e < 9; ++e) Rs[e] = frames[e * kBlock + src];
#pragma unroll
        for (int e = 0; e < 3; ++e) ps[e] = frames[(9 + e) * kBlock + src];
        if (exact_pair<S::kGimbal>(M, X, q, Rs, ps, obs[it & 0xFFFFu])) sm_hit[src] = 1;
      }
    }
    // a lane whose pairs did not all fit tests its whole cell list itself (scenes far denser than
    // the list is sized for)
    if (__any(spilled)) {
#pragma unroll 1
      for (int j = 0; j < np; ++j) {
        const int q = X->prim_order[pb + j];
        const unsigned todo = spilled ? (ge - gb) * kGridGroup : 0u;
#pragma unroll 1
        for (unsigned e = 0; __any(e < todo); ++e) {
          if (e < todo) {
            const unsigned ob = A.grid.items[static_cast<size_t>(gb) * kGridGroup + e];
            if (ob != kGridNone && exact_pair<S::kGimbal>(M, X, q, R, p, obs[ob])) sm_hit[tid] = 1;
          }
        }
      }
    }
    __syncthreads();
#ifndef MVK_PROFS
    PROF_T(t_l2);
    PROF_STAGE(3, t_l1, t_l2);
#endif
    const bool hit = mine && sm_hit[tid] != 0;
    PROF_LANES(13, hit);
    if (hit) {  // the screen wrote this state as valid-so-far
      if (!EDGE) {
        if (A.flags) A.flags[gi] = 0;
        if (A.mask) atomicAnd(reinterpret_cast<unsigned long long*>(A.mask) + (gi >> 6), ~(1ull << (gi & 63)));
      } else {
        uint64_t edge;
        int step, nd_e;
        edge_step(A, gi, edge, step, nd_e);
        mark_invalid_step(A, edge, step);
      }
    }
  }
  }
#ifndef MVK_PROFS
  if (blockIdx.x < n_chunks) PROF_STAGE_FLUSH;
#endif
}

template <int ROBOT, int NL, bool EDGE>
__global__ void __launch_bounds__(kBlock, 3)  // three blocks per CU is also what the launcher sizes the grid for
exact_kernel(const EvalArgs A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
             const DevObs* __restrict__ obs, const DevObsF* __restrict__ obsf) {
  __shared__ __align__(16) unsigned char lds[ExactLds<ROBOT, NL>::kBytes];
  exact_body<ROBOT, NL, EDGE>(A, M, X, obs, obsf, lds);
}

// ------------------------------------------------------------------------
// small batches: the whole flag path in ONE launch
// ------------------------------------------------------------------------
// A planner iteration, a densified path, a single isValid call: a few thousand lanes at most, where the two launches,
// the two copies and the stream synchronisation cost more than the arithmetic.  For batches of at most
// kFusedMaxBlocks blocks -- every block resident at once, one per CU if need be -- screen and exact stage run in one
// kernel around a grid barrier (an arrival counter in the call's counter set; the wait is bounded, a block that
// gives up says so in the done word and the host falls back to the two launches).  Inputs are read from, verdicts
// written to, pinned host memory directly; the last block to finish raises the done word the host polls.

// the second half of a fused launch: grid barrier, exact stage, hand-over of the verdicts by the last block
template <int ROBOT, int NL, bool EDGE>
DEV void fused_tail(const EvalArgs& A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
                    const DevObs* __restrict__ obs, const DevObsF* __restrict__ obsf, unsigned char* lds, unsigned* sm_go) {
  const bool second = A.do_coll && A.n_obs > 0;
  if (second) {
    // ---- grid barrier: every queue entry and record is in place before any block reads them ----
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(A.fused_bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0, go = 1u;
      while (__hip_atomic_load(A.fused_bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
        __builtin_amdgcn_s_sleep(16);  // ~0.4 us between polls: a few hundred pollers must not crowd the arrivals out of the L2 channel
        if (++spins > A.fused_spin_limit) { go = 0u; break; }
      }
      (*sm_go) = go;
    }
    __syncthreads();
    if (!(*sm_go)) {  // (the host repeats the call with two launches: tests/test_gpu_fused_giveup.py forces it)
      if (threadIdx.x == 0) {
        *A.host_done = A.fused_seq | 0x80000000u;
        __threadfence_system();
      }
      return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    exact_body<ROBOT, NL, EDGE>(A, M, X, obs, obsf, lds);
  }
  // ---- completion: the last block to get here hands the verdicts to the host ----
  __syncthreads();
  // (the verdicts live in device memory -- flags, or first_tag -- until here: device-scope ordering is enough for
  // every block but the last, which alone pays for a system-scope release)
  if (threadIdx.x == 0)
    (*sm_go) = __hip_atomic_fetch_add(A.fused_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
  __syncthreads();
  if (!(*sm_go)) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (EDGE) {
    for (uint64_t e = threadIdx.x; e < A.n_edges; e += kBlock) {
      const unsigned v = __hip_atomic_load(A.first_tag + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int nd_e = A.nd ? (A.nd[e] < 1 ? 1 : A.nd[e]) : A.nd_uniform;
      A.host_first[e] = (v >> 16) == (A.fused_seq & 0xFFFFu) ? static_cast<int>(0xFFFFu - (v & 0xFFFFu)) : nd_e + 1;
    }
  } else {  // flags: device buffer -> pinned host block, 16 bytes per lane and step (both padded to 64 bytes)
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(A.flags);
    uint4* __restrict__ dst = reinterpret_cast<uint4*>(A.host_flags);
    for (uint64_t i = threadIdx.x; i < (A.n + 15) / 16; i += kBlock) dst[i] = src[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();  // one system-scope release for the whole block (cumulative over the barrier)
    *A.host_done = A.fused_seq;
  }
}

template <int ROBOT, int NL, bool EDGE>
__global__ void __launch_bounds__(kBlock, (Shape<ROBOT, NL>::kScreenWaves))
fused_kernel(const EvalArgs A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
             const DevObs* __restrict__ obs, const DevObsF* __restrict__ obsf) {
  using S = Shape<ROBOT, NL>;
  constexpr int NJ = S::kJoint;
  constexpr size_t kCs = sizeof(double) * 2 * NJ * kBlock, kRingB = sizeof(unsigned) * (kBlock / 64) * kRingWords * kRing,
                   kResB = sizeof(unsigned) * kBlock, kThrB = sizeof(float) * S::kSeg * kSegPrimMax * 2;
  constexpr size_t kScreenBytes = kCs + kRingB + kResB + kThrB, kExactBytes = ExactLds<ROBOT, NL>::kBytes;
  __shared__ __align__(16) unsigned char lds[kScreenBytes > kExactBytes ? kScreenBytes : kExactBytes];
  __shared__ unsigned sm_go;
  Smem sm{(lds_double*)lds, nullptr};
  const ScreenLds L{(lds_uint*)(lds + kCs) + (threadIdx.x >> 6) * (kRingWords * kRing), (lds_uint*)(lds + kCs + kRingB) + (threadIdx.x & ~63),
                    (lds_float*)(lds + kCs + kRingB + kResB)};
  if (threadIdx.x < S::kSeg * kSegPrimMax) {
    float h2, m2;
    screen_thresholds(A, X, threadIdx.x / kSegPrimMax, threadIdx.x % kSegPrimMax, A.f32_delta, h2, m2);
    L.thr[2 * threadIdx.x] = h2;
    L.thr[2 * threadIdx.x + 1] = m2;
  }
  __syncthreads();
  const uint64_t gi = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const bool in_range = gi < A.n;
  screen_body<ROBOT, NL, EDGE, false>(A, M, X, obsf, sm, L, gi, in_range ? gi : (A.n - 1), in_range, 0ull);

  fused_tail<ROBOT, NL, EDGE>(A, M, X, obs, obsf, lds, &sm_go);
}


// Edge batches of a planner iteration (a few hundred edges, ~20 k steps): at one wave per SIMD the lane of a step runs
// walk 1 and then the collision screen, two chains of dependent work.  Here a block takes 128 steps and gives each
// step TWO lanes in different waves: waves 2-3 run walk 1 (model_update), waves 0-1 the collision screen (own
// cos/sin and root matrix, a few dozen instructions) and the hand-off to the exact stage.  In edge mode the halves need
// not meet: either marks the step invalid in the edge's tag word.  A state that is unstable is still screened (its lane
// would otherwise idle); marks are idempotent.  Twice the blocks of fused_kernel, all resident at once.
template <int ROBOT, int NL>
__global__ void __launch_bounds__(kBlock, 2)  // (at most one block per CU ever runs: registers over occupancy)
fused_split_kernel(const EvalArgs A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
                   const DevObs* __restrict__ obs, const DevObsF* __restrict__ obsf) {
  using S = Shape<ROBOT, NL>;
  constexpr int NJ = S::kJoint;
  constexpr int kHalf = kBlock / 2;
  constexpr size_t kCs = sizeof(double) * 2 * NJ * kBlock, kRingB = sizeof(unsigned) * (kBlock / 64) * kRingWords * kRing,
                   kResB = sizeof(unsigned) * kBlock, kThrB = sizeof(float) * S::kSeg * kSegPrimMax * 2;
  constexpr size_t kScreenBytes = kCs + kRingB + kResB + kThrB, kExactBytes = ExactLds<ROBOT, NL>::kBytes;
  __shared__ __align__(16) unsigned char lds[kScreenBytes > kExactBytes ? kScreenBytes : kExactBytes];
  __shared__ unsigned sm_go;
  Smem sm{(lds_double*)lds, nullptr};
  const ScreenLds L{(lds_uint*)(lds + kCs) + (threadIdx.x >> 6) * (kRingWords * kRing), (lds_uint*)(lds + kCs + kRingB) + (threadIdx.x & ~63),
                    (lds_float*)(lds + kCs + kRingB + kResB)};
  if (threadIdx.x < S::kSeg * kSegPrimMax) {
    float h2, m2;
    screen_thresholds(A, X, threadIdx.x / kSegPrimMax, threadIdx.x % kSegPrimMax, A.f32_delta, h2, m2);
    L.thr[2 * threadIdx.x] = h2;
    L.thr[2 * threadIdx.x + 1] = m2;
  }
  __syncthreads();
  const int tid = threadIdx.x;
  const bool model_half = tid >= kHalf;
  const uint64_t gi = static_cast<uint64_t>(blockIdx.x) * kHalf + static_cast<uint64_t>(tid & (kHalf - 1));
  const bool in_range = gi < A.n;
  const uint64_t li = in_range ? gi : (A.n - 1);
  // the counters the NEXT call on this ctx appends to (as screen_body)
  if (!model_half && (gi < static_cast<uint64_t>(S::kSeg * kQueueShards) || (gi >= static_cast<uint64_t>(kDoneSlot) && gi <= static_cast<uint64_t>(kRefineSlot))))
    A.queue_count_next[gi * kQueueCounterPitch] = 0u;
  double st[7 + NJ];
  uint64_t edge;
  int step;
  fetch_state<NJ, true, true>(A, li, st, edge, step);
  if (model_half) {
    StateOut o;
    double Rw[9];
    bool finite;
    model_update<ROBOT, NL, false>(A, M, X, st, sm, li, o, Rw, finite, nullptr);
    const bool ok = finite && (!A.do_stab || o.stable);
    const bool band = !finite || (A.do_stab && fabs(o.margin) <= A.eps_margin);
    if (in_range && !ok) mark_invalid_step(A, edge, step);
    if (A.band && in_range && band) A.band[edge] = 1;
  } else {
    double Rw[9];
    quat_to_mat(st[3], st[4], st[5], st[6], Rw);
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 7 + NJ; ++k) finite = finite && isfinite(st[k]);
#pragma unroll
    for (int k = 0; k < 9; ++k) finite = finite && isfinite(Rw[k]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      double sj, cj;
      sincos_cw(st[7 + j], sj, cj);
      sm.cs[(2 * j) * kBlock + tid] = cj;
      sm.cs[(2 * j + 1) * kBlock + tid] = sj;
    }
    const bool live = finite && A.do_coll && A.n_obs > 0;
    bool hit = false;
    unsigned segmask = 0u;
    if (__any(live)) collide_screen<ROBOT, NL>(A, M, X, obsf, st, Rw, sm, L, live, hit, segmask, nullptr);
    if (in_range && hit) mark_invalid_step(A, edge, step);
    const unsigned todo = (in_range && live && !hit) ? segmask : 0u;
    if (__any(todo != 0u)) {
      enqueue_undecided<ROBOT, NL, false>(A, gi, todo);
      if (todo != 0u) {
        double2* __restrict__ dst = reinterpret_cast<double2*>(A.queue_rec + gi * S::kRecPad);
        dst[0] = make_double2(st[0], st[1]);
        dst[1] = make_double2(st[2], st[3]);
        dst[2] = make_double2(st[4], st[5]);
        dst[3] = make_double2(st[6], 0.0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) dst[4 + j] = make_double2(sm.cs[(2 * j) * kBlock + tid], sm.cs[(2 * j + 1) * kBlock + tid]);
      }
    }
  }
  fused_tail<ROBOT, NL, true>(A, M, X, obs, obsf, lds, &sm_go);
}

// Full path = two launches on one stream: the statics want registers and have no gathers, the distance
// walks are chains of dependent gathers (cell -> ids -> boxes) and want occupancy, so each gets its own kernel.
//   model_kernel    : walk 1 in full (FK, CoG, statics, feasible-control distances, margins, details).  With the
//                     collision stage on it leaves, per state, the joints' cos/sin rows and the five scalars the
//                     verdict and the statistics need (FullRec rows, lane-contiguous: coalesced both ways);
//                     without it (mode 0, empty world) it writes the verdicts itself.
//   distance_kernel : the two grid walks and the narrowphase drain (collide_full), then verdict, band, statistics.
// Rows of the hand-off, each [ld] doubles: cos/sin of joint j at 2j / 2j+1, then
constexpr int kFullRecMargin = 0, kFullRecLamMin = 1, kFullRecLamMax = 2, kFullRecRp = 3, kFullRecFinite = 4, kFullRecScalars = 5;

// verdict, band, per-state outputs and the block's partial statistics (motion_planning.cpp:615-632 extrema)
DEV void full_finish(const EvalArgs& A, uint64_t gi, bool in_range, bool finite, double margin, double lam_min, double lam_max,
                     double rp_margin, double mind, bool stable, double margin_err = 0.0, double dist_tol = 0.0) {
  StateOut o;
  o.margin = margin; o.lam_min = lam_min; o.lam_max = lam_max; o.rp_margin = rp_margin; o.stable = stable;
  o.min_dist = (A.do_coll && finite) ? mind : INFINITY;
  o.collision = A.do_coll && (mind <= 0.0);
  o.valid = finite && (!A.do_stab || o.stable) && !o.collision;
  // (fp32 variant: margin_err is the half width of the margin's enclosure, eps_dist the variant's distance tolerance)
  o.band = !finite || (A.do_stab && (margin_err > 0.0 ? !(fabs(o.margin) > A.eps_margin + margin_err) : fabs(o.margin) <= A.eps_margin)) ||
           (A.do_coll && fabs(o.min_dist) <= fmax(A.eps_dist, dist_tol));
  if (!finite) { o.margin = -INFINITY; o.min_dist = NAN; o.stable = false; o.collision = false; }

  if (in_range) {
    if (A.flags) A.flags[gi] = o.valid ? 1 : 0;
    if (A.band) A.band[gi] = o.band ? 1 : 0;
    if (A.min_dist) A.min_dist[gi] = o.min_dist;
    if (A.margin) A.margin[gi] = o.margin;
    if (A.min_dist_f32) A.min_dist_f32[gi] = static_cast<float>(o.min_dist);
  }
  if (A.mask) {
    const unsigned long long bal = __ballot(in_range && o.valid);
    if ((threadIdx.x & 63) == 0 && (gi - (gi & 63)) < A.n) A.mask[gi >> 6] = bal;
  }

  if (A.stats_partial) {
    __shared__ StatsPartial sm_part[kBlock / 64];
    StatsPartial mine;
    const long long idx = static_cast<long long>(gi);
    mine.valid_count = (in_range && o.valid) ? 1 : 0;
    mine.band_count = (in_range && o.band) ? 1 : 0;
    const bool use = in_range;
    mine.min_dist = (use && !isnan(o.min_dist)) ? o.min_dist : INFINITY;  mine.min_dist_index = idx;
    mine.min_margin = (use && !isnan(o.margin)) ? o.margin : INFINITY;    mine.min_margin_index = idx;
    mine.max_thrust = (use && !isnan(o.lam_max)) ? o.lam_max : -INFINITY; mine.max_thrust_index = idx;
    mine.min_thrust = (use && !isnan(o.lam_min)) ? o.lam_min : INFINITY;  mine.min_thrust_index = idx;
    mine.min_rp = (use && !isnan(o.rp_margin)) ? o.rp_margin : INFINITY;  mine.min_rp_index = idx;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mine.valid_count += __shfl_xor(mine.valid_count, off);
      mine.band_count += __shfl_xor(mine.band_count, off);
      auto red_min = [&](double& v, long long& ix) {
        const double ov = __shfl_xor(v, off);
        const long long oi = __shfl_xor(ix, off);
        if (ov < v || (ov == v && oi < ix)) { v = ov; ix = oi; }
      };
      auto red_max = [&](double& v, long long& ix) {
        const double ov = __shfl_xor(v, off);
        const long long oi = __shfl_xor(ix, off);
        if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
      };
      red_min(mine.min_dist, mine.min_dist_index);
      red_min(mine.min_margin, mine.min_margin_index);
      red_max(mine.max_thrust, mine.max_thrust_index);
      red_min(mine.min_thrust, mine.min_thrust_index);
      red_min(mine.min_rp, mine.min_rp_index);
    }
    if ((threadIdx.x & 63) == 0) sm_part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
      StatsPartial acc = sm_part[0];
      for (int w = 1; w < kBlock / 64; ++w) {
        const StatsPartial& q = sm_part[w];
        acc.valid_count += q.valid_count;
        acc.band_count += q.band_count;
        if (q.min_dist < acc.min_dist || (q.min_dist == acc.min_dist && q.min_dist_index < acc.min_dist_index)) { acc.min_dist = q.min_dist; acc.min_dist_index = q.min_dist_index; }
        if (q.min_margin < acc.min_margin || (q.min_margin == acc.min_margin && q.min_margin_index < acc.min_margin_index)) { acc.min_margin = q.min_margin; acc.min_margin_index = q.min_margin_index; }
        if (q.max_thrust > acc.max_thrust || (q.max_thrust == acc.max_thrust && q.max_thrust_index < acc.max_thrust_index)) { acc.max_thrust = q.max_thrust; acc.max_thrust_index = q.max_thrust_index; }
        if (q.min_thrust < acc.min_thrust || (q.min_thrust == acc.min_thrust && q.min_thrust_index < acc.min_thrust_index)) { acc.min_thrust = q.min_thrust; acc.min_thrust_index = q.min_thrust_index; }
        if (q.min_rp < acc.min_rp || (q.min_rp == acc.min_rp && q.min_rp_index < acc.min_rp_index)) { acc.min_rp = q.min_rp; acc.min_rp_index = q.min_rp_index; }
      }
      static_cast<StatsPartial*>(A.stats_partial)[blockIdx.x] = acc;
    }
  }
}

template <int ROBOT, int NL>
__global__ void __launch_bounds__(kBlock, 2)
model_kernel(const EvalArgs A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X) {
  using S = Shape<ROBOT, NL>;
  constexpr int NJ = S::kJoint;
  __shared__ double sm_cs[2 * NJ * kBlock];
  Smem sm{(lds_double*)sm_cs, nullptr};

  const uint64_t gi = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const bool in_range = gi < A.n;
  const uint64_t li = in_range ? gi : (A.n - 1);

  double st[7 + NJ];
  uint64_t edge;
  int step;
  fetch_state<NJ, false, true>(A, li, st, edge, step);

  StateOut o;
  double thrust[NL];
  double Rw[9];
  bool finite;
  model_update<ROBOT, NL, true>(A, M, X, st, sm, li, o, Rw, finite, thrust);

  if (in_range) {
    if (A.fc_t_min) A.fc_t_min[gi] = o.fc_t_min;
    if (A.fc_rp_min) A.fc_rp_min[gi] = o.fc_rp_min;
    if (A.rp_margin) A.rp_margin[gi] = o.rp_margin;
    if (A.thrust) {
#pragma unroll
      for (int i = 0; i < NL; ++i) A.thrust[i * A.ld + gi] = thrust[i];
    }
  }
  if (A.full_rec) {  // the collision stage follows in distance_kernel
    double* __restrict__ rec = A.full_rec;
#pragma unroll
    for (int r = 0; r < 2 * NJ; ++r) rec[static_cast<uint64_t>(r) * A.full_rec_ld + gi] = sm.cs[r * kBlock + threadIdx.x];
    double* __restrict__ sc = rec + static_cast<uint64_t>(2 * NJ) * A.full_rec_ld;
    sc[static_cast<uint64_t>(kFullRecMargin) * A.full_rec_ld + gi] = o.margin;
    sc[static_cast<uint64_t>(kFullRecLamMin) * A.full_rec_ld + gi] = o.lam_min;
    sc[static_cast<uint64_t>(kFullRecLamMax) * A.full_rec_ld + gi] = o.lam_max;
    sc[static_cast<uint64_t>(kFullRecRp) * A.full_rec_ld + gi] = o.rp_margin;
    sc[static_cast<uint64_t>(kFullRecFinite) * A.full_rec_ld + gi] = finite ? 1.0 : 0.0;
    return;
  }
  full_finish(A, gi, in_range, finite, o.margin, o.lam_min, o.lam_max, o.rp_margin, INFINITY, o.stable);
}

// Occupancy over registers: the walks wait on dependent gathers, so the kernel is compiled for kDistanceWaves waves
// per SIMD and its LDS (candidate lists, first-walk bounds, the wave's exact-distance batch) is sized to allow them.
#ifndef MVK_DISTANCE_WAVES
#define MVK_DISTANCE_WAVES 2
#endif
constexpr int kDistanceWaves = MVK_DISTANCE_WAVES;
// model32_kernel: walk 1 of the fp32 variant (mvk_eval_batch_f32_dev).  The statics run in fp32 with their error bounds
// (model_update_f32), which enclose the fp64 margin: [lo, hi] -> midpoint and half width, written as floats; the
// verdict takes the midpoint, the band every state whose enclosure reaches within eps_margin of zero.  The hand-off
// rows for the distance stage carry fp64 cos/sin of the joints (the chain's frames are then as in the fp64 path: the
// fp32 part of the distance stage is its broadphase and its narrowphase), the midpoint, the half width and 'finite'.
template <int ROBOT, int NL>
__global__ void __launch_bounds__(kBlock, 2)
model32_kernel(const EvalArgs A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X) {
  using S = Shape<ROBOT, NL>;
  constexpr int NJ = S::kJoint;
  __shared__ float sm_csf[2 * NJ * kBlock];
  const uint64_t gi = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const bool in_range = gi < A.n;
  const uint64_t li = in_range ? gi : (A.n - 1);
  double st[7 + NJ];
  uint64_t edge;
  int step;
  fetch_state<NJ, false, true>(A, li, st, edge, step);
  double Rw[9];
  quat_to_mat(st[3], st[4], st[5], st[6], Rw);
  bool finite = true;
#pragma unroll
  for (int k = 0; k < 7 + NJ; ++k) finite = finite && isfinite(st[k]);
#pragma unroll
  for (int k = 0; k < 9; ++k) finite = finite && isfinite(Rw[k]);
  float Rf[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) Rf[k] = static_cast<float>(Rw[k]);
  float mid = INFINITY, err = 0.0f;
  if (A.do_stab) {
    float dbg[3], mdbg[2];
    (void)model_update_f32<ROBOT, NL>(A, X, st, Rf, (lds_float*)sm_csf, dbg, mdbg);
    mid = 0.5f * (mdbg[0] + mdbg[1]);
    err = 0.5f * (mdbg[1] - mdbg[0]) + 2.0f * kF32U * fabsf(mid);
    if (dbg[2] != 0.0f || !(err < INFINITY) || !(fabsf(mid) < INFINITY)) { err = INFINITY; mid = isfinite(mid) ? mid : 0.0f; }  // not bounded in fp32
  }
  if (!finite) { mid = -INFINITY; err = 0.0f; }
  if (in_range) {
    if (A.margin_f32) A.margin_f32[gi] = mid;
    if (A.margin_err_f32) A.margin_err_f32[gi] = err;
  }
  const double margin = static_cast<double>(mid), margin_err = static_cast<double>(err);
  if (A.full_rec) {  // the collision stage follows in distance_kernel
    double* __restrict__ rec = A.full_rec;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      double sj, cj;
      sincos_cw(st[7 + j], sj, cj);
      rec[static_cast<uint64_t>(2 * j) * A.full_rec_ld + gi] = cj;
      rec[static_cast<uint64_t>(2 * j + 1) * A.full_rec_ld + gi] = sj;
    }
    double* __restrict__ sc = rec + static_cast<uint64_t>(2 * NJ) * A.full_rec_ld;
    sc[static_cast<uint64_t>(kFullRecMargin) * A.full_rec_ld + gi] = margin;
    sc[static_cast<uint64_t>(kFullRecLamMin) * A.full_rec_ld + gi] = margin_err;  // (this variant: the half width rides in the thrust row)
    sc[static_cast<uint64_t>(kFullRecLamMax) * A.full_rec_ld + gi] = NAN;
    sc[static_cast<uint64_t>(kFullRecRp) * A.full_rec_ld + gi] = NAN;
    sc[static_cast<uint64_t>(kFullRecFinite) * A.full_rec_ld + gi] = finite ? 1.0 : 0.0;
    return;
  }
  full_finish(A, gi, in_range, finite, margin, NAN, NAN, NAN, INFINITY, margin >= 0.0, margin_err);
}

template <int ROBOT, int NL, bool F32N = false>
__global__ void __launch_bounds__(kBlock, kDistanceWaves)
distance_kernel(const EvalArgs A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
                const DevObs* __restrict__ obs) {
  using S = Shape<ROBOT, NL>;
  constexpr int NJ = S::kJoint;
  __shared__ unsigned short sm_list[kListCap * kBlock];
  __shared__ double sm_ex_item[(kBlock / 64) * kExactBatch * kExactItem];
  __shared__ unsigned long long sm_ex_res[(kBlock / 64) * kExactBatch];
  __shared__ float sm_upper[kUpperSlots * kBlock];
  __shared__ unsigned sm_dring[(kBlock / 64) * kDRingWords * kRing];
  __shared__ unsigned sm_dcnt[kBlock];
  __shared__ unsigned long long sm_mres[kBlock];
  __shared__ unsigned long long sm_gmin[kBlock];
  const uint64_t gi = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const bool in_range = gi < A.n;
  const uint64_t li = in_range ? gi : (A.n - 1);
  Smem sm{nullptr, (lds_ushort*)sm_list, (lds_double*)sm_ex_item + (threadIdx.x >> 6) * kExactBatch * kExactItem,
          (lds_u64*)sm_ex_res + (threadIdx.x >> 6) * kExactBatch, nullptr, nullptr, (lds_float*)sm_upper,
          A.full_rec, A.full_rec_ld, gi,   // (the record rows are written for every launched lane: gi, not li)
          (lds_uint*)sm_dring + (threadIdx.x >> 6) * (kDRingWords * kRing), (lds_uint*)sm_dcnt + (threadIdx.x & ~63),
          (lds_u64*)sm_mres + (threadIdx.x & ~63), (lds_u64*)sm_gmin + (threadIdx.x & ~63)};

  double st[7 + NJ];
  uint64_t edge;
  int step;
  fetch_state<NJ, false, false>(A, li, st, edge, step);  // the root pose; the joints come as cos/sin rows
  double Rw[9];
  quat_to_mat(st[3], st[4], st[5], st[6], Rw);
  const double* __restrict__ sc = A.full_rec + static_cast<uint64_t>(2 * NJ) * A.full_rec_ld;
  const bool finite = sc[static_cast<uint64_t>(kFullRecFinite) * A.full_rec_ld + gi] != 0.0;

  double mind = INFINITY;
  if (__any(finite)) mind = collide_full<ROBOT, NL, F32N>(A, M, X, obs, st, Rw, sm, finite);
  PROF_WAVE(28);

  const double margin = sc[static_cast<uint64_t>(kFullRecMargin) * A.full_rec_ld + gi];
  if constexpr (F32N) {  // (model32_kernel's rows: the thrust row carries the margin's half width)
    // the variant's distance tolerance at this state (include/mvk.h: mvk_f32_distance_tolerance)
    const double tol = MVK_F32_DIST_TOL_ULPS * 5.9604644775390625e-8 * (fmax(fabs(st[0]), fmax(fabs(st[1]), fabs(st[2]))) + X->reach + 1.0);
    full_finish(A, gi, in_range, finite, margin, NAN, NAN, NAN, mind, margin >= 0.0, sc[static_cast<uint64_t>(kFullRecLamMin) * A.full_rec_ld + gi], tol);
    return;
  }
  full_finish(A, gi, in_range, finite, margin, sc[static_cast<uint64_t>(kFullRecLamMin) * A.full_rec_ld + gi],
              sc[static_cast<uint64_t>(kFullRecLamMax) * A.full_rec_ld + gi], sc[static_cast<uint64_t>(kFullRecRp) * A.full_rec_ld + gi],
              mind, margin >= 0.0);
}

// Exact distances of the box pairs some lanes of a wave hold (`need`), computed by the whole wave as in collide_full:
// kExactBatch pairs at a time go to LDS with their two half-extent triples (the primitive differs from lane to lane
// here), every lane takes (pair, edge) items, an LDS atomic min merges each pair's 24 squared edge distances.  Returns
// the pair's distance to its lane (+inf to the others): the same bits as box_box_exact.
constexpr int kExactItemAB = 18;  // relative rotation 9, centre 3, box half extents 3, primitive half extents 3
DEV double exact_wave_ab(lds_double* item, lds_u64* res, bool need, const double Rr[9], const double t[3], const double h[3],
                         const double a[3]) {
  const int lane = threadIdx.x & 63;
  double out = INFINITY;
  unsigned long long pend = __ballot(need);
#pragma unroll 1
  while (pend) {
    const int rank = __popcll(pend & ((1ull << lane) - 1ull));
    const bool mine = need && ((pend >> lane) & 1ull) && rank < kExactBatch;
    const int E = min(static_cast<int>(__popcll(pend)), kExactBatch);
    if (mine) {
#pragma unroll
      for (int c = 0; c < 9; ++c) item[rank * kExactItemAB + c] = Rr[c];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        item[rank * kExactItemAB + 9 + c] = t[c];
        item[rank * kExactItemAB + 12 + c] = h[c];
        item[rank * kExactItemAB + 15 + c] = a[c];
      }
      res[rank] = 0x7FF0000000000000ull;  // +inf
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int it = lane; it < 24 * E; it += 64) {
      const int pr = it / 24, ed = it - 24 * pr;
      double iR[9], it_[3], ih[3], ia[3];
#pragma unroll
      for (int c = 0; c < 9; ++c) iR[c] = item[pr * kExactItemAB + c];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        it_[c] = item[pr * kExactItemAB + 9 + c];
        ih[c] = item[pr * kExactItemAB + 12 + c];
        ia[c] = item[pr * kExactItemAB + 15 + c];
      }
      const bool second = ed >= 12;  // edges 0..11: the primitive's against the box; 12..23: the box's against the primitive
      double cR[9], cc[3], ca[3], cb[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) cR[3 * r + c] = second ? iR[3 * c + r] : iR[3 * r + c];
        const double tb = -fma(iR[r], it_[0], fma(iR[3 + r], it_[1], iR[6 + r] * it_[2]));
        cc[r] = second ? tb : it_[r];
        ca[r] = second ? ih[r] : ia[r];
        cb[r] = second ? ia[r] : ih[r];
      }
      const double s2 = box_edge_to_box2(second ? ed - 12 : ed, cc, cR, ca, cb);
      __hip_atomic_fetch_min(&res[pr], static_cast<unsigned long long>(__double_as_longlong(s2)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (mine) out = sqrt(__longlong_as_double(static_cast<long long>(res[rank])));
    pend &= ~__ballot(mine);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  return out;
}

// The same with the closest points (avoidance rows): an LDS atomic min per pair over its 24 squared distances, a second
// one over the edge indices that reach it (the lowest wins, as in the one-lane form box_box_exact_cp: robot edges
// before box edges, ascending), and the owner lane repeats that one edge for its points.  Same arithmetic per edge,
// so the result keeps its bits.  pr on the primitive, pq on the box, both in the box frame.
DEV double exact_cp_wave_ab(lds_double* item, lds_u64* res, lds_double* s2s, lds_uint* win, bool need, const double Rr[9],
                            const double t[3], const double h[3], const double a[3], double pr[3], double pq[3]) {
  const int lane = threadIdx.x & 63;
  double out = INFINITY;
  // one (pair, edge): inputs as box_box_exact_cp builds them for its two halves
  auto edge_item = [&](const double iR[9], const double it_[3], const double ih[3], const double ia[3], int ed, double y[3], double z[3]) {
    const bool second = ed >= 12;
    double cR[9], cc[3], ca[3], cb[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) cR[3 * r + c] = second ? iR[3 * c + r] : iR[3 * r + c];
      const double tb = -fma(iR[r], it_[0], fma(iR[3 + r], it_[1], iR[6 + r] * it_[2]));
      cc[r] = second ? tb : it_[r];
      ca[r] = second ? ih[r] : ia[r];
      cb[r] = second ? ia[r] : ih[r];
    }
    return box_edge_to_box_cp2(second ? ed - 12 : ed, cc, cR, ca, cb, y, z);
  };
  unsigned long long pend = __ballot(need);
#pragma unroll 1
  while (pend) {
    const int rank = __popcll(pend & ((1ull << lane) - 1ull));
    const bool mine = need && ((pend >> lane) & 1ull) && rank < kCpBatch;
    const int E = min(static_cast<int>(__popcll(pend)), kCpBatch);
    if (mine) {
#pragma unroll
      for (int c = 0; c < 9; ++c) item[rank * kExactItemAB + c] = Rr[c];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        item[rank * kExactItemAB + 9 + c] = t[c];
        item[rank * kExactItemAB + 12 + c] = h[c];
        item[rank * kExactItemAB + 15 + c] = a[c];
      }
      res[rank] = 0x7FF0000000000000ull;  // +inf
      win[rank] = 0xFFFFFFFFu;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int it = lane; it < 24 * E; it += 64) {
      const int pr_ = it / 24, ed = it - 24 * pr_;
      double iR[9], it_[3], ih[3], ia[3], y[3], z[3];
#pragma unroll
      for (int c = 0; c < 9; ++c) iR[c] = item[pr_ * kExactItemAB + c];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        it_[c] = item[pr_ * kExactItemAB + 9 + c];
        ih[c] = item[pr_ * kExactItemAB + 12 + c];
        ia[c] = item[pr_ * kExactItemAB + 15 + c];
      }
      const double s2 = edge_item(iR, it_, ih, ia, ed, y, z);
      s2s[it] = s2;
      __hip_atomic_fetch_min(&res[pr_], static_cast<unsigned long long>(__double_as_longlong(s2)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int it = lane; it < 24 * E; it += 64) {
      const int pr_ = it / 24, ed = it - 24 * pr_;
      if (static_cast<unsigned long long>(__double_as_longlong(s2s[it])) == res[pr_])
        __hip_atomic_fetch_min(&win[pr_], static_cast<unsigned>(ed), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (mine) {
      const int ed = static_cast<int>(win[rank]);
      double y[3], z[3];
      const double s2 = edge_item(Rr, t, h, a, ed, y, z);
      if (ed < 12) {  // y on the robot's edge, z on the box (box frame)
#pragma unroll
        for (int c = 0; c < 3; ++c) { pr[c] = y[c]; pq[c] = z[c]; }
      } else {        // y on the box's edge, z on the robot (primitive frame): x_B = Rr x_A + t
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          pq[c] = fma(Rr[3 * c], y[0], fma(Rr[3 * c + 1], y[1], fma(Rr[3 * c + 2], y[2], t[c])));
          pr[c] = fma(Rr[3 * c], z[0], fma(Rr[3 * c + 1], z[1], fma(Rr[3 * c + 2], z[2], t[c])));
        }
      }
      out = sqrt(s2);
    }
    pend &= ~__ballot(mine);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  return out;
}

// The same stage for small batches (mvk_is_valid, the states of a densified path): distance_kernel gives a state one
// lane, which then walks its primitives and their candidate lists one after the other -- 0.16 ms for a single state,
// whatever the size of the batch up to a chip-full.  Here a state has a wave of its own (kSmallGroup lanes): lane (qi, part) takes
// primitive qi of the chain order and every K-th box of its candidate list, K = kSmallGroup / P slices per primitive.
// Everything is fp64: per lane the centre distance against the bound and the separating-axis gap; the exact edge
// distances of the box pairs whose gap undercuts the minimum are computed by the whole wave (exact_wave_ab: one lane
// alone would take ~20 us per pair); the running minimum is shared by the group after every round.
// The result is the minimum of the same per-pair values as in distance_kernel (a pair is skipped only when its lower
// bound, centre distance - bounding radius, cannot undercut max(minimum, 0)), so it carries the same bits.
// Candidates: the distance grid's cell list of the primitive's centre (complete within dgrid_radius); a primitive the
// list cannot answer (rb + max(minimum, 0) beyond the radius) goes on to the wider second grid, then to all boxes.
static_assert(kSmallGroup >= MVK_MAX_PRIM && kBlock % kSmallGroup == 0 && kSmallGroup <= 64 && 64 % kSmallGroup == 0, "one group per state inside a wave");
// the wave's work on one state: the minimum signed distance over its primitives and the boxes (`sub` = lane within the
// state's group, R / p = the root frame, cs = the joints' cos/sin); returned to every lane of the group
template <int ROBOT, int NL>
DEV double distance_small_body(const EvalArgs& A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
                               const DevObs* __restrict__ obs, bool in_range, bool finite, int sub, double R[9], double p[3],
                               const double* cs, lds_double* ex_item, lds_u64* ex_res) {
  using S = Shape<ROBOT, NL>;
  const int P = M->n_prim;
  const int K = kSmallGroup / P;  // >= 1 (P <= MVK_MAX_PRIM)
  const bool active = in_range && finite && sub < P * K;
  const int qi = sub % P, part = sub / P;
  const int q = X->prim_order[qi];
  int myk = 0;
#pragma unroll 1
  for (int k = 1; k < S::kSeg; ++k) myk = qi >= X->seg_prim_begin[k] ? k : myk;
  // the chain up to this lane's segment: advance_frame with the joint index and axis known at compile time
  static_for<1, S::kSeg>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if (k <= myk) {
      double pn[3];
      xform(R, p, M->seg_t[k][0], M->seg_t[k][1], M->seg_t[k][2], pn);
      p[0] = pn[0]; p[1] = pn[1]; p[2] = pn[2];
      rot_cols<S::axis(k)>(R, cs[2 * S::joint(k)], cs[2 * S::joint(k) + 1]);
    }
  });
  double pc[3], PR[9];
  prim_pose<S::kGimbal, true>(M, q, R, p, pc, PR);
  const int type = M->prim_type[q];
  const bool sphere = type == MVK_PRIM_SPHERE;
  const double rb = X->prim_bound[q];
  const double d0 = M->prim_dim[q][0], d1 = M->prim_dim[q][1], d2 = M->prim_dim[q][2];

  double best = INFINITY;  // the group's running minimum (after group_min: the same in all its lanes)
  auto group_min = [&](double v) {
#pragma unroll
    for (int off = kSmallGroup / 2; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    return v;
  };
  // two boxes per round, their loads independent: centre distances first, the narrowphase for those within reach
  auto centre = [&](int ob, double t[3]) -> double {
    if (ob == kGridNone) return INFINITY;
    to_box_frame(obs[ob], pc, t);
    return point_box_dist2(t, obs[ob].h);
  };
  // (called by the whole wave: the exact stage is shared)
  const double a[3] = {d0, d1, d2};
  auto finish = [&](int ob, const double t[3], double l2) {
    const double thr = rb + fmax(best, 0.0);
    const bool reach = ob != kGridNone && l2 <= thr * thr * (1.0 + 1e-12);
    bool need = false;
    double Rr[9], h[3];
    if (reach) {
      if (type == MVK_PRIM_BOX) {  // as narrow<true>, with the exact distance left to the wave
        const DevObs& O = obs[ob];
        box_rel(O, PR, Rr);
        h[0] = O.h[0]; h[1] = O.h[1]; h[2] = O.h[2];
        const double g = box_box_sat<true>(t, Rr, a, h, best + kSatSlack);
        if (g <= 0.0) best = fmin(best, g);
        else need = g < best + kSatSlack;
      } else {
        best = fmin(best, sphere ? sqrt(l2) - d0 : narrow<true>(type, PR, d0, d1, d2, obs[ob], t, best));
      }
    }
    best = fmin(best, exact_wave_ab(ex_item, ex_res, need, Rr, t, h, a));
  };
  auto scan = [&](const uint16_t* __restrict__ ids, int n_items) {  // ids == nullptr: boxes 0 .. n_items-1
#pragma unroll 1
    for (int e = part; __any(e < n_items); e += 2 * K) {
      const int e1 = e + K;
      const int ob0 = e < n_items ? (ids ? ids[e] : e) : kGridNone;
      const int ob1 = e1 < n_items ? (ids ? ids[e1] : e1) : kGridNone;
      double t0[3], t1[3];
      const double l0 = centre(ob0, t0), l1 = centre(ob1, t1);
      finish(ob0, t0, l0);
      finish(ob1, t1, l1);
      best = group_min(best);
    }
  };
  auto scan_cell = [&](const GridDev& G, bool who) {
    unsigned gb, ge;
    grid_cell(G, pc, who, gb, ge);
    scan(G.items + static_cast<size_t>(gb) * kGridGroup, static_cast<int>(ge - gb) * kGridGroup);
  };
  const double pad = 1e-9;
  bool todo = active;  // primitives whose candidates are not complete yet
  if (A.dgrid.start != nullptr) {
    scan_cell(A.dgrid, todo);
    todo = todo && !(rb + fmax(best, 0.0) + pad <= A.dgrid_radius);
    if (A.dgrid2.start != nullptr && A.dgrid2_radius > A.dgrid_radius && __any(todo)) {
      scan_cell(A.dgrid2, todo);
      todo = todo && !(rb + fmax(best, 0.0) + pad <= A.dgrid2_radius);
    }
  }
  if (A.fgrid.start != nullptr && __any(todo)) {  // the far grid: complete at any distance, inside its volume
    unsigned gb, ge;
    const bool inside = grid_cell_in(A.fgrid, pc, todo, gb, ge);
    scan(A.fgrid.items + static_cast<size_t>(gb) * kGridGroup, static_cast<int>(ge - gb) * kGridGroup);
    todo = todo && !inside;
  }
  if (__any(todo)) scan(nullptr, todo ? A.n_obs : 0);
  const double mind = group_min(best);
  return mind;
}

template <int ROBOT, int NL>
__global__ void __launch_bounds__(kBlock)
distance_small_kernel(const EvalArgs A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
                      const DevObs* __restrict__ obs) {
  using S = Shape<ROBOT, NL>;
  constexpr int NJ = S::kJoint;
  __shared__ double sm_ex_item[(kBlock / 64) * kExactBatch * kExactItemAB];
  __shared__ unsigned long long sm_ex_res[(kBlock / 64) * kExactBatch];
  const int tid = threadIdx.x, sub = tid & (kSmallGroup - 1);
  lds_double* ex_item = (lds_double*)sm_ex_item + (tid >> 6) * kExactBatch * kExactItemAB;
  lds_u64* ex_res = (lds_u64*)sm_ex_res + (tid >> 6) * kExactBatch;
  const uint64_t gi = static_cast<uint64_t>(blockIdx.x) * (kBlock / kSmallGroup) + static_cast<uint64_t>(tid / kSmallGroup);
  const bool in_range = gi < A.n;
  const uint64_t li = in_range ? gi : (A.n - 1);

  double st[7 + NJ];
  uint64_t edge;
  int step;
  fetch_state<NJ, false, false>(A, li, st, edge, step);
  // the joints' cos/sin rows of the model kernel, all loads in flight at once
  double cs[2 * NJ];
#pragma unroll
  for (int r = 0; r < 2 * NJ; ++r) cs[r] = A.full_rec[static_cast<uint64_t>(r) * A.full_rec_ld + li];
  double R[9], p[3] = {st[0], st[1], st[2]};
  quat_to_mat(st[3], st[4], st[5], st[6], R);
  const double* __restrict__ sc = A.full_rec + static_cast<uint64_t>(2 * NJ) * A.full_rec_ld;
  const bool finite = sc[static_cast<uint64_t>(kFullRecFinite) * A.full_rec_ld + li] != 0.0;

  const double mind = distance_small_body<ROBOT, NL>(A, M, X, obs, in_range, finite, sub, R, p, cs, ex_item, ex_res);

  const bool report = in_range && sub == 0;
  const double margin = sc[static_cast<uint64_t>(kFullRecMargin) * A.full_rec_ld + li];
  full_finish(A, gi, report, finite, margin, sc[static_cast<uint64_t>(kFullRecLamMin) * A.full_rec_ld + li],
              sc[static_cast<uint64_t>(kFullRecLamMax) * A.full_rec_ld + li], sc[static_cast<uint64_t>(kFullRecRp) * A.full_rec_ld + li],
              mind, margin >= 0.0);
  // a host caller polling a pinned word instead of synchronising the stream (outputs in pinned memory): every block
  // releases its writes to the system, the last one to arrive raises the word
  if (A.host_done) {
    __syncthreads();
    if (tid == 0) {
      __threadfence_system();
      const unsigned before = atomicAdd(A.fused_done, 1u);
      if (before + 1u == A.small_target) {
        __threadfence_system();
        *A.host_done = A.fused_seq;
      }
    }
  }
}

// The full path of a few states in ONE launch (mvk_is_valid, clearance queries): a block takes two states and gives
// each a wave for the distance stage (as above, cos/sin taken here) and a wave for walk 1 in full (model_update with
// its detail outputs, every lane the same state); the halves meet in LDS for verdict, band and statistics.  Saves the
// model kernel's launch and the hand-off rows.
template <int ROBOT, int NL>
__global__ void __launch_bounds__(kBlock)
full_small_kernel(const EvalArgs A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
                  const DevObs* __restrict__ obs) {
  using S = Shape<ROBOT, NL>;
  constexpr int NJ = S::kJoint;
  __shared__ double sm_cs[2 * NJ * kBlock];
  __shared__ double sm_ex_item[(kBlock / 64) * kExactBatch * kExactItemAB];
  __shared__ unsigned long long sm_ex_res[(kBlock / 64) * kExactBatch];
  __shared__ double sm_model[2][6];  // per state of the block: margin, lam_min, lam_max, rp margin, finite
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, slot = wv & 1;
  const bool model_wave = wv >= 2;
  lds_double* ex_item = (lds_double*)sm_ex_item + wv * kExactBatch * kExactItemAB;
  lds_u64* ex_res = (lds_u64*)sm_ex_res + wv * kExactBatch;
  const uint64_t gi = static_cast<uint64_t>(blockIdx.x) * 2 + static_cast<uint64_t>(slot);
  const bool in_range = gi < A.n;
  const uint64_t li = in_range ? gi : (A.n - 1);
  double st[7 + NJ];
  uint64_t edge;
  int step;
  fetch_state<NJ, false, true, true>(A, li, st, edge, step);
  double mind = INFINITY;
  if (model_wave) {
    Smem sm{(lds_double*)sm_cs, nullptr};
    StateOut o;
    double thrust[NL];
    double Rw[9];
    bool finite;
    model_update<ROBOT, NL, true>(A, M, X, st, sm, li, o, Rw, finite, thrust);
    if (lane == 0) {
      if (in_range) {
        if (A.fc_t_min) A.fc_t_min[gi] = o.fc_t_min;
        if (A.fc_rp_min) A.fc_rp_min[gi] = o.fc_rp_min;
        if (A.rp_margin) A.rp_margin[gi] = o.rp_margin;
        if (A.thrust) {
#pragma unroll
          for (int i = 0; i < NL; ++i) A.thrust[i * A.ld + gi] = thrust[i];
        }
      }
      sm_model[slot][0] = o.margin; sm_model[slot][1] = o.lam_min; sm_model[slot][2] = o.lam_max;
      sm_model[slot][3] = o.rp_margin; sm_model[slot][4] = finite ? 1.0 : 0.0;
    }
  } else {
    double R[9], p[3] = {st[0], st[1], st[2]};
    quat_to_mat(st[3], st[4], st[5], st[6], R);
    bool finite = true;  // as model_update decides it
#pragma unroll
    for (int k = 0; k < 7 + NJ; ++k) finite = finite && isfinite(st[k]);
#pragma unroll
    for (int k = 0; k < 9; ++k) finite = finite && isfinite(R[k]);
    double cs[2 * NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) sincos_cw(st[7 + j], cs[2 * j + 1], cs[2 * j]);
    mind = distance_small_body<ROBOT, NL>(A, M, X, obs, in_range, finite, lane, R, p, cs, ex_item, ex_res);
  }
  __syncthreads();
  const double margin = sm_model[slot][0];
  const bool finite = sm_model[slot][4] != 0.0;
  full_finish(A, gi, !model_wave && in_range && lane == 0, finite, margin, sm_model[slot][1], sm_model[slot][2], sm_model[slot][3], mind, margin >= 0.0);
  if (A.host_done) {
    __syncthreads();
    if (tid == 0) {
      __threadfence_system();
      const unsigned before = atomicAdd(A.fused_done, 1u);
      if (before + 1u == A.small_target) {
        __threadfence_system();
        *A.host_done = A.fused_seq;
      }
    }
  }
}


// Collision-avoidance constraint rows: differential_kinematics CollisionAvoidance::getConstraint
// (src/constraint/collision_avoidance.cpp:102-277) for a batch of states.  One lane per state;
// per collision primitive the minimum distance over the world boxes with nearest points (:147-153),
// the normal from the environment to the robot (:255), the row n^T J_lin over
// [root v (world), root w (world), planned joints] (:259-260) and the damped bound (:262-265).
#ifndef MVK_AVOID_WAVES
#define MVK_AVOID_WAVES 2
#endif
#ifndef MVK_AVOID_RING   // 1: the cell walks go through the wave's ring (two passes); 0: every lane walks its own (one pass)
#define MVK_AVOID_RING 0  // (measured, profiles/r03/avoid_flat_drain.txt: the one pass is 2-3 % ahead once the walks stop early)
#endif
template <int ROBOT, int NL>
__global__ void __launch_bounds__(kBlock, MVK_AVOID_WAVES)  // as the full kernel: occupancy over spills
avoid_kernel(const EvalArgs A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
             const DevObs* __restrict__ obs) {
  using S = Shape<ROBOT, NL>;
  constexpr int NJ = S::kJoint, W = 6 + NJ;
  __shared__ double sm_cs[2 * NJ * kBlock];
  __shared__ unsigned short sm_list[kListCap * kBlock];
  __shared__ double sm_ex_item[(kBlock / 64) * kCpBatch * kExactItem];
  __shared__ unsigned long long sm_ex_res[(kBlock / 64) * kCpBatch];
  __shared__ double sm_ex_s2[(kBlock / 64) * 24 * kCpBatch];
  __shared__ unsigned sm_ex_win[(kBlock / 64) * kCpBatch];
  // broadphase ring of each wave (dist_push / dist_pop); during a drain the same words hold the owners' nearest points
  __shared__ __align__(8) unsigned sm_dring[(kBlock / 64) * kDRingWords * kRing];
  __shared__ unsigned sm_dcnt[kBlock], sm_upper[kBlock], sm_gob[kBlock], sm_rob[kBlock], sm_emin[kBlock];
  __shared__ unsigned long long sm_mres[kBlock], sm_gmin[kBlock];
  static_assert(kDRingWords * kRing * 4 >= 6 * 64 * 8, "the ring holds [6][64] doubles");
  static_assert(24 * kCpBatch >= 64, "the batch buffer holds [64] keys");
  const int wv = threadIdx.x >> 6;
  Smem sm{(lds_double*)sm_cs, (lds_ushort*)sm_list, (lds_double*)sm_ex_item + wv * kCpBatch * kExactItem, (lds_u64*)sm_ex_res + wv * kCpBatch,
          (lds_double*)sm_ex_s2 + wv * 24 * kCpBatch, (lds_uint*)sm_ex_win + wv * kCpBatch};
  sm.upper = (lds_float*)sm_upper;
  sm.gcs = nullptr;
  sm.dring = (lds_uint*)sm_dring + wv * (kDRingWords * kRing);
  sm.dcnt = (lds_uint*)sm_dcnt + wv * 64;
  sm.mres = (lds_u64*)sm_mres + wv * 64;
  sm.gmin = (lds_u64*)sm_gmin + wv * 64;
  lds_uint* gob = (lds_uint*)sm_gob + wv * 64;        // flat drain: the owner's best box id
  lds_uint* rob = (lds_uint*)sm_rob + wv * 64;        // ... a round's best box id per owner
  lds_uint* kmask = (lds_uint*)sm_emin + wv * 64;     // ... the list entries an owner keeps after the centre bound, one bit each
  static_assert(kListCap <= 32, "one mask bit per list entry");
  // the list index of the j-th kept entry
  auto nth_kept = [](unsigned m, unsigned j) {
    unsigned pos = 0u;
#pragma unroll
    for (unsigned w = 16u; w >= 1u; w >>= 1) {
      const unsigned c = static_cast<unsigned>(__popc((m >> pos) & ((1u << w) - 1u)));
      if (j >= c) { pos += w; j -= c; }
    }
    return pos;
  };
  unsigned head = 0, qn = 0;  // the wave's ring (wave-uniform)
  const int tid = threadIdx.x, lane = tid & 63;
  const uint64_t gi = static_cast<uint64_t>(blockIdx.x) * kBlock + tid;
  const bool in_range = gi < A.n;
  const uint64_t li = in_range ? gi : (A.n - 1);

  double st[7 + NJ];
  uint64_t edge;
  int step;
  fetch_state<NJ, false, true>(A, li, st, edge, step);
  double Rw[9];
  quat_to_mat(st[3], st[4], st[5], st[6], Rw);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    double sj, cj;
    sincos_cw(st[7 + j], sj, cj);
    sm.cs[(2 * j) * kBlock + tid] = cj;
    sm.cs[(2 * j + 1) * kBlock + tid] = sj;
  }
  double R[9], p[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = Rw[k];
  p[0] = st[0]; p[1] = st[1]; p[2] = st[2];
  bool ok = true;
  PROF_ACC_DECL;
  const bool use_grid = A.dgrid.start != nullptr;
  const bool grouped = use_grid && X->n_grp > 0;  // (the sweep over all boxes of small scenes keeps one primitive at a time)
  const bool f32_safe = fmax(fabs(st[0]), fmax(fabs(st[1]), fabs(st[2]))) + X->reach <= A.f32_range;
#pragma unroll 1
  for (int k = 0; k < S::kSeg; ++k) {
    if (k > 0) advance_frame(M, sm, tid, k, R, p);
    const int gfirst = grouped ? X->seg_grp_begin[k] : X->seg_prim_begin[k], glast = grouped ? X->seg_grp_begin[k + 1] : X->seg_prim_begin[k + 1];
#pragma unroll 1
    for (int g = gfirst; g < glast; ++g) {
      // One cell walk per GROUP of co-located primitives (DevExtras::grp_begin, as in the full path): about the
      // leader's centre, with the radius that keeps the list complete for every member (grp_rq); every member then
      // drains that one list -- its own bound removes what the wider radius let in.
      const int qb = grouped ? X->grp_begin[g] : g, qe = grouped ? X->grp_begin[g + 1] : g + 1;
      double pcl[3];
      prim_pose<S::kGimbal, false>(M, X->prim_order[grouped ? X->grp_lead[g] : g], R, p, pcl, nullptr);
      const double rq = grouped ? X->grp_rq[g] : X->prim_bound[X->prim_order[qb]];
      PROF_T0(t_c);
      int cnt = 0;
      int e_near = 0;  // the list index of the candidate nearest by centre distance, where the broadphase knows it
      unsigned cell_b = 0, cell_e = 0;
      bool from_grid = false;
      const uint16_t* __restrict__ cell_items = A.dgrid.items;
      if (use_grid) {
        // The cell of the primitive's centre lists every box within the grid's radius.  Its walk is shared out over
        // the wave through the ring, as in the full path (dist_push / dist_pop): first the smallest centre distance
        // of the list -- the primitive's own nearest box sets the threshold here, no bound is known beforehand --
        // then the boxes within bounding radius + that distance join the lane's candidate list.
        const bool on = in_range && f32_safe;
#if MVK_AVOID_RING
        lds_uint* up = reinterpret_cast<lds_uint*>(sm.upper) + tid;
        *up = 0x7F800000u;  // +inf
        const bool inside = grid_cell_in(A.dgrid, pcl, on, cell_b, cell_e);
        dist_push(sm, inside, pcl, cell_b, cell_e, 0u, kRangeCell, 0.0f, head, qn);
#pragma unroll 1
        while (qn > 0u) dist_pop<false>(A, A.obsf, sm, head, qn);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float best2f = __uint_as_float(*up);
        const double upd = static_cast<double>(sqrtf(best2f)) * 1.000001 + static_cast<double>(A.f32_delta);
        // (+inf: nothing listed; beyond the radius the list is not complete for the threshold)
        const bool answered = inside && best2f < INFINITY && rq + upd + 2.0 * static_cast<double>(A.f32_delta) <= A.dgrid_radius;
        const float rbf = static_cast<float>(rq) * 1.000001f + 2.0f * A.f32_delta;
        const float tf = (rbf + sqrtf(best2f) * 1.000001f) * 1.000001f;
        sm.dcnt[lane] = 0u;
        dist_push(sm, answered, pcl, cell_b, cell_e, 0u, kRangeCell, tf * tf, head, qn);
#pragma unroll 1
        while (qn > 0u) dist_pop<true>(A, A.obsf, sm, head, qn);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bool scan = in_range && !answered;
        if (answered) cnt = static_cast<int>(sm.dcnt[lane]);
#else
        // (every lane walks its own cell's list, once: the threshold follows the running minimum)
        bool scan = in_range;
        if (grid_collect_nearest(A, A.dgrid, A.dgrid_radius, sm, on, pcl, rq, cnt, cell_b, cell_e, &e_near)) scan = false;
#endif
        if (A.fgrid.start != nullptr && __any(scan && f32_safe)) {  // too far for the grid's radius: the far grid's list
          int c3 = 0;
          unsigned b3 = 0, e3 = 0;
          const bool in_far = grid_cell_in(A.fgrid, pcl, scan && f32_safe, b3, e3);
          far_for_lanes_f32(A, sm, in_far, pcl, rq, b3, e3, c3);
          if (in_far) {
            scan = false;
            cnt = c3; cell_b = b3; cell_e = e3;
            cell_items = A.fgrid.items;
            e_near = 0;
          }
        }
        from_grid = !scan;
        double b2 = INFINITY;
        int c2 = 0;
        if (scan) e_near = 0;
        sweep_for_lanes_f32(A, sm, scan && f32_safe, pcl, rq, INFINITY, cnt);
        const bool slow = scan && !f32_safe;
        sweep_for_lanes(A, obs, sm, slow, pcl, rq, INFINITY, b2, c2);
#ifdef MVK_PROF_COUNTS
        PROF_LANES(37, scan);
#endif
        if (slow) cnt = c2;
      } else {
        auto centre_dist2 = [&](const DevObs& O) {
          double t[3];
          to_box_frame(O, pcl, t);
          return point_box_dist2(t, O.h);
        };
        // the primitive's origin lies inside it: the nearest box to the origin bounds the separation from above
        double best2 = INFINITY;
#pragma unroll 2
        for (int ob = 0; ob < A.n_obs; ++ob) best2 = fmin(best2, centre_dist2(obs[ob]));
        const double thr = rq + sqrt(best2);
        const double thr2 = thr * thr * (1.0 + 1e-12);
#pragma unroll 2
        for (int ob = 0; ob < A.n_obs; ++ob) {
          if (centre_dist2(obs[ob]) <= thr2) {
            if (cnt < kListCap) sm.list[cnt * kBlock + tid] = static_cast<unsigned short>(ob);
            ++cnt;
          }
        }
      }
      const bool overflow = cnt > kListCap;   // as in the full kernel: the cell's whole list, or every box
      const bool from_cell = overflow && from_grid && in_range;
      const int todo = (use_grid && !in_range) ? 0 : (overflow ? (from_cell ? static_cast<int>(cell_e - cell_b) * kGridGroup : A.n_obs) : cnt);
      auto cand = [&](int e) -> int {  // kGridNone: padding, skip
        return !overflow ? sm.list[e * kBlock + tid] : (from_cell ? cell_items[static_cast<size_t>(cell_b) * kGridGroup + e] : e);
      };
      PROF_CLK(0, t_c);
#ifdef MVK_PROF_COUNTS
      PROF_LANES(36, in_range); PROF_LANES(38, in_range && overflow); PROF_SUM(39, in_range ? cnt : 0);
#endif
#pragma unroll 1
      for (int qm = qb; qm < qe; ++qm) {
      const int q = X->prim_order[qm];
      const int type = M->prim_type[q];
      double pc[3], PR[9];
      prim_pose<S::kGimbal, true>(M, q, R, p, pc, PR);
      const double rb = X->prim_bound[q];
      const double d0 = M->prim_dim[q][0], d1 = M->prim_dim[q][1], d2 = M->prim_dim[q][2];
      PROF_T0(t_d);
      double best = INFINITY, br[3] = {NAN, NAN, NAN}, be[3] = {NAN, NAN, NAN};
      int bi = -1;
      // Flat drain (no lane's list overflowed, the common case): a primitive's lists have very different lengths --
      // ten candidates on average, thirty in the longest of 64 -- so instead of every lane scanning its own, the wave
      // numbers the (lane, candidate) pairs through and takes them 64 at a time: a lane fetches the owner's pose
      // (cross-lane reads), computes the pair, and the results of a round merge into their owners' records in LDS --
      // (distance key, box id, nearest points), the minimum of (distance, box id) as the oracle keeps it.
      const bool flat = use_grid && !__any(in_range && overflow);
      const int wave_tid = tid - lane;
      lds_u64* gkey = sm.mres;      // the owner's best distance, as an ordered key
      lds_u64* rkey = sm.gmin;      // a round's best per owner
      lds_double* pts = reinterpret_cast<lds_double*>(sm.dring);  // [6][64] world points of the best pair (the ring is idle here)
      // merge one result per lane (`has`) into its owner's record; several lanes may serve one owner in a round
      auto merge = [&](bool has, int src, double d, int ob, const double wr[3], const double we[3]) {
        if (!__any(has)) return;
        rkey[lane] = ~0ull;
        rob[lane] = 0xFFFFFFFFu;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long k = ord_key(d + 0.0);  // (-0 -> +0: one key per value)
        if (has) __hip_atomic_fetch_min(&rkey[src], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool top = has && rkey[src] == k;
        if (top) __hip_atomic_fetch_min(&rob[src], static_cast<unsigned>(ob), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (top && rob[src] == static_cast<unsigned>(ob)) {  // the round's winner for this owner: one lane
          const unsigned long long gk = gkey[src];
          if (k < gk || (k == gk && static_cast<unsigned>(ob) < gob[src])) {
            gkey[src] = k;
            gob[src] = static_cast<unsigned>(ob);
#pragma unroll
            for (int c = 0; c < 3; ++c) { pts[c * 64 + src] = wr[c]; pts[(3 + c) * 64 + src] = we[c]; }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      };
      auto to_world = [&](const DevObs& O, const double pr[3], const double pq[3], double wr[3], double we[3]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {  // box frame -> world (NaN points stay NaN)
          wr[c] = fma(O.R[3 * c], pr[0], fma(O.R[3 * c + 1], pr[1], fma(O.R[3 * c + 2], pr[2], O.c[c])));
          we[c] = fma(O.R[3 * c], pq[0], fma(O.R[3 * c + 1], pq[1], fma(O.R[3 * c + 2], pq[2], O.c[c])));
        }
      };
      unsigned T = 0u;
      // the owner of pair i of the wave's numbering, and its index in the owner's list
      auto owner_of = [&](unsigned i, bool act, int& src, unsigned& e) {
        unsigned lo = 0u, hi = 63u;  // the first lane whose inclusive count exceeds i
#pragma unroll
        for (int st_ = 0; st_ < 6; ++st_) {
          const unsigned mid = (lo + hi) >> 1;
          if (sm.dcnt[mid] > i) hi = mid; else lo = mid + 1u;
        }
        src = act ? static_cast<int>(lo) : lane;
        e = act ? i - (src > 0 ? sm.dcnt[src - 1] : 0u) : 0u;
      };
      auto number_pairs = [&](unsigned mine) {  // inclusive counts into dcnt, the wave's total into T
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned v = __shfl_up(incl, o);
          if (lane >= o) incl += v;
        }
        T = __shfl(incl, 63);
        sm.dcnt[lane] = incl;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      };
      if (flat) {
        gkey[lane] = ord_key(INFINITY);
        gob[lane] = 0xFFFFFFFFu;
#pragma unroll
        for (int c = 0; c < 6; ++c) pts[c * 64 + lane] = NAN;
        // Round 0: every lane its own candidate nearest by centre distance (the broadphase knows it), in full -- for a
        // box the separating-axis gap and, when separated, the exact closest points.  Its distance is close to the
        // primitive's minimum, and a cheap bound then removes most of the other candidates before any costly test.
        const int e0 = todo > 0 ? min(e_near, todo - 1) : -1;
        PROF_T0(t_f);
        if (__any(e0 >= 0)) {
          if (e0 >= 0) {
            const int ob = sm.list[e0 * kBlock + tid];
            const DevObs O = obs[ob];
            double t[3], pr[3] = {NAN, NAN, NAN}, pq[3] = {NAN, NAN, NAN}, wr[3], we[3], d;
            to_box_frame(O, pc, t);
            if (type == MVK_PRIM_BOX) {
              const double a_[3] = {d0, d1, d2};
              double Rr[9];
              box_rel(O, PR, Rr);
              d = box_box_sat<true>(t, Rr, a_, O.h, INFINITY);  // (<= 0: in contact, no points)
              if (d > 0.0) d = box_box_exact_cp(t, Rr, a_, O.h, pr, pq);
            } else {
              d = narrow_cp(type, PR, d0, d1, d2, O, t, pr, pq, INFINITY);
            }
            to_world(O, pr, pq, wr, we);
            gkey[lane] = ord_key(d + 0.0);
            gob[lane] = static_cast<unsigned>(ob);
#pragma unroll
            for (int c = 0; c < 3; ++c) { pts[c * 64 + lane] = wr[c]; pts[(3 + c) * 64 + lane] = we[c]; }
          }
        }
        PROF_CLK(4, t_f);
        kmask[lane] = 0u;
        number_pairs(static_cast<unsigned>(todo));
        // the others: a pair whose centre distance less the primitive's bounding radius exceeds the owner's distance
        // cannot realise the minimum (nor tie with it); the survivors are marked in the owner's mask (one bit per list
        // entry: the list itself stays as it is for the other members of the group)
        PROF_T0(t_a);
#pragma unroll 1
        for (unsigned base = 0; base < T; base += 64u) {
          const unsigned i = base + static_cast<unsigned>(lane);
          const bool act = i < T;
          int src; unsigned e;
          owner_of(i, act, src, e);
          double spc[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) spc[c] = __shfl(pc[c], src);
          const int se0 = __shfl(e0, src);
          bool keep = false;
          unsigned ob = 0u;
          if (act && static_cast<int>(e) != se0) {
            ob = sm.list[e * kBlock + wave_tid + src];
            const DevObs& O = obs[ob];
            double t[3];
            to_box_frame(O, spc, t);
            const double lim = (rb + fmax(ord_val(gkey[src]), 0.0)) * (1.0 + 1e-9) + 1e-9;
            keep = point_box_dist2(t, O.h) <= lim * lim;
          }
          if (keep) __hip_atomic_fetch_or(&kmask[src], 1u << e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        PROF_CLK(3, t_a);
        number_pairs(static_cast<unsigned>(__popc(kmask[lane])));
#ifdef MVK_PROF_COUNTS
        PROF_SUM(45, __popc(kmask[lane]));
#endif
      }
      if (type == MVK_PRIM_BOX) {
        // Every separated primitive needs at least one exact closest-point computation (24 edge-to-box closest
        // pairs, ~30x a separating-axis gap), so the order matters: pass A takes the gap of every candidate
        // (contacts are final there) and finds the smallest; its exact distance comes first and prunes most of the
        // rest; pass B scans the others with the gap alone, and the pairs whose gap still undercuts their owner's
        // minimum get the exact code, the wave working on a batch of them together.  The oracle keeps
        // the first minimum in ascending box order: out of order, that is the minimum of (distance, box id).
        const double a[3] = {d0, d1, d2};
        auto take = [&](double d, int ob, const DevObs& O, const double pr[3], const double pq[3]) {
          if (d < best || (d == best && ob < bi)) {
            best = d;
            bi = ob;
#pragma unroll
            for (int c = 0; c < 3; ++c) {  // box frame -> world (NaN points stay NaN)
              br[c] = fma(O.R[3 * c], pr[0], fma(O.R[3 * c + 1], pr[1], fma(O.R[3 * c + 2], pr[2], O.c[c])));
              be[c] = fma(O.R[3 * c], pq[0], fma(O.R[3 * c + 1], pq[1], fma(O.R[3 * c + 2], pq[2], O.c[c])));
            }
          }
        };
        const double nan3[3] = {NAN, NAN, NAN};
        // one (pair, edge): inputs as box_box_exact_cp builds them for its two halves
        auto edge_item = [&](const double iR[9], const double it_[3], const double ih[3], int ed, double y[3], double z[3]) {
          const bool second = ed >= 12;
          double cR[9], cc[3], ca[3], cb[3];
#pragma unroll
          for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) cR[3 * r + c] = second ? iR[3 * c + r] : iR[3 * r + c];
            const double tb = -fma(iR[r], it_[0], fma(iR[3 + r], it_[1], iR[6 + r] * it_[2]));
            cc[r] = second ? tb : it_[r];
            ca[r] = second ? ih[r] : a[r];
            cb[r] = second ? a[r] : ih[r];
          }
          return box_edge_to_box_cp2(second ? ed - 12 : ed, cc, cR, ca, cb, y, z);
        };
        // The pairs that need the exact closest points (`need`: relative pose Rr, t; box ob_n), computed by the whole
        // wave, kCpBatch at a time: (pair, edge) items over all lanes, an LDS atomic min per pair over its 24 squared
        // distances, a second one over the edge indices that reach it (the lowest wins, as in the one-lane form: robot
        // edges before box edges, ascending), and the lane that holds the pair repeats that one edge for its points.
        // Same arithmetic per edge, so the result keeps its bits.  `deliver` hands (distance, box, points) over.
        auto run_cp = [&](bool need, int ob_n, const double Rr[9], const double t[3], auto&& deliver) {
          unsigned long long pend = __ballot(need);
#pragma unroll 1
          while (pend) {
            const int rank = __popcll(pend & ((1ull << lane) - 1ull));
            const bool mine = need && ((pend >> lane) & 1ull) && rank < kCpBatch;
            const int E = min(static_cast<int>(__popcll(pend)), kCpBatch);
            double h[3] = {0.0, 0.0, 0.0};
            if (mine) {
              const DevObs& O = obs[ob_n];
              h[0] = O.h[0]; h[1] = O.h[1]; h[2] = O.h[2];
#pragma unroll
              for (int c = 0; c < 9; ++c) sm.ex_item[rank * kExactItem + c] = Rr[c];
#pragma unroll
              for (int c = 0; c < 3; ++c) { sm.ex_item[rank * kExactItem + 9 + c] = t[c]; sm.ex_item[rank * kExactItem + 12 + c] = h[c]; }
              sm.ex_res[rank] = 0x7FF0000000000000ull;  // +inf
              sm.ex_win[rank] = 0xFFFFFFFFu;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll 1
            for (int it = lane; it < 24 * E; it += 64) {
              const int pr_ = it / 24, ed = it - 24 * pr_;
              double iR[9], it_[3], ih[3], y[3], z[3];
#pragma unroll
              for (int c = 0; c < 9; ++c) iR[c] = sm.ex_item[pr_ * kExactItem + c];
#pragma unroll
              for (int c = 0; c < 3; ++c) { it_[c] = sm.ex_item[pr_ * kExactItem + 9 + c]; ih[c] = sm.ex_item[pr_ * kExactItem + 12 + c]; }
              const double s2 = edge_item(iR, it_, ih, ed, y, z);
              sm.ex_s2[it] = s2;
              __hip_atomic_fetch_min(&sm.ex_res[pr_], static_cast<unsigned long long>(__double_as_longlong(s2)), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll 1
            for (int it = lane; it < 24 * E; it += 64) {
              const int pr_ = it / 24, ed = it - 24 * pr_;
              if (static_cast<unsigned long long>(__double_as_longlong(sm.ex_s2[it])) == sm.ex_res[pr_])
                __hip_atomic_fetch_min(&sm.ex_win[pr_], static_cast<unsigned>(ed), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double dd_ = INFINITY, pr[3] = {NAN, NAN, NAN}, pq[3] = {NAN, NAN, NAN};
            if (mine) {
              const int ed = static_cast<int>(sm.ex_win[rank]);
              double y[3], z[3];
              const double s2 = edge_item(Rr, t, h, ed, y, z);
              if (ed < 12) {  // y on the robot's edge, z on the box (box frame)
#pragma unroll
                for (int c = 0; c < 3; ++c) { pr[c] = y[c]; pq[c] = z[c]; }
              } else {        // y on the box's edge, z on the robot (primitive frame): x_B = Rr x_A + t
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                  pq[c] = fma(Rr[3 * c], y[0], fma(Rr[3 * c + 1], y[1], fma(Rr[3 * c + 2], y[2], t[c])));
                  pr[c] = fma(Rr[3 * c], z[0], fma(Rr[3 * c + 1], z[1], fma(Rr[3 * c + 2], z[2], t[c])));
                }
              }
              dd_ = sqrt(s2);
            }
            deliver(mine, dd_, pr, pq);
            pend &= ~__ballot(mine);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
        };
        if (flat) {
          // what round 0 and the bound left: the separating-axis gap against the owner's distance, contacts merged
          // at once, and the exact closest points for the pairs whose gap still undercuts it
#pragma unroll 1
          for (unsigned base = 0; base < T; base += 64u) {
            const unsigned i = base + static_cast<unsigned>(lane);
            const bool act = i < T;
            int src; unsigned e;
            owner_of(i, act, src, e);
            double spc[3], sPR[9];
#pragma unroll
            for (int c = 0; c < 3; ++c) spc[c] = __shfl(pc[c], src);
#pragma unroll
            for (int c = 0; c < 9; ++c) sPR[c] = __shfl(PR[c], src);
            bool need = false, contact = false;
            double g = INFINITY, Rr[9], t[3];
            int ob = 0;
            rkey[lane] = ~0ull;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (act) {
              ob = sm.list[nth_kept(kmask[src], e) * kBlock + wave_tid + src];
              const DevObs& O = obs[ob];
              to_box_frame(O, spc, t);
              box_rel(O, sPR, Rr);
              const double cut = ord_val(gkey[src]) + kSatSlack;  // (kSatSlack: the minimum does not depend on the order of the pairs)
              // (a positive cutoff: the early exit on the face axes then only ever takes separated pairs)
              g = box_box_sat<true>(t, Rr, a, O.h, fmax(cut, kSatSlack));
              contact = g <= 0.0;
              need = g > 0.0 && g <= cut;
              if (need) __hip_atomic_fetch_min(&rkey[src], ord_key(g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // per owner the pair with the smallest gap first (read before the merges below reuse rkey)
            const bool first = need && rkey[src] == ord_key(g);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            merge(contact, src, g, ob, nan3, nan3);  // in contact: no points
#ifdef MVK_PROF_COUNTS
            PROF_SUM(46, need ? 1 : 0);
#endif
            if (__any(need)) {
              auto deliver = [&](bool mine, double d, const double pr[3], const double pq[3]) {
                double wr[3] = {NAN, NAN, NAN}, we[3] = {NAN, NAN, NAN};
                if (mine) to_world(obs[ob], pr, pq, wr, we);
                merge(mine, src, d, ob, wr, we);
              };
              run_cp(first, ob, Rr, t, deliver);
              // the others are re-tested against the result
              const bool again = need && !first && g <= ord_val(gkey[src]) + kSatSlack;
              run_cp(again, ob, Rr, t, deliver);
            }
          }
        } else {
        double gmin = INFINITY;
        int emin = -1;
        PROF_T0(t_a);
#pragma unroll 1
        for (int e = 0; __any(e < todo); ++e) {
          const int ob = e < todo ? cand(e) : kGridNone;
          if (ob != kGridNone) {
            const DevObs& O = obs[ob];
            double Rr[9], t[3];
            to_box_frame(O, pc, t);
            box_rel(O, PR, Rr);
            const double g = box_box_sat<true>(t, Rr, a, O.h, gmin);
            if (g <= 0.0) take(g, ob, O, nan3, nan3);  // in contact: no points
            else if (g < gmin) { gmin = g; emin = e; }
          }
        }
        PROF_CLK(3, t_a);
        PROF_T0(t_f);
        {
          const bool first = emin >= 0 && gmin <= best + kSatSlack;
          if (__any(first)) {
            if (first) {
              const int ob = cand(emin);
              const DevObs O = obs[ob];
              double Rr[9], t[3], pr[3], pq[3];
              to_box_frame(O, pc, t);
              box_rel(O, PR, Rr);
              const double d = box_box_exact_cp(t, Rr, a, O.h, pr, pq);
              take(d, ob, O, pr, pq);
            }
          }
        }
        PROF_CLK(4, t_f);
        int e = 0;
#pragma unroll 1
        while (__any(e < todo)) {
          bool need = false;
          int ob_n = 0;
          double Rr[9], t[3];
#pragma unroll 1
          while (e < todo && !need) {
            const int ob = e != emin ? cand(e) : kGridNone;
            if (ob != kGridNone) {
              const DevObs& O = obs[ob];
              to_box_frame(O, pc, t);
              box_rel(O, PR, Rr);
              const double g = box_box_sat<true>(t, Rr, a, O.h, best + kSatSlack);
              if (g > 0.0 && g <= best + kSatSlack) {  // (kSatSlack: the minimum does not depend on the order of the pairs)
                need = true;
                ob_n = ob;
              }
            }
            ++e;
          }
          run_cp(need, ob_n, Rr, t, [&](bool mine, double d, const double pr[3], const double pq[3]) {
            if (mine) take(d, ob_n, obs[ob_n], pr, pq);
          });
        }
        }
      } else if (flat) {
        // closed-form primitives: one round per 64 pairs, distance and nearest points at once
#pragma unroll 1
        for (unsigned base = 0; base < T; base += 64u) {
          const unsigned i = base + static_cast<unsigned>(lane);
          const bool act = i < T;
          int src; unsigned e;
          owner_of(i, act, src, e);
          double spc[3], sPR[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int c = 0; c < 3; ++c) spc[c] = __shfl(pc[c], src);
#pragma unroll
          for (int c = 0; c < 3; ++c) sPR[3 * c + 2] = __shfl(PR[3 * c + 2], src);  // (a capsule's axis: all narrow_cp() reads)
          double d = INFINITY, wr[3] = {NAN, NAN, NAN}, we[3] = {NAN, NAN, NAN};
          int ob = 0;
          if (act) {
            ob = sm.list[nth_kept(kmask[src], e) * kBlock + wave_tid + src];
            const DevObs O = obs[ob];
            double t[3], pr[3], pq[3];
            to_box_frame(O, spc, t);
            d = narrow_cp(type, sPR, d0, d1, d2, O, t, pr, pq, INFINITY);
            to_world(O, pr, pq, wr, we);
          }
          merge(act, src, d, ob, wr, we);
        }
      } else {
#pragma unroll 1
        for (int e = 0; __any(e < todo); ++e) {
          const int ob = e < todo ? cand(e) : kGridNone;
          if (ob != kGridNone) {
            const DevObs O = obs[ob];
            double t[3], pr[3], pq[3];
            to_box_frame(O, pc, t);
            const double d = narrow_cp(type, PR, d0, d1, d2, O, t, pr, pq, best);
            if (d < best || (d == best && ob < bi)) {  // the first minimum in ascending box order, as in the oracle, whatever the list's order
              best = d;
              bi = ob;
#pragma unroll
              for (int c = 0; c < 3; ++c) {  // box frame -> world
                br[c] = fma(O.R[3 * c], pr[0], fma(O.R[3 * c + 1], pr[1], fma(O.R[3 * c + 2], pr[2], O.c[c])));
                be[c] = fma(O.R[3 * c], pq[0], fma(O.R[3 * c + 1], pq[1], fma(O.R[3 * c + 2], pq[2], O.c[c])));
              }
            }
          }
        }
      }
      if (flat) {  // the owner's record
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned go = gob[lane];
        if (go != 0xFFFFFFFFu) {
          best = ord_val(gkey[lane]);
          bi = static_cast<int>(go);
#pragma unroll
          for (int c = 0; c < 3; ++c) { br[c] = pts[c * 64 + lane]; be[c] = pts[(3 + c) * 64 + lane]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      PROF_CLK(1, t_d);
      PROF_T0(t_r);
      const bool usable = bi >= 0 && best > 0.0;
      ok = ok && (bi < 0 || best > 0.0);
      // normal from the environment to the robot, and the row over the chain up to this segment
      double nrm[3] = {br[0] - be[0], br[1] - be[1], br[2] - be[2]};
      const double inv = 1.0 / sqrt(fma(nrm[0], nrm[0], fma(nrm[1], nrm[1], nrm[2] * nrm[2])));
#pragma unroll
      for (int c = 0; c < 3; ++c) nrm[c] = usable ? nrm[c] * inv : 0.0;
      double row[W];
#pragma unroll
      for (int c = 0; c < W; ++c) row[c] = 0.0;
      if (A.av_full_body) {
        const double rx = br[0] - st[0], ry = br[1] - st[1], rz = br[2] - st[2];
        row[0] = nrm[0]; row[1] = nrm[1]; row[2] = nrm[2];
        row[3] = usable ? fma(ry, nrm[2], -rz * nrm[1]) : 0.0;
        row[4] = usable ? fma(rz, nrm[0], -rx * nrm[2]) : 0.0;
        row[5] = usable ? fma(rx, nrm[1], -ry * nrm[0]) : 0.0;
      }
      {
        // joint s rotates about column axis(s) of frame s-1 through the origin of frame s
        double Rj[9], pj[3];
#pragma unroll
        for (int c = 0; c < 9; ++c) Rj[c] = Rw[c];
        pj[0] = st[0]; pj[1] = st[1]; pj[2] = st[2];
        static_for<1, S::kSeg>([&](auto sc_) {
          constexpr int s = decltype(sc_)::value;
          if (s <= k) {
            double pn[3];
            xform(Rj, pj, M->seg_t[s][0], M->seg_t[s][1], M->seg_t[s][2], pn);
            pj[0] = pn[0]; pj[1] = pn[1]; pj[2] = pn[2];
            constexpr int ax = S::axis(s);
            const double ax_w[3] = {Rj[ax], Rj[3 + ax], Rj[6 + ax]};
            const double rx = br[0] - pj[0], ry = br[1] - pj[1], rz = br[2] - pj[2];
            const double vx = fma(ax_w[1], rz, -ax_w[2] * ry), vy = fma(ax_w[2], rx, -ax_w[0] * rz),
                         vz = fma(ax_w[0], ry, -ax_w[1] * rx);
            row[6 + S::joint(s)] = usable ? fma(nrm[0], vx, fma(nrm[1], vy, nrm[2] * vz)) : 0.0;
            rot_cols<ax>(Rj, sm.cs[(2 * S::joint(s)) * kBlock + tid], sm.cs[(2 * S::joint(s) + 1) * kBlock + tid]);
          }
        });
      }
      if (in_range) {
        if (A.av_dist) A.av_dist[q * A.ld + gi] = best;
        if (A.av_box) A.av_box[q * A.ld + gi] = bi;
        if (A.av_lb)
          A.av_lb[q * A.ld + gi] = (usable && best < A.av_range) ? -A.av_gain * ((best - A.av_forbid) / (A.av_range - A.av_forbid)) : -1e6;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (A.av_p_env) A.av_p_env[(3 * q + c) * A.ld + gi] = usable ? be[c] : NAN;
          if (A.av_p_robot) A.av_p_robot[(3 * q + c) * A.ld + gi] = usable ? br[c] : NAN;
        }
        if (A.av_row) {
#pragma unroll
          for (int c = 0; c < W; ++c) A.av_row[(static_cast<uint64_t>(W) * q + c) * A.ld + gi] = row[c];
        }
      }
      PROF_CLK(2, t_r);
      }  // members
    }
  }
  PROF_ACC_FLUSH(40);
  PROF_WAVE(47);
  if (in_range && A.av_ok) A.av_ok[gi] = ok ? 1 : 0;
}

// Flag path for a handful of states (OMPL's isValid, one state per call): fused_kernel gives the state one lane, which
// runs walk 1 and the collision walk one after the other -- 19 us for a single state.  Here a state has a block of
// two waves that work side by side: wave 1 runs walk 1 (model_update, every lane the same state: a chain of dependent
// fp64 that lanes cannot share), wave 0 the collision test as distance_small_kernel lays it out -- lane (qi, part)
// walks the chain to the segment of primitive qi and tests every K-th box of that segment's cell of the flag grid
// with exact_pair (fp64; the fp32 screen of the large kernels only ever removes pairs that cannot touch, so the
// verdict is the same OR over the same exact tests).  Verdict to pinned host memory, done word by the last block.
// EDGE: the states are the steps of a few edges (OMPL's checkMotion, one edge per call): block gi takes lane gi of
// the flattened steps, an invalid step goes into the edge's tag word as in fused_kernel, the last block turns the tags
// into first_invalid in the pinned answer block.
template <int ROBOT, int NL, bool EDGE>
__global__ void __launch_bounds__(128)
flag_small_kernel(const EvalArgs A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
                  const DevObs* __restrict__ obs) {
  using S = Shape<ROBOT, NL>;
  constexpr int NJ = S::kJoint;
  __shared__ double sm_cs[2 * NJ * kBlock];  // model_update's cos/sin slots (pitch kBlock)
  __shared__ unsigned sm_verdict[3];         // finite, stable, hit
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint64_t gi = blockIdx.x;            // one state per block (gridDim.x == A.n)
  if (tid < 3) sm_verdict[tid] = 0u;
  __syncthreads();

  double st[7 + NJ];
  uint64_t edge;
  int step;
  fetch_state<NJ, EDGE, true, true>(A, gi, st, edge, step);
  if (wv == 1) {
    Smem sm{(lds_double*)sm_cs, nullptr};
    StateOut o;
    double Rw[9];
    bool finite;
    model_update<ROBOT, NL, false>(A, M, X, st, sm, gi, o, Rw, finite, nullptr);
    if (lane == 0) {
      sm_verdict[0] = finite ? 1u : 0u;
      sm_verdict[1] = o.stable ? 1u : 0u;
    }
  } else if (A.do_coll && A.n_obs > 0) {
    double cs[2 * NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) sincos_cw(st[7 + j], cs[2 * j + 1], cs[2 * j]);
    double R[9], p[3] = {st[0], st[1], st[2]};
    quat_to_mat(st[3], st[4], st[5], st[6], R);
    const int P = M->n_prim;
    const int K = 64 / P;
    const bool active = lane < P * K;
    const int qi = lane % P, part = lane / P;
    const int q = X->prim_order[qi];
    int myk = 0;
#pragma unroll 1
    for (int k = 1; k < S::kSeg; ++k) myk = qi >= X->seg_prim_begin[k] ? k : myk;
    static_for<1, S::kSeg>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if (k <= myk) {
        double pn[3];
        xform(R, p, M->seg_t[k][0], M->seg_t[k][1], M->seg_t[k][2], pn);
        p[0] = pn[0]; p[1] = pn[1]; p[2] = pn[2];
        rot_cols<S::axis(k)>(R, cs[2 * S::joint(k)], cs[2 * S::joint(k) + 1]);
      }
    });
    // the segment's cell of the flag grid lists every box within the largest segment radius of the sphere's centre
    double sc[3];
    xform(R, p, X->seg_sph[myk][0], X->seg_sph[myk][1], X->seg_sph[myk][2], sc);
    unsigned gb, ge;
    grid_cell(A.grid, sc, active, gb, ge);
    const int n_items = static_cast<int>(ge - gb) * kGridGroup;
    const uint16_t* __restrict__ ids = A.grid.items + static_cast<size_t>(gb) * kGridGroup;
    bool hit = false;
#pragma unroll 1
    for (int e = part; e < n_items && !hit; e += K) {
      const int ob = ids[e];
      if (ob != kGridNone) hit = exact_pair<S::kGimbal>(M, X, q, R, p, obs[ob]);
    }
    if (__any(hit) && lane == 0) sm_verdict[2] = 1u;
  }
  __syncthreads();
  const bool valid = sm_verdict[0] != 0u && (!A.do_stab || sm_verdict[1] != 0u) && sm_verdict[2] == 0u;
  if (!EDGE) {
    if (tid == 0) {
      A.host_flags[gi] = valid ? 1 : 0;
      __threadfence_system();
      const unsigned before = atomicAdd(A.fused_done, 1u);
      if (before + 1u == A.small_target) {
        __threadfence_system();
        *A.host_done = A.fused_seq;
      }
    }
  } else {
    __shared__ unsigned sm_last;
    if (tid == 0) {
      if (!valid) mark_invalid_step(A, edge, step);
      sm_last = __hip_atomic_fetch_add(A.fused_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == A.small_target ? 1u : 0u;
    }
    __syncthreads();
    if (!sm_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (uint64_t e = tid; e < A.n_edges; e += 128) {
      const unsigned v = __hip_atomic_load(A.first_tag + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int nd_e = A.nd ? (A.nd[e] < 1 ? 1 : A.nd[e]) : A.nd_uniform;
      A.host_first[e] = (v >> 16) == (A.fused_seq & 0xFFFFu) ? static_cast<int>(0xFFFFu - (v & 0xFFFFu)) : nd_e + 1;
    }
    __syncthreads();
    if (tid == 0) {
      __threadfence_system();
      *A.host_done = A.fused_seq;
    }
  }
}

// The avoidance rows for the one state a QP iteration of the differential-kinematics planner asks about
// (collision_avoidance.cpp:102-277 runs once per iteration), and for small batches: a wave per state, as
// distance_small_kernel.  Lane (qi, part) scans every K-th box of the candidates of primitive qi, K = 64 / P; the K
// partial results of a primitive are merged through LDS by (distance, box id), which is the oracle's "first minimum
// in ascending box order"; lane (qi, 0) then builds the row.  Candidates: the avoidance grid's cell list of the
// primitive's centre, complete within its radius; a primitive whose nearest listed box is farther than that (or
// that has none) goes over all boxes.  Exact closest points by the whole wave (exact_cp_wave_ab).
template <int ROBOT, int NL>
__global__ void __launch_bounds__(kBlock)
avoid_small_kernel(const EvalArgs A, const mvk_model* __restrict__ M, const DevExtras* __restrict__ X,
                   const DevObs* __restrict__ obs) {
  using S = Shape<ROBOT, NL>;
  constexpr int NJ = S::kJoint, W = 6 + NJ;
  constexpr int kRed = 8;  // doubles per lane in the merge: distance, box id, the two points
  __shared__ double sm_item[(kBlock / 64) * kCpBatch * kExactItemAB];
  __shared__ unsigned long long sm_res[(kBlock / 64) * kCpBatch];
  __shared__ double sm_s2[(kBlock / 64) * 24 * kCpBatch];
  __shared__ unsigned sm_win[(kBlock / 64) * kCpBatch];
  __shared__ double sm_red[(kBlock / 64) * 64 * kRed];
  constexpr int kSweepKeep = 24;  // boxes a lane keeps per round of the sweep over all boxes
  __shared__ unsigned short sm_keep[(kBlock / 64) * kSweepKeep * 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  lds_ushort* keep = (lds_ushort*)sm_keep + wv * kSweepKeep * 64;
  lds_double* ex_item = (lds_double*)sm_item + wv * kCpBatch * kExactItemAB;
  lds_u64* ex_res = (lds_u64*)sm_res + wv * kCpBatch;
  lds_double* ex_s2 = (lds_double*)sm_s2 + wv * 24 * kCpBatch;
  lds_uint* ex_win = (lds_uint*)sm_win + wv * kCpBatch;
  lds_double* red = (lds_double*)sm_red + wv * 64 * kRed;
  const uint64_t gi = static_cast<uint64_t>(blockIdx.x) * (kBlock / 64) + static_cast<uint64_t>(wv);
  const bool in_range = gi < A.n;
  const uint64_t li = in_range ? gi : (A.n - 1);

  double st[7 + NJ];
  uint64_t edge;
  int step;
  fetch_state<NJ, false, true>(A, li, st, edge, step);
  double Rw[9];
  quat_to_mat(st[3], st[4], st[5], st[6], Rw);
  double cs[2 * NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) sincos_cw(st[7 + j], cs[2 * j + 1], cs[2 * j]);

  const int P = M->n_prim;
  const int K = 64 / P;  // >= 2 (P <= MVK_MAX_PRIM = 32)
  const bool active = in_range && lane < P * K;
  const int qi = lane % P, part = lane / P;
  const int q = X->prim_order[qi];
  int myk = 0;
#pragma unroll 1
  for (int k = 1; k < S::kSeg; ++k) myk = qi >= X->seg_prim_begin[k] ? k : myk;
  double R[9], p[3] = {st[0], st[1], st[2]};
#pragma unroll
  for (int c = 0; c < 9; ++c) R[c] = Rw[c];
  static_for<1, S::kSeg>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if (k <= myk) {
      double pn[3];
      xform(R, p, M->seg_t[k][0], M->seg_t[k][1], M->seg_t[k][2], pn);
      p[0] = pn[0]; p[1] = pn[1]; p[2] = pn[2];
      rot_cols<S::axis(k)>(R, cs[2 * S::joint(k)], cs[2 * S::joint(k) + 1]);
    }
  });
  double pc[3], PR[9];
  prim_pose<S::kGimbal, true>(M, q, R, p, pc, PR);
  const int type = M->prim_type[q];
  const double rb = X->prim_bound[q];
  const double d0 = M->prim_dim[q][0], d1 = M->prim_dim[q][1], d2 = M->prim_dim[q][2];
  const double a[3] = {d0, d1, d2};

  // this lane's best so far; `cut` also knows the primitive's best after a merge (pruning only)
  double best = INFINITY, cut = INFINITY, br[3] = {NAN, NAN, NAN}, be[3] = {NAN, NAN, NAN};
  int bi = -1;
  auto take = [&](double d, int ob, const double pr[3], const double pq[3]) {  // points in the frame of box ob
    if (d < best || (d == best && ob < bi)) {
      const DevObs& O = obs[ob];
      best = d;
      bi = ob;
      cut = fmin(cut, d);
#pragma unroll
      for (int c = 0; c < 3; ++c) {  // box frame -> world (NaN points stay NaN)
        br[c] = fma(O.R[3 * c], pr[0], fma(O.R[3 * c + 1], pr[1], fma(O.R[3 * c + 2], pr[2], O.c[c])));
        be[c] = fma(O.R[3 * c], pq[0], fma(O.R[3 * c + 1], pq[1], fma(O.R[3 * c + 2], pq[2], O.c[c])));
      }
    }
  };
  const double nan3[3] = {NAN, NAN, NAN};
  // one box for this lane (kGridNone: none), by the whole wave: the exact stage is shared
  auto process = [&](int ob) {
    bool need = false, reach = false;
    double Rr[9], t[3] = {0.0, 0.0, 0.0}, h[3] = {0.0, 0.0, 0.0};
    if (ob != kGridNone) {
      to_box_frame(obs[ob], pc, t);
      const double l2 = point_box_dist2(t, obs[ob].h);
      const double thr = rb + fmax(cut, 0.0) + kSatSlack;  // centre distance - bounding radius bounds the pair from below
      reach = l2 <= thr * thr * (1.0 + 1e-12);
    }
    if (!__any(reach)) return;  // (most rounds of a sweep over all boxes)
    if (ob != kGridNone) {
      const DevObs& O = obs[ob];
      if (reach) {
        if (type == MVK_PRIM_BOX) {
          box_rel(O, PR, Rr);
          h[0] = O.h[0]; h[1] = O.h[1]; h[2] = O.h[2];
          const double g = box_box_sat<true>(t, Rr, a, h, cut + kSatSlack);
          if (g <= 0.0) take(g, ob, nan3, nan3);  // in contact: no points
          else need = g <= cut + kSatSlack;
        } else {
          double pr[3], pq[3];
          const double d = narrow_cp(type, PR, d0, d1, d2, O, t, pr, pq, cut);
          take(d, ob, pr, pq);
        }
      }
    }
    double pr[3], pq[3];
    const double d = exact_cp_wave_ab(ex_item, ex_res, ex_s2, ex_win, need, Rr, t, h, a, pr, pq);
    if (need) take(d, ob, pr, pq);
  };
  auto scan = [&](const uint16_t* __restrict__ ids, int n_items) {  // ids == nullptr: boxes 0 .. n_items-1
#pragma unroll 1
    for (int e = part; __any(e < n_items); e += K) process(e < n_items ? (ids ? static_cast<int>(ids[e]) : e) : kGridNone);
  };
  // merge of the K partial results of a primitive: every lane learns its primitive's best distance (`cut`); lane
  // (qi, 0) takes the winner's record
  auto merge = [&](bool final_round) {
    red[lane * kRed + 0] = best;
    red[lane * kRed + 1] = static_cast<double>(bi);
#pragma unroll
    for (int c = 0; c < 3; ++c) { red[lane * kRed + 2 + c] = br[c]; red[lane * kRed + 5 + c] = be[c]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double mb = INFINITY;
    int mi = -1, ml = lane;
#pragma unroll 1
    for (int j = 0; j < K; ++j) {
      const int l = qi + j * P;
      const double d = red[l * kRed + 0];
      const int b = static_cast<int>(red[l * kRed + 1]);
      if (b >= 0 && (mi < 0 || d < mb || (d == mb && b < mi))) { mb = d; mi = b; ml = l; }
    }
    cut = fmin(cut, mb);
    if (final_round && part == 0 && mi >= 0) {
      best = mb;
      bi = mi;
#pragma unroll
      for (int c = 0; c < 3; ++c) { br[c] = red[ml * kRed + 2 + c]; be[c] = red[ml * kRed + 5 + c]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  bool todo = active;  // primitives whose candidates are not complete yet
  if (A.dgrid.start != nullptr) {
    unsigned gb, ge;
    grid_cell(A.dgrid, pc, active, gb, ge);
    scan(A.dgrid.items + static_cast<size_t>(gb) * kGridGroup, static_cast<int>(ge - gb) * kGridGroup);
    merge(false);
    todo = active && !(rb + fmax(cut, 0.0) + 2.0 * kSatSlack <= A.dgrid_radius);
  }
  // The sweep over a complete set of boxes (`ids`: a far-grid cell's list; nullptr: all boxes in order) for the lanes
  // `who`.  First the box nearest to the primitive's centre: its distance bounds the primitive's from above.  Then the
  // boxes whose centre distance can undercut that are collected per lane (a cheap pass: one in a few dozen survives)
  // and only those go through the narrowphase, which the wave runs in lockstep.
  // (two stages through one body: the far grid's cell for the lanes inside its volume, then all boxes for the rest)
#pragma unroll 1
  for (int stage = 0; stage < 2; ++stage) {
    bool who = todo;
    const uint16_t* __restrict__ ids = nullptr;
    int n_items = A.n_obs;
    if (stage == 0) {
      if (A.fgrid.start == nullptr) continue;
      unsigned gb, ge;
      who = grid_cell_in(A.fgrid, pc, todo, gb, ge);
      ids = A.fgrid.items + static_cast<size_t>(gb) * kGridGroup;
      n_items = static_cast<int>(ge - gb) * kGridGroup;
    }
    if (!__any(who)) continue;
    auto box_at = [&](int e) -> int { return ids ? static_cast<int>(ids[e]) : e; };
    double l2min = INFINITY;
    int bmin = -1;
#pragma unroll 1
    for (int e = part; __any(who && e < n_items); e += K)
      if (who && e < n_items) {
        const int ob = box_at(e);
        if (ob != kGridNone) {
          double t[3];
          to_box_frame(obs[ob], pc, t);
          const double l2 = point_box_dist2(t, obs[ob].h);
          if (l2 < l2min) { l2min = l2; bmin = ob; }
        }
      }
    red[lane * kRed] = l2min;
    red[lane * kRed + 1] = static_cast<double>(bmin);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int j = 0; j < K; ++j) {
      const double l2 = red[(qi + j * P) * kRed];
      if (l2 < l2min) { l2min = l2; bmin = static_cast<int>(red[(qi + j * P) * kRed + 1]); }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (who) cut = fmin(cut, sqrt(l2min) * (1.0 + 1e-12));
    process(who && part == 0 && bmin >= 0 ? bmin : kGridNone);  // (bmin < 0: an empty scene)
    merge(false);
    int e = part;
#pragma unroll 1
    while (__any(who && e < n_items)) {
      int cnt = 0;
      const double thr = rb + fmax(cut, 0.0) + kSatSlack;
      const double thr2 = thr * thr * (1.0 + 1e-12);
#pragma unroll 1
      while (__any(who && e < n_items && cnt < kSweepKeep)) {
        if (who && e < n_items && cnt < kSweepKeep) {
          const int ob = box_at(e);
          if (ob != kGridNone) {
            double t[3];
            to_box_frame(obs[ob], pc, t);
            if (point_box_dist2(t, obs[ob].h) <= thr2) {
              keep[cnt * 64 + lane] = static_cast<unsigned short>(ob);
              ++cnt;
            }
          }
          e += K;
        }
      }
#pragma unroll 1
      for (int i = 0; __any(i < cnt); ++i) process(i < cnt ? static_cast<int>(keep[i * 64 + lane]) : kGridNone);
      merge(false);
    }
    todo = todo && !who;
  }
  merge(true);

  const bool report = active && part == 0;
  const bool usable = bi >= 0 && best > 0.0;
  const bool okq = bi < 0 || best > 0.0;
  const bool ok = !__any(report && !okq);
  if (report) {
    // normal from the environment to the robot, and the row over the chain up to this segment (as avoid_kernel)
    double nrm[3] = {br[0] - be[0], br[1] - be[1], br[2] - be[2]};
    const double inv = 1.0 / sqrt(fma(nrm[0], nrm[0], fma(nrm[1], nrm[1], nrm[2] * nrm[2])));
#pragma unroll
    for (int c = 0; c < 3; ++c) nrm[c] = usable ? nrm[c] * inv : 0.0;
    double row[W];
#pragma unroll
    for (int c = 0; c < W; ++c) row[c] = 0.0;
    if (A.av_full_body) {
      const double rx = br[0] - st[0], ry = br[1] - st[1], rz = br[2] - st[2];
      row[0] = nrm[0]; row[1] = nrm[1]; row[2] = nrm[2];
      row[3] = usable ? fma(ry, nrm[2], -rz * nrm[1]) : 0.0;
      row[4] = usable ? fma(rz, nrm[0], -rx * nrm[2]) : 0.0;
      row[5] = usable ? fma(rx, nrm[1], -ry * nrm[0]) : 0.0;
    }
    {
      double Rj[9], pj[3];
#pragma unroll
      for (int c = 0; c < 9; ++c) Rj[c] = Rw[c];
      pj[0] = st[0]; pj[1] = st[1]; pj[2] = st[2];
      static_for<1, S::kSeg>([&](auto sc_) {
        constexpr int s = decltype(sc_)::value;
        if (s <= myk) {
          double pn[3];
          xform(Rj, pj, M->seg_t[s][0], M->seg_t[s][1], M->seg_t[s][2], pn);
          pj[0] = pn[0]; pj[1] = pn[1]; pj[2] = pn[2];
          constexpr int ax = S::axis(s);
          const double ax_w[3] = {Rj[ax], Rj[3 + ax], Rj[6 + ax]};
          const double rx = br[0] - pj[0], ry = br[1] - pj[1], rz = br[2] - pj[2];
          const double vx = fma(ax_w[1], rz, -ax_w[2] * ry), vy = fma(ax_w[2], rx, -ax_w[0] * rz),
                       vz = fma(ax_w[0], ry, -ax_w[1] * rx);
          row[6 + S::joint(s)] = usable ? fma(nrm[0], vx, fma(nrm[1], vy, nrm[2] * vz)) : 0.0;
          rot_cols<ax>(Rj, cs[2 * S::joint(s)], cs[2 * S::joint(s) + 1]);
        }
      });
    }
    if (A.av_dist) A.av_dist[q * A.ld + gi] = best;
    if (A.av_box) A.av_box[q * A.ld + gi] = bi;
    if (A.av_lb)
      A.av_lb[q * A.ld + gi] = (usable && best < A.av_range) ? -A.av_gain * ((best - A.av_forbid) / (A.av_range - A.av_forbid)) : -1e6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (A.av_p_env) A.av_p_env[(3 * q + c) * A.ld + gi] = usable ? be[c] : NAN;
      if (A.av_p_robot) A.av_p_robot[(3 * q + c) * A.ld + gi] = usable ? br[c] : NAN;
    }
    if (A.av_row) {
#pragma unroll
      for (int c = 0; c < W; ++c) A.av_row[(static_cast<uint64_t>(W) * q + c) * A.ld + gi] = row[c];
    }
  }
  if (in_range && lane == 0 && A.av_ok) A.av_ok[gi] = ok ? 1 : 0;
  if (A.host_done) {  // as distance_small_kernel: the last block raises the pinned done word
    __syncthreads();
    if (tid == 0) {
      __threadfence_system();
      const unsigned before = atomicAdd(A.fused_done, 1u);
      if (before + 1u == A.small_target) {
        __threadfence_system();
        *A.host_done = A.fused_seq;
      }
    }
  }
}

// ------------------------------------------------------------------------
// small helper kernels
// ------------------------------------------------------------------------
// the no-allocation entry point's counter set must start at zero; a kernel, not hipMemsetAsync: captured into a
// graph the memset node did not take effect on replay (tests/test_zz_no_alloc.py: a few verdicts differed from the
// second launch of the graph on), a kernel node does
__global__ void zero_words_kernel(uint32_t* p, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0u;
}
__global__ void edge_init_kernel(int32_t* first_invalid, const int32_t* nd, int32_t nd_uniform, uint64_t n) {
  const uint64_t e = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e < n) {
    const int v = nd ? nd[e] : nd_uniform;
    first_invalid[e] = (v < 1 ? 1 : v) + 1;
  }
}
__global__ void edge_finish_kernel(const int32_t* first_invalid, const int32_t* nd, int32_t nd_uniform,
                                   uint8_t* ok, uint64_t n) {
  const uint64_t e = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e < n) {
    const int v = nd ? nd[e] : nd_uniform;
    ok[e] = first_invalid[e] == (v < 1 ? 1 : v) + 1 ? 1 : 0;
  }
}
__global__ void stats_final_kernel(const StatsPartial* part, uint64_t n, StatsPartial* out) {
  // one block of 256 threads strides over the partials, then a serial merge in LDS
  __shared__ StatsPartial sm[256];
  StatsPartial acc;
  acc.valid_count = 0; acc.band_count = 0;
  acc.min_dist = INFINITY; acc.min_dist_index = -1;
  acc.min_margin = INFINITY; acc.min_margin_index = -1;
  acc.max_thrust = -INFINITY; acc.max_thrust_index = -1;
  acc.min_thrust = INFINITY; acc.min_thrust_index = -1;
  acc.min_rp = INFINITY; acc.min_rp_index = -1;
  auto merge = [](StatsPartial& a, const StatsPartial& q) {
    a.valid_count += q.valid_count;
    a.band_count += q.band_count;
    auto lt = [](double v, long long i, double w, long long j) { return v < w || (v == w && (j < 0 || (i >= 0 && i < j))); };
    if (lt(q.min_dist, q.min_dist_index, a.min_dist, a.min_dist_index)) { a.min_dist = q.min_dist; a.min_dist_index = q.min_dist_index; }
    if (lt(q.min_margin, q.min_margin_index, a.min_margin, a.min_margin_index)) { a.min_margin = q.min_margin; a.min_margin_index = q.min_margin_index; }
    if (lt(-q.max_thrust, q.max_thrust_index, -a.max_thrust, a.max_thrust_index)) { a.max_thrust = q.max_thrust; a.max_thrust_index = q.max_thrust_index; }
    if (lt(q.min_thrust, q.min_thrust_index, a.min_thrust, a.min_thrust_index)) { a.min_thrust = q.min_thrust; a.min_thrust_index = q.min_thrust_index; }
    if (lt(q.min_rp, q.min_rp_index, a.min_rp, a.min_rp_index)) { a.min_rp = q.min_rp; a.min_rp_index = q.min_rp_index; }
  };
  for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) merge(acc, part[i]);
  sm[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    StatsPartial tot = sm[0];
    for (int t = 1; t < 256; ++t) merge(tot, sm[t]);
    *out = tot;
  }
}

// ev (nullable): three events recorded on the launch stream around the kernels
// (before the first, between first and second, after the last) for mvk_last_kernel_ms
template <int ROBOT, int NL>
int launch_shape(const EvalArgs& a, bool edge_mode, hipStream_t s, hipEvent_t* ev) {
  const uint64_t blocks = (a.n + kBlock - 1) / kBlock;
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7fffffffULL || a.n > 0xffffffffULL) return hipErrorInvalidValue;  // 32-bit queue entries
  dim3 g(static_cast<unsigned>(blocks)), b(kBlock);
  if (ev) (void)hipEventRecord(ev[0], s);
  if (a.full && !edge_mode && a.small_full == 2) {  // a few states: walk 1 and the distance stage in one launch, two states per block
    const dim3 gs(static_cast<unsigned>((a.n + 1) / 2));
    hipLaunchKernelGGL((full_small_kernel<ROBOT, NL>), gs, b, 0, s, a, a.model, a.extras, a.obs);
    if (ev) { (void)hipEventRecord(ev[1], s); (void)hipEventRecord(ev[2], s); }
    return hipGetLastError();
  }
  if (a.full && !edge_mode && a.f32_variant) {
    hipLaunchKernelGGL((model32_kernel<ROBOT, NL>), g, b, 0, s, a, a.model, a.extras);
    if (ev) (void)hipEventRecord(ev[1], s);
    if (a.full_rec) hipLaunchKernelGGL((distance_kernel<ROBOT, NL, true>), g, b, 0, s, a, a.model, a.extras, a.obs);
    if (ev) (void)hipEventRecord(ev[2], s);
    return hipGetLastError();
  }
  if (a.full && !edge_mode) {
    hipLaunchKernelGGL((model_kernel<ROBOT, NL>), g, b, 0, s, a, a.model, a.extras);
    if (ev) (void)hipEventRecord(ev[1], s);
    if (a.full_rec && a.small_full) {
      const dim3 gs(static_cast<unsigned>((a.n + (kBlock / kSmallGroup) - 1) / (kBlock / kSmallGroup)));
      hipLaunchKernelGGL((distance_small_kernel<ROBOT, NL>), gs, b, 0, s, a, a.model, a.extras, a.obs);
    } else if (a.full_rec) {
      hipLaunchKernelGGL((distance_kernel<ROBOT, NL>), g, b, 0, s, a, a.model, a.extras, a.obs);
    }
    if (ev) (void)hipEventRecord(ev[2], s);
    return hipGetLastError();
  }
  if (a.fused) {
    if (blocks > static_cast<uint64_t>(kFusedHardMax)) return hipErrorInvalidValue;
    if (edge_mode && a.fused == 2) {
      const dim3 g2(static_cast<unsigned>((a.n + kBlock / 2 - 1) / (kBlock / 2)));
      hipLaunchKernelGGL((fused_split_kernel<ROBOT, NL>), g2, b, 0, s, a, a.model, a.extras, a.obs, a.obsf);
    } else if (edge_mode) hipLaunchKernelGGL((fused_kernel<ROBOT, NL, true>), g, b, 0, s, a, a.model, a.extras, a.obs, a.obsf);
    else hipLaunchKernelGGL((fused_kernel<ROBOT, NL, false>), g, b, 0, s, a, a.model, a.extras, a.obs, a.obsf);
    if (ev) { (void)hipEventRecord(ev[1], s); (void)hipEventRecord(ev[2], s); }
    return hipGetLastError();
  }
  // exact stage: as many blocks as the chip holds at once stride over the chunks of all queues
  const uint64_t chunks_max = blocks * Shape<ROBOT, NL>::kSeg, resident = a.exact_blocks ? a.exact_blocks : 768;
  dim3 g2(static_cast<unsigned>(chunks_max < resident ? chunks_max : resident));
  dim3 g3(a.refine_blocks ? a.refine_blocks : 1u);
  if (edge_mode) {
    if (a.f32_walk) {
      hipLaunchKernelGGL((screen32_kernel<ROBOT, NL, true>), g, b, 0, s, a, a.model, a.extras, a.obsf);
      hipLaunchKernelGGL((screen_kernel<ROBOT, NL, true, true>), g3, b, 0, s, a, a.model, a.extras, a.obsf);
    } else {
      hipLaunchKernelGGL((screen_kernel<ROBOT, NL, true, false>), g, b, 0, s, a, a.model, a.extras, a.obsf);
    }
    if (ev) (void)hipEventRecord(ev[1], s);
    if (a.do_coll && a.n_obs > 0) hipLaunchKernelGGL((exact_kernel<ROBOT, NL, true>), g2, b, 0, s, a, a.model, a.extras, a.obs, a.obsf);
  } else {
    if (a.f32_walk) {
      hipLaunchKernelGGL((screen32_kernel<ROBOT, NL, false>), g, b, 0, s, a, a.model, a.extras, a.obsf);
      hipLaunchKernelGGL((screen_kernel<ROBOT, NL, false, true>), g3, b, 0, s, a, a.model, a.extras, a.obsf);
    } else {
      hipLaunchKernelGGL((screen_kernel<ROBOT, NL, false, false>), g, b, 0, s, a, a.model, a.extras, a.obsf);
    }
    if (ev) (void)hipEventRecord(ev[1], s);
    if (a.do_coll && a.n_obs > 0) hipLaunchKernelGGL((exact_kernel<ROBOT, NL, false>), g2, b, 0, s, a, a.model, a.extras, a.obs, a.obsf);
  }
  if (ev) (void)hipEventRecord(ev[2], s);
  return hipGetLastError();
}

}  // namespace

#if defined(MVK_PROF) || defined(MVK_PROFT)
extern "C" int mvk_debug_prof(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(g_prof)) != hipSuccess) return -1;
#ifdef MVK_PROFT
  {
    static unsigned long long rec[kProfWaves * 8];
    if (hipMemcpyFromSymbol(rec, HIP_SYMBOL(g_rec), sizeof(rec)) != hipSuccess) return -1;
    if (out) {
      unsigned long long first = ~0ull, last = 0;
      for (int w = 0; w < kProfWaves; ++w) {
        const unsigned long long* r = rec + 8 * w;
        if (!r[0]) continue;
        out[15] += 1;
        first = r[1] < first ? r[1] : first;
        last = r[2] > last ? r[2] : last;
        for (int i = 0; i < 4; ++i) { out[16 + 2 * i] += r[3 + i]; if (r[3 + i] > out[17 + 2 * i]) out[17 + 2 * i] = r[3 + i]; }
      }
      out[40] = first; out[41] = last;
    }
    if (reset) { std::memset(rec, 0, sizeof(rec)); if (hipMemcpyToSymbol(HIP_SYMBOL(g_rec), rec, sizeof(rec)) != hipSuccess) return -1; }
  }
#endif
  if (reset) { unsigned long long z[48] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

bool kernel_supported(int robot_type, int n_links) {
  if (robot_type == MVK_ROBOT_HYDRUS) return n_links == 4 || n_links == 6 || n_links == 8;
  if (robot_type == MVK_ROBOT_DRAGON) return n_links == 4;
  return false;
}

int launch_eval(const EvalArgs& a, int robot_type, int n_links, bool edge_mode, void* stream, void* events) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipEvent_t* ev = static_cast<hipEvent_t*>(events);
  if (robot_type == MVK_ROBOT_HYDRUS) {
    switch (n_links) {
      case 4: return launch_shape<MVK_ROBOT_HYDRUS, 4>(a, edge_mode, s, ev);
      case 6: return launch_shape<MVK_ROBOT_HYDRUS, 6>(a, edge_mode, s, ev);
      case 8: return launch_shape<MVK_ROBOT_HYDRUS, 8>(a, edge_mode, s, ev);
      default: break;
    }
  } else if (robot_type == MVK_ROBOT_DRAGON) {
    if (n_links == 4) return launch_shape<MVK_ROBOT_DRAGON, 4>(a, edge_mode, s, ev);
  }
  return hipErrorInvalidValue;
}

template <bool EDGE>
int launch_flag_small_t(const EvalArgs& a, int robot_type, int n_links, hipStream_t s) {
  const dim3 g(static_cast<unsigned>(a.n)), b(128);
  if (robot_type == MVK_ROBOT_HYDRUS && n_links == 4) hipLaunchKernelGGL((flag_small_kernel<MVK_ROBOT_HYDRUS, 4, EDGE>), g, b, 0, s, a, a.model, a.extras, a.obs);
  else if (robot_type == MVK_ROBOT_HYDRUS && n_links == 6) hipLaunchKernelGGL((flag_small_kernel<MVK_ROBOT_HYDRUS, 6, EDGE>), g, b, 0, s, a, a.model, a.extras, a.obs);
  else if (robot_type == MVK_ROBOT_HYDRUS && n_links == 8) hipLaunchKernelGGL((flag_small_kernel<MVK_ROBOT_HYDRUS, 8, EDGE>), g, b, 0, s, a, a.model, a.extras, a.obs);
  else if (robot_type == MVK_ROBOT_DRAGON && n_links == 4) hipLaunchKernelGGL((flag_small_kernel<MVK_ROBOT_DRAGON, 4, EDGE>), g, b, 0, s, a, a.model, a.extras, a.obs);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
int launch_flag_small(const EvalArgs& a, int robot_type, int n_links, bool edge_mode, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a.n == 0) return hipSuccess;
  if (a.n > 1024 || !a.host_done || (edge_mode ? (!a.host_first || !a.first_tag) : !a.host_flags)) return hipErrorInvalidValue;
  return edge_mode ? launch_flag_small_t<true>(a, robot_type, n_links, s) : launch_flag_small_t<false>(a, robot_type, n_links, s);
}
int launch_avoid(const EvalArgs& a, int robot_type, int n_links, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a.small_full) {  // a wave per state
    const dim3 gs(static_cast<unsigned>((a.n + (kBlock / 64) - 1) / (kBlock / 64))), bs(kBlock);
    if (robot_type == MVK_ROBOT_HYDRUS && n_links == 4) hipLaunchKernelGGL((avoid_small_kernel<MVK_ROBOT_HYDRUS, 4>), gs, bs, 0, s, a, a.model, a.extras, a.obs);
    else if (robot_type == MVK_ROBOT_HYDRUS && n_links == 6) hipLaunchKernelGGL((avoid_small_kernel<MVK_ROBOT_HYDRUS, 6>), gs, bs, 0, s, a, a.model, a.extras, a.obs);
    else if (robot_type == MVK_ROBOT_HYDRUS && n_links == 8) hipLaunchKernelGGL((avoid_small_kernel<MVK_ROBOT_HYDRUS, 8>), gs, bs, 0, s, a, a.model, a.extras, a.obs);
    else if (robot_type == MVK_ROBOT_DRAGON && n_links == 4) hipLaunchKernelGGL((avoid_small_kernel<MVK_ROBOT_DRAGON, 4>), gs, bs, 0, s, a, a.model, a.extras, a.obs);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  const uint64_t blocks = (a.n + kBlock - 1) / kBlock;
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
  const dim3 g(static_cast<unsigned>(blocks)), b(kBlock);
  if (robot_type == MVK_ROBOT_HYDRUS && n_links == 4) hipLaunchKernelGGL((avoid_kernel<MVK_ROBOT_HYDRUS, 4>), g, b, 0, s, a, a.model, a.extras, a.obs);
  else if (robot_type == MVK_ROBOT_HYDRUS && n_links == 6) hipLaunchKernelGGL((avoid_kernel<MVK_ROBOT_HYDRUS, 6>), g, b, 0, s, a, a.model, a.extras, a.obs);
  else if (robot_type == MVK_ROBOT_HYDRUS && n_links == 8) hipLaunchKernelGGL((avoid_kernel<MVK_ROBOT_HYDRUS, 8>), g, b, 0, s, a, a.model, a.extras, a.obs);
  else if (robot_type == MVK_ROBOT_DRAGON && n_links == 4) hipLaunchKernelGGL((avoid_kernel<MVK_ROBOT_DRAGON, 4>), g, b, 0, s, a, a.model, a.extras, a.obs);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
int launch_zero_words(uint32_t* p, uint64_t n_words, void* stream) {
  if (n_words == 0) return hipSuccess;
  const unsigned blocks = static_cast<unsigned>(std::min<uint64_t>((n_words + 255) / 256, 64));
  hipLaunchKernelGGL(zero_words_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, static_cast<uint32_t>(n_words));
  return hipGetLastError();
}
int launch_edge_init(int32_t* first_invalid, const int32_t* nd, int32_t nd_uniform, uint64_t n_edges, void* stream) {
  if (n_edges == 0) return hipSuccess;
  hipLaunchKernelGGL(edge_init_kernel, dim3(static_cast<unsigned>((n_edges + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), first_invalid, nd, nd_uniform, n_edges);
  return hipGetLastError();
}
int launch_edge_finish(const int32_t* first_invalid, const int32_t* nd, int32_t nd_uniform, uint8_t* ok,
                       uint64_t n_edges, void* stream) {
  if (n_edges == 0) return hipSuccess;
  hipLaunchKernelGGL(edge_finish_kernel, dim3(static_cast<unsigned>((n_edges + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), first_invalid, nd, nd_uniform, ok, n_edges);
  return hipGetLastError();
}
int launch_stats_final(const StatsPartial* partial, uint64_t n_partial, StatsPartial* out, void* stream) {
  hipLaunchKernelGGL(stats_final_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), partial,
                     n_partial, out);
  return hipGetLastError();
}

}  // namespace mvk
