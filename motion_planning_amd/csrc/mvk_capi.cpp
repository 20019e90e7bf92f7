// mvk_capi.cpp -- the C ABI of include/mvk.h: context ownership (device copies of
// the model blob and the obstacle set, staging buffers, a private stream) and the
// entry points that marshal a call into one kernel launch.
//
// There is deliberately no CPU evaluation path in this library: when no HIP
// device is usable every compute entry point returns MVK_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "mvk_internal.h"
#include "../../include/mvk_debug.h"

namespace mvk {

static thread_local char g_err[512] = "";
static thread_local int g_code = MVK_OK;  // the code of the latest fail() on this thread (for entry points that return a pointer)

int last_code() { return g_code; }

int fail(int code, const char* fmt, ...) {
  g_code = code;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

namespace {

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess)                                                             \
      return fail(MVK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                  __FILE__, __LINE__);                                                 \
  } while (0)

struct Buf {  // grow-only device buffer
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return MVK_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    HIP_TRY(hipMalloc(&p, want));
    cap = want;
    return MVK_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct PinBuf {  // grow-only pinned host buffer (async copies from pageable memory are staged and slow)
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return MVK_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = std::max<size_t>(bytes + bytes / 4 + 4096, size_t{1} << 18);  // small calls never re-pin
    HIP_TRY(hipHostMalloc(&p, want, hipHostMallocCoherent));  // fine-grained: kernels write answers and done words here that the host reads while they run
    cap = want;
    return MVK_OK;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

void quat_to_rows(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double s = 2.0 / (x * x + y * y + z * z + w * w);
  R[0] = 1.0 - s * (y * y + z * z); R[1] = s * (x * y - w * z);       R[2] = s * (x * z + w * y);
  R[3] = s * (x * y + w * z);       R[4] = 1.0 - s * (x * x + z * z); R[5] = s * (y * z - w * x);
  R[6] = s * (x * z - w * y);       R[7] = s * (y * z + w * x);       R[8] = 1.0 - s * (x * x + y * y);
}

}  // namespace
}  // namespace mvk

using namespace mvk;

namespace mvk {
void grid_build_threads(int threads);  // mvk_grid.cpp: host threads of the build_grid calls made from the calling thread
int edges_submit_two_launches(mvk_ctx* c);
}

struct mvk_ctx {
  int device = 0;
  mvk_model model{};
  mvk_params params{};
  DevExtras extras{};
  mvk_model* d_model = nullptr;
  DevExtras* d_extras = nullptr;
  Buf obs, obsf, grid_start, grid_items, dgrid_start[2], dgrid_items[2], dgrid_glb[2], fgrid_start, fgrid_items;
  // flag path hand-off between the screen and the exact stage (see EvalArgs).  Two alternating sets
  // of queue counters: call k appends to set k%2 and zeroes set (k+1)%2, so no memset sits between
  // launches.  Calls on one ctx must therefore be ordered with respect to each other (one stream at a time).
  static constexpr size_t kQueueCounterSet = static_cast<size_t>(MVK_MAX_SEG) * kQueueShards * kQueueCounterPitch;  // words
  Buf queue, queue_rec, queue_counts, refine_list;
  int flag_precision = MVK_PRECISION_F64;  // mvk_set_flag_precision
  float f32_delta_fk = 0.0f;
  float f32_bound_scale = 1.0f;
  const uint32_t* last_refine_counter = nullptr;  // device address of the last fp32-variant call's refine counter
  // small host batches in one launch (fused_kernel): a pinned control block holding the done word the kernel
  // raises, the 16-bit call sequence, the per-edge tag array, and what the last fused call still owes the caller
  PinBuf pin_ctl;
  Buf first_tag;
  uint32_t fused_seq = 0;
  int fused_max_blocks = kFusedMaxBlocks;  // blocks of kBlock lanes up to which a host call takes one fused launch
  bool fused_enabled = true;
  long long fused_giveups = 0;  // host calls repeated with two launches because the fused launch's barrier gave up (mvk_debug.h)
  uint32_t fused_spin_limit = 1u << 16;  // ~30 ms of polling before a block of the fused launch gives up on the grid barrier
  bool split_enabled = true;       // edge batches of the fused path: two lanes per step (MVK_NO_SPLIT=1: one)
  Buf small_done;                  // distance_small_kernel: blocks finished, over all polled calls (never reset)
  uint32_t small_total = 0;        // ... and the count the last polled call ends on
  bool tag_clear_needed = false;   // the call sequence number has wrapped since first_tag was last cleared
  uint64_t flag_small_max = kFlagSmallMax;    // flag path: states per host call that get a block each (flag_small_kernel)
  uint64_t small_avoid_max = kSmallAvoidMax;  // the same for the avoidance rows (avoid_small_kernel)
  uint64_t small_full_max = kSmallFullMax;  // states per call up to which the full path takes distance_small_kernel
  bool small_full_enabled = true;  // full path: distance_small_kernel for small batches (MVK_NO_SMALL_FULL=1: never)
  bool fused_in_flight = false;   // (edge checks begun with _begin)
  size_t edges_n = 0, edges_dld = 0, edges_blob_bytes = 0;
  uint64_t edges_total = 0;
  bool edges_have_nd = false;
  int32_t edges_nu = 1;
  uint64_t call_index = 0;
  int n_cu = 256;
  // mvk_check_motion_batch_begin / _end
  bool edges_pending = false;
  size_t edges_in_flight = 0, edges_nd_off = 0;
  // optional kernel timing (mvk_set_profiling): events around the kernels of the last call
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  bool profiling = false, timed = false;
  GridDev grid{};
  GridDev dgrid[2] = {};  // distance grids: [0] full path, [1] avoidance rows (start == nullptr: not in use for this scene)
  double dgrid_radius[2] = {0.0, 0.0};
  GridDev fgrid{};  // far grid: nearest-candidate lists without a radius (mvk_grid.cpp build_far_grid), behind the two above
  int n_obs = 0;
  float f32_delta = 0.0f;
  double f32_range = 0.0;
  hipStream_t stream = nullptr;
  void* static_scratch = nullptr;  // this GPU's copy of the no-allocation entry point's scratch array (resolved at mvk_create)
  // Calls on one ctx share its scratch (queues, records, counters, statistics partials, staging), so they must
  // execute in the order they were issued even when the caller hands different streams to the _dev entry points:
  // every call that touches the scratch records `order_ev` behind its work, and a call on another stream makes
  // that stream wait for it first (enter_stream / leave_stream).
  hipEvent_t order_ev = nullptr;
  hipStream_t order_stream = nullptr;
  bool order_armed = false;
  bool order_recorded = false;  // order_ev already sits behind the last call's work (caller-created streams: eagerly)
  // staging for the host-pointer entry points
  Buf in_a, in_b, flags, band, mind, margin, det[6], nd, ok, first, edge_off, av[7];
  // host edge checks (the planner's inner loop): one pinned upload blob, one pinned download
  Buf edge_blob;
  PinBuf pin_up, pin_down;
  // statistics of the last FULL batch
  Buf partial, stats_out, full_rec;
  uint64_t last_blocks = 0, last_n = 0;
  bool stats_valid = false;
};

namespace {

int check_model_shape(const mvk_model* m) {
  if (m->abi_version != MVK_ABI_VERSION) return fail(MVK_ERR_INVALID_ARG, "model abi_version %d != %d", m->abi_version, MVK_ABI_VERSION);
  if (!kernel_supported(m->robot_type, m->n_rotor))
    return fail(MVK_ERR_CAPACITY, "no kernel instantiated for robot_type %d with %d links", m->robot_type, m->n_rotor);
  const bool dragon = m->robot_type == MVK_ROBOT_DRAGON;
  const int NL = m->n_rotor;
  const int ns = dragon ? 2 * NL - 1 : NL, nj = dragon ? 2 * (NL - 1) : NL - 1;
  if (m->n_seg != ns || m->n_joint != nj || m->has_gimbal != (dragon ? 1 : 0) || m->wrench_dof != (dragon ? 3 : 4))
    return fail(MVK_ERR_INVALID_ARG, "model does not have the chain shape of its robot_type");
  for (int k = 1; k < ns; ++k) {
    const int axis = dragon ? ((k & 1) ? 1 : 2) : 2;
    if (m->seg_axis[k] != axis || m->seg_joint[k] != k - 1)
      return fail(MVK_ERR_INVALID_ARG, "segment %d: joint axis / index differ from the compiled chain shape", k);
  }
  for (int i = 0; i < NL; ++i)
    if (m->rotor_seg[i] != (dragon ? 2 * i : i)) return fail(MVK_ERR_INVALID_ARG, "rotor %d rides on an unexpected segment", i);
  if (m->n_prim < 0 || m->n_prim > MVK_MAX_PRIM) return fail(MVK_ERR_CAPACITY, "too many collision primitives");
  for (int q = 0; q < m->n_prim; ++q) {
    if (m->prim_type[q] < MVK_PRIM_SPHERE || m->prim_type[q] > MVK_PRIM_BOX) return fail(MVK_ERR_INVALID_ARG, "primitive %d: unknown kind", q);
    if (m->prim_seg[q] < 0 || m->prim_seg[q] >= ns) return fail(MVK_ERR_INVALID_ARG, "primitive %d: bad segment", q);
    if (m->prim_frame[q] != MVK_FRAME_SEG) {
      if (!dragon) return fail(MVK_ERR_INVALID_ARG, "primitive %d: gimbal frame on a robot without gimbals", q);
      const int i = m->prim_rotor[q];
      if (i < 0 || i >= NL || m->rotor_seg[i] != m->prim_seg[q]) return fail(MVK_ERR_INVALID_ARG, "primitive %d: gimbal / segment mismatch", q);
    }
  }
  for (int k = 0; k < ns; ++k) {
    int on_seg = 0;
    for (int q = 0; q < m->n_prim; ++q) on_seg += m->prim_seg[q] == k;
    if (on_seg > kSegPrimMax) return fail(MVK_ERR_CAPACITY, "segment %d carries %d collision primitives (max %d)", k, on_seg, kSegPrimMax);
  }
  return MVK_OK;
}

void derive_extras(const mvk_model* m, DevExtras* x) {
  std::memset(x, 0, sizeof(*x));
  x->tan_edf_max_tilt = std::tan(m->edf_max_tilt);
  {
    double mass = 0.0;
    for (int k = 0; k < m->n_seg; ++k) mass += m->seg_mass[k];
    x->inv_mass = 1.0 / mass;
  }
  // chain order of the primitives: by segment, and within a segment spheres and capsules before boxes.  Their exact
  // distances are closed forms that lower the running minimum at once, so the boxes of the segment (and everything
  // after) meet a tighter bound and fewer box pairs reach the 24-edge exact stage; results do not depend on the order.
  int at = 0;
  for (int k = 0; k < m->n_seg; ++k) {
    x->seg_prim_begin[k] = at;
    for (int pass = 0; pass < 2; ++pass)
      for (int q = 0; q < m->n_prim; ++q)
        if (m->prim_seg[q] == k && (m->prim_type[q] == MVK_PRIM_BOX) == (pass == 1)) x->prim_order[at++] = q;
  }
  for (int k = m->n_seg; k <= MVK_MAX_SEG; ++k) x->seg_prim_begin[k] = at;
  // bounding sphere of each segment's primitives (level-1 cull) and the robot's reach
  double chain = 0.0;
  x->reach = 0.0;
  std::vector<double> bound(m->n_prim);
  for (int q = 0; q < m->n_prim; ++q) {
    const double* d = m->prim_dim[q];
    bound[q] = m->prim_type[q] == MVK_PRIM_SPHERE ? d[0]
               : m->prim_type[q] == MVK_PRIM_CAPSULE ? d[0] + d[1]
                                                      : std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  }
  auto norm3 = [](const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
  for (int k = 0; k < m->n_seg; ++k) {
    if (k > 0) chain += norm3(m->seg_t[k]);
    double c[3] = {0, 0, 0};
    int cntk = 0;
    auto anchor = [&](int q) { return m->prim_frame[q] == MVK_FRAME_SEG ? m->prim_pos[q] : m->rotor_pos[m->prim_rotor[q]]; };
    for (int q = 0; q < m->n_prim; ++q)
      if (m->prim_seg[q] == k) {
        for (int a = 0; a < 3; ++a) c[a] += anchor(q)[a];
        ++cntk;
      }
    double r = 0.0;
    if (cntk) {
      for (int a = 0; a < 3; ++a) c[a] /= cntk;
      for (int q = 0; q < m->n_prim; ++q)
        if (m->prim_seg[q] == k) {
          const double* an = anchor(q);
          const double dd[3] = {an[0] - c[0], an[1] - c[1], an[2] - c[2]};
          // a primitive on a gimbal frame swings about the gimbal centre by |prim_pos|
          const double ext = bound[q] + (m->prim_frame[q] == MVK_FRAME_SEG ? 0.0 : norm3(m->prim_pos[q]));
          r = std::fmax(r, norm3(dd) + ext);
        }
      x->reach = std::fmax(x->reach, chain + norm3(c) + r);
    }
    SegTab& T = x->seg_tab[k];
    for (int a = 0; a < 3; ++a) { T.t[a] = m->seg_t[k][a]; T.sph[a] = c[a]; }
    T.axis = m->seg_axis[k];
    T.joint = m->seg_joint[k];
    T.np = 0;
    for (int oi = x->seg_prim_begin[k]; oi < x->seg_prim_begin[k + 1]; ++oi)  // (the same order as prim_order: slot j <-> prim_order[begin + j])
      if (const int q = x->prim_order[oi]; T.np < kSegPrimMax) {
        for (int a = 0; a < 3; ++a) {
          T.prim_pos[T.np][a] = m->prim_pos[q][a];
          T.gimbal_pos[T.np][a] = m->prim_frame[q] == MVK_FRAME_SEG ? 0.0 : m->rotor_pos[m->prim_rotor[q]][a];
        }
        T.prim_frame[T.np] = m->prim_frame[q];
        if (m->prim_frame[q] != MVK_FRAME_SEG) T.pad = 1;
        ++T.np;
      }
    x->seg_sph[k][0] = c[0]; x->seg_sph[k][1] = c[1]; x->seg_sph[k][2] = c[2];
    x->seg_sph[k][3] = r * (1.0 + 1e-12) + 1e-12;
  }
  // fp32 copy of the model for the fp32 variant of the flag path, and the error constants of its frame walk:
  // a rotation entry picks up <= 2.5e-7 per joint (fp32 sincos of an angle rounded to fp32) and <= 8 u per
  // segment of rounding (u = 2^-24), plus the conversion of the root rotation; a frame origin that error
  // times the lever (the robot's reach from the root)
  {
    ModelF32& f = x->f32;
    double mass = 0.0;
    for (int k = 0; k < m->n_seg; ++k) {
      mass += m->seg_mass[k];
      f.seg_mass[k] = static_cast<float>(m->seg_mass[k]);
      for (int a = 0; a < 3; ++a) {
        f.seg_t[k][a] = static_cast<float>(m->seg_t[k][a]);
        f.seg_mcom[k][a] = static_cast<float>(m->seg_mass[k] * m->seg_com[k][a]);
      }
    }
    for (int i = 0; i < m->n_rotor; ++i) {
      for (int a = 0; a < 3; ++a) f.rotor_pos[i][a] = static_cast<float>(m->rotor_pos[i][a]);
      f.rotor_ks[i] = static_cast<float>(m->m_f_rate * m->rotor_sigma[i]);
    }
    f.inv_mass = static_cast<float>(1.0 / mass);
    f.weight = static_cast<float>(mass * m->gravity);
    f.thrust_min = static_cast<float>(m->thrust_min);
    f.thrust_max = static_cast<float>(m->thrust_max);
    f.fc_t_thre = static_cast<float>(m->fc_t_min_thre);
    f.fc_rp_thre = static_cast<float>(m->fc_rp_min_thre);
    f.edf_offset_y = static_cast<float>(m->edf_offset_y);
    f.rotor_roll_z = static_cast<float>(m->rotor_roll_z);
    f.edf_two_r = static_cast<float>(2.0 * m->edf_radius);
    f.tan_edf_max_tilt = static_cast<float>(x->tan_edf_max_tilt);
    const double u = 5.9604645e-8;
    f.err_u = static_cast<float>(2.5e-7 * m->n_joint + 8.0 * u * (m->n_seg + 1));
    f.err_p = static_cast<float>((2.5e-7 * m->n_joint + 8.0 * u * (m->n_seg + 1)) * x->reach);
    for (int k = 0; k < m->n_seg; ++k) {
      for (int a = 0; a < 4; ++a) x->seg_sph_f[k][a] = static_cast<float>(x->seg_sph[k][a]);
      for (int j = 0; j < kSegPrimMax; ++j)
        for (int a = 0; a < 3; ++a) {
          x->seg_prim_pos_f[k][j][a] = static_cast<float>(x->seg_tab[k].prim_pos[j][a]);
          x->seg_gimbal_pos_f[k][j][a] = static_cast<float>(x->seg_tab[k].gimbal_pos[j][a]);
        }
    }
  }
  for (int q = 0; q < m->n_prim; ++q) {
    const double* d = m->prim_dim[q];
    switch (m->prim_type[q]) {
      case MVK_PRIM_SPHERE: x->prim_bound[q] = d[0]; x->prim_inner[q] = d[0]; break;
      case MVK_PRIM_CAPSULE: x->prim_bound[q] = d[0] + d[1]; x->prim_inner[q] = d[0]; break;
      default:
        x->prim_bound[q] = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        x->prim_inner[q] = std::fmin(d[0], std::fmin(d[1], d[2]));
    }
  }
  // Groups of co-located primitives (distance kernel: one broadphase per group).  Consecutive primitives of a segment
  // in drain order join the current group while their centre stays within kGroupTol of its first member's centre in
  // every configuration: |anchor - anchor_first| + the swing of a primitive on a gimbal frame about its anchor.
  {
    const bool off = std::getenv("MVK_NO_PRIM_GROUPS") && std::getenv("MVK_NO_PRIM_GROUPS")[0] == '1';
    const double kGroupTol = 0.08;
    auto anchor_of = [&](int q) { return m->prim_frame[q] == MVK_FRAME_SEG ? m->prim_pos[q] : m->rotor_pos[m->prim_rotor[q]]; };
    auto swing_of = [&](int q) {
      const double* v = m->prim_pos[q];
      return m->prim_frame[q] == MVK_FRAME_SEG ? 0.0 : std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    };
    auto apart = [&](int q, int l) {  // how far q's centre can be from l's, in any configuration
      if (q == l) return 0.0;
      const double *a = anchor_of(q), *b = anchor_of(l);
      return std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2])) + swing_of(q) + swing_of(l);
    };
    int ng = 0;
    x->grp_rb_max = 0.0;
    for (int k = 0; k < m->n_seg; ++k) {
      x->seg_grp_begin[k] = ng;
      int first = -1;
      for (int oi = x->seg_prim_begin[k]; oi < x->seg_prim_begin[k + 1]; ++oi) {
        const int q = x->prim_order[oi];
        if (first < 0 || off || apart(q, first) > kGroupTol) {  // a new group, q opens it
          x->grp_begin[ng++] = oi;
          first = q;
        }
      }
    }
    x->grp_begin[ng] = x->seg_prim_begin[m->n_seg];
    // the broadphase is centred on the member that needs the smallest radius to contain all of them (the largest
    // member, as a rule: its own list is then no longer than without the group)
    for (int g = 0; g < ng; ++g) {
      double best = INFINITY;
      for (int li = x->grp_begin[g]; li < x->grp_begin[g + 1]; ++li) {
        double r = 0.0;
        for (int oi = x->grp_begin[g]; oi < x->grp_begin[g + 1]; ++oi)
          r = std::fmax(r, x->prim_bound[x->prim_order[oi]] + apart(x->prim_order[oi], x->prim_order[li]));
        if (r < best) { best = r; x->grp_lead[g] = li; }
      }
      x->grp_rb[g] = x->grp_begin[g + 1] - x->grp_begin[g] == 1 ? best : best * (1.0 + 1e-12) + 1e-12;
      // avoidance rows: no bound is shared between primitives there.  With L the leader centre's distance to its nearest
      // box, a member at offset o from the leader is at most L + o from that box, and a box at centre distance l from
      // the leader is at least l - o - (member's radius) from the member: the member's nearest box has l <= L + 2 o + radius
      double rq = 0.0;
      for (int oi = x->grp_begin[g]; oi < x->grp_begin[g + 1]; ++oi)
        rq = std::fmax(rq, x->prim_bound[x->prim_order[oi]] + 2.0 * apart(x->prim_order[oi], x->prim_order[x->grp_lead[g]]));
      x->grp_rq[g] = x->grp_begin[g + 1] - x->grp_begin[g] == 1 ? rq : rq * (1.0 + 1e-12) + 1e-12;
    }
    for (int k = m->n_seg; k <= MVK_MAX_SEG; ++k) x->seg_grp_begin[k] = ng;
    for (int g = ng; g <= MVK_MAX_PRIM; ++g) x->grp_begin[g] = x->seg_prim_begin[m->n_seg];
    for (int g = 0; g < ng; ++g) x->grp_rb_max = std::fmax(x->grp_rb_max, x->grp_rb[g]);
    x->n_grp = off ? 0 : ng;
  }
}

int boxes_to_obs(const mvk_box* boxes, size_t n, std::vector<DevObs>& host) {
  if (n >= static_cast<size_t>(kGridNone)) return fail(MVK_ERR_CAPACITY, "at most %d obstacles (16-bit candidate ids)", kGridNone - 1);
  if (n && !boxes) return fail(MVK_ERR_INVALID_ARG, "null obstacle array");
  host.resize(n);
  for (size_t i = 0; i < n; ++i) {
    DevObs& o = host[i];
    for (int k = 0; k < 3; ++k) {
      o.c[k] = boxes[i].center[k];
      o.h[k] = 0.5 * boxes[i].size[k];
      if (!(boxes[i].size[k] >= 0.0)) return fail(MVK_ERR_INVALID_ARG, "obstacle %zu: negative or NaN size", i);
    }
    quat_to_rows(boxes[i].quat, o.R);
    for (int k = 0; k < 9; ++k)
      if (!std::isfinite(o.R[k])) return fail(MVK_ERR_INVALID_ARG, "obstacle %zu: invalid quaternion", i);
    o.bound = std::sqrt(o.h[0] * o.h[0] + o.h[1] * o.h[1] + o.h[2] * o.h[2]);
  }
  return MVK_OK;
}

// Host threads mvk_set_scene may use (the four grids are independent builds, the probe distances independent sums):
// at most four, fewer on small hosts; MVK_SCENE_THREADS=1 keeps everything on the calling thread.  Workers only touch
// their own outputs and never call HIP or fail() (whose text is per thread): every upload and every error is the caller's.
int scene_threads() {
  int want = 4;
  if (const char* ov = std::getenv("MVK_SCENE_THREADS")) if (std::atoi(ov) >= 1) want = std::atoi(ov);
  const unsigned hw = std::thread::hardware_concurrency();
  if (hw >= 1 && static_cast<unsigned>(want) > hw) want = static_cast<int>(hw);
  return std::max(1, std::min(want, 16));
}

// runs job(0) .. job(n_jobs - 1), on up to `threads` threads (the caller's included); a thread that cannot be started
// leaves its jobs to the caller
template <class F>
void run_jobs(int n_jobs, int threads, F&& job) {
  std::atomic<int> next{0};
  auto work = [&] { for (int k = next.fetch_add(1); k < n_jobs; k = next.fetch_add(1)) job(k); };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads && t < n_jobs; ++t) {
    try { pool.emplace_back(work); } catch (...) { break; }
  }
  work();
  for (auto& th : pool) th.join();
}

// Distance grids of the distance kernels (full path: minimum distance per state; avoidance rows: nearest box per
// primitive).  A primitive whose centre lies in a cell sees in that cell's list every box within
// radius = (largest primitive bounding radius + slack) of the centre, so whenever an upper bound <= slack of
// the distance sought is known, the list holds every box that can realise the minimum and the sweep over all
// boxes is not needed.  The slack is read off the scene: a percentile of the distance from a point of the
// obstacles' bounding volume to the nearest box.  Grid 0 takes the 25th percentile (short lists for the states close
// to something, which is most of them), grid 1 the 95th: the full path asks grid 0 first and falls back to grid 1 for
// the states it cannot answer, the avoidance rows (nearest box of every primitive on its own) use grid 1.  Measured on
// c2 / c2k / c3 / c5 (profiles/pct_sweep.sh).  Scenes where the lists would not be much shorter than the whole set
// keep the sweep (scalar loads at wave-uniform addresses).
int upload_distance_grids(mvk_ctx* c, const std::vector<DevObs>& host) {
  for (int g = 0; g < 2; ++g) {
    c->dgrid[g] = GridDev{};
    c->dgrid_radius[g] = 0.0;
  }
  c->fgrid = GridDev{};
  const size_t n = host.size();
  if (n < 64) return MVK_OK;
  // diagnostics (profiles/full_time.py, tests): keep the sweep over all boxes for scenes set from now on
  if (const char* off = std::getenv("MVK_NO_DISTANCE_GRID")) if (off[0] == '1') return MVK_OK;
  double rb = c->extras.grp_rb_max;  // (>= every primitive's bounding radius: the distance kernel asks about group leaders)
  for (int q = 0; q < c->model.n_prim; ++q) rb = std::fmax(rb, c->extras.prim_bound[q]);
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      lo[k] = std::fmin(lo[k], host[i].c[k] - host[i].bound);
      hi[k] = std::fmax(hi[k], host[i].c[k] + host[i].bound);
    }
  // 2048 probe points, fewer for very large scenes (the probes cost kProbe * n point-to-box distances on the host)
  const int kProbe = static_cast<int>(std::min<size_t>(2048, std::max<size_t>(256, (size_t{1} << 23) / n)));
  std::vector<double> probe(3 * kProbe), dist(kProbe);
  uint64_t rng = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < 3 * kProbe; ++i) {
    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
    const double u = static_cast<double>(rng >> 11) * (1.0 / 9007199254740992.0);
    probe[i] = lo[i % 3] + u * (hi[i % 3] - lo[i % 3]);
  }
  const int threads = scene_threads();
  const auto t_probe0 = std::chrono::steady_clock::now();
  constexpr int kProbeChunk = 64;
  run_jobs((kProbe + kProbeChunk - 1) / kProbeChunk, threads, [&](int job) {
  for (int i = job * kProbeChunk; i < std::min(kProbe, (job + 1) * kProbeChunk); ++i) {
    double best2 = INFINITY;
    for (size_t b = 0; b < n; ++b) {
      const DevObs& o = host[b];
      const double d[3] = {probe[3 * i] - o.c[0], probe[3 * i + 1] - o.c[1], probe[3 * i + 2] - o.c[2]};
      double s2 = 0.0;
      for (int a = 0; a < 3; ++a) {
        const double t = std::fabs(o.R[a] * d[0] + o.R[3 + a] * d[1] + o.R[6 + a] * d[2]) - o.h[a];
        if (t > 0.0) s2 += t * t;
      }
      best2 = std::fmin(best2, s2);
    }
    dist[i] = std::sqrt(best2);
  }
  });
  std::sort(dist.begin(), dist.end());
  const auto t_build0 = std::chrono::steady_clock::now();
  // the three builds side by side (jobs 0, 1: the distance grids; job 2: the far grid), then the uploads from here
  GridHost dgh[2], fgh;
  bool dbuilt[2] = {false, false}, fbuilt = false;
  double dmean[2] = {0.0, 0.0};
  const bool by_bound = !(std::getenv("MVK_NO_GRID_BOUNDS") && std::getenv("MVK_NO_GRID_BOUNDS")[0] == '1');
  bool far_wanted = true;
  if (const char* off = std::getenv("MVK_NO_FAR_GRID")) if (off[0] == '1') far_wanted = false;  // diagnostics, tests
  double diag = 0.0;
  for (int k = 0; k < 3; ++k) diag += (hi[k] - lo[k]) * (hi[k] - lo[k]);
  const double far_margin = std::fmin(std::fmax(c->extras.reach, 0.5), 4.0);
  const double far_pad = 8.0 * static_cast<double>(c->f32_delta) + 4e-6 * (std::sqrt(diag) + 2.0 * far_margin) + 1e-6;
  // reach: the largest radius term any walk asks with -- the avoidance rows' group walks use grp_rq (a member off the
  // leader's centre by o needs the boxes within radius + 2 o of the leader's nearest-box distance)
  double far_reach = rb;
  for (int gi = 0; gi < c->extras.n_grp; ++gi) far_reach = std::fmax(far_reach, c->extras.grp_rq[gi]);
  run_jobs(3, threads, [&](int job) {
    if (job == 2) {
      grid_build_threads(threads);
      // (built whenever a distance grid may exist; dropped below if neither does)
      if (far_wanted) fbuilt = build_far_grid(host.data(), n, far_reach, far_margin, far_pad, fgh) == MVK_OK;
      return;
    }
    const int g = job;
    grid_build_threads(threads);
    int pct = g == 0 ? 25 : 95;  // (profiles/pct_sweep.sh: with the second grid behind it the first one pays off small)
    if (const char* ov = std::getenv("MVK_DISTANCE_GRID_PCT")) if (std::atoi(ov) > 0 && std::atoi(ov) < 100) pct = std::atoi(ov);  // diagnostics
    if (g == 1) if (const char* ov = std::getenv("MVK_DISTANCE_GRID_PCT2")) if (std::atoi(ov) > 0 && std::atoi(ov) < 100) pct = std::atoi(ov);
    double slack = std::fmin(std::fmax(dist[(kProbe * pct) / 100], 0.05), 4.0);
    // lists sorted by the box-to-cell lower bound, so that a walk stops early (diagnostics: MVK_NO_GRID_BOUNDS=1).
    // The second grid's cells: with bounds, finer cells end the walks sooner (a box's bound is taken against the whole
    // cell); the first grid's cells are as fine as the cell budget allows already
    double cell_frac = (by_bound && g == 1) ? 0.25 : 0.5;
    if (const char* ov = std::getenv("MVK_GRID_CELL_FRAC")) if (g == 1 && std::atof(ov) >= 0.05 && std::atof(ov) <= 1.0) cell_frac = std::atof(ov);  // diagnostics
    double rbg = rb;  // (the avoidance rows' walks ask with grp_rq)
    if (g == 1) for (int gi = 0; gi < c->extras.n_grp; ++gi) rbg = std::fmax(rbg, c->extras.grp_rq[gi]);
    bool built = false;
    for (int attempt = 0; attempt < 3 && !built; ++attempt, slack *= 0.5) built = build_grid(host.data(), n, rbg + slack, dgh[g], by_bound, 4.0 * static_cast<double>(c->f32_delta), cell_frac) == MVK_OK;
    if (!built) return;  // too dense for the grid's budget: the sweep stays
    double mean_len = 0.0;
    for (int i = 0; i < kProbe; ++i) {
      const uint16_t* ids;
      mean_len += grid_query(dgh[g], &probe[3 * i], &ids);
    }
    dmean[g] = mean_len / kProbe;
    dbuilt[g] = true;
  });
  if (std::getenv("MVK_DISTANCE_GRID_TRACE"))
    std::fprintf(stderr, "mvk: %d host thread(s): probe distances %.1f ms, distance grids + far grid %.1f ms\n", threads,
                 std::chrono::duration<double, std::milli>(t_build0 - t_probe0).count(),
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build0).count());
  for (int g = 0; g < 2; ++g) {
    if (!dbuilt[g]) continue;
    GridHost& gh = dgh[g];
    const double mean_len = dmean[g];
    if (mean_len > 0.25 * static_cast<double>(n)) continue;
    if (int rc = c->dgrid_start[g].reserve(sizeof(uint32_t) * gh.start.size())) return rc;
    if (int rc = c->dgrid_items[g].reserve(sizeof(uint16_t) * (gh.items.size() + kGridGroup))) return rc;
    HIP_TRY(hipMemcpy(c->dgrid_start[g].p, gh.start.data(), sizeof(uint32_t) * gh.start.size(), hipMemcpyHostToDevice));
    if (!gh.items.empty())
      HIP_TRY(hipMemcpy(c->dgrid_items[g].p, gh.items.data(), sizeof(uint16_t) * gh.items.size(), hipMemcpyHostToDevice));
    GridDev& d = c->dgrid[g];
    for (int k = 0; k < 3; ++k) { d.origin[k] = gh.origin[k]; d.n[k] = gh.n[k]; }
    d.inv_h = 1.0 / gh.h;
    d.pad = 0;
    d.start = static_cast<const uint32_t*>(c->dgrid_start[g].p);
    d.items = static_cast<const uint16_t*>(c->dgrid_items[g].p);
    d.glb = nullptr;
    if (!gh.glb.empty()) {
      if (int rc = c->dgrid_glb[g].reserve(sizeof(float) * (gh.glb.size() + 1))) return rc;
      HIP_TRY(hipMemcpy(c->dgrid_glb[g].p, gh.glb.data(), sizeof(float) * gh.glb.size(), hipMemcpyHostToDevice));
      d.glb = static_cast<const float*>(c->dgrid_glb[g].p);
    }
    c->dgrid_radius[g] = gh.radius;
    if (std::getenv("MVK_DISTANCE_GRID_TRACE"))
      std::fprintf(stderr, "mvk: distance grid %d: radius %.3f (slack %.3f), cell %.3f, %d x %d x %d cells, mean list %.1f of %zu boxes\n",
                   g, gh.radius, gh.radius - rb, gh.h, gh.n[0], gh.n[1], gh.n[2], mean_len, n);
  }
  // Behind them, for the primitives neither can answer (farther from every box than the second grid's slack: open
  // space, the rim of the obstacles' volume): the far grid, whose lists are complete at any distance.  It replaces
  // the sweep over all boxes inside the boxes' bound inflated by the robot's reach.
  const bool far_on = far_wanted && (c->dgrid[0].start != nullptr || c->dgrid[1].start != nullptr);
  if (far_on) {
    const double margin = far_margin;
    GridHost& gh = fgh;
    if (fbuilt) {
      if (int rc = c->fgrid_start.reserve(sizeof(uint32_t) * gh.start.size())) return rc;
      if (int rc = c->fgrid_items.reserve(sizeof(uint16_t) * (gh.items.size() + kGridGroup))) return rc;
      HIP_TRY(hipMemcpy(c->fgrid_start.p, gh.start.data(), sizeof(uint32_t) * gh.start.size(), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(c->fgrid_items.p, gh.items.data(), sizeof(uint16_t) * gh.items.size(), hipMemcpyHostToDevice));
      GridDev& d = c->fgrid;
      for (int k = 0; k < 3; ++k) { d.origin[k] = gh.origin[k]; d.n[k] = gh.n[k]; }
      d.inv_h = 1.0 / gh.h;
      d.pad = 0;
      d.start = static_cast<const uint32_t*>(c->fgrid_start.p);
      d.items = static_cast<const uint16_t*>(c->fgrid_items.p);
      if (std::getenv("MVK_DISTANCE_GRID_TRACE"))
        std::fprintf(stderr, "mvk: far grid: margin %.2f, cell %.3f, %d x %d x %d cells, mean list %.1f of %zu boxes\n", margin, gh.h, gh.n[0],
                     gh.n[1], gh.n[2], static_cast<double>(gh.items.size()) / static_cast<double>(gh.start.size() - 1), n);
    }
  }
  return MVK_OK;
}

int upload_scene(mvk_ctx* c, const mvk_box* boxes, size_t n) {
  std::vector<DevObs> host;
  if (int rc = boxes_to_obs(boxes, n, host)) return rc;
  // fp32 copy for the level-1 cull, and the coordinate range / error bound it is trusted for:
  // every fp32 rounding in the centre-distance evaluation is <= 2^-24 relative to a coordinate
  // difference of at most 2*range; the accumulated bound on the distance error is < 4e-6 * range.
  std::vector<DevObsF> hostf(n + 1);
  double range = 64.0;
  for (size_t i = 0; i < n; ++i) {
    for (int k = 0; k < 3; ++k) {
      hostf[i].c[k] = static_cast<float>(host[i].c[k]);
      hostf[i].h[k] = static_cast<float>(host[i].h[k]);
      range = std::fmax(range, std::fabs(host[i].c[k]) + host[i].bound);
    }
    for (int k = 0; k < 9; ++k) hostf[i].R[k] = static_cast<float>(host[i].R[k]);
    hostf[i].pad = 0.0f;
  }
  if (!(range < 1e6)) return fail(MVK_ERR_INVALID_ARG, "obstacle coordinates beyond 1e6 m");
  // entry n: the box the grid's padding id resolves to (far away, never a candidate)
  std::memset(&hostf[n], 0, sizeof(DevObsF));
  hostf[n].c[0] = hostf[n].c[1] = hostf[n].c[2] = 3.0e18f;
  hostf[n].R[0] = hostf[n].R[4] = hostf[n].R[8] = 1.0f;
  // broadphase grid for the largest segment sphere (level-1 cull of the flag path)
  double rq = 0.0;
  for (int k = 0; k < c->model.n_seg; ++k) rq = std::fmax(rq, c->extras.seg_sph[k][3]);
  // fp32 variant of the flag path: its frame walk places a segment centre within fk_err of the true one inside the
  // fp32 coordinate range (<= 16 u of relative rounding per segment on coordinates up to `range`, plus the fp32
  // cos/sin), so the cell it looks up must list every box within rq of any point that close: the grid is built
  // for rq + 2 fk_err (a few tenths of a millimetre), and the fp32 thresholds are padded by f32_delta + fk_err
  const double fk_err = 16.0 * 5.9604645e-8 * (c->model.n_seg + 1) * range + static_cast<double>(c->extras.f32.err_p);
  rq += 2.0 * fk_err;
  GridHost gh;
  const auto t_flag0 = std::chrono::steady_clock::now();
  grid_build_threads(scene_threads());
  if (int rc = build_grid(host.data(), n, rq, gh)) return rc;
  if (std::getenv("MVK_DISTANCE_GRID_TRACE"))
    std::fprintf(stderr, "mvk: flag grid built in %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_flag0).count());
  if (int rc = c->obs.reserve(sizeof(DevObs) * (n ? n : 1))) return rc;
  if (int rc = c->obsf.reserve(sizeof(DevObsF) * (n + 1))) return rc;
  if (int rc = c->grid_start.reserve(sizeof(uint32_t) * gh.start.size())) return rc;
  if (int rc = c->grid_items.reserve(sizeof(uint16_t) * (gh.items.size() + kGridGroup))) return rc;
  if (n) HIP_TRY(hipMemcpy(c->obs.p, host.data(), sizeof(DevObs) * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->obsf.p, hostf.data(), sizeof(DevObsF) * (n + 1), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->grid_start.p, gh.start.data(), sizeof(uint32_t) * gh.start.size(), hipMemcpyHostToDevice));
  if (!gh.items.empty())
    HIP_TRY(hipMemcpy(c->grid_items.p, gh.items.data(), sizeof(uint16_t) * gh.items.size(), hipMemcpyHostToDevice));
  for (int k = 0; k < 3; ++k) { c->grid.origin[k] = gh.origin[k]; c->grid.n[k] = gh.n[k]; }
  c->grid.inv_h = 1.0 / gh.h;
  c->grid.pad = 0;
  c->grid.start = static_cast<const uint32_t*>(c->grid_start.p);
  c->grid.items = static_cast<const uint16_t*>(c->grid_items.p);
  c->n_obs = static_cast<int>(n);
  c->f32_range = range;
  c->f32_delta = static_cast<float>(4e-6 * range);
  c->f32_delta_fk = static_cast<float>(4e-6 * range + fk_err);
  return upload_distance_grids(c, host);
}

int check_params(const mvk_model* m, const mvk_params* p) {
  if (p->planning_mode < 0 || p->planning_mode > 2) return fail(MVK_ERR_INVALID_ARG, "planning_mode must be 0..2");
  if (p->motion_type < 0 || p->motion_type > 1) return fail(MVK_ERR_INVALID_ARG, "motion_type must be 0 (SE2) or 1 (SE3)");
  if (!(p->eps_dist >= 0.0) || !(p->eps_margin >= 0.0)) return fail(MVK_ERR_INVALID_ARG, "negative band width");
  (void)m;
  return MVK_OK;
}

int layout_rows(const mvk_ctx* c, int layout) {
  if (layout == MVK_LAYOUT_POSE7) return 7 + c->model.n_joint;
  if (layout == MVK_LAYOUT_OMPL) return space_dim(&c->params, c->model.n_joint);
  return fail(MVK_ERR_INVALID_ARG, "unknown layout %d", layout);
}

void base_args(const mvk_ctx* c, EvalArgs& a) {
  std::memset(&a, 0, sizeof(a));
  const mvk_params& p = c->params;
  a.planning_mode = p.planning_mode;
  a.motion_type = p.motion_type;
  // gating of motion_planning.cpp:138-151 (stability) and :153 (collision)
  a.do_stab = p.check_stability < 0 ? (p.planning_mode != MVK_MODE_ONLY_BASE || p.motion_type == MVK_SE3)
                                    : (p.check_stability != 0);
  a.do_coll = p.check_collision < 0 ? (p.planning_mode != MVK_MODE_ONLY_JOINTS) : (p.check_collision != 0);
  a.eps_dist = p.eps_dist;
  a.eps_margin = p.eps_margin;
  a.model = c->d_model;
  a.extras = c->d_extras;
  a.obs = static_cast<const DevObs*>(c->obs.p);
  a.obsf = static_cast<const DevObsF*>(c->obsf.p);
  a.grid = c->grid;
  a.dgrid = c->dgrid[0];
  a.dgrid_radius = c->dgrid_radius[0];
  a.dgrid2 = c->dgrid[1];
  a.dgrid2_radius = c->dgrid_radius[1];
  a.fgrid = c->fgrid;
  a.f32_delta = c->f32_delta;
  a.f32_range = c->f32_range;
  a.f32_delta_fk = c->f32_delta_fk;
  a.f32_bound_scale = c->f32_bound_scale;
  a.n_obs = c->n_obs;
  a.nd_uniform = 1;
}

// Ordering of the calls that use the ctx's scratch buffers (see mvk_ctx::order_ev): a call on a stream other than
// the last one's first makes its stream wait for the last call's work; every call leaves its own mark behind.
//
// The last call's stream may be the CALLER's (every _dev entry point takes one), and by the time the next call
// arrives it may have been destroyed -- and HIP does not survive being asked about a destroyed stream (a query, a
// record, even hipStreamIsCapturing on the stale handle is a use after free).  So a caller-created stream is only
// ever touched while the call that was handed it is still running:
//   * a call on a caller-created stream records the mark (order_ev) behind its work EAGERLY, before it returns;
//   * a call on the ctx's own stream or on the null stream (neither can be destroyed) records nothing: the mark is
//     recorded lazily, when a call on another stream needs it -- calls that stay on one of these streams, the usual
//     case and bench.py's, never pay for an event between their launches;
//   * later calls use the EVENT only (hipStreamWaitEvent / hipEventSynchronize), never the old stream handle.
// A call that is being captured into a graph ran nothing and leaves no mark; a call captured while the ctx's last
// call may still be running elsewhere is refused (HIP forbids the waits that would be needed, see below).
bool stream_capturing(hipStream_t s) {
  if (!s) return false;  // the legacy stream cannot be captured (and asking about it while another stream captures is itself an error)
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  return st != hipStreamCaptureStatusNone;
}
// host wait for whatever the ctx last enqueued, without using a caller's stream handle
int drain_order_stream(mvk_ctx* c) {
  if (!c->order_armed) return MVK_OK;
  const hipError_t e = c->order_recorded ? hipEventSynchronize(c->order_ev) : hipStreamSynchronize(c->order_stream);
  c->order_armed = false;
  if (e != hipSuccess) {  // (e.g. the event's stream died with work pending): one device-wide wait instead
    (void)hipGetLastError();
    HIP_TRY(hipDeviceSynchronize());
  }
  return MVK_OK;
}
int enter_stream(mvk_ctx* c, hipStream_t s) {
  if (!c->order_armed || s == c->order_stream) return MVK_OK;
  if (stream_capturing(s)) {
    // The new call is being captured: a wait on an event recorded outside the capture would either fail or drop
    // out of the graph, and HIP refuses every synchronising call from a capturing thread (and invalidates the
    // capture for the attempt).  So nothing of the runtime is asked here: either the ctx knows its last call has
    // finished (every blocking entry point disarms the mark once it has waited), or the caller is told.
    return fail(MVK_ERR_INVALID_ARG, "graph capture: the previous call on this ctx may still be running on another stream; "
                "synchronise that stream (or call a blocking entry point) before capturing");
  }
  hipError_t e = hipSuccess;
  if (!c->order_recorded) {  // the lazy mark: order_stream is the ctx's own or the null stream here, alive by construction
    e = hipEventRecord(c->order_ev, c->order_stream);
    c->order_recorded = e == hipSuccess;
  }
  if (e == hipSuccess) e = hipStreamWaitEvent(s, c->order_ev, 0);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return drain_order_stream(c);
  }
  return MVK_OK;
}
int leave_stream(mvk_ctx* c, hipStream_t s) {
  const bool callers = s != c->stream && s != nullptr;
  // a call captured into a graph ran nothing: there is no work of it to order later calls behind (the graph's
  // launches are the caller's to order against other users of the ctx, mvk.h)
  if (callers && stream_capturing(s)) return MVK_OK;
  c->order_stream = s;
  c->order_armed = true;
  c->order_recorded = false;
  if (callers) {  // while the stream is certainly alive
    if (hipEventRecord(c->order_ev, s) == hipSuccess) c->order_recorded = true;
    else {        // cannot mark it: wait for it here, once, rather than trust the handle later
      (void)hipGetLastError();
      (void)hipStreamSynchronize(s);
      c->order_armed = false;
    }
  }
  return MVK_OK;
}
// host wait for the last call on the ctx, whatever stream it went to
int wait_last_call(mvk_ctx* c) { return drain_order_stream(c); }
// a blocking entry point's wait for its own work: afterwards nothing of the ctx is in flight on `s`
int sync_stream(mvk_ctx* c, hipStream_t s) {
  HIP_TRY(hipStreamSynchronize(s));
  if (c->order_armed && c->order_stream == s) c->order_armed = false;
  return MVK_OK;
}

// ---- small host batches: one launch, no copies (fused_kernel) --------------------------------------------------
// eligible: flag path in fp64, at most kFusedMaxBlocks blocks of lanes (all resident at once)
bool fused_eligible(const mvk_ctx* c, uint64_t lanes) {
  return c->fused_enabled && c->flag_precision == MVK_PRECISION_F64 && lanes > 0 &&
         lanes <= static_cast<uint64_t>(c->fused_max_blocks) * kBlock;
}
// sequence number of a call that reports through the pinned done word (16 bits: first_tag packs it)
uint32_t next_seq(mvk_ctx* c) {
  if (c->fused_seq >= 0xFFFFu) { c->fused_seq = 0u; c->tag_clear_needed = true; }
  return ++c->fused_seq;
}
// after prepare_queue: the barrier / completion counters of this call's set, the next sequence number, the done word
int fused_arm(mvk_ctx* c, EvalArgs& a, hipStream_t s, size_t n_edges) {
  a.fused = 1;
  a.fused_bar = a.queue_count + static_cast<size_t>(MVK_MAX_SEG * kQueueShards - 2) * kQueueCounterPitch;
  a.fused_done = a.queue_count + static_cast<size_t>(MVK_MAX_SEG * kQueueShards - 3) * kQueueCounterPitch;
  a.fused_seq = next_seq(c);
  a.fused_spin_limit = c->fused_spin_limit;
  a.host_done = static_cast<volatile uint32_t*>(c->pin_ctl.p);
  if (n_edges) {
    const size_t before = c->first_tag.cap;
    if (int rc = c->first_tag.reserve(n_edges * sizeof(uint32_t))) return rc;
    // tags of an older call lose against this call's sequence number -- except after the 16-bit number has wrapped
    // (or in fresh memory): then the array is cleared once
    if (c->tag_clear_needed || c->first_tag.cap != before) {
      HIP_TRY(hipMemsetAsync(c->first_tag.p, 0, c->first_tag.cap, s));
      c->tag_clear_needed = false;
    }
    a.first_tag = static_cast<uint32_t*>(c->first_tag.p);
  }
  return MVK_OK;
}
// waits for the done word of call `seq`: a short poll (the kernel raises it from the device), then the stream.
// returns 1 when the kernel's grid barrier gave up (the caller repeats the call with two launches)
int fused_wait(mvk_ctx* c, uint32_t seq, hipStream_t s, int* gave_up) {
  volatile uint32_t* done = static_cast<volatile uint32_t*>(c->pin_ctl.p);
  *gave_up = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0;; ++spin) {
    const uint32_t v = __atomic_load_n(const_cast<const uint32_t*>(const_cast<uint32_t*>(done)), __ATOMIC_ACQUIRE);
    if ((v & 0x7FFFFFFFu) == seq) {
      if (v & 0x80000000u) { *gave_up = 1; ++c->fused_giveups; if (int rc_ = sync_stream(c, s)) return rc_; }
      // the done word is the last thing the launch writes: nothing of the ctx's scratch is in use any more
      if (c->order_armed && c->order_stream == s) c->order_armed = false;
      return MVK_OK;
    }
    __builtin_ia32_pause();
    if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
  }
  if (int rc_ = sync_stream(c, s)) return rc_;  // a long kernel (or a busy GPU): let the runtime wait
  const uint32_t v = *done;
  if ((v & 0x7FFFFFFFu) != seq) return fail(MVK_ERR_HIP, "fused flag path: the kernel finished without raising its done word");
  *gave_up = (v & 0x80000000u) ? 1 : 0;
  if (*gave_up) ++c->fused_giveups;
  return MVK_OK;
}

// flag path: size the hand-off buffers between the screen stage and the exact stage for n lanes and
// hand this call its set of counters
int prepare_queue(mvk_ctx* c, EvalArgs& a, uint64_t n) {
  if (n > 0xffffffffULL) return fail(MVK_ERR_CAPACITY, "at most 2^32-1 states (or edge steps) per call");
  const uint64_t n_seg = static_cast<uint64_t>(c->model.n_seg);
  const uint64_t waves = (n + kBlock - 1) / kBlock * (kBlock / 64);
  const uint64_t stride = (waves + kQueueShards - 1) / kQueueShards * 64;
  const uint64_t rec_doubles = 8 + 2 * static_cast<uint64_t>(c->model.n_joint);
  if (int rc = c->queue.reserve(n_seg * kQueueShards * stride * sizeof(uint32_t))) return rc;
  if (int rc = c->queue_rec.reserve(waves * 64 * rec_doubles * sizeof(double))) return rc;
  uint32_t* counts = static_cast<uint32_t*>(c->queue_counts.p);
  a.queue = static_cast<uint32_t*>(c->queue.p);
  a.queue_stride = stride;
  a.queue_count = counts + (c->call_index & 1) * mvk_ctx::kQueueCounterSet;
  a.queue_count_next = counts + ((c->call_index + 1) & 1) * mvk_ctx::kQueueCounterSet;
  ++c->call_index;
  a.queue_rec = static_cast<double*>(c->queue_rec.p);
  // the exact stage runs as many blocks as the chip holds at once (about three per CU)
  a.exact_blocks = static_cast<uint32_t>(3 * c->n_cu);
  if (c->flag_precision == MVK_PRECISION_F32_REFINED) {
    if (int rc = c->refine_list.reserve(waves * 64 * sizeof(uint32_t))) return rc;
    a.f32_walk = 1;
    a.refine_list = static_cast<uint32_t*>(c->refine_list.p);
    a.refine_count = a.queue_count + static_cast<size_t>(MVK_MAX_SEG * kQueueShards - 1) * kQueueCounterPitch;
    c->last_refine_counter = a.refine_count;
    const uint64_t blocks = (n + kBlock - 1) / kBlock;
    a.refine_blocks = static_cast<uint32_t>(std::min<uint64_t>(blocks, static_cast<uint64_t>(2 * c->n_cu)));
  }
  return MVK_OK;
}

struct F32Out { float* min_dist; float* margin; float* margin_err; double dist_tol; };

int run_eval_dev(mvk_ctx* c, const double* d_soa, size_t n, size_t ld, int layout, uint8_t* d_flags,
                 uint64_t* d_mask, double* d_min_dist, double* d_margin, uint8_t* d_band,
                 const mvk_detail_out* det, bool force_full, hipStream_t s, uint32_t poll_seq = 0,
                 const double* inline_rows = nullptr /* host rows of a single state, pitch 1: travel in the kernel arguments */,
                 const F32Out* f32 = nullptr /* the fp32 variant of the distance / margin outputs */) {
  if (c && c->edges_pending) return fail(MVK_ERR_INVALID_ARG, "an edge check begun with mvk_check_motion_batch_begin is still in flight on this ctx: call mvk_check_motion_batch_end first");
  if (!c) return fail(MVK_ERR_INVALID_ARG, "null ctx");
  if (n == 0) return MVK_OK;
  if (!d_soa || ld < n) return fail(MVK_ERR_INVALID_ARG, "bad state array (null, or ld < n)");
  const int rows = layout_rows(c, layout);
  if (rows < 0) return rows;
  HIP_TRY(hipSetDevice(c->device));
  EvalArgs a;
  base_args(c, a);
  a.soa = d_soa;
  a.n = n;
  a.ld = ld;
  a.layout = layout;
  a.flags = d_flags;
  a.mask = d_mask;
  a.min_dist = d_min_dist;
  a.margin = d_margin;
  a.band = d_band;
  bool want_detail = false;
  if (det) {
    a.fc_t_min = det->fc_t_min; a.fc_rp_min = det->fc_rp_min; a.thrust = det->thrust;
    a.rp_margin = det->rp_position_margin; a.cog = det->cog; a.gimbal = det->gimbal;
    want_detail = det->fc_t_min || det->fc_rp_min || det->thrust || det->rp_position_margin || det->cog || det->gimbal;
  }
  a.full = (force_full || d_min_dist || d_margin || d_band || want_detail) ? 1 : 0;
  if (f32) {
    a.full = 1;
    a.f32_variant = 1;
    a.min_dist_f32 = f32->min_dist;
    a.margin_f32 = f32->margin;
    a.margin_err_f32 = f32->margin_err;
    a.eps_dist = std::fmax(a.eps_dist, f32->dist_tol);  // the band takes what the variant cannot tell from a contact
  }
  c->stats_valid = false;
  if (a.full) {
    const uint64_t blocks = (n + kBlock - 1) / kBlock;
    // small batches with a collision stage: distance_small_kernel, a group of lanes per state (and a partial per
    // block of kBlock / kSmallGroup states)
    a.small_full = (!f32 && a.do_coll && a.n_obs > 0 && !d_mask && n <= c->small_full_max && c->small_full_enabled) ? 1 : 0;
    if (a.small_full && n <= kFullOneLaunchMax) a.small_full = 2;  // walk 1 in the same launch, two states per block
    if (a.small_full == 2 && n == 1 && inline_rows) {
      a.inline_on = 1;
      for (int r = 0; r < rows; ++r) a.inline_state[r] = inline_rows[r];
    }
    const uint64_t stat_blocks = a.small_full == 2 ? (n + 1) / 2 : a.small_full ? (n + (kBlock / kSmallGroup) - 1) / (kBlock / kSmallGroup) : blocks;
    if (poll_seq) {  // eval_host: outputs in pinned memory, completion through the pinned done word
      if (!a.small_full) return fail(MVK_ERR_INVALID_ARG, "internal: polled completion asked of a call that does not take distance_small_kernel");
      a.host_done = static_cast<volatile uint32_t*>(c->pin_ctl.p);
      a.fused_seq = poll_seq;
      a.fused_done = static_cast<uint32_t*>(c->small_done.p);
      a.small_target = c->small_total + static_cast<uint32_t>(stat_blocks);  // (committed once the launch has succeeded)
    }
    if (int rc = c->partial.reserve(stat_blocks * sizeof(StatsPartial))) return rc;
    a.stats_partial = c->partial.p;
    c->last_blocks = stat_blocks;
    c->last_n = n;
    if (a.do_coll && a.n_obs > 0 && a.small_full != 2) {  // the distance kernel follows: hand-off rows for every launched lane
      const uint64_t rec_ld = blocks * kBlock;
      if (int rc = c->full_rec.reserve((2 * static_cast<uint64_t>(c->model.n_joint) + 5) * rec_ld * sizeof(double))) return rc;
      a.full_rec = static_cast<double*>(c->full_rec.p);
      a.full_rec_ld = rec_ld;
    }
  } else if (int rc = prepare_queue(c, a, n)) {
    return rc;
  }
  if (int rc = enter_stream(c, s)) return rc;
  const int e = launch_eval(a, c->model.robot_type, c->model.n_rotor, false, s, c->profiling ? c->ev : nullptr);
  c->timed = c->profiling;
  if (e != hipSuccess) return fail(MVK_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(e)));
  if (poll_seq) c->small_total = a.small_target;
  c->stats_valid = a.full != 0;
  return leave_stream(c, s);
}

// copy a [rows][n] SoA block with row pitch ld (elements) between host and device
int copy_rows(void* dst, size_t dst_ld, const void* src, size_t src_ld, size_t elem, size_t n, size_t rows,
              hipMemcpyKind kind, hipStream_t s) {
  HIP_TRY(hipMemcpy2DAsync(dst, dst_ld * elem, src, src_ld * elem, n * elem, rows, kind, s));
  return MVK_OK;
}

size_t round_up(size_t n, size_t m) { return (n + m - 1) / m * m; }

}  // namespace

// ---------------------------------------------------------------------------
extern "C" int mvk_abi_version(void) { return MVK_ABI_VERSION; }
extern "C" size_t mvk_model_sizeof(void) { return sizeof(mvk_model); }
extern "C" size_t mvk_params_sizeof(void) { return sizeof(mvk_params); }
extern "C" const char* mvk_last_error(void) { return g_err; }

extern "C" int mvk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int usable = 0;
  for (int d = 0; d < n; ++d) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++usable;
  }
  return usable;
}

extern "C" int mvk_broadphase_query(const mvk_box* boxes, size_t n_boxes, double radius, const double* points,
                                    size_t n_points, int32_t* counts, int32_t* ids, int cap) {
  if (!(radius >= 0.0) || (n_points && (!points || !counts)) || cap < 0 || (cap > 0 && !ids))
    return fail(MVK_ERR_INVALID_ARG, "mvk_broadphase_query: bad argument");
  std::vector<DevObs> host;
  if (int rc = boxes_to_obs(boxes, n_boxes, host)) return rc;
  GridHost g;
  if (int rc = build_grid(host.data(), n_boxes, radius, g)) return rc;
  for (size_t i = 0; i < n_points; ++i) {
    const uint16_t* list = nullptr;
    const int padded = grid_query(g, points + 3 * i, &list);
    int k = 0;
    for (int e = 0; e < padded; ++e)
      if (list[e] != kGridNone) {
        if (k < cap) ids[i * static_cast<size_t>(cap) + k] = list[e];
        ++k;
      }
    counts[i] = k;
  }
  return MVK_OK;
}

extern "C" int mvk_debug_grid_bounds(const mvk_box* boxes, size_t n_boxes, double radius, double pad, const double* points,
                                     size_t n_points, int32_t* counts, int32_t* ids, float* bounds, int cap) {
  if (!(radius >= 0.0) || !(pad >= 0.0) || (n_points && (!points || !counts)) || cap < 0 || (cap > 0 && (!ids || !bounds)))
    return fail(MVK_ERR_INVALID_ARG, "mvk_debug_grid_bounds: bad argument");
  std::vector<DevObs> host;
  if (int rc = boxes_to_obs(boxes, n_boxes, host)) return rc;
  GridHost g;
  if (int rc = build_grid(host.data(), n_boxes, radius, g, true, pad)) return rc;
  for (size_t i = 0; i < n_points; ++i) {
    const uint16_t* list = nullptr;
    const int padded = grid_query(g, points + 3 * i, &list);
    const size_t group0 = padded ? static_cast<size_t>(list - g.items.data()) / kGridGroup : 0;
    int k = 0;
    for (int e = 0; e < padded; ++e)
      if (list[e] != kGridNone) {
        if (k < cap) {
          ids[i * static_cast<size_t>(cap) + k] = list[e];
          bounds[i * static_cast<size_t>(cap) + k] = g.glb[group0 + e / kGridGroup];
        }
        ++k;
      }
    counts[i] = k;
  }
  return MVK_OK;
}

extern "C" int mvk_far_grid_query(const mvk_box* boxes, size_t n_boxes, double reach, double margin, const double* points,
                                  size_t n_points, int32_t* counts, int32_t* ids, int cap) {
  if (!(reach >= 0.0) || !(margin >= 0.0) || (n_points && (!points || !counts)) || cap < 0 || (cap > 0 && !ids))
    return fail(MVK_ERR_INVALID_ARG, "mvk_far_grid_query: bad argument");
  std::vector<DevObs> host;
  if (int rc = boxes_to_obs(boxes, n_boxes, host)) return rc;
  GridHost g;
  if (int rc = build_far_grid(host.data(), n_boxes, reach, margin, 1e-6, g)) return rc;
  for (size_t i = 0; i < n_points; ++i) {
    const uint16_t* list = nullptr;
    const int padded = grid_query(g, points + 3 * i, &list);
    int k = 0;
    for (int e = 0; e < padded; ++e)
      if (list[e] != kGridNone) {
        if (k < cap) ids[i * static_cast<size_t>(cap) + k] = list[e];
        ++k;
      }
    counts[i] = padded ? k : -1;  // (a cell's list is never empty: it holds the box nearest to its centre)
  }
  return MVK_OK;
}

extern "C" mvk_ctx* mvk_create(const mvk_model* model, const mvk_box* boxes, size_t n_boxes,
                               const mvk_params* params, int device) {
  if (!model || !params) {
    fail(MVK_ERR_INVALID_ARG, "mvk_create: null model or params");
    return nullptr;
  }
  if (check_model_shape(model) != MVK_OK || check_params(model, params) != MVK_OK) return nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    fail(MVK_ERR_NO_DEVICE, "mvk_create: no HIP device available (this library has no CPU path)");
    return nullptr;
  }
  if (device < 0 || device >= count) {
    fail(MVK_ERR_INVALID_ARG, "mvk_create: device %d out of range (0..%d)", device, count - 1);
    return nullptr;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    fail(MVK_ERR_NO_DEVICE, "mvk_create: device %d is not gfx950 (kernels are built for MI355X only)", device);
    return nullptr;
  }
  mvk_ctx* c = new (std::nothrow) mvk_ctx();
  if (!c) {
    fail(MVK_ERR_CAPACITY, "mvk_create: out of host memory");
    return nullptr;
  }
  c->device = device;
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c->model = *model;
  c->params = *params;
  derive_extras(model, &c->extras);
  auto bail = [&](const char* what, hipError_t e) -> mvk_ctx* {
    fail(MVK_ERR_HIP, "mvk_create: %s: %s", what, hipGetErrorString(e));
    mvk_destroy(c);
    return nullptr;
  };
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  if ((e = hipEventCreateWithFlags(&c->order_ev, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = static_cast<hipError_t>(static_scratch_for_current_device(device, &c->static_scratch))) != hipSuccess)
    return bail("hipGetSymbolAddress(scratch)", e);
  if ((e = hipMalloc(reinterpret_cast<void**>(&c->d_model), sizeof(mvk_model))) != hipSuccess) return bail("hipMalloc(model)", e);
  if ((e = hipMalloc(reinterpret_cast<void**>(&c->d_extras), sizeof(DevExtras))) != hipSuccess) return bail("hipMalloc(extras)", e);
  if ((e = hipMemcpy(c->d_model, &c->model, sizeof(mvk_model), hipMemcpyHostToDevice)) != hipSuccess) return bail("upload model", e);
  if ((e = hipMemcpy(c->d_extras, &c->extras, sizeof(DevExtras), hipMemcpyHostToDevice)) != hipSuccess) return bail("upload extras", e);
  if (upload_scene(c, boxes, n_boxes) != MVK_OK) {
    mvk_destroy(c);
    return nullptr;
  }
  if (c->queue_counts.reserve(2 * mvk_ctx::kQueueCounterSet * sizeof(uint32_t)) != MVK_OK) {
    mvk_destroy(c);
    return nullptr;
  }
  if ((e = hipMemset(c->queue_counts.p, 0, 2 * mvk_ctx::kQueueCounterSet * sizeof(uint32_t))) != hipSuccess) return bail("hipMemset(queue counters)", e);
  if (c->small_done.reserve(256) != MVK_OK) {
    mvk_destroy(c);
    return nullptr;
  }
  if ((e = hipMemset(c->small_done.p, 0, 256)) != hipSuccess) return bail("hipMemset(small_done)", e);
  if (c->pin_ctl.reserve(4096) != MVK_OK) {
    mvk_destroy(c);
    return nullptr;
  }
  std::memset(c->pin_ctl.p, 0, 4096);
  if (const char* off = std::getenv("MVK_NO_SMALL_FULL")) c->small_full_enabled = off[0] != '1';  // diagnostics, tests
  if (const char* v = std::getenv("MVK_SMALL_FULL_MAX")) if (std::atoll(v) > 0) c->small_full_max = c->small_avoid_max = static_cast<uint64_t>(std::atoll(v));  // diagnostics
  if (const char* v = std::getenv("MVK_FLAG_SMALL_MAX")) if (std::atoll(v) > 0 && std::atoll(v) <= 1024) c->flag_small_max = static_cast<uint64_t>(std::atoll(v));  // diagnostics
  if (const char* v = std::getenv("MVK_FUSED_MAX_BLOCKS")) if (std::atoi(v) > 0 && std::atoi(v) <= kFusedHardMax) c->fused_max_blocks = std::atoi(v);  // diagnostics
  if (const char* off = std::getenv("MVK_NO_SPLIT")) c->split_enabled = off[0] != '1';  // diagnostics
  if (const char* v = std::getenv("MVK_FUSED_SPIN_LIMIT")) c->fused_spin_limit = static_cast<uint32_t>(std::strtoul(v, nullptr, 10));  // diagnostics: 0 forces the give-up path
  if (const char* off = std::getenv("MVK_NO_FUSED")) c->fused_enabled = off[0] != '1';  // diagnostics: always two launches
  return c;
}

extern "C" void mvk_destroy(mvk_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)drain_order_stream(c);  // the last call may be on a caller's stream (possibly gone by now)
  if (c->stream) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamDestroy(c->stream);
  }
  for (hipEvent_t& e : c->ev)
    if (e) (void)hipEventDestroy(e);
  if (c->order_ev) (void)hipEventDestroy(c->order_ev);
  if (c->d_model) (void)hipFree(c->d_model);
  if (c->d_extras) (void)hipFree(c->d_extras);
  Buf* all[] = {&c->obs, &c->obsf, &c->grid_start, &c->grid_items, &c->dgrid_start[0], &c->dgrid_items[0], &c->dgrid_start[1], &c->dgrid_items[1], &c->dgrid_glb[0], &c->dgrid_glb[1], &c->fgrid_start, &c->fgrid_items, &c->queue, &c->queue_rec, &c->queue_counts, &c->refine_list, &c->in_a, &c->in_b, &c->flags, &c->band, &c->mind, &c->margin, &c->det[0], &c->det[1],
                &c->det[2], &c->det[3], &c->det[4], &c->det[5], &c->nd, &c->ok, &c->first, &c->edge_off,
                &c->partial, &c->stats_out, &c->full_rec, &c->small_done, &c->av[0], &c->av[1], &c->av[2], &c->av[3], &c->av[4], &c->av[5], &c->av[6]};
  for (Buf* b : all) b->release();
  c->edge_blob.release();
  c->first_tag.release();
  c->pin_up.release();
  c->pin_down.release();
  c->pin_ctl.release();
  delete c;
}

extern "C" int mvk_set_scene(mvk_ctx* c, const mvk_box* boxes, size_t n_boxes) {
  if (c && c->edges_pending) return fail(MVK_ERR_INVALID_ARG, "an edge check begun with mvk_check_motion_batch_begin is still in flight on this ctx: call mvk_check_motion_batch_end first");
  if (!c) return fail(MVK_ERR_INVALID_ARG, "null ctx");
  HIP_TRY(hipSetDevice(c->device));
  if (int rc = wait_last_call(c)) return rc;  // the last call may be on a caller's stream and still reading the old scene
  if (int rc_ = sync_stream(c, c->stream)) return rc_;
  return upload_scene(c, boxes, n_boxes);
}

extern "C" int mvk_set_params(mvk_ctx* c, const mvk_params* p) {
  if (c && c->edges_pending) return fail(MVK_ERR_INVALID_ARG, "an edge check begun with mvk_check_motion_batch_begin is still in flight on this ctx: call mvk_check_motion_batch_end first");
  if (!c || !p) return fail(MVK_ERR_INVALID_ARG, "null argument");
  if (int rc = check_params(&c->model, p)) return rc;
  c->params = *p;
  return MVK_OK;
}

extern "C" int mvk_get_params(const mvk_ctx* c, mvk_params* out) {
  if (!c || !out) return fail(MVK_ERR_INVALID_ARG, "null argument");
  *out = c->params;
  return MVK_OK;
}

extern "C" int mvk_set_flag_precision(mvk_ctx* c, int precision) {
  if (!c) return fail(MVK_ERR_INVALID_ARG, "null ctx");
  if (c->edges_pending) return fail(MVK_ERR_INVALID_ARG, "an edge check begun with mvk_check_motion_batch_begin is still in flight on this ctx: call mvk_check_motion_batch_end first");
  if (precision != MVK_PRECISION_F64 && precision != MVK_PRECISION_F32_REFINED)
    return fail(MVK_ERR_INVALID_ARG, "mvk_set_flag_precision: unknown precision %d", precision);
  c->flag_precision = precision;
  c->f32_bound_scale = 1.0f;
  if (const char* e = std::getenv("MVK_F32_BOUND_SCALE")) {  // diagnostics: shrink the fp32 error bounds (tests of their safety margin)
    const double v = std::atof(e);
    if (v > 0.0 && v <= 64.0) c->f32_bound_scale = static_cast<float>(v);
  }
  return MVK_OK;
}

extern "C" int mvk_last_refined_count(mvk_ctx* c, uint64_t* out) {
  if (!c || !out) return fail(MVK_ERR_INVALID_ARG, "null argument");
  if (!c->last_refine_counter) return fail(MVK_ERR_INVALID_ARG, "mvk_last_refined_count: no call with MVK_PRECISION_F32_REFINED yet");
  HIP_TRY(hipSetDevice(c->device));
  if (int rc = wait_last_call(c)) return rc;  // the last call on this ctx, whatever its stream
  uint32_t v = 0;
  HIP_TRY(hipMemcpy(&v, c->last_refine_counter, sizeof(v), hipMemcpyDeviceToHost));
  *out = v;
  return MVK_OK;
}

extern "C" int mvk_set_profiling(mvk_ctx* c, int enable) {
  if (!c) return fail(MVK_ERR_INVALID_ARG, "null ctx");
  HIP_TRY(hipSetDevice(c->device));
  if (enable)
    for (hipEvent_t& e : c->ev)
      if (!e) HIP_TRY(hipEventCreate(&e));
  c->profiling = enable != 0;
  c->timed = false;
  return MVK_OK;
}

extern "C" int mvk_last_kernel_ms(mvk_ctx* c, float* first_ms, float* second_ms) {
  if (!c || !first_ms || !second_ms) return fail(MVK_ERR_INVALID_ARG, "null argument");
  if (!c->timed) return fail(MVK_ERR_INVALID_ARG, "mvk_last_kernel_ms: no timed call (enable mvk_set_profiling first)");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventSynchronize(c->ev[2]));
  HIP_TRY(hipEventElapsedTime(first_ms, c->ev[0], c->ev[1]));
  HIP_TRY(hipEventElapsedTime(second_ms, c->ev[1], c->ev[2]));
  return MVK_OK;
}

extern "C" int mvk_state_dim(const mvk_ctx* c, int layout) {
  if (!c) return fail(MVK_ERR_INVALID_ARG, "null ctx");
  return layout_rows(c, layout);
}

// ---- batched predicate --------------------------------------------------------
extern "C" int mvk_is_valid_batch_dev(mvk_ctx* c, const double* d_soa, size_t n, size_t ld, int layout,
                                      uint8_t* d_flags, double* d_min_dist, double* d_margin, uint8_t* d_band,
                                      void* stream) {
  return run_eval_dev(c, d_soa, n, ld, layout, d_flags, nullptr, d_min_dist, d_margin, d_band, nullptr, false,
                      static_cast<hipStream_t>(stream));
}

extern "C" int mvk_is_valid_mask_dev(mvk_ctx* c, const double* d_soa, size_t n, size_t ld, int layout,
                                     uint64_t* d_mask, void* stream) {
  if (!d_mask) return fail(MVK_ERR_INVALID_ARG, "null mask");
  return run_eval_dev(c, d_soa, n, ld, layout, nullptr, d_mask, nullptr, nullptr, nullptr, nullptr, false,
                      static_cast<hipStream_t>(stream));
}

// Diagnostic (include/mvk_debug.h): host calls whose fused launch gave up at its grid barrier
extern "C" long long mvk_debug_fused_giveups(mvk_ctx* c, int reset) {
  if (!c) return -1;
  const long long v = c->fused_giveups;
  if (reset) c->fused_giveups = 0;
  return v;
}

// Diagnostic (include/mvk_debug.h; tests/test_gpu_f32.py, profiles/f32_bounds.py): one pass of the fp32 screen
// stage over a batch that also writes, per state, what its stability classification worked with: d_lo = the
// smallest (slack - error bound) over the tests it could bound, d_hi = the smallest (slack + error bound), d_cls =
// 1 certainly stable / 0 certainly unstable / -1 uncertain (refined in fp64) / -2 non-finite input.
extern "C" int mvk_debug_f32_bounds(mvk_ctx* c, const double* d_soa, size_t n, size_t ld, int layout, uint8_t* d_flags,
                                    double* d_lo, double* d_hi, double* d_cls, void* stream) {
  if (!c || !d_soa || !d_lo || !d_hi || !d_cls || ld < n) return fail(MVK_ERR_INVALID_ARG, "mvk_debug_f32_bounds: bad argument");
  if (n == 0) return MVK_OK;
  const int rows = layout_rows(c, layout);
  if (rows < 0) return rows;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int keep = c->flag_precision;
  c->flag_precision = MVK_PRECISION_F32_REFINED;
  EvalArgs a;
  base_args(c, a);
  a.soa = d_soa; a.n = n; a.ld = ld; a.layout = layout; a.flags = d_flags;
  a.fc_t_min = d_lo; a.fc_rp_min = d_hi; a.rp_margin = d_cls;
  c->stats_valid = false;
  const int rc = prepare_queue(c, a, n);
  c->flag_precision = keep;
  if (rc) return rc;
  if (int rc2 = enter_stream(c, s)) return rc2;
  const int e = launch_eval(a, c->model.robot_type, c->model.n_rotor, false, s, nullptr);
  if (e != hipSuccess) return fail(MVK_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(e)));
  return leave_stream(c, s);
}

// The flag path with its hand-off buffers in the library's static __device__ scratch array: no hipMalloc, no
// hipFree, no allocating routine on this path (the ctx's grow-only buffers are not touched).
extern "C" int mvk_is_valid_mask_noalloc_dev(mvk_ctx* c, const double* d_soa, size_t n, size_t ld, int layout,
                                             uint64_t* d_mask, void* stream) {
  if (!c) return fail(MVK_ERR_INVALID_ARG, "null ctx");
  if (c->edges_pending) return fail(MVK_ERR_INVALID_ARG, "an edge check begun with mvk_check_motion_batch_begin is still in flight on this ctx: call mvk_check_motion_batch_end first");
  if (!d_mask) return fail(MVK_ERR_INVALID_ARG, "null mask");
  if (n == 0) return MVK_OK;
  if (n > MVK_NOALLOC_MAX_STATES)
    return fail(MVK_ERR_CAPACITY, "mvk_is_valid_mask_noalloc_dev: %zu states exceed the static scratch (at most %d per call)", n, MVK_NOALLOC_MAX_STATES);
  if (!d_soa || ld < n) return fail(MVK_ERR_INVALID_ARG, "bad state array (null, or ld < n)");
  const int rows = layout_rows(c, layout);
  if (rows < 0) return rows;
  if (!c->static_scratch) return fail(MVK_ERR_HIP, "static scratch not resolved on device %d", c->device);
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  EvalArgs a;
  base_args(c, a);
  a.soa = d_soa;
  a.n = n;
  a.ld = ld;
  a.layout = layout;
  a.mask = d_mask;
  // scratch layout: [counter set of this call][counter set the kernel zeroes "for the next call"][queues][records]
  unsigned char* base = static_cast<unsigned char*>(c->static_scratch);
  const uint64_t waves = (n + kBlock - 1) / kBlock * (kBlock / 64);
  a.queue_stride = (waves + kQueueShards - 1) / kQueueShards * 64;
  a.queue_count = reinterpret_cast<uint32_t*>(base);
  a.queue_count_next = reinterpret_cast<uint32_t*>(base + kStaticCounterSetBytes);
  a.queue = reinterpret_cast<uint32_t*>(base + 2 * kStaticCounterSetBytes);
  a.queue_rec = reinterpret_cast<double*>(base + 2 * kStaticCounterSetBytes + kStaticQueueBytes);
  a.exact_blocks = static_cast<uint32_t>(3 * c->n_cu);
  // what the kernels use must fit what the array holds (n <= MVK_NOALLOC_MAX_STATES and the model's capacities)
  const uint64_t need_queue = static_cast<uint64_t>(c->model.n_seg) * kQueueShards * a.queue_stride * sizeof(uint32_t);
  const uint64_t need_rec = waves * 64 * (8 + 2 * static_cast<uint64_t>(c->model.n_joint)) * sizeof(double);
  if (need_queue > kStaticQueueBytes || need_rec > kStaticRecordBytes)
    return fail(MVK_ERR_CAPACITY, "mvk_is_valid_mask_noalloc_dev: batch does not fit the static scratch");
  c->stats_valid = false;
  if (int rc = enter_stream(c, s)) return rc;
  // the array keeps the last call's contents (any ctx of this GPU): the counters must start at zero
  if (const int ez = launch_zero_words(reinterpret_cast<uint32_t*>(base), kStaticCounterSetBytes / sizeof(uint32_t), s))
    return fail(MVK_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(ez)));
  // (under graph capture the profiling events would be recorded into the graph: left out)
  const bool timed = c->profiling && !stream_capturing(s);
  const int e = launch_eval(a, c->model.robot_type, c->model.n_rotor, false, s, timed ? c->ev : nullptr);
  c->timed = timed;
  if (e != hipSuccess) return fail(MVK_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(e)));
  return leave_stream(c, s);
}

extern "C" int mvk_sweep_mask_dev(mvk_ctx* c, const double* root_pose7, uint64_t first_index, size_t n, uint64_t seed,
                                  uint64_t* d_mask, void* stream) {
  if (c && c->edges_pending) return fail(MVK_ERR_INVALID_ARG, "an edge check begun with mvk_check_motion_batch_begin is still in flight on this ctx: call mvk_check_motion_batch_end first");
  if (!c || !root_pose7 || !d_mask) return fail(MVK_ERR_INVALID_ARG, "null argument");
  if (n == 0) return MVK_OK;
  HIP_TRY(hipSetDevice(c->device));
  EvalArgs a;
  base_args(c, a);
  a.n = n;
  a.ld = n;
  a.layout = MVK_LAYOUT_POSE7;
  a.mask = d_mask;
  a.sweep_on = 1;
  a.sweep_first = first_index;
  a.sweep_seed = seed;
  for (int k = 0; k < 7; ++k) a.sweep_pose[k] = root_pose7[k];
  for (int j = 0; j < c->model.n_joint; ++j) {
    a.sweep_lo[j] = c->params.joint_low[j];
    a.sweep_span[j] = c->params.joint_high[j] - c->params.joint_low[j];
  }
  c->stats_valid = false;
  if (int rc = prepare_queue(c, a, n)) return rc;
  if (int rc = enter_stream(c, static_cast<hipStream_t>(stream))) return rc;
  const int e = launch_eval(a, c->model.robot_type, c->model.n_rotor, false, static_cast<hipStream_t>(stream), c->profiling ? c->ev : nullptr);
  c->timed = c->profiling;
  if (e != hipSuccess) return fail(MVK_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(e)));
  return leave_stream(c, static_cast<hipStream_t>(stream));
}

extern "C" int mvk_sweep_states(const mvk_params* prm, int J, const double* root_pose7, uint64_t first_index, size_t n,
                                uint64_t seed, double* soa, size_t ld) {
  if (!prm || !root_pose7 || (n && !soa) || ld < n || J < 0 || J > MVK_MAX_JOINT) return fail(MVK_ERR_INVALID_ARG, "mvk_sweep_states: bad argument");
  for (size_t q = 0; q < n; ++q) {
    const uint64_t i = first_index + q;
    for (int r = 0; r < 7; ++r) soa[r * ld + q] = root_pose7[r];
    for (int b = 0; b < (J + 3) / 4; ++b) {
      uint32_t u[4];
      philox4x32_10(static_cast<uint32_t>(i), static_cast<uint32_t>(i >> 32), static_cast<uint32_t>(b), 0u,
                    static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), u);
      for (int k = 0; k < 4 && 4 * b + k < J; ++k) {
        const int j = 4 * b + k;
        soa[(7 + j) * ld + q] = std::fma(prm->joint_high[j] - prm->joint_low[j], sweep_unit(u[k]), prm->joint_low[j]);
      }
    }
  }
  return MVK_OK;
}

extern "C" int mvk_eval_batch_dev(mvk_ctx* c, const double* d_soa, size_t n, size_t ld, int layout,
                                  uint8_t* d_flags, double* d_min_dist, double* d_margin, uint8_t* d_band,
                                  const mvk_detail_out* d_detail, void* stream) {
  return run_eval_dev(c, d_soa, n, ld, layout, d_flags, nullptr, d_min_dist, d_margin, d_band, d_detail, true,
                      static_cast<hipStream_t>(stream));
}

// fp32 variant of the distance / margin outputs (include/mvk.h): model32_kernel + distance_kernel<.., true>
extern "C" double mvk_f32_distance_tolerance(mvk_ctx* c, double coord_abs_max) {
  if (!c || !(coord_abs_max >= 0.0)) return NAN;
  // one fp32 rounding is 2^-24 relative to a coordinate of the state's primitives or of the boxes near them
  // (|root position| + the robot's reach, + 1 m for the boxes' own extent); the narrowphase chains a few dozen
  return MVK_F32_DIST_TOL_ULPS * 5.9604644775390625e-8 * (coord_abs_max + c->extras.reach + 1.0);
}

extern "C" int mvk_eval_batch_f32_dev(mvk_ctx* c, const double* d_soa, size_t n, size_t ld, int layout, uint8_t* d_flags,
                                      float* d_min_dist, float* d_margin, float* d_margin_err, uint8_t* d_band, void* stream) {
  if (!c) return fail(MVK_ERR_INVALID_ARG, "null ctx");
  const F32Out f32{d_min_dist, d_margin, d_margin_err, 0.0};  // (the band's distance tolerance is per state: distance_kernel)
  return run_eval_dev(c, d_soa, n, ld, layout, d_flags, nullptr, nullptr, nullptr, d_band, nullptr, true, static_cast<hipStream_t>(stream), 0,
                      nullptr, &f32);
}

namespace {
int eval_host(mvk_ctx* c, const double* soa, size_t n, size_t ld, int layout, uint8_t* flags, double* min_dist,
              double* margin, uint8_t* band, const mvk_detail_out* det, bool force_full) {
  if (!c) return fail(MVK_ERR_INVALID_ARG, "null ctx");
  if (c->edges_pending) return fail(MVK_ERR_INVALID_ARG, "an edge check is in flight on this ctx: call mvk_check_motion_batch_end first");
  if (n == 0) return MVK_OK;
  if (!soa || ld < n) return fail(MVK_ERR_INVALID_ARG, "bad state array (null, or ld < n)");
  const int rows = layout_rows(c, layout);
  if (rows < 0) return rows;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  if (int rc = enter_stream(c, s)) return rc;  // the staging buffers may still feed a call on a caller's stream
  const size_t dld = round_up(n, 64);
  const int NL = c->model.n_rotor;
  if (int rc = c->in_a.reserve(dld * rows * sizeof(double))) return rc;
  // Small batches (down to the single state of an OMPL isValid call, or the few hundred states of a densified
  // path) are shaped for latency like the edge checks: the rows packed into one pinned block, the outputs gathered
  // in another, every copy asynchronous, one synchronisation.  Large ones copy straight from / to the caller's memory.
  // Flag-only requests of at most kFusedMaxBlocks blocks (a single isValid, a densified path): ONE launch that reads
  // the rows from, and writes the verdicts to, pinned host memory and raises a done word there -- no copy, no
  // stream synchronisation (fused_kernel).
  if (flags && !min_dist && !margin && !band && !det && !force_full && fused_eligible(c, n)) {
    if (int rc = c->pin_up.reserve(dld * rows * sizeof(double))) return rc;
    if (int rc = c->pin_down.reserve(dld)) return rc;
    double* up = static_cast<double*>(c->pin_up.p);
    for (int r = 0; r < rows; ++r) std::memcpy(up + static_cast<size_t>(r) * dld, soa + static_cast<size_t>(r) * ld, n * sizeof(double));
    EvalArgs a;
    base_args(c, a);
    if (n <= c->flag_small_max && c->small_full_enabled) {
      // a state or a few (OMPL's isValid): a block of two waves per state, walk 1 and the collision test side by side
      a.soa = up;
      a.n = n;
      a.ld = dld;
      a.layout = layout;
      if (n == 1) {  // a single state travels in the kernel arguments
        a.inline_on = 1;
        for (int r = 0; r < rows; ++r) a.inline_state[r] = soa[static_cast<size_t>(r) * ld];
      }
      a.host_flags = static_cast<uint8_t*>(c->pin_down.p);
      a.host_done = static_cast<volatile uint32_t*>(c->pin_ctl.p);
      a.fused_seq = next_seq(c);
      a.fused_done = static_cast<uint32_t*>(c->small_done.p);
      a.small_target = c->small_total + static_cast<uint32_t>(n);
      c->stats_valid = false;
      if (c->profiling) (void)hipEventRecord(c->ev[0], s);
      const int e = launch_flag_small(a, c->model.robot_type, c->model.n_rotor, false, s);
      if (e != hipSuccess) return fail(MVK_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(e)));
      c->small_total = a.small_target;
      if (c->profiling) { (void)hipEventRecord(c->ev[1], s); (void)hipEventRecord(c->ev[2], s); }
      c->timed = c->profiling;
      if (int rc = leave_stream(c, s)) return rc;
      int gave_up = 0;
      if (int rc = fused_wait(c, a.fused_seq, s, &gave_up)) return rc;
      std::memcpy(flags, c->pin_down.p, n);
      return MVK_OK;
    }
    a.soa = up;  // a few states: the kernel reads the pinned rows itself; more: one copy up first
    if (n > 1024) {
      HIP_TRY(hipMemcpyAsync(c->in_a.p, up, dld * rows * sizeof(double), hipMemcpyHostToDevice, s));
      a.soa = static_cast<const double*>(c->in_a.p);
    }
    a.n = n;
    a.ld = dld;
    a.layout = layout;
    if (int rc = c->flags.reserve(dld)) return rc;
    a.flags = static_cast<uint8_t*>(c->flags.p);
    a.host_flags = static_cast<uint8_t*>(c->pin_down.p);
    c->stats_valid = false;
    if (int rc = prepare_queue(c, a, n)) return rc;
    if (int rc = fused_arm(c, a, s, 0)) return rc;
    const int e = launch_eval(a, c->model.robot_type, c->model.n_rotor, false, s, c->profiling ? c->ev : nullptr);
    c->timed = c->profiling;
    if (e != hipSuccess) return fail(MVK_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(e)));
    if (int rc = leave_stream(c, s)) return rc;
    int gave_up = 0;
    if (int rc = fused_wait(c, a.fused_seq, s, &gave_up)) return rc;
    if (!gave_up) {
      std::memcpy(flags, c->pin_down.p, n);
      return MVK_OK;
    }
    // (the grid barrier gave up -- the blocks were not co-resident in time; the two-launch path below repeats the call)
  }
  // Full path on a handful of states without detail rows (mvk_is_valid, clearance queries): the kernels read the rows
  // from, and write the outputs to, pinned host memory, and the last block of distance_small_kernel raises the done
  // word -- no copy, no stream synchronisation
  if (!det && n <= 1024 && n <= c->small_full_max && (force_full || min_dist || margin || band)) {
    EvalArgs probe;
    base_args(c, probe);
    if (probe.do_coll && probe.n_obs > 0 && c->small_full_enabled) {
      if (int rc = c->pin_up.reserve(dld * rows * sizeof(double))) return rc;
      if (int rc = c->pin_down.reserve(2 * dld + 2 * dld * sizeof(double))) return rc;
      double* up = static_cast<double*>(c->pin_up.p);
      for (int r = 0; r < rows; ++r) std::memcpy(up + static_cast<size_t>(r) * dld, soa + static_cast<size_t>(r) * ld, n * sizeof(double));
      unsigned char* down = static_cast<unsigned char*>(c->pin_down.p);
      double* down_d = reinterpret_cast<double*>(down + 2 * dld);
      const uint32_t seq = next_seq(c);
      double one[MVK_MAX_STATE_DIM];
      if (n == 1) for (int r = 0; r < rows; ++r) one[r] = soa[static_cast<size_t>(r) * ld];
      if (int rc = run_eval_dev(c, up, n, dld, layout, flags ? down : nullptr, nullptr, min_dist ? down_d : nullptr,
                                margin ? down_d + dld : nullptr, band ? down + dld : nullptr, nullptr, force_full, s, seq,
                                n == 1 ? one : nullptr))
        return rc;
      int gave_up = 0;
      if (int rc = fused_wait(c, seq, s, &gave_up)) return rc;
      if (flags) std::memcpy(flags, down, n);
      if (band) std::memcpy(band, down + dld, n);
      if (min_dist) std::memcpy(min_dist, down_d, n * sizeof(double));
      if (margin) std::memcpy(margin, down_d + dld, n * sizeof(double));
      return MVK_OK;
    }
  }
  const bool staged = n <= (1u << 16);
  if (staged) {
    if (int rc = c->pin_up.reserve(dld * rows * sizeof(double))) return rc;
    double* up = static_cast<double*>(c->pin_up.p);
    for (int r = 0; r < rows; ++r) std::memcpy(up + static_cast<size_t>(r) * dld, soa + static_cast<size_t>(r) * ld, n * sizeof(double));
    HIP_TRY(hipMemcpyAsync(c->in_a.p, up, dld * rows * sizeof(double), hipMemcpyHostToDevice, s));
  } else {
    if (int rc = copy_rows(c->in_a.p, dld, soa, ld, sizeof(double), n, rows, hipMemcpyHostToDevice, s)) return rc;
  }
  if (flags) if (int rc = c->flags.reserve(dld)) return rc;
  if (band) if (int rc = c->band.reserve(dld)) return rc;
  if (min_dist) if (int rc = c->mind.reserve(dld * sizeof(double))) return rc;
  if (margin) if (int rc = c->margin.reserve(dld * sizeof(double))) return rc;
  mvk_detail_out dd{};
  const size_t det_rows[6] = {1, 1, static_cast<size_t>(NL), 1, 3, static_cast<size_t>(2 * NL)};
  double* host_det[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (det) {
    host_det[0] = det->fc_t_min; host_det[1] = det->fc_rp_min; host_det[2] = det->thrust;
    host_det[3] = det->rp_position_margin; host_det[4] = det->cog; host_det[5] = det->gimbal;
    if (det->gimbal && !c->model.has_gimbal) return fail(MVK_ERR_INVALID_ARG, "gimbal output requested for a robot without gimbals");
    double** dev_slot[6] = {&dd.fc_t_min, &dd.fc_rp_min, &dd.thrust, &dd.rp_position_margin, &dd.cog, &dd.gimbal};
    for (int k = 0; k < 6; ++k)
      if (host_det[k]) {
        if (int rc = c->det[k].reserve(dld * det_rows[k] * sizeof(double))) return rc;
        *dev_slot[k] = static_cast<double*>(c->det[k].p);
      }
  }
  // the device detail rows use pitch dld, which run_eval_dev reads from `ld`
  if (int rc = run_eval_dev(c, static_cast<const double*>(c->in_a.p), n, dld, layout,
                            flags ? static_cast<uint8_t*>(c->flags.p) : nullptr, nullptr,
                            min_dist ? static_cast<double*>(c->mind.p) : nullptr,
                            margin ? static_cast<double*>(c->margin.p) : nullptr,
                            band ? static_cast<uint8_t*>(c->band.p) : nullptr, det ? &dd : nullptr, force_full, s))
    return rc;
  if (staged) {
    // pinned block: [flags dld][band dld][min_dist dld][margin dld][detail rows, pitch dld]
    size_t det_total = 0;
    for (int k = 0; k < 6; ++k) if (host_det[k]) det_total += det_rows[k];
    if (int rc = c->pin_down.reserve(2 * dld + (2 + det_total) * dld * sizeof(double))) return rc;
    unsigned char* down = static_cast<unsigned char*>(c->pin_down.p);
    double* down_d = reinterpret_cast<double*>(down + 2 * dld);
    if (flags) HIP_TRY(hipMemcpyAsync(down, c->flags.p, n, hipMemcpyDeviceToHost, s));
    if (band) HIP_TRY(hipMemcpyAsync(down + dld, c->band.p, n, hipMemcpyDeviceToHost, s));
    if (min_dist) HIP_TRY(hipMemcpyAsync(down_d, c->mind.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
    if (margin) HIP_TRY(hipMemcpyAsync(down_d + dld, c->margin.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
    size_t at = 2;
    for (int k = 0; k < 6; ++k)
      if (host_det[k]) {
        HIP_TRY(hipMemcpyAsync(down_d + at * dld, c->det[k].p, det_rows[k] * dld * sizeof(double), hipMemcpyDeviceToHost, s));
        at += det_rows[k];
      }
    if (int rc_ = sync_stream(c, s)) return rc_;
    if (flags) std::memcpy(flags, down, n);
    if (band) std::memcpy(band, down + dld, n);
    if (min_dist) std::memcpy(min_dist, down_d, n * sizeof(double));
    if (margin) std::memcpy(margin, down_d + dld, n * sizeof(double));
    at = 2;
    for (int k = 0; k < 6; ++k)
      if (host_det[k]) {
        for (size_t r = 0; r < det_rows[k]; ++r) std::memcpy(host_det[k] + r * ld, down_d + (at + r) * dld, n * sizeof(double));
        at += det_rows[k];
      }
    return MVK_OK;
  }
  if (flags) HIP_TRY(hipMemcpyAsync(flags, c->flags.p, n, hipMemcpyDeviceToHost, s));
  if (band) HIP_TRY(hipMemcpyAsync(band, c->band.p, n, hipMemcpyDeviceToHost, s));
  if (min_dist) HIP_TRY(hipMemcpyAsync(min_dist, c->mind.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
  if (margin) HIP_TRY(hipMemcpyAsync(margin, c->margin.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
  for (int k = 0; k < 6; ++k)
    if (host_det[k])
      if (int rc = copy_rows(host_det[k], ld, c->det[k].p, dld, sizeof(double), n, det_rows[k], hipMemcpyDeviceToHost, s)) return rc;
  if (int rc_ = sync_stream(c, s)) return rc_;
  return MVK_OK;
}
}  // namespace

extern "C" int mvk_is_valid_batch(mvk_ctx* c, const double* soa, size_t n, size_t ld, int layout, uint8_t* flags,
                                  double* min_dist, double* margin, uint8_t* band) {
  if (!flags && !min_dist && !margin && !band) return fail(MVK_ERR_INVALID_ARG, "no output requested");
  return eval_host(c, soa, n, ld, layout, flags, min_dist, margin, band, nullptr, false);
}

extern "C" int mvk_eval_batch(mvk_ctx* c, const double* soa, size_t n, size_t ld, int layout, uint8_t* flags,
                              double* min_dist, double* margin, uint8_t* band, const mvk_detail_out* detail) {
  return eval_host(c, soa, n, ld, layout, flags, min_dist, margin, band, detail, true);
}

extern "C" int mvk_is_valid(mvk_ctx* c, const double* state, mvk_result* out) {
  if (!c || !state || !out) return fail(MVK_ERR_INVALID_ARG, "null argument");
  uint8_t flag = 0, band = 0;
  double md = 0.0, mg = 0.0;
  if (int rc = eval_host(c, state, 1, 1, MVK_LAYOUT_OMPL, &flag, &md, &mg, &band, nullptr, true)) return rc;
  EvalArgs a;
  base_args(c, a);
  out->valid = flag;
  out->band = band;
  out->min_dist = md;
  out->margin = mg;
  out->stable = mg >= 0.0 ? 1 : 0;
  out->collision = (a.do_coll && md <= 0.0) ? 1 : 0;
  return MVK_OK;
}

extern "C" int mvk_reduce_stats(mvk_ctx* c, mvk_stats* out) {
  if (!c || !out) return fail(MVK_ERR_INVALID_ARG, "null argument");
  if (!c->stats_valid) return fail(MVK_ERR_INVALID_ARG, "mvk_reduce_stats: the last call on this ctx was not a full evaluation");
  HIP_TRY(hipSetDevice(c->device));
  if (int rc = c->stats_out.reserve(sizeof(StatsPartial))) return rc;
  // The evaluating kernel may have been enqueued on a caller stream: wait for the device.
  HIP_TRY(hipDeviceSynchronize());
  const int e = launch_stats_final(static_cast<const StatsPartial*>(c->partial.p), c->last_blocks,
                                   static_cast<StatsPartial*>(c->stats_out.p), c->stream);
  if (e != hipSuccess) return fail(MVK_ERR_HIP, "stats kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(e)));
  StatsPartial h;
  HIP_TRY(hipMemcpyAsync(&h, c->stats_out.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  if (int rc_ = sync_stream(c, c->stream)) return rc_;
  out->n = c->last_n;
  out->valid_count = h.valid_count;
  out->band_count = h.band_count;
  out->min_dist = h.min_dist; out->min_dist_index = h.min_dist_index;
  out->min_margin = h.min_margin; out->min_margin_index = h.min_margin_index;
  out->max_thrust = h.max_thrust; out->max_thrust_index = h.max_thrust_index;
  out->min_thrust = h.min_thrust; out->min_thrust_index = h.min_thrust_index;
  out->min_rp_position_margin = h.min_rp; out->min_rp_position_margin_index = h.min_rp_index;
  return MVK_OK;
}

// ---- collision-avoidance constraint rows ------------------------------------------
extern "C" int mvk_avoid_params_default(mvk_avoid_params* p) {
  if (!p) return fail(MVK_ERR_INVALID_ARG, "null argument");
  p->damping_gain = 0.01;      // collision_avoidance.cpp:68
  p->constraint_range = 0.03;  // :69
  p->forbidden_range = 0.01;   // :70
  p->full_body = 1;
  return MVK_OK;
}

namespace {
int run_avoid_dev(mvk_ctx* c, const double* d_soa, size_t n, size_t ld, int layout, const mvk_avoid_params* prm,
                  const mvk_avoid_out* o, void* stream, uint32_t poll_seq) {
  if (c && c->edges_pending) return fail(MVK_ERR_INVALID_ARG, "an edge check begun with mvk_check_motion_batch_begin is still in flight on this ctx: call mvk_check_motion_batch_end first");
  if (!c || !prm || !o) return fail(MVK_ERR_INVALID_ARG, "null argument");
  if (n == 0) return MVK_OK;
  if (!d_soa || ld < n) return fail(MVK_ERR_INVALID_ARG, "bad state array (null, or ld < n)");
  if (!(prm->constraint_range > prm->forbidden_range)) return fail(MVK_ERR_INVALID_ARG, "constraint_range must exceed forbidden_range");
  const int rows = layout_rows(c, layout);
  if (rows < 0) return rows;
  HIP_TRY(hipSetDevice(c->device));
  EvalArgs a;
  base_args(c, a);
  a.soa = d_soa;
  a.n = n;
  a.ld = ld;
  a.layout = layout;
  a.dgrid = c->dgrid[1];
  a.dgrid_radius = c->dgrid_radius[1];
  a.av_gain = prm->damping_gain;
  a.av_range = prm->constraint_range;
  a.av_forbid = prm->forbidden_range;
  a.av_full_body = prm->full_body;
  a.av_dist = o->dist; a.av_box = o->box; a.av_p_env = o->p_env; a.av_p_robot = o->p_robot;
  a.av_row = o->row; a.av_lb = o->lb; a.av_ok = o->ok;
  // small batches (the one state of a QP iteration): avoid_small_kernel, a wave per state
  a.small_full = (n <= c->small_avoid_max && c->small_full_enabled) ? 1 : 0;
  if (poll_seq) {  // host entry: outputs in pinned memory, completion through the pinned done word
    if (!a.small_full) return fail(MVK_ERR_INVALID_ARG, "internal: polled completion asked of a call that does not take avoid_small_kernel");
    a.host_done = static_cast<volatile uint32_t*>(c->pin_ctl.p);
    a.fused_seq = poll_seq;
    a.fused_done = static_cast<uint32_t*>(c->small_done.p);
    a.small_target = c->small_total + static_cast<uint32_t>((n + (kBlock / 64) - 1) / (kBlock / 64));
  }
  c->stats_valid = false;
  if (int rc = enter_stream(c, static_cast<hipStream_t>(stream))) return rc;
  const int e = launch_avoid(a, c->model.robot_type, c->model.n_rotor, stream);
  if (e != hipSuccess) return fail(MVK_ERR_HIP, "avoidance kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(e)));
  if (poll_seq) c->small_total = a.small_target;
  return leave_stream(c, static_cast<hipStream_t>(stream));
}
}  // namespace

extern "C" int mvk_collision_avoidance_batch_dev(mvk_ctx* c, const double* d_soa, size_t n, size_t ld, int layout,
                                                 const mvk_avoid_params* prm, const mvk_avoid_out* o, void* stream) {
  return run_avoid_dev(c, d_soa, n, ld, layout, prm, o, stream, 0);
}

extern "C" int mvk_collision_avoidance_batch(mvk_ctx* c, const double* soa, size_t n, size_t ld, int layout,
                                             const mvk_avoid_params* prm, const mvk_avoid_out* o) {
  if (!c || !prm || !o) return fail(MVK_ERR_INVALID_ARG, "null argument");
  if (c->edges_pending) return fail(MVK_ERR_INVALID_ARG, "an edge check begun with mvk_check_motion_batch_begin is still in flight on this ctx: call mvk_check_motion_batch_end first");
  if (n == 0) return MVK_OK;
  if (!soa || ld < n) return fail(MVK_ERR_INVALID_ARG, "bad state array (null, or ld < n)");
  if (!(prm->constraint_range > prm->forbidden_range)) return fail(MVK_ERR_INVALID_ARG, "constraint_range must exceed forbidden_range");
  const int rows = layout_rows(c, layout);
  if (rows < 0) return rows;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  if (int rc = enter_stream(c, s)) return rc;  // the staging buffers may still feed a call on a caller's stream
  const size_t dld = round_up(n, 64), P = static_cast<size_t>(c->model.n_prim), W = 6 + static_cast<size_t>(c->model.n_joint);
  void* host[7] = {o->dist, o->box, o->p_env, o->p_robot, o->row, o->lb, o->ok};
  const size_t nrows[7] = {P, P, 3 * P, 3 * P, W * P, P, 1};
  const size_t elem[7] = {8, 4, 8, 8, 8, 8, 1};
  // A few states (the one state of a QP iteration): the kernel reads the rows from, and writes every output to,
  // pinned host memory and its last block raises the done word -- no copy, no stream synchronisation
  if (n <= 256 && n <= c->small_avoid_max && c->small_full_enabled) {
    if (int rc = c->pin_up.reserve(dld * rows * sizeof(double))) return rc;
    double* up = static_cast<double*>(c->pin_up.p);
    for (int r = 0; r < rows; ++r) std::memcpy(up + static_cast<size_t>(r) * dld, soa + static_cast<size_t>(r) * ld, n * sizeof(double));
    size_t off[7], total = 0;
    for (int k = 0; k < 7; ++k) {
      off[k] = total;
      if (host[k]) total += round_up(dld * nrows[k] * elem[k], 256);
    }
    if (int rc = c->pin_down.reserve(total ? total : 256)) return rc;
    unsigned char* down = static_cast<unsigned char*>(c->pin_down.p);
    void* pd[7];
    for (int k = 0; k < 7; ++k) pd[k] = host[k] ? down + off[k] : nullptr;
    mvk_avoid_out d{static_cast<double*>(pd[0]), static_cast<int32_t*>(pd[1]), static_cast<double*>(pd[2]), static_cast<double*>(pd[3]),
                    static_cast<double*>(pd[4]), static_cast<double*>(pd[5]), static_cast<uint8_t*>(pd[6])};
    const uint32_t seq = next_seq(c);
    static const bool trace = std::getenv("MVK_TRACE_SMALL") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (int rc = run_avoid_dev(c, up, n, dld, layout, prm, &d, s, seq)) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    int gave_up = 0;
    if (int rc = fused_wait(c, seq, s, &gave_up)) return rc;
    const auto t2 = std::chrono::steady_clock::now();
    for (int k = 0; k < 7; ++k)
      if (host[k])
        for (size_t r = 0; r < nrows[k]; ++r)
          std::memcpy(static_cast<unsigned char*>(host[k]) + r * ld * elem[k], static_cast<unsigned char*>(pd[k]) + r * dld * elem[k], n * elem[k]);
    if (trace) {
      const auto t3 = std::chrono::steady_clock::now();
      auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
      std::fprintf(stderr, "mvk: avoidance rows n=%zu: launch %.1f us, wait %.1f us, copy out %.1f us\n", n, us(t0, t1), us(t1, t2), us(t2, t3));
    }
    return MVK_OK;
  }
  if (int rc = c->in_a.reserve(dld * rows * sizeof(double))) return rc;
  // up to 8192 states: rows and outputs staged through pinned blocks, one contiguous copy each (as eval_host)
  const bool staged = n <= 8192;
  if (staged) {
    if (int rc = c->pin_up.reserve(dld * rows * sizeof(double))) return rc;
    double* up = static_cast<double*>(c->pin_up.p);
    for (int r = 0; r < rows; ++r) std::memcpy(up + static_cast<size_t>(r) * dld, soa + static_cast<size_t>(r) * ld, n * sizeof(double));
    HIP_TRY(hipMemcpyAsync(c->in_a.p, up, dld * rows * sizeof(double), hipMemcpyHostToDevice, s));
  } else if (int rc = copy_rows(c->in_a.p, dld, soa, ld, sizeof(double), n, rows, hipMemcpyHostToDevice, s)) {
    return rc;
  }
  void* dev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int k = 0; k < 7; ++k)
    if (host[k]) {
      if (int rc = c->av[k].reserve(dld * nrows[k] * elem[k])) return rc;
      dev[k] = c->av[k].p;
    }
  mvk_avoid_out d{static_cast<double*>(dev[0]), static_cast<int32_t*>(dev[1]), static_cast<double*>(dev[2]),
                  static_cast<double*>(dev[3]), static_cast<double*>(dev[4]), static_cast<double*>(dev[5]),
                  static_cast<uint8_t*>(dev[6])};
  if (int rc = mvk_collision_avoidance_batch_dev(c, static_cast<const double*>(c->in_a.p), n, dld, layout, prm, &d, s)) return rc;
  if (staged) {
    size_t off[7], total = 0;
    for (int k = 0; k < 7; ++k) {
      off[k] = total;
      if (host[k]) total += round_up(dld * nrows[k] * elem[k], 256);
    }
    if (int rc = c->pin_down.reserve(total ? total : 256)) return rc;
    unsigned char* down = static_cast<unsigned char*>(c->pin_down.p);
    for (int k = 0; k < 7; ++k)
      if (host[k]) HIP_TRY(hipMemcpyAsync(down + off[k], dev[k], dld * nrows[k] * elem[k], hipMemcpyDeviceToHost, s));
    if (int rc_ = sync_stream(c, s)) return rc_;
    for (int k = 0; k < 7; ++k)
      if (host[k])
        for (size_t r = 0; r < nrows[k]; ++r)
          std::memcpy(static_cast<unsigned char*>(host[k]) + r * ld * elem[k], down + off[k] + r * dld * elem[k], n * elem[k]);
    return MVK_OK;
  }
  for (int k = 0; k < 7; ++k)
    if (host[k])
      if (int rc = copy_rows(host[k], ld, dev[k], dld, elem[k], n, nrows[k], hipMemcpyDeviceToHost, s)) return rc;
  if (int rc_ = sync_stream(c, s)) return rc_;
  return MVK_OK;
}

// ---- motion validation -----------------------------------------------------------
namespace {
// d_edge_off: exclusive prefix of clamp(nd,1) (device), total = number of steps
int run_edges_dev(mvk_ctx* c, const double* d_s1, const double* d_s2, size_t n_edges, size_t ld,
                  const int32_t* d_nd, const int64_t* d_edge_off, uint64_t total, int32_t nd_uniform,
                  uint8_t* d_ok, int32_t* d_first, hipStream_t s, bool first_is_initialised = false,
                  const uint32_t* d_edge_block = nullptr, int32_t* fused_host_first = nullptr) {
  EvalArgs a;
  base_args(c, a);
  a.edge_block = d_edge_block;
  a.s1 = d_s1;
  a.s2 = d_s2;
  a.nd = d_nd;
  a.edge_off = d_edge_off;
  a.nd_uniform = nd_uniform < 1 ? 1 : nd_uniform;
  a.n = total;
  a.n_edges = n_edges;
  a.ld = ld;
  a.layout = MVK_LAYOUT_OMPL;
  a.first_invalid = d_first;
  c->stats_valid = false;
  if (int rc = prepare_queue(c, a, total)) return rc;
  if (int rc = enter_stream(c, s)) return rc;
  if (fused_host_first) {  // one launch; the arrays are pinned host memory the kernel reads and writes directly
    if (int rc = fused_arm(c, a, s, n_edges)) return rc;
    if (c->split_enabled) a.fused = 2;  // fused_split_kernel: walk 1 and the collision screen of a step in two waves
    a.host_first = fused_host_first;
    a.first_invalid = nullptr;
    const int ef = launch_eval(a, c->model.robot_type, c->model.n_rotor, true, s, c->profiling ? c->ev : nullptr);
    c->timed = c->profiling;
    if (ef != hipSuccess) return fail(MVK_ERR_HIP, "edge kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(ef)));
    return leave_stream(c, s);
  }
  int e = first_is_initialised ? static_cast<int>(hipSuccess) : launch_edge_init(d_first, d_nd, a.nd_uniform, n_edges, s);
  if (e == hipSuccess) e = launch_eval(a, c->model.robot_type, c->model.n_rotor, true, s, c->profiling ? c->ev : nullptr);
  c->timed = c->profiling;
  if (e == hipSuccess && d_ok) e = launch_edge_finish(d_first, d_nd, a.nd_uniform, d_ok, n_edges, s);
  if (e != hipSuccess) return fail(MVK_ERR_HIP, "edge kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(e)));
  return leave_stream(c, s);
}

int prefix_counts(const int32_t* nd, size_t n, std::vector<int64_t>& off, uint64_t& total) {
  off.resize(n);
  total = 0;
  for (size_t e = 0; e < n; ++e) {
    off[e] = static_cast<int64_t>(total);
    total += static_cast<uint64_t>(nd[e] < 1 ? 1 : nd[e]);
  }
  return MVK_OK;
}
}  // namespace

extern "C" int mvk_check_motion_batch_dev(mvk_ctx* c, const double* d_s1, const double* d_s2, size_t n_edges,
                                          size_t ld, const int32_t* d_nd, int32_t nd_uniform, uint8_t* d_ok,
                                          int32_t* d_first_invalid, void* stream) {
  if (c && c->edges_pending) return fail(MVK_ERR_INVALID_ARG, "an edge check begun with mvk_check_motion_batch_begin is still in flight on this ctx: call mvk_check_motion_batch_end first");
  if (!c) return fail(MVK_ERR_INVALID_ARG, "null ctx");
  if (n_edges == 0) return MVK_OK;
  if (!d_s1 || !d_s2 || ld < n_edges) return fail(MVK_ERR_INVALID_ARG, "bad edge arrays (null, or ld < n_edges)");
  if (!d_ok && !d_first_invalid) return fail(MVK_ERR_INVALID_ARG, "no output requested");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  int32_t* d_first = d_first_invalid;
  if (!d_first) {
    if (int rc = c->first.reserve(n_edges * sizeof(int32_t))) return rc;
    d_first = static_cast<int32_t*>(c->first.p);
  }
  uint64_t total = 0;
  const int64_t* d_off = nullptr;
  if (d_nd) {
    // per-edge counts: the (edge, step) -> lane map needs their prefix sums
    std::vector<int32_t> h(n_edges);
    HIP_TRY(hipMemcpyAsync(h.data(), d_nd, n_edges * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (int rc_ = sync_stream(c, s)) return rc_;
    std::vector<int64_t> off;
    prefix_counts(h.data(), n_edges, off, total);
    if (int rc = c->edge_off.reserve(n_edges * sizeof(int64_t))) return rc;
    HIP_TRY(hipMemcpyAsync(c->edge_off.p, off.data(), n_edges * sizeof(int64_t), hipMemcpyHostToDevice, s));
    if (int rc_ = sync_stream(c, s)) return rc_;  // `off` is a local
    d_off = static_cast<const int64_t*>(c->edge_off.p);
  } else {
    total = static_cast<uint64_t>(n_edges) * static_cast<uint64_t>(nd_uniform < 1 ? 1 : nd_uniform);
  }
  return run_edges_dev(c, d_s1, d_s2, n_edges, ld, d_nd, d_off, total, nd_uniform, d_ok, d_first, s);
}

extern "C" int mvk_check_motion_batch_begin(mvk_ctx* c, const double* s1, const double* s2, size_t n_edges, size_t ld,
                                            const int32_t* nd, int32_t nd_uniform) {
  if (!c) return fail(MVK_ERR_INVALID_ARG, "null ctx");
  if (c->edges_pending) return fail(MVK_ERR_INVALID_ARG, "mvk_check_motion_batch_begin: the last begin has not been ended");
  c->edges_in_flight = 0;
  c->edges_pending = true;
  if (n_edges == 0) return MVK_OK;
  c->edges_pending = false;
  if (!s1 || !s2 || ld < n_edges) return fail(MVK_ERR_INVALID_ARG, "bad edge arrays (null, or ld < n_edges)");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  // This is the planner's inner loop (hundreds of small calls per solve), so the call is shaped for
  // latency: everything the kernels read -- both state blocks, the per-edge counts, their prefix and
  // the initial first_invalid = nd + 1 -- is packed into ONE pinned blob and uploaded with one copy;
  // the verdict comes back with one copy (ok is derived from first_invalid on the host).
  const size_t rows = static_cast<size_t>(space_dim(&c->params, c->model.n_joint));
  const size_t dld = round_up(n_edges, 64);
  c->edges_n = n_edges;
  size_t steps_max = 0;  // upper bound of the lane count, for the size of the per-block hints
  if (nd) for (size_t e = 0; e < n_edges; ++e) steps_max += static_cast<size_t>(nd[e] < 1 ? 1 : nd[e]);
  const size_t n_blk = nd ? (steps_max + kBlock - 1) / kBlock : 0;
  const size_t off_s2 = rows * dld * sizeof(double), off_off = 2 * off_s2, off_nd = off_off + dld * sizeof(int64_t),
               off_first = off_nd + dld * sizeof(int32_t), off_blk = off_first + dld * sizeof(int32_t),
               bytes = off_blk + round_up(n_blk, 64) * sizeof(uint32_t);
  if (int rc = c->pin_up.reserve(bytes)) return rc;
  if (int rc = c->pin_down.reserve(dld * sizeof(int32_t))) return rc;
  if (int rc = c->edge_blob.reserve(bytes)) return rc;
  char* up = static_cast<char*>(c->pin_up.p);
  double* h1 = reinterpret_cast<double*>(up);
  double* h2 = reinterpret_cast<double*>(up + off_s2);
  int64_t* hoff = reinterpret_cast<int64_t*>(up + off_off);
  int32_t* hnd = reinterpret_cast<int32_t*>(up + off_nd);
  int32_t* hfirst = reinterpret_cast<int32_t*>(up + off_first);
  for (size_t r = 0; r < rows; ++r) {
    std::memcpy(h1 + r * dld, s1 + r * ld, n_edges * sizeof(double));
    std::memcpy(h2 + r * dld, s2 + r * ld, n_edges * sizeof(double));
  }
  uint64_t total = 0;
  const int32_t nu = nd_uniform < 1 ? 1 : nd_uniform;
  for (size_t e = 0; e < n_edges; ++e) {
    const int32_t v = nd ? (nd[e] < 1 ? 1 : nd[e]) : nu;
    hoff[e] = static_cast<int64_t>(total);
    hnd[e] = v;
    hfirst[e] = v + 1;
    total += static_cast<uint64_t>(v);
  }
  if (nd) {  // the edge holding the first lane of every block of kBlock lanes (saves the kernels a binary search)
    uint32_t* hblk = reinterpret_cast<uint32_t*>(up + off_blk);
    size_t e = 0;
    for (size_t b = 0; b < n_blk; ++b) {
      const int64_t lane = static_cast<int64_t>(b * kBlock);
      while (e + 1 < n_edges && hoff[e + 1] <= lane) ++e;
      hblk[b] = static_cast<uint32_t>(e);
    }
  }
  int32_t nd_max = nu;
  if (nd) for (size_t e = 0; e < n_edges; ++e) nd_max = std::max(nd_max, hnd[e]);
  c->fused_in_flight = false;
  c->edges_blob_bytes = bytes;
  c->edges_total = total;
  c->edges_have_nd = nd != nullptr;
  c->edges_nu = nu;
  c->edges_dld = dld;
  if (fused_eligible(c, total) && nd_max < 0xFFFF && total <= c->flag_small_max && c->small_full_enabled) {
    // an edge or a few (OMPL's checkMotion): a block of two waves per step (the blob goes to device memory first:
    // lane -> (edge, step) -> rows is a chain of dependent reads, 2 us each over PCIe)
    EvalArgs a;
    base_args(c, a);
    char* dev = static_cast<char*>(c->edge_blob.p);
    a.edge_block = nd ? reinterpret_cast<const uint32_t*>(dev + off_blk) : nullptr;
    a.s1 = reinterpret_cast<const double*>(dev);
    a.s2 = reinterpret_cast<const double*>(dev + off_s2);
    a.nd = nd ? reinterpret_cast<const int32_t*>(dev + off_nd) : nullptr;
    a.edge_off = nd ? reinterpret_cast<const int64_t*>(dev + off_off) : nullptr;
    a.nd_uniform = nu;
    a.n = total;
    a.n_edges = n_edges;
    a.ld = dld;
    a.layout = MVK_LAYOUT_OMPL;
    c->stats_valid = false;
    if (int rc = enter_stream(c, s)) return rc;
    if (n_edges == 1) {
      // one edge (OMPL's checkMotion): its end states travel in the kernel arguments, nothing is copied up
      a.inline_on = 1;
      for (size_t r = 0; r < rows; ++r) { a.inline_state[r] = s1[r * ld]; a.inline_state2[r] = s2[r * ld]; }
      a.s1 = a.s2 = nullptr;
      a.nd = nullptr;
      a.edge_off = nullptr;
      a.edge_block = nullptr;
      a.nd_uniform = hnd[0];
    } else {
      HIP_TRY(hipMemcpyAsync(dev, up, bytes, hipMemcpyHostToDevice, s));
    }
    a.fused_seq = next_seq(c);
    a.host_done = static_cast<volatile uint32_t*>(c->pin_ctl.p);
    {
      const size_t before = c->first_tag.cap;
      if (int rc = c->first_tag.reserve(n_edges * sizeof(uint32_t))) return rc;
      if (c->tag_clear_needed || c->first_tag.cap != before) {
        HIP_TRY(hipMemsetAsync(c->first_tag.p, 0, c->first_tag.cap, s));
        c->tag_clear_needed = false;
      }
      a.first_tag = static_cast<uint32_t*>(c->first_tag.p);
    }
    a.host_first = static_cast<int32_t*>(c->pin_down.p);
    a.fused_done = static_cast<uint32_t*>(c->small_done.p);
    a.small_target = c->small_total + static_cast<uint32_t>(total);
    if (c->profiling) (void)hipEventRecord(c->ev[0], s);
    const int e = launch_flag_small(a, c->model.robot_type, c->model.n_rotor, true, s);
    if (e != hipSuccess) return fail(MVK_ERR_HIP, "edge kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(e)));
    c->small_total = a.small_target;
    if (c->profiling) { (void)hipEventRecord(c->ev[1], s); (void)hipEventRecord(c->ev[2], s); }
    c->timed = c->profiling;
    if (int rc = leave_stream(c, s)) return rc;
    c->fused_in_flight = true;
    c->edges_in_flight = n_edges;
    c->edges_nd_off = off_nd;
    c->edges_pending = true;
    return MVK_OK;
  }
  if (fused_eligible(c, total) && nd_max < 0xFFFF) {
    // one copy up (every lane of an edge reads the edge's rows: they belong in device memory, where the cache holds
    // them), one launch; the kernel leaves first_invalid in the pinned answer block and raises the done word
    char* dev = static_cast<char*>(c->edge_blob.p);
    if (int rc = enter_stream(c, s)) return rc;
    HIP_TRY(hipMemcpyAsync(dev, up, bytes, hipMemcpyHostToDevice, s));
    if (int rc = run_edges_dev(c, reinterpret_cast<const double*>(dev), reinterpret_cast<const double*>(dev + off_s2), n_edges, dld,
                               nd ? reinterpret_cast<const int32_t*>(dev + off_nd) : nullptr,
                               nd ? reinterpret_cast<const int64_t*>(dev + off_off) : nullptr, total, nu, nullptr, nullptr, s, true,
                               nd ? reinterpret_cast<const uint32_t*>(dev + off_blk) : nullptr, static_cast<int32_t*>(c->pin_down.p)))
      return rc;
    c->fused_in_flight = true;
    c->edges_in_flight = n_edges;
    c->edges_nd_off = off_nd;
    c->edges_pending = true;
    return MVK_OK;
  }
  // larger batches: two launches (the exact stage then has the whole chip, not only this call's blocks), one copy
  // down, one synchronisation (a third kernel handing the verdicts to a polling host was 3-6 us slower: profiles/r02)
  return edges_submit_two_launches(c);
}

// the two-launch form of an edge check whose pinned block is ready (mvk_check_motion_batch_begin; and _end when
// the fused kernel's grid barrier gave up)
int mvk::edges_submit_two_launches(mvk_ctx* c) {
  hipStream_t s = c->stream;
  const size_t n_edges = c->edges_n, dld = c->edges_dld, bytes = c->edges_blob_bytes;
  const size_t rows = static_cast<size_t>(space_dim(&c->params, c->model.n_joint));
  const size_t off_s2 = rows * dld * sizeof(double), off_off = 2 * off_s2, off_nd = off_off + dld * sizeof(int64_t),
               off_first = off_nd + dld * sizeof(int32_t), off_blk = off_first + dld * sizeof(int32_t);
  const bool nd = c->edges_have_nd;
  const uint64_t total = c->edges_total;
  const int32_t nu = c->edges_nu;
  char* up = static_cast<char*>(c->pin_up.p);
  char* dev = static_cast<char*>(c->edge_blob.p);
  if (int rc = enter_stream(c, s)) return rc;
  HIP_TRY(hipMemcpyAsync(dev, up, bytes, hipMemcpyHostToDevice, s));
  if (int rc = run_edges_dev(c, reinterpret_cast<const double*>(dev), reinterpret_cast<const double*>(dev + off_s2), n_edges, dld,
                             nd ? reinterpret_cast<const int32_t*>(dev + off_nd) : nullptr,
                             nd ? reinterpret_cast<const int64_t*>(dev + off_off) : nullptr, total, nu, nullptr,
                             reinterpret_cast<int32_t*>(dev + off_first), s, /*first_is_initialised=*/true,
                             nd ? reinterpret_cast<const uint32_t*>(dev + off_blk) : nullptr))
    return rc;
  HIP_TRY(hipMemcpyAsync(c->pin_down.p, dev + off_first, n_edges * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  c->edges_in_flight = n_edges;
  c->edges_nd_off = off_nd;
  c->edges_pending = true;
  return MVK_OK;
}

extern "C" int mvk_check_motion_batch_end(mvk_ctx* c, uint8_t* ok, int32_t* first_invalid) {
  if (!c) return fail(MVK_ERR_INVALID_ARG, "null ctx");
  if (!c->edges_pending) return fail(MVK_ERR_INVALID_ARG, "mvk_check_motion_batch_end without a begin");
  c->edges_pending = false;
  const size_t n_edges = c->edges_in_flight;
  if (n_edges == 0) return MVK_OK;
  // wait first, whatever the arguments: once the ctx reads as idle nothing may still be using its pinned blocks
  HIP_TRY(hipSetDevice(c->device));
  if (c->fused_in_flight) {
    c->fused_in_flight = false;
    int gave_up = 0;
    if (int rc = fused_wait(c, c->fused_seq, c->stream, &gave_up)) return rc;
    if (gave_up) {  // (never seen) repeat with two launches
      if (int rc = edges_submit_two_launches(c)) return rc;
      c->edges_pending = false;
      if (int rc_ = sync_stream(c, c->stream)) return rc_;
    }
  } else {
    if (int rc_ = sync_stream(c, c->stream)) return rc_;
  }
  if (!ok && !first_invalid) return fail(MVK_ERR_INVALID_ARG, "no output requested");
  const int32_t* down = static_cast<const int32_t*>(c->pin_down.p);
  const int32_t* hnd = reinterpret_cast<const int32_t*>(static_cast<const char*>(c->pin_up.p) + c->edges_nd_off);
  for (size_t e = 0; e < n_edges; ++e) {
    if (first_invalid) first_invalid[e] = down[e];
    if (ok) ok[e] = down[e] == hnd[e] + 1 ? 1 : 0;
  }
  return MVK_OK;
}

extern "C" int mvk_check_motion_batch(mvk_ctx* c, const double* s1, const double* s2, size_t n_edges, size_t ld,
                                      const int32_t* nd, int32_t nd_uniform, uint8_t* ok, int32_t* first_invalid) {
  if (!c) return fail(MVK_ERR_INVALID_ARG, "null ctx");
  if (n_edges == 0) return MVK_OK;
  if (!ok && !first_invalid) return fail(MVK_ERR_INVALID_ARG, "no output requested");
  if (int rc = mvk_check_motion_batch_begin(c, s1, s2, n_edges, ld, nd, nd_uniform)) return rc;
  return mvk_check_motion_batch_end(c, ok, first_invalid);
}
