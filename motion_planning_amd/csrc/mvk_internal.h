// mvk_internal.h -- declarations shared by the translation units of libmvk.so.
// Not part of the public ABI (that is include/mvk.h).
#ifndef MVK_INTERNAL_H_
#define MVK_INTERNAL_H_

#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/mvk.h"

namespace mvk {

// record an error text for mvk_last_error() and return `code`
int fail(int code, const char* fmt, ...);

// ---- device-side views -------------------------------------------------
// World box in the form the kernels read: centre, rotation (row-major, columns
// are the box axes in world coordinates) and half extents.  16 doubles = 128 B.
struct DevObs {
  double c[3];
  double R[9];
  double h[3];
  double bound;  // circumscribed-sphere radius
};

// fp32 copy for the level-1 cull: 16 floats = 64 B (four 16-byte loads per lane).
struct alignas(64) DevObsF {
  float c[3];
  float R[9];
  float h[3];
  float pad;
};

// Model constants derived once on the host.
constexpr int kSegPrimMax = 3;           // collision primitives riding on one segment (checked by mvk_create)

// Everything walk 2 of the screen stage reads for one chain segment, in one place: one batch of
// independent scalar loads per segment instead of chains through prim_order[] and prim_rotor[].
struct SegTab {
  double t[3];                          // origin in the parent segment frame (mvk_model::seg_t)
  double sph[3];                        // centre of the segment's bounding sphere
  double prim_pos[kSegPrimMax][3];      // collision origins, in their frame
  double gimbal_pos[kSegPrimMax][3];    // gimbal centre carrying the primitive (frame != SEG)
  int32_t axis, joint, np, pad;         // pad: 1 when a primitive of the segment rides on a gimbal frame
  int32_t prim_frame[kSegPrimMax + 1];  // MVK_FRAME_*
};

// fp32 copy of the model constants for the fp32 variant of the flag path (walk 1 and the frames of walk 2 in
// fp32): wave-uniform scalar loads of floats, no conversions in the kernel.  The err_* members are the bounds of
// the fp32 evaluation errors the stability classification works with (derive_extras).
struct ModelF32 {
  float seg_t[MVK_MAX_SEG][3];
  float seg_mcom[MVK_MAX_SEG][3];  // mass * centre of mass
  float seg_mass[MVK_MAX_SEG];
  float rotor_pos[MVK_MAX_ROTOR][3];
  float rotor_ks[MVK_MAX_ROTOR];   // m_f_rate * sigma
  float inv_mass, weight;          // 1 / total mass, total mass * gravity
  float thrust_min, thrust_max, fc_t_thre, fc_rp_thre;
  float edf_offset_y, edf_two_r, tan_edf_max_tilt, rotor_roll_z;
  float err_p;   // position of a frame origin relative to the root, incl. the joints' cos/sin errors [m]
  float err_u;   // a component of a rotation matrix / thrust axis
  float pad_[1];
};

struct DevExtras {
  double tan_edf_max_tilt;
  double prim_bound[MVK_MAX_PRIM];  // radius about the primitive origin containing it
  double prim_inner[MVK_MAX_PRIM];  // radius about the origin contained in it
  int32_t seg_prim_begin[MVK_MAX_SEG + 1];  // primitives sorted by segment
  // groups of co-located primitives of a segment (consecutive in chain order): the distance kernel runs ONE broadphase
  // per group about its leader's centre with radius grp_rb (contains every member) and every member's narrowphase over
  // that list.  n_grp == 0: one primitive at a time (MVK_NO_PRIM_GROUPS=1, diagnostics / identity tests)
  int32_t n_grp;
  int32_t grp_begin[MVK_MAX_PRIM + 1];
  int32_t grp_lead[MVK_MAX_PRIM];  // index into prim_order of the member the broadphase is centred on
  int32_t seg_grp_begin[MVK_MAX_SEG + 1];
  double grp_rb[MVK_MAX_PRIM];
  double grp_rq[MVK_MAX_PRIM];   // avoidance rows: the cell walk's radius term -- max over members of (bounding radius + twice the distance to the leader's centre)
  double grp_rb_max;  // >= every prim_bound: the radius the distance grids and the far grid are built for
  int32_t prim_order[MVK_MAX_PRIM];
  double seg_sph[MVK_MAX_SEG][4];  // per segment: sphere (centre in the segment frame, radius) holding its primitives
  double reach;                    // bound on the distance of any primitive point from the root origin
  double inv_mass;                 // 1 / total mass
  SegTab seg_tab[MVK_MAX_SEG];
  ModelF32 f32;
  float seg_sph_f[MVK_MAX_SEG][4];             // seg_sph in fp32
  float seg_prim_pos_f[MVK_MAX_SEG][kSegPrimMax][3];    // SegTab::prim_pos in fp32
  float seg_gimbal_pos_f[MVK_MAX_SEG][kSegPrimMax][3];  // SegTab::gimbal_pos in fp32
};

// ---- broadphase grid (mvk_grid.cpp) ---------------------------------------
constexpr int kGridGroup = 4;             // candidate ids fetched per 8-byte load
constexpr int kGridNone = 0xFFFF;         // padding id
constexpr uint64_t kGridMaxCells = 1u << 17;
constexpr uint64_t kGridMaxItems = 1u << 23;

struct GridHost {
  double origin[3], h, radius;
  int32_t n[3];
  std::vector<uint32_t> start;  // [ncell + 1], in groups of kGridGroup ids
  std::vector<uint16_t> items;  // [start[ncell] * kGridGroup]
  // by_bound grids: per group, a lower bound of the distance from ANY point of the cell to the group's boxes and to
  // every box after them in the cell's list (lists sorted by that bound, ascending): a walk that looks for boxes within
  // some distance of a point stops at the first group whose bound exceeds it.  [start[ncell]]; empty otherwise
  std::vector<float> glb;
};
// every box within `radius` of a point is listed in the cell holding the point.  by_bound: lists sorted by the
// box-to-cell lower bound instead of by id, with GridHost::glb (the bounds less `pad`, the slack for the device's fp32
// distances).  cell_frac: cell side as a fraction of the radius, where the cell budget allows
int build_grid(const DevObs* obs, size_t n, double radius, GridHost& g, bool by_bound = false, double pad = 0.0, double cell_frac = 0.5);
// far grid: the cell holding a point x lists every box b with l(x, b) <= reach + min over boxes of l(x, .), inside the
// boxes' bound inflated by `margin` (pad: slack for the device's fp32 centre distances)
int build_far_grid(const DevObs* obs, size_t n, double reach, double margin, double pad, GridHost& g);
// host lookup (tests / diagnostics): returns the padded candidate count, ids in *ids
int grid_query(const GridHost& g, const double c[3], const uint16_t** ids);

struct GridDev {
  double origin[3], inv_h;
  int32_t n[3];
  int32_t pad;
  const uint32_t* start;
  const uint16_t* items;
  const float* glb;  // per group lower bounds of a by_bound grid (GridHost::glb), or nullptr
};

struct EvalArgs {
  // state source: either a SoA batch ...
  const double* soa;
  // ... or edges (s1, s2 in MVK_LAYOUT_OMPL) densified in-kernel
  const double* s1;
  const double* s2;
  const int32_t* nd;         // per-edge segment counts or nullptr
  const int64_t* edge_off;   // exclusive prefix of nd (when nd != nullptr)
  const uint32_t* edge_block;  // optional hint: the edge holding lane b * kBlock, per block b of lanes (else binary search from 0)
  int32_t nd_uniform;
  uint64_t n;   // samples (or total edge steps)
  uint64_t n_edges;
  uint64_t ld;
  int32_t layout;
  int32_t planning_mode, motion_type;
  int32_t do_stab, do_coll;
  int32_t full;  // 1: exact min distance / margin / band for every sample, no early exit
  double eps_dist, eps_margin;
  const mvk_model* model;
  const DevExtras* extras;
  const DevObs* obs;
  const DevObsF* obsf;   // n_obs + 1 entries: the last one is the padding id's never-hit box
  GridDev grid;
  GridDev dgrid;        // full path: distance grid keyed by primitive centres (start == nullptr: scan all boxes)
  double dgrid_radius;  // every box within this distance of a point is listed in the point's dgrid cell
  int32_t n_obs;
  float f32_delta;   // bound on the fp32 centre-distance error inside f32_range [m]
  double f32_range;  // coordinate magnitude up to which f32_delta holds
  // fp32 variant of the flag path: f32_delta plus the bound of the fp32 frame walk's position error inside
  // f32_range, and a scale on the stability error bounds (1; tests shrink it to show the safety margin)
  float f32_delta_fk;
  float f32_bound_scale;
  int32_t f32_walk;        // 1: screen stage in fp32, uncertain states refined in fp64
  uint32_t* refine_list;   // [n] state indices the fp32 screen hands to the fp64 screen
  uint32_t* refine_count;  // one counter in the call's counter set
  uint32_t refine_blocks;  // grid of the refine launch
  // small batches, one launch (fused_kernel): grid barrier and completion counters in the call's counter set, the
  // verdicts go straight to pinned host memory and a done word there tells the host (no copies, no stream sync)
  int32_t fused;           // 1: launch fused_kernel
  uint32_t fused_seq;      // this call's sequence number (16 bits used by first_tag)
  uint32_t* fused_bar;     // arrivals at the grid barrier
  uint32_t* fused_done;    // blocks that have finished
  uint32_t fused_spin_limit;  // polls of the grid barrier before a block gives up (diagnostics: MVK_FUSED_SPIN_LIMIT)
  uint32_t* first_tag;     // edge mode: [n_edges] (seq << 16 | 0xFFFF - step), atomicMax: no initialisation needed
  int32_t* host_first;     // edge mode: pinned host [n_edges] first_invalid, written by the last block
  uint8_t* host_flags;     // state mode: pinned host copy of flags[n], written by the last block
  volatile uint32_t* host_done;  // pinned host: seq when every verdict is in place, seq | 0x80000000: barrier timed out
  // outputs (nullable)
  uint8_t* flags;
  uint64_t* mask;
  double* min_dist;
  double* margin;
  uint8_t* band;
  // fp32 variant of the distance / margin outputs (mvk_eval_batch_f32_dev): model32_kernel + distance_kernel<.., true>
  int f32_variant = 0;
  float* min_dist_f32 = nullptr;    // [n]
  float* margin_f32 = nullptr;      // [n] midpoint of the fp32 enclosure of the margin
  float* margin_err_f32 = nullptr;  // [n] its half width (+inf: a test could not be bounded in fp32)
  double* fc_t_min;
  double* fc_rp_min;
  double* thrust;
  double* rp_margin;
  double* cog;
  double* gimbal;
  int32_t* first_invalid;  // edge mode: atomicMin target
  // flag path: states the fp32 screen leaves undecided, for the exact stage (ctx-owned)
  // flag path hand-off from the screen stage to the exact stage: queues of state indices, one per
  // (chain segment, shard); a screen wave appends to shard (wave index % kQueueShards) of a segment
  uint32_t* queue;             // [n_seg * kQueueShards][queue_stride]
  uint64_t queue_stride;       // capacity of one queue: every lane of every kQueueShards-th wave
  uint32_t* queue_count;       // [n_seg * kQueueShards] counters kQueueCounterPitch words apart, appended to by this call (zero on entry)
  uint32_t* queue_count_next;  // same shape, zeroed by this call for the next one
  double* queue_rec;           // [lanes launched][8 + 2 * n_joint] per state: root pose, pad, (cos, sin) of every joint
  uint32_t exact_blocks;       // grid of the exact stage
  // per-block partial statistics (full mode) or nullptr
  void* stats_partial;
  // full path hand-off from model_kernel to distance_kernel (ctx-owned; nullptr: no collision stage, the model
  // kernel finishes the states itself): rows of full_rec_ld doubles -- cos/sin of every joint, then five scalars
  double* full_rec;
  uint64_t full_rec_ld;
  // full path, small batches (distance_small_kernel): kSmallGroup lanes per state; the second, wider distance grid
  // serves the primitives the first one cannot answer before the sweep over all boxes does
  int32_t small_full;     // 1: model_kernel + distance_small_kernel; 2: full_small_kernel (one launch, two states per block)
  uint32_t small_target;  // host_done set: the block that brings *fused_done to this count raises the done word
  GridDev dgrid2;
  double dgrid2_radius;
  // a host call of ONE state carries its rows in the kernel arguments (flag_small_kernel, full_small_kernel): the
  // kernel then starts without the dependent read of the pinned staging block over PCIe (one edge: without the copy up)
  int32_t inline_on;
  double inline_state[MVK_MAX_STATE_DIM];
  double inline_state2[MVK_MAX_STATE_DIM];  // a host call of ONE edge: its end states in inline_state / inline_state2
  GridDev fgrid;          // far grid (start == nullptr: none): complete nearest-candidate lists at any distance, inside its volume
  // C-space sweep (mvk_sweep_mask_dev): states are generated from the sample index, nothing is read
  int32_t sweep_on;
  uint64_t sweep_first, sweep_seed;
  double sweep_pose[7];
  double sweep_lo[MVK_MAX_JOINT], sweep_span[MVK_MAX_JOINT];
  // collision-avoidance rows (avoid_kernel): parameters and nullable outputs, row pitch ld
  double av_gain, av_range, av_forbid;
  int32_t av_full_body;
  double* av_dist;     // [P][ld]
  int32_t* av_box;     // [P][ld]
  double* av_p_env;    // [3P][ld]
  double* av_p_robot;  // [3P][ld]
  double* av_row;      // [(6+J)P][ld]
  double* av_lb;       // [P][ld]
  uint8_t* av_ok;      // [ld]
};

struct StatsPartial {
  unsigned long long valid_count, band_count;
  double min_dist;   long long min_dist_index;
  double min_margin; long long min_margin_index;
  double max_thrust; long long max_thrust_index;
  double min_thrust; long long min_thrust_index;
  double min_rp;     long long min_rp_index;
};

constexpr int kBlock = 256;
constexpr int kQueueShards = 8;         // queues per segment (MVK_MAX_SEG * kQueueShards counters per set)
constexpr int kQueueCounterPitch = 64;  // uint32 words between two counters: one 256-byte line each
constexpr int kSmallGroup = 64;          // distance_small_kernel: lanes per state (>= MVK_MAX_PRIM)
constexpr uint64_t kSmallFullMax = 4096;  // states per call up to which the full path takes distance_small_kernel (profiles/small_full_time.py)
constexpr uint64_t kFullOneLaunchMax = 256;  // ... and up to which walk 1 joins it in the same launch (full_small_kernel)
constexpr uint64_t kSmallAvoidMax = 4096; // ... and the avoidance rows avoid_small_kernel
constexpr uint64_t kFlagSmallMax = 256;   // flag path: states per host call up to which a state gets a block of its own (flag_small_kernel)
constexpr int kFusedHardMax = 96;       // most blocks a fused launch may have (MVK_FUSED_MAX_BLOCKS can raise the default up to here)
constexpr int kFusedMaxBlocks = 28;     // fused kernels: blocks of kBlock lanes (every block resident at once); beyond ~7 k lanes two launches win (profiles/latency.py)


// ---- the no-allocation flag path (mvk_is_valid_mask_noalloc_dev) -------------------------------------------
// Its hand-off buffers live in ONE file-scope __device__ array of mvk_kernels.hip instead of ctx-owned hipMalloc
// memory: [counter set][counter set][queues][records], sized for MVK_NOALLOC_MAX_STATES states of the largest
// model the blob can describe (MVK_MAX_SEG segments, MVK_MAX_JOINT joints).  Every GPU holds its own copy.
constexpr size_t kStaticCounterSetBytes = static_cast<size_t>(MVK_MAX_SEG) * kQueueShards * kQueueCounterPitch * sizeof(uint32_t);
constexpr size_t kStaticQueueBytes = static_cast<size_t>(MVK_MAX_SEG) * MVK_NOALLOC_MAX_STATES * sizeof(uint32_t);
constexpr size_t kStaticRecordBytes = static_cast<size_t>(MVK_NOALLOC_MAX_STATES) * (8 + 2 * MVK_MAX_JOINT) * sizeof(double);
constexpr size_t kStaticScratchBytes = 2 * kStaticCounterSetBytes + kStaticQueueBytes + kStaticRecordBytes;  // 22.1 MiB
// address of the calling thread's current device's copy of that array (resolved once per device with
// hipGetSymbolAddress, mvk_kernels.hip); returns hipError_t as int
int static_scratch_for_current_device(int device, void** ptr);

// launchers implemented in mvk_kernels.hip; return hipError_t as int
// events: nullptr, or hipEvent_t[3] recorded before / between / after the kernels of the call
int launch_eval(const EvalArgs& a, int robot_type, int n_links, bool edge_mode, void* stream, void* events);
int launch_avoid(const EvalArgs& a, int robot_type, int n_links, void* stream);
int launch_flag_small(const EvalArgs& a, int robot_type, int n_links, bool edge_mode, void* stream);
int launch_zero_words(uint32_t* p, uint64_t n_words, void* stream);
int launch_edge_init(int32_t* first_invalid, const int32_t* nd, int32_t nd_uniform, uint64_t n_edges, void* stream);
int launch_edge_finish(const int32_t* first_invalid, const int32_t* nd, int32_t nd_uniform, uint8_t* ok,
                       uint64_t n_edges, void* stream);
int launch_stats_final(const StatsPartial* partial, uint64_t n_partial, StatsPartial* out, void* stream);
bool kernel_supported(int robot_type, int n_links);

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): the counter-based
// generator of the C-space sweep.  Shared by host and device code so that both produce the same states.
#if defined(__HIPCC__)
#define MVK_HD __host__ __device__ inline
#else
#define MVK_HD inline
#endif
MVK_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0, p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
    const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0, n1 = static_cast<uint32_t>(p1);
    const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1, n3 = static_cast<uint32_t>(p0);
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// uniform in (0, 1) from 32 random bits, exactly representable steps of 2^-32
MVK_HD double sweep_unit(uint32_t u) { return (static_cast<double>(u) + 0.5) * (1.0 / 4294967296.0); }

// host OMPL arithmetic (mvk_space.cpp)
int space_dim(const mvk_params* p, int n_joint);

}  // namespace mvk

#endif
