/*
 * mvk.h -- C ABI of the MI355X-native batched state / motion validity engine.
 *
 * Drop-in boundary for the validity hot path of the reference planner
 * (tongtybj/motion_planning, package sampling_based_method).  Every entry point
 * cites the reference interface it replaces (paths relative to the reference
 * root).  Plain pointers and sizes only; no C++ / torch types cross this ABI.
 *
 * Conventions
 *   - return 0 = MVK_OK, negative = error (text via mvk_last_error()).
 *   - quaternions are (x, y, z, w), as geometry_msgs::Quaternion.
 *   - batched state arrays are SoA: element (row r, sample i) at soa[r*ld + i]
 *     (ld >= n is the row pitch in elements), so a wavefront reads 64
 *     consecutive doubles per row (coalesced 512 B).
 *   - "host" entry points take host pointers and copy; "_dev" entry points take
 *     device pointers (HBM resident) plus a hipStream_t passed as void*: the work
 *     is enqueued on that stream and the call returns without synchronising.
 *   - calls on one ctx share its scratch buffers, so the library keeps them in
 *     issue order on the device: a _dev call on another stream than the ctx's
 *     previous call first makes its stream wait (hipStreamWaitEvent) for that
 *     call's work.  A ctx is still for one host thread at a time.
 *   - there is NO CPU fallback: every compute entry point fails with
 *     MVK_ERR_NO_DEVICE when no gfx950 device / HIP runtime is usable.
 */
#ifndef MVK_H_
#define MVK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVK_ABI_VERSION 1

/* ---- capacities of the POD model blob --------------------------------- */
#define MVK_MAX_SEG 16   /* chain segments: link bodies + inter-link joint bodies */
#define MVK_MAX_ROTOR 8  /* = max links */
#define MVK_MAX_JOINT 14 /* planned joints J */
#define MVK_MAX_PRIM 32  /* robot collision primitives P */
#define MVK_MAX_STATE_DIM (7 + MVK_MAX_JOINT)

/* ---- error codes -------------------------------------------------------- */
enum {
  MVK_OK = 0,
  MVK_ERR_INVALID_ARG = -1,
  MVK_ERR_NO_DEVICE = -2, /* HIP runtime / gfx950 device not usable */
  MVK_ERR_HIP = -3,       /* a HIP call failed (text in mvk_last_error) */
  MVK_ERR_CAPACITY = -4,  /* model / scene exceeds a compiled-in capacity */
  MVK_ERR_IO = -5,
  MVK_ERR_NO_SOLUTION = -6
};

/* ---- enums mirroring the reference ------------------------------------- */
/* sampling_based_method/msg/PlanningMode.msg:2-4 */
enum { MVK_MODE_ONLY_JOINTS = 0, MVK_MODE_ONLY_BASE = 1, MVK_MODE_JOINTS_AND_BASE = 2 };
/* sampling_based_method/msg/PlanningMode.msg:11-12 */
enum { MVK_SE2 = 0, MVK_SE3 = 1 };
/* squeeze_navigation/include/squeeze_navigation/planner/base_plugin.h:108-109 */
enum { MVK_GAP_HORIZONTAL = 0, MVK_GAP_VERTICAL = 1 };

enum { MVK_ROBOT_HYDRUS = 0, MVK_ROBOT_DRAGON = 1 };
/* robot collision primitive kinds (V-spec primitive set) */
enum { MVK_PRIM_SPHERE = 0, MVK_PRIM_CAPSULE = 1, MVK_PRIM_BOX = 2 };
/* frame a primitive is attached to */
enum { MVK_FRAME_SEG = 0, MVK_FRAME_GIMBAL_ROLL = 1, MVK_FRAME_GIMBAL_PITCH = 2 };
/* batched state layouts */
enum {
  MVK_LAYOUT_POSE7 = 0, /* rows: px py pz qx qy qz qw j1..jJ  (7+J rows) -- the
                           state after the unpack of motion_planning.cpp:81-129 */
  MVK_LAYOUT_OMPL = 1   /* the OMPL state of the (mode, motion type):
                           mode 0: j1..jJ
                           mode 1: SE2 x y yaw | SE3 x y z qx qy qz qw
                           mode 2: (mode-1 rows) then j1..jJ              */
};

/* ---- robot model: POD description of a serial multilink aerial robot ---- *
 * Stands in for aerial_robot_model::RobotModel / HydrusRobotModel /
 * Dragon::HydrusLikeRobotModel (third-party, absent from the reference tree;
 * call sites multilink_state.cpp:107-116, motion_planning.cpp:150).
 * Chain: F_0 = root (= link1 = baselink, motion_planning.cpp:188),
 *        F_k = F_{k-1} * Trans(seg_t[k]) * Rot(seg_axis[k], q[seg_joint[k]]). */
typedef struct mvk_model {
  int32_t abi_version;
  int32_t robot_type; /* MVK_ROBOT_* */
  int32_t n_seg, n_joint, n_rotor, n_prim;
  int32_t wrench_dof; /* 4: z,roll,pitch,yaw (HYDRUS)  3: z,roll,pitch (DRAGON) */
  int32_t has_gimbal; /* rotors sit on 2-DOF (roll,pitch) gimbals solved to be vertical */

  int32_t seg_axis[MVK_MAX_SEG];  /* -1 fixed, 0 x, 1 y, 2 z */
  int32_t seg_joint[MVK_MAX_SEG]; /* planned-joint index or -1 */
  int32_t seg_link[MVK_MAX_SEG];  /* link id (0..N-1) when the segment is a link body, else -1 */
  double seg_t[MVK_MAX_SEG][3];   /* origin in the parent segment frame */
  double seg_mass[MVK_MAX_SEG];
  double seg_com[MVK_MAX_SEG][3]; /* centre of mass in the segment frame */

  int32_t rotor_seg[MVK_MAX_ROTOR];
  double rotor_pos[MVK_MAX_ROTOR][3]; /* gimbal roll-axis origin (rotor origin of a robot without gimbals) in its segment frame */
  double rotor_sigma[MVK_MAX_ROTOR];  /* +-1 spin direction */
  /* gimballed rotors: the thrust-frame origin (what aerial_robot_model calls the rotor origin: statics, the
   * roll/pitch position margin, the EDF pair test) sits rotor_roll_z along the z axis of the gimbal ROLL frame
   * above rotor_pos.  DRAGON: 0.037, read off the reference's logs. */
  double rotor_roll_z;

  double edf_offset_y; /* DRAGON dual EDFs sit at +-edf_offset_y of the gimbal-pitch frame */
  double edf_radius;
  double edf_max_tilt;

  double gravity;
  double m_f_rate; /* rotor drag-moment / thrust ratio (kappa) */
  double thrust_min, thrust_max;
  double fc_t_min_thre;  /* feasible-control torque distance threshold */
  double fc_rp_min_thre; /* feasible-control roll/pitch distance threshold */
  double link_length;    /* informational: rp_position_margin is normalised by |rotor 2 - rotor 1| */

  int32_t prim_type[MVK_MAX_PRIM];  /* MVK_PRIM_* */
  int32_t prim_seg[MVK_MAX_PRIM];   /* segment the primitive rides on */
  int32_t prim_frame[MVK_MAX_PRIM]; /* MVK_FRAME_* */
  int32_t prim_rotor[MVK_MAX_PRIM]; /* rotor whose gimbal carries it (frame != SEG) */
  double prim_pos[MVK_MAX_PRIM][3]; /* collision origin in that frame */
  double prim_rot[MVK_MAX_PRIM][9]; /* collision origin rotation, row-major */
  double prim_dim[MVK_MAX_PRIM][3]; /* SPHERE: r,-,- | CAPSULE: r, half length (axis = local z),- | BOX: half extents */

  double joint_low[MVK_MAX_JOINT], joint_high[MVK_MAX_JOINT];
} mvk_model;

/* World obstacle.  Mirrors shape_msgs::SolidPrimitive BOX + geometry_msgs::Pose
 * as filled by squeeze_navigation/src/planner/sampling_based_method.cpp:66-71:
 * size = full side lengths ("dimensions"), pose = centre. */
typedef struct mvk_box {
  double center[3];
  double quat[4];
  double size[3];
} mvk_box;

/* Planner-side parameters; defaults follow motion_planning.cpp:465-532. */
typedef struct mvk_params {
  int32_t planning_mode; /* MVK_MODE_*  (default 0, motion_planning.cpp:487) */
  int32_t motion_type;   /* MVK_SE2|MVK_SE3 (default 0, :488) */
  /* -1 = reference gating (stability iff mode in {0,2} or SE3, :138-151;
   * collision iff mode != 0, :153); 0 / 1 force the stage off / on. */
  int32_t check_stability;
  int32_t check_collision;
  double eps_dist;   /* boundary band half width on min distance [m] (1e-6) */
  double eps_margin; /* boundary band half width on stability slacks (1e-6) */
  double pos_low[3], pos_high[3]; /* motion_planning.cpp:471-476 */
  double joint_low[MVK_MAX_JOINT], joint_high[MVK_MAX_JOINT]; /* :478-484 */
  double validity_resolution;        /* state_validity_check_res, :527 (0.03) */
  /* :528 (20).  Counts in mode 0 only: in mode 1 the reference sets it on the SE2/SE3
   * compound space (:244), which never reads it (the reference's mode-1 log
   * path_log/hydrus/planning_log.txt has ceil(d / (extent * res)) states per edge). */
  int32_t valid_segment_count_factor;
  /* Quirk Q4 (motion_planning.cpp:252-258): mode 2 never installs resolution /
   * factor, so OMPL defaults (0.01, 1) apply.  1 = reproduce that. */
  int32_t mode2_ompl_defaults;
  double joint_subspace_weight; /* :255 (0.001) */
} mvk_params;

typedef struct mvk_result {
  uint8_t valid;
  uint8_t band; /* within eps of a constraint boundary, or non-finite input */
  uint8_t stable;
  uint8_t collision;
  double min_dist; /* signed separation to the obstacle set (+inf when none / skipped) */
  double margin;   /* min stability slack (>= 0 <=> stabilityCheck() true) */
} mvk_result;

/* Optional per-sample detail outputs (NULL = not wanted).  Rows as documented. */
typedef struct mvk_detail_out {
  double* fc_t_min;           /* [n] */
  double* fc_rp_min;          /* [n] */
  double* thrust;             /* [n_rotor][ld] static thrust lambda (getStaticThrust) */
  double* rp_position_margin; /* [n] legacy margin logged as minimum_var (motion_planning.cpp:615-620) */
  double* cog;                /* [3][ld] world CoG position (multilink_state.cpp:115-117) */
  double* gimbal;             /* [2*n_rotor][ld] solved (roll,pitch) per gimbal; DRAGON only */
} mvk_detail_out;

/* Reduction over the most recent batch of a ctx (motion_planning.cpp:546-642
 * keeps the same running extrema over a path). */
typedef struct mvk_stats {
  uint64_t n, valid_count, band_count;
  double min_dist;
  int64_t min_dist_index;
  double min_margin;
  int64_t min_margin_index;
  double max_thrust; /* max over samples of max_i lambda_i (max_force_, :623-627) */
  int64_t max_thrust_index;
  double min_thrust; /* min over samples of min_i lambda_i */
  int64_t min_thrust_index;
  double min_rp_position_margin; /* min_var_ (:615-620) */
  int64_t min_rp_position_margin_index;
} mvk_stats;

typedef struct mvk_ctx mvk_ctx;

/* ---- library -------------------------------------------------------------- */
int mvk_abi_version(void);
size_t mvk_model_sizeof(void);
size_t mvk_params_sizeof(void);
const char* mvk_last_error(void);
/* number of usable gfx950 devices (0 when none / no runtime) */
int mvk_device_count(void);

/* ---- model / scene / params factories (host only, no device needed) ------ */
/* Parametric N-link HYDRUS (planar yaw chain) and DRAGON (pitch+yaw joints,
 * gimballed dual rotors).  The reference loads these from URDFs that live in
 * the hydrus / dragon packages (hydrus_passing_planning.launch:19-22,
 * dragon_passing_planning.launch:21); geometry hints: hydrus.yaml:31-33,
 * quad.urdf.xacro:7-12.  Numbers are frozen in DESIGN.md section 3. */
int mvk_model_hydrus(int n_links, mvk_model* out);
int mvk_model_dragon(int n_links, mvk_model* out);
/* defaults of motion_planning.cpp:465-532 with the model's joint limits */
int mvk_params_default(const mvk_model* model, int planning_mode, int motion_type, mvk_params* out);
/* The gap worlds of SamplingBasedMethod::setCollisionWorld
 * (squeeze_navigation/src/planner/sampling_based_method.cpp:76-113 / :114-198).
 * Returns the number of boxes written (<= cap), negative on error. */
int mvk_scene_horizontal_gap(double gap_left_x, double gap_y_offset, double gap_left_width,
                             double gap_right_width, mvk_box* out, int cap);
int mvk_scene_vertical_gap(double gap_x_width, double gap_y_width, double gap_height,
                           int planning_walls, double ceiling_offset, double side_wall_width,
                           mvk_box* out, int cap);

/* Broadphase diagnostic (host only, no device needed).  mvk_create / mvk_set_scene
 * index the world boxes in a uniform grid -- the role of FCL's broadphase manager
 * under PlanningScene::checkCollision (call site motion_planning.cpp:156-161) -- whose
 * cells list every box within `radius` of any of their points.  This entry point
 * builds that same grid and returns, for each query point, the ids its cell lists
 * (a superset of the boxes within `radius` of the point; tests check exactly that).
 * counts[i] = number of ids for point i (may exceed cap; only cap are written). */
int mvk_broadphase_query(const mvk_box* boxes, size_t n_boxes, double radius, const double* points /*[n][3]*/,
                         size_t n_points, int32_t* counts, int32_t* ids /*[n][cap]*/, int cap);
/* The same for the far grid of the distance paths (the role of FCL's distance traversal of the broadphase tree under
 * CollisionAvoidance::getConstraint, collision_avoidance.cpp:147-153: nearest objects first, at any distance): a cell
 * lists every box whose distance to a point of the cell is within `reach` of the smallest such distance, so the boxes
 * that can be nearest to a primitive of bounding radius <= reach centred there are all listed.  Covers the boxes'
 * bound inflated by `margin`; counts[i] = -1 for a point outside it (the kernels then sweep all boxes). */
int mvk_far_grid_query(const mvk_box* boxes, size_t n_boxes, double reach, double margin, const double* points /*[n][3]*/,
                       size_t n_points, int32_t* counts, int32_t* ids /*[n][cap]*/, int cap);

/* ---- context --------------------------------------------------------------- */
/* Replaces MotionPlanning's ownership of robot_model_ptr_ + planning_scene_
 * (motion_planning.h:128,133; planInit motion_planning.cpp:354-361). */
mvk_ctx* mvk_create(const mvk_model* model, const mvk_box* boxes, size_t n_boxes,
                    const mvk_params* params, int device);
void mvk_destroy(mvk_ctx* ctx);
/* MotionPlanning::setScene (motion_planning.cpp:167-175): replaces the obstacle set. */
int mvk_set_scene(mvk_ctx* ctx, const mvk_box* boxes, size_t n_boxes);
int mvk_set_params(mvk_ctx* ctx, const mvk_params* params);
int mvk_get_params(const mvk_ctx* ctx, mvk_params* out);
/* Arithmetic of the flag path (the entry points that return verdicts only: mvk_is_valid_batch[_dev] without
 * distance / margin / band outputs, mvk_is_valid_mask_dev, mvk_sweep_mask_dev, mvk_check_motion_batch*).
 * The reference computes everything in fp64 (motion_planning.cpp:76-164 over KDL / Eigen / FCL doubles), and so
 * does MVK_PRECISION_F64, the default.  MVK_PRECISION_F32_REFINED is the fp32 variant BASELINE.json's north star
 * asks for: the model update, the stability tests and the frames of the collision screen run in fp32 with
 * first-order bounds on their evaluation errors; a verdict is taken in fp32 only where the bounds leave no
 * doubt, every other state (and every possible contact, as always) is evaluated again in fp64.  The flags are
 * therefore the SAME as with MVK_PRECISION_F64 outside the 1e-6 boundary band (tests/test_gpu_f32.py); tolerance
 * of the fp32 stage itself: slack errors bounded by ~1e-4 relative, positions by 1e-3 m.
 * Distances, margins and details are always fp64 (full path). */
enum { MVK_PRECISION_F64 = 0, MVK_PRECISION_F32_REFINED = 1 };
int mvk_set_flag_precision(mvk_ctx* ctx, int precision);
/* number of states the last MVK_PRECISION_F32_REFINED call on this ctx handed to the fp64 screen (waits for that
 * call; measurement only, like mvk_last_kernel_ms) */
int mvk_last_refined_count(mvk_ctx* ctx, uint64_t* out);

/* Kernel timing for benchmarks (no counterpart in the reference, which only times the whole
 * solve: motion_planning.cpp:406-410).  When enabled, every batched call records HIP events on
 * its launch stream around its kernels; mvk_last_kernel_ms waits for the last call and returns
 * the duration of its first kernel (the screen kernel of the flag path, or the full-evaluation
 * kernel) and of its second kernel (the exact stage of the flag path; 0 when there is none). */
int mvk_set_profiling(mvk_ctx* ctx, int enable);
int mvk_last_kernel_ms(mvk_ctx* ctx, float* first_ms, float* second_ms);
/* rows of a state in the given layout for the ctx's (mode, motion type) */
int mvk_state_dim(const mvk_ctx* ctx, int layout);

/* ---- the predicate --------------------------------------------------------- */
/* MotionPlanning::isStateValid (motion_planning.cpp:76-164) for ONE state in
 * MVK_LAYOUT_OMPL.  Runs the same device kernel on a batch of one.
 * MotionPlanning::checkCollision(const MultilinkState&) (motion_planning.cpp:802-810:
 * collision of an arbitrary state, no stability test, whatever the planning mode) is
 * this call on a ctx whose params have check_stability = 0, check_collision = 1, read
 * through out->collision.  The reference's version always returns true and only warns
 * (quirk Q8); that is not reproduced.  tests/test_gate_overrides.py covers both. */
int mvk_is_valid(mvk_ctx* ctx, const double* state, mvk_result* out);

/* Batched predicate (north star: isValid(states[N])).  flags[i] = 1 valid.
 * min_dist / margin / band may be NULL; asking for any of them selects the
 * full evaluation (no early exit), otherwise the flag-only fast path runs. */
int mvk_is_valid_batch(mvk_ctx* ctx, const double* soa, size_t n, size_t ld, int layout,
                       uint8_t* flags, double* min_dist, double* margin, uint8_t* band);
int mvk_is_valid_batch_dev(mvk_ctx* ctx, const double* d_soa, size_t n, size_t ld, int layout,
                           uint8_t* d_flags, double* d_min_dist, double* d_margin,
                           uint8_t* d_band, void* stream);
/* same, with the validity flags packed one bit per sample (bit i%64 of word
 * i/64, little endian) -- the form that is all-gathered across GPUs. */
int mvk_is_valid_mask_dev(mvk_ctx* ctx, const double* d_soa, size_t n, size_t ld, int layout,
                          uint64_t* d_mask, void* stream);

/* mvk_is_valid_mask_dev for callers that cannot allocate.  No counterpart in the reference; it is for
 * callers that must not reach hipMalloc / hipFree: code under HIP graph capture, tight serving loops,
 * harnesses that forbid allocation.  The other batched entry points keep grow-only buffers per ctx and
 * may allocate when a batch is larger than any before; this one allocates nothing, ever: the flag
 * path's hand-off queues, records and counters live in one file-scope __device__ array of the library
 * (22.1 MiB per GPU, held until the process ends), so
 *   - n is limited to MVK_NOALLOC_MAX_STATES per call (MVK_ERR_CAPACITY beyond, nothing is launched);
 *   - every ctx on one GPU shares that array: calls of this entry point that overlap on one GPU
 *     (two streams, two ctxs, two threads) race.  Issue them one after another in stream order.
 * Same arguments as mvk_is_valid_mask_dev and the same result bits as its fp64 path: this entry point always
 * computes in fp64, whatever mvk_set_flag_precision selected (tests/test_zz_no_alloc.py).
 * Under graph capture (`stream` is capturing): the call captures a memset node and its kernels and records
 * nothing about itself in the ctx, so launches of the graph are the caller's to keep in order with other calls
 * on the ctx and with other users of the array.  HIP refuses synchronising calls from a capturing thread, so the
 * library cannot wait for an earlier asynchronous call itself: if the ctx's previous call went to ANOTHER stream
 * through a *_dev entry point and nothing has waited for it since, the call is refused (MVK_ERR_INVALID_ARG,
 * nothing captured) -- synchronise that stream, or make any blocking call on the ctx, before the capture begins
 * (tests/test_zz_no_alloc.py::test_noalloc_call_captured_into_a_graph_replays). */
#define MVK_NOALLOC_MAX_STATES 65536
int mvk_is_valid_mask_noalloc_dev(mvk_ctx* ctx, const double* d_soa, size_t n, size_t ld, int layout,
                                  uint64_t* d_mask, void* stream);

/* C-space sweep (BASELINE config 5; the reference's manual joint-space validity sweep is
 * sampling_based_method/scripts/state_validation_l4.py:46-68).  Sample i (i = first_index ..
 * first_index + n - 1) has the fixed root pose `root_pose7` (px py pz qx qy qz qw) and joints
 *   j = joint_low[j] + (joint_high[j] - joint_low[j]) * (u + 0.5) / 2^32,
 *   u = philox4x32-10(counter = {lo32(i), hi32(i), j / 4, 0}, key = {lo32(seed), hi32(seed)})[j % 4]
 * with the joint bounds of the ctx's params: the states are generated in the kernel from the
 * sample index, no input is read (a 64 M-sample sweep sharded over 8 GPUs moves 8 MiB of mask).
 * Bit i - first_index of d_mask is the verdict of sample i.  mvk_sweep_states materialises the
 * same states on the host (SoA, MVK_LAYOUT_POSE7 rows; no device needed) for checking. */
int mvk_sweep_mask_dev(mvk_ctx* ctx, const double* root_pose7, uint64_t first_index, size_t n, uint64_t seed,
                       uint64_t* d_mask, void* stream);
int mvk_sweep_states(const mvk_params* params, int n_joint, const double* root_pose7, uint64_t first_index,
                     size_t n, uint64_t seed, double* soa, size_t ld);

/* number of valid samples among the first n_bits of a packed mask, written to *d_count (device):
 * the scalar that is all-reduced across GPUs next to the mask all-gather */
int mvk_mask_count_dev(const uint64_t* d_mask, size_t n_bits, uint64_t* d_count, void* stream);

/* ---- multi-GPU exchange (BASELINE north star: "RCCL over xGMI is used only to gather valid-sample
 * masks and reduce min-distance/margin statistics") -------------------------------------------------------------------
 * No counterpart in the reference, which is one process (squeeze_navigation/src/planner/sampling_based_method.cpp:214-217
 * -> MotionPlanning::plan, single-threaded ros::spin).  One process per GPU; every rank holds the whole model and scene
 * and evaluates the index range mvk_shard_range gives it with the entry points above (no data-path collective); these
 * calls are the only communication.  RCCL is bound at run time (librccl.so.1): MVK_ERR_NO_DEVICE when it is absent.
 *   mvk_comm_unique_id   rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by whatever
 *                        channel the job has (file, socket, MPI, a torch.distributed store)
 *   mvk_comm_create      ncclCommInitRank on `device`; collective: every rank of the job calls it
 *   mvk_gather_mask_dev  ncclAllGather of the packed validity masks: every rank contributes words_per_rank words
 *                        (its shard, zero padded), d_all receives world * words_per_rank words in rank order
 *   mvk_allreduce_count_dev  ncclAllReduce(sum) of uint64 counters in place (e.g. mvk_mask_count_dev's output)
 *   mvk_allreduce_stats  the mvk_stats of every rank's last batch (mvk_reduce_stats) combined: counts summed, extrema
 *                        with their GLOBAL sample index (local index + index_offset, ties -> smallest index); one
 *                        ncclAllGather of 104 bytes per rank + a host merge; blocks until `stream` has delivered it
 *   mvk_merge_stats      that merge alone, on the host (records that travelled by another channel)
 *   mvk_exchange_masks_dev  one step's whole exchange in one host call, shaped for a loop that overlaps it with the
 *                        next steps' kernels: waits (by event, on the device) for `producer_stream`, all-gathers the
 *                        shard on `exchange_stream`, counts the valid bits of the gathered words into *d_count there
 *                        (nullable), and marks buffer set `slot` (0..7) busy; mvk_exchange_wait(slot) blocks the host
 *                        until that set's exchange is through (slot -1: all sets) -- call it before refilling the set
 * Collectives are enqueued on `stream`; a communicator serves one stream at a time. */
#define MVK_COMM_ID_BYTES 128
typedef struct mvk_comm mvk_comm;
int mvk_comm_unique_id(unsigned char* id /* [MVK_COMM_ID_BYTES] */);
int mvk_comm_create(const unsigned char* id, int rank, int world, int device, mvk_comm** out);
int mvk_comm_destroy(mvk_comm* comm);
int mvk_comm_rank(const mvk_comm* comm);
int mvk_comm_world(const mvk_comm* comm);
/* contiguous shard [lo, hi) of n samples for `rank`, aligned to 64 so that mask words never straddle ranks;
 * words_per_rank = the (padded) mask words every rank contributes to mvk_gather_mask_dev */
int mvk_shard_range(size_t n, int rank, int world, size_t* lo, size_t* hi, size_t* words_per_rank);
int mvk_gather_mask_dev(mvk_comm* comm, const uint64_t* d_shard, size_t words_per_rank, uint64_t* d_all, void* stream);
int mvk_allreduce_count_dev(mvk_comm* comm, uint64_t* d_count, size_t n_counts, void* stream);
int mvk_exchange_masks_dev(mvk_comm* comm, int slot, const uint64_t* d_shard, size_t words_per_rank, uint64_t* d_all,
                           uint64_t* d_count, void* producer_stream, void* exchange_stream);
int mvk_exchange_wait(mvk_comm* comm, int slot);
int mvk_allreduce_stats(mvk_comm* comm, const mvk_stats* local, uint64_t index_offset, mvk_stats* global, void* stream);
int mvk_merge_stats(const mvk_stats* locals, const uint64_t* offsets, int world, mvk_stats* global);

/* ---- pipelined batches: several contexts, several streams --------------------------------------------------------
 * No counterpart in the reference (isStateValid takes one state per call, motion_planning.cpp:76-164); part of the
 * batched entry point the north star adds.  Calls of ONE ctx are ordered on the GPU -- the flag path's exact stage
 * runs behind its screen stage and before the ctx's next call, since a caller may read call k's mask as soon as call k
 * has finished on its stream -- so a stream of independent batches through one ctx leaves the chip a twelfth busy
 * during every exact stage.  A pipe owns `depth` contexts (same model, scene and parameters), each with a stream of its
 * own, and hands consecutive batches to them round-robin: the exact stage of batch k runs beside the screen stage of
 * batch k + 1.  Per batch the semantics are those of the ctx entry point it calls (mvk_is_valid_mask_dev,
 * mvk_sweep_mask_dev, mvk_eval_batch_dev): same kernels, same bits.  1 M-state batches, HYDRUS-6, 1000 boxes: 0.123 ms
 * per batch with depth 2 against 0.140 through one ctx (bench.py: `value` against `value_one_ctx`).
 *   producer_stream: the stream the batch's inputs are produced on (the batch is ordered behind everything enqueued
 *     there so far), or MVK_PIPE_NO_PRODUCER when they are ready.  NULL is the null stream, as everywhere.
 *   ticket: names the lane the batch went to; mvk_pipe_wait(pipe, ticket, consumer_stream) orders consumer_stream
 *     behind the lane's LATEST batch (no host wait), with MVK_PIPE_HOST it blocks the host until that batch is done.
 *     A lane runs its batches in order, so a ticket stays good until `depth` further batches have been submitted; the
 *     output buffers of a batch must not be handed to another batch before it has been waited for.
 *   mvk_pipe_drain: blocks the host until every submitted batch is done.  mvk_pipe_ctx: a lane's ctx (statistics,
 *     profiling, parameters; do not call it while the lane has work in flight on another stream).  mvk_pipe_stream: a
 *     lane's stream, for work that has to follow a batch without a host wait (the ticket of a batch is its lane).
 * A pipe is thread-compatible like a ctx: one host thread at a time.  Memory: `depth` times a ctx's buffers. */
typedef struct mvk_pipe mvk_pipe;
#define MVK_PIPE_NO_PRODUCER ((void*)(intptr_t)-1)
#define MVK_PIPE_HOST ((void*)(intptr_t)-1)
int mvk_pipe_create(const mvk_model* model, const mvk_box* boxes, size_t n_boxes, const mvk_params* params, int device,
                    int depth /* 1..8 */, mvk_pipe** out);
void mvk_pipe_destroy(mvk_pipe* pipe);
int mvk_pipe_depth(const mvk_pipe* pipe);
mvk_ctx* mvk_pipe_ctx(mvk_pipe* pipe, int lane);
void* mvk_pipe_stream(mvk_pipe* pipe, int lane);   /* the lane's hipStream_t: e.g. the producer stream of mvk_exchange_masks_dev */
int mvk_pipe_set_scene(mvk_pipe* pipe, const mvk_box* boxes, size_t n_boxes);   /* waits for every lane first */
int mvk_pipe_is_valid_mask(mvk_pipe* pipe, const double* d_soa, size_t n, size_t ld, int layout, uint64_t* d_mask,
                           void* producer_stream, int* ticket);
int mvk_pipe_sweep_mask(mvk_pipe* pipe, const double* root_pose7, uint64_t first_index, size_t n, uint64_t seed,
                        uint64_t* d_mask, int* ticket);
int mvk_pipe_eval_batch(mvk_pipe* pipe, const double* d_soa, size_t n, size_t ld, int layout, uint8_t* d_flags,
                        double* d_min_dist, double* d_margin, uint8_t* d_band, void* producer_stream, int* ticket);
int mvk_pipe_wait(mvk_pipe* pipe, int ticket, void* consumer_stream);
int mvk_pipe_drain(mvk_pipe* pipe);

/* Full evaluation with per-sample model quantities (what MultilinkState /
 * addState expose: multilink_state.cpp:98-137, motion_planning.cpp:546-642). */
int mvk_eval_batch(mvk_ctx* ctx, const double* soa, size_t n, size_t ld, int layout,
                   uint8_t* flags, double* min_dist, double* margin, uint8_t* band,
                   const mvk_detail_out* detail);
int mvk_eval_batch_dev(mvk_ctx* ctx, const double* d_soa, size_t n, size_t ld, int layout,
                       uint8_t* d_flags, double* d_min_dist, double* d_margin, uint8_t* d_band,
                       const mvk_detail_out* d_detail, void* stream);
/* The fp32 variant of the distance and margin outputs (the reference is all fp64, motion_planning.cpp:76-164; the
 * north star asks for an fp32 variant reported with its own tolerance).  Same states, same verdict rule, outputs as
 * floats:
 *   d_min_dist   [n]  minimum signed distance with the narrowphase (separating-axis gaps, edge-to-box and
 *                     segment-to-box distances) evaluated in fp32 on the fp32 copy of the boxes; the chain's frames
 *                     stay fp64.  TOLERANCE, per state: |d_min_dist - min_dist(fp64)| <=
 *                     mvk_f32_distance_tolerance(ctx, X) = MVK_F32_DIST_TOL_ULPS * 2^-24 * (X + reach + 1 m), X the
 *                     largest |coordinate| of the state's root position and `reach` the robot's: 3e-5 m for a HYDRUS
 *                     5 m from the origin.  Observed: 2e-7 .. 1e-6 m on the benchmark scenes
 *                     (tests/test_gpu_f32_dist.py, bench.py's fp32_distances leg).  Scenes without a distance grid
 *                     (< 64 boxes) and lanes whose candidate list overflowed are evaluated in fp64.
 *   d_margin     [n]  midpoint of an ENCLOSURE of the fp64 stability margin computed in fp32 with running error
 *                     bounds (the walk of MVK_PRECISION_F32_REFINED), d_margin_err [n] its half width: the fp64
 *                     margin lies in [d_margin - d_margin_err, d_margin + d_margin_err]; +inf where a test could not
 *                     be bounded in fp32 (a nearly singular wrench matrix).  The tolerance is per state, not a constant.
 *   d_flags      [n]  finite && d_margin >= 0 && d_min_dist > 0 (as mvk_eval_batch_dev, on the fp32 values)
 *   d_band       [n]  1 where the fp32 values cannot settle the verdict: |d_min_dist| <= max(eps_dist, the state's
 *                     tolerance above), or the margin's enclosure reaches within eps_margin of zero.  Outside the band the
 *                     flags equal the fp64 flags.
 * Any output may be NULL.  Statistics (mvk_reduce_stats) afterwards are those of the fp32 values. */
#define MVK_F32_DIST_TOL_ULPS 64.0
double mvk_f32_distance_tolerance(mvk_ctx* ctx, double coord_abs_max);
int mvk_eval_batch_f32_dev(mvk_ctx* ctx, const double* d_soa, size_t n, size_t ld, int layout, uint8_t* d_flags,
                           float* d_min_dist, float* d_margin, float* d_margin_err, uint8_t* d_band, void* stream);
/* extrema over the last *_batch / eval call on this ctx (device reduction) */
int mvk_reduce_stats(mvk_ctx* ctx, mvk_stats* out);

/* ---- collision-avoidance constraint rows --------------------- *
 * differential_kinematics CollisionAvoidance::getConstraint
 * (differential_kinematics/src/constraint/collision_avoidance.cpp:102-277) for a batch of
 * states: per robot collision primitive q (model order) the minimum distance to the world
 * boxes with nearest points (:147-153), the closest box, the row  n^T J_lin  (:255-260) over
 * the columns [root linear velocity (world), root angular velocity (world), planned joints
 * j1..jJ], and the lower bound damplingBound(dist, -gain, range, forbidden) (:262-265,
 * constraint/base_plugin.h:108-113; -1e6 = unconstrained).  ok[i] = 0 where getConstraint
 * would fail because a distance is <= 0 (:214-218); such primitives get NaN points and a
 * zero row.  Gimbal angles count as fixed.  Every output pointer may be NULL. */
typedef struct mvk_avoid_params {
  double damping_gain;     /* collision_damping_gain, default 0.01 (:68) */
  double constraint_range; /* collision_distance_constraint_range, default 0.03 (:69) */
  double forbidden_range;  /* collision_distance_forbidden_range, default 0.01 (:70) */
  int32_t full_body;       /* 0: the six root columns are zeroed (:267) */
} mvk_avoid_params;
typedef struct mvk_avoid_out {
  double* dist;    /* [n_prim][ld] */
  int32_t* box;    /* [n_prim][ld]  index of the closest box, -1 in an empty world */
  double* p_env;   /* [3*n_prim][ld]  nearest point on the environment, world frame */
  double* p_robot; /* [3*n_prim][ld]  nearest point on the robot primitive, world frame */
  double* row;     /* [(6+J)*n_prim][ld] */
  double* lb;      /* [n_prim][ld] */
  uint8_t* ok;     /* [ld] */
} mvk_avoid_out;
int mvk_avoid_params_default(mvk_avoid_params* out);
int mvk_collision_avoidance_batch(mvk_ctx* ctx, const double* soa, size_t n, size_t ld, int layout,
                                  const mvk_avoid_params* prm, const mvk_avoid_out* out);
/* device pointers; output rows use the same pitch ld as the states */
int mvk_collision_avoidance_batch_dev(mvk_ctx* ctx, const double* d_soa, size_t n, size_t ld, int layout,
                                      const mvk_avoid_params* prm, const mvk_avoid_out* d_out, void* stream);

/* ---- motion validation ------------------------------------------------------- */
/* ompl::base::DiscreteMotionValidator::checkMotion bound at
 * motion_planning.cpp:238,249 (and the densification loop :434-444).
 * s1/s2 are SoA batches in MVK_LAYOUT_OMPL.  nd[e] (or the scalar nd_uniform
 * when nd == NULL) is validSegmentCount; states j/nd, j = 1..nd are tested
 * (s1 is assumed valid, as OMPL does).  ok[e] = 1 when all are valid;
 * first_invalid[e] = smallest invalid j, or nd+1 when none (so the
 * lastValid fraction of the 3-argument checkMotion is (first_invalid-1)/nd). */
int mvk_check_motion_batch(mvk_ctx* ctx, const double* s1_soa, const double* s2_soa,
                           size_t n_edges, size_t ld, const int32_t* nd, int32_t nd_uniform,
                           uint8_t* ok, int32_t* first_invalid);
/* The same call in two halves (no counterpart in the reference: OMPL's checkMotion blocks).
 * begin copies the edges, enqueues the transfers and the kernels on the ctx's stream and returns;
 * end waits for them and writes the verdicts.  Exactly one end per begin, no other call on the
 * ctx in between.  mvk_check_motion_batch is begin followed by end. */
int mvk_check_motion_batch_begin(mvk_ctx* ctx, const double* s1_soa, const double* s2_soa,
                                 size_t n_edges, size_t ld, const int32_t* nd, int32_t nd_uniform);
int mvk_check_motion_batch_end(mvk_ctx* ctx, uint8_t* ok, int32_t* first_invalid);
int mvk_check_motion_batch_dev(mvk_ctx* ctx, const double* d_s1, const double* d_s2,
                               size_t n_edges, size_t ld, const int32_t* d_nd, int32_t nd_uniform,
                               uint8_t* d_ok, int32_t* d_first_invalid, void* stream);

/* OMPL state-space arithmetic (host, scalar, no device needed; used by the
 * adapter and the planner): StateSpace::distance / interpolate /
 * validSegmentCount / getMaximumExtent of the space planInit builds for
 * (planning_mode, motion_type) (motion_planning.cpp:192-258).  States are in
 * MVK_LAYOUT_OMPL; n_joint = J. */
int mvk_space_dim(const mvk_params* p, int n_joint);
double mvk_space_distance(const mvk_params* p, int n_joint, const double* s1, const double* s2);
int mvk_space_interpolate(const mvk_params* p, int n_joint, const double* s1, const double* s2,
                          double t, double* out);
int mvk_space_valid_segment_count(const mvk_params* p, int n_joint, const double* s1,
                                  const double* s2);
double mvk_space_max_extent(const mvk_params* p, int n_joint);
/* the unpack of motion_planning.cpp:81-129: OMPL state -> px py pz qx qy qz qw j1..jJ */
int mvk_space_unpack(const mvk_params* p, int n_joint, const double* s, double* pose7j);


/* ---- planner ------------------------------------------ *
 * Host-side RRT-Connect that drives the batched checker.  The reference plans
 * with OMPL RRT* (motion_planning.cpp:391-408); BASELINE.json's metric asks for
 * an RRT-Connect gap-pass solve time, so this is the build's own planner with
 * OMPL's RRTConnect structure (two trees, extend by `range`, greedy connect),
 * batched: every iteration grows `batch` random extensions with one edge-check
 * launch and answers them with one connect launch. */
typedef struct mvk_checker {
  /* both callbacks return 0 on success; arrays are SoA in MVK_LAYOUT_OMPL */
  int (*check_states)(void* user, const double* soa, size_t n, size_t ld, uint8_t* ok);
  int (*check_motions)(void* user, const double* s1, const double* s2, size_t n_edges, size_t ld,
                       const int32_t* nd, uint8_t* ok, int32_t* first_invalid);
  void* user;
  /* optional (may be NULL, both or neither): the same edge check in two halves.  begin takes the
   * edges and returns once the work is under way; end blocks and delivers the verdicts of the
   * last begin.  With them the planner searches its trees while the GPU checks. */
  int (*check_motions_begin)(void* user, const double* s1, const double* s2, size_t n_edges, size_t ld,
                             const int32_t* nd);
  int (*check_motions_end)(void* user, uint8_t* ok, int32_t* first_invalid);
} mvk_checker;

typedef struct mvk_plan_request {
  const double* start; /* MVK_LAYOUT_OMPL */
  const double* goal;
  double range;        /* max extension; <= 0: 0.2 * maximum extent (OMPL's default) */
  double time_limit_s; /* solving_time_limit (motion_planning.cpp:530) */
  uint64_t seed;
  int32_t max_iterations; /* <= 0: unlimited */
  int32_t batch;          /* extensions per iteration (>= 1) */
} mvk_plan_request;

typedef struct mvk_plan_result {
  int32_t solved;
  int32_t n_waypoints;
  double solve_ms;      /* wall time of the search (planning_time of the log, in ms) */
  double path_length;   /* sum of space distances along the way-points (motion_cost) */
  uint64_t iterations, state_checks, edge_checks, edge_steps; /* edges / interpolation steps submitted to the checker */
  int32_t tree_sizes[2];
} mvk_plan_result;

/* generic form: validity through callbacks (lets the host logic run against any checker) */
int mvk_plan_rrt_connect_cb(const mvk_params* params, int n_joint, const mvk_checker* checker,
                            const mvk_plan_request* req, double* waypoints /*[cap][dim] row-major*/,
                            int cap, mvk_plan_result* out);
/* bound to the GPU context: MotionPlanning::plan (motion_planning.cpp:377-463) */
int mvk_plan_rrt_connect(mvk_ctx* ctx, const mvk_plan_request* req, double* waypoints, int cap,
                         mvk_plan_result* out);

/* Solution densification exactly as motion_planning.cpp:424-446: for every pair of
 * consecutive way-points addState(s1), the nd-1 interior states, addState(s2) -- so
 * interior way-points appear twice, as in the reference's logs.  Returns the number of
 * states written (row-major [n][dim]); negative on error / insufficient cap. */
int mvk_path_densify(const mvk_params* params, int n_joint, const double* waypoints, int n_waypoints,
                     double* states, int cap);

/* Path log in the reference's text format (savePath/loadPath, motion_planning.cpp:644-800;
 * SURVEY.md 6.1).  States are the unpacked px py pz qx qy qz qw j1..jJ rows. */
typedef struct mvk_path_log {
  int32_t n_states, n_joint, planning_mode;
  double planning_time, motion_cost, minimum_var;
  int32_t minimum_var_state_entry;
  double start[MVK_MAX_STATE_DIM], goal[MVK_MAX_STATE_DIM];
} mvk_path_log;
int mvk_path_log_save(const char* file, const mvk_path_log* hdr, const double* states /*[n][7+J]*/);
/* reads the header and up to cap states; returns the number of states in the file */
int mvk_path_log_load(const char* file, int n_joint, mvk_path_log* hdr, double* states, int cap);

/* ---- path post-processing -------------------------------- *
 * What squeeze_motion_planner::Base::plan runs on the planner's discrete path
 * (squeeze_navigation/include/squeeze_navigation/planner/base_plugin.h:72-85).
 * "cog state" rows: CoG position, desired baselink attitude (= root attitude, the baselink is
 * the root link: motion_planning.cpp:188), joints -- the triple MultilinkState keeps. */
/* MultilinkState::convertRootPose2CogPose / convertCogPose2RootPose
 * (aerial_motion_planning_msgs/src/multilink_state.cpp:98-118 / 42-61); host, one state */
int mvk_root_to_cog(const mvk_model* model, const double* root_state /*[7+J]*/, double* cog_state);
int mvk_cog_to_root(const mvk_model* model, const double* cog_state /*[7+J]*/, double* root_state);
/* ContinuousPathGenerator::discretePathResampling
 * (aerial_motion_planning_msgs/src/continuous_path_generator.cpp:119-230); parameters
 * resampling_angular_rate (1.0) and resampling_seg_diff_thresh (0.2) (:124-125).
 * States are row major [n][7+J]; returns the number of states written, negative on error. */
int mvk_path_resample(const double* cog_states, int n, int n_joint, double angular_rate, double seg_diff_thresh,
                      double* out, int cap);
/* ContinuousPathGenerator::discretePathSmoothing (continuous_path_generator.cpp:57-117): low-pass filter over the
 * ROOT states of a path (root xyz + joints in one filter, root orientation in another, both of factor filter_rate,
 * default 0.1 (:61); the filters start on the first state and the last state is repeated while
 * index < n + 1 / filter_rate in floating point (:83), so the output has n + 1 / filter_rate states when that is an
 * integer (0.1: n + 10) and n + ceil(1 / filter_rate) otherwise (0.3: n + 4), and ends on the last one).  PARITY UNPINNED: the filters themselves
 * (FirFilter / FirFilterQuaternion of the external kalman_filter package) are absent; restated as a moving average
 * over round(1 / filter_rate) inputs.  out == NULL: returns the number of states it would write. */
int mvk_path_smooth(const double* states, int n, int n_joint, double filter_rate, double* out, int cap);
/* ContinuousPathGenerator::calcContinuousPath (:232-282) + BsplineRos::initialize
 * (bspline_ros/src/bspline_ros.cpp:45-86): clamped uniform B-spline of `degree` (bspline_degree,
 * default 5) over [0, period] through the control points (cog xyz, roll pitch yaw with a
 * continuous yaw, joints), first and last state doubled.  Curve dimension = 6 + J. */
typedef struct mvk_spline mvk_spline;
mvk_spline* mvk_spline_create(const double* cog_states, int n, int n_joint, double period, int degree);
void mvk_spline_destroy(mvk_spline* s);
int mvk_spline_dim(const mvk_spline* s);
/* BsplineRos::evaluate (bspline_ros.cpp:88-110; getPositionVector / getVelocityVector of
 * continuous_path_generator.cpp:47-55): derive-th derivative at nt times (clamped to the period).
 * Host: out [nt][dim] row major.  Device: d_t [nt], d_out [dim][ld] SoA, one lane per time. */
int mvk_spline_eval(const mvk_spline* s, const double* t, int nt, int derive, double* out);
int mvk_spline_eval_dev(mvk_spline* s, int device, const double* d_t, size_t nt, int derive, double* d_out,
                        size_t ld, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MVK_H_ */
