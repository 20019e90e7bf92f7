// mvk_model.cpp -- host-side factories: parametric robot models, default
// planner parameters and the gap worlds.  No device code in this file.
//
// The reference loads its robots from URDFs of the hydrus / dragon packages
// (squeeze_navigation/launch/hydrus_passing_planning.launch:19-22,
// dragon_passing_planning.launch:21) which are not part of the reference tree,
// so the numbers below are this build's frozen synthetic descriptions
// (DESIGN.md section 3).  In-reference hints that are honoured:
//   link_length 0.6, protector radius 0.2
//       (aerial_manipulation/aerial_transportation/config/hydrus.yaml:31-33)
//   joint range +-1.58, edf_max_tilt 0.1
//       (squeeze_navigation/config/sampling_based_method/dragon_quad.yaml:43-58)
//   collision link sets {link*, thrust*} / {link*, inter_joint*, gimbal*_{roll,pitch}_module}
//       (sampling_based_method/config/{hydrus,dragon}/quad/moveit.srdf)
// Pinned by the reference's planner logs (tests/golden/fit_dragon_model.py):
//   DRAGON link 0.424 (pitch axis at its end), yaw axis 0.0515 behind the pitch axis, gimbal roll axis
//   origin at mid-link, thrust-frame origin 0.037 up the roll frame's z axis: with these four numbers the
//   rp_position_margin column of nine logs (2,761 states) is reproduced to the printed digits;
//   HYDRUS rotor at mid-link: reproduces minimum_var 0.452857 @ state 200 of path_log/hydrus/hydrus_planning_log.txt.
#include <cmath>
#include <cstring>

#include "mvk_internal.h"

namespace {

void set3(double* d, double x, double y, double z) { d[0] = x; d[1] = y; d[2] = z; }
void identity9(double* r) {
  std::memset(r, 0, 9 * sizeof(double));
  r[0] = r[4] = r[8] = 1.0;
}

int add_prim(mvk_model* m, int type, int seg, int frame, int rotor, double px, double py, double pz,
             double d0, double d1, double d2) {
  int q = m->n_prim++;
  m->prim_type[q] = type;
  m->prim_seg[q] = seg;
  m->prim_frame[q] = frame;
  m->prim_rotor[q] = rotor;
  set3(m->prim_pos[q], px, py, pz);
  identity9(m->prim_rot[q]);
  set3(m->prim_dim[q], d0, d1, d2);
  return q;
}

void blank(mvk_model* m) {
  std::memset(m, 0, sizeof(*m));
  m->abi_version = MVK_ABI_VERSION;
  for (int k = 0; k < MVK_MAX_SEG; ++k) {
    m->seg_axis[k] = -1;
    m->seg_joint[k] = -1;
    m->seg_link[k] = -1;
  }
  m->gravity = 9.80665;
}

}  // namespace

extern "C" int mvk_model_hydrus(int n_links, mvk_model* m) {
  if (!m || n_links < 3 || n_links > MVK_MAX_ROTOR) return mvk::fail(MVK_ERR_INVALID_ARG, "mvk_model_hydrus: n_links must be 3..8");
  blank(m);
  const double L = 0.6;
  m->robot_type = MVK_ROBOT_HYDRUS;
  m->n_seg = n_links;
  m->n_joint = n_links - 1;
  m->n_rotor = n_links;
  m->wrench_dof = 4;
  m->has_gimbal = 0;
  m->link_length = L;
  m->m_f_rate = -0.0143;
  m->thrust_min = 2.0;
  m->thrust_max = 18.0;
  m->fc_t_min_thre = 0.2;
  m->fc_rp_min_thre = 3.0;  // hydrus_quad.yaml:66
  for (int k = 0; k < n_links; ++k) {
    m->seg_link[k] = k;
    if (k > 0) {
      m->seg_axis[k] = 2;  // yaw joint at the end of the previous link
      m->seg_joint[k] = k - 1;
      set3(m->seg_t[k], L, 0.0, 0.0);
    }
    // link1 carries battery + processor
    m->seg_mass[k] = k == 0 ? 1.30 : 1.00;
    set3(m->seg_com[k], k == 0 ? 0.28 : 0.30, 0.0, k == 0 ? -0.03 : -0.02);
    m->rotor_seg[k] = k;
    set3(m->rotor_pos[k], 0.5 * L, 0.0, 0.05);
    m->rotor_sigma[k] = (k % 2 == 0) ? 1.0 : -1.0;
    // "link<k>": body bar; "thrust<k>": propeller protector (r = 0.2) as a short
    // vertical capsule (the URDF cylinder has no closed-form box distance)
    add_prim(m, MVK_PRIM_BOX, k, MVK_FRAME_SEG, -1, 0.5 * L, 0.0, 0.0, 0.5 * L, 0.025, 0.025);
    add_prim(m, MVK_PRIM_CAPSULE, k, MVK_FRAME_SEG, -1, 0.5 * L, 0.0, 0.05, 0.2, 0.03, 0.0);
  }
  for (int j = 0; j < m->n_joint; ++j) {
    m->joint_low[j] = -1.58;
    m->joint_high[j] = 1.58;
  }
  return MVK_OK;
}

extern "C" int mvk_model_dragon(int n_links, mvk_model* m) {
  if (!m || n_links < 3 || n_links > MVK_MAX_ROTOR || 2 * n_links - 1 > MVK_MAX_SEG ||
      2 * (n_links - 1) > MVK_MAX_JOINT)
    return mvk::fail(MVK_ERR_INVALID_ARG, "mvk_model_dragon: n_links must be 3..8");
  blank(m);
  const double L = 0.424;
  m->robot_type = MVK_ROBOT_DRAGON;
  m->n_seg = 2 * n_links - 1;
  m->n_joint = 2 * (n_links - 1);
  m->n_rotor = n_links;
  m->wrench_dof = 3;  // z, roll, pitch: vectoring handles yaw
  m->has_gimbal = 1;
  m->link_length = L;
  m->m_f_rate = 0.0;  // contra-rotating dual fans
  m->thrust_min = 10.0;  // path_log/dragon/..._param.txt:13-14 (f_min 10, f_max 26)
  m->thrust_max = 26.0;
  m->fc_t_min_thre = 0.0;   // off (fc_t is identically 0 with kappa = 0)
  m->fc_rp_min_thre = 2.0;
  m->rotor_roll_z = 0.037;  // thrust-frame origin above the roll axis (pinned by the logs' margin column)
  m->edf_offset_y = 0.09;
  m->edf_radius = 0.065;
  m->edf_max_tilt = 0.1;  // dragon_quad.yaml:56
  const double zx[9] = {0, 0, 1, 0, 1, 0, -1, 0, 0};  // maps local z onto the link x axis
  for (int i = 0; i < n_links; ++i) {
    int ks = 2 * i;  // segment of link i+1
    m->seg_link[ks] = i;
    m->seg_mass[ks] = 1.85;  // pipe + gimbal module + dual EDF, lumped
    set3(m->seg_com[ks], 0.5 * L, 0.0, -0.01);
    if (i > 0) {
      int kj = ks - 1;  // inter_joint<i>: pitch about y at the end of link i
      m->seg_axis[kj] = 1;
      m->seg_joint[kj] = 2 * (i - 1);
      set3(m->seg_t[kj], L, 0.0, 0.0);
      m->seg_mass[kj] = 0.12;
      set3(m->seg_com[kj], 0.0, 0.0, 0.0);
      m->seg_axis[ks] = 2;  // yaw about z
      m->seg_joint[ks] = 2 * (i - 1) + 1;
      set3(m->seg_t[ks], 0.0515, 0.0, 0.0);  // pitch axis -> yaw axis (pinned by the logs' margin column)
    }
    m->rotor_seg[i] = ks;
    set3(m->rotor_pos[i], 0.5 * L, 0.0, 0.0);
    m->rotor_sigma[i] = (i % 2 == 0) ? 1.0 : -1.0;
    int q = add_prim(m, MVK_PRIM_CAPSULE, ks, MVK_FRAME_SEG, -1, 0.5 * L, 0.0, 0.0, 0.025, 0.19, 0.0);
    std::memcpy(m->prim_rot[q], zx, sizeof(zx));
    if (i > 0) add_prim(m, MVK_PRIM_SPHERE, ks - 1, MVK_FRAME_SEG, -1, 0.0, 0.0, 0.0, 0.045, 0.0, 0.0);
    add_prim(m, MVK_PRIM_BOX, ks, MVK_FRAME_GIMBAL_ROLL, i, 0.0, 0.0, 0.0, 0.03, 0.11, 0.03);
    add_prim(m, MVK_PRIM_BOX, ks, MVK_FRAME_GIMBAL_PITCH, i, 0.0, 0.0, 0.0, 0.07, 0.155, 0.035);
  }
  for (int j = 0; j < m->n_joint; ++j) {
    m->joint_low[j] = -1.58;  // dragon_quad.yaml:43-46
    m->joint_high[j] = 1.58;
  }
  return MVK_OK;
}

// Defaults of MotionPlanning::rosParamInit (motion_planning.cpp:465-532).
extern "C" int mvk_params_default(const mvk_model* model, int planning_mode, int motion_type,
                                  mvk_params* p) {
  if (!model || !p || planning_mode < 0 || planning_mode > 2 || motion_type < 0 || motion_type > 1)
    return mvk::fail(MVK_ERR_INVALID_ARG, "mvk_params_default: bad argument");
  std::memset(p, 0, sizeof(*p));
  p->planning_mode = planning_mode;
  p->motion_type = motion_type;
  p->check_stability = -1;
  p->check_collision = -1;
  p->eps_dist = 1e-6;
  p->eps_margin = 1e-6;
  set3(p->pos_low, -2.0, -2.2, -2.2);   // :471,473,475
  set3(p->pos_high, 5.0, 2.2, 2.2);     // :472,474,476
  for (int j = 0; j < model->n_joint; ++j) {
    p->joint_low[j] = model->joint_low[j];  // :482-483 default to the model's limits
    p->joint_high[j] = model->joint_high[j];
  }
  p->validity_resolution = 0.03;         // :527
  p->valid_segment_count_factor = 20;    // :528
  p->mode2_ompl_defaults = 1;            // quirk Q4, :252-258
  p->joint_subspace_weight = 0.001;      // :255
  return MVK_OK;
}

namespace {
// every wall of setCollisionWorld shares orientation.w = 1 (:67)
void wall(mvk_box* b, double cx, double cy, double cz, double dx, double dy, double dz) {
  set3(b->center, cx, cy, cz);
  b->quat[0] = b->quat[1] = b->quat[2] = 0.0;
  b->quat[3] = 1.0;
  set3(b->size, dx, dy, dz);
}
}  // namespace

// SamplingBasedMethod::setCollisionWorld, gap_type == HORIZONTAL_GAP
// (squeeze_navigation/src/planner/sampling_based_method.cpp:76-113).
extern "C" int mvk_scene_horizontal_gap(double gap_left_x, double gap_y_offset, double gap_left_width,
                                        double gap_right_width, mvk_box* out, int cap) {
  if (!out || cap < 4) return mvk::fail(MVK_ERR_INVALID_ARG, "mvk_scene_horizontal_gap: need room for 4 boxes");
  const double tall = 10.0;                        // :89
  const double y_c = (2.5 + gap_y_offset) / 2.0;   // :92
  // :91-96 left wall; :98-101 right wall re-uses the left wall's dimensions (quirk Q2)
  wall(out + 0, gap_left_x + gap_left_width / 2.0, +y_c, 0.0, gap_left_width, 2.5, tall);
  wall(out + 1, gap_left_x + gap_right_width / 2.0, -y_c, 0.0, gap_left_width, 2.5, tall);
  // :103-112 the two long boundary walls
  wall(out + 2, 1.0, +2.5, 0.0, 8.0, 0.6, tall);
  wall(out + 3, 1.0, -2.5, 0.0, 8.0, 0.6, tall);
  return 4;
}

// gap_type == VERTICAL_GAP (:114-198)
extern "C" int mvk_scene_vertical_gap(double gap_x_width, double gap_y_width, double gap_height,
                                      int planning_walls, double ceiling_offset, double side_wall_width,
                                      mvk_box* out, int cap) {
  const int need = planning_walls ? 10 : 4;
  if (!out || cap < need) return mvk::fail(MVK_ERR_INVALID_ARG, "mvk_scene_vertical_gap: output too small");
  const double len = 10.0, thick = 0.05;  // :117,:129
  int n = 0;
  const double sx[4] = {+1, 0, -1, 0}, sy[4] = {0, +1, 0, -1};  // :124-146, plates around the hole
  for (int k = 0; k < 4; ++k)
    wall(out + n++, sx[k] * (gap_x_width / 2.0 + len / 2.0), sy[k] * (gap_y_width / 2.0 + len / 2.0),
         gap_height, len, len, thick);
  if (planning_walls) {
    wall(out + n++, 0.0, 0.0, ceiling_offset + gap_height, len, len, thick);  // :158-165 ceiling
    wall(out + n++, 0.0, 0.0, 0.0, len, len, thick);                          // :168-170 ground
    for (int s = +1; s >= -1; s -= 2)                                         // :173-184
      wall(out + n++, s * side_wall_width / 2.0, 0.0, gap_height / 2.0, thick, side_wall_width, gap_height);
    for (int s = +1; s >= -1; s -= 2)                                         // :186-196
      wall(out + n++, 0.0, s * side_wall_width / 2.0, gap_height / 2.0, side_wall_width, thick, gap_height);
  }
  return n;
}
