// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  PARITY UNPINNED.
// Plain-C entry points so tests/ and bench.py's cpu_baseline leg can drive the oracle via ctypes.
#include <cstring>
#include "orc_loam.h"
#include "orc_img.h"
#include <algorithm>

using namespace orc;

namespace {
int copy_cloud(const Cloud& c, float* buf, int cap) {
  int n = (int)c.size();
  if (buf) {
    int m = n < cap ? n : cap;
    std::memcpy(buf, c.data(), sizeof(PointXYZI) * (size_t)m);
  }
  return n;
}
template <class T>
int copy_vec(const std::vector<T>& v, T* buf, int cap) {
  int n = (int)v.size();
  if (buf) {
    int m = n < cap ? n : cap;
    if (m > 0) std::memcpy(buf, v.data(), sizeof(T) * (size_t)m);
  }
  return n;
}
void pack_summary(const SolveSummary& s, double* trace, int cap_iters, int* n_iters, double* H0, double* g0, int* term,
                  double* costs2) {
  int n = (int)s.iterations.size();
  if (n_iters) *n_iters = n;
  if (trace) {
    for (int i = 0; i < n && i < cap_iters; i++) {
      const IterationSummary& it = s.iterations[i];
      double* r = trace + 8 * i;
      r[0] = it.cost; r[1] = it.cost_change; r[2] = it.gradient_max_norm; r[3] = it.step_norm;
      r[4] = it.relative_decrease; r[5] = it.trust_region_radius; r[6] = it.step_is_valid; r[7] = it.step_is_successful;
    }
  }
  if (H0) std::memcpy(H0, s.H0, sizeof(double) * 36);
  if (g0) std::memcpy(g0, s.g0, sizeof(double) * 6);
  if (term) *term = s.termination;
  if (costs2) { costs2[0] = s.initial_cost; costs2[1] = s.final_cost; }
}
struct HelloFunctor {
  double c;
  template <class T> bool operator()(const T* x, const T* /*t*/, T* r) const { r[0] = T(c) - x[0]; return true; }
};
}  // namespace

extern "C" {

struct orc_handle {
  Pipeline* p;
};

orc_handle* orc_create(int scan_line, double minimum_range, float line_res, float plane_res, int mapping_skip_frame,
                       int detach_vo_lo, int with_mapping) {
  Config c;
  c.scan_line = scan_line; c.minimum_range = minimum_range;
  c.mapping_line_resolution = line_res; c.mapping_plane_resolution = plane_res;
  c.mapping_skip_frame = mapping_skip_frame; c.detach_VO_LO = detach_vo_lo != 0;
  orc_handle* h = new orc_handle;
  h->p = new Pipeline(c, with_mapping != 0);
  return h;
}
void orc_destroy(orc_handle* h) { if (h) { delete h->p; delete h; } }

int orc_process(orc_handle* h, const float* xyz_pad4, int n) { return h->p->process(xyz_pad4, n) ? 0 : -1; }
int orc_scan_registration(orc_handle* h, const float* xyz_pad4, int n) {
  return scan_registration(xyz_pad4, n, h->p->cfg, &h->p->sr) ? 0 : -1;
}
// ---- the façade stage by stage, with the hand-overs in the caller's reach (lidar_odometry_mapping.cpp:73-154 passes clouds and the pose from
// stage to stage BY VALUE: LaserOdometry::input / LaserMapping::input deep-copy whatever they are handed, laser_odometry.cpp:141-145,
// laser_mapping.cpp:172-181) — tests of vloam_set_odometry_input / vloam_set_mapping_input edit them in between
int orc_stage_sr(orc_handle* h, const float* xyz_pad4, int n) {
  h->p->lm.reset();
  return scan_registration(xyz_pad4, n, h->p->cfg, &h->p->sr) ? 0 : -1;
}
// which: 0 laserCloud, 1 cornerPointsSharp, 2 cornerPointsLessSharp, 3 surfPointsFlat, 4 surfPointsLessFlat (what LaserOdometry::input will copy)
int orc_set_sr_cloud(orc_handle* h, int which, const float* xyzi, int n) {
  ScanRegistrationResult& s = h->p->sr;
  Cloud* c[5] = {&s.laserCloud, &s.cornerPointsSharp, &s.cornerPointsLessSharp, &s.surfPointsFlat, &s.surfPointsLessFlat};
  if (which < 0 || which > 4 || n < 0) return -1;
  c[which]->resize((size_t)n);
  if (n) std::memcpy(c[which]->data(), xyzi, sizeof(PointXYZI) * (size_t)n);
  return 0;
}
int orc_stage_lo(orc_handle* h) { h->p->lo.input(h->p->sr); h->p->lo.solveLO(); return 0; }
// null cloud / pose = LaserOdometry::output's own
int orc_stage_map(orc_handle* h, const float* corner, int nc, const float* surf, int ns, const float* full, int nf, const double* q, const double* t) {
  Pipeline& p = *h->p;
  if (!p.do_mapping) return -1;
  Cloud c = p.lo.laserCloudCornerLast, s = p.lo.laserCloudSurfLast, f = p.lo.laserCloudFullRes;
  if (corner) { c.resize((size_t)nc); if (nc) std::memcpy(c.data(), corner, sizeof(PointXYZI) * (size_t)nc); }
  if (surf) { s.resize((size_t)ns); if (ns) std::memcpy(s.data(), surf, sizeof(PointXYZI) * (size_t)ns); }
  if (full) { f.resize((size_t)nf); if (nf) std::memcpy(f.data(), full, sizeof(PointXYZI) * (size_t)nf); }
  Quat<double> qq = p.lo.q_w_curr;
  V3<double> tt = p.lo.t_w_curr;
  if (q) { qq.x = q[0]; qq.y = q[1]; qq.z = q[2]; qq.w = q[3]; }
  if (t) { tt.x = t[0]; tt.y = t[1]; tt.z = t[2]; }
  p.lm.input(c, s, f, qq, tt, p.lo.skip_frame);
  if (!p.lo.skip_frame) p.lm.solveMapping();
  return 0;
}
void orc_set_vo_prior(orc_handle* h, const double* q, const double* t) { h->p->lo.set_vo_prior(q, t); }
void orc_stage_ms(orc_handle* h, double* ms3) { for (int i = 0; i < 3; i++) ms3[i] = h->p->stage_ms[i]; }

// which: 0 full, 1 sharp, 2 lessSharp, 3 flat, 4 lessFlat, 5 cornerLast, 6 surfLast,
//        7 map corner stack, 8 map surf stack, 9 cornerFromMap, 10 surfFromMap, 11 registered full-res
int orc_get_cloud(orc_handle* h, int which, float* buf, int cap) {
  Pipeline& p = *h->p;
  switch (which) {
    case 0: return copy_cloud(p.sr.laserCloud, buf, cap);
    case 1: return copy_cloud(p.sr.cornerPointsSharp, buf, cap);
    case 2: return copy_cloud(p.sr.cornerPointsLessSharp, buf, cap);
    case 3: return copy_cloud(p.sr.surfPointsFlat, buf, cap);
    case 4: return copy_cloud(p.sr.surfPointsLessFlat, buf, cap);
    case 5: return copy_cloud(p.lo.laserCloudCornerLast, buf, cap);
    case 6: return copy_cloud(p.lo.laserCloudSurfLast, buf, cap);
    case 7: return copy_cloud(p.lm.laserCloudCornerStack, buf, cap);
    case 8: return copy_cloud(p.lm.laserCloudSurfStack, buf, cap);
    case 9: return copy_cloud(p.lm.laserCloudCornerFromMap, buf, cap);
    case 10: return copy_cloud(p.lm.laserCloudSurfFromMap, buf, cap);
    case 11: { Cloud c; p.lm.registered_cloud(&c); return copy_cloud(c, buf, cap); }
  }
  return -1;
}
// which: 0 sortInd, 1 picked, 2 label, 3 scanStartInd, 4 scanEndInd, 5 sharpInd, 6 lessSharpInd, 7 flatInd
int orc_get_sr_ints(orc_handle* h, int which, int* buf, int cap) {
  ScanRegistrationResult& s = h->p->sr;
  switch (which) {
    case 0: return copy_vec(s.cloudSortInd, buf, cap);
    case 1: return copy_vec(s.cloudNeighborPicked, buf, cap);
    case 2: return copy_vec(s.cloudLabel, buf, cap);
    case 3: return copy_vec(s.scanStartInd, buf, cap);
    case 4: return copy_vec(s.scanEndInd, buf, cap);
    case 5: return copy_vec(s.sharpInd, buf, cap);
    case 6: return copy_vec(s.lessSharpInd, buf, cap);
    case 7: return copy_vec(s.flatInd, buf, cap);
  }
  return -1;
}
int orc_get_sr_curvature(orc_handle* h, float* buf, int cap) { return copy_vec(h->p->sr.cloudCurvature, buf, cap); }
void orc_get_sr_scalars(orc_handle* h, float* start_end_ori2, int* half_passed_at, int* n_after_s1) {
  start_end_ori2[0] = h->p->sr.startOri; start_end_ori2[1] = h->p->sr.endOri;
  *half_passed_at = h->p->sr.halfPassedAt; *n_after_s1 = h->p->sr.n_after_s1;
}

void orc_get_lo_pose(orc_handle* h, double* q_w, double* t_w, double* q_lc, double* t_lc) {
  LaserOdometry& lo = h->p->lo;
  q_w[0] = lo.q_w_curr.x; q_w[1] = lo.q_w_curr.y; q_w[2] = lo.q_w_curr.z; q_w[3] = lo.q_w_curr.w;
  t_w[0] = lo.t_w_curr.x; t_w[1] = lo.t_w_curr.y; t_w[2] = lo.t_w_curr.z;
  for (int i = 0; i < 4; i++) q_lc[i] = lo.para_q[i];
  for (int i = 0; i < 3; i++) t_lc[i] = lo.para_t[i];
}
int orc_lo_num_outer(orc_handle* h) { return (int)h->p->lo.debug.size(); }
// corner: n x 3 ints (i,a,b); plane: n x 4 ints (i,a,b,c)
int orc_get_lo_corr(orc_handle* h, int outer, int* corner, int cap_c, int* n_corner, int* plane, int cap_p, int* n_plane) {
  if (outer < 0 || outer >= (int)h->p->lo.debug.size()) return -1;
  const LOIterationDebug& d = h->p->lo.debug[outer];
  *n_corner = (int)d.corner.size();
  *n_plane = (int)d.plane.size();
  if (corner) for (int i = 0; i < *n_corner && i < cap_c; i++) { corner[3 * i] = d.corner[i].i; corner[3 * i + 1] = d.corner[i].a; corner[3 * i + 2] = d.corner[i].b; }
  if (plane) for (int i = 0; i < *n_plane && i < cap_p; i++) { plane[4 * i] = d.plane[i].i; plane[4 * i + 1] = d.plane[i].a; plane[4 * i + 2] = d.plane[i].b; plane[4 * i + 3] = d.plane[i].c; }
  return 0;
}
// trace rows: cost, cost_change, gradient_max_norm, step_norm, relative_decrease, radius, valid, successful
int orc_get_lo_solve(orc_handle* h, int outer, double* qt_in7, double* qt_out7, double* trace, int cap_iters, int* n_iters,
                     double* H0, double* g0, int* termination, double* costs2, double* residuals0, int cap_res, int* n_res) {
  if (outer < 0 || outer >= (int)h->p->lo.debug.size()) return -1;
  const LOIterationDebug& d = h->p->lo.debug[outer];
  for (int i = 0; i < 4; i++) { qt_in7[i] = d.q_in[i]; qt_out7[i] = d.q_out[i]; }
  for (int i = 0; i < 3; i++) { qt_in7[4 + i] = d.t_in[i]; qt_out7[4 + i] = d.t_out[i]; }
  pack_summary(d.summary, trace, cap_iters, n_iters, H0, g0, termination, costs2);
  if (n_res) *n_res = (int)d.summary.raw_residuals0.size();
  if (residuals0) copy_vec(d.summary.raw_residuals0, residuals0, cap_res);
  return 0;
}

void orc_get_map_pose(orc_handle* h, double* q_w, double* t_w, double* q_wmap_wodom, double* t_wmap_wodom) {
  LaserMapping& lm = h->p->lm;
  for (int i = 0; i < 4; i++) q_w[i] = lm.parameters[i];
  for (int i = 0; i < 3; i++) t_w[i] = lm.parameters[4 + i];
  q_wmap_wodom[0] = lm.q_wmap_wodom.x; q_wmap_wodom[1] = lm.q_wmap_wodom.y; q_wmap_wodom[2] = lm.q_wmap_wodom.z; q_wmap_wodom[3] = lm.q_wmap_wodom.w;
  t_wmap_wodom[0] = lm.t_wmap_wodom.x; t_wmap_wodom[1] = lm.t_wmap_wodom.y; t_wmap_wodom[2] = lm.t_wmap_wodom.z;
}
// the pose LaserMapping::publish hands to VloamTF::world_MOT_base_last (laser_mapping.cpp:718-757): q_w_curr after a mapped
// frame, the high-frequency pose after a skipped one
void orc_get_map_published_pose(orc_handle* h, double* q, double* t) {
  LaserMapping& lm = h->p->lm;
  if (lm.skip_frame) {
    q[0] = lm.q_w_curr_highfreq.x; q[1] = lm.q_w_curr_highfreq.y; q[2] = lm.q_w_curr_highfreq.z; q[3] = lm.q_w_curr_highfreq.w;
    t[0] = lm.t_w_curr_highfreq.x; t[1] = lm.t_w_curr_highfreq.y; t[2] = lm.t_w_curr_highfreq.z;
  } else {
    for (int i = 0; i < 4; i++) q[i] = lm.parameters[i];
    for (int i = 0; i < 3; i++) t[i] = lm.parameters[4 + i];
  }
}
int orc_map_num_outer(orc_handle* h) { return (int)h->p->lm.debug.size(); }
int orc_get_map_solve(orc_handle* h, int outer, double* qt_in7, double* qt_out7, double* trace, int cap_iters, int* n_iters,
                      double* H0, double* g0, int* termination, double* costs2, int* corner_surf_num2, double* residuals0,
                      int cap_res, int* n_res) {
  if (outer < 0 || outer >= (int)h->p->lm.debug.size()) return -1;
  const MapIterationDebug& d = h->p->lm.debug[outer];
  for (int i = 0; i < 4; i++) { qt_in7[i] = d.q_in[i]; qt_out7[i] = d.q_out[i]; }
  for (int i = 0; i < 3; i++) { qt_in7[4 + i] = d.t_in[i]; qt_out7[4 + i] = d.t_out[i]; }
  pack_summary(d.summary, trace, cap_iters, n_iters, H0, g0, termination, costs2);
  corner_surf_num2[0] = d.corner_num; corner_surf_num2[1] = d.surf_num;
  if (n_res) *n_res = (int)d.summary.raw_residuals0.size();
  if (residuals0) copy_vec(d.summary.raw_residuals0, residuals0, cap_res);
  return 0;
}
// accepted map factors of outer round `outer`: stack indices + geometry (corner: a,b = 6 doubles; surf: n,d = 4)
int orc_get_map_factors(orc_handle* h, int outer, int* corner_idx, double* corner_ab, int cap_c, int* surf_idx, double* surf_plane, int cap_s) {
  if (outer < 0 || outer >= (int)h->p->lm.debug.size()) return -1;
  const MapIterationDebug& d = h->p->lm.debug[outer];
  for (int i = 0; i < (int)d.corner_idx.size() && i < cap_c; i++) {
    if (corner_idx) corner_idx[i] = d.corner_idx[i];
    if (corner_ab) for (int k = 0; k < 6; k++) corner_ab[6 * i + k] = d.corner_ab[i][k];
  }
  for (int i = 0; i < (int)d.surf_idx.size() && i < cap_s; i++) {
    if (surf_idx) surf_idx[i] = d.surf_idx[i];
    if (surf_plane) for (int k = 0; k < 4; k++) surf_plane[4 * i + k] = d.surf_plane[i][k];
  }
  return 0;
}
// map bookkeeping: centre offsets, total points, valid cube list, per-cube sizes
void orc_get_map_info(orc_handle* h, int* cen3, long long* total2, int* valid_ind, int cap_valid, int* n_valid) {
  LaserMapping& lm = h->p->lm;
  cen3[0] = lm.cenW(); cen3[1] = lm.cenH(); cen3[2] = lm.cenD();
  total2[0] = (long long)lm.map_points_corner(); total2[1] = (long long)lm.map_points_surf();
  *n_valid = (int)lm.validInd.size();
  for (int i = 0; i < *n_valid && i < cap_valid; i++) valid_ind[i] = lm.validInd[i];
}
int orc_get_map_cube(orc_handle* h, int which /*0 corner,1 surf*/, int cube, float* buf, int cap) {
  if (cube < 0 || cube >= LaserMapping::laserCloudNum) return -1;
  return copy_cloud(which == 0 ? h->p->lm.cube_corner(cube) : h->p->lm.cube_surf(cube), buf, cap);
}

// ---------------------------------------------------------------- standalone pieces
int orc_voxel_grid(const float* xyzi, int n, float leaf, float* out, int cap) {
  Cloud in((size_t)n);
  if (n > 0) std::memcpy(in.data(), xyzi, sizeof(PointXYZI) * (size_t)n);
  Cloud o = voxel_grid(in, leaf);
  return copy_cloud(o, out, cap);
}
void orc_knn(const float* xyzi, int n, const float* queries_xyz, int nq, int k, int* idx, float* d2, int use_tree) {
  Cloud pts((size_t)n);
  if (n > 0) std::memcpy(pts.data(), xyzi, sizeof(PointXYZI) * (size_t)n);
  KdTree tree;
  if (use_tree) tree.build(pts);
  for (int i = 0; i < nq; i++) {
    for (int j = 0; j < k; j++) { idx[i * k + j] = -1; d2[i * k + j] = -1.0f; }
    if (use_tree) tree.knn(queries_xyz + 3 * i, k, idx + i * k, d2 + i * k);
    else knn_brute(pts, queries_xyz + 3 * i, k, idx + i * k, d2 + i * k);
  }
}

// type: 0 LidarEdgeFactor (geom = a[3], b[3]), 1 LidarPlaneFactor (geom = j,l,m [9]), 2 LidarPlaneNormFactor (geom = n[3], d).
// Returns nres; residual[nres]; jac_local = nres x 6 row-major in the tangent space of (q ⊞ δ, t + δt)
// exactly as Ceres assembles it (autodiff global Jacobian x EigenQuaternionParameterization Jacobian); no loss applied.
int orc_eval_lidar_factor_s(int type, const double* curr3, const double* geom, const double* q4, const double* t3, double s_ratio, double* residual,
                            double* jac_local);
int orc_eval_lidar_factor(int type, const double* curr3, const double* geom, const double* q4, const double* t3, double* residual,
                          double* jac_local) {
  return orc_eval_lidar_factor_s(type, curr3, geom, q4, t3, 1.0, residual, jac_local);
}
// s_ratio: the functors' interpolation ratio s (lidarFactor.hpp:26-33: q slerped from identity by s, t scaled by s); 1.0 when DISTORTION == false
int orc_eval_lidar_factor_s(int type, const double* curr3, const double* geom, const double* q4, const double* t3, double s_ratio, double* residual,
                            double* jac_local) {
  std::unique_ptr<CostFunction> f;
  Vec3d c(curr3[0], curr3[1], curr3[2]);
  if (type == 0) f.reset(new AutoDiffCost<LidarEdgeFactor, 3, 4>(LidarEdgeFactor(c, Vec3d(geom[0], geom[1], geom[2]), Vec3d(geom[3], geom[4], geom[5]), s_ratio)));
  else if (type == 1) f.reset(new AutoDiffCost<LidarPlaneFactor, 1, 4>(LidarPlaneFactor(c, Vec3d(geom[0], geom[1], geom[2]), Vec3d(geom[3], geom[4], geom[5]), Vec3d(geom[6], geom[7], geom[8]), s_ratio)));
  else if (type == 2) f.reset(new AutoDiffCost<LidarPlaneNormFactor, 1, 4>(LidarPlaneNormFactor(c, Vec3d(geom[0], geom[1], geom[2]), geom[3])));
  else return -1;
  double j0[12], j1[9];
  f->Evaluate(q4, t3, residual, j0, j1);
  const double* x = q4;
  const double P[12] = {x[3], x[2], -x[1], -x[2], x[3], x[0], x[1], -x[0], x[3], -x[0], -x[1], -x[2]};
  for (int k = 0; k < f->nres; k++) {
    for (int a = 0; a < 3; a++) {
      double s = 0; for (int b = 0; b < 4; b++) s += j0[k * 4 + b] * P[b * 3 + a];
      jac_local[k * 6 + a] = s;
      jac_local[k * 6 + 3 + a] = j1[k * 3 + a];
    }
  }
  return f->nres;
}

// Generic driver for the LM restatement.  Factors: rows of 16 doubles {type, payload...}:
//   type 0 edge      : curr[3], a[3], b[3]
//   type 1 plane     : curr[3], j[3], l[3], m[3]
//   type 2 planenorm : curr[3], n[3], d
//   type 3 CostFunctor32 : x0,y0,z0,x1_bar,y1_bar
//   type 4 CostFunctor22 : x0_bar,y0_bar,x1_bar,y1_bar
//   type 5 scalar test   : r = payload[0] - p0[0]     (Ceres "hello world", VO/README.md:40-50)
// p0 has 4 entries (quaternion xyzw) when quaternion != 0, else 3.
int orc_solve(const double* factors, int nf, int quaternion, double huber_a, int max_iters, double* p0, double* p1, double* trace,
              int cap_iters, int* n_iters, double* H0, double* g0, int* termination, double* costs2) {
  Problem prob;
  for (int i = 0; i < nf; i++) {
    const double* f = factors + 16 * i;
    const int type = (int)f[0];
    const double* p = f + 1;
    Vec3d c(p[0], p[1], p[2]);
    if (quaternion) {
      if (type == 0) prob.Add(new AutoDiffCost<LidarEdgeFactor, 3, 4>(LidarEdgeFactor(c, Vec3d(p[3], p[4], p[5]), Vec3d(p[6], p[7], p[8]), 1.0)));
      else if (type == 1) prob.Add(new AutoDiffCost<LidarPlaneFactor, 1, 4>(LidarPlaneFactor(c, Vec3d(p[3], p[4], p[5]), Vec3d(p[6], p[7], p[8]), Vec3d(p[9], p[10], p[11]), 1.0)));
      else if (type == 2) prob.Add(new AutoDiffCost<LidarPlaneNormFactor, 1, 4>(LidarPlaneNormFactor(c, Vec3d(p[3], p[4], p[5]), p[6])));
      else return -1;
    } else {
      if (type == 3) prob.Add(new AutoDiffCost<CostFunctor32, 2, 3>(CostFunctor32(p[0], p[1], p[2], p[3], p[4])));
      else if (type == 4) prob.Add(new AutoDiffCost<CostFunctor22, 1, 3>(CostFunctor22(p[0], p[1], p[2], p[3])));
      else if (type == 5) prob.Add(new AutoDiffCost<HelloFunctor, 1, 3>(HelloFunctor{p[0]}));
      else return -1;
    }
  }
  SolveOptions opt;
  opt.max_num_iterations = max_iters;
  opt.huber_a = huber_a;
  opt.quaternion_block0 = quaternion != 0;
  SolveSummary s;
  prob.Solve(opt, p0, p1, &s);
  pack_summary(s, trace, cap_iters, n_iters, H0, g0, termination, costs2);
  return 0;
}

// ---- image front-end (orc_img.h)
int orc_good_features(const unsigned char* img, int w, int h, int max_corners, double quality, double min_distance, int block_size, float* xy_out,
                      int cap, float* eig_out) {
  std::vector<float> eig;
  std::vector<ImgCorner> c = good_features_to_track(img, w, h, max_corners, quality, min_distance, block_size, eig_out ? &eig : nullptr);
  if (eig_out) std::copy(eig.begin(), eig.end(), eig_out);
  const int n = (int)c.size();
  for (int i = 0; i < n && i < cap; i++) { xy_out[2 * i] = c[i].x; xy_out[2 * i + 1] = c[i].y; }
  return n;
}
// level < 0: only the number of levels.  img_out [w_l * h_l], deriv_out [2 * w_l * h_l], wh [2]
int orc_pyramid_level(const unsigned char* img, int w, int h, int win, int max_level, int level, unsigned char* img_out, short* deriv_out, int* wh) {
  Pyramid P;
  P.build(img, w, h, win, max_level);
  if (level < 0) return P.levels();
  if (level >= P.levels()) return -1;
  std::copy(P.img[level].begin(), P.img[level].end(), img_out);
  std::copy(P.deriv[level].begin(), P.deriv[level].end(), deriv_out);
  wh[0] = P.w[level]; wh[1] = P.h[level];
  return P.levels();
}
int orc_pyr_lk(const unsigned char* prev, const unsigned char* next, int w, int h, const float* pts, int n, float* next_pts, unsigned char* status, int win,
               int max_level, int max_count, double epsilon) {
  Pyramid P, N;
  P.build(prev, w, h, win, max_level);
  N.build(next, w, h, win, max_level);
  std::vector<ImgCorner> a((size_t)n), b;
  for (int i = 0; i < n; i++) a[i] = ImgCorner{pts[2 * i], pts[2 * i + 1]};
  std::vector<uint8_t> st;
  calc_optical_flow_pyr_lk(P, N, a, &b, &st, win, max_count, epsilon);
  for (int i = 0; i < n; i++) { next_pts[2 * i] = b[i].x; next_pts[2 * i + 1] = b[i].y; status[i] = st[i]; }
  return 0;
}

// (queryIdx, trainIdx) pairs of ImageUtil::matchDescriptors (BF, NORM_HAMMING); returns the number of matches
int orc_bf_match_hamming(const unsigned char* d0, int n0, const unsigned char* d1, int n1, int bytes, int knn, int* q_out, int* t_out, int cap) {
  std::vector<std::pair<int, int>> m = bf_match_hamming(d0, n0, d1, n1, bytes, knn != 0);
  for (size_t i = 0; i < m.size() && (int)i < cap; i++) { q_out[i] = m[i].first; t_out[i] = m[i].second; }
  return (int)m.size();
}

int orc_gaussian_blur(const unsigned char* img, int w, int h, unsigned char* out) { gaussian_blur_7x7_s2(img, w, h, out); return 0; }
// kept_idx [cap], desc [cap][32]; returns the number of keypoints that survive the border filter
int orc_orb_descriptors(const unsigned char* img, int w, int h, const float* kps_xy, int n, const signed char* pattern, int* kept_idx, unsigned char* desc, int cap) {
  std::vector<ImgCorner> kps((size_t)n);
  for (int i = 0; i < n; i++) { kps[(size_t)i].x = kps_xy[2 * i]; kps[(size_t)i].y = kps_xy[2 * i + 1]; }
  std::vector<int> kept;
  std::vector<uint8_t> d;
  orb_descriptors(img, w, h, kps, pattern, &kept, &d);
  const int m = (int)kept.size() < cap ? (int)kept.size() : cap;
  for (int i = 0; i < m; i++) kept_idx[i] = kept[(size_t)i];
  if (m > 0) std::memcpy(desc, d.data(), (size_t)m * 32);
  return (int)kept.size();
}
int orc_clahe(const unsigned char* img, int w, int h, double clip_limit, int tiles, unsigned char* out) {
  clahe_apply(img, w, h, clip_limit, tiles, out);
  return 0;
}

}  // extern "C"
