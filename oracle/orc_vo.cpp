// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  PARITY UNPINNED.
// Line references are into /root/reference/src/visual_odometry/.
#include "orc_vo.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace orc {

// src/point_cloud_util.cpp:148-174.  Eigen evaluates tilde * A^T * B^T * C^T left to right as three
// f32 GEMMs whose K=4 inner products accumulate in k order.
void DepthMap::projectPointCloud(const float* xyz, int n, const VOCalib& c) {
  point_cloud_2d.clear();
  point_cloud_2d.reserve((size_t)n * 3);
  for (int i = 0; i < n; i++) {
    const float t[4] = {xyz[4 * i], xyz[4 * i + 1], xyz[4 * i + 2], 1.0f};
    float a[4], b[4], p[3];
    for (int r = 0; r < 4; r++) a[r] = ((t[0] * c.cam_T_velo[r * 4 + 0] + t[1] * c.cam_T_velo[r * 4 + 1]) + t[2] * c.cam_T_velo[r * 4 + 2]) + t[3] * c.cam_T_velo[r * 4 + 3];
    for (int r = 0; r < 4; r++) b[r] = ((a[0] * c.rect0_T_cam[r * 4 + 0] + a[1] * c.rect0_T_cam[r * 4 + 1]) + a[2] * c.rect0_T_cam[r * 4 + 2]) + a[3] * c.rect0_T_cam[r * 4 + 3];
    for (int r = 0; r < 3; r++) p[r] = ((b[0] * c.P_rect0[r * 4 + 0] + b[1] * c.P_rect0[r * 4 + 1]) + b[2] * c.P_rect0[r * 4 + 2]) + b[3] * c.P_rect0[r * 4 + 3];
    if (!(p[2] > 0.1f)) continue;  // :156-158 (Eigen compares the f32 array with Scalar(0.1); NaN fails the test and is dropped)
    const float inv = 1.0f / p[2];  // Eigen::inverse(col(2).array()) then a product, :171-173
    point_cloud_2d.push_back(p[0] * inv);
    point_cloud_2d.push_back(p[1] * inv);
    point_cloud_2d.push_back(p[2]);
  }
}

// src/point_cloud_util.cpp:205-260
void DepthMap::downsamplePointCloud() {
  new_width = (int)std::ceil(static_cast<float>(IMG_WIDTH) / static_cast<float>(downsample_grid_size));
  new_height = (int)std::ceil(static_cast<float>(IMG_HEIGHT) / static_cast<float>(downsample_grid_size));
  const size_t nb = (size_t)new_width * new_height;
  bucket_x.assign(nb, 0.0f); bucket_y.assign(nb, 0.0f); bucket_depth.assign(nb, 0.0f); bucket_count.assign(nb, 0);
  const int rows = (int)(point_cloud_2d.size() / 3);
  for (int i = 0; i < rows; ++i) {
    const float u = point_cloud_2d[3 * i], v = point_cloud_2d[3 * i + 1], d = point_cloud_2d[3 * i + 2];
    int index_x = static_cast<int>(u / downsample_grid_size);
    int index_y = static_cast<int>(v / downsample_grid_size);
    if (index_x >= 0 && index_x < new_width && index_y >= 0 && index_y < new_height) {
      const size_t b = (size_t)index_x * new_height + index_y;
      if (bucket_count[b] == 0) {
        bucket_x[b] = u; bucket_y[b] = v; bucket_depth[b] = d;
      } else {  // "incremental averaging" with the count BEFORE the increment — reproduced verbatim (:230-235)
        bucket_x[b] += (u - bucket_x[b]) / bucket_count[b];
        bucket_y[b] += (v - bucket_y[b]) / bucket_count[b];
        bucket_depth[b] += (d - bucket_depth[b]) / bucket_count[b];
      }
      ++bucket_count[b];
    }
  }
}

// src/point_cloud_util.cpp:302-387
float DepthMap::queryDepth(float x, float y, int searching_radius) const {
  int index_x = static_cast<int>(x / downsample_grid_size);
  int index_y = static_cast<int>(y / downsample_grid_size);
  struct Nb { float x, y, d, dist; };
  Nb nbs[64];
  int cnt = 0;
  for (int ix = index_x - searching_radius; ix <= index_x + searching_radius; ++ix) {
    for (int iy = index_y - searching_radius; iy <= index_y + searching_radius; ++iy) {
      if (ix >= 0 && ix < new_width && iy >= 0 && iy < new_height && bucket_count[(size_t)ix * new_height + iy] > 0) {
        const size_t b = (size_t)ix * new_height + iy;
        Nb nb;
        nb.x = bucket_x[b]; nb.y = bucket_y[b]; nb.d = bucket_depth[b];
        nb.dist = std::sqrt(std::pow(x - nb.x, 2) + std::pow(y - nb.y, 2));  // std::pow(float,int) -> double
        if (cnt < 64) nbs[cnt++] = nb;
      }
    }
  }
  if (cnt < 10) return -1.0f;
  // std::sort in the reference (unstable); canonical tie order = scan order
  std::stable_sort(nbs, nbs + cnt, [](const Nb& a, const Nb& b) { return a.dist < b.dist; });
  float z = (nbs[0].d * nbs[1].dist * nbs[2].dist + nbs[1].d * nbs[0].dist * nbs[2].dist + nbs[2].d * nbs[0].dist * nbs[1].dist) /
            (0.0001f + nbs[1].dist * nbs[2].dist + nbs[0].dist * nbs[2].dist + nbs[0].dist * nbs[1].dist);
  return z;
}

// Column-pivoted Householder QR in f32 (algorithm family of Eigen's ColPivHouseholderQR; column
// norms are recomputed per step instead of down-dated).  Device code mirrors this op for op.
void solve3x3_colpiv_qr_f32(const float A_[9], const float b_[3], float x[3]) {
  float A[3][3], b[3] = {b_[0], b_[1], b_[2]};
  int perm[3] = {0, 1, 2};
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) A[r][c] = A_[r * 3 + c];
  for (int k = 0; k < 3; k++) {
    int best = k; float bestn = -1.0f;
    for (int c = k; c < 3; c++) {
      float s = 0.0f;
      for (int r = k; r < 3; r++) s += A[r][c] * A[r][c];
      if (s > bestn) { bestn = s; best = c; }
    }
    if (best != k) {
      for (int r = 0; r < 3; r++) std::swap(A[r][k], A[r][best]);
      std::swap(perm[k], perm[best]);
    }
    float nrm = std::sqrt(bestn);
    if (nrm == 0.0f) continue;
    float alpha = A[k][k] > 0.0f ? -nrm : nrm;
    float v[3] = {0, 0, 0};
    v[k] = A[k][k] - alpha;
    float vtv = v[k] * v[k];
    for (int r = k + 1; r < 3; r++) { v[r] = A[r][k]; vtv += v[r] * v[r]; }
    if (vtv != 0.0f) {
      for (int c = k + 1; c < 3; c++) {
        float s = 0.0f;
        for (int r = k; r < 3; r++) s += v[r] * A[r][c];
        s = 2.0f * s / vtv;
        for (int r = k; r < 3; r++) A[r][c] -= s * v[r];
      }
      float s = 0.0f;
      for (int r = k; r < 3; r++) s += v[r] * b[r];
      s = 2.0f * s / vtv;
      for (int r = k; r < 3; r++) b[r] -= s * v[r];
    }
    A[k][k] = alpha;
    for (int r = k + 1; r < 3; r++) A[r][k] = 0.0f;
  }
  float y[3];
  for (int k = 2; k >= 0; k--) {
    float s = b[k];
    for (int c = k + 1; c < 3; c++) s -= A[k][c] * y[c];
    y[k] = s / A[k][k];
  }
  for (int k = 0; k < 3; k++) x[perm[k]] = y[k];
}

// src/visual_odometry.cpp:254-450
void VisualOdometry::solveNlsAll(const int* prev_uv, const int* curr_uv, int n_match, const double* init_angles, const double* init_t) {
  Problem problem;
  if (!init_angles || !init_t) {
    for (int j = 0; j < 3; ++j) { angles_0to1[j] = 0.0; t_0to1[j] = 0.0; }
  } else {
    for (int j = 0; j < 3; ++j) { angles_0to1[j] = init_angles[j]; t_0to1[j] = init_t[j]; }
  }
  counter32 = counter22 = 0;
  match_debug.clear();
  float K[9];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) K[r * 3 + c] = calib.P_rect0[r * 4 + c];  // leftCols(3)
  for (int j = 0; j < n_match; ++j) {
    const int prev_pt_x = prev_uv[2 * j], prev_pt_y = prev_uv[2 * j + 1];
    const int curr_pt_x = curr_uv[2 * j], curr_pt_y = curr_uv[2 * j + 1];
    VOMatchDebug md;
    md.kind = 0; md.depth0 = 0; for (double& o : md.obs) o = 0;
    if (remove_VO_outlier > 0) {
      if (std::pow(prev_pt_x - curr_pt_x, 2) + std::pow(prev_pt_y - curr_pt_y, 2) > remove_VO_outlier * remove_VO_outlier) {
        match_debug.push_back(md);
        continue;
      }
    }
    const float depth0 = maps[1 - i].queryDepth((float)prev_pt_x, (float)prev_pt_y);
    md.depth0 = depth0;
    float p0[3], p1[3], r0[3], r1[3];
    if (depth0 > 0) {
      p0[0] = prev_pt_x * depth0; p0[1] = prev_pt_y * depth0; p0[2] = depth0;
      p1[0] = (float)curr_pt_x; p1[1] = (float)curr_pt_y; p1[2] = 1.0f;
      solve3x3_colpiv_qr_f32(K, p0, r0);
      solve3x3_colpiv_qr_f32(K, p1, r1);
      md.kind = 32;
      md.obs[0] = r0[0]; md.obs[1] = r0[1]; md.obs[2] = r0[2];
      md.obs[3] = static_cast<double>(r1[0]) / static_cast<double>(r1[2]);
      md.obs[4] = static_cast<double>(r1[1]) / static_cast<double>(r1[2]);
      problem.Add(new AutoDiffCost<CostFunctor32, 2, 3>(CostFunctor32(md.obs[0], md.obs[1], md.obs[2], md.obs[3], md.obs[4])));
      ++counter32;
    } else {
      p0[0] = (float)prev_pt_x; p0[1] = (float)prev_pt_y; p0[2] = 1.0f;
      p1[0] = (float)curr_pt_x; p1[1] = (float)curr_pt_y; p1[2] = 1.0f;
      solve3x3_colpiv_qr_f32(K, p0, r0);
      solve3x3_colpiv_qr_f32(K, p1, r1);
      md.kind = 22;
      md.obs[0] = static_cast<double>(r0[0]) / static_cast<double>(r0[2]);
      md.obs[1] = static_cast<double>(r0[1]) / static_cast<double>(r0[2]);
      md.obs[2] = static_cast<double>(r1[0]) / static_cast<double>(r1[2]);
      md.obs[3] = static_cast<double>(r1[1]) / static_cast<double>(r1[2]);
      problem.Add(new AutoDiffCost<CostFunctor22, 1, 3>(CostFunctor22(md.obs[0], md.obs[1], md.obs[2], md.obs[3])));
      ++counter22;
    }
    match_debug.push_back(md);
  }
  SolveOptions options;  // visual_odometry.cpp:67-68
  options.max_num_iterations = 100;
  options.huber_a = 0.1;
  options.quaternion_block0 = false;
  problem.Solve(options, angles_0to1, t_0to1, &summary);
}

}  // namespace orc

// ---------------------------------------------------------------- C entry points (ctypes)
using namespace orc;
extern "C" {
struct orc_vo_handle { VisualOdometry vo; };
orc_vo_handle* orc_vo_create(const float* cam_T_velo16, const float* rect0_T_cam16, const float* P_rect0_12, int remove_outlier) {
  orc_vo_handle* h = new orc_vo_handle;
  std::memcpy(h->vo.calib.cam_T_velo, cam_T_velo16, sizeof(float) * 16);
  std::memcpy(h->vo.calib.rect0_T_cam, rect0_T_cam16, sizeof(float) * 16);
  std::memcpy(h->vo.calib.P_rect0, P_rect0_12, sizeof(float) * 12);
  h->vo.remove_VO_outlier = remove_outlier;
  return h;
}
void orc_vo_destroy(orc_vo_handle* h) { delete h; }
void orc_vo_reset(orc_vo_handle* h) { h->vo.reset(); }
void orc_vo_process_point_cloud(orc_vo_handle* h, const float* xyz_pad4, int n) { h->vo.processPointCloud(xyz_pad4, n); }
// which_map: 0 = current (i), 1 = previous (1-i)
int orc_vo_get_buckets(orc_vo_handle* h, int which_map, float* bx, float* by, float* bd, int* bc, int cap) {
  const DepthMap& m = h->vo.maps[which_map == 0 ? h->vo.i : 1 - h->vo.i];
  const int nb = m.new_width * m.new_height;
  for (int k = 0; k < nb && k < cap; k++) { bx[k] = m.bucket_x[k]; by[k] = m.bucket_y[k]; bd[k] = m.bucket_depth[k]; bc[k] = m.bucket_count[k]; }
  return nb;
}
int orc_vo_get_points2d(orc_vo_handle* h, int which_map, float* uvd, int cap_rows) {
  const DepthMap& m = h->vo.maps[which_map == 0 ? h->vo.i : 1 - h->vo.i];
  const int rows = (int)(m.point_cloud_2d.size() / 3);
  if (uvd) std::memcpy(uvd, m.point_cloud_2d.data(), sizeof(float) * 3 * (size_t)std::min(rows, cap_rows));
  return rows;
}
float orc_vo_query_depth(orc_vo_handle* h, int which_map, float x, float y) {
  return h->vo.maps[which_map == 0 ? h->vo.i : 1 - h->vo.i].queryDepth(x, y);
}
// init7 = angles[3], t[3], use_init flag.  Outputs angles/t; counters {32, 22}; trace as in orc_get_lo_solve.
void orc_vo_solve(orc_vo_handle* h, const int* prev_uv, const int* curr_uv, int n_match, const double* init_angles, const double* init_t,
                  double* angles3, double* t3, int* counters2, double* trace, int cap_iters, int* n_iters, double* H0, double* g0,
                  int* termination, double* costs2) {
  h->vo.solveNlsAll(prev_uv, curr_uv, n_match, init_angles, init_t);
  for (int k = 0; k < 3; k++) { angles3[k] = h->vo.angles_0to1[k]; t3[k] = h->vo.t_0to1[k]; }
  counters2[0] = h->vo.counter32; counters2[1] = h->vo.counter22;
  const SolveSummary& s = h->vo.summary;
  int n = (int)s.iterations.size();
  if (n_iters) *n_iters = n;
  if (trace) for (int i = 0; i < n && i < cap_iters; i++) {
    const IterationSummary& it = s.iterations[i];
    double* r = trace + 8 * i;
    r[0] = it.cost; r[1] = it.cost_change; r[2] = it.gradient_max_norm; r[3] = it.step_norm;
    r[4] = it.relative_decrease; r[5] = it.trust_region_radius; r[6] = it.step_is_valid; r[7] = it.step_is_successful;
  }
  if (H0) std::memcpy(H0, s.H0, sizeof(double) * 36);
  if (g0) std::memcpy(g0, s.g0, sizeof(double) * 6);
  if (termination) *termination = s.termination;
  if (costs2) { costs2[0] = s.initial_cost; costs2[1] = s.final_cost; }
}
// per-match debug rows: kind, depth0, obs[5]  (7 doubles)
int orc_vo_get_match_debug(orc_vo_handle* h, double* rows7, int cap) {
  const auto& md = h->vo.match_debug;
  for (int k = 0; k < (int)md.size() && k < cap; k++) {
    rows7[7 * k] = md[k].kind; rows7[7 * k + 1] = md[k].depth0;
    for (int j = 0; j < 5; j++) rows7[7 * k + 2 + j] = md[k].obs[j];
  }
  return (int)md.size();
}
}
