// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  PARITY UNPINNED.
//
// CPU restatement of the reference's per-scan LiDAR odometry hot path:
//   ScanRegistration::input   src/lidar_odometry_mapping/src/scan_registration.cpp:131-449
//   LaserOdometry::solveLO    src/lidar_odometry_mapping/src/laser_odometry.cpp:187-536
//   LaserMapping::solveMapping src/lidar_odometry_mapping/src/laser_mapping.cpp:198-708
//   façade call order          src/lidar_odometry_mapping/src/lidar_odometry_mapping.cpp:65-154
#pragma once
#include <array>
#include <vector>
#include "orc_ceres.h"
#include "orc_factors.h"
#include "orc_pcl.h"

namespace orc {

struct Config {
  // LOM/launch/loam_velodyne_HDL_64_kitti.launch:3-16, MAIN/launch/vloam_main.launch:4
  int scan_line = 64;
  double minimum_range = 5.0;
  int mapping_skip_frame = 1;
  float mapping_line_resolution = 0.4f;
  float mapping_plane_resolution = 0.8f;
  bool detach_VO_LO = true;
};

// ------------------------------------------------------------------ ScanRegistration
struct ScanRegistrationResult {
  Cloud laserCloud, cornerPointsSharp, cornerPointsLessSharp, surfPointsFlat, surfPointsLessFlat;
  // debug / parity hooks
  std::vector<float> cloudCurvature;
  std::vector<int> cloudSortInd, cloudNeighborPicked, cloudLabel;
  std::vector<int> scanStartInd, scanEndInd;
  std::vector<int> sharpInd, lessSharpInd, flatInd;  // indices into laserCloud, in emission order
  float startOri = 0, endOri = 0;
  int halfPassedAt = -1;  // index (after S1 compaction) of the point that flipped halfPassed, -1 if none
  int n_after_s1 = 0;
};
// Returns false when no point survives NaN / minimum-range removal (the reference would read
// points[0] of an empty cloud, scan_registration.cpp:166 — undefined behaviour there).
bool scan_registration(const float* xyz_pad4, int n, const Config& cfg, ScanRegistrationResult* out);

// ------------------------------------------------------------------ LaserOdometry
struct CornerCorr { int i, a, b; };
struct PlaneCorr { int i, a, b, c; };
struct LOIterationDebug {
  std::vector<CornerCorr> corner;
  std::vector<PlaneCorr> plane;
  double q_in[4], t_in[3];  // parameters the data association ran with
  SolveSummary summary;
  double q_out[4], t_out[3];
};

class LaserOdometry {
 public:
  explicit LaserOdometry(const Config& c) : cfg(c) { reset_all(); }
  void reset_all();
  void set_vo_prior(const double q_xyzw[4], const double t[3]);  // vloam_tf->velo_last_VOT_velo_curr
  void input(const ScanRegistrationResult& sr);                  // laser_odometry.cpp:135-146
  void solveLO();                                                // laser_odometry.cpp:187-536
  // outputs (laser_odometry.cpp:610-629)
  Quat<double> q_w_curr;
  V3<double> t_w_curr;
  double para_q[4], para_t[3];  // q_last_curr (x,y,z,w), t_last_curr
  Cloud laserCloudCornerLast, laserCloudSurfLast, laserCloudFullRes;
  bool skip_frame = false;
  int frameCount = 0;
  int corner_correspondence = 0, plane_correspondence = 0;
  std::vector<LOIterationDebug> debug;  // one entry per outer iteration of the last solveLO

  void TransformToStart(const PointXYZI& pi, PointXYZI* po) const;  // laser_odometry.cpp:149-167

 private:
  Config cfg;
  bool systemInited = false;
  Cloud cornerPointsSharp, cornerPointsLessSharp, surfPointsFlat, surfPointsLessFlat;
  KdTree kdtreeCornerLast, kdtreeSurfLast;
  double prior_q[4], prior_t[3];
};

// ------------------------------------------------------------------ LaserMapping
struct MapIterationDebug {
  int corner_num = 0, surf_num = 0;
  double q_in[4], t_in[3];
  SolveSummary summary;
  double q_out[4], t_out[3];
  // per accepted factor: index into the down-sampled stack + geometry, for parity checks
  std::vector<int> corner_idx, surf_idx;
  std::vector<std::array<double, 6>> corner_ab;   // point_a, point_b
  std::vector<std::array<double, 4>> surf_plane;  // unit normal, negative_OA_dot_norm
};

class LaserMapping {
 public:
  explicit LaserMapping(const Config& c);
  void reset() {}  // laser_mapping.cpp:127-131 (valid-cube counters are locals here)
  void input(const Cloud& cornerLast, const Cloud& surfLast, const Cloud& fullRes, const Quat<double>& q_wodom_curr,
             const V3<double>& t_wodom_curr, bool skip_frame);  // laser_mapping.cpp:167-196
  void solveMapping();                                          // laser_mapping.cpp:198-708
  // state / outputs
  double parameters[7];  // q_w_curr (x,y,z,w), t_w_curr
  Quat<double> q_w_curr_highfreq{0, 0, 0, 1};  // laser_mapping.cpp:186-190 (what publish() reports on a skipped frame, :743-757)
  V3<double> t_w_curr_highfreq{0, 0, 0};
  bool skip_frame = false;
  Quat<double> q_wmap_wodom, q_wodom_curr;
  V3<double> t_wmap_wodom, t_wodom_curr;
  int frameCount = 0;
  // debug
  Cloud laserCloudCornerStack, laserCloudSurfStack, laserCloudCornerFromMap, laserCloudSurfFromMap;
  std::vector<MapIterationDebug> debug;
  std::vector<int> validInd;
  size_t map_points_corner() const;
  size_t map_points_surf() const;
  const Cloud& cube_corner(int ind) const { return laserCloudCornerArray[ind]; }
  const Cloud& cube_surf(int ind) const { return laserCloudSurfArray[ind]; }
  int cenW() const { return laserCloudCenWidth; }
  int cenH() const { return laserCloudCenHeight; }
  int cenD() const { return laserCloudCenDepth; }
  void registered_cloud(Cloud* out) const;  // publish(): fullRes -> map frame (laser_mapping.cpp:795-799)

  static const int laserCloudWidth = 21, laserCloudHeight = 21, laserCloudDepth = 11;
  static const int laserCloudNum = laserCloudWidth * laserCloudHeight * laserCloudDepth;

 private:
  Config cfg;
  int laserCloudCenWidth = 10, laserCloudCenHeight = 10, laserCloudCenDepth = 5;
  std::vector<Cloud> laserCloudCornerArray, laserCloudSurfArray;
  Cloud laserCloudCornerLast, laserCloudSurfLast, laserCloudFullRes;
  void pointAssociateToMap(const PointXYZI& pi, PointXYZI* po) const;  // laser_mapping.cpp:146-155
  void transformUpdate();                                               // laser_mapping.cpp:140-144
  template <class Shift> void roll(Shift);
};

// ------------------------------------------------------------------ façade
// LidarOdometryMapping: reset -> scanRegistrationIO -> laserOdometryIO -> laserMappingIO
class Pipeline {
 public:
  explicit Pipeline(const Config& c, bool with_mapping) : cfg(c), lo(c), lm(c), do_mapping(with_mapping) {}
  bool process(const float* xyz_pad4, int n);
  Config cfg;
  ScanRegistrationResult sr;
  LaserOdometry lo;
  LaserMapping lm;
  bool do_mapping;
  double stage_ms[3] = {0, 0, 0};
};

}  // namespace orc
