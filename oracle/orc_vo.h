// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  PARITY UNPINNED.
//
// CPU restatement of the depth-enhanced visual-odometry residual stack (config 4):
//   PointCloudUtil::projectPointCloud     src/visual_odometry/src/point_cloud_util.cpp:148-174
//   PointCloudUtil::downsamplePointCloud  src/visual_odometry/src/point_cloud_util.cpp:205-260
//   PointCloudUtil::queryDepth            src/visual_odometry/src/point_cloud_util.cpp:302-387
//   VisualOdometry::solveNlsAll           src/visual_odometry/src/visual_odometry.cpp:254-450
// The image front-end (OpenCV feature matching) is out of scope: matched pixel pairs are inputs.
#pragma once
#include <vector>
#include "orc_ceres.h"
#include "orc_factors.h"

namespace orc {

struct VOCalib {           // row-major, f32, exactly as the caller holds them
  float cam_T_velo[16];    // point_cloud_util.h:43
  float rect0_T_cam[16];   // 4x4; the ROS path fills only the 3x3 block, (3,3) stays 0 (visual_odometry.cpp:140-144)
  float P_rect0[12];       // 3x4
};

class DepthMap {  // the slice of PointCloudUtil the hot path touches
 public:
  static const int IMG_HEIGHT = 375, IMG_WIDTH = 1242;  // point_cloud_util.h:41-42
  int downsample_grid_size = 5;
  int new_width = 0, new_height = 0;
  std::vector<float> point_cloud_2d;  // n_front x 3 (u, v, depth)
  std::vector<float> bucket_x, bucket_y, bucket_depth;  // [new_width][new_height]
  std::vector<int> bucket_count;
  void projectPointCloud(const float* xyz_pad4, int n, const VOCalib& c);
  void downsamplePointCloud();
  float queryDepth(float x, float y, int searching_radius = 2) const;
};

struct VOMatchDebug {
  int kind;  // 0 skipped (outlier gate), 32, 22
  float depth0;
  double obs[5];
};

class VisualOdometry {
 public:
  VOCalib calib;
  DepthMap maps[2];
  int count = -1, i = 0;       // visual_odometry.cpp:86-90  reset(): ++count; i = count % 2
  double angles_0to1[3] = {0, 0, 0}, t_0to1[3] = {0, 0, 0};
  int remove_VO_outlier = 100;  // vloam_main.launch:6
  int counter32 = 0, counter22 = 0;
  SolveSummary summary;
  std::vector<VOMatchDebug> match_debug;
  void reset() { ++count; i = count % 2; }
  void processPointCloud(const float* xyz_pad4, int n) { maps[i].projectPointCloud(xyz_pad4, n, calib); maps[i].downsamplePointCloud(); }
  // prev_uv / curr_uv: integer pixel coordinates (the reference truncates keypoint floats to int,
  // visual_odometry.cpp:283-294).  init_angles/init_t: the LO prior cam0_curr_LOT_cam0_prev as
  // angle-axis + t, or nullptr for reset_VO_to_identity.
  void solveNlsAll(const int* prev_uv, const int* curr_uv, int n_match, const double* init_angles, const double* init_t);
};

// 3x3 f32 column-pivoted Householder QR solve — stand-in for
// P_rect0.leftCols(3).colPivHouseholderQr().solve(v) (visual_odometry.cpp:350-355).
void solve3x3_colpiv_qr_f32(const float A[9], const float b[3], float x[3]);

}  // namespace orc
