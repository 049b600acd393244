// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  PARITY UNPINNED.
// Line references are into /root/reference/src/lidar_odometry_mapping/.
#include "orc_loam.h"
#include <chrono>
#include <cmath>
#include <limits>

namespace orc {

static const double kEps = std::numeric_limits<double>::epsilon();

// ====================================================================== ScanRegistration
// src/scan_registration.cpp:131-449.  Float/double promotion follows the C++ expressions there,
// with the float overloads of atan/sqrt/atan2 (the TU pulls in <math.h> via ROS/OpenCV headers and
// `using std::atan2`, scan_registration.h:57-59).
bool scan_registration(const float* in, int n, const Config& cfg, ScanRegistrationResult* out) {
  *out = ScanRegistrationResult();
  const int N_SCANS = cfg.scan_line;
  const double scanPeriod = 0.1;  // scan_registration.h:84
  struct P3 { float x, y, z; };
  std::vector<P3> pin;
  pin.reserve(n);
  // :157 pcl::removeNaNFromPointCloud
  for (int i = 0; i < n; i++) {
    const float x = in[4 * i], y = in[4 * i + 1], z = in[4 * i + 2];
    if (!std::isfinite(x) || !std::isfinite(y) || !std::isfinite(z)) continue;
    pin.push_back({x, y, z});
  }
  // :158 removeClosedPointCloud(:100-129), thres is the double MINIMUM_RANGE passed as float
  {
    const float thres = static_cast<float>(cfg.minimum_range);
    size_t j = 0;
    for (size_t i = 0; i < pin.size(); i++) {
      if (pin[i].x * pin[i].x + pin[i].y * pin[i].y + pin[i].z * pin[i].z < thres * thres) continue;
      pin[j++] = pin[i];
    }
    pin.resize(j);
  }
  int cloudSize = (int)pin.size();
  out->n_after_s1 = cloudSize;
  if (cloudSize == 0) return false;

  // :166-176
  float startOri = -atan2f(pin[0].y, pin[0].x);
  float endOri = -atan2f(pin[cloudSize - 1].y, pin[cloudSize - 1].x) + 2 * M_PI;
  if (endOri - startOri > 3 * M_PI) endOri -= 2 * M_PI;
  else if (endOri - startOri < M_PI) endOri += 2 * M_PI;
  out->startOri = startOri;
  out->endOri = endOri;

  // :183-267
  bool halfPassed = false;
  int count = cloudSize;
  std::vector<Cloud> laserCloudScans(N_SCANS);
  for (int i = 0; i < cloudSize; i++) {
    PointXYZI point;
    point.x = pin[i].x; point.y = pin[i].y; point.z = pin[i].z;
    float angle = atanf(point.z / sqrtf(point.x * point.x + point.y * point.y)) * 180 / M_PI;
    int scanID = 0;
    if (N_SCANS == 16) {
      scanID = int((angle + 15) / 2 + 0.5);
      if (scanID > (N_SCANS - 1) || scanID < 0) { count--; continue; }
    } else if (N_SCANS == 32) {
      scanID = int((angle + 92.0 / 3.0) * 3.0 / 4.0);
      if (scanID > (N_SCANS - 1) || scanID < 0) { count--; continue; }
    } else if (N_SCANS == 64) {
      if (angle >= -8.83) scanID = int((2 - angle) * 3.0 + 0.5);
      else scanID = N_SCANS / 2 + int((-8.83 - angle) * 2.0 + 0.5);
      if (angle > 2 || angle < -24.33 || scanID > 50 || scanID < 0) { count--; continue; }
    } else {
      return false;  // ROS_BREAK()
    }
    float ori = -atan2f(point.y, point.x);
    if (!halfPassed) {
      if (ori < startOri - M_PI / 2) ori += 2 * M_PI;
      else if (ori > startOri + M_PI * 3 / 2) ori -= 2 * M_PI;
      if (ori - startOri > M_PI) { halfPassed = true; out->halfPassedAt = i; }
    } else {
      ori += 2 * M_PI;
      if (ori < endOri - M_PI * 3 / 2) ori += 2 * M_PI;
      else if (ori > endOri + M_PI / 2) ori -= 2 * M_PI;
    }
    float relTime = (ori - startOri) / (endOri - startOri);
    point.intensity = scanID + scanPeriod * relTime;
    laserCloudScans[scanID].push_back(point);
  }
  cloudSize = count;

  // :276-281
  std::vector<int> scanStartInd(N_SCANS, 0), scanEndInd(N_SCANS, 0);
  Cloud& laserCloud = out->laserCloud;
  for (int i = 0; i < N_SCANS; i++) {
    scanStartInd[i] = (int)laserCloud.size() + 5;
    laserCloud.insert(laserCloud.end(), laserCloudScans[i].begin(), laserCloudScans[i].end());
    scanEndInd[i] = (int)laserCloud.size() - 6;
  }
  out->scanStartInd = scanStartInd;
  out->scanEndInd = scanEndInd;

  // :288-307 (the reference's member arrays persist across frames; every element that is later READ
  // lies in [5, cloudSize-5) and is re-initialised here, so zero-filling the rest is equivalent)
  std::vector<float>& cloudCurvature = out->cloudCurvature;
  std::vector<int>& cloudSortInd = out->cloudSortInd;
  std::vector<int>& cloudNeighborPicked = out->cloudNeighborPicked;
  std::vector<int>& cloudLabel = out->cloudLabel;
  cloudCurvature.assign(cloudSize, 0.0f);
  cloudSortInd.assign(cloudSize, 0);
  cloudNeighborPicked.assign(cloudSize, 0);
  cloudLabel.assign(cloudSize, 0);
  const Cloud& P = laserCloud;
  for (int i = 5; i < cloudSize - 5; i++) {
    float diffX = P[i - 5].x + P[i - 4].x + P[i - 3].x + P[i - 2].x + P[i - 1].x - 10 * P[i].x + P[i + 1].x + P[i + 2].x +
                  P[i + 3].x + P[i + 4].x + P[i + 5].x;
    float diffY = P[i - 5].y + P[i - 4].y + P[i - 3].y + P[i - 2].y + P[i - 1].y - 10 * P[i].y + P[i + 1].y + P[i + 2].y +
                  P[i + 3].y + P[i + 4].y + P[i + 5].y;
    float diffZ = P[i - 5].z + P[i - 4].z + P[i - 3].z + P[i - 2].z + P[i - 1].z - 10 * P[i].z + P[i + 1].z + P[i + 2].z +
                  P[i + 3].z + P[i + 4].z + P[i + 5].z;
    cloudCurvature[i] = diffX * diffX + diffY * diffY + diffZ * diffZ;
    cloudSortInd[i] = i;
    cloudNeighborPicked[i] = 0;
    cloudLabel[i] = 0;
  }

  auto spread = [&](int ind) {  // :353-376 == :397-420
    for (int l = 1; l <= 5; l++) {
      float diffX = P[ind + l].x - P[ind + l - 1].x;
      float diffY = P[ind + l].y - P[ind + l - 1].y;
      float diffZ = P[ind + l].z - P[ind + l - 1].z;
      if (diffX * diffX + diffY * diffY + diffZ * diffZ > 0.05) break;
      cloudNeighborPicked[ind + l] = 1;
    }
    for (int l = -1; l >= -5; l--) {
      float diffX = P[ind + l].x - P[ind + l + 1].x;
      float diffY = P[ind + l].y - P[ind + l + 1].y;
      float diffZ = P[ind + l].z - P[ind + l + 1].z;
      if (diffX * diffX + diffY * diffY + diffZ * diffZ > 0.05) break;
      cloudNeighborPicked[ind + l] = 1;
    }
  };

  // :312-440
  for (int i = 0; i < N_SCANS; i++) {
    if (scanEndInd[i] - scanStartInd[i] < 6) continue;
    Cloud surfPointsLessFlatScan;
    for (int j = 0; j < 6; j++) {
      int sp = scanStartInd[i] + (scanEndInd[i] - scanStartInd[i]) * j / 6;
      int ep = scanStartInd[i] + (scanEndInd[i] - scanStartInd[i]) * (j + 1) / 6 - 1;
      auto comp = [&](const int& a, const int& b) { return cloudCurvature[a] < cloudCurvature[b]; };
#ifdef ORC_STD_SORT
      std::sort(cloudSortInd.begin() + sp, cloudSortInd.begin() + ep + 1, comp);  // :323 literally
#else
      std::stable_sort(cloudSortInd.begin() + sp, cloudSortInd.begin() + ep + 1, comp);  // canonical tie order: index asc
#endif
      int largestPickedNum = 0;
      for (int k = ep; k >= sp; k--) {
        int ind = cloudSortInd[k];
        if (cloudNeighborPicked[ind] == 0 && cloudCurvature[ind] > 0.1) {
          largestPickedNum++;
          if (largestPickedNum <= 2) {
            cloudLabel[ind] = 2;
            out->cornerPointsSharp.push_back(P[ind]); out->sharpInd.push_back(ind);
            out->cornerPointsLessSharp.push_back(P[ind]); out->lessSharpInd.push_back(ind);
          } else if (largestPickedNum <= 20) {
            cloudLabel[ind] = 1;
            out->cornerPointsLessSharp.push_back(P[ind]); out->lessSharpInd.push_back(ind);
          } else {
            break;
          }
          cloudNeighborPicked[ind] = 1;
          spread(ind);
        }
      }
      int smallestPickedNum = 0;
      for (int k = sp; k <= ep; k++) {
        int ind = cloudSortInd[k];
        if (cloudNeighborPicked[ind] == 0 && cloudCurvature[ind] < 0.1) {
          cloudLabel[ind] = -1;
          out->surfPointsFlat.push_back(P[ind]); out->flatInd.push_back(ind);
          smallestPickedNum++;
          if (smallestPickedNum >= 4) break;  // 4th flat point does not suppress its neighbours (:390-394)
          cloudNeighborPicked[ind] = 1;
          spread(ind);
        }
      }
      for (int k = sp; k <= ep; k++)
        if (cloudLabel[k] <= 0) surfPointsLessFlatScan.push_back(P[k]);
    }
    Cloud ds = voxel_grid(surfPointsLessFlatScan, 0.2f);  // :433-437
    out->surfPointsLessFlat.insert(out->surfPointsLessFlat.end(), ds.begin(), ds.end());
  }
  return true;
}

// ====================================================================== LaserOdometry
void LaserOdometry::reset_all() {  // laser_odometry.cpp:41-117
  systemInited = false;
  q_w_curr = Quat<double>(0, 0, 0, 1);
  t_w_curr = V3<double>(0, 0, 0);
  para_q[0] = para_q[1] = para_q[2] = 0.0; para_q[3] = 1.0;
  para_t[0] = para_t[1] = para_t[2] = 0.0;
  prior_q[0] = prior_q[1] = prior_q[2] = 0.0; prior_q[3] = 1.0;
  prior_t[0] = prior_t[1] = prior_t[2] = 0.0;
  frameCount = 0;
  laserCloudCornerLast.clear(); laserCloudSurfLast.clear(); laserCloudFullRes.clear();
}

void LaserOdometry::set_vo_prior(const double q[4], const double t[3]) {
  for (int i = 0; i < 4; i++) prior_q[i] = q[i];
  for (int i = 0; i < 3; i++) prior_t[i] = t[i];
}

void LaserOdometry::input(const ScanRegistrationResult& sr) {
  laserCloudFullRes = sr.laserCloud;
  cornerPointsSharp = sr.cornerPointsSharp;
  cornerPointsLessSharp = sr.cornerPointsLessSharp;
  surfPointsFlat = sr.surfPointsFlat;
  surfPointsLessFlat = sr.surfPointsLessFlat;
}

void LaserOdometry::TransformToStart(const PointXYZI& pi, PointXYZI* po) const {
  const double s = 1.0;  // DISTORTION == false (laser_odometry.h:90)
  Quat<double> q_last_curr(para_q[0], para_q[1], para_q[2], para_q[3]);
  Quat<double> q_point_last = slerp_from_identity(s, q_last_curr, kEps);
  V3<double> t_point_last = s * V3<double>(para_t[0], para_t[1], para_t[2]);
  V3<double> point(pi.x, pi.y, pi.z);
  V3<double> un_point = rotate(q_point_last, point) + t_point_last;
  po->x = un_point.x; po->y = un_point.y; po->z = un_point.z;
  po->intensity = pi.intensity;
}

void LaserOdometry::solveLO() {
  const double DISTANCE_SQ_THRESHOLD = 25, NEARBY_SCAN = 2.5;  // laser_odometry.h:93-95
  debug.clear();
  if (!systemInited) {
    systemInited = true;
  } else {
    const int cornerPointsSharpNum = (int)cornerPointsSharp.size();
    const int surfPointsFlatNum = (int)surfPointsFlat.size();
    const Cloud& CL = laserCloudCornerLast;
    const Cloud& SL = laserCloudSurfLast;
    for (size_t opti_counter = 0; opti_counter < 2; ++opti_counter) {
      corner_correspondence = 0;
      plane_correspondence = 0;
      Problem problem;
      if (!cfg.detach_VO_LO) {  // :223-236
        for (int i = 0; i < 4; i++) para_q[i] = prior_q[i];
        for (int i = 0; i < 3; i++) para_t[i] = prior_t[i];
      }
      debug.emplace_back();
      LOIterationDebug& dbg = debug.back();
      for (int i = 0; i < 4; i++) dbg.q_in[i] = para_q[i];
      for (int i = 0; i < 3; i++) dbg.t_in[i] = para_t[i];

      PointXYZI pointSel;
      int pointSearchInd[1];
      float pointSearchSqDis[1];
      // :266-350 corner features
      for (int i = 0; i < cornerPointsSharpNum; ++i) {
        TransformToStart(cornerPointsSharp[i], &pointSel);
        const float qp[3] = {pointSel.x, pointSel.y, pointSel.z};
        if (kdtreeCornerLast.knn(qp, 1, pointSearchInd, pointSearchSqDis) < 1) continue;
        int closestPointInd = -1, minPointInd2 = -1;
        if (pointSearchSqDis[0] < DISTANCE_SQ_THRESHOLD) {
          closestPointInd = pointSearchInd[0];
          int closestPointScanID = int(CL[closestPointInd].intensity);
          double minPointSqDis2 = DISTANCE_SQ_THRESHOLD;
          for (int j = closestPointInd + 1; j < (int)CL.size(); ++j) {
            if (int(CL[j].intensity) <= closestPointScanID) continue;
            if (int(CL[j].intensity) > (closestPointScanID + NEARBY_SCAN)) break;
            double pointSqDis = (CL[j].x - pointSel.x) * (CL[j].x - pointSel.x) + (CL[j].y - pointSel.y) * (CL[j].y - pointSel.y) +
                                (CL[j].z - pointSel.z) * (CL[j].z - pointSel.z);
            if (pointSqDis < minPointSqDis2) { minPointSqDis2 = pointSqDis; minPointInd2 = j; }
          }
          for (int j = closestPointInd - 1; j >= 0; --j) {
            if (int(CL[j].intensity) >= closestPointScanID) continue;
            if (int(CL[j].intensity) < (closestPointScanID - NEARBY_SCAN)) break;
            double pointSqDis = (CL[j].x - pointSel.x) * (CL[j].x - pointSel.x) + (CL[j].y - pointSel.y) * (CL[j].y - pointSel.y) +
                                (CL[j].z - pointSel.z) * (CL[j].z - pointSel.z);
            if (pointSqDis < minPointSqDis2) { minPointSqDis2 = pointSqDis; minPointInd2 = j; }
          }
        }
        if (minPointInd2 >= 0) {
          Vec3d curr_point(cornerPointsSharp[i].x, cornerPointsSharp[i].y, cornerPointsSharp[i].z);
          Vec3d last_point_a(CL[closestPointInd].x, CL[closestPointInd].y, CL[closestPointInd].z);
          Vec3d last_point_b(CL[minPointInd2].x, CL[minPointInd2].y, CL[minPointInd2].z);
          const double s = 1.0;
          problem.Add(new AutoDiffCost<LidarEdgeFactor, 3, 4>(LidarEdgeFactor(curr_point, last_point_a, last_point_b, s)));
          dbg.corner.push_back({i, closestPointInd, minPointInd2});
          corner_correspondence++;
        }
      }
      // :353-444 plane features
      for (int i = 0; i < surfPointsFlatNum; ++i) {
        TransformToStart(surfPointsFlat[i], &pointSel);
        const float qp[3] = {pointSel.x, pointSel.y, pointSel.z};
        if (kdtreeSurfLast.knn(qp, 1, pointSearchInd, pointSearchSqDis) < 1) continue;
        int closestPointInd = -1, minPointInd2 = -1, minPointInd3 = -1;
        if (pointSearchSqDis[0] < DISTANCE_SQ_THRESHOLD) {
          closestPointInd = pointSearchInd[0];
          int closestPointScanID = int(SL[closestPointInd].intensity);
          double minPointSqDis2 = DISTANCE_SQ_THRESHOLD, minPointSqDis3 = DISTANCE_SQ_THRESHOLD;
          for (int j = closestPointInd + 1; j < (int)SL.size(); ++j) {
            if (int(SL[j].intensity) > (closestPointScanID + NEARBY_SCAN)) break;
            double pointSqDis = (SL[j].x - pointSel.x) * (SL[j].x - pointSel.x) + (SL[j].y - pointSel.y) * (SL[j].y - pointSel.y) +
                                (SL[j].z - pointSel.z) * (SL[j].z - pointSel.z);
            if (int(SL[j].intensity) <= closestPointScanID && pointSqDis < minPointSqDis2) {
              minPointSqDis2 = pointSqDis; minPointInd2 = j;
            } else if (int(SL[j].intensity) > closestPointScanID && pointSqDis < minPointSqDis3) {
              minPointSqDis3 = pointSqDis; minPointInd3 = j;
            }
          }
          for (int j = closestPointInd - 1; j >= 0; --j) {
            if (int(SL[j].intensity) < (closestPointScanID - NEARBY_SCAN)) break;
            double pointSqDis = (SL[j].x - pointSel.x) * (SL[j].x - pointSel.x) + (SL[j].y - pointSel.y) * (SL[j].y - pointSel.y) +
                                (SL[j].z - pointSel.z) * (SL[j].z - pointSel.z);
            if (int(SL[j].intensity) >= closestPointScanID && pointSqDis < minPointSqDis2) {
              minPointSqDis2 = pointSqDis; minPointInd2 = j;
            } else if (int(SL[j].intensity) < closestPointScanID && pointSqDis < minPointSqDis3) {
              minPointSqDis3 = pointSqDis; minPointInd3 = j;
            }
          }
          if (minPointInd2 >= 0 && minPointInd3 >= 0) {
            Vec3d curr_point(surfPointsFlat[i].x, surfPointsFlat[i].y, surfPointsFlat[i].z);
            Vec3d last_point_a(SL[closestPointInd].x, SL[closestPointInd].y, SL[closestPointInd].z);
            Vec3d last_point_b(SL[minPointInd2].x, SL[minPointInd2].y, SL[minPointInd2].z);
            Vec3d last_point_c(SL[minPointInd3].x, SL[minPointInd3].y, SL[minPointInd3].z);
            const double s = 1.0;
            problem.Add(new AutoDiffCost<LidarPlaneFactor, 1, 4>(LidarPlaneFactor(curr_point, last_point_a, last_point_b, last_point_c, s)));
            dbg.plane.push_back({i, closestPointInd, minPointInd2, minPointInd3});
            plane_correspondence++;
          }
        }
      }
      // :457-463
      SolveOptions options;
      options.max_num_iterations = 4;
      options.huber_a = 0.1;
      options.quaternion_block0 = true;
      problem.Solve(options, para_q, para_t, &dbg.summary);
      for (int i = 0; i < 4; i++) dbg.q_out[i] = para_q[i];
      for (int i = 0; i < 3; i++) dbg.t_out[i] = para_t[i];
    }
    // :477-478
    Quat<double> q_last_curr(para_q[0], para_q[1], para_q[2], para_q[3]);
    V3<double> t_last_curr(para_t[0], para_t[1], para_t[2]);
    t_w_curr = t_w_curr + rotate(q_w_curr, t_last_curr);
    q_w_curr = qmul(q_w_curr, q_last_curr);
  }
  // :511-526 (TransformToEnd block is `if (0)`)
  laserCloudCornerLast.swap(cornerPointsLessSharp);
  laserCloudSurfLast.swap(surfPointsLessFlat);
  kdtreeCornerLast.build(laserCloudCornerLast);
  kdtreeSurfLast.build(laserCloudSurfLast);
  frameCount++;
  // output() :610-629
  skip_frame = !(frameCount % cfg.mapping_skip_frame == 0);
}

// ====================================================================== LaserMapping
LaserMapping::LaserMapping(const Config& c) : cfg(c) {  // laser_mapping.cpp:40-125, laser_mapping.h:76-78
  laserCloudCornerArray.resize(laserCloudNum);
  laserCloudSurfArray.resize(laserCloudNum);
  parameters[0] = parameters[1] = parameters[2] = 0.0; parameters[3] = 1.0;
  parameters[4] = parameters[5] = parameters[6] = 0.0;
  q_wmap_wodom = Quat<double>(0, 0, 0, 1); t_wmap_wodom = V3<double>(0, 0, 0);
  q_wodom_curr = Quat<double>(0, 0, 0, 1); t_wodom_curr = V3<double>(0, 0, 0);
}

size_t LaserMapping::map_points_corner() const { size_t s = 0; for (auto& c : laserCloudCornerArray) s += c.size(); return s; }
size_t LaserMapping::map_points_surf() const { size_t s = 0; for (auto& c : laserCloudSurfArray) s += c.size(); return s; }

void LaserMapping::transformUpdate() {
  Quat<double> q_w_curr(parameters[0], parameters[1], parameters[2], parameters[3]);
  V3<double> t_w_curr(parameters[4], parameters[5], parameters[6]);
  q_wmap_wodom = qmul(q_w_curr, qinverse(q_wodom_curr));
  t_wmap_wodom = t_w_curr - rotate(q_wmap_wodom, t_wodom_curr);
}

void LaserMapping::pointAssociateToMap(const PointXYZI& pi, PointXYZI* po) const {
  Quat<double> q_w_curr(parameters[0], parameters[1], parameters[2], parameters[3]);
  V3<double> t_w_curr(parameters[4], parameters[5], parameters[6]);
  V3<double> point_curr(pi.x, pi.y, pi.z);
  V3<double> point_w = rotate(q_w_curr, point_curr) + t_w_curr;
  po->x = point_w.x; po->y = point_w.y; po->z = point_w.z;
  po->intensity = pi.intensity;
}

void LaserMapping::registered_cloud(Cloud* out) const {
  out->resize(laserCloudFullRes.size());
  for (size_t i = 0; i < laserCloudFullRes.size(); i++) pointAssociateToMap(laserCloudFullRes[i], &(*out)[i]);
}

void LaserMapping::input(const Cloud& cornerLast, const Cloud& surfLast, const Cloud& fullRes, const Quat<double>& q_wodom_curr_,
                         const V3<double>& t_wodom_curr_, bool skip_frame_) {
  skip_frame = skip_frame_;
  if (!skip_frame) {
    laserCloudCornerLast = cornerLast;
    laserCloudSurfLast = surfLast;
    laserCloudFullRes = fullRes;
  }
  q_wodom_curr = q_wodom_curr_;
  t_wodom_curr = t_wodom_curr_;
  // :186-195 transformAssociateToMap: a skipped frame only refreshes the high-frequency pose that publish() sends out
  Quat<double> q = qmul(q_wmap_wodom, q_wodom_curr);
  V3<double> t = rotate(q_wmap_wodom, t_wodom_curr) + t_wmap_wodom;
  if (skip_frame) {
    q_w_curr_highfreq = q; t_w_curr_highfreq = t;
  } else {
    parameters[0] = q.x; parameters[1] = q.y; parameters[2] = q.z; parameters[3] = q.w;
    parameters[4] = t.x; parameters[5] = t.y; parameters[6] = t.z;
  }
}

void LaserMapping::solveMapping() {
  const int W = laserCloudWidth, H = laserCloudHeight, D = laserCloudDepth;
  auto IDX = [&](int i, int j, int k) { return i + W * j + W * H * k; };
  debug.clear();
  // :207-216
  int centerCubeI = int((parameters[4] + 25.0) / 50.0) + laserCloudCenWidth;
  int centerCubeJ = int((parameters[5] + 25.0) / 50.0) + laserCloudCenHeight;
  int centerCubeK = int((parameters[6] + 25.0) / 50.0) + laserCloudCenDepth;
  if (parameters[4] + 25.0 < 0) centerCubeI--;
  if (parameters[5] + 25.0 < 0) centerCubeJ--;
  if (parameters[6] + 25.0 < 0) centerCubeK--;

  // :218-402 grid roll.  Moving the clouds is equivalent to the reference's shared_ptr shuffling.
  auto shift_line = [&](auto idx_of, int len, bool up) {
    // up: a[len-1] <- a[len-2] <- ... <- a[0]; a[0] = cleared(old a[len-1])
    for (int pass = 0; pass < 2; pass++) {
      std::vector<Cloud>& A = pass == 0 ? laserCloudCornerArray : laserCloudSurfArray;
      if (up) {
        Cloud saved = std::move(A[idx_of(len - 1)]);
        for (int t = len - 1; t >= 1; t--) A[idx_of(t)] = std::move(A[idx_of(t - 1)]);
        saved.clear();
        A[idx_of(0)] = std::move(saved);
      } else {
        Cloud saved = std::move(A[idx_of(0)]);
        for (int t = 0; t < len - 1; t++) A[idx_of(t)] = std::move(A[idx_of(t + 1)]);
        saved.clear();
        A[idx_of(len - 1)] = std::move(saved);
      }
    }
  };
  while (centerCubeI < 3) {
    for (int j = 0; j < H; j++) for (int k = 0; k < D; k++) shift_line([&](int t) { return IDX(t, j, k); }, W, true);
    centerCubeI++; laserCloudCenWidth++;
  }
  while (centerCubeI >= W - 3) {
    for (int j = 0; j < H; j++) for (int k = 0; k < D; k++) shift_line([&](int t) { return IDX(t, j, k); }, W, false);
    centerCubeI--; laserCloudCenWidth--;
  }
  while (centerCubeJ < 3) {
    for (int i = 0; i < W; i++) for (int k = 0; k < D; k++) shift_line([&](int t) { return IDX(i, t, k); }, H, true);
    centerCubeJ++; laserCloudCenHeight++;
  }
  while (centerCubeJ >= H - 3) {
    for (int i = 0; i < W; i++) for (int k = 0; k < D; k++) shift_line([&](int t) { return IDX(i, t, k); }, H, false);
    centerCubeJ--; laserCloudCenHeight--;
  }
  while (centerCubeK < 3) {
    for (int i = 0; i < W; i++) for (int j = 0; j < H; j++) shift_line([&](int t) { return IDX(i, j, t); }, D, true);
    centerCubeK++; laserCloudCenDepth++;
  }
  while (centerCubeK >= D - 3) {
    for (int i = 0; i < W; i++) for (int j = 0; j < H; j++) shift_line([&](int t) { return IDX(i, j, t); }, D, false);
    centerCubeK--; laserCloudCenDepth--;
  }

  // :404-420
  validInd.clear();
  for (int i = centerCubeI - 2; i <= centerCubeI + 2; i++)
    for (int j = centerCubeJ - 2; j <= centerCubeJ + 2; j++)
      for (int k = centerCubeK - 1; k <= centerCubeK + 1; k++)
        if (i >= 0 && i < W && j >= 0 && j < H && k >= 0 && k < D) validInd.push_back(IDX(i, j, k));

  // :422-430
  laserCloudCornerFromMap.clear();
  laserCloudSurfFromMap.clear();
  for (int ind : validInd) {
    laserCloudCornerFromMap.insert(laserCloudCornerFromMap.end(), laserCloudCornerArray[ind].begin(), laserCloudCornerArray[ind].end());
    laserCloudSurfFromMap.insert(laserCloudSurfFromMap.end(), laserCloudSurfArray[ind].begin(), laserCloudSurfArray[ind].end());
  }
  const int laserCloudCornerFromMapNum = (int)laserCloudCornerFromMap.size();
  const int laserCloudSurfFromMapNum = (int)laserCloudSurfFromMap.size();

  // :432-440
  laserCloudCornerStack = voxel_grid(laserCloudCornerLast, cfg.mapping_line_resolution);
  laserCloudSurfStack = voxel_grid(laserCloudSurfLast, cfg.mapping_plane_resolution);
  const int laserCloudCornerStackNum = (int)laserCloudCornerStack.size();
  const int laserCloudSurfStackNum = (int)laserCloudSurfStack.size();

  if (laserCloudCornerFromMapNum > 10 && laserCloudSurfFromMapNum > 50) {  // :448
    KdTree kdtreeCornerFromMap, kdtreeSurfFromMap;
    kdtreeCornerFromMap.build(laserCloudCornerFromMap);
    kdtreeSurfFromMap.build(laserCloudSurfFromMap);
    const Cloud& CM = laserCloudCornerFromMap;
    const Cloud& SM = laserCloudSurfFromMap;
    int pointSearchInd[5];
    float pointSearchSqDis[5];
    PointXYZI pointOri, pointSel;
    for (int iterCount = 0; iterCount < 2; iterCount++) {
      Problem problem;
      debug.emplace_back();
      MapIterationDebug& dbg = debug.back();
      for (int i = 0; i < 4; i++) dbg.q_in[i] = parameters[i];
      for (int i = 0; i < 3; i++) dbg.t_in[i] = parameters[4 + i];
      int corner_num = 0;
      for (int i = 0; i < laserCloudCornerStackNum; i++) {  // :472-517
        pointOri = laserCloudCornerStack[i];
        pointAssociateToMap(pointOri, &pointSel);
        const float qp[3] = {pointSel.x, pointSel.y, pointSel.z};
        if (kdtreeCornerFromMap.knn(qp, 5, pointSearchInd, pointSearchSqDis) < 5) continue;
        if (pointSearchSqDis[4] < 1.0) {
          Vec3d nearCorners[5];
          Vec3d center(0, 0, 0);
          for (int j = 0; j < 5; j++) {
            Vec3d tmp(CM[pointSearchInd[j]].x, CM[pointSearchInd[j]].y, CM[pointSearchInd[j]].z);
            center = center + tmp;
            nearCorners[j] = tmp;
          }
          center = center / 5.0;
          double covMat[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
          for (int j = 0; j < 5; j++) {
            Vec3d z = nearCorners[j] - center;
            const double zz[3] = {z.x, z.y, z.z};
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) covMat[a][b] = covMat[a][b] + zz[a] * zz[b];
          }
          double evals[3], evecs[3][3];
          sym_eig3(covMat, evals, evecs);
          Vec3d unit_direction(evecs[0][2], evecs[1][2], evecs[2][2]);
          Vec3d curr_point(pointOri.x, pointOri.y, pointOri.z);
          if (evals[2] > 3 * evals[1]) {
            Vec3d point_on_line = center;
            Vec3d point_a = 0.1 * unit_direction + point_on_line;
            Vec3d point_b = -0.1 * unit_direction + point_on_line;
            problem.Add(new AutoDiffCost<LidarEdgeFactor, 3, 4>(LidarEdgeFactor(curr_point, point_a, point_b, 1.0)));
            dbg.corner_idx.push_back(i);
            dbg.corner_ab.push_back({point_a.x, point_a.y, point_a.z, point_b.x, point_b.y, point_b.z});
            corner_num++;
          }
        }
      }
      int surf_num = 0;
      for (int i = 0; i < laserCloudSurfStackNum; i++) {  // :538-581
        pointOri = laserCloudSurfStack[i];
        pointAssociateToMap(pointOri, &pointSel);
        const float qp[3] = {pointSel.x, pointSel.y, pointSel.z};
        if (kdtreeSurfFromMap.knn(qp, 5, pointSearchInd, pointSearchSqDis) < 5) continue;
        if (pointSearchSqDis[4] < 1.0) {
          double matA0[15], matB0[5], nrm[3];
          for (int j = 0; j < 5; j++) {
            matA0[j * 3 + 0] = SM[pointSearchInd[j]].x;
            matA0[j * 3 + 1] = SM[pointSearchInd[j]].y;
            matA0[j * 3 + 2] = SM[pointSearchInd[j]].z;
            matB0[j] = -1.0;
          }
          if (!householder_ls(matA0, matB0, 5, 3, nrm)) continue;
          Vec3d norm_(nrm[0], nrm[1], nrm[2]);
          double negative_OA_dot_norm = 1 / norm(norm_);
          norm_ = norm_ / norm(norm_);
          bool planeValid = true;
          for (int j = 0; j < 5; j++) {
            if (std::fabs(norm_.x * SM[pointSearchInd[j]].x + norm_.y * SM[pointSearchInd[j]].y + norm_.z * SM[pointSearchInd[j]].z +
                          negative_OA_dot_norm) > 0.2) {
              planeValid = false;
              break;
            }
          }
          Vec3d curr_point(pointOri.x, pointOri.y, pointOri.z);
          if (planeValid) {
            problem.Add(new AutoDiffCost<LidarPlaneNormFactor, 1, 4>(LidarPlaneNormFactor(curr_point, norm_, negative_OA_dot_norm)));
            dbg.surf_idx.push_back(i);
            dbg.surf_plane.push_back({norm_.x, norm_.y, norm_.z, negative_OA_dot_norm});
            surf_num++;
          }
        }
      }
      dbg.corner_num = corner_num;
      dbg.surf_num = surf_num;
      SolveOptions options;  // :609-617
      options.max_num_iterations = 4;
      options.huber_a = 0.1;
      options.quaternion_block0 = true;
      problem.Solve(options, parameters, parameters + 4, &dbg.summary);
      for (int i = 0; i < 4; i++) dbg.q_out[i] = parameters[i];
      for (int i = 0; i < 3; i++) dbg.t_out[i] = parameters[4 + i];
    }
  }
  transformUpdate();  // :636

  // :639-683
  PointXYZI pointSel;
  auto insert = [&](const Cloud& stack, std::vector<Cloud>& arr) {
    for (size_t i = 0; i < stack.size(); i++) {
      pointAssociateToMap(stack[i], &pointSel);
      int cubeI = int((pointSel.x + 25.0) / 50.0) + laserCloudCenWidth;
      int cubeJ = int((pointSel.y + 25.0) / 50.0) + laserCloudCenHeight;
      int cubeK = int((pointSel.z + 25.0) / 50.0) + laserCloudCenDepth;
      if (pointSel.x + 25.0 < 0) cubeI--;
      if (pointSel.y + 25.0 < 0) cubeJ--;
      if (pointSel.z + 25.0 < 0) cubeK--;
      if (cubeI >= 0 && cubeI < W && cubeJ >= 0 && cubeJ < H && cubeK >= 0 && cubeK < D) arr[IDX(cubeI, cubeJ, cubeK)].push_back(pointSel);
    }
  };
  insert(laserCloudCornerStack, laserCloudCornerArray);
  insert(laserCloudSurfStack, laserCloudSurfArray);

  // :689-702
  for (int ind : validInd) {
    laserCloudCornerArray[ind] = voxel_grid(laserCloudCornerArray[ind], cfg.mapping_line_resolution);
    laserCloudSurfArray[ind] = voxel_grid(laserCloudSurfArray[ind], cfg.mapping_plane_resolution);
  }
  frameCount++;
}

// ====================================================================== façade
bool Pipeline::process(const float* xyz_pad4, int n) {
  typedef std::chrono::steady_clock clk;
  auto t0 = clk::now();
  lm.reset();  // LidarOdometryMapping::reset (lidar_odometry_mapping.cpp:65-71)
  if (!scan_registration(xyz_pad4, n, cfg, &sr)) return false;
  auto t1 = clk::now();
  lo.input(sr);
  lo.solveLO();
  auto t2 = clk::now();
  if (do_mapping) {
    lm.input(lo.laserCloudCornerLast, lo.laserCloudSurfLast, lo.laserCloudFullRes, lo.q_w_curr, lo.t_w_curr, lo.skip_frame);
    if (!lo.skip_frame) lm.solveMapping();
  }
  auto t3 = clk::now();
  stage_ms[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
  stage_ms[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
  stage_ms[2] = std::chrono::duration<double, std::milli>(t3 - t2).count();
  return true;
}

}  // namespace orc
