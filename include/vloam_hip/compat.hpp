// vloam_hip/compat.hpp — the reference's C++ class surface on top of the libvloam_hip.so C ABI.
//
// Drop-in for the classes the façade drives:
//   vloam::ScanRegistration   scan_registration.h:71-77   init / reset / input / output
//   vloam::LaserOdometry      laser_odometry.h:70-84      init / input / solveLO / output
//   vloam::LaserMapping       laser_mapping.h:85-94       init / reset / input / solveMapping
//   vloam::LidarOdometryMapping  lidar_odometry_mapping.cpp:65-154  reset / scanRegistrationIO / laserOdometryIO / laserMappingIO
// Same method names, argument meaning and call order.  Clouds are a PCL-free POD vector by default; define
// VLOAM_HIP_WITH_PCL (and have PCL on the include path) to get overloads taking pcl::PointCloud — that adapter is
// compile-guarded and untested here because PCL / ROS are absent from this image.  Errors: the reference aborts
// (ROS_BREAK) or returns void; here a failing ABI call throws std::runtime_error carrying vloam_last_error().
#pragma once
#include <array>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "c_api.h"

#ifdef VLOAM_HIP_WITH_PCL
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#endif

namespace vloam {

struct PointXYZI {  // payload of pcl::PointXYZI (common.h:42); on input `intensity` is the pad word of pcl::PointXYZ
  float x, y, z, intensity;
};
typedef std::vector<PointXYZI> Cloud;
typedef std::array<double, 4> Quaterniond;  // (x, y, z, w) == Eigen::Quaterniond::coeffs()
typedef std::array<double, 3> Vector3d;

inline void check(vloam_status s) {
  if (s != VLOAM_OK) throw std::runtime_error(std::string("vloam_hip: ") + vloam_last_error());
}

class Session {  // one vloam_handle == one sequence on one GPU; shared by the three stage objects
 public:
  explicit Session(int device = 0, const vloam_config* cfg = nullptr) {
    vloam_config c;
    if (cfg) c = *cfg; else vloam_default_config(&c);
    check(vloam_create(&c, device, &h_));
  }
  ~Session() { vloam_destroy(h_); }
  Session(const Session&) = delete;
  Session& operator=(const Session&) = delete;
  vloam_handle* get() const { return h_; }
  Cloud features(int which) const {
    int n = 0;
    check(vloam_get_features(h_, which, nullptr, 0, &n));
    Cloud c(static_cast<size_t>(n));
    if (n) check(vloam_get_features(h_, which, &c[0].x, n, &n));
    return c;
  }

 private:
  vloam_handle* h_ = nullptr;
};

class ScanRegistration {
 public:
  explicit ScanRegistration(std::shared_ptr<Session> s) : s_(std::move(s)) {}
  void init() {}                                   // parameters were bound at vloam_create
  void reset() { check(vloam_reset_frame(s_->get())); }
  void input(const Cloud& laserCloudIn) { check(vloam_scan_registration(s_->get(), laserCloudIn.empty() ? nullptr : &laserCloudIn[0].x, (int)laserCloudIn.size())); }
#ifdef VLOAM_HIP_WITH_PCL
  void input(const pcl::PointCloud<pcl::PointXYZ>& in) {  // pcl::PointXYZ is 16 bytes: x, y, z, pad
    check(vloam_scan_registration(s_->get(), reinterpret_cast<const float*>(in.points.data()), (int)in.points.size()));
  }
#endif
  void publish() {}                                // ROS topics are out of scope
  void output(Cloud& laserCloud, Cloud& cornerPointsSharp, Cloud& cornerPointsLessSharp, Cloud& surfPointsFlat, Cloud& surfPointsLessFlat) {
    laserCloud = s_->features(0); cornerPointsSharp = s_->features(1); cornerPointsLessSharp = s_->features(2);
    surfPointsFlat = s_->features(3); surfPointsLessFlat = s_->features(4);
  }

 private:
  std::shared_ptr<Session> s_;
};

class LaserOdometry {
 public:
  explicit LaserOdometry(std::shared_ptr<Session> s) : s_(std::move(s)) {}
  void init() {}
  // the five clouds stay resident in HBM; the reference deep-copies them here (laser_odometry.cpp:141-145)
  void input(const Cloud&, const Cloud&, const Cloud&, const Cloud&, const Cloud&) {}
  void input() {}
  void setVOPrior(const Quaterniond& q, const Vector3d& t) { check(vloam_set_lo_prior(s_->get(), q.data(), t.data())); }  // vloam_tf->velo_last_VOT_velo_curr
  void solveLO() { check(vloam_laser_odometry(s_->get(), q_w_curr.data(), t_w_curr.data(), q_last_curr.data(), t_last_curr.data())); }
  void publish() {}
  void output(Quaterniond& q_w_curr_, Vector3d& t_w_curr_, Cloud& laserCloudCornerLast, Cloud& laserCloudSurfLast, Cloud& laserCloudFullRes, bool& skip_frame) {
    q_w_curr_ = q_w_curr; t_w_curr_ = t_w_curr;
    laserCloudCornerLast = s_->features(5); laserCloudSurfLast = s_->features(6); laserCloudFullRes = s_->features(0);
    skip_frame = false;  // mapping_skip_frame handling lives inside vloam_laser_mapping
  }
  Quaterniond q_w_curr{{0, 0, 0, 1}}, q_last_curr{{0, 0, 0, 1}};
  Vector3d t_w_curr{{0, 0, 0}}, t_last_curr{{0, 0, 0}};

 private:
  std::shared_ptr<Session> s_;
};

class LaserMapping {
 public:
  explicit LaserMapping(std::shared_ptr<Session> s) : s_(std::move(s)) {}
  void init() {}
  void reset() {}
  void input(const Cloud&, const Cloud&, const Cloud&, const Quaterniond&, const Vector3d&, const bool&) {}
  void input() {}
  void solveMapping() { check(vloam_laser_mapping(s_->get(), q_w_curr.data(), t_w_curr.data())); }
  void publish() {}
  Cloud registeredCloud() { return s_->features(11); }  // /velodyne_cloud_registered (laser_mapping.cpp:795-805)
  Quaterniond q_w_curr{{0, 0, 0, 1}};
  Vector3d t_w_curr{{0, 0, 0}};

 private:
  std::shared_ptr<Session> s_;
};

class LidarOdometryMapping {
 public:
  explicit LidarOdometryMapping(int device = 0, const vloam_config* cfg = nullptr)
      : session(std::make_shared<Session>(device, cfg)), scan_registration(session), laser_odometry(session), laser_mapping(session) {}
  void init() {}
  void reset() { scan_registration.reset(); laser_mapping.reset(); }
  void scanRegistrationIO(const Cloud& laserCloudIn) { scan_registration.input(laserCloudIn); }
  void laserOdometryIO() { laser_odometry.input(); laser_odometry.solveLO(); laser_odometry.publish(); }
  void laserMappingIO() { laser_mapping.input(); laser_mapping.solveMapping(); laser_mapping.publish(); }
  std::shared_ptr<Session> session;
  ScanRegistration scan_registration;
  LaserOdometry laser_odometry;
  LaserMapping laser_mapping;
};

}  // namespace vloam
