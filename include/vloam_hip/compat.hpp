// vloam_hip/compat.hpp — the reference's C++ class surface on top of the libvloam_hip.so C ABI.
//
// Drop-in for the classes the façade drives:
//   vloam::ScanRegistration   scan_registration.h:71-77   init / reset / input / output
//   vloam::LaserOdometry      laser_odometry.h:70-84      init / input / solveLO / output
//   vloam::LaserMapping       laser_mapping.h:85-94       init / reset / input / solveMapping
//   vloam::LidarOdometryMapping  lidar_odometry_mapping.cpp:65-154  reset / scanRegistrationIO / laserOdometryIO / laserMappingIO
//   vloam::VisualOdometry     visual_odometry.h:36-58     init / reset / processImage / setUpPointCloud / processPointCloud / solveNlsAll
//                             (optical_flow_match = true; shares a Session with the LiDAR stages or owns one)
// Same method names, argument meaning and call order; the stage classes are default-constructible like the reference's (they then share
// Session::get_default()), or take the Session they work on.  Clouds are a PCL-free POD vector by default; define
// VLOAM_HIP_WITH_PCL (and have PCL on the include path) to get overloads with the reference's own signatures — ScanRegistration::input
// taking pcl::PointCloud<pcl::PointXYZ>, and the pcl::PointCloud<PointType>::Ptr forms of ScanRegistration::output, LaserOdometry::input /
// output and LaserMapping::input; that adapter is compile-guarded and untested here because PCL / ROS are absent from this image.  Errors: the reference aborts
// (ROS_BREAK) or returns void; here a failing ABI call throws std::runtime_error carrying vloam_last_error().
//
// What stays in HBM: the reference hands clouds from stage to stage by value (LaserOdometry::input deep-copies the five
// scan-registration clouds, laser_odometry.cpp:141-145; LaserMapping::input copies three more, laser_mapping.cpp:167-181).
// Here the stages read each other's results on the device, so input(...) does not upload anything — but it CHECKS that the
// clouds it is given are the ones the previous stage produced (size + first / last point) and throws std::invalid_argument
// otherwise: a caller that edits or substitutes clouds between the stages is told so instead of being silently ignored.
// init(std::shared_ptr<TF>&) accepts (and ignores) the reference's vloam_tf blackboard: the pose hand-overs it carries are
// device-resident (vloam_set_lo_prior / vloam_process_frame are the VO coupling points).
#pragma once
#include <array>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "c_api.h"

#ifdef VLOAM_HIP_WITH_OPENCV
#include <opencv2/core.hpp>
#endif
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

inline bool same_cloud(const Cloud& a, const Cloud& b) {  // cheap identity check: size, first and last point bit for bit
  if (a.size() != b.size()) return false;
  if (a.empty()) return true;
  auto eq = [](const PointXYZI& p, const PointXYZI& q) { return p.x == q.x && p.y == q.y && p.z == q.z; };
  return eq(a.front(), b.front()) && eq(a.back(), b.back());
}

#ifdef VLOAM_HIP_WITH_PCL
// pcl::PointCloud<pcl::PointXYZI> <-> Cloud (the reference's PointType, common.h:42).  Compile-guarded and UNTESTED here: PCL is absent
// from this image.  The Ptr overloads below give the three stage classes the exact signatures of scan_registration.h:74-76,
// laser_odometry.h:76-84 and laser_mapping.h:88-91, so that a caller that drives them one by one (lidar_odometry_mapping.cpp:73-154) links
// unchanged.
typedef pcl::PointCloud<pcl::PointXYZI> PclCloud;
inline void to_pcl(const Cloud& c, PclCloud::Ptr& out) {
  if (!out) out.reset(new PclCloud());
  out->points.resize(c.size());
  for (size_t i = 0; i < c.size(); i++) { out->points[i].x = c[i].x; out->points[i].y = c[i].y; out->points[i].z = c[i].z; out->points[i].intensity = c[i].intensity; }
  out->width = static_cast<uint32_t>(c.size()); out->height = 1; out->is_dense = true;
}
inline Cloud from_pcl(const PclCloud::Ptr& in) {
  Cloud c(in ? in->points.size() : 0);
  for (size_t i = 0; i < c.size(); i++) { c[i].x = in->points[i].x; c[i].y = in->points[i].y; c[i].z = in->points[i].z; c[i].intensity = in->points[i].intensity; }
  return c;
}
#endif

class Session {  // one vloam_handle == one sequence on one GPU; shared by the three stage objects
 public:
  explicit Session(int device = 0, const vloam_config* cfg = nullptr) {
    if (cfg) config = *cfg; else vloam_default_config(&config);
    check(vloam_create(&config, device, &h_));
  }
  vloam_config config;
  int frames_done = 0;   // sweeps whose laser odometry has run == LaserOdometry::frameCount
  ~Session() { vloam_destroy(h_); }
  Session(const Session&) = delete;
  Session& operator=(const Session&) = delete;
  vloam_handle* get() const { return h_; }
  // The session behind DEFAULT-CONSTRUCTED stage objects (the reference's classes are default-constructible, laser_odometry.h:66-68, and
  // its façade holds them as plain members): created with the launch-file defaults on device 0 the first time one is needed; bind another
  // one (other device / config) with set_default() before the stage objects are constructed.  The slot itself is never destroyed (a
  // heap object that outlives static destruction): a process that forgets set_default(nullptr) leaves the handle to the operating system
  // instead of calling vloam_destroy after the HIP runtime's own static destructors have run; release it with set_default(nullptr) for a
  // clean shutdown.  Access is serialised (several threads may default-construct stage objects at once).
  static std::shared_ptr<Session>& default_slot() { static std::shared_ptr<Session>* s = new std::shared_ptr<Session>(); return *s; }
  static std::mutex& default_mutex() { static std::mutex* m = new std::mutex(); return *m; }
  static void set_default(std::shared_ptr<Session> s) { std::lock_guard<std::mutex> g(default_mutex()); default_slot() = std::move(s); }
  static std::shared_ptr<Session> get_default() {
    std::lock_guard<std::mutex> g(default_mutex());
    auto& s = default_slot();
    if (!s) s = std::make_shared<Session>();
    return s;
  }
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
  ScanRegistration() : s_(Session::get_default()) {}   // like the reference's: the stages of a process share Session::get_default()
  explicit ScanRegistration(std::shared_ptr<Session> s) : s_(std::move(s)) {}
  void init() {}                                   // parameters were bound at vloam_create
  template <class TF> void init(std::shared_ptr<TF>&) {}
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
#ifdef VLOAM_HIP_WITH_PCL
  void output(PclCloud::Ptr& laserCloud_, PclCloud::Ptr& cornerPointsSharp_, PclCloud::Ptr& cornerPointsLessSharp_, PclCloud::Ptr& surfPointsFlat_,
              PclCloud::Ptr& surfPointsLessFlat_) {   // scan_registration.h:75-77
    PclCloud::Ptr* out[5] = {&laserCloud_, &cornerPointsSharp_, &cornerPointsLessSharp_, &surfPointsFlat_, &surfPointsLessFlat_};
    for (int k = 0; k < 5; k++) to_pcl(s_->features(k), *out[k]);
  }
#endif

 private:
  std::shared_ptr<Session> s_;
};

class LaserOdometry {
 public:
  LaserOdometry() : s_(Session::get_default()) {}   // like the reference's: the stages of a process share Session::get_default()
  explicit LaserOdometry(std::shared_ptr<Session> s) : s_(std::move(s)) {}
  void init() {}
  template <class TF> void init(std::shared_ptr<TF>&) {}   // laser_odometry.h:70 takes the vloam_tf blackboard
  // laser_odometry.cpp:135-146.  The five clouds are already resident in HBM: nothing is uploaded, but foreign clouds are refused.
  void input(const Cloud& laserCloud, const Cloud& cornerPointsSharp, const Cloud& cornerPointsLessSharp, const Cloud& surfPointsFlat,
             const Cloud& surfPointsLessFlat) {
    const Cloud* given[5] = {&laserCloud, &cornerPointsSharp, &cornerPointsLessSharp, &surfPointsFlat, &surfPointsLessFlat};
    for (int k = 0; k < 5; k++)
      if (!same_cloud(*given[k], s_->features(k)))
        throw std::invalid_argument("vloam_hip: LaserOdometry::input was given a cloud that is not ScanRegistration::output's (the stages exchange "
                                    "clouds on the device; substituted clouds are not uploaded)");
  }
  void input() {}
#ifdef VLOAM_HIP_WITH_PCL
  void input(const PclCloud::Ptr& laserCloud_, const PclCloud::Ptr& cornerPointsSharp_, const PclCloud::Ptr& cornerPointsLessSharp_,
             const PclCloud::Ptr& surfPointsFlat_, const PclCloud::Ptr& surfPointsLessFlat_) {   // laser_odometry.h:76-80
    input(from_pcl(laserCloud_), from_pcl(cornerPointsSharp_), from_pcl(cornerPointsLessSharp_), from_pcl(surfPointsFlat_), from_pcl(surfPointsLessFlat_));
  }
  void output(Quaterniond& q_w_curr_, Vector3d& t_w_curr_, PclCloud::Ptr& laserCloudCornerLast_, PclCloud::Ptr& laserCloudSurfLast_,
              PclCloud::Ptr& laserCloudFullRes_, bool& skip_frame) {   // laser_odometry.h:82-84 (the pose as (x, y, z, w) / (x, y, z) arrays: Eigen is not a dependency of this header)
    Cloud a, b, c;
    output(q_w_curr_, t_w_curr_, a, b, c, skip_frame);
    to_pcl(a, laserCloudCornerLast_); to_pcl(b, laserCloudSurfLast_); to_pcl(c, laserCloudFullRes_);
  }
#endif
  void setVOPrior(const Quaterniond& q, const Vector3d& t) { check(vloam_set_lo_prior(s_->get(), q.data(), t.data())); }  // vloam_tf->velo_last_VOT_velo_curr
  void solveLO() {
    check(vloam_laser_odometry(s_->get(), q_w_curr.data(), t_w_curr.data(), q_last_curr.data(), t_last_curr.data()));
    s_->frames_done++;   // laser_odometry.cpp:535 frameCount++
  }
  void publish() {}
  void output(Quaterniond& q_w_curr_, Vector3d& t_w_curr_, Cloud& laserCloudCornerLast, Cloud& laserCloudSurfLast, Cloud& laserCloudFullRes, bool& skip_frame) {
    q_w_curr_ = q_w_curr; t_w_curr_ = t_w_curr;
    laserCloudCornerLast = s_->features(5); laserCloudSurfLast = s_->features(6); laserCloudFullRes = s_->features(0);
    skip_frame = (s_->frames_done % s_->config.mapping_skip_frame) != 0;  // laser_odometry.cpp:618 (vloam_laser_mapping applies the same rule)
  }
  Quaterniond q_w_curr{{0, 0, 0, 1}}, q_last_curr{{0, 0, 0, 1}};
  Vector3d t_w_curr{{0, 0, 0}}, t_last_curr{{0, 0, 0}};

 private:
  std::shared_ptr<Session> s_;
};

class LaserMapping {
 public:
  LaserMapping() : s_(Session::get_default()) {}   // like the reference's: the stages of a process share Session::get_default()
  explicit LaserMapping(std::shared_ptr<Session> s) : s_(std::move(s)) {}
  void init() {}
  template <class TF> void init(std::shared_ptr<TF>&) {}   // laser_mapping.h:85
  void reset() {}
  // laser_mapping.cpp:167-196: clouds and odometry pose are read on the device; what is passed in must be LaserOdometry::output's
  void input(const Cloud& laserCloudCornerLast, const Cloud& laserCloudSurfLast, const Cloud& laserCloudFullRes, const Quaterniond&, const Vector3d&,
             const bool& skip_frame) {
    if (!same_cloud(laserCloudCornerLast, s_->features(5)) || !same_cloud(laserCloudSurfLast, s_->features(6)) || !same_cloud(laserCloudFullRes, s_->features(0)))
      throw std::invalid_argument("vloam_hip: LaserMapping::input was given a cloud that is not LaserOdometry::output's");
    if (skip_frame != ((s_->frames_done % s_->config.mapping_skip_frame) != 0))
      throw std::invalid_argument("vloam_hip: LaserMapping::input: skip_frame differs from frameCount % mapping_skip_frame (laser_odometry.cpp:618)");
  }
  void input() {}
#ifdef VLOAM_HIP_WITH_PCL
  void input(const PclCloud::Ptr& laserCloudCornerLast_, const PclCloud::Ptr& laserCloudSurfLast_, const PclCloud::Ptr& laserCloudFullRes_,
             const Quaterniond& q_wodom_curr_, const Vector3d& t_wodom_curr_, const bool& skip_frame_) {   // laser_mapping.h:88-91
    input(from_pcl(laserCloudCornerLast_), from_pcl(laserCloudSurfLast_), from_pcl(laserCloudFullRes_), q_wodom_curr_, t_wodom_curr_, skip_frame_);
  }
#endif
  void solveMapping() { check(vloam_laser_mapping(s_->get(), q_w_curr.data(), t_w_curr.data())); }
  Cloud map() {   // /laser_cloud_map (laser_mapping.cpp:778-793)
    long long n = 0;
    check(vloam_get_map(s_->get(), nullptr, 0, &n));
    Cloud c(static_cast<size_t>(n));
    if (n) check(vloam_get_map(s_->get(), &c[0].x, n, &n));
    return c;
  }
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
  template <class TF> void init(std::shared_ptr<TF>&) {}   // lidar_odometry_mapping.h: init(std::shared_ptr<VloamTF>&)
  void reset() { scan_registration.reset(); laser_mapping.reset(); }
  void scanRegistrationIO(const Cloud& laserCloudIn) { scan_registration.input(laserCloudIn); }
#ifdef VLOAM_HIP_WITH_PCL
  void scanRegistrationIO(const pcl::PointCloud<pcl::PointXYZ>& laserCloudIn) { scan_registration.input(laserCloudIn); }  // lidar_odometry_mapping.cpp:73
#endif
  void laserOdometryIO() { laser_odometry.input(); laser_odometry.solveLO(); laser_odometry.publish(); }
  void laserMappingIO() { laser_mapping.input(); laser_mapping.solveMapping(); laser_mapping.publish(); }
  std::shared_ptr<Session> session;
  ScanRegistration scan_registration;
  LaserOdometry laser_odometry;
  LaserMapping laser_mapping;
};

// visual_odometry.h:36-58 in its optical-flow configuration.  processImage takes the 8-bit grey image as a pointer (cv::Mat::data /
// cols / rows / step); define VLOAM_HIP_WITH_OPENCV for the cv::Mat overload.  The session needs cfg.image_width / image_height.
// The reference reads its initial guess from the vloam_tf blackboard (cam0_curr_LOT_cam0_prev, visual_odometry.cpp:258-281): here the
// caller leaves it in angles_0to1 / t_0to1 before solveNlsAll (zeros = reset_VO_to_identity), which also returns the estimate there.
class VisualOdometry {
 public:
  explicit VisualOdometry(std::shared_ptr<Session> s) : s_(std::move(s)) {}
  void init() {}
  template <class TF> void init(std::shared_ptr<TF>&) {}   // visual_odometry.h:40 takes the vloam_tf blackboard
  void reset() { ++count; i = count % 2; }                    // visual_odometry.cpp:85-89
  void processImage(const unsigned char* gray, int width, int height, int stride) {   // visual_odometry.cpp:91-132
    check(vloam_vo_process_image(s_->get(), gray, width, height, stride));
    int n = 0;
    keypoints.assign(2 * 1024, 0.f);
    check(vloam_vo_get_keypoints(s_->get(), keypoints.data(), 1024, &n));
    keypoints.resize(2 * static_cast<size_t>(n));
    prev_uv.assign(2 * 1024, 0); curr_uv.assign(2 * 1024, 0);
    int m = 0;
    check(vloam_vo_get_flow_matches(s_->get(), prev_uv.data(), curr_uv.data(), 1024, &m));   // the match loop's pairs (:296-308)
    prev_uv.resize(2 * static_cast<size_t>(m)); curr_uv.resize(2 * static_cast<size_t>(m));
  }
#ifdef VLOAM_HIP_WITH_OPENCV
  void processImage(const cv::Mat& img00) { processImage(img00.data, img00.cols, img00.rows, static_cast<int>(img00.step)); }
#endif
  void setUpPointCloud(const vloam_calib& calib) { check(vloam_vo_set_calib(s_->get(), &calib)); }   // visual_odometry.cpp:134-155
  void processPointCloud(const Cloud& cloud) {                                                        // visual_odometry.cpp:157-186
    check(vloam_vo_process_point_cloud(s_->get(), cloud.empty() ? nullptr : &cloud[0].x, static_cast<int>(cloud.size())));
  }
  void solveNlsAll() {                                                                                 // visual_odometry.cpp:254-450
    int counters[2] = {0, 0};
    check(vloam_vo_solve(s_->get(), prev_uv.data(), curr_uv.data(), static_cast<int>(prev_uv.size() / 2), angles_0to1, t_0to1, counters));
    counter32 = counters[0]; counter22 = counters[1];
  }
  void publish() {}
  int i = 0, count = -1;
  std::vector<float> keypoints;        // (x, y) of keypoints[i], goodFeaturesToTrack order
  std::vector<int> prev_uv, curr_uv;   // integer pixel pairs of the tracked corners (previous image -> this image)
  double angles_0to1[3] = {0, 0, 0}, t_0to1[3] = {0, 0, 0};
  int counter32 = 0, counter22 = 0;

 private:
  std::shared_ptr<Session> s_;
};

}  // namespace vloam
