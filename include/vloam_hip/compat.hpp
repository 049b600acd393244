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
// Session::get_default()), or take the Session they work on.  Clouds are a PCL-free POD vector (vloam::Cloud) — AND every method that takes
// or fills clouds / poses in the reference is also a TEMPLATE over the caller's own types, so that the reference's signatures compile
// without this header including PCL or Eigen:
//   * cloud pointers: anything that dereferences to an object with a `points` vector of {x, y, z, intensity} plus width / height / is_dense
//     (pcl::PointCloud<PointType>::Ptr: scan_registration.h:75-77, laser_odometry.h:76-84, laser_mapping.h:88-91);
//   * ScanRegistration::input / scanRegistrationIO: any cloud object whose `points` are 16-byte {x, y, z, pad} (pcl::PointCloud<pcl::PointXYZ>);
//   * poses: any quaternion class with x() y() z() w() and vector class with x() y() z() (Eigen::Quaterniond / Eigen::Vector3d:
//     laser_odometry.h:82, laser_mapping.h:90-91), or anything indexable (std::array, double*).
// tests/test_cpp_compat_types.py instantiates every one of them with stand-in pcl / Eigen / cv / ceres / tf2 types (tests/stubs/), and
// tests/test_gpu_cpp_boundary.py runs the reference's façade sequence through them on the GPU.  VLOAM_HIP_WITH_PCL / VLOAM_HIP_WITH_OPENCV
// only add the includes of the real libraries.  Errors: the reference aborts (ROS_BREAK) or returns void; here a failing ABI call throws
// std::runtime_error carrying vloam_last_error().
//
// What stays in HBM: the reference hands clouds from stage to stage by value (LaserOdometry::input deep-copies the five
// scan-registration clouds, laser_odometry.cpp:141-145; LaserMapping::input copies three more, laser_mapping.cpp:167-181).
// Here the stages read each other's results on the device, so input(...) does not upload anything when it is handed what the previous
// stage produced (size + first / last point are compared).  A caller that edits or substitutes clouds between the stages gets the
// reference's semantics all the same: the foreign clouds are uploaded with vloam_set_stage_clouds and the stage works on them.
// init(std::shared_ptr<TF>&) accepts (and ignores) the reference's vloam_tf blackboard: the pose hand-overs it carries are
// device-resident (vloam_set_lo_prior / vloam_process_frame are the VO coupling points).
#pragma once
#include <array>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "c_api.h"

#include <cmath>
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

namespace detail {
// ---- pose classes: x() y() z() [w()] accessors (Eigen) are preferred, operator[] (std::array, pointers) otherwise
template <class Q> inline auto quat_set(Q& q, const double* v, int) -> decltype(q.w() = v[3], void()) { q.x() = v[0]; q.y() = v[1]; q.z() = v[2]; q.w() = v[3]; }
template <class Q> inline auto quat_set(Q& q, const double* v, long) -> decltype(q[3] = v[3], void()) { for (int i = 0; i < 4; i++) q[i] = v[i]; }
template <class V> inline auto vec_set(V& t, const double* v, int) -> decltype(t.z() = v[2], void()) { t.x() = v[0]; t.y() = v[1]; t.z() = v[2]; }
template <class V> inline auto vec_set(V& t, const double* v, long) -> decltype(t[2] = v[2], void()) { for (int i = 0; i < 3; i++) t[i] = v[i]; }
template <class Q> inline auto quat_get(const Q& q, double* v, int) -> decltype(double(q.w()), void()) { v[0] = q.x(); v[1] = q.y(); v[2] = q.z(); v[3] = q.w(); }
template <class Q> inline auto quat_get(const Q& q, double* v, long) -> decltype(double(q[3]), void()) { for (int i = 0; i < 4; i++) v[i] = q[i]; }
template <class V> inline auto vec_get(const V& t, double* v, int) -> decltype(double(t.z()), void()) { v[0] = t.x(); v[1] = t.y(); v[2] = t.z(); }
template <class V> inline auto vec_get(const V& t, double* v, long) -> decltype(double(t[2]), void()) { for (int i = 0; i < 3; i++) v[i] = t[i]; }
// ---- cloud pointers: boost::shared_ptr / std::shared_ptr of a pcl::PointCloud-like object
template <class P> using CloudPtrLike = typename std::enable_if<std::is_class<P>::value &&
    std::is_convertible<decltype((*std::declval<const P&>()).points.size()), size_t>::value &&
    std::is_convertible<decltype((*std::declval<const P&>()).points[0].intensity), float>::value, int>::type;
// ---- cloud objects with 16-byte points (pcl::PointCloud<pcl::PointXYZ>: x, y, z, pad)
template <class C> using XyzCloudLike = typename std::enable_if<std::is_class<C>::value &&
    sizeof(decltype(std::declval<const C&>().points[0])) == 16 &&
    std::is_convertible<decltype(std::declval<const C&>().points[0].x), float>::value, int>::type;
}  // namespace detail

inline void check(vloam_status s) {
  if (s != VLOAM_OK) throw std::runtime_error(std::string("vloam_hip: ") + vloam_last_error());
}

inline bool same_cloud(const Cloud& a, const Cloud& b) {  // cheap identity check: size, first and last point bit for bit
  if (a.size() != b.size()) return false;
  if (a.empty()) return true;
  auto eq = [](const PointXYZI& p, const PointXYZI& q) { return p.x == q.x && p.y == q.y && p.z == q.z; };
  return eq(a.front(), b.front()) && eq(a.back(), b.back());
}

// cloud pointer <-> Cloud (the reference's pcl::PointCloud<PointType>::Ptr, PointType = pcl::PointXYZI, common.h:42); a null pointer is
// given an object first, like the boost::make_shared calls of lidar_odometry_mapping.cpp:54-62
template <class P, detail::CloudPtrLike<P> = 0>
inline void to_cloud_ptr(const Cloud& c, P& out) {
  if (!out) out.reset(new typename P::element_type());
  out->points.resize(c.size());
  for (size_t i = 0; i < c.size(); i++) { out->points[i].x = c[i].x; out->points[i].y = c[i].y; out->points[i].z = c[i].z; out->points[i].intensity = c[i].intensity; }
  out->width = static_cast<decltype(out->width)>(c.size()); out->height = 1; out->is_dense = true;
}
template <class P, detail::CloudPtrLike<P> = 0>
inline Cloud from_cloud_ptr(const P& in) {
  Cloud c(in ? in->points.size() : 0);
  for (size_t i = 0; i < c.size(); i++) { c[i].x = in->points[i].x; c[i].y = in->points[i].y; c[i].z = in->points[i].z; c[i].intensity = in->points[i].intensity; }
  return c;
}

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
  // scan_registration.h:74: input(const pcl::PointCloud<pcl::PointXYZ>&) — pcl::PointXYZ is 16 bytes: x, y, z, pad
  template <class XyzCloud, detail::XyzCloudLike<XyzCloud> = 0>
  void input(const XyzCloud& in) {
    check(vloam_scan_registration(s_->get(), in.points.empty() ? nullptr : reinterpret_cast<const float*>(&in.points[0]), (int)in.points.size()));
  }
  void publish() {}                                // ROS topics are out of scope
  void output(Cloud& laserCloud, Cloud& cornerPointsSharp, Cloud& cornerPointsLessSharp, Cloud& surfPointsFlat, Cloud& surfPointsLessFlat) {
    laserCloud = s_->features(0); cornerPointsSharp = s_->features(1); cornerPointsLessSharp = s_->features(2);
    surfPointsFlat = s_->features(3); surfPointsLessFlat = s_->features(4);
  }
  template <class P, detail::CloudPtrLike<P> = 0>
  void output(P& laserCloud_, P& cornerPointsSharp_, P& cornerPointsLessSharp_, P& surfPointsFlat_, P& surfPointsLessFlat_) {   // scan_registration.h:75-77
    P* out[5] = {&laserCloud_, &cornerPointsSharp_, &cornerPointsLessSharp_, &surfPointsFlat_, &surfPointsLessFlat_};
    for (int k = 0; k < 5; k++) to_cloud_ptr(s_->features(k), *out[k]);
  }

 private:
  std::shared_ptr<Session> s_;
};

class LaserOdometry {
 public:
  LaserOdometry() : s_(Session::get_default()) {}   // like the reference's: the stages of a process share Session::get_default()
  explicit LaserOdometry(std::shared_ptr<Session> s) : s_(std::move(s)) {}
  void init() {}
  template <class TF> void init(std::shared_ptr<TF>&) {}   // laser_odometry.h:70 takes the vloam_tf blackboard
  // laser_odometry.cpp:135-146 deep-copies the five clouds it is handed.  What scan registration produced is already resident in HBM: such
  // clouds (size + first / last point equal) cost nothing; anything else is uploaded and the odometry of this sweep — and, for the two
  // less-clouds, of the next sweep, whose CornerLast / SurfLast they become (laser_odometry.cpp:506-526) — works on it.
  void input(const Cloud& laserCloud, const Cloud& cornerPointsSharp, const Cloud& cornerPointsLessSharp, const Cloud& surfPointsFlat,
             const Cloud& surfPointsLessFlat) {
    const Cloud* given[5] = {&laserCloud, &cornerPointsSharp, &cornerPointsLessSharp, &surfPointsFlat, &surfPointsLessFlat};
    const float* ptr[5]; int n[5]; bool any = false;
    for (int k = 0; k < 5; k++) {
      const bool same = same_cloud(*given[k], s_->features(k));
      ptr[k] = same ? nullptr : (given[k]->empty() ? &kEmptyPoint().x : &(*given[k])[0].x);   // null = keep the device's cloud
      n[k] = (int)given[k]->size();
      any = any || !same;
    }
    if (any) check(vloam_set_odometry_input(s_->get(), ptr[0], n[0], ptr[1], n[1], ptr[2], n[2], ptr[3], n[3], ptr[4], n[4]));
  }
  void input() {}
  template <class P, detail::CloudPtrLike<P> = 0>
  void input(const P& laserCloud_, const P& cornerPointsSharp_, const P& cornerPointsLessSharp_, const P& surfPointsFlat_, const P& surfPointsLessFlat_) {   // laser_odometry.h:76-80
    input(from_cloud_ptr(laserCloud_), from_cloud_ptr(cornerPointsSharp_), from_cloud_ptr(cornerPointsLessSharp_), from_cloud_ptr(surfPointsFlat_),
          from_cloud_ptr(surfPointsLessFlat_));
  }
  template <class Q, class V>
  void setVOPrior(const Q& q, const V& t) {   // vloam_tf->velo_last_VOT_velo_curr
    double qv[4], tv[3];
    detail::quat_get(q, qv, 0); detail::vec_get(t, tv, 0);
    check(vloam_set_lo_prior(s_->get(), qv, tv));
  }
  void solveLO() {
    check(vloam_laser_odometry(s_->get(), q_w_curr.data(), t_w_curr.data(), q_last_curr.data(), t_last_curr.data()));
    s_->frames_done++;   // laser_odometry.cpp:535 frameCount++
  }
  void publish() {}
  // laser_odometry.h:82-84: output(Eigen::Quaterniond&, Eigen::Vector3d&, Ptr&, Ptr&, Ptr&, bool&) — the pose classes are duck-typed
  // (x() y() z() w() or operator[]), the clouds are vloam::Cloud or cloud pointers
  template <class Q, class V>
  void output(Q& q_w_curr_, V& t_w_curr_, Cloud& laserCloudCornerLast, Cloud& laserCloudSurfLast, Cloud& laserCloudFullRes, bool& skip_frame) {
    detail::quat_set(q_w_curr_, q_w_curr.data(), 0); detail::vec_set(t_w_curr_, t_w_curr.data(), 0);
    laserCloudCornerLast = s_->features(5); laserCloudSurfLast = s_->features(6); laserCloudFullRes = s_->features(0);
    skip_frame = (s_->frames_done % s_->config.mapping_skip_frame) != 0;  // laser_odometry.cpp:618 (vloam_laser_mapping applies the same rule)
  }
  template <class Q, class V, class P, detail::CloudPtrLike<P> = 0>
  void output(Q& q_w_curr_, V& t_w_curr_, P& laserCloudCornerLast_, P& laserCloudSurfLast_, P& laserCloudFullRes_, bool& skip_frame) {
    Cloud a, b, c;
    output(q_w_curr_, t_w_curr_, a, b, c, skip_frame);
    to_cloud_ptr(a, laserCloudCornerLast_); to_cloud_ptr(b, laserCloudSurfLast_); to_cloud_ptr(c, laserCloudFullRes_);
  }
  Quaterniond q_w_curr{{0, 0, 0, 1}}, q_last_curr{{0, 0, 0, 1}};
  Vector3d t_w_curr{{0, 0, 0}}, t_last_curr{{0, 0, 0}};

 private:
  static const PointXYZI& kEmptyPoint() { static const PointXYZI p = {0.f, 0.f, 0.f, 0.f}; return p; }   // a non-null address for an empty substituted cloud
  std::shared_ptr<Session> s_;
};

class LaserMapping {
 public:
  LaserMapping() : s_(Session::get_default()) {}   // like the reference's: the stages of a process share Session::get_default()
  explicit LaserMapping(std::shared_ptr<Session> s) : s_(std::move(s)) {}
  void init() {}
  template <class TF> void init(std::shared_ptr<TF>&) {}   // laser_mapping.h:85
  void reset() {}
  // laser_mapping.cpp:167-196 copies the three clouds (unless skip_frame) and the odometry pose.  What LaserOdometry::output returned is read on
  // the device; a foreign cloud or pose is uploaded (vloam_set_mapping_input) and this sweep's mapping works on it — the odometry's own
  // CornerLast / SurfLast are untouched, like the reference's separate copies.
  template <class Q, class V>
  void input(const Cloud& laserCloudCornerLast, const Cloud& laserCloudSurfLast, const Cloud& laserCloudFullRes, const Q& q_wodom_curr_, const V& t_wodom_curr_,
             const bool& skip_frame) {
    if (skip_frame != ((s_->frames_done % s_->config.mapping_skip_frame) != 0))   // (the flag only ever comes from LaserOdometry::output; the device applies the same rule)
      throw std::invalid_argument("vloam_hip: LaserMapping::input: skip_frame differs from frameCount % mapping_skip_frame (laser_odometry.cpp:618)");
    skip_frame_ = skip_frame;
    const Cloud* given[3] = {&laserCloudCornerLast, &laserCloudSurfLast, &laserCloudFullRes};
    static const int which[3] = {5, 6, 0};
    const float* ptr[3] = {nullptr, nullptr, nullptr}; int n[3] = {0, 0, 0}; bool any = false;
    for (int k = 0; k < 3 && !skip_frame; k++) {
      const bool same = same_cloud(*given[k], s_->features(which[k]));
      ptr[k] = same ? nullptr : (given[k]->empty() ? &kEmptyPoint().x : &(*given[k])[0].x);
      n[k] = (int)given[k]->size();
      any = any || !same;
    }
    double q[4], t[3], qd[4], td[3];
    detail::quat_get(q_wodom_curr_, q, 0); detail::vec_get(t_wodom_curr_, t, 0);
    check(vloam_get_odometry_pose(s_->get(), qd, td));
    bool same_pose = true;
    for (int i = 0; i < 4; i++) same_pose = same_pose && q[i] == qd[i];
    for (int i = 0; i < 3; i++) same_pose = same_pose && t[i] == td[i];
    if (any || !same_pose) check(vloam_set_mapping_input(s_->get(), ptr[0], n[0], ptr[1], n[1], ptr[2], n[2], same_pose ? nullptr : q, same_pose ? nullptr : t));
  }
  void input() { skip_frame_ = (s_->frames_done % s_->config.mapping_skip_frame) != 0; }
  template <class P, class Q, class V, detail::CloudPtrLike<P> = 0>
  void input(const P& laserCloudCornerLast_, const P& laserCloudSurfLast_, const P& laserCloudFullRes_, const Q& q_wodom_curr_, const V& t_wodom_curr_,
             const bool& skip_frame) {   // laser_mapping.h:88-91
    input(from_cloud_ptr(laserCloudCornerLast_), from_cloud_ptr(laserCloudSurfLast_), from_cloud_ptr(laserCloudFullRes_), q_wodom_curr_, t_wodom_curr_, skip_frame);
  }
  // The reference's façade calls solveMapping() only `if (!skip_frame)` and publish() always (lidar_odometry_mapping.cpp:128-133); the
  // high-frequency pose of a skipped sweep comes out of publish() there.  vloam_laser_mapping does both (it applies the skip rule itself) and
  // closes the sweep, so either call order works: solveMapping() on every sweep, or the reference's conditional call followed by publish().
  void solveMapping() { run(); solved_ = true; }
  void publish() { if (!solved_) run(); solved_ = false; }
  Cloud map() {   // /laser_cloud_map (laser_mapping.cpp:778-793)
    long long n = 0;
    check(vloam_get_map(s_->get(), nullptr, 0, &n));
    Cloud c(static_cast<size_t>(n));
    if (n) check(vloam_get_map(s_->get(), &c[0].x, n, &n));
    return c;
  }
  Cloud registeredCloud() { return s_->features(11); }  // /velodyne_cloud_registered (laser_mapping.cpp:795-805)
  Quaterniond q_w_curr{{0, 0, 0, 1}}, q_w_curr_highfreq{{0, 0, 0, 1}};   // laser_mapping.h:141-155: a skipped sweep only moves the high-frequency pose
  Vector3d t_w_curr{{0, 0, 0}}, t_w_curr_highfreq{{0, 0, 0}};

 private:
  static const PointXYZI& kEmptyPoint() { static const PointXYZI p = {0.f, 0.f, 0.f, 0.f}; return p; }
  void run() {
    Quaterniond q; Vector3d t;
    check(vloam_laser_mapping(s_->get(), q.data(), t.data()));
    if (skip_frame_) { q_w_curr_highfreq = q; t_w_curr_highfreq = t; } else { q_w_curr = q; t_w_curr = t; q_w_curr_highfreq = q; t_w_curr_highfreq = t; }
  }
  std::shared_ptr<Session> s_;
  bool skip_frame_ = false, solved_ = false;
};

class LidarOdometryMapping {
 public:
  explicit LidarOdometryMapping(int device = 0, const vloam_config* cfg = nullptr)
      : session(std::make_shared<Session>(device, cfg)), scan_registration(session), laser_odometry(session), laser_mapping(session) {}
  void init() {}
  template <class TF> void init(std::shared_ptr<TF>&) {}   // lidar_odometry_mapping.h: init(std::shared_ptr<VloamTF>&)
  void reset() { scan_registration.reset(); laser_mapping.reset(); }
  void scanRegistrationIO(const Cloud& laserCloudIn) { scan_registration.input(laserCloudIn); }
  template <class XyzCloud, detail::XyzCloudLike<XyzCloud> = 0>
  void scanRegistrationIO(const XyzCloud& laserCloudIn) { scan_registration.input(laserCloudIn); }  // lidar_odometry_mapping.cpp:73: const pcl::PointCloud<pcl::PointXYZ>&
  void laserOdometryIO() { laser_odometry.input(); laser_odometry.solveLO(); laser_odometry.publish(); }
  void laserMappingIO() { laser_mapping.input(); laser_mapping.solveMapping(); laser_mapping.publish(); }
  std::shared_ptr<Session> session;
  ScanRegistration scan_registration;
  LaserOdometry laser_odometry;
  LaserMapping laser_mapping;
};

// tf2::Transform as far as the callback reads it (vloam_main_node.cpp:160 hands VO->cam0_curr_T_cam0_last to VloamTF::VO2VeloAndBase):
// rotation as a quaternion (x, y, z, w) + origin; as<tf2::Transform>() builds the caller's own class through setOrigin / setRotation.
struct Transform {
  double q[4] = {0, 0, 0, 1}, origin[3] = {0, 0, 0};
  template <class T> T as() const {
    T out;
    auto o = out.getOrigin(); o.setValue(origin[0], origin[1], origin[2]); out.setOrigin(o);
    auto r = out.getRotation(); r.setValue(q[0], q[1], q[2], q[3]); out.setRotation(r);
    return out;
  }
};

// visual_odometry.h:36-96, optical-flow configuration or (setOrbPattern) the ORB + brute-force one.  processImage takes the 8-bit grey image as a pointer (cv::Mat::data /
// cols / rows / step) or any matrix class with those members (cv::Mat).  The session needs cfg.image_width / image_height.
// The reference reads its initial guess from the vloam_tf blackboard (cam0_curr_LOT_cam0_prev, visual_odometry.cpp:258-281): here the
// caller leaves it in angles_0to1 / t_0to1 before solveNlsAll (zeros = reset_VO_to_identity), which also returns the estimate there and —
// like visual_odometry.cpp:425-430 — as the transform cam0_curr_T_cam0_last.
class VisualOdometry {
 public:
  VisualOdometry() : s_(Session::get_default()) {}            // visual_odometry.h:39
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
  template <class Mat> auto processImage(const Mat& img00) -> decltype(img00.data, img00.cols, img00.rows, void()) {   // visual_odometry.h:44: const cv::Mat&
    processImage(img00.data, img00.cols, img00.rows, static_cast<int>(img00.step));
  }
  // optical_flow_match = false (the launch default, vloam_main.launch:10; read at visual_odometry.cpp:50-52): ORB descriptors + brute-force matches
  // instead of optical flow.  OpenCV's sampling table (orb.cpp: bit_pattern_31_, 256 x (x0, y0, x1, y1)) is library data the caller hands in.
  void setOrbPattern(const signed char* bit_pattern_31_256x4) { check(vloam_vo_set_orb_pattern(s_->get(), bit_pattern_31_256x4)); }
  void setUpPointCloud(const vloam_calib& calib) { check(vloam_vo_set_calib(s_->get(), &calib)); }   // visual_odometry.cpp:134-155
  void processPointCloud(const Cloud& cloud) {                                                        // visual_odometry.cpp:157-186
    check(vloam_vo_process_point_cloud(s_->get(), cloud.empty() ? nullptr : &cloud[0].x, static_cast<int>(cloud.size())));
  }
  template <class XyzCloud, detail::XyzCloudLike<XyzCloud> = 0>
  void processPointCloud(const XyzCloud& point_cloud_pcl) {                                           // the pcl::PointCloud<pcl::PointXYZ> the callback converts (vloam_main_node.cpp:147-150)
    check(vloam_vo_process_point_cloud(s_->get(), point_cloud_pcl.points.empty() ? nullptr : reinterpret_cast<const float*>(&point_cloud_pcl.points[0]),
                                       static_cast<int>(point_cloud_pcl.points.size())));
  }
  void solveNlsAll() {                                                                                 // visual_odometry.cpp:254-450
    int counters[2] = {0, 0};
    check(vloam_vo_solve(s_->get(), prev_uv.data(), curr_uv.data(), static_cast<int>(prev_uv.size() / 2), angles_0to1, t_0to1, counters));
    counter32 = counters[0]; counter22 = counters[1];
    // visual_odometry.cpp:425-430: origin = t_0to1, rotation = Quaternion(axis = angles / |angles|, |angles|) — tf2::Quaternion::setRotation:
    // s = sin(angle / 2) / |axis| with a unit axis.  A zero angle divides by zero in the reference (NaN transform); the same arithmetic here.
    angle = std::sqrt(angles_0to1[0] * angles_0to1[0] + angles_0to1[1] * angles_0to1[1] + angles_0to1[2] * angles_0to1[2]);
    const double ax[3] = {angles_0to1[0] / angle, angles_0to1[1] / angle, angles_0to1[2] / angle};
    const double d = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]), sn = std::sin(angle * 0.5) / d;
    cam0_curr_T_cam0_last.q[0] = ax[0] * sn; cam0_curr_T_cam0_last.q[1] = ax[1] * sn; cam0_curr_T_cam0_last.q[2] = ax[2] * sn;
    cam0_curr_T_cam0_last.q[3] = std::cos(angle * 0.5);
    for (int a = 0; a < 3; a++) cam0_curr_T_cam0_last.origin[a] = t_0to1[a];
  }
  void publish() {}
  int i = 0, count = -1;
  std::vector<float> keypoints;        // (x, y) of keypoints[i], goodFeaturesToTrack order
  std::vector<int> prev_uv, curr_uv;   // integer pixel pairs of the tracked corners (previous image -> this image)
  double angles_0to1[3] = {0, 0, 0}, t_0to1[3] = {0, 0, 0};
  int counter32 = 0, counter22 = 0;
  double angle = 0;                    // visual_odometry.h:95
  Transform cam0_curr_T_cam0_last;     // visual_odometry.h:96 (identity before the first solve)

 private:
  std::shared_ptr<Session> s_;
};

}  // namespace vloam
