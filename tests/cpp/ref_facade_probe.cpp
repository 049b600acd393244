// Test program: vloam_hip/compat.hpp + factors.hpp driven with the reference's OWN argument types — pcl::PointCloud<PointType>::Ptr clouds,
// Eigen::Quaterniond / Eigen::Vector3d poses, cv::Mat images, ceres::CostFunction* factories, tf2::Transform — as stand-ins from tests/stubs/
// (PCL, Eigen, OpenCV, Ceres and tf2 are absent from this image; the stand-ins only carry the member names the adapters are templated over).
// The façade below holds the members of lidar_odometry_mapping.h:52-75 and makes the calls of lidar_odometry_mapping.cpp:73-154 in their order
// with their argument lists; only the logging / timing lines are not here.
//   probe --selfcheck                         CPU only: conversions, pose accessors, Create() factories, Transform::as<tf2::Transform>()
//   probe <sweeps.bin> <n> <pts> <skip>       GPU: n sweeps through the façade, one line of results per sweep
#define VLOAM_HIP_WITH_PCL 1
#define VLOAM_HIP_WITH_OPENCV 1
#define VLOAM_HIP_WITH_CERES 1
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>
#include <Eigen/Dense>
#include <tf2/LinearMath/Transform.h>
#include "vloam_hip/compat.hpp"
#include "vloam_hip/factors.hpp"

typedef pcl::PointXYZI PointType;   // common.h:42

struct FakeTF {};   // stands in for vloam::VloamTF

class LidarOdometryMapping {   // lidar_odometry_mapping.h:40-80
 public:
  explicit LidarOdometryMapping(std::shared_ptr<vloam::Session> s) : scan_registration(s), laser_odometry(s), laser_mapping(s) {}
  void init(std::shared_ptr<FakeTF>& vloam_tf_) {
    scan_registration.init(vloam_tf_);
    laser_odometry.init(vloam_tf_);
    laser_mapping.init(vloam_tf_);
    laserCloud = boost::make_shared<pcl::PointCloud<PointType>>();
    cornerPointsSharp = boost::make_shared<pcl::PointCloud<PointType>>();
    cornerPointsLessSharp = boost::make_shared<pcl::PointCloud<PointType>>();
    surfPointsFlat = boost::make_shared<pcl::PointCloud<PointType>>();
    surfPointsLessFlat = boost::make_shared<pcl::PointCloud<PointType>>();
    // laserCloudCornerLast / SurfLast / FullRes deliberately left null here: the adapters must give a null Ptr an object
  }
  void reset() { scan_registration.reset(); laser_mapping.reset(); }
  void scanRegistrationIO(const pcl::PointCloud<pcl::PointXYZ>& laserCloudIn) {
    scan_registration.input(laserCloudIn);
    scan_registration.output(laserCloud, cornerPointsSharp, cornerPointsLessSharp, surfPointsFlat, surfPointsLessFlat);
  }
  void laserOdometryIO() {
    laser_odometry.input(laserCloud, cornerPointsSharp, cornerPointsLessSharp, surfPointsFlat, surfPointsLessFlat);
    laser_odometry.solveLO();
    laser_odometry.publish();
    laser_odometry.output(q_wodom_curr, t_wodom_curr, laserCloudCornerLast, laserCloudSurfLast, laserCloudFullRes, skip_frame);
  }
  void laserMappingIO() {
    laser_mapping.input(laserCloudCornerLast, laserCloudSurfLast, laserCloudFullRes, q_wodom_curr, t_wodom_curr, skip_frame);
    if (!skip_frame) laser_mapping.solveMapping();
    laser_mapping.publish();
  }
  vloam::ScanRegistration scan_registration;
  pcl::PointCloud<PointType>::Ptr laserCloud, cornerPointsSharp, cornerPointsLessSharp, surfPointsFlat, surfPointsLessFlat;
  vloam::LaserOdometry laser_odometry;
  Eigen::Quaterniond q_wodom_curr, q_w_curr;
  Eigen::Vector3d t_wodom_curr, t_w_curr;
  pcl::PointCloud<PointType>::Ptr laserCloudCornerLast, laserCloudSurfLast, laserCloudFullRes;
  bool skip_frame = false;
  vloam::LaserMapping laser_mapping;
};

static int selfcheck() {
  // cloud pointer <-> vloam::Cloud, a null pointer gets an object
  vloam::Cloud c(3);
  for (int i = 0; i < 3; i++) { c[i].x = 1.f + i; c[i].y = 2.f + i; c[i].z = 3.f + i; c[i].intensity = 10.f + i; }
  pcl::PointCloud<PointType>::Ptr p;
  vloam::to_cloud_ptr(c, p);
  if (!p || p->points.size() != 3 || p->width != 3 || p->height != 1 || !p->is_dense || p->points[2].intensity != 12.f || p->points[1].y != 3.f) return 1;
  const vloam::Cloud back = vloam::from_cloud_ptr(p);
  if (back.size() != 3 || std::memcmp(back.data(), c.data(), sizeof(vloam::PointXYZI) * 3) != 0) return 2;
  // pose accessors: Eigen-style classes and plain arrays
  const double qv[4] = {0.1, 0.2, 0.3, 0.9}, tv[3] = {1, 2, 3};
  Eigen::Quaterniond q; Eigen::Vector3d t;
  vloam::detail::quat_set(q, qv, 0); vloam::detail::vec_set(t, tv, 0);
  if (q.x() != 0.1 || q.w() != 0.9 || t.z() != 3) return 3;
  vloam::Quaterniond qa; vloam::Vector3d ta; double out[4];
  vloam::detail::quat_set(qa, qv, 0); vloam::detail::vec_set(ta, tv, 0);
  vloam::detail::quat_get(q, out, 0);
  if (qa[3] != 0.9 || ta[1] != 2 || out[2] != 0.3) return 4;
  // the Create() factories of lidarFactor.hpp:47-52,95-101,129-134 and ceres_cost_function.h:87-92,176-181 (Eigen::Vector3d call sites: laser_odometry.cpp:337-347)
  const Eigen::Vector3d cp(1.5, -2.0, 0.7), pa(1.0, 0.5, 0.2), pb(1.2, 0.1, 1.4), pm(-0.4, 2.0, 0.3), nrm(0.0, 0.6, 0.8);
  const double qq[4] = {0.01, -0.02, 0.03, 0.9993}, tt[3] = {0.8, -0.1, 0.05};
  const double* params[2] = {qq, tt};
  double r1[3], r2[3];
  std::unique_ptr<ceres::CostFunction> f1(vloam::factors::LidarEdgeFactor::Create(cp, pa, pb, 1.0));
  vloam::factors::LidarEdgeFactor e(cp, pa, pb, 1.0);
  f1->Evaluate(params, r1, nullptr); e(qq, tt, r2);
  if (std::memcmp(r1, r2, sizeof(r1)) != 0) return 5;
  std::unique_ptr<ceres::CostFunction> f2(vloam::factors::LidarPlaneFactor::Create(cp, pa, pb, pm, 1.0));
  vloam::factors::LidarPlaneFactor pl(cp, pa, pb, pm, 1.0);
  f2->Evaluate(params, r1, nullptr); pl(qq, tt, r2);
  if (r1[0] != r2[0]) return 6;
  std::unique_ptr<ceres::CostFunction> f3(vloam::factors::LidarPlaneNormFactor::Create(cp, nrm, 0.25));
  vloam::factors::LidarPlaneNormFactor pn(cp, nrm, 0.25);
  f3->Evaluate(params, r1, nullptr); pn(qq, tt, r2);
  if (r1[0] != r2[0]) return 7;
  const double aa[3] = {0.01, -0.02, 0.005};
  const double* vparams[2] = {aa, tt};
  std::unique_ptr<ceres::CostFunction> f4(vloam::factors::CostFunctor32::Create(0.1, -0.2, 7.5, 0.11, -0.19));
  vloam::factors::CostFunctor32 c32(0.1, -0.2, 7.5, 0.11, -0.19);
  f4->Evaluate(vparams, r1, nullptr); c32(aa, tt, r2);
  if (r1[0] != r2[0] || r1[1] != r2[1]) return 8;
  std::unique_ptr<ceres::CostFunction> f5(vloam::factors::CostFunctor22::Create(0.1, -0.2, 0.11, -0.19));
  vloam::factors::CostFunctor22 c22(0.1, -0.2, 0.11, -0.19);
  f5->Evaluate(vparams, r1, nullptr); c22(aa, tt, r2);
  if (r1[0] != r2[0]) return 9;
  // VisualOdometry::cam0_curr_T_cam0_last as the callback hands it on (vloam_main_node.cpp:160)
  vloam::Transform T;
  T.q[0] = 0.1; T.q[1] = 0.2; T.q[2] = 0.3; T.q[3] = 0.9; T.origin[0] = 4; T.origin[1] = 5; T.origin[2] = 6;
  const tf2::Transform t2 = T.as<tf2::Transform>();
  if (t2.getOrigin().getY() != 5 || t2.getRotation().getW() != 0.9 || t2.getRotation().getX() != 0.1) return 10;
  std::printf("selfcheck OK\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && std::strcmp(argv[1], "--selfcheck") == 0) return selfcheck();
  if (argc < 5) return 64;
  const int n_sweeps = std::atoi(argv[2]), n_pts = std::atoi(argv[3]), skip = std::atoi(argv[4]);
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 65;
  vloam_config cfg; vloam_default_config(&cfg); cfg.mapping_skip_frame = skip;
  auto session = std::make_shared<vloam::Session>(0, &cfg);
  LidarOdometryMapping LOAM(session);
  auto tf = std::make_shared<FakeTF>();
  LOAM.init(tf);
  pcl::PointCloud<pcl::PointXYZ> point_cloud_pcl;
  for (int k = 0; k < n_sweeps; k++) {
    point_cloud_pcl.points.resize((size_t)n_pts);
    if (std::fread(point_cloud_pcl.points.data(), sizeof(pcl::PointXYZ), (size_t)n_pts, f) != (size_t)n_pts) return 2;
    LOAM.reset();
    LOAM.scanRegistrationIO(point_cloud_pcl);   // vloam_main_node.cpp:166-168
    if (k == 2) {   // a caller that edits a hand-over: every third surfPointsLessFlat point dropped before LaserOdometry::input (uploaded: vloam_set_odometry_input)
      pcl::PointCloud<PointType>::Ptr thin = boost::make_shared<pcl::PointCloud<PointType>>();
      for (size_t i = 0; i < LOAM.surfPointsLessFlat->points.size(); i++) if (i % 3 != 1) thin->points.push_back(LOAM.surfPointsLessFlat->points[i]);
      LOAM.surfPointsLessFlat = thin;
    }
    LOAM.laserOdometryIO();
    if (k == 3) {   // a caller that edits a hand-over: every second corner point dropped before LaserMapping::input
      pcl::PointCloud<PointType>::Ptr thin = boost::make_shared<pcl::PointCloud<PointType>>();
      for (size_t i = 0; i < LOAM.laserCloudCornerLast->points.size(); i += 2) thin->points.push_back(LOAM.laserCloudCornerLast->points[i]);
      LOAM.laserCloudCornerLast = thin;
    }
    LOAM.laserMappingIO();
    std::printf("%d %zu %zu %zu %zu %zu %d", k, LOAM.laserCloud->points.size(), LOAM.cornerPointsSharp->points.size(), LOAM.cornerPointsLessSharp->points.size(),
                LOAM.surfPointsFlat->points.size(), LOAM.surfPointsLessFlat->points.size(), (int)LOAM.skip_frame);
    std::printf(" %.17g %.17g %.17g %.17g", LOAM.q_wodom_curr.x(), LOAM.q_wodom_curr.y(), LOAM.q_wodom_curr.z(), LOAM.q_wodom_curr.w());
    std::printf(" %.17g %.17g %.17g", LOAM.t_wodom_curr.x(), LOAM.t_wodom_curr.y(), LOAM.t_wodom_curr.z());
    const vloam::Quaterniond& qm = LOAM.skip_frame ? LOAM.laser_mapping.q_w_curr_highfreq : LOAM.laser_mapping.q_w_curr;
    const vloam::Vector3d& tm = LOAM.skip_frame ? LOAM.laser_mapping.t_w_curr_highfreq : LOAM.laser_mapping.t_w_curr;
    for (int i = 0; i < 4; i++) std::printf(" %.17g", qm[i]);
    for (int i = 0; i < 3; i++) std::printf(" %.17g", tm[i]);
    std::printf(" %zu %zu\n", LOAM.laserCloudCornerLast->points.size(), LOAM.laserCloudSurfLast->points.size());
  }
  std::printf("map %zu\n", LOAM.laser_mapping.map().size());
  {   // visual_odometry.h:36-96: default-constructible, processImage(const cv::Mat&), processPointCloud of the callback's pcl cloud
    vloam::Session::set_default(session);
    vloam::VisualOdometry VO;
    VO.init(tf);
    VO.reset();
    cv::Mat img;   // (the session has no image front-end: the call must fail with the library's error, not crash)
    std::vector<unsigned char> px(64 * 64, 0);
    img.data = px.data(); img.cols = 64; img.rows = 64; img.step.v = 64;
    bool threw = false;
    try { VO.processImage(img); } catch (const std::runtime_error&) { threw = true; }
    std::printf("vo %d %d\n", (int)threw, VO.count);
    vloam::Session::set_default(nullptr);
  }
  return 0;
}
