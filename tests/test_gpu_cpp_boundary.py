"""-m gpu: the C++ class surface (include/vloam_hip/compat.hpp) executed on the GPU.

A C++ program written against vloam::LidarOdometryMapping / ScanRegistration / LaserOdometry / LaserMapping exactly the way the
reference's façade drives them (lidar_odometry_mapping.cpp:65-154: reset -> scanRegistrationIO -> laserOdometryIO -> laserMappingIO,
with the explicit output() / input() hand-overs of scan_registration.h:71-77, laser_odometry.h:70-84, laser_mapping.h:85-94) is
compiled with g++ against libvloam_hip.so, fed synthetic sweeps from a file, and its poses / feature counts / skip flags are
compared with the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r'''
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>
#include "vloam_hip/compat.hpp"
struct FakeTF {};   // stands in for vloam::VloamTF: init(std::shared_ptr<VloamTF>&) must accept it
int main(int argc, char** argv) {
  const int n_sweeps = std::atoi(argv[2]), n_pts = std::atoi(argv[3]), skip = std::atoi(argv[4]);
  std::FILE* f = std::fopen(argv[1], "rb");
  vloam_config cfg; vloam_default_config(&cfg); cfg.mapping_skip_frame = skip;
  vloam::LidarOdometryMapping LOAM(0, &cfg);
  auto tf = std::make_shared<FakeTF>();
  LOAM.init(tf);
  for (int k = 0; k < n_sweeps; k++) {
    vloam::Cloud in((size_t)n_pts);
    if (std::fread(in.data(), sizeof(vloam::PointXYZI), (size_t)n_pts, f) != (size_t)n_pts) return 2;
    LOAM.reset();
    // the reference's laserOdometryIO / laserMappingIO spelled out (lidar_odometry_mapping.cpp:84-154)
    LOAM.scanRegistrationIO(in);
    vloam::Cloud full, sharp, lessSharp, flat, lessFlat;
    LOAM.scan_registration.output(full, sharp, lessSharp, flat, lessFlat);
    LOAM.laser_odometry.input(full, sharp, lessSharp, flat, lessFlat);
    LOAM.laser_odometry.solveLO();
    vloam::Quaterniond q; vloam::Vector3d t; vloam::Cloud cornerLast, surfLast, fullRes; bool skip_frame = false;
    LOAM.laser_odometry.output(q, t, cornerLast, surfLast, fullRes, skip_frame);
    LOAM.laser_mapping.input(cornerLast, surfLast, fullRes, q, t, skip_frame);
    LOAM.laser_mapping.solveMapping();
    std::printf("%d %zu %zu %zu %zu %zu %d", k, full.size(), sharp.size(), lessSharp.size(), flat.size(), lessFlat.size(), (int)skip_frame);
    for (int i = 0; i < 4; i++) std::printf(" %.17g", q[i]);
    for (int i = 0; i < 3; i++) std::printf(" %.17g", t[i]);
    // what LaserMapping::publish reports: q_w_curr after a mapped sweep, the high-frequency pose after a skipped one (laser_mapping.cpp:718-757)
    for (int i = 0; i < 4; i++) std::printf(" %.17g", skip_frame ? LOAM.laser_mapping.q_w_curr_highfreq[i] : LOAM.laser_mapping.q_w_curr[i]);
    for (int i = 0; i < 3; i++) std::printf(" %.17g", skip_frame ? LOAM.laser_mapping.t_w_curr_highfreq[i] : LOAM.laser_mapping.t_w_curr[i]);
    std::printf("\n");
    if (k == 1) {   // a substituted cloud at the wrong moment (the sweep's odometry has run) is an ORDER error, not silence
      vloam::Cloud bogus = sharp; bogus.pop_back();
      bool threw = false;
      try { LOAM.laser_odometry.input(full, bogus, lessSharp, flat, lessFlat); } catch (const std::runtime_error&) { threw = true; }
      if (!threw) return 3;
    }
  }
  std::printf("map %zu registered %zu\n", LOAM.laser_mapping.map().size(), LOAM.laser_mapping.registeredCloud().size());
  {   // default-constructed stage objects (laser_odometry.h:66-68) share Session::get_default(): a fresh sequence, first sweep again
    vloam::ScanRegistration sr; vloam::LaserOdometry lo; vloam::LaserMapping lm;
    sr.init(); lo.init(); lm.init();
    std::rewind(f);
    vloam::Cloud in((size_t)n_pts);
    if (std::fread(in.data(), sizeof(vloam::PointXYZI), (size_t)n_pts, f) != (size_t)n_pts) return 2;
    sr.input(in);
    vloam::Cloud full, sharp, lessSharp, flat, lessFlat;
    sr.output(full, sharp, lessSharp, flat, lessFlat);
    lo.input(full, sharp, lessSharp, flat, lessFlat);
    lo.solveLO();
    std::printf("default %zu %zu %zu %zu %zu\n", full.size(), sharp.size(), lessSharp.size(), flat.size(), lessFlat.size());
  }
  vloam::Session::set_default(nullptr);   // release the device before the runtime's own static destructors run
  return 0;
}
'''


def qdist(a, b):
    return min(np.linalg.norm(a - b), np.linalg.norm(a + b))


@pytest.mark.parametrize("skip", [1, 2])
def test_cpp_facade_runs_on_the_gpu_and_matches_the_oracle(tmp_path, orc, sweeps, vl, skip):
    n, shape = 5, (64, 512)
    clouds = [sweeps(shape[0], shape[1], k) for k in range(n)]
    data = tmp_path / "sweeps.bin"
    np.stack(clouds).astype(np.float32).tofile(data)
    src, exe = tmp_path / "probe.cpp", tmp_path / "probe"
    src.write_text(PROBE)
    libdir = os.path.join(ROOT, "vloam-cmu-16833_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lvloam_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe), str(data), str(n), str(clouds[0].shape[0]), str(skip)]).decode().strip().split("\n")
    assert len(out) == n + 2
    o = orc.Oracle(with_mapping=True, mapping_skip_frame=skip)
    for k in range(n):
        f = out[k].split()
        o.process(clouds[k])
        counts = [int(v) for v in f[1:6]]
        assert counts == [o.cloud(w).shape[0] for w in range(5)], "frame %d feature counts" % k
        assert int(f[6]) == (1 if ((k + 1) % skip) != 0 else 0), "skip_frame of LaserOdometry::output (laser_odometry.cpp:618)"
        v = np.array([float(x) for x in f[7:]])
        qw, tw, _, _ = o.lo_pose()
        qm, tm = o.map_published_pose()
        assert qdist(v[0:4], qw) < 1e-8 and np.linalg.norm(v[4:7] - tw) < 1e-8, "frame %d odometry pose" % k
        assert qdist(v[7:11], qm) < 1e-8 and np.linalg.norm(v[11:14] - tm) < 1e-8, "frame %d mapping pose" % k
    last = out[n].split()
    info = o.map_info()
    assert int(last[1]) == info["total_corner"] + info["total_surf"] and int(last[3]) == o.cloud(0).shape[0]
    # default-constructed stage objects: a fresh sequence on the process-wide default session, the first sweep's feature clouds again
    assert out[n + 1].split()[0] == "default" and out[n + 1].split()[1:] == out[0].split()[1:6]


VO_PROBE = r'''
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>
#include "vloam_hip/compat.hpp"
int main(int argc, char** argv) {
  const int n_frames = std::atoi(argv[3]), n_pts = std::atoi(argv[4]), W = std::atoi(argv[5]), H = std::atoi(argv[6]);
  std::FILE* fc = std::fopen(argv[1], "rb");
  std::FILE* fi = std::fopen(argv[2], "rb");
  vloam_config cfg; vloam_default_config(&cfg); cfg.with_mapping = 0; cfg.image_width = W; cfg.image_height = H;
  auto session = std::make_shared<vloam::Session>(0, &cfg);
  vloam::VisualOdometry VO(session);
  VO.init();
  vloam_calib calib;
  if (std::fread(&calib, sizeof(calib), 1, fc) != 1) return 2;
  VO.setUpPointCloud(calib);
  for (int k = 0; k < n_frames; k++) {
    vloam::Cloud cloud((size_t)n_pts);
    std::vector<unsigned char> img((size_t)W * H);
    if (std::fread(cloud.data(), sizeof(vloam::PointXYZI), (size_t)n_pts, fc) != (size_t)n_pts) return 2;
    if (std::fread(img.data(), 1, img.size(), fi) != img.size()) return 2;
    VO.reset();                          // the callback's order: vloam_main_node.cpp:133-160
    VO.processImage(img.data(), W, H, W);
    VO.processPointCloud(cloud);
    if (VO.count > 0) {
      for (int a = 0; a < 3; a++) { VO.angles_0to1[a] = 0; VO.t_0to1[a] = 0; }   // reset_VO_to_identity
      VO.solveNlsAll();
    }
    std::printf("%d %zu %zu %d %d", k, VO.keypoints.size() / 2, VO.prev_uv.size() / 2, VO.counter32, VO.counter22);
    for (int a = 0; a < 3; a++) std::printf(" %.17g", VO.angles_0to1[a]);
    for (int a = 0; a < 3; a++) std::printf(" %.17g", VO.t_0to1[a]);
    std::printf("\n");
  }
  return 0;
}
'''


def test_cpp_visual_odometry_class_from_images(tmp_path, orc, synth, vl):
    """vloam::VisualOdometry (compat.hpp) driven like the reference's callback: reset -> processImage -> processPointCloud -> solveNlsAll,
    image front-end + depth-enhanced VO on the device; keypoint / match counts, counters and the estimate against the oracle."""
    n, W, H = 3, 1242, 375
    seq = synth.SynthSequence(n_rings=64, n_azimuth=512, n_sweeps=n + 1)
    clouds = [seq.sweep(k) for k in range(n)]
    images = [synth.render_image(seq, k, W, H) for k in range(n)]
    cam_T_velo, rect0_T_cam, P = synth.kitti_like_calib()
    with open(tmp_path / "clouds.bin", "wb") as f:
        f.write(np.concatenate([cam_T_velo.reshape(-1), rect0_T_cam.reshape(-1), P.reshape(-1)]).astype(np.float32).tobytes())
        for c in clouds:
            f.write(np.ascontiguousarray(c, dtype=np.float32).tobytes())
    np.stack(images).tofile(tmp_path / "images.bin")
    src, exe = tmp_path / "vo_probe.cpp", tmp_path / "vo_probe"
    src.write_text(VO_PROBE)
    libdir = os.path.join(ROOT, "vloam-cmu-16833_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lvloam_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe), str(tmp_path / "clouds.bin"), str(tmp_path / "images.bin"), str(n), str(clouds[0].shape[0]), str(W), str(H)])
    rows = [l.split() for l in out.decode().strip().split("\n")]
    assert len(rows) == n
    vo = orc.VOOracle(cam_T_velo, rect0_T_cam, P)
    prev = None
    for k in range(n):
        vo.reset()
        corners = orc.good_features(images[k])
        vo.process_point_cloud(clouds[k])
        assert int(rows[k][1]) == corners.shape[0]
        if k == 0:
            assert int(rows[k][2]) == 0
        else:
            tracked, status = orc.pyr_lk(prev, images[k], corners)
            pu, cu = orc.flow_matches(corners, tracked, status)
            assert int(rows[k][2]) == pu.shape[0]
            r = vo.solve(pu, cu, np.zeros(3), np.zeros(3))
            assert (int(rows[k][3]), int(rows[k][4])) == (r["counter32"], r["counter22"])
            v = np.array([float(x) for x in rows[k][5:11]])
            assert np.linalg.norm(v[:3] - r["angles"]) < 1e-8 and np.linalg.norm(v[3:] - r["t"]) < 1e-8
        prev = images[k]


def test_reference_typed_facade_on_the_gpu(tmp_path, orc, sweeps, vl):
    """tests/cpp/ref_facade_probe.cpp: the façade of lidar_odometry_mapping.cpp:73-154 — pcl::PointCloud<PointType>::Ptr hand-overs,
    Eigen::Quaterniond / Vector3d poses, `if (!skip_frame) solveMapping(); publish();` — through compat.hpp's templates with the stand-in types of
    tests/stubs/, on the GPU, against the oracle driven stage by stage; at sweep 2 the caller thins surfPointsLessFlat before LaserOdometry::input (uploaded by
    vloam_set_odometry_input), at sweep 3 laserCloudCornerLast before LaserMapping::input (vloam_set_mapping_input)."""
    n, shape, skip = 6, (64, 512), 2
    clouds = [sweeps(shape[0], shape[1], k) for k in range(n)]
    data = tmp_path / "sweeps.bin"
    np.stack(clouds).astype(np.float32).tofile(data)
    exe = tmp_path / "ref_probe"
    libdir = os.path.join(ROOT, "vloam-cmu-16833_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "tests", "stubs"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "ref_facade_probe.cpp"), "-o", str(exe),
                           "-L", libdir, "-lvloam_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe), str(data), str(n), str(clouds[0].shape[0]), str(skip)]).decode().strip().split("\n")
    assert len(out) == n + 2, out
    o = orc.Oracle(with_mapping=True, mapping_skip_frame=skip)
    for k in range(n):
        f = out[k].split()
        assert o.stage_sr(clouds[k]) == 0
        if k == 2:   # the probe thins surfPointsLessFlat before LaserOdometry::input at sweep 2
            lf = o.cloud(4)
            o.set_sr_cloud(4, lf[np.arange(lf.shape[0]) % 3 != 1])
        counts = [o.cloud(w).shape[0] for w in range(5)]
        o.stage_lo()
        corner = o.cloud(5)
        thin = corner[::2] if k == 3 else None
        assert o.stage_map(corner=thin) == 0
        assert [int(v) for v in f[1:6]] == counts, "sweep %d feature counts" % k
        skipped = ((k + 1) % skip) != 0
        assert int(f[6]) == int(skipped)
        v = np.array([float(x) for x in f[7:21]])
        qw, tw, _, _ = o.lo_pose()
        qm, tm = o.map_published_pose()
        assert qdist(v[0:4], qw) < 1e-8 and np.linalg.norm(v[4:7] - tw) < 1e-8, "sweep %d odometry pose" % k
        assert qdist(v[7:11], qm) < 1e-8 and np.linalg.norm(v[11:14] - tm) < 1e-8, "sweep %d mapping pose" % k
        assert int(f[21]) == (thin.shape[0] if k == 3 else corner.shape[0]) and int(f[22]) == o.cloud(6).shape[0]
    info = o.map_info()
    assert int(out[n].split()[1]) == info["total_corner"] + info["total_surf"]
    assert out[n + 1].split() == ["vo", "1", "0"]
