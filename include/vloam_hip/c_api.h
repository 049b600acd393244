/* libvloam_hip.so — C ABI of the MI355X-native (gfx950, HIP) VLOAM per-scan odometry hot path.
 *
 * The reference (YukunXia/VLOAM-CMU-16833) has no FFI layer; the seam this ABI replaces is the C++
 * class surface its façade drives plus the Ceres cost-functor surface:
 *   vloam::ScanRegistration   src/lidar_odometry_mapping/include/lidar_odometry_mapping/scan_registration.h:71-77
 *   vloam::LaserOdometry      src/lidar_odometry_mapping/include/lidar_odometry_mapping/laser_odometry.h:70-84
 *   vloam::LaserMapping       src/lidar_odometry_mapping/include/lidar_odometry_mapping/laser_mapping.h:85-94
 *   façade call order         src/lidar_odometry_mapping/src/lidar_odometry_mapping.cpp:65-154
 *   VO residual stack         src/visual_odometry/src/visual_odometry.cpp:157-186,254-450
 * include/vloam_hip/compat.hpp presents those classes on top of this ABI; INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Conventions: plain C, opaque handle, caller-owned buffers, int status (0 = OK, < 0 = error), no
 * exceptions across the boundary.  One handle = one sequence on one HIP device (or n_sessions sequences advanced in lock
 * step, vloam_create_batch), with its own HIP streams.  A handle is not thread-safe; distinct handles are independent.  Quaternions are (x, y, z, w) — the layout of
 * the reference's para_q / parameters arrays (laser_odometry.h:126-130, laser_mapping.h:141-143).
 * Clouds are packed float4: (x, y, z, pad) on input == pcl::PointXYZ, (x, y, z, intensity) on
 * output == the payload of pcl::PointXYZI (common.h:42).
 */
#ifndef VLOAM_HIP_C_API_H
#define VLOAM_HIP_C_API_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vloam_handle vloam_handle;
typedef int vloam_status;

enum {
  VLOAM_OK = 0,
  VLOAM_ERR_INVALID = -1,     /* bad argument / unsupported scan_line (the reference ROS_BREAK()s) */
  VLOAM_ERR_HIP = -2,         /* a HIP runtime call failed; vloam_last_error() has the text */
  VLOAM_ERR_CAPACITY = -3,    /* more points / ring length / map entries than the handle was sized for */
  VLOAM_ERR_EMPTY = -4,       /* no point survived NaN / minimum_range removal (reference: UB, scan_registration.cpp:166) */
  VLOAM_ERR_NO_DEVICE = -5,   /* no usable gfx950 device: there is NO CPU fallback */
  VLOAM_ERR_ORDER = -6        /* stage called out of the façade's order */
};

/* Tunables = the ROS parameters the reference reads in its init() functions:
 * LOM/launch/loam_velodyne_HDL_64_kitti.launch:3-16, MAIN/launch/vloam_main.launch:4-10. */
typedef struct vloam_config {
  int scan_line;                   /* 16 | 32 | 64                                   (64)   */
  double minimum_range;            /* removeClosedPointCloud threshold [m]           (5.0)  */
  int mapping_skip_frame;          /*                                                (1)    */
  float mapping_line_resolution;   /* corner VoxelGrid leaf [m]                      (0.4)  */
  float mapping_plane_resolution;  /* surf VoxelGrid leaf [m]                        (0.8)  */
  int detach_VO_LO;                /* 1: LO warm-starts from its own last estimate   (1)    */
  int reset_VO_to_identity;        /*                                                (0)    */
  int remove_VO_outlier;           /* pixel gate on matches                          (100)  */
  int with_mapping;                /* run laserMapping inside vloam_process_scan     (1)    */
  int max_points;                  /* capacity of one sweep                          (262144) */
  int max_frames;                  /* capacity of the on-device trajectory log       (8192) */
  int map_capacity_log2;           /* voxel-hash slots = 2^n per feature kind        (22)   */
  int debug;                       /* 1: keep parity-hook arrays (curvature, sort order, …) */
  int timing;                      /* 1: HIP-event per-stage timing (synchronises every sweep)   */
  int image_width;                 /* capacity of the image front-end (KITTI: 1242 x 375); 0 = none (0) */
  int image_height;
  int CLAHE;                       /* 1: cv::createCLAHE(2.0)->apply on every image first (vloam_main.launch:8)  (0) */
} vloam_config;

typedef struct vloam_calib {  /* row-major f32, as PointCloudUtil holds them (point_cloud_util.h:43-46) */
  float cam_T_velo[16];
  float rect0_T_cam[16];
  float P_rect0[12];
} vloam_calib;

void vloam_default_config(vloam_config* cfg);
const char* vloam_last_error(void);
const char* vloam_version(void);

vloam_status vloam_create(const vloam_config* cfg, int device, vloam_handle** out);
vloam_status vloam_destroy(vloam_handle* h);

/* ---- Batched execution: one handle, n_sessions independent sequences advanced in lock step.  Every kernel of the sweep chain is
 * launched once per sweep for ALL sessions (session index in blockIdx.z), so n_sessions sequences cost one launch chain — the way to
 * fill the chip with a path whose single-sequence form is a latency chain.  Each session owns an identical arena of
 * device state; integer / index / f32 point results (feature clouds, picks, down-sampled scan features, image key points) are bit-identical
 * to running the sequence alone, the f64 poses agree to round-off (~1e-13: a batch adds the partial sums of its solves over 4 / 6
 * workgroups, a single sequence over 8).  vloam_create == vloam_create_batch(…, 1, …).
 * vloam_batch_process_scan[_device]: session b gets sweep xyz_pad4[b] with n[b] points (arrays of n_sessions entries).
 * vloam_select_session: which session the getters below (trajectory, features, counts, map, parity hooks) read; 0 after creation.
 * The single-sequence entry points (vloam_scan_registration*, vloam_laser_*, vloam_process_scan*, vloam_process_frame*) return
 * VLOAM_ERR_INVALID on a handle with more than one session.
 * Co-residency bound: the Levenberg-Marquardt solves of a sweep run as cooperating workgroups (batch: 4 for the odometry, 6 for the
 * mapping; a single sequence: 8 + 8, all eight of a solve on ONE XCD — odometry on XCD 2, mapping on XCD 6 — for the first two
 * single-sequence handles alive in a process, spread over the XCDs for further ones; one compute unit's worth of registers each) that
 * exchange partial sums INSIDE one launch, so all workgroups of a solve must be
 * resident together; a session of a batch can have one odometry and one mapping solve in flight, i.e. 10 such workgroups.  vloam_create_batch
 * refuses (VLOAM_ERR_CAPACITY) when 10 * n_sessions exceeds the device's compute-unit count (256 on MI355X, so n_sessions <= 24 — the
 * library's own bound, kMaxBatch — is always accepted there); several batched handles on ONE device share that budget — keep the sum of their sessions within it.  The one-XCD
 * placement of a single sequence's solves holds at most 4 solves per XCD (32 compute units, 8 per solve).  When the workgroups of a solve
 * are NOT resident together after all (another process on the GPU, a CU-masked queue, a partitioned device), the solve does not fail: a
 * workgroup that still waits for its partners after ~0.5 s gives up, the lead workgroup runs the whole solve on its own, vloam_get_health
 * counts it, and from the next ENQUEUE on (the host polls a host-mapped word, no synchronisation needed) the handle launches one-workgroup
 * solves only (slower per solve, no co-residency needed). */
vloam_status vloam_create_batch(const vloam_config* cfg, int device, int n_sessions, vloam_handle** out);
vloam_status vloam_batch_size(vloam_handle* h, int* n_sessions);
vloam_status vloam_batch_process_scan_device(vloam_handle* h, const void* const* d_xyz_pad4, const int* n);
vloam_status vloam_batch_process_scan(vloam_handle* h, const float* const* xyz_pad4, const int* n);
vloam_status vloam_select_session(vloam_handle* h, int session);
/* one coupled VLOAM frame (vloam_process_frame_device, below) for every session: prev_uv[b] / curr_uv[b] = session b's n_match[b] pixel
 * matches in host memory; vloam_vo_set_calib / vloam_set_extrinsics apply to all sessions (one sensor rig) */
vloam_status vloam_batch_process_frame_device(vloam_handle* h, const void* const* d_xyz_pad4, const int* n, const int* const* prev_uv,
                                              const int* const* curr_uv, const int* n_match);
vloam_status vloam_batch_process_frame(vloam_handle* h, const float* const* xyz_pad4, const int* n, const int* const* prev_uv,
                                       const int* const* curr_uv, const int* n_match);   /* sweeps in host memory */

/* == LidarOdometryMapping::reset (lidar_odometry_mapping.cpp:65-71) */
vloam_status vloam_reset_frame(vloam_handle* h);

/* == ScanRegistration::input (scan_registration.cpp:131-449).  xyz_pad4: n packed float4 in HOST memory. */
vloam_status vloam_scan_registration(vloam_handle* h, const float* xyz_pad4, int n);
/* same, input already resident in this device's HBM */
vloam_status vloam_scan_registration_device(vloam_handle* h, const void* d_xyz_pad4, int n);

/* == ScanRegistration::output (scan_registration.cpp:501-512).
 * which: 0 laserCloud, 1 cornerPointsSharp, 2 cornerPointsLessSharp, 3 surfPointsFlat, 4 surfPointsLessFlat,
 *        5 laserCloudCornerLast, 6 laserCloudSurfLast (LaserOdometry::output, laser_odometry.cpp:610-629),
 *        7 laserCloudCornerStack, 8 laserCloudSurfStack (down-sampled scan features inside mapping),
 *        11 full-resolution cloud registered in the map frame (LaserMapping::publish, laser_mapping.cpp:795-799).
 * Copies min(n, cap) points to the HOST buffer xyzi4 and returns the true count in *n. Synchronises the stream. */
vloam_status vloam_get_features(vloam_handle* h, int which, float* xyzi4, int cap, int* n);

/* == the /laser_cloud_map product of LaserMapping::publish (laser_mapping.cpp:778-793): the corner cloud then the surf cloud of every
 * cube of the 21 x 21 x 11 window, cube index ascending, each cube cloud in its VoxelGrid order.  The reference publishes it every
 * map_pub_number frames; here the caller decides when to ask.  Copies min(*n, cap) points to the HOST buffer, true count in *n. */
vloam_status vloam_get_map(vloam_handle* h, float* xyzi4, long long cap, long long* n);

/* == vloam_tf->velo_last_VOT_velo_curr, read by solveLO when detach_VO_LO == 0 (laser_odometry.cpp:223-236) */
vloam_status vloam_set_lo_prior(vloam_handle* h, const double q_xyzw[4], const double t[3]);

/* == LaserOdometry::input + solveLO + output (laser_odometry.cpp:135-146,187-536,610-629).
 * Outputs (any may be NULL): world pose q_w_curr/t_w_curr and the frame-to-frame q_last_curr/t_last_curr. */
vloam_status vloam_laser_odometry(vloam_handle* h, double q_w[4], double t_w[3], double q_lc[4], double t_lc[3]);

/* LaserOdometry::input with clouds that are NOT the ones vloam_scan_registration left on the device — the reference deep-copies whatever it is
 * handed (laser_odometry.cpp:135-146), so a caller may edit or replace the five clouds between the stages.  Call between
 * vloam_scan_registration and vloam_laser_odometry; host pointers to packed (x, y, z, intensity) floats, intensity = scan line +
 * SCAN_PERIOD * relTime as scan registration writes it (scan_registration.cpp:262-264); a NULL cloud keeps the device's.  The clouds replace
 * the sweep's own: this sweep's odometry, the next sweep's CornerLast / SurfLast (laser_odometry.cpp:506-526) and the mapping stage's input
 * all see them.  Capacity: 768 / 7 680 / 1 536 points for cornerPointsSharp / LessSharp / surfPointsFlat, max_points for the other two.
 * Stated limit: the two less-clouds must keep scan registration's ordering (scan lines ascending up to the r / r - 1 jitter of
 * int(intensity)); the adjacent-line walks of laser_odometry.cpp:294-324,371-428 are evaluated from per-line first / last indices.
 * Both stage-input calls are for single-sequence handles driven stage by stage (VLOAM_ERR_INVALID on n_sessions > 1: a batch is enqueued whole). */
vloam_status vloam_set_odometry_input(vloam_handle* h, const float* laserCloud, int n_full, const float* cornerPointsSharp, int n_sharp,
                                      const float* cornerPointsLessSharp, int n_less_sharp, const float* surfPointsFlat, int n_flat,
                                      const float* surfPointsLessFlat, int n_less_flat);

/* q_w_curr / t_w_curr of the sweep in progress (after vloam_laser_odometry) or of the last finished sweep: what LaserOdometry::output hands to
 * LaserMapping::input (laser_odometry.cpp:610-616). */
vloam_status vloam_get_odometry_pose(vloam_handle* h, double q_w[4], double t_w[3]);

/* LaserMapping::input with clouds / an odometry pose that are NOT LaserOdometry::output's (laser_mapping.cpp:167-196 copies what it is handed; on a
 * sweep skipped by mapping_skip_frame only the pose).  Call between vloam_laser_odometry and vloam_laser_mapping; NULL keeps the device's.  Only
 * this sweep's mapping (and the /velodyne_cloud_registered product, vloam_get_features(11)) sees them — the odometry keeps its own
 * CornerLast / SurfLast, like the reference's separate copies.  Stated limit: q_wodom_curr must be a unit quaternion (as any
 * Eigen::Quaterniond an odometry produces): the solver's closed-form Jacobians are those of a rotation, the reference's autodiff
 * differentiates Eigen's un-normalised q * v — with |q|^2 = 1 + 5e-6 the map poses part by 1e-9 (measured, tests/test_gpu_stage_inputs.py). */
vloam_status vloam_set_mapping_input(vloam_handle* h, const float* laserCloudCornerLast, int n_corner, const float* laserCloudSurfLast, int n_surf,
                                     const float* laserCloudFullRes, int n_full, const double q_wodom_curr[4], const double t_wodom_curr[3]);

/* == LaserMapping::input + solveMapping + the pose of publish() (laser_mapping.cpp:167-196,198-708,718-757).  Outputs the
 * map-frame pose publish() reports: q_w_curr / t_w_curr after a mapped sweep, the high-frequency pose
 * q_wmap_wodom * q_wodom_curr after a sweep skipped by mapping_skip_frame. */
vloam_status vloam_laser_mapping(vloam_handle* h, double q_map[4], double t_map[3]);

/* Whole façade for one sweep, enqueued with NO host synchronisation:
 * reset -> scanRegistrationIO -> laserOdometryIO -> laserMappingIO (MAIN/src/vloam_main_node.cpp:134,166-168).
 * d_xyz_pad4 is DEVICE memory and must stay valid until vloam_sync().  Poses go to the on-device trajectory log.
 * The odometry / mapping of a sweep may only be ENQUEUED by a later call (they trail the scan registration by one / two
 * sweeps so that no live cross-stream wait is needed); every call that returns results, vloam_sync() and the stage-wise
 * entry points first enqueue whatever is still owed, so this is invisible except through the raw device pointer below. */
vloam_status vloam_process_scan_device(vloam_handle* h, const void* d_xyz_pad4, int n);
/* same with a HOST buffer — what the reference's callback hands over (a pcl::PointCloud<pcl::PointXYZ>'s points, scan_registration.cpp:131-152).
 * The sweep is copied into one of four device input buffers on a copy stream of the handle; the call returns once the copy is enqueued, and the
 * sweep itself — like its odometry and mapping — is ENQUEUED BY THE NEXT CALL on the handle, whatever that call is (another sweep of any kind, a
 * stage-wise call, a getter, vloam_sync): by then the copy has landed, so the 2 MB cross the host link beside the previous sweep's scan
 * registration and no stream waits for another (0.96 - 0.98 x the device-resident rate; profiles/r06_host_input.txt).  The sweep counts
 * (vloam_frame_count, trajectory rows) from the moment it is handed over.  What can be refused is refused by THIS call (null / empty / oversized
 * cloud, a full trajectory log); an enqueue failure of the deferred sweep (VLOAM_ERR_HIP) is reported by the call that enqueues it.
 *   pageable memory (malloc, std::vector, a ROS message): the runtime has taken its copy of the sweep when the call returns — the buffer may
 *     be reused at once; the calling thread pays the staging memcpy (~2 MB per sweep);
 *   pinned memory (hipHostMalloc / hipHostRegister): read by DMA AFTER the call returns — leave the buffer unchanged until the next
 *     vloam_sync(); no host-side copy (bench.py: host_input).
 * vloam_batch_process_scan works the same way.  Every other entry point that takes a host sweep (vloam_scan_registration, vloam_process_frame*,
 * vloam_vo_process_point_cloud) copies it on the scan-registration stream in front of its first reader, nothing deferred (the memory rules are
 * the same); VLOAM_STAGE_INLINE=1 in the environment selects that form for the two whole-sweep calls as well. */
vloam_status vloam_process_scan(vloam_handle* h, const float* xyz_pad4, int n);
vloam_status vloam_sync(vloam_handle* h);

/* Trajectory log: per processed frame 14 doubles {q_w_curr[4], t_w_curr[3]} (laser odometry, world_LOT_base_last)
 * followed by {q_map[4], t_map[3]} (mapping, world_MOT_base_last).  first..first+count-1 -> HOST buffer. */
vloam_status vloam_get_trajectory(vloam_handle* h, int first, int count, double* poses14);
vloam_status vloam_frame_count(vloam_handle* h, int* frames);
/* device address + byte size of the trajectory log (for an RCCL gather across GPUs); rows are only complete
 * after vloam_sync() */
vloam_status vloam_trajectory_device_ptr(vloam_handle* h, void** d_ptr, long long* bytes);

/* == VisualOdometry::processPointCloud + solveNlsAll (visual_odometry.cpp:157-186,254-450).
 * prev_uv/curr_uv: n_match integer pixel pairs (the reference truncates keypoints to int, :283-294).
 * angle_axis/t: in = initial guess (cam0_curr_LOT_cam0_prev) unless reset_VO_to_identity, out = estimate. */
vloam_status vloam_vo_set_calib(vloam_handle* h, const vloam_calib* calib);
vloam_status vloam_vo_process_point_cloud(vloam_handle* h, const float* xyz_pad4, int n);
vloam_status vloam_vo_solve(vloam_handle* h, const int* prev_uv, const int* curr_uv, int n_match, double angle_axis[3],
                            double t[3], int counters32_22[2]);

/* ---- The coupled per-frame VLOAM loop (configs[3]): MAIN/src/vloam_main_node.cpp:125-180.
 * vloam_set_extrinsics: base_T_cam0 and velo_T_cam0 as VloamTF::processStaticTransform leaves them (vloam_tf.cpp:55-56), row-major 4x4.
 * vloam_process_frame[_device]: one callback() with no host round trip — VO->reset / LOAM->reset, processPointCloud on the sweep that is
 * already in HBM (visual_odometry.cpp:157-186), solveNlsAll for every frame but the first (initial guess = cam0_curr_LOT_cam0_prev of the
 * previous frame unless reset_VO_to_identity, :258-281), VloamTF::VO2VeloAndBase (vloam_tf.cpp:59-75), scanRegistrationIO, laserOdometryIO
 * (combined mode detach_VO_LO == 0: para_q / para_t are overwritten by velo_last_VOT_velo_curr at the top of BOTH outer rounds,
 * laser_odometry.cpp:223-236; publish() refreshes cam0_curr_LOT_cam0_prev, :563-567), laserMappingIO.  prev_uv / curr_uv: n_match integer
 * pixel pairs in HOST memory, previous frame -> this frame (the image front-end is outside this library); ignored for the first frame.
 * Like the reference, a VO solve that returns a zero rotation angle makes every later pose NaN (visual_odometry.cpp:427-430 divides by
 * it); vloam_sync() then reports VLOAM_ERR_INVALID. */
vloam_status vloam_set_extrinsics(vloam_handle* h, const double base_T_cam0[16], const double velo_T_cam0[16]);
vloam_status vloam_process_frame_device(vloam_handle* h, const void* d_xyz_pad4, int n, const int* prev_uv, const int* curr_uv, int n_match);
vloam_status vloam_process_frame(vloam_handle* h, const float* xyz_pad4, int n, const int* prev_uv, const int* curr_uv, int n_match);
/* world_VOT_base_last per frame as {q xyzw, t} (7 doubles): the VO leg next to the LO / MO legs of vloam_get_trajectory */
vloam_status vloam_get_vo_trajectory(vloam_handle* h, int first, int count, double* poses7);
/* last frame: VO estimate (angles_0to1, t_0to1), counter32 / counter22, and velo_last_VOT_velo_curr derived from it (any may be NULL) */
vloam_status vloam_get_vo_result(vloam_handle* h, double angle_axis[3], double t[3], int counters32_22[2], double prior_q[4], double prior_t[3]);

/* The ORB + brute-force configuration of VisualOdometry::processImage — optical_flow_match = false, the reference's LAUNCH DEFAULT
 * (vloam_main/launch/vloam_main.launch:10; visual_odometry.cpp:106-116): ImageUtil::descKeypoints = cv::ORB::create()->compute on the Shi-Tomasi
 * corners (image_util.cpp:162-212: border filter at 31 px, 7 x 7 Gaussian blur, the 256 steered binary tests — the provided keypoints carry
 * angle -1, which OpenCV uses as -1 degree), then ImageUtil::matchDescriptors with BF / NORM_HAMMING / 2-NN + ratio 0.8 (:214-296).
 * OpenCV's sampling pattern (modules/features2d/src/orb.cpp: bit_pattern_31_, 256 tests x (x0, y0, x1, y1)) is learned DATA of the library that
 * is not in the reference tree: the caller hands it in.  pattern_256x4 != NULL switches the handle's image front-end (all sessions) to this
 * configuration, NULL back to optical flow; not in the middle of a sequence.  vloam_vo_process_image*, vloam_process_frame_image* and the
 * batched forms then work as before; vloam_vo_get_flow_matches returns the descriptor matches' pixel pairs, vloam_vo_get_flow nothing. */
vloam_status vloam_vo_set_orb_pattern(vloam_handle* h, const signed char* pattern_256x4);
/* the keypoints of the latest image that survive ORB's border filter (what DMatch indices refer to: descKeypoints edits the caller's vector)
 * and their 32-byte descriptors */
vloam_status vloam_vo_get_descriptors(vloam_handle* h, float* xy, unsigned char* desc32, int cap, int* n);

/* ---- Image front-end of the visual odometry, optical-flow configuration (vloam_main.launch: optical_flow_match = true); needs
 * cfg.image_width / image_height > 0.  Replaces, on the device:
 *   ImageUtil::detKeypoints (ShiTomasi)   src/visual_odometry/src/image_util.cpp:13-36    cv::goodFeaturesToTrack(img, 1024, 0.03, 7.5, 5)
 *   ImageUtil::calculateOpticalFlow       src/visual_odometry/src/image_util.cpp:351-372  cv::calcOpticalFlowPyrLK(15 x 15, maxLevel = 2 i.e. three pyramid levels, 10 / 0.03)
 *   VisualOdometry::processImage          src/visual_odometry/src/visual_odometry.cpp:91-132 (the NEW image's corners are tracked from the
 *                                         previous image into the new one, :121-122)
 * vloam_vo_process_image[_device]: one 8-bit grey image (row stride in bytes); every image of a sequence has the same size.  With
 *   cfg.CLAHE = 1 the image first goes through cv::createCLAHE(2.0)->apply (visual_odometry.cpp:31,97-100), on the device.
 * vloam_vo_get_keypoints: the corners of the last image, (x, y) pairs in goodFeaturesToTrack's order.
 * vloam_vo_get_flow: for the last image, per corner: where it sits in the previous image (= the corner itself), where it was tracked to
 *   in the new image, and calcOpticalFlowPyrLK's status byte (n = 0 after the first image).
 * vloam_vo_get_flow_matches: the match loop's integer pixel pairs of the tracked corners (visual_odometry.cpp:296-308), ready for
 *   vloam_vo_solve / vloam_process_frame.
 * vloam_process_frame_image[_device]: vloam_process_frame with the matches taken from the image instead of from the caller — cloud and
 *   image of one frame in, nothing comes back to the host (the image work rides on a stream of its own next to scan registration).
 * vloam_vo_match_descriptors: ImageUtil::matchDescriptors (image_util.cpp:221-296) for binary descriptors in host memory —
 *   BFMatcher(NORM_HAMMING): knnMatch k = 2 + ratio test 0.8 (select_knn != 0, the reference's SelectType::KNN) or match with crossCheck
 *   (select_knn == 0); ties resolve like cv::batchDistance (lower index).  Matches come back in query order.
 * Of the ORB + brute-force configuration (optical_flow_match = false) only the descriptor EXTRACTION is not provided: OpenCV's learned
 * ORB sampling table cannot be restated without the library — compute the descriptors with OpenCV at the corners of
 * vloam_vo_get_keypoints and match them here. */
vloam_status vloam_vo_process_image(vloam_handle* h, const unsigned char* gray, int width, int height, int stride);
vloam_status vloam_vo_process_image_device(vloam_handle* h, const void* d_gray, int width, int height, int stride);
vloam_status vloam_vo_get_keypoints(vloam_handle* h, float* xy, int cap, int* n);
vloam_status vloam_vo_get_flow(vloam_handle* h, float* prev_xy, float* curr_xy, unsigned char* status, int cap, int* n);
vloam_status vloam_vo_get_flow_matches(vloam_handle* h, int* prev_uv, int* curr_uv, int cap, int* n);
vloam_status vloam_vo_match_descriptors(vloam_handle* h, const unsigned char* desc_prev, int n_prev, const unsigned char* desc_curr, int n_curr, int bytes_per_desc,
                                        int select_knn, int* query_idx, int* train_idx, int cap, int* n_matches);
vloam_status vloam_process_frame_image_device(vloam_handle* h, const void* d_xyz_pad4, int n, const void* d_gray, int width, int height, int stride);
vloam_status vloam_process_frame_image(vloam_handle* h, const float* xyz_pad4, int n, const unsigned char* gray, int width, int height, int stride);
/* Batched coupled frames from raw inputs (handles of vloam_create_batch with an image front-end): session b gets sweep xyz_pad4[b] and the
 * grey image gray[b]; all images of a call share width / height / stride.  The sessions' images run one after the other on the image
 * stream (each session has the image buffers of its own arena); everything behind them is the batched frame chain.  The getters
 * vloam_vo_get_keypoints / _flow / _flow_matches read the session chosen with vloam_select_session. */
vloam_status vloam_batch_process_frame_image_device(vloam_handle* h, const void* const* d_xyz_pad4, const int* n, const void* const* d_gray, int width, int height,
                                                    int stride);
vloam_status vloam_batch_process_frame_image(vloam_handle* h, const float* const* xyz_pad4, const int* n, const unsigned char* const* gray, int width, int height,
                                             int stride);

/* Parity hooks (tests only; need cfg.debug = 1 for the per-point arrays).  Copies up to cap elements of
 * the named array into buf (element type given per item) and returns the element count in *n.
 *   stage 0 (scan registration): item 0 curvature f32[N2], 1 sort order i32[N2], 2 picked i32[N2],
 *       3 label i32[N2], 4 scanStartInd i32[64], 5 scanEndInd i32[64], 6 sharp idx i32, 7 lessSharp idx i32,
 *       8 flat idx i32, 9 scalars f32{startOri,endOri,halfPassedAt,n_after_s1,N2}
 *   stage 1 (laser odometry, item = outer*16 + k): k=0 corner corr i32[n][3] (i,a,b; -1 = none),
 *       1 plane corr i32[n][4], 2 LM record f64 (see vloam_device.h LMRecord), 3 per-factor residuals f64
 *   stage 2 (laser mapping, same k layout; corr rows are f64 geometry: corner a,b[6] / plane n,d[4] / flag)
 *   stage 3 (VO): 0 buckets, 1 per-match rows, 2 LM record */
vloam_status vloam_debug_get(vloam_handle* h, int stage, int item, void* buf, long long cap_bytes, long long* n_bytes);

/* Per-kernel HIP-event timer for bench.py's roofline line: brackets every launch of the kernel whose __global__
 * symbol is `name` (e.g. "k_lo_assoc") with an event pair on the handle's own stream.  "" disables.
 * vloam_profile_read returns the summed milliseconds and launch count since the last read. */
vloam_status vloam_profile_kernel(vloam_handle* h, const char* name, int max_launches);
vloam_status vloam_profile_read(vloam_handle* h, double* total_ms, int* launches);
/* name "*" brackets every launch of every kernel; vloam_profile_read_table then returns the per-kernel sums, ms[k] / launches[k]
 * for kernel id k < n_kernels (ids 0..vloam_profile_kernel_count()-1, symbol names through vloam_profile_kernel_name).  This is
 * also where the reference's per-substage timers map to ("sort q time" / "seperate points" -> k_sr_ring,
 * "data association" -> k_lo_assoc / k_map_assoc + k_map_fit, "solver time" -> k_lm_compact + k_lm_solve,
 * "build tree" -> k_lo_grid_*, "filter time" -> k_map_ds_*, "add points" -> k_map_insert + k_map_finalize). */
vloam_status vloam_profile_read_table(vloam_handle* h, int n_kernels, double* ms, int* launches);
int vloam_profile_kernel_count(void);
const char* vloam_profile_kernel_name(int k);

/* Health counters (synchronises): out8[0] cooperative Levenberg-Marquardt solves that found their partner workgroups missing and finished on ONE
 * workgroup instead (same factors, same trust-region loop; the pose agrees with the cooperative form to round-off, the solve is ~0.5 s late
 * because the workgroup first waited for its partners), summed over the sessions; out8[1] 1 once the handle has reacted to that by launching
 * one-workgroup solves only (vloam_sync does, see the co-residency note at vloam_create_batch); out8[2] voxel-table rebuilds; out8[3..7] 0. */
vloam_status vloam_get_health(vloam_handle* h, long long out8[8]);

/* Timing of the last vloam_sync()ed scans: HIP-event milliseconds accumulated per stage
 * {scanRegistration, laserOdometry, laserMapping, vo} and number of scans covered. */
vloam_status vloam_get_stage_ms(vloam_handle* h, double ms4[4], int* scans);
/* Measured per-run counts that DESIGN.md's algorithmic-byte formulas need:
 * {N_in, N2, n_sharp, n_lessSharp, n_flat, n_lessFlat, C, S, F_o_corner, F_o_plane, E_o, n_c, n_s, K_m, E_m, M} of the last scan. */
vloam_status vloam_get_counts(vloam_handle* h, long long counts16[16]);

#ifdef __cplusplus
}
#endif
#endif
