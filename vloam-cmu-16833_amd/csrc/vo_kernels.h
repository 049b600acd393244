// Depth-enhanced VO residual stack on gfx950 — host-visible interface (vo_kernels.hip).
// Restates PointCloudUtil::{projectPointCloud, downsamplePointCloud, queryDepth} and VisualOdometry::solveNlsAll
// (/root/reference/src/visual_odometry/src/point_cloud_util.cpp:148-174,205-260,302-387; visual_odometry.cpp:254-450).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "../../include/vloam_hip/c_api.h"
#include "vloam_device.h"

namespace vloam {

constexpr int kImgW = 1242, kImgH = 375, kGrid = 5;          // point_cloud_util.h:41-42, visual_odometry.cpp:56
constexpr int kBW = 249, kBH = 75, kBuckets = kBW * kBH;     // ceil(1242/5) x ceil(375/5)
constexpr int kVoMaxMatches = 8192;

struct DepthMapDev {
  float *bx, *by, *bd;  // [kBuckets]  bucket_x / bucket_y / bucket_depth
  int* bc;              // [kBuckets]  bucket_count
  __host__ __device__ void rebase(size_t off) { rbp(bx, off); rbp(by, off); rbp(bd, off); rbp(bc, off); }
};
struct VoMatchCounts { int n[kMaxBatch]; };   // pixel matches per session of a batched handle (kernel argument by value)

struct VOContext {
  vloam_calib* d_calib = nullptr;
  bool have_calib = false;
  // The reference ping-pongs two PointCloudUtil objects (i = count % 2).  Here the depth maps rotate with the handle's buffer sets:
  // the scan-registration stream (which builds them) runs up to kSets - 1 sweeps ahead of the odometry stream (which reads the
  // previous frame's map in k_vo_match).
  static constexpr int kSets = kBufferSets;
  DepthMapDev maps[kSets];
  int count = -1, i = 0;     // VisualOdometry::reset(): ++count; i = count % kSets
  float4* uvd = nullptr;     // [max_points] (u, v, depth, bucket id as int bits; -1 = not in a bucket)
  int* bcount = nullptr;     // [kBuckets + 1] scratch counts -> offsets
  int* bfill = nullptr;      // [kBuckets]
  int* seg = nullptr;        // [max_points]
  int* d_prev = nullptr;     // [kVoMaxMatches][2]
  int* d_curr = nullptr;
  int* d_prev_set[kSets] = {};  // per buffer set (coupled frame loop: the solve of frame k is enqueued after the matches of frame k + 1 arrive)
  int* d_curr_set[kSets] = {};
  VoMatchCounts n_match_set[kSets] = {};
  FactorTable F{};
  LMRecord* rec = nullptr;
  double* x = nullptr;       // [6] angle-axis, t
  double* match_dbg = nullptr;  // [kVoMaxMatches][7] kind, depth0, obs[5]
  int* counters = nullptr;   // [2] counter32, counter22
  int max_points = 0;
  Sess se;                   // sessions of the handle (every VO buffer lives in the session arenas)
};

vloam_status vo_layout(VOContext* v, const vloam_config& cfg, Arena& A);
vloam_status vo_set_calib(VOContext* v, hipStream_t st, const vloam_calib* c);
vloam_status vo_process_point_cloud(VOContext* v, hipStream_t st, const float4* d_in, int n);
vloam_status vo_solve(VOContext* v, const vloam_config& cfg, hipStream_t st, const int* prev_uv, const int* curr_uv, int n_match,
                      double aa[3], double t[3], int counters[2]);
// coupled frame loop (vloam_process_frame*): depth map of frame `frame` on the scan-registration stream ...
vloam_status vo_depth_enqueue(VOContext* v, hipStream_t st, const BatchIn& bi, int frame, const int* const* prev_uv, const int* const* curr_uv,
                              const int* n_match, ProfHook* ph);
// ... and match + solve on the odometry stream, initial guess from lo->tf.cam0_curr_LOT_cam0_prev (VO:258-281); result stays in v->x
vloam_status vo_solve_enqueue(VOContext* v, const vloam_config& cfg, hipStream_t st, int frame, LOState* lo, ProfHook* ph);
vloam_status vo_debug_get(VOContext* v, int item, void* buf, long long cap, long long* n);

}  // namespace vloam
