// Host-visible launch interface of the single-workgroup Levenberg–Marquardt solver (lm_solve.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "vloam_device.h"

namespace vloam {

// Runs the whole ceres::Solve() over the accepted slots of F (type != 0; the association kernels also count them per
// 64-slot row in F.rowcnt).  Two launches: k_lm_compact (grid) + k_lm_solve (one workgroup).
// d_x: 7 doubles (q xyzw, t) when quat, else 6 (angle-axis, t); updated in place like ceres::Solve.
// d_enable (optional): device int; 0 skips the solve entirely (mapping gate, laser_mapping.cpp:448).
// n_edge_slots: slots [0, n_edge_slots) hold LidarEdgeFactors, the rest plane factors (multiple of 64; ignored when !quat).
void lm_launch(hipStream_t st, Sess se, const FactorTable& F, int n_edge_slots, double* d_x, LMRecord* d_rec, int max_iters, double huber_a, bool quat,
               const int* d_enable, ProfHook* ph = nullptr, LOState* fin_lo = nullptr, double* fin_traj = nullptr,
               hipEvent_t done = nullptr);
// fin_lo / fin_traj: when set, the solve's last act is LaserOdometry's pose integration + trajectory row (saves a launch)
// done: event bound to the solve dispatch (recorded when it completes)

// Cooperative solves: rank candidate sync-word slots (n_cand slots of kLmSyncDoubles doubles, stride_bytes apart, in `pool`, which carries 256 spare bytes behind the last slot) by
// the measured round trip of the grid barrier at that address; order_out = slot indices, fastest first.  0 on success.
int lm_sync_calibrate(hipStream_t st, double* pool, int n_cand, size_t stride_bytes, int* order_out);

}  // namespace vloam
