// Host-visible launch interface of the single-workgroup Levenberg–Marquardt solver (lm_solve.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "vloam_device.h"

namespace vloam {

// Solves over slots [0, n) of F where n = *d_n_slots (device) if d_n_slots != nullptr, else n_slots_fixed.
// d_x: 7 doubles (q xyzw, t) when quat, else 6 (angle-axis, t); updated in place like ceres::Solve.
// d_enable (optional): device int; 0 skips the solve entirely (mapping gate, laser_mapping.cpp:448).
void lm_launch(hipStream_t st, const FactorTable& F, const int* d_n_slots, int n_slots_fixed, double* d_x, LMRecord* d_rec, int max_iters,
               double huber_a, bool quat, const int* d_enable, ProfHook* ph = nullptr);

}  // namespace vloam
