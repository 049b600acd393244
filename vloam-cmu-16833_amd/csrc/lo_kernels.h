// Host-visible launch interface of the laser-odometry kernels (lo_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "vloam_device.h"

namespace vloam {

// Slot layout of the LO factor table: [0, kMaxSharp) corner features, [kMaxSharp, kMaxLoFactors) plane features.
// corr: [kMaxLoFactors][4] ints (feature index or -1, closest, 2nd, 3rd).
void lo_assoc_launch(hipStream_t st, const float4* sharp, const float4* flat, const FrameScalars* Sc, const float4* CL, const float4* SL,
                     const FrameScalars* Sp, const LOState* lo, const FactorTable& F, int* corr, ProfHook* ph = nullptr);
void lo_set_prior_launch(hipStream_t st, LOState* lo);
void lo_finish_launch(hipStream_t st, LOState* lo, double* traj_row14, bool integrate, ProfHook* ph = nullptr);

}  // namespace vloam
