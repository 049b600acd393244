// Host-visible launch interface of the laser-odometry kernels (lo_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "vloam_device.h"

namespace vloam {

// Uniform 1 m hash grid over one sweep's lessSharp / lessFlat cloud (the device-side stand-in for the two
// pcl::KdTreeFLANN::setInputCloud calls at laser_odometry.cpp:525-526): bucket = hash(floor(x), floor(y), floor(z)).
constexpr int kGridBucketsCorner = 1 << 14, kGridBucketsSurf = 1 << 16;
struct LoGrid {
  int* start[2];   // [buckets + 1] exclusive offsets (after the scan), kind 0 corner / 1 surf
  int* fill[2];    // [buckets] scatter cursors
  int* items[2];   // [n] point indices grouped by bucket
  int mask[2];
};
void lo_grid_build_launch(hipStream_t st, const float4* less_sharp, const float4* less_flat, const FrameScalars* S, const LoGrid& G,
                          ProfHook* ph = nullptr);

// Slot layout of the LO factor table: [0, kMaxSharp) corner features, [kMaxSharp, kMaxLoFactors) plane features.
// corr: [kMaxLoFactors][4] ints (feature index or -1, closest, 2nd, 3rd).
void lo_assoc_launch(hipStream_t st, const float4* sharp, const float4* flat, const FrameScalars* Sc, const float4* CL, const float4* SL,
                     const FrameScalars* Sp, const LoGrid& G, const LOState* lo, const FactorTable& F, int* corr, ProfHook* ph = nullptr);
void lo_set_prior_launch(hipStream_t st, LOState* lo);
void lo_finish_launch(hipStream_t st, LOState* lo, double* traj_row14, bool integrate, ProfHook* ph = nullptr);

}  // namespace vloam
