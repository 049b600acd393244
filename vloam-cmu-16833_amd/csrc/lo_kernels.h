// Host-visible launch interface of the laser-odometry kernels (lo_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "vloam_device.h"

namespace vloam {

// Two uniform hash grids over one sweep's lessSharp / lessFlat cloud (the device-side stand-in for the two
// pcl::KdTreeFLANN::setInputCloud calls at laser_odometry.cpp:525-526): a 1 m grid for the expanding exact search and a 5 m
// grid whose 27-cell neighbourhood covers DISTANCE_SQ_THRESHOLD = 25 for the rare queries without a close neighbour.
// Grid index g = kind + 2 * level (kind 0 corner / 1 surf, level 0 = 1 m / 1 = 5 m); bucket = hash(cell) & mask.
constexpr int kGridBuckets[4] = {1 << 13, 1 << 15, 1 << 14, 1 << 15};   // 5 m level: 1 024 / 2 048 cell slots x 16 ring groups (contiguous per cell, lo_kernels.hip coarse_bucket)
constexpr int kStopLen = kMaxRings + 4;
constexpr int kGridMaxBuckets = 1 << 15;
struct LoGrid {
  int* cnt[4];     // [buckets] points per bucket (count pass); counted back down to zero by the scatter pass
  int* start[4];   // [buckets + 1] exclusive offsets
  float4* pts[4];  // [n] the points grouped by bucket: (x, y, z, bits: index | ring << 24)
  int* occ;        // [2 kinds][first, last][kMaxRings] first / last index of every stored scan line (armed: INT_MAX / -1)
  int* stops;      // [2 kinds][2][kStopLen] where the reference's adjacent-line walks break (see k_lo_assoc)
  int mask[4];
  __host__ __device__ void rebase(size_t off) {
    // (never null: no select against nullptr here — it would cost the kernels the global-address-space inference for these
    // dynamically indexed members and turn their accesses into FLAT instructions)
#pragma unroll
    for (int g = 0; g < 4; g++) {
      cnt[g] = (int*)((char*)cnt[g] + off); start[g] = (int*)((char*)start[g] + off); pts[g] = (float4*)((char*)pts[g] + off);
    }
    occ = (int*)((char*)occ + off); stops = (int*)((char*)stops + off);
  }
};
void lo_grid_build_launch(hipStream_t st, Sess se, const float4* less_sharp, const float4* less_flat, const FrameScalars* S, const LoGrid& G,
                          ProfHook* ph = nullptr);

// Slot layout of the LO factor table: [0, kMaxSharp) corner features, [kMaxSharp, kMaxLoFactors) plane features.
// corr: [kMaxLoFactors][4] ints (feature index or -1, closest, 2nd, 3rd).
void lo_assoc_launch(hipStream_t st, Sess se, const float4* sharp, const float4* flat, const FrameScalars* Sc, const float4* CL, const float4* SL,
                     const FrameScalars* Sp, const LoGrid& G, const LOState* lo, const FactorTable& F, int* corr, long long* dbg_cyc,
                     int* queue /* [kMaxLoFactors] left-over slots of the 16-lane form, or null */, int* queue_n /* [2] */, int launch_no, ProfHook* ph = nullptr);
// copy_to_para: LO:223-236 (combined mode).  vo_row7 != nullptr: first publish this frame's visual odometry (see k_lo_set_prior).
void lo_set_prior_launch(hipStream_t st, Sess se, LOState* lo, bool copy_to_para = true, const double* vo_x = nullptr, bool vo_solved = false,
                         double* vo_row7 = nullptr, int* err = nullptr);
void lo_finish_launch(hipStream_t st, Sess se, LOState* lo, double* traj_row14, bool integrate, ProfHook* ph = nullptr);

}  // namespace vloam
