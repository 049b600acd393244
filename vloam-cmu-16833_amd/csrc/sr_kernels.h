// Host-visible launch interface of the scan-registration kernels (sr_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "vloam_device.h"

namespace vloam {

struct SRBuffers {
  FrameScalars* S;
  int* sticky_err;      // handle-wide sticky error word: the per-sweep S->error (rewritten every sweep, per buffer set) is folded into it
  signed char* sid;     // [max_points] ring id or -1
  float* ori;           // [max_points] raw -atan2(y, x)
  int* blockhist;       // [nblk_max][kMaxRings]
  int* blockoff;        // [nblk_max][kMaxRings]
  float4* cloud;        // [max_points] ring-major (x, y, z, intensity) == laserCloud
  int* sharp_idx;       // [kMaxRings][kSectors][2]
  int* less_sharp_idx;  // [kMaxRings][kSectors][20]
  int* flat_idx;        // [kMaxRings][kSectors][4]
  float4* ring_ds;      // [kMaxRings][kMaxRingLen] per-ring VoxelGrid(0.2) output
  float4* sharp;        // [kMaxSharp]      cornerPointsSharp
  float4* less_sharp;   // [kMaxLessSharp]  cornerPointsLessSharp (this frame's ping-pong half)
  float4* flat;         // [kMaxFlat]       surfPointsFlat
  float4* less_flat;    // [max_points]     surfPointsLessFlat   (this frame's ping-pong half)
  // parity hooks (cfg.debug)
  float* dbg_curv;
  int *dbg_sort, *dbg_picked, *dbg_label;
  long long* dbg_cyc;   // [kMaxRings][8] shader-clock cycles of k_sr_ring's phases (debug)
  int* dbg_feat_idx;    // [3][kMaxLessSharp]
};

hipError_t sr_init();
hipError_t sr_launch(hipStream_t st, const SRBuffers& b, const float4* d_in, int n, int N_SCANS, float min_range, bool debug, ProfHook* ph = nullptr,
                     hipEvent_t done = nullptr);  // `done`: recorded when the feature clouds are complete

}  // namespace vloam
