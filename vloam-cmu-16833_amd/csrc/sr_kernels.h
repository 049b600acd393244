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
  __host__ __device__ void rebase(size_t off) {
    rbp(S, off); rbp(sticky_err, off); rbp(sid, off); rbp(ori, off); rbp(blockhist, off); rbp(blockoff, off); rbp(cloud, off);
    rbp(sharp_idx, off); rbp(less_sharp_idx, off); rbp(flat_idx, off); rbp(ring_ds, off); rbp(sharp, off); rbp(less_sharp, off);
    rbp(flat, off); rbp(less_flat, off); rbp(dbg_curv, off); rbp(dbg_sort, off); rbp(dbg_picked, off); rbp(dbg_label, off);
    rbp(dbg_cyc, off); rbp(dbg_feat_idx, off);
  }
};

hipError_t sr_init();
// bi: the sweeps of the B sessions (device pointers + point counts); `done`: recorded when the feature clouds are complete
// ring_watch: host-mapped [sessions], set by the small ring tier when a ring nears its capacity; big_tier: also launch the 4096-point tier
hipError_t sr_launch(hipStream_t st, const SRBuffers& b, const BatchIn& bi, Sess se, int N_SCANS, float min_range, int debug_level, ProfHook* ph = nullptr,
                     hipEvent_t done = nullptr, int* ring_watch = nullptr, bool big_tier = true);

// clouds uploaded into a buffer set by the caller (vloam_set_odometry_input / vloam_set_mapping_input): counts (< 0 = keep) and the VoxelGrid boxes
// of the two less-clouds re-derived from them; single session
void sr_adopt_launch(hipStream_t st, const SRBuffers& b, int n_full, int n_sharp, int n_less_sharp, int n_flat, int n_less_flat);

}  // namespace vloam
