// laserMapping on gfx950 — host-visible interface (map_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "../../include/vloam_hip/c_api.h"
#include "sr_kernels.h"
#include "vloam_device.h"

namespace vloam {

struct MapContext {
  MapState* state = nullptr;
  int* error = nullptr;
};

vloam_status map_create(MapContext* m, const vloam_config& cfg, hipStream_t st, std::vector<void*>& allocs);
vloam_status map_enqueue(MapContext* m, const vloam_config& cfg, hipStream_t st, const SRBuffers& cur, LOState* lo, double* traj_row14, bool skip_frame);
vloam_status map_get_cloud(MapContext* m, hipStream_t st, int which, const SRBuffers& cur, float* xyzi4, int cap, int* n);
vloam_status map_error(MapContext* m, int* err_bits);
vloam_status map_debug_get(MapContext* m, int item, void* buf, long long cap, long long* n);
vloam_status map_counts(MapContext* m, long long c[16]);

}  // namespace vloam
