// Depth-enhanced VO residual stack on gfx950 — host-visible interface (vo_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "../../include/vloam_hip/c_api.h"
#include "vloam_device.h"

namespace vloam {

struct VOContext {
  int dummy = 0;
};

vloam_status vo_create(VOContext* v, const vloam_config& cfg, hipStream_t st, std::vector<void*>& allocs);
vloam_status vo_set_calib(VOContext* v, hipStream_t st, const vloam_calib* c);
vloam_status vo_process_point_cloud(VOContext* v, hipStream_t st, const float4* d_in, int n);
vloam_status vo_solve(VOContext* v, const vloam_config& cfg, hipStream_t st, const int* prev_uv, const int* curr_uv, int n_match,
                      double aa[3], double t[3], int counters[2]);
vloam_status vo_debug_get(VOContext* v, int item, void* buf, long long cap, long long* n);

}  // namespace vloam
