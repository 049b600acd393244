#include "vo_kernels.h"
namespace vloam {
vloam_status vo_create(VOContext*, const vloam_config&, hipStream_t, std::vector<void*>&) { return VLOAM_OK; }
vloam_status vo_set_calib(VOContext*, hipStream_t, const vloam_calib*) { return VLOAM_ERR_INVALID; }
vloam_status vo_process_point_cloud(VOContext*, hipStream_t, const float4*, int) { return VLOAM_ERR_INVALID; }
vloam_status vo_solve(VOContext*, const vloam_config&, hipStream_t, const int*, const int*, int, double*, double*, int*) { return VLOAM_ERR_INVALID; }
vloam_status vo_debug_get(VOContext*, int, void*, long long, long long*) { return VLOAM_ERR_INVALID; }
}
