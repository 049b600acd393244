#include "map_kernels.h"
namespace vloam {
vloam_status map_create(MapContext* m, const vloam_config&, hipStream_t, std::vector<void*>& allocs) {
  void* p = nullptr;
  if (hipMalloc(&p, sizeof(MapState) + 256) != hipSuccess) return VLOAM_ERR_HIP;
  (void)hipMemset(p, 0, sizeof(MapState) + 256);
  allocs.push_back(p);
  m->state = (MapState*)p;
  return VLOAM_OK;
}
vloam_status map_enqueue(MapContext*, const vloam_config&, hipStream_t, const SRBuffers&, LOState*, double*, bool) { return VLOAM_OK; }
vloam_status map_get_cloud(MapContext*, hipStream_t, int, const SRBuffers&, float*, int, int*) { return VLOAM_ERR_INVALID; }
vloam_status map_error(MapContext*, int* e) { *e = 0; return VLOAM_OK; }
vloam_status map_debug_get(MapContext*, int, void*, long long, long long*) { return VLOAM_ERR_INVALID; }
vloam_status map_counts(MapContext*, long long*) { return VLOAM_OK; }
}
