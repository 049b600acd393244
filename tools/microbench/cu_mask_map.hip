// Which compute unit does bit b of a hipExtStreamCreateWithCUMask mask select on MI355X?  One single-bit stream per probe, a kernel that
// reads HW_REG_XCC_ID / HW_REG_HW_ID.  (Needed to keep other streams off the compute units a cooperative solve is placed on.)
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/cu_mask_map.hip -o /tmp/cu_mask_map && /tmp/cu_mask_map
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_where(int* out) {
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID [3:0]
  const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID [15:0]
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = (int)xcc; out[2 * blockIdx.x + 1] = (int)hw; }
}
int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("multiProcessorCount %d\n", ncu);
  int* out; CK(hipMalloc(&out, 4096));
  int h[64];
  printf("plain launch of 16 blocks: block -> xcc\n ");
  hipLaunchKernelGGL(k_where, dim3(16), dim3(64), 0, 0, out);
  CK(hipDeviceSynchronize()); CK(hipMemcpy(h, out, 128, hipMemcpyDeviceToHost));
  for (int b = 0; b < 16; b++) printf(" %d->%d", b, h[2 * b] & 15);
  printf("\nsingle-bit CU masks: bit -> (xcc, se, sh, cu)\n");
  for (int b = 0; b < ncu; b += (b < 40 ? 1 : 13)) {
    uint32_t mask[16] = {0};
    mask[b >> 5] = 1u << (b & 31);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)((ncu + 31) / 32), mask) != hipSuccess) { printf(" bit %d: stream creation failed\n", b); continue; }
    hipLaunchKernelGGL(k_where, dim3(1), dim3(64), 0, s, out);
    CK(hipStreamSynchronize(s)); CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
    printf(" bit %3d -> xcc %d se %d sh %d cu %2d\n", b, h[0] & 15, (h[1] >> 13) & 7, (h[1] >> 12) & 1, (h[1] >> 8) & 15);
    CK(hipStreamDestroy(s));
  }
  return 0;
}
