#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned xor1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned xor2(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned xor4(unsigned v) {
  int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xA, false);
  return (unsigned)__builtin_amdgcn_update_dpp(t, (int)v, 0x12C, 0xf, 0x5, false);
}
__device__ __forceinline__ unsigned xor8(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned xor16(unsigned v) {
  auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  return (threadIdx.x & 16) ? r[0] : r[1];
}
__device__ __forceinline__ unsigned xor32(unsigned v) {
  auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return (threadIdx.x & 32) ? r[0] : r[1];
}
__global__ void k(unsigned* o) {
  unsigned v = threadIdx.x;
  o[threadIdx.x * 6 + 0] = xor1(v); o[threadIdx.x * 6 + 1] = xor2(v); o[threadIdx.x * 6 + 2] = xor4(v);
  o[threadIdx.x * 6 + 3] = xor8(v); o[threadIdx.x * 6 + 4] = xor16(v); o[threadIdx.x * 6 + 5] = xor32(v);
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 6 * 4); k<<<1, 64>>>(d); unsigned h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int js[6] = {1, 2, 4, 8, 16, 32}; int bad = 0;
  for (int q = 0; q < 6; q++) { int b = 0; for (int l = 0; l < 64; l++) if (h[l * 6 + q] != (unsigned)(l ^ js[q])) b++; printf("xor %d: %d wrong (lane 0 got %u, lane 5 got %u, lane 20 got %u, lane 40 got %u)\n", js[q], b, h[q], h[5*6+q], h[20*6+q], h[40*6+q]); bad += b; }
  return bad != 0;
}
