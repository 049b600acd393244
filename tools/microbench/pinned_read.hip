// How fast does a KERNEL read a 2 MB sweep out of pinned host memory (zero-copy over PCIe), against hipMemcpyAsync of the same buffer — on the
// GPU (HIP events) and on the HOST (time inside the API calls)?  Behind the host-input path of c_api.cpp.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/pinned_read tools/microbench/pinned_read.hip && /tmp/pinned_read
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_copy(const float4* __restrict__ src, float4* __restrict__ dst, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}
template <int PER>
__global__ void k_copy_per(const float4* __restrict__ src, float4* __restrict__ dst, int n) {
  float4 p[PER];
#pragma unroll
  for (int e = 0; e < PER; e++) { const int i = (blockIdx.x * PER + e) * blockDim.x + threadIdx.x; p[e] = i < n ? src[i] : make_float4(0, 0, 0, 0); }
#pragma unroll
  for (int e = 0; e < PER; e++) { const int i = (blockIdx.x * PER + e) * blockDim.x + threadIdx.x; if (i < n) dst[i] = p[e]; }
}
__global__ void k_first(const float4* __restrict__ src, int* out, int n, int trip) {   // 16 walkers x one trip of `trip` points
  const int t0 = blockIdx.x * trip;
  int found = 0;
  for (int i = t0 + threadIdx.x; i < t0 + trip && i < n; i += blockDim.x) found |= src[i].x != 0.f;
  if (__ballot(found) && (threadIdx.x & 63) == 0) atomicOr(out, 1);
}

int main() {
  const int n = 131072, R = 8, reps = 200;
  std::vector<float4*> host(R);
  for (auto& h : host) { CK(hipHostMalloc((void**)&h, n * sizeof(float4), hipHostMallocDefault)); for (int i = 0; i < n; i++) h[i] = make_float4(1.f + i, 2.f, 3.f, 0.f); }
  float4* dev; CK(hipMalloc((void**)&dev, (size_t)n * sizeof(float4) * R));
  int* flag; CK(hipMalloc((void**)&flag, 4));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto&& body) {
    for (int k = 0; k < 20; k++) body(k);
    hipStreamSynchronize(st);
    double host_us = 0;
    hipEventRecord(e0, st);
    for (int k = 0; k < reps; k++) {
      auto a = std::chrono::steady_clock::now();
      body(k);
      host_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
    }
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %7.1f us on the stream per sweep (%5.1f GB/s)   %6.1f us of host time per call\n", name, 1e3 * ms / reps, 2.097152e6 / (1e3 * ms / reps) / 1e3, host_us / reps);
  };
  run("hipMemcpyAsync, pinned -> device", [&](int k) { hipMemcpyAsync(dev + (size_t)(k % R) * n, host[k % R], n * sizeof(float4), hipMemcpyHostToDevice, st); });
  run("kernel, 128 x 1024 lanes, one float4 per lane", [&](int k) { hipLaunchKernelGGL(k_copy, dim3(128), dim3(1024), 0, st, host[k % R], dev + (size_t)(k % R) * n, n); });
  run("kernel, 512 x 256 lanes, one float4 per lane", [&](int k) { hipLaunchKernelGGL(k_copy, dim3(512), dim3(256), 0, st, host[k % R], dev + (size_t)(k % R) * n, n); });
  run("kernel, 128 x 256 lanes, four float4 per lane in flight", [&](int k) { hipLaunchKernelGGL(k_copy_per<4>, dim3(128), dim3(256), 0, st, host[k % R], dev + (size_t)(k % R) * n, n); });
  run("kernel, 32 x 256 lanes, sixteen float4 per lane in flight", [&](int k) { hipLaunchKernelGGL(k_copy_per<16>, dim3(32), dim3(256), 0, st, host[k % R], dev + (size_t)(k % R) * n, n); });
  run("kernel, device -> device (the same kernel on resident data)", [&](int k) { hipLaunchKernelGGL(k_copy, dim3(128), dim3(1024), 0, st, dev + (size_t)((k + 1) % R) * n, dev + (size_t)(k % R) * n, n); });
  run("first/last style: 16 walkers x 4096 points from pinned", [&](int k) { hipLaunchKernelGGL(k_first, dim3(16), dim3(256), 0, st, host[k % R], flag, n, 4096); });
  run("first/last style: 16 walkers x 1024 points from pinned", [&](int k) { hipLaunchKernelGGL(k_first, dim3(16), dim3(256), 0, st, host[k % R], flag, n, 1024); });
  run("first/last style: 16 walkers x 4096 points from device", [&](int k) { hipLaunchKernelGGL(k_first, dim3(16), dim3(256), 0, st, dev + (size_t)(k % R) * n, flag, n, 4096); });
  hipPointerAttribute_t at;
  auto a = std::chrono::steady_clock::now();
  int ok = 0;
  for (int k = 0; k < 1000; k++) ok += hipPointerGetAttributes(&at, host[k % R] + 17) == hipSuccess && at.type == hipMemoryTypeHost;
  printf("hipPointerGetAttributes on a pinned interior pointer: %.2f us per call (%d of 1000 say host; devicePointer %s)\n",
         std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count() / 1000, ok, at.devicePointer ? "set" : "null");
  {
    void* d1 = nullptr;
    hipPointerGetAttributes(&at, host[0] + 17);
    hipHostGetDevicePointer(&d1, host[0] + 17, 0);
    printf("interior pointer host + 17: hipPointerGetAttributes.devicePointer - host = %lld float4, hipHostGetDevicePointer - host = %lld float4\n",
           (long long)((float4*)at.devicePointer - host[0]), (long long)((float4*)d1 - host[0]));
  }
  std::vector<float4> pageable(n);
  a = std::chrono::steady_clock::now();
  int bad = 0;
  for (int k = 0; k < 1000; k++) { bad += hipPointerGetAttributes(&at, pageable.data() + 17) != hipSuccess || at.type != hipMemoryTypeHost; (void)hipGetLastError(); }
  printf("hipPointerGetAttributes on a pageable pointer: %.2f us per call (%d of 1000 refused / not host)\n",
         std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count() / 1000, bad);
  return 0;
}
