// Round trip of the cooperative LM solve's grid barrier (publish partials, agent-scope atomic barrier, read all partials) as a
// function of the ADDRESS of the sync line and of the XCD placement of the 4 workgroups.  hipcc --offload-arch=gfx950 -O3.
// Output of one run: profiles/r01_grid_barrier_microbench.txt (motivates lm_sync_calibrate, lm_solve.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// grid barrier micro-benchmark: nb working blocks, `stride` = blockIdx spacing between working blocks (8 -> same XCD)
__global__ __launch_bounds__(256) void k_bar(unsigned* bar, double* part, int nwork, int stride, int iters, double* out) {
  if (blockIdx.x % stride != 0) return;
  const int b = blockIdx.x / stride;
  if (b >= nwork) return;
  double acc = 0;
  for (int it = 0; it < iters; it++) {
    // each block publishes a partial (32 doubles), barrier, then everyone reads all partials
    if (threadIdx.x < 32) __hip_atomic_store(&part[(it & 1) * 32 * 64 + b * 32 + threadIdx.x], (double)(it + b + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(it + 1) * (unsigned)nwork;
      while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (threadIdx.x < 32) {
      double s = 0;
      for (int q = 0; q < nwork; q++) s += __hip_atomic_load(&part[(it & 1) * 32 * 64 + q * 32 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc += s;
    }
  }
  if (threadIdx.x < 32 && b == 0) out[threadIdx.x] = acc;
}
int main() {
  unsigned* bar; double *part, *out; char* big;
  hipMalloc(&big, 256 << 20); hipMalloc(&out, 32 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int stride : {1, 8}) for (size_t off : {(size_t)0, (size_t)256, (size_t)1024, (size_t)4096, (size_t)8192, (size_t)16384, (size_t)65536, (size_t)(1 << 20), (size_t)(2 << 20), (size_t)(3 << 20), (size_t)(16 << 20), (size_t)(17 << 20), (size_t)(64 << 20), (size_t)(100 << 20) + 4096 * 3, (size_t)(200 << 20) + 256 * 5}) {
    const int nwork = 4;
    bar = (unsigned*)(big + off); part = (double*)(big + off + 64);
    for (int rep = 0; rep < 2; rep++) {
      hipMemset(bar, 0, 4);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_bar, dim3(nwork * stride), dim3(256), 0, 0, bar, part, nwork, stride, iters, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("stride %d off %10zu: %.3f us per (publish + barrier + read-all)\n", stride, off, 1e3 * ms / iters);
    }
  }
  return 0;
}
