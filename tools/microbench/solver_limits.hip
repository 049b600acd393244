// Two facts the Levenberg-Marquardt kernel (csrc/lm_solve.hip) is designed around, measured on the box it runs on:
//  (A) what one exchange of tagged 8-byte granules between NB workgroups costs — agent scope (the memory-side path every XCD sees) with the
//      workgroups dealt round-robin over the XCDs as a plain launch places them, the same with all workgroups on ONE XCD (every eighth
//      workgroup of an 8x launch works), and — same XCD — through that XCD's L2 only (sc0 loads: miss in the compute unit's vector cache,
//      hit in L2; plain stores write through to L2);
//  (B) what a dependent / independent f64 FMA costs a wavefront that is alone on its SIMD, and with a second wavefront next to it.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/solver_limits.hip -o /tmp/solver_limits && /tmp/solver_limits
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;

__device__ __forceinline__ u64 load_sc0(const u64* p) {
  u64 v;
  asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void store_plain(u64* p, u64 v) {
  asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
}

// mode 0: agent-scope atomics; 1: sc0 loads + plain stores (L2 of the XCD).  stride8: only workgroups with blockIdx.x % 8 == 0 work.
template <int NB>
__global__ __launch_bounds__(256) void k_exchange(u64* gran, int iters, int mode, int stride8, long long* out) {
  int blk = blockIdx.x;
  if (stride8) { if (blk & 7) return; blk >>= 3; }
  if (blk >= NB) return;
  const int tid = threadIdx.x;
  long long t0 = 0;
  double acc = 0;
  int timeouts = 0;
  for (int it = 0; it <= iters; it++) {
    if (it == 1) t0 = clock64();   // iteration 0 warms up / lines the workgroups up
    __syncthreads();
    if (tid < 28) {
      u64* g = gran + (size_t)(it & 1) * 8 * 64;
      const u64 tag = (u64)(it + 1) << 32;
      const u64 val = tag | (u64)(unsigned)(tid + blk);
      if (mode == 0) {
        __hip_atomic_store(&g[blk * 64 + tid], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&g[blk * 64 + 32 + tid], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        store_plain(&g[blk * 64 + tid], val);
        store_plain(&g[blk * 64 + 32 + tid], val);
      }
      u64 lo[NB], hi[NB];
      bool got = false;
      for (int spins = 0; spins < (1 << 12); spins++) {
#pragma unroll
        for (int q = 0; q < NB; q++) {
          if (mode == 0) {
            lo[q] = __hip_atomic_load(&g[q * 64 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hi[q] = __hip_atomic_load(&g[q * 64 + 32 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            lo[q] = load_sc0(&g[q * 64 + tid]);
            hi[q] = load_sc0(&g[q * 64 + 32 + tid]);
          }
        }
        bool all = true;
#pragma unroll
        for (int q = 0; q < NB; q++) all = all && (lo[q] >> 32 << 32) == tag && (hi[q] >> 32 << 32) == tag;
        if (all) { got = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (!got) timeouts++;
#pragma unroll
      for (int q = 0; q < NB; q++) acc += (double)(lo[q] & 0xffff);
    }
  }
  __syncthreads();
  if (tid == 0 && blk == 0) { out[0] = clock64() - t0; out[1] = timeouts; }
}

// CHAINS independent chains of dependent f64 FMAs per lane, `len` FMAs each; blockDim = 256 (one wavefront per SIMD) or 512 (two)
template <int CHAINS>
__global__ void k_fma(int len, double seed, long long* out, double* sink) {
  double a[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x * 1e-9;
  const double m = 1.0 - 1e-12, b = 1e-13;
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < len; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) a[c] = fma(a[c], m, b);
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s += a[c];
  if (s == 12345.678) sink[0] = s;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

int main() {
  u64* gran; long long* out; double* sink;
  CK(hipMalloc(&gran, 2 * 8 * 64 * sizeof(u64) + 4096));
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&sink, 64));
  long long h[2];
  const int iters = 200;
  printf("(A) exchange of 28 x 2 tagged granules between NB workgroups, shader cycles per exchange (incl. one __syncthreads)\n");
  for (int nb = 4; nb <= 6; nb += 2)
    for (int cfg = 0; cfg < 3; cfg++) {
      const int mode = cfg == 2 ? 1 : 0, stride8 = cfg >= 1 ? 1 : 0;
      double best = 1e30, sum = 0;
      for (int rep = 0; rep < 5; rep++) {
        CK(hipMemset(gran, 0, 2 * 8 * 64 * sizeof(u64)));
        const int grid = stride8 ? nb * 8 : nb;
        if (nb == 4) hipLaunchKernelGGL(k_exchange<4>, dim3(grid), dim3(256), 0, 0, gran, iters, mode, stride8, out);
        else hipLaunchKernelGGL(k_exchange<6>, dim3(grid), dim3(256), 0, 0, gran, iters, mode, stride8, out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        const double c = (double)h[0] / iters;
        if (c < best) best = c;
        sum += c;
      }
      printf("  NB = %d  %-44s  best %7.0f  mean %7.0f cycles  (exchanges that timed out in the last run: %lld)\n", nb,
             cfg == 0 ? "agent scope, workgroups over the XCDs" : (cfg == 1 ? "agent scope, all workgroups on one XCD" : "through the XCD's L2 (sc0 loads), one XCD"), best, sum / 5, h[1]);
      fflush(stdout);
    }
  printf("(B) f64 FMA: cycles per FMA instruction of one wavefront (clock64 ticks at 100 MHz are NOT used: s_memtime = shader clock)\n");
  const int len = 4096;
  for (int threads = 256; threads <= 1024; threads *= 2) {
    for (int chains = 1; chains <= 8; chains *= 2) {
      if (chains == 1) hipLaunchKernelGGL(k_fma<1>, dim3(1), dim3(threads), 0, 0, len, 0.5, out, sink);
      if (chains == 2) hipLaunchKernelGGL(k_fma<2>, dim3(1), dim3(threads), 0, 0, len, 0.5, out, sink);
      if (chains == 4) hipLaunchKernelGGL(k_fma<4>, dim3(1), dim3(threads), 0, 0, len, 0.5, out, sink);
      if (chains == 8) hipLaunchKernelGGL(k_fma<8>, dim3(1), dim3(threads), 0, 0, len, 0.5, out, sink);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
      printf("  %4d threads (%d wavefront%s per SIMD), %d independent chain%s per lane: %6.2f cycles per FMA of a wavefront, %6.2f per FMA issued on the SIMD\n", threads, threads / 256,
             threads > 256 ? "s" : "", chains, chains > 1 ? "s" : "", (double)h[0] / ((double)len * chains), (double)h[0] / ((double)len * chains * (threads / 256)));
      fflush(stdout);
    }
  }
  return 0;
}
