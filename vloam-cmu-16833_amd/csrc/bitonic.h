// Bitonic sorting network on u64 keys in LDS for a workgroup of `nthreads` lanes (a multiple of 64): strides below 64 are cross-lane
// exchanges on keys held in registers, only the wide strides go through LDS behind a workgroup barrier.  Shared by the scan-registration
// ring kernel (voxel runs of a scan line, sr_kernels.hip) and the mapping stage's scan-feature VoxelGrid (map_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace vloam {

typedef unsigned long long u64;

// One stage of the network for `nthreads` cooperating lanes: each lane fetches both operands of up to four compare-exchanges
// before writing any of them back (the exchanges of a stage touch disjoint pairs), so the LDS round trips overlap.
__device__ __forceinline__ void bitonic_stage(u64* a, int P, int j, int k, int tid, int nthreads) {
  for (int t0 = tid; t0 < P / 2; t0 += 4 * nthreads) {
    u64 x[4], y[4];
    int ii[4], ll[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int t = t0 + u * nthreads;
      ii[u] = ((t & ~(j - 1)) << 1) | (t & (j - 1));
      ll[u] = ii[u] | j;
      if (t < P / 2) { x[u] = a[ii[u]]; y[u] = a[ll[u]]; }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int t = t0 + u * nthreads;
      if (t < P / 2 && ((x[u] > y[u]) == ((ii[u] & k) == 0))) { a[ii[u]] = y[u]; a[ll[u]] = x[u]; }
    }
  }
}

// Merge levels k_lo .. k_hi (doubling) of an ASCENDING bitonic network, strides min(k / 2, 64) .. 1, on 128-element blocks held in
// registers: wavefront w of the workgroup owns blocks w, w + nwaves, ...; lane l holds elements l and l + 64 of its block, so a stride
// below 64 is a cross-lane exchange and stride 64 the lane's own pair — no LDS round trip, no barrier between these stages.
__device__ __forceinline__ void bitonic_reg_stages(u64* a, int P, int k_lo, int k_hi, int tid, int nthreads) {
  const int lane = tid & 63, wv = tid >> 6, nwaves = nthreads >> 6;
  for (int blk = wv; blk * 128 < P; blk += nwaves) {
    const int base = blk * 128;
    u64 a0 = a[base + lane], a1 = a[base + 64 + lane];
    for (int k = k_lo; k <= k_hi; k <<= 1) {
      const bool up0 = ((base + lane) & k) == 0, up1 = ((base + 64 + lane) & k) == 0;
      if (k > 64) {  // stride 64 (both elements see the same direction: bit k lies above bit 6)
        const bool swap = (a0 > a1) == up0;
        const u64 t0 = swap ? a1 : a0, t1 = swap ? a0 : a1;
        a0 = t0; a1 = t1;
      }
      for (int j = (k > 64 ? 32 : k >> 1); j > 0; j >>= 1) {
        const u64 b0 = __shfl_xor(a0, j), b1 = __shfl_xor(a1, j);
        const bool lower = (lane & j) == 0;   // the lower lane of an ascending pair keeps the smaller key: one compare, one exchange decision
        a0 = ((a0 > b0) == (up0 == lower)) ? b0 : a0;
        a1 = ((a1 > b1) == (up1 == lower)) ? b1 : a1;
      }
    }
    a[base + lane] = a0; a[base + 64 + lane] = a1;
  }
}

// The same with 256-element blocks, four keys per lane (elements l, l + 64, l + 128, l + 192): strides 128 and 64 are the lane's own pairs,
// anything below a cross-lane exchange.  2 048 run keys are then eight blocks — one per wavefront — and only the strides >= 256 (6 of
// the 66 stages) go through LDS behind a workgroup barrier.
__device__ __forceinline__ void bitonic_reg_stages4(u64* a, int P, int k_lo, int k_hi, int tid, int nthreads) {
  const int lane = tid & 63, wv = tid >> 6, nwaves = nthreads >> 6;
  for (int blk = wv; blk * 256 < P; blk += nwaves) {
    const int base = blk * 256;
    u64 v[4];
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = a[base + e * 64 + lane];
    for (int k = k_lo; k <= k_hi; k <<= 1) {
      bool up[4];
#pragma unroll
      for (int e = 0; e < 4; e++) up[e] = ((base + e * 64 + lane) & k) == 0;
      if (k > 128) {  // stride 128: (0, 2) and (1, 3); both ends of a pair see the same direction (bit k lies above bit 7)
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const u64 x = v[e], y = v[e + 2];
          const bool swap = (x > y) == up[e];
          v[e] = swap ? y : x; v[e + 2] = swap ? x : y;
        }
      }
      if (k > 64) {   // stride 64: (0, 1) and (2, 3)
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const u64 x = v[e], y = v[e + 1];
          const bool swap = (x > y) == up[e];
          v[e] = swap ? y : x; v[e + 1] = swap ? x : y;
        }
      }
      for (int j = (k > 64 ? 32 : k >> 1); j > 0; j >>= 1) {
        const bool lower = (lane & j) == 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const u64 b = __shfl_xor(v[e], j);
          v[e] = ((v[e] > b) == (up[e] == lower)) ? b : v[e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) a[base + e * 64 + lane] = v[e];
  }
}

// Ascending sort of P2 keys (a power of two >= 256) by the whole workgroup: 256-key blocks in registers (four keys per lane), strides >= 256
// through LDS.  Ends behind a workgroup barrier.
__device__ __forceinline__ void block_bitonic_sort_u64(u64* a, int P2, int tid, int nthreads) {
  bitonic_reg_stages4(a, P2, 2, 256, tid, nthreads);
  __syncthreads();
  for (int k = 512; k <= P2; k <<= 1) {
    for (int j = k >> 1; j >= 256; j >>= 1) {
      bitonic_stage(a, P2, j, k, tid, nthreads);
      __syncthreads();
    }
    bitonic_reg_stages4(a, P2, k, k, tid, nthreads);
    __syncthreads();
  }
}

}  // namespace vloam
