// Groups of G lanes inside a 64-lane wavefront (G = 16, 32 or 64): several small queries share one wavefront, each served by an
// aligned group of G lanes.  A group of 16 is one DPP row, so its scans and reductions are DPP row operations (quad swaps, row
// shifts / rotations: a handful of cycles per step, no LDS crossbar); 32- and 64-lane groups add the gfx9 row_bcast carries or a
// ds_bpermute hop.  Every primitive must be called with ALL 64 lanes active (wavefront-uniform control flow): a group whose query
// is finished keeps executing with neutral values.
#pragma once
#include <hip/hip_runtime.h>

namespace vloam {

typedef unsigned long long u64;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int sw_dpp_or0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }  // lanes without a source see 0

// inclusive prefix sum inside every aligned group of G lanes
template <int G>
__device__ __forceinline__ int grp_scan_incl(int v) {
  static_assert(G == 16 || G == 32 || G == 64, "a group is one, two or four DPP rows");
  v += sw_dpp_or0<0x111, 0xf>(v);  // row_shr:1
  v += sw_dpp_or0<0x112, 0xf>(v);  // row_shr:2
  v += sw_dpp_or0<0x114, 0xf>(v);  // row_shr:4
  v += sw_dpp_or0<0x118, 0xf>(v);  // row_shr:8
  if (G >= 32) v += sw_dpp_or0<0x142, 0xa>(v);  // row_bcast15: rows 1 and 3 take in lane 15 of the row below
  if (G == 64) v += sw_dpp_or0<0x143, 0xc>(v);  // row_bcast31: rows 2 and 3 take in lane 31
  return v;
}

// sum over the group, on every lane of the group
template <int G>
__device__ __forceinline__ int grp_sum(int v) {
  v += sw_dpp_or0<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += sw_dpp_or0<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += sw_dpp_or0<0x124, 0xf>(v);  // row_ror:4
  v += sw_dpp_or0<0x128, 0xf>(v);  // row_ror:8 -> the row's sum on every lane of the row
  if (G >= 32) v += __shfl_xor(v, 16);
  if (G == 64) v += __shfl_xor(v, 32);
  return v;
}

template <int CTRL>
__device__ __forceinline__ u64 sw_dpp_u64(u64 v) {   // both halves through the same lane permutation; lanes without a source see ~0
  const int lo = __builtin_amdgcn_update_dpp(-1, (int)(unsigned)(v & 0xffffffffull), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(-1, (int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, false);
  return ((u64)(unsigned)hi << 32) | (u64)(unsigned)lo;
}

// minimum over the group, on every lane of the group
template <int G>
__device__ __forceinline__ u64 grp_min_u64(u64 v) {
  u64 o;
  o = sw_dpp_u64<0xB1>(v); v = o < v ? o : v;
  o = sw_dpp_u64<0x4E>(v); v = o < v ? o : v;
  o = sw_dpp_u64<0x124>(v); v = o < v ? o : v;
  o = sw_dpp_u64<0x128>(v); v = o < v ? o : v;
  if (G >= 32) { o = __shfl_xor(v, 16); v = o < v ? o : v; }
  if (G == 64) { o = __shfl_xor(v, 32); v = o < v ? o : v; }
  return v;
}

// LDS hand-over between lanes of ONE wavefront.  LDS operations of a wavefront are issued and retire in order, so the hardware needs
// nothing; the wavefront-scope fence only keeps the COMPILER from caching or reordering LDS accesses across the exchange.  (A
// workgroup-scope fence here costs an s_waitcnt vmcnt(0): the wavefront would sit out every global store it still has in flight —
// measured in k_map_assoc: 12 k of 45 k cycles per wavefront behind the candidate-cache stores.)
__device__ __forceinline__ void sw_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

}  // namespace vloam
