// scanRegistration on gfx950: NaN / min-range removal, ring + relTime labelling, stable per-ring
// compaction, 11-tap curvature, greedy sharp / flat pick per sector (== walking the sorted sector), per-ring VoxelGrid(0.2).
// Restates ScanRegistration::input, /root/reference/src/lidar_odometry_mapping/src/scan_registration.cpp:131-449
// (cited per step as "SR:<line>").  Integer / index results are bit-identical to the CPU oracle; f32
// arithmetic follows the reference's expression order with FMA contraction disabled at compile time.
//
// Kernels (one sweep = 6 launches, no host synchronisation):
//   k_sr_first_last  16 WGs      first / last surviving point: eight strided walkers from each end of the cloud (their records are folded by
//                                every k_sr_label workgroup)                                                                    SR:157-176
//   k_sr_label       n/1024 WGs  scanID, raw ori, halfPassed pivot (atomicMin), ring histogram SR:186-262   (1 024 lanes x 1 point for one or two
//   k_sr_scatter     n/1024 WGs  ring offsets + per-WG bases (SR:276-281), relTime / intensity, stable scatter SR:264-266   sessions, 256 x 4 for a batch)
//   k_sr_ring        1 WG/ring   LDS-resident ring: curvature, sort-free picks (wavefront arg-max rounds, six sectors
//                                at once + boundary fixed point), lessFlat + VoxelGrid(0.2) over voxel runs  SR:288-439
//                                (two capacity tiers: 2 176 points in 78.75 KB of LDS = two rings per CU; the 4 096-point tier follows on every
//                                sweep — its full grid while long rings are around, one catch-all workgroup otherwise)
//   k_sr_compact     1 WG/ring   ring/sector-ordered feature clouds (+ per-ring bounding boxes of the less-clouds for the mapping
//                                stage's VoxelGrid)                                                 SR:338-344,388,439
#include <hip/hip_runtime.h>
#include <limits.h>
#include <float.h>
#include <type_traits>
#include <math.h>
#include "sr_kernels.h"
#include "bitonic.h"
#include "fdlibm_f32.h"
#include <cstdlib>

namespace vloam {

typedef unsigned long long u64;

__device__ __forceinline__ void lds_fence_wave() {
  // LDS ops of one wavefront retire in order; this only has to stop the compiler from caching or
  // reordering LDS accesses across the point where lanes exchange data.
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// Wavefront-wide unsigned max / min through DPP row operations (no LDS crossbar, a handful of cycles per step): quad swaps,
// row rotations, then the row_bcast15 / row_bcast31 carries of gfx9; the result lands in lane 63 and is broadcast.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_max_step(unsigned v) {
  const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);  // lanes without a source keep 0
  return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = dpp_max_step<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_max_step<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_max_step<0x124, 0xf>(v);  // row_ror:4
  v = dpp_max_step<0x128, 0xf>(v);  // row_ror:8   -> every lane holds the max of its row of 16
  v = dpp_max_step<0x142, 0xa>(v);  // row_bcast15 -> rows 1 and 3 fold in the row below
  v = dpp_max_step<0x143, 0xc>(v);  // row_bcast31 -> rows 2 and 3 fold in lane 31
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) { return ~wave_max_u32(~v); }
// The same steps for f32 min / max (finite values) and an integer sum; lanes without a source combine with themselves (min / max) or 0 (sum).
template <int CTRL, int ROW_MASK, bool MAX>
__device__ __forceinline__ float dpp_fminmax_step(float v) {
  const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
  return MAX ? fmaxf(o, v) : fminf(o, v);
}
template <bool MAX>
__device__ __forceinline__ float wave_fminmax(float v) {
  v = dpp_fminmax_step<0xB1, 0xf, MAX>(v);
  v = dpp_fminmax_step<0x4E, 0xf, MAX>(v);
  v = dpp_fminmax_step<0x124, 0xf, MAX>(v);
  v = dpp_fminmax_step<0x128, 0xf, MAX>(v);
  v = dpp_fminmax_step<0x142, 0xa, MAX>(v);
  v = dpp_fminmax_step<0x143, 0xc, MAX>(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add_step(int v) { return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int wave_sum_i32(int v) {
  v = dpp_add_step<0xB1, 0xf>(v);
  v = dpp_add_step<0x4E, 0xf>(v);
  v = dpp_add_step<0x124, 0xf>(v);
  v = dpp_add_step<0x128, 0xf>(v);
  v = dpp_add_step<0x142, 0xa>(v);
  v = dpp_add_step<0x143, 0xc>(v);
  return __builtin_amdgcn_readlane(v, 63);
}

// SR:157 removeNaNFromPointCloud + SR:100-129 removeClosedPointCloud
__device__ __forceinline__ bool sr_survives_s1(float x, float y, float z, float thres) {
  if (!isfinite(x) || !isfinite(y) || !isfinite(z)) return false;
  if (x * x + y * y + z * z < thres * thres) return false;
  return true;
}

// atan2f / atanf exactly as glibc (fdlibm float) computes them on the reference's platform — fdlibm_f32.h; OCML's bodies differ from them in the
// last bit on ~10 % of the arguments, which moves returns that sit on a scan-line bin edge or on an unwrap threshold.  One out-of-line body
// each: every caller, every batch size, the same instruction sequence.
__device__ __noinline__ float sr_atan2f(float y, float x) { return fd_atan2f(y, x); }
__device__ __noinline__ float sr_atanf(float v) { return fd_atanf(v); }

// SR:192-226.  Returns the ring id or -1 when the point is dropped.
__device__ __forceinline__ int sr_scan_id(float x, float y, float z, int N_SCANS) {
  float angle = (float)((double)(sr_atanf(z / sqrtf(x * x + y * y)) * 180) / M_PI);
  // a return at the origin itself (0 / 0; it survives S1 only with minimum_range <= 0): int(NaN) is INT_MIN on the reference's x86-64
  // (cvttsd2si), i.e. "scanID < 0" in every branch below — dropped; the GPU's conversion would give 0, i.e. scan line 0 / 32
  if (angle != angle) return -1;
  int scanID = 0;
  if (N_SCANS == 16) {
    scanID = int((double)((angle + 15) / 2) + 0.5);
    if (scanID > (N_SCANS - 1) || scanID < 0) return -1;
  } else if (N_SCANS == 32) {
    scanID = int(((double)angle + 92.0 / 3.0) * 3.0 / 4.0);
    if (scanID > (N_SCANS - 1) || scanID < 0) return -1;
  } else {
    if ((double)angle >= -8.83) scanID = int((double)(2 - angle) * 3.0 + 0.5);
    else scanID = N_SCANS / 2 + int((-8.83 - (double)angle) * 2.0 + 0.5);
    if (angle > 2 || (double)angle < -24.33 || scanID > 50 || scanID < 0) return -1;
  }
  return scanID;
}

// SR:237-244, the !halfPassed branch
__device__ __forceinline__ float sr_ori_first_half(float ori, float startOri) {
  if ((double)ori < (double)startOri - M_PI / 2) ori = (float)((double)ori + 2 * M_PI);
  else if ((double)ori > (double)startOri + M_PI * 3 / 2) ori = (float)((double)ori - 2 * M_PI);
  return ori;
}
// SR:253-261, the halfPassed branch
__device__ __forceinline__ float sr_ori_second_half(float ori, float endOri) {
  ori = (float)((double)ori + 2 * M_PI);
  if ((double)ori < (double)endOri - M_PI * 3 / 2) ori = (float)((double)ori + 2 * M_PI);
  else if ((double)ori > (double)endOri + M_PI / 2) ori = (float)((double)ori - 2 * M_PI);
  return ori;
}

// ------------------------------------------------------------------------------------------------
// First / last surviving point (SR:157-176) in two steps without a single-workgroup stage on the critical path: every 256-lane
// workgroup of k_sr_first_last reports the first / last surviving point of its slice (one pass over the input at full width);
// every workgroup of k_sr_label folds the ~512 slice records itself (4 KB from L2) and derives startOri / endOri locally.
// 256-lane workgroups: small enough to slip onto CUs whose register file is mostly taken by the previous sweep's odometry
// kernels (the stages of consecutive sweeps overlap), where a 1024-lane workgroup would have to wait for them to drain.
constexpr int kFLThreads = 256;
constexpr int kFLPoints = 4096;   // points a workgroup looks at per trip (sixteen loads in flight per lane: a cloud that opens with a few thousand dropped returns is still one trip)
// 2 x kFLWalkers workgroups per session: kFLWalkers walk the cloud from the front — walker j looks at trips j, j + kFLWalkers, ... and stops at
// its first trip with a surviving point — and kFLWalkers from the back.  The first surviving point of the cloud is the smallest of the
// front walkers' answers (every k_sr_label workgroup folds the records), the last one the largest of the back walkers'.  On a real sweep that
// is one trip each, 32 KB per walker of the 2 MB cloud (rounds 1 - 4 read all of it for these two indices); a ring-major cloud that closes
// with ten rings of dropped returns (the synthetic one does: five trips, 15.6 us with ONE walker per end) still is one trip deep.
// slice[j] = (first, -1) for a front walker, (INT_MAX, last) for a back walker.
constexpr int kFLWalkers = 8;
__global__ __launch_bounds__(kFLThreads) void k_sr_first_last(BatchIn bi, float thres, int2* __restrict__ slice, size_t ss) {
  VL_SESSION(ss); RB(slice);
  const float4* __restrict__ in = bi.in[blockIdx.z];
  const int n = bi.n[blockIdx.z];
  __shared__ int s_first, s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const bool back = blockIdx.x >= kFLWalkers;
  const int walker = blockIdx.x - (back ? kFLWalkers : 0);
  if (tid == 0) { s_first = INT_MAX; s_last = -1; }
  __syncthreads();
  const int ntrip = (n + kFLPoints - 1) / kFLPoints;
  for (int trip = walker; trip < ntrip; trip += kFLWalkers) {
    const int t0 = (back ? ntrip - 1 - trip : trip) * kFLPoints;
    float4 p[kFLPoints / kFLThreads];
#pragma unroll
    for (int e = 0; e < kFLPoints / kFLThreads; e++) {
      const int i = t0 + e * kFLThreads + tid;
      p[e] = i < n ? in[i] : make_float4(NAN, NAN, NAN, 0.f);
    }
    int first = INT_MAX, last = -1;
#pragma unroll
    for (int e = 0; e < kFLPoints / kFLThreads; e++) {
      const int i = t0 + e * kFLThreads + tid;
      const bool v = i < n && sr_survives_s1(p[e].x, p[e].y, p[e].z, thres);
      const unsigned long long m = __ballot(v);
      if (m != 0ull) {
        const int base = i - lane;
        first = min(first, base + __ffsll((long long)m) - 1);
        last = max(last, base + 63 - __clzll((long long)m));
      }
    }
    if (lane == 0 && last >= 0) { atomicMin(&s_first, first); atomicMax(&s_last, last); }
    __syncthreads();
    if (s_last >= 0) break;   // (uniform: read behind the barrier)
    __syncthreads();
  }
  if (tid == 0) slice[blockIdx.x] = back ? make_int2(INT_MAX, s_last) : make_int2(s_first, -1);
}

// ------------------------------------------------------------------------------------------------
// blk: per-workgroup results for k_sr_scatter — [0, nblk): candidate pivot (SR:246-249) or INT_MAX, [nblk, 2 nblk): points
// surviving S1.  Plain stores, no counters to re-arm between sweeps.
// A label / scatter workgroup owns kLabelBlock points and comes in two shapes: kLabelThreads = 256 lanes with four points each (chunk e =
// points e * 256 .. e * 256 + 255) — a quarter of the wavefronts, what a batch of sessions wants (the chip is short of wave slots there) —
// and 1024 lanes with one point each for one or two sessions, where the 128 workgroups of a sweep leave the chip mostly empty and four
// dependent rounds per lane only add latency (B = 1: label 9.3 -> 7.4 us, scatter 12.2 -> 7.0 us).  Same blocks, same results.
template <int kLabelThreads>
__global__ __launch_bounds__(kLabelThreads) void k_sr_label(BatchIn bi, float thres, int N_SCANS,
                                                           FrameScalars* S, signed char* __restrict__ sid,
                                                           float* __restrict__ ori_raw, int* __restrict__ blockhist,
                                                           const int2* __restrict__ slice, int nslice, int* __restrict__ blk, size_t ss) {
  VL_SESSION(ss); RB(S); RB(sid); RB(ori_raw); RB(blockhist); RB(slice); RB(blk);
  constexpr int kLabelPer = kLabelBlock / kLabelThreads;
  const float4* __restrict__ in = bi.in[blockIdx.z];
  const int n = bi.n[blockIdx.z];
  __shared__ int hist[kMaxRings];
  __shared__ int s_istar, s_cnt, s_first, s_last;
  __shared__ float s_start;
  const int tid = threadIdx.x, lane = tid & 63;
  // this workgroup's points first: their loads are in flight while the slice records are folded
  float4 p[kLabelPer];
#pragma unroll
  for (int e = 0; e < kLabelPer; e++) {
    const int i = blockIdx.x * kLabelBlock + e * kLabelThreads + tid;
    p[e] = i < n ? in[i] : make_float4(NAN, NAN, NAN, 0.f);
  }
  if (tid < kMaxRings) hist[tid] = 0;
  if (tid == 0) { s_istar = INT_MAX; s_cnt = 0; s_first = INT_MAX; s_last = -1; }
  __syncthreads();
  {
    int f = INT_MAX, l = -1;
    for (int b = tid; b < nslice; b += kLabelThreads) { const int2 fl = slice[b]; f = min(f, fl.x); l = max(l, fl.y); }
    for (int d = 32; d > 0; d >>= 1) { f = min(f, __shfl_xor(f, d)); l = max(l, __shfl_xor(l, d)); }
    if (lane == 0) { atomicMin(&s_first, f); atomicMax(&s_last, l); }
  }
  __syncthreads();
  if (tid == 0) {
    float startOri = 0.f, endOri = 0.f;
    if (s_last >= 0) {
      const float4 pf = in[s_first], pl = in[s_last];
      startOri = -sr_atan2f(pf.y, pf.x);                                                // SR:166
      endOri = (float)((double)(-sr_atan2f(pl.y, pl.x)) + 2 * M_PI);                    // SR:167
      if ((double)(endOri - startOri) > 3 * M_PI) endOri = (float)((double)endOri - 2 * M_PI);        // SR:169-172
      else if ((double)(endOri - startOri) < M_PI) endOri = (float)((double)endOri + 2 * M_PI);      // SR:173-176
    }
    s_start = startOri;
    if (blockIdx.x == 0) {  // the sweep's first writer of the scalars: also clears the error word
      S->first_valid = s_first == INT_MAX ? -1 : s_first;
      S->last_valid = s_last;
      S->startOri = startOri; S->endOri = endOri;
      S->error = s_last < 0 ? kErrEmpty : 0;
    }
  }
  __syncthreads();
  const float startOri = s_start;
  int nv = 0, istar = INT_MAX;
#pragma unroll
  for (int e = 0; e < kLabelPer; e++) {
    const int i = blockIdx.x * kLabelBlock + e * kLabelThreads + tid;
    int id = -1;
    bool v1 = false;
    if (i < n) {
      v1 = sr_survives_s1(p[e].x, p[e].y, p[e].z, thres);
      if (v1) {
        id = sr_scan_id(p[e].x, p[e].y, p[e].z, N_SCANS);
        float ori = -sr_atan2f(p[e].y, p[e].x);  // SR:234
        ori_raw[i] = ori;
        if (id >= 0) {
          atomicAdd(&hist[id], 1);
          float o = sr_ori_first_half(ori, startOri);
          if ((double)(o - startOri) > M_PI) istar = min(istar, i);  // SR:246-249 candidate pivot
        }
      }
      sid[i] = (signed char)id;
    }
    nv += __popcll(__ballot(v1));
  }
  if (istar != INT_MAX) atomicMin(&s_istar, istar);
  if (lane == 0 && nv) atomicAdd(&s_cnt, nv);
  __syncthreads();
  if (tid < kMaxRings) blockhist[blockIdx.x * kMaxRings + tid] = hist[tid];
  if (tid == 0) { blk[blockIdx.x] = s_istar; blk[gridDim.x + blockIdx.x] = s_cnt; }
}

// ------------------------------------------------------------------------------------------------
// Stable scatter into the ring-major cloud.  Every workgroup first derives, from the per-WG ring histograms of k_sr_label,
// the ring offsets (SR:276-281) and its own base inside every ring — 32 KB of L2 reads per WG instead of a separate
// single-workgroup scan kernel on the critical path.
template <int kLabelThreads>
__global__ __launch_bounds__(kLabelThreads) void k_sr_scatter(BatchIn bi, FrameScalars* S,
                                                             const signed char* __restrict__ sid, const float* __restrict__ ori_raw,
                                                             const int* __restrict__ blockhist, int nblk, float4* __restrict__ cloud,
                                                             const int* __restrict__ blk, size_t ss) {
  VL_SESSION(ss); RB(S); RB(sid); RB(ori_raw); RB(blockhist); RB(cloud); RB(blk);
  const float4* __restrict__ in = bi.in[blockIdx.z];
  const int n = bi.n[blockIdx.z];
  constexpr int kLabelPer = kLabelBlock / kLabelThreads;
  constexpr int kChunks = kLabelBlock / 64, kWaves = kLabelThreads / 64;   // 64-point chunks of the workgroup's points, in input order: chunk c = e * kWaves + wave
  __shared__ int wcnt[kChunks][kMaxRings];
  __shared__ int s_istar, s_nvalid;
  __shared__ int part_before[kWaves][kMaxRings], part_all[kWaves][kMaxRings];
  __shared__ int ring_base[kMaxRings];   // ring offset + points of this ring in earlier workgroups
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // this workgroup's points, ring ids and raw azimuths first (in flight during the prologue)
  float4 p[kLabelPer];
  float oraw[kLabelPer];
  int ids[kLabelPer];
#pragma unroll
  for (int e = 0; e < kLabelPer; e++) {
    const int i = blockIdx.x * kLabelBlock + e * kLabelThreads + tid;
    ids[e] = (i < n) ? (int)sid[i] : -1;
    p[e] = i < n ? in[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    oraw[e] = i < n ? ori_raw[i] : 0.f;
  }
  for (int k = tid; k < kChunks * kMaxRings; k += kLabelThreads) (&wcnt[0][0])[k] = 0;
  if (tid == 0) { s_istar = INT_MAX; s_nvalid = 0; }
  __syncthreads();
  {
    // the pivot of the sweep = the smallest candidate of any label workgroup; the S1 survivor count is a debug scalar
    int mi = INT_MAX, cnt = 0;
    for (int b = tid; b < nblk; b += kLabelThreads) { mi = min(mi, blk[b]); cnt += blk[nblk + b]; }
    for (int d = 32; d > 0; d >>= 1) { mi = min(mi, __shfl_xor(mi, d)); cnt += __shfl_xor(cnt, d); }
    if (lane == 0) { atomicMin(&s_istar, mi); atomicAdd(&s_nvalid, cnt); }
  }
  {
    // wavefront w sums blocks w, w + kWaves, ... for ring = lane
    int before = 0, all = 0;
    for (int b = wave; b < nblk; b += kWaves) {
      const int h = blockhist[b * kMaxRings + lane];
      all += h;
      if (b < (int)blockIdx.x) before += h;
    }
    part_before[wave][lane] = before;
    part_all[wave][lane] = all;
  }
  __syncthreads();
  if (tid < kMaxRings) {
    int before = 0, all = 0;
    for (int w = 0; w < kWaves; w++) { before += part_before[w][tid]; all += part_all[w][tid]; }
    // exclusive prefix of the ring totals across the 64 rings of this wavefront
    int inc = all;
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (tid >= d) inc += t; }
    const int roff = inc - all;
    ring_base[tid] = roff + before;
    if (blockIdx.x == 0) {
      S->ring_count[tid] = all;
      S->ring_off[tid] = roff;
      S->scanStartInd[tid] = roff + 5;         // SR:278
      S->scanEndInd[tid] = roff + all - 6;     // SR:280
      if (tid == kMaxRings - 1) { S->N2 = inc; S->ring_off[kMaxRings] = inc; S->istar = s_istar; S->n_after_s1 = s_nvalid; }
    }
  }
  // stable rank of every point among the same-ring points of its 64-point chunk
  int rank[kLabelPer];
#pragma unroll
  for (int e = 0; e < kLabelPer; e++) {
    const int id = ids[e];
    rank[e] = 0;
    bool pending = id >= 0;
    while (true) {
      u64 act = __ballot(pending);
      if (act == 0) break;
      int leader = __ffsll((long long)act) - 1;
      int lid = __shfl(id, leader);
      u64 same = __ballot(pending && id == lid);
      if (pending && id == lid) {
        rank[e] = __popcll(same & ((1ull << lane) - 1ull));
        pending = false;
      }
      if (lane == leader) wcnt[e * kWaves + wave][lid] = __popcll(same);
    }
  }
  __syncthreads();
  const float startOri = S->startOri, endOri = S->endOri;
  const int istar = s_istar;
#pragma unroll
  for (int e = 0; e < kLabelPer; e++) {
    const int id = ids[e];
    if (id < 0) continue;
    const int i = blockIdx.x * kLabelBlock + e * kLabelThreads + tid;
    int base = ring_base[id];
    for (int c = 0; c < e * kWaves + wave; c++) base += wcnt[c][id];
    float ori = oraw[e];
    if (i <= istar) ori = sr_ori_first_half(ori, startOri);
    else ori = sr_ori_second_half(ori, endOri);
    float relTime = (ori - startOri) / (endOri - startOri);  // SR:264
    float4 o;
    o.x = p[e].x; o.y = p[e].y; o.z = p[e].z;
    o.w = (float)((double)id + 0.1 * (double)relTime);  // SR:265, scanPeriod = 0.1 (scan_registration.h:84)
    cloud[base + rank[e]] = o;
  }
}

// ------------------------------------------------------------------------------------------------
// Block-wide helpers for k_sr_ring (512 threads = 8 wavefronts).
constexpr int kRingThreads = 512;

// In-place exclusive scan of an LDS int array (n <= 16 * kRingThreads); returns the total.  Per-thread chunk sums, a shuffle scan inside
// every wavefront, the eight wavefront totals through LDS: three workgroup barriers (a Hillis-Steele scan over 512 partial sums took 18).
__device__ int block_exclusive_scan(int* a, int n, int* scratch /* kRingThreads ints */) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int per = (n + kRingThreads - 1) / kRingThreads;
  const int lo = tid * per, hi = min(lo + per, n);
  int s = 0;
  for (int k = lo; k < hi; k++) s += a[k];
  int inc = s;
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  __syncthreads();   // (scratch may still be read by the caller's previous use)
  if (lane == 63) scratch[wv] = inc;
  __syncthreads();
  int base = 0, total = 0;
  for (int w = 0; w < kRingThreads / 64; w++) { const int t = scratch[w]; base += w < wv ? t : 0; total += t; }
  int run = base + inc - s;
  for (int k = lo; k < hi; k++) { int v = a[k]; a[k] = run; run += v; }
  __syncthreads();
  return total;
}

__device__ __forceinline__ void bitonic_step(u64* a, int t, int j, int k) {
  const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
  const int l = i | j;
  const bool up = (i & k) == 0;
  const u64 x = a[i], y = a[l];
  if ((x > y) == up) { a[i] = y; a[l] = x; }
}

// SR:353-376 == SR:397-420.  Serial neighbour suppression around local index ind (one lane).
__device__ void sr_spread(int ind, const float* px, const float* py, const float* pz, unsigned char* picked) {
  for (int l = 1; l <= 5; l++) {
    float dx = px[ind + l] - px[ind + l - 1];
    float dy = py[ind + l] - py[ind + l - 1];
    float dz = pz[ind + l] - pz[ind + l - 1];
    if ((double)(dx * dx + dy * dy + dz * dz) > 0.05) break;
    picked[ind + l] = 1;
  }
  for (int l = -1; l >= -5; l--) {
    float dx = px[ind + l] - px[ind + l + 1];
    float dy = py[ind + l] - py[ind + l + 1];
    float dz = pz[ind + l] - pz[ind + l + 1];
    if ((double)(dx * dx + dy * dy + dz * dz) > 0.05) break;
    picked[ind + l] = 1;
  }
}

// Two capacity tiers of the LDS-resident ring.  CAP = kRingCapSmall covers every real HDL-64E / HDL-32 / VLP-16 ring (a revolution
// has <= ~2 100 firings) in 78.75 KB of LDS, so TWO rings share a CU (and other kernels' workgroups still find LDS next to one);
// the kMaxRingLen tier (146 KB, one workgroup per CU) is launched right behind it and only works on rings the small tier had to
// leave alone (len > kRingCapSmall).
constexpr int kRingCapSmall = 2176, kSectCapSmall = 512, kRingWatch = 2144;   // an HDL-64E revolution at 10 Hz has <= 2 083 firings per laser
template <int CAP, int SECT>
constexpr size_t sr_ring_keys_bytes() { return sizeof(u64) * kSectors * SECT > (size_t)12 * CAP ? sizeof(u64) * kSectors * SECT : (size_t)12 * CAP; }

template <int CAP, int SECT, bool BIG_TIER>
__global__ __launch_bounds__(kRingThreads) void k_sr_ring(const float4* __restrict__ cloud, FrameScalars* S, int* __restrict__ sharp_idx,
                                                          int* __restrict__ less_sharp_idx, int* __restrict__ flat_idx,
                                                          float4* __restrict__ ring_ds, float* __restrict__ dbg_curv,
                                                          int* __restrict__ dbg_sort, int* __restrict__ dbg_picked,
                                                          int* __restrict__ dbg_label, long long* __restrict__ dbg_cyc /* [rings][8] */,
                                                          int* ring_watch /* host-mapped [sessions] */, int big_follows, size_t ss) {
  VL_SESSION(ss); RB(cloud); RB(S); RB(sharp_idx); RB(less_sharp_idx); RB(flat_idx); RB(ring_ds); RB(dbg_curv); RB(dbg_sort); RB(dbg_picked);
  RB(dbg_label); RB(dbg_cyc);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* px = (float*)smem;                          // [CAP]
  float* py = px + CAP;
  float* pz = py + CAP;
  float* pi = pz + CAP;                              // intensity
  u64* keys = (u64*)(pi + CAP);                      // debug sort: [kSectors * SECT]; VoxelGrid: run keys [CAP] + voxel ids [CAP]
  int* iscratch = (int*)((unsigned char*)keys + sr_ring_keys_bytes<CAP, SECT>());  // [CAP] heads / ranks
  int* scan_tmp = iscratch + CAP;                    // [kRingThreads]
  unsigned char* picked = (unsigned char*)(scan_tmp + kRingThreads);  // [CAP]
  signed char* label = (signed char*)(picked + CAP);                 // [CAP]
  unsigned char* gap = (unsigned char*)(label + CAP);                // [CAP]
  unsigned char* reachb = gap + CAP;                                 // [CAP]
  int* s_sp = (int*)(reachb + CAP);                                   // [kSectors]   (all LDS lives in the dynamic
  int* s_ep = s_sp + 8;                                              // [kSectors]    region so its base stays 16-B aligned)
  float* s_red = (float*)(s_ep + 8);                                 // [6]
  int* s_ncand_p = (int*)(s_red + 8);
  int* s_leak_lo = s_ncand_p + 8;                                     // [kSectors] lowest / highest local index marked by
  int* s_leak_hi = s_leak_lo + 8;                                    // [kSectors] sector s's picks (unclipped)
  int* s_zone = s_leak_hi + 8;                                       // [kSectors] incoming spill the sector was computed with
  int* s_any = s_zone + 8;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // One workgroup per ring (gridDim.x == kMaxRings) — or, as the catch-all behind the small tier on sweeps for which the host has
  // not launched the full big tier, ONE workgroup per session that walks the rings and works on the (normally zero) oversized ones.
  auto one_ring = [&](const int r) {
  long long tstamp[8];
  int nstamp = 0;
#define SR_STAMP() do { if (dbg_cyc) tstamp[nstamp] = clock64(); nstamp++; } while (0)
  SR_STAMP();
  const int len = S->ring_count[r], off = S->ring_off[r];
  const int start = off + 5, end = off + len - 6;  // SR:278-280
  if (BIG_TIER && len <= kRingCapSmall) return;    // the small tier has done this ring
  if (tid < kSectors * 3) (&S->sect_cnt[r][0][0])[tid] = 0;
  if (tid == 0) S->ring_ds_cnt[r] = 0;
  // The big tier asks for 146 KB of LDS just to start, which on a busy chip (batches) costs tens of microseconds even when it has
  // nothing to do: the host only launches its full grid while rings near the small tier's capacity have been seen (ring_watch, a
  // host-mapped word it polls without synchronising) or during the first sweeps; otherwise a single catch-all workgroup per session
  // follows the small tier, so that a ring which outgrows the small tier without that warning is still processed (slowly: the rings
  // one after the other, until the host has seen the watch word).
  if (!BIG_TIER && tid == 0 && len > kRingWatch && ring_watch) __hip_atomic_store(&ring_watch[blockIdx.z], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (len > CAP) { if ((BIG_TIER || !big_follows) && tid == 0) atomicOr(&S->error, kErrRingTooLong); return; }   // small tier: left to the big tier
  if (end - start < 6) return;  // SR:314

  for (int l = tid; l < len; l += kRingThreads) {
    float4 p = cloud[off + l];
    px[l] = p.x; py[l] = p.y; pz[l] = p.z; pi[l] = p.w;
    picked[l] = 0; label[l] = 0;  // SR:305-306
  }
  if (tid < kSectors) {
    s_sp[tid] = start + (end - start) * tid / 6;            // SR:319
    s_ep[tid] = start + (end - start) * (tid + 1) / 6 - 1;  // SR:320
  }
  __syncthreads();
  SR_STAMP();

  // ---- neighbour-suppression reach of every point, computed once in parallel (SR:353-376 walks outwards while consecutive
  // points are closer than sqrt(0.05) m): gap bit l = 1 when dist2(p[l+1], p[l]) > 0.05; reach[l] = (how many of l-1..l-5) |
  // (how many of l+1..l+5) << 4 a pick at l would mark.  The gap bits of 64 consecutive points are one ballot word; a point then
  // reads its five bits forward and five bits backward out of three words (count of trailing / leading zeros) instead of walking up to
  // ten bytes, each a dependent LDS round trip.
  u64* gapw = (u64*)gap;   // [CAP / 64 + 1] ballot words (the byte array's place)
  for (int base = wave * 64; base < len; base += kRingThreads) {   // wavefront-uniform
    const int l = base + lane;
    bool g = true;
    if (l + 1 < len) {
      const float dx = px[l + 1] - px[l], dy = py[l + 1] - py[l], dz = pz[l + 1] - pz[l];
      g = (double)(dx * dx + dy * dy + dz * dz) > 0.05;
    }
    const u64 m = __ballot(g);
    if (lane == 0) gapw[base >> 6] = m;
  }
  __syncthreads();
  for (int l = 5 + tid; l < len - 5; l += kRingThreads) {  // picks are >= 5 away from both ring ends
    const int c = l >> 6, b = l & 63;
    const u64 w0 = gapw[c], wn = gapw[c + 1], wp = c > 0 ? gapw[c - 1] : 0ull;
    // forward: bits l .. l+4 (SR:353-364 stops at the first gap)
    const unsigned fw = (unsigned)(((w0 >> b) | ((wn << 1) << (63 - b))) & 31ull);
    const int f = min(5, (int)__builtin_ctz(fw | 32u));
    // backward: bits l-1 .. l-5, bit l-1 on top (SR:365-376)
    const unsigned bw = b >= 5 ? (unsigned)((w0 >> (b - 5)) & 31ull) : (unsigned)(((wp >> (59 + b)) | (w0 << (5 - b))) & 31ull);
    const int k = min(5, (int)__builtin_clz((bw << 27) | (1u << 26)));
    reachb[l] = (unsigned char)(k | (f << 4));
  }

  auto curvature = [&](int i) {  // SR:288-303
    const float dX = px[i - 5] + px[i - 4] + px[i - 3] + px[i - 2] + px[i - 1] - 10 * px[i] + px[i + 1] + px[i + 2] + px[i + 3] + px[i + 4] + px[i + 5];
    const float dY = py[i - 5] + py[i - 4] + py[i - 3] + py[i - 2] + py[i - 1] - 10 * py[i] + py[i + 1] + py[i + 2] + py[i + 3] + py[i + 4] + py[i + 5];
    const float dZ = pz[i - 5] + pz[i - 4] + pz[i - 3] + pz[i - 2] + pz[i - 1] - 10 * pz[i] + pz[i + 1] + pz[i + 2] + pz[i + 3] + pz[i + 4] + pz[i + 5];
    return dX * dX + dY * dY + dZ * dZ;
  };

  // ---- debug only: the reference's std::sort of every sector (SR:323, canonical tie order: index ascending), as a
  // wavefront-local bitonic network in LDS.  The production path below never sorts.
  if (dbg_sort && wave < kSectors) {
    const int sp = s_sp[wave] - off, ep = s_ep[wave] - off;  // local indices
    const int seclen = ep - sp + 1;
    int P = 2;
    while (P < seclen) P <<= 1;
    u64* K = keys + wave * SECT;
    for (int t = lane; t < P; t += 64) {
      u64 key = ~0ull;
      if (t < seclen) key = ((u64)__float_as_uint(curvature(sp + t)) << 32) | (unsigned)(sp + t);  // c >= 0: bits order like the value
      K[t] = key;
    }
    lds_fence_wave();
    for (int k = 2; k <= P; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        bitonic_stage(K, P, j, k, lane, 64);
        lds_fence_wave();
      }
    for (int t = lane; t < seclen; t += 64) dbg_sort[off + sp + t] = off + (int)(K[t] & 0xffffffffu);
  }
  __syncthreads();
  SR_STAMP();

  // ---- greedy picks (SR:325-422).  The reference sorts each sector by curvature and walks the sorted list from the top (2
  // sharp + 18 less-sharp picks) and from the bottom (4 flat picks), skipping points suppressed by earlier picks.  Walking a
  // sorted list and taking the first eligible entry is the same as taking the arg-max (arg-min) over the eligible entries,
  // and at most 24 entries are ever taken — so nothing is sorted: every lane keeps the curvatures of its sector points in
  // registers (point t of the sector lives in lane t % 64, slot t / 64) with one eligibility bit each, and a pick is one
  // wavefront arg-max (two DPP reductions: curvature bits, then index among the ties — descending (c, index) order for the
  // sharp walk, ascending for the flat walk, exactly the sorted list's order).
  //
  // The sectors of a ring are walked in order by the reference and a pick's neighbour suppression can spill over the sector
  // boundary (SR:353-376), so sector s + 1 formally depends on sector s.  The spill reaches at most 5 points and only
  // matters if one of them would otherwise be selected, so all six sectors run at once (one wavefront each) without incoming
  // marks; afterwards the boundaries are checked in order and a sector is redone — with the marks of its predecessor applied
  // first — only when a spilled-on point had been selected.  Marks are clipped to the own sector while picking (a later
  // sector must not disturb an earlier one); the full extents are applied at the end so that `picked` ends up exactly like
  // cloudNeighborPicked.
  // A redone sector does not start over: its picks are taken in priority order, so every pick made BEFORE the first one the new spill
  // invalidates stands — the picks of both walks are kept in LDS, the kept ones are replayed (marks only, all at once, one lane each)
  // and the walk resumes behind them.  On average half of the second pass; a spill that hits a flat pick keeps the whole sharp walk.
  unsigned* cbits = (unsigned*)keys;      // [CAP] curvature bits of the sector points (a redo reads them back instead of 33 LDS reads per point)
  constexpr int kPickSlots = kMaxLessSharpPerSect + kMaxFlatPerSect;
  int* s_plist = iscratch + 64;                         // [kSectors][kPickSlots] local index of the walks' picks, in pick order
  int* s_pcnt = s_plist + kSectors * kPickSlots;        // [kSectors][2] picks of the sharp / flat walk
  auto run_sector_q = [&](auto kq_tag, int s, int in_hi, bool redo, int keep_sharp, bool sharp_final, int keep_flat) {
    constexpr int kQ = decltype(kq_tag)::value;   // register slots per lane: sector length <= 64 kQ
    // (the same in every lane: in scalar registers, so that what derives from them — the lane masks of suppress() — is scalar too)
    const int sp_l = __builtin_amdgcn_readfirstlane(s_sp[s] - off), ep_l = __builtin_amdgcn_readfirstlane(s_ep[s] - off);
    const int seclen = ep_l - sp_l + 1;
    int* plist = s_plist + s * kPickSlots;
    if (redo) {  // forget the previous result, then apply the predecessor's spill
      for (int l = sp_l + lane; l <= ep_l; l += 64) { picked[l] = (l <= in_hi) ? 1 : 0; label[l] = 0; }
      lds_fence_wave();
    }
    // A pick is a chain of dependent instructions on a wavefront that is (almost) alone on its SIMD — ~8 cycles each —, so the walk is
    // written for few instructions per pick: the lane's candidates are kept PRE-MASKED (an ineligible slot holds the walk's neutral
    // value: the lane's best is a max3 / min3 tree), the winner's local index travels with its suppression reach in one word (no LDS
    // round trip for the reach), the lane that owns the winner is found by ballot (a second reduction only when two lanes tie), and the
    // picks stay in registers (lane k: pick k) until the walk is over — labels, index lists and the pick list are written once, in parallel.
    unsigned cb[kQ];                      // curvature bits of point sp_l + q * 64 + lane
    unsigned pay[kQ];                     // its local index << 8 | reach byte
    unsigned sharp_bits = 0, flat_bits = 0, elig = 0, inside = 0;
#pragma unroll
    for (int q = 0; q < kQ; q++) {
      cb[q] = 0; pay[q] = 0;
      if (q * 64 < seclen) {
        const int t = q * 64 + lane;
        if (t < seclen) {
          const int i = sp_l + t;
          inside |= 1u << q;
          if (redo) {
            cb[q] = cbits[i];
          } else {
            const float c0 = curvature(i);
            if (dbg_curv) dbg_curv[off + i] = c0;
            cb[q] = __float_as_uint(c0);   // c >= 0: the bit pattern orders like the value
            cbits[i] = cb[q];
          }
          pay[q] = ((unsigned)i << 8) | (unsigned)reachb[i];
          // SR:330 / SR:383 compare the f32 curvature with the double 0.1, which lies between two neighbouring floats (0.099999994 and
          // 0.1f = 0.100000001): (double)c > 0.1 <=> c >= 0.1f and (double)c < 0.1 <=> c < 0.1f — no f64 convert and compare per point
          const float c = __uint_as_float(cb[q]);
          if (c >= 0.1f) sharp_bits |= 1u << q;
          if (c < 0.1f) flat_bits |= 1u << q;
          if (i > in_hi) elig |= 1u << q;
        }
      }
    }
    int* o_sharp = sharp_idx + (r * kSectors + s) * kMaxSharpPerSect;
    int* o_less = less_sharp_idx + (r * kSectors + s) * kMaxLessSharpPerSect;
    int* o_flat = flat_idx + (r * kSectors + s) * kMaxFlatPerSect;
    int n_less = 0, n_flat = 0;
    unsigned my_sharp = 0, my_flat = 0;   // lane k: pick k of the sharp / flat walk (local index << 8 | reach)
    int leak_lo = INT_MAX, leak_hi = -1;
    unsigned v[kQ];                       // the running walk's pre-masked candidates
    // SR:353-376 around the winner, as far as the walk itself needs it: the (at most 11) covered points of the sector leave the
    // candidates.  The marks themselves (cloudNeighborPicked) are written after the walk, by one lane per pick.
    auto suppress = [&](unsigned pw, unsigned neutral) {
      const int lf = (int)(pw >> 8);
      const int lo_m = lf - (int)(pw & 15u), hi_m = lf + (int)((pw >> 4) & 15u);
      const int tlo = max(lo_m - sp_l, 0), thi = min(hi_m - sp_l, seclen - 1);
      const unsigned d = (unsigned)(lane - tlo), span = (unsigned)(thi - tlo);
#pragma unroll
      for (int q = 0; q < kQ; q++)
        if (d + (unsigned)(q * 64) <= span) v[q] = neutral;
    };
    // the marks (clipped to the sector) and extents of picks [from, to) of a walk: lane k marks around pick k
    auto apply_marks = [&](unsigned my, int from, int to) {
      int lo_m = INT_MAX, hi_m = -1;
      if (lane >= from && lane < to) {
        const int lf = (int)(my >> 8);
        lo_m = lf - (int)(my & 15u); hi_m = lf + (int)((my >> 4) & 15u);
        for (int l = max(lo_m, sp_l); l <= min(hi_m, ep_l); l++) picked[l] = 1;
      }
      leak_lo = min(leak_lo, (int)wave_min_u32((unsigned)lo_m));
      leak_hi = max(leak_hi, (int)wave_max_u32((unsigned)(hi_m + 1)) - 1);
    };
    // which of the lane's points no pick has marked (the walks themselves only keep v[] current)
    auto refresh_elig = [&]() {
      lds_fence_wave();
#pragma unroll
      for (int q = 0; q < kQ; q++)
        if (((inside >> q) & 1u) && picked[sp_l + q * 64 + lane]) elig &= ~(1u << q);
    };
    // kept picks [first, first + n) of the pick list come back into the registers (lane k: pick k); the first n_marking of them mark
    auto replay = [&](int first, int n, int n_marking, int sharp_walk) {
      unsigned my = 0;
      if (lane < n) { const int lf = plist[first + lane]; my = ((unsigned)lf << 8) | (unsigned)reachb[lf]; }
      if (sharp_walk) my_sharp = my; else my_flat = my;
      apply_marks(my, 0, n_marking);
    };
    // SR:327-378, descending curvature
    if (keep_sharp > 0) { replay(0, keep_sharp, keep_sharp, 1); n_less = keep_sharp; }
    if (!sharp_final) {
      if (redo) refresh_elig();
#pragma unroll
      for (int q = 0; q < kQ; q++) v[q] = ((elig & sharp_bits) >> q) & 1u ? cb[q] : 0u;   // candidates have c > 0.1, i.e. non-zero bits
      for (int picks = keep_sharp + 1; picks <= kMaxLessSharpPerSect; picks++) {
        unsigned lb = v[0];
#pragma unroll
        for (int q = 1; q < kQ; q++) lb = max(lb, v[q]);
        const unsigned mh = wave_max_u32(lb);
        if (mh == 0) break;
        unsigned pl = 0;
#pragma unroll
        for (int q = 0; q < kQ; q++) if (v[q] == mh) pl = pay[q];   // ascending: the higher index wins ties
        const u64 tie = __ballot(lb == mh);
        const unsigned pw = (tie & (tie - 1ull)) == 0ull ? (unsigned)__builtin_amdgcn_readlane((int)pl, __ffsll((long long)tie) - 1) : wave_max_u32(pl);
        if (lane == n_less) my_sharp = pw;
        n_less++;
        suppress(pw, 0u);
      }
      apply_marks(my_sharp, keep_sharp, n_less);
    }
    // SR:380-422, ascending curvature
    if (keep_flat > 0) { replay(kMaxLessSharpPerSect, keep_flat, min(keep_flat, kMaxFlatPerSect - 1), 0); n_flat = keep_flat; }
    refresh_elig();
#pragma unroll
    for (int q = 0; q < kQ; q++) v[q] = ((elig & flat_bits) >> q) & 1u ? cb[q] : 0xffffffffu;   // (c < 0.1: never all ones)
    for (int picks = keep_flat + 1;; picks++) {
      unsigned lb = v[0];
#pragma unroll
      for (int q = 1; q < kQ; q++) lb = min(lb, v[q]);
      const unsigned mh = wave_min_u32(lb);
      if (mh == 0xffffffffu) break;
      unsigned pl = 0xffffffffu;
#pragma unroll
      for (int q = kQ - 1; q >= 0; q--) if (v[q] == mh) pl = pay[q];   // descending: the lower index wins ties
      const u64 tie = __ballot(lb == mh);
      const unsigned pw = (tie & (tie - 1ull)) == 0ull ? (unsigned)__builtin_amdgcn_readlane((int)pl, __ffsll((long long)tie) - 1) : wave_min_u32(pl);
      if (lane == n_flat) my_flat = pw;
      n_flat++;
      if (picks >= kMaxFlatPerSect) break;  // the 4th flat point is emitted but not suppressed (SR:390-394)
      suppress(pw, 0xffffffffu);
    }
    apply_marks(my_flat, keep_flat, min(n_flat, kMaxFlatPerSect - 1));
    // labels, index lists, pick list: one lane per pick
    const int n_sharp = min(n_less, kMaxSharpPerSect);
    if (lane < n_less) {
      const int lf = (int)(my_sharp >> 8);
      label[lf] = lane < kMaxSharpPerSect ? 2 : 1;
      o_less[lane] = off + lf; plist[lane] = lf;
      if (lane < n_sharp) o_sharp[lane] = off + lf;
    }
    if (lane < n_flat) { const int lf = (int)(my_flat >> 8); label[lf] = -1; o_flat[lane] = off + lf; plist[kMaxLessSharpPerSect + lane] = lf; }
    lds_fence_wave();
    if (lane == 0) {
      S->sect_cnt[r][s][0] = n_sharp; S->sect_cnt[r][s][1] = n_less; S->sect_cnt[r][s][2] = n_flat;
      s_leak_lo[s] = leak_lo; s_leak_hi[s] = leak_hi;
      s_zone[s] = in_hi;
      s_pcnt[2 * s] = n_less; s_pcnt[2 * s + 1] = n_flat;
    }
  };
  auto run_sector = [&](int s, int in_hi, bool redo, int keep_sharp, bool sharp_final, int keep_flat) {
    if (s_ep[s] - s_sp[s] + 1 <= 6 * 64) run_sector_q(std::integral_constant<int, 6>{}, s, in_hi, redo, keep_sharp, sharp_final, keep_flat);   // HDL-64E: ~330 points per sector
    else run_sector_q(std::integral_constant<int, SECT / 64>{}, s, in_hi, redo, keep_sharp, sharp_final, keep_flat);
  };
  if (wave < kSectors) run_sector(wave, -1, false, 0, false, 0);
  __syncthreads();
  SR_STAMP();
  // fixed point over the boundaries: a sector is redone when the spill it was computed with differs from its predecessors'
  // current spill in a way that can matter.  Sector 0 never changes, so after round k sectors 0..k are final.
  for (int round = 0; round < kSectors - 1; round++) {
    if (tid == 0) *s_any = 0;
    __syncthreads();
    bool redo = false, sharp_final = false;
    int in_hi = -1, keep_sharp = 0, keep_flat = 0;
    if (wave >= 1 && wave < kSectors) {
      const int sp_l = s_sp[wave] - off, ep_l = s_ep[wave] - off;
      for (int q = 0; q < wave; q++) in_hi = max(in_hi, s_leak_hi[q]);  // a spill reaches 5 points: several sectors when they are tiny
      in_hi = min(in_hi, ep_l);
      if (in_hi < sp_l) in_hi = -1;
      const int used = s_zone[wave];
      if (in_hi > used) {        // the spill grew: matters only if a newly covered point (at most 5) had been selected
        const int nl = s_pcnt[2 * wave], nf = s_pcnt[2 * wave + 1];
        int lf = -1;
        if (lane < kMaxLessSharpPerSect) { if (lane < nl) lf = s_plist[wave * kPickSlots + lane]; }
        else if (lane < kPickSlots) { if (lane - kMaxLessSharpPerSect < nf) lf = s_plist[wave * kPickSlots + lane]; }
        const unsigned long long hit = __ballot(lf > used && lf <= in_hi);   // (picks lie inside the sector, i.e. at or above sp_l)
        redo = hit != 0ull;
        if (redo) {   // everything picked before the first invalidated pick stands
          const int k = __ffsll((long long)hit) - 1;
          if (k < kMaxLessSharpPerSect) { keep_sharp = k; }
          else { keep_sharp = nl; sharp_final = true; keep_flat = k - kMaxLessSharpPerSect; }
        }
        if (!redo && lane == 0) s_zone[wave] = in_hi;
      } else if (in_hi < used) {  // the spill shrank: points that were blocked are free again — from the start
        redo = true;
      }
      if (redo && lane == 0) *s_any = 1;
    }
    __syncthreads();  // every wavefront has read its predecessor's spill before anything is redone
    if (*s_any == 0) break;
    if (redo) run_sector(wave, in_hi, true, keep_sharp, sharp_final, keep_flat);
    __syncthreads();
  }
  if (wave == 0) {
    // full mark extents (beyond the own sector), as the reference leaves them behind
    for (int s = 0; s < kSectors; s++) {
      const int sp_l = s_sp[s] - off, ep_l = s_ep[s] - off;
      const int lo_m = s_leak_lo[s], hi_m = s_leak_hi[s];
      if (hi_m < 0) continue;
      if (lane < 8) {
        const int lb = lo_m + lane, lf = ep_l + 1 + lane;
        if (lb < sp_l) picked[lb] = 1;
        if (lf <= hi_m) picked[lf] = 1;
      }
    }
  }
  __syncthreads();
  SR_STAMP();
  if (dbg_picked) for (int l = tid; l < len; l += kRingThreads) { dbg_picked[off + l] = picked[l]; dbg_label[off + l] = label[l]; }

  // ---- lessFlat (SR:424-430) + per-ring pcl::VoxelGrid leaf 0.2 (SR:433-437)
  const int c_lo = 5, c_hi = len - 7;  // local range covered by the six sectors: [start, end-1]

  // bounding box of the candidates (getMinMax3D)
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  int mycnt = 0;
  for (int l = c_lo + tid; l <= c_hi; l += kRingThreads)
    if (label[l] <= 0) {
      mn[0] = fminf(mn[0], px[l]); mx[0] = fmaxf(mx[0], px[l]);
      mn[1] = fminf(mn[1], py[l]); mx[1] = fmaxf(mx[1], py[l]);
      mn[2] = fminf(mn[2], pz[l]); mx[2] = fmaxf(mx[2], pz[l]);
      mycnt++;
    }
  for (int a = 0; a < 3; a++) { mn[a] = wave_fminmax<false>(mn[a]); mx[a] = wave_fminmax<true>(mx[a]); }   // (DPP row steps: no LDS crossbar)
  mycnt = wave_sum_i32(mycnt);
  float* wred = (float*)scan_tmp;  // [8 waves][6] + counts
  if (lane == 0) {
    for (int a = 0; a < 3; a++) { wred[wave * 8 + a] = mn[a]; wred[wave * 8 + 3 + a] = mx[a]; }
    ((int*)wred)[wave * 8 + 6] = mycnt;
  }
  __syncthreads();
  // (every lane folds the eight wavefronts' results itself — broadcast reads — instead of waiting for one lane behind a second barrier)
  int ncand = 0;
  for (int a = 0; a < 3; a++) { mn[a] = wred[a]; mx[a] = wred[3 + a]; }
  for (int w = 0; w < kRingThreads / 64; w++) {
    for (int a = 0; a < 3; a++) { mn[a] = fminf(mn[a], wred[w * 8 + a]); mx[a] = fmaxf(mx[a], wred[w * 8 + 3 + a]); }
    ncand += ((int*)wred)[w * 8 + 6];
  }
  if (ncand == 0) return;
  const float inv = 1.0f / 0.2f;  // inverse_leaf_size_
  const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
  float4* out = ring_ds + (size_t)r * kMaxRingLen;
  if (dx * dy * dz > (long long)INT_MAX) {
    // PCL: "Leaf size is too small for the input dataset" -> output = input.  Keep input order.
    for (int l = tid; l < len; l += kRingThreads) iscratch[l] = (l >= c_lo && l <= c_hi && label[l] <= 0) ? 1 : 0;
    __syncthreads();
    block_exclusive_scan(iscratch, len, scan_tmp);
    for (int l = c_lo + tid; l <= c_hi; l += kRingThreads)
      if (label[l] <= 0) out[iscratch[l]] = make_float4(px[l], py[l], pz[l], pi[l]);
    if (tid == 0) S->ring_ds_cnt[r] = ncand;
    return;
  }
  int min_b[3], div_b[3];
  for (int a = 0; a < 3; a++) {
    min_b[a] = (int)floorf(mn[a] * inv);
    div_b[a] = (int)floorf(mx[a] * inv) - min_b[a] + 1;
  }
  // Consecutive ring points mostly fall into the same 0.2 m voxel, so the sort runs over RUNS (maximal stretches of
  // consecutive candidates sharing a voxel), not points: key = (voxel index, first point of the run).  Sorted runs of one
  // voxel are in input order and so are the points inside a run, hence summing run after run reproduces the
  // input-order f32 sums of pcl::VoxelGrid exactly, with a network several times smaller.
  // run key = voxel index : first point : last point (32 : 20 : 12 bits; a run is known by its first point, so the order is (voxel,
  // first point)).  The run's head writes the voxel and its own position, the point behind its last point ORs in where the run ended:
  // the centroid pass below then walks [first, last] without testing every point's voxel (a dependent LDS trip per point otherwise).
  //
  // Every wavefront takes a CONTIGUOUS stretch of the ring, 64 consecutive points per trip: a point's predecessor sits in the lane
  // below, the run heads of a trip are one ballot word and a head's run number is a population count — no voxel array, no flag array,
  // no workgroup scan (that was five passes over LDS behind five barriers; this is two passes behind two).
  u64* K2 = keys;                      // [<= CAP] run keys
  unsigned* K2w = (unsigned*)K2;       // [2 t] low word, [2 t + 1] high word
  constexpr int kTrips = (CAP + kRingThreads - 1) / kRingThreads;
  const int seg = ((len + kRingThreads - 1) / kRingThreads) * 64;   // points per wavefront (<= 64 kTrips)
  const int seg0 = wave * seg;
  auto voxel_of = [&](int l) -> int {  // -1: not a lessFlat candidate
    if (l < c_lo || l > c_hi || label[l] > 0) return -1;
    const int ijk0 = (int)(floorf(px[l] * inv) - (float)min_b[0]);
    const int ijk1 = (int)(floorf(py[l] * inv) - (float)min_b[1]);
    const int ijk2 = (int)(floorf(pz[l] * inv) - (float)min_b[2]);
    return ijk0 + ijk1 * div_b[0] + ijk2 * div_b[0] * div_b[1];
  };
  int vidx[kTrips];
  u64 headm[kTrips], pcandm[kTrips];   // trip's run heads / lanes whose predecessor is a candidate
  int nheads = 0;
  {
    int carry = seg0 > 0 && seg0 - 1 < len ? voxel_of(seg0 - 1) : -2;   // the point in front of the stretch (-2: none)
    carry = __builtin_amdgcn_readfirstlane(carry);
#pragma unroll
    for (int it = 0; it < kTrips; it++) {
      vidx[it] = -1; headm[it] = 0ull; pcandm[it] = 0ull;
      if (it * 64 < seg) {
        const int l = seg0 + it * 64 + lane;
        const int v = l < len ? voxel_of(l) : -1;
        int pv = __shfl_up(v, 1);
        if (lane == 0) pv = carry;
        carry = __builtin_amdgcn_readlane(v, 63);
        vidx[it] = v;
        headm[it] = __ballot(v >= 0 && pv != v);
        pcandm[it] = __ballot(pv >= 0);
        nheads += __popcll(headm[it]);
        if (l < CAP) K2[l] = 0ull;   // (run numbers never exceed point numbers: every key that will be ORed together starts from zero)
      }
    }
  }
  int* s_heads = scan_tmp + 64;        // [8] (behind the bounding-box slots: a slow wavefront may still be reading those)
  if (lane == 0) s_heads[wave] = nheads;
  __syncthreads();
  int nrun = 0, running = 0;
  for (int w = 0; w < kRingThreads / 64; w++) { const int t = s_heads[w]; running += w < wave ? t : 0; nrun += t; }
  int P2 = 2;
  while (P2 < nrun) P2 <<= 1;
  for (int t = nrun + tid; t < P2 && t < CAP; t += kRingThreads) K2[t] = ~0ull;   // the sorting network's padding
#pragma unroll
  for (int it = 0; it < kTrips; it++) {
    if (it * 64 < seg) {
      const int l = seg0 + it * 64 + lane;
      const int v = vidx[it];
      const u64 hm = headm[it];
      const bool head = (hm >> lane) & 1ull;
      const int run = running + __popcll(hm & ((1ull << lane) - 1ull));   // heads in front of l
      if (head) { K2w[2 * run + 1] = (unsigned)v; atomicOr(&K2w[2 * run], (unsigned)l << 12); }
      if (((pcandm[it] >> lane) & 1ull) && (head || v < 0)) atomicOr(&K2w[2 * (run - 1)], (unsigned)(l - 1));   // l - 1 closed its run
      running += __popcll(hm);
    }
  }
  __syncthreads();
  SR_STAMP();
  // a stage with stride j <= 64 only moves data inside 128-element blocks that belong to one wavefront (64 consecutive
  // compare-exchanges), so only the wide strides need a workgroup barrier
  if (P2 > CAP) {
    // More runs than the largest power of two the key array holds (only the small tier's 2 176-point rings can get here, with a
    // voxel of its own for almost every point): rank by counting instead of padding the network to 4 096 keys.  Keys are unique.
    constexpr int NQ = (CAP + kRingThreads - 1) / kRingThreads;
    u64 mine[NQ];
    int rk[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int t = tid + q * kRingThreads;
      mine[q] = t < nrun ? K2[t] : ~0ull;
      int below = 0;
      if (t < nrun) for (int e = 0; e < nrun; e++) below += K2[e] < mine[q];
      rk[q] = below;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; q++) if (tid + q * kRingThreads < nrun) K2[rk[q]] = mine[q];
  } else if (P2 < 128) {
    for (int k = 2; k <= P2; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        bitonic_stage(K2, P2, j, k, tid, kRingThreads);
        const int next_j = j > 1 ? j >> 1 : k;  // first stride of the next merge level
        if (j > 64 || next_j > 64) __syncthreads(); else lds_fence_wave();
      }
  } else if (P2 < 256 * (kRingThreads / 64)) {
    // up to 1 024 keys: 128-key blocks (two keys per lane) keep all eight wavefronts busy; strides <= 64 run in registers, the strides >= 128
    // go through LDS behind a workgroup barrier
    bitonic_reg_stages(K2, P2, 2, 128, tid, kRingThreads);
    __syncthreads();
    for (int k = 256; k <= P2; k <<= 1) {
      for (int j = k >> 1; j >= 128; j >>= 1) {
        bitonic_stage(K2, P2, j, k, tid, kRingThreads);
        __syncthreads();
      }
      bitonic_reg_stages(K2, P2, k, k, tid, kRingThreads);
      __syncthreads();
    }
  } else {
    // 2 048 keys and more: 256-key blocks (four keys per lane, one block per wavefront at 2 048); strides <= 128 run in registers, only the
    // strides >= 256 — 6 of the 66 stages at 2 048 run keys — go through LDS (41.5 k -> 37 k cycles; at 1 024 keys the wider blocks would
    // leave four wavefronts idle: 19 k -> 26 k)
    bitonic_reg_stages4(K2, P2, 2, 256, tid, kRingThreads);
    __syncthreads();
    for (int k = 512; k <= P2; k <<= 1) {
      for (int j = k >> 1; j >= 256; j >>= 1) {
        bitonic_stage(K2, P2, j, k, tid, kRingThreads);
        __syncthreads();
      }
      bitonic_reg_stages4(K2, P2, k, k, tid, kRingThreads);
      __syncthreads();
    }
  }
  __syncthreads();
  SR_STAMP();
  // voxel heads -> output rank, the same way (contiguous stretches of the sorted keys, ballot words, population counts), then the centroids
  int nvox = 0;
  {
    const int qseg = ((nrun + kRingThreads - 1) / kRingThreads) * 64, q0 = wave * qseg;
    u64 kreg[kTrips], hm[kTrips];
    unsigned carry = q0 > 0 && q0 - 1 < nrun ? (unsigned)(K2[q0 - 1] >> 32) : 0xffffffffu;   // (no voxel has that index: idx <= INT_MAX)
    int mine = 0;
#pragma unroll
    for (int it = 0; it < kTrips; it++) {
      kreg[it] = 0ull; hm[it] = 0ull;
      if (it * 64 < qseg) {
        const int t = q0 + it * 64 + lane;
        const u64 k0 = t < nrun ? K2[t] : ~0ull;
        const unsigned vid = (unsigned)(k0 >> 32);
        unsigned pv = (unsigned)__shfl_up((int)vid, 1);
        if (lane == 0) pv = carry;
        carry = (unsigned)__builtin_amdgcn_readlane((int)vid, 63);
        kreg[it] = k0;
        hm[it] = __ballot(t < nrun && vid != pv);
        mine += __popcll(hm[it]);
      }
    }
    int* s_vox = scan_tmp + 72;   // [8]
    if (lane == 0) s_vox[wave] = mine;
    __syncthreads();
    int running = 0;
    for (int w = 0; w < kRingThreads / 64; w++) { const int c = s_vox[w]; running += w < wave ? c : 0; nvox += c; }
#pragma unroll
    for (int it = 0; it < kTrips; it++) {
      if (it * 64 < qseg) {
        const int t = q0 + it * 64 + lane;
        if ((hm[it] >> lane) & 1ull) {
          const unsigned vid = (unsigned)(kreg[it] >> 32);
          float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;  // CentroidPoint<PointXYZI>: f32 sums in input order
          int npt = 0;
          u64 k = kreg[it];
          for (int u = t;;) {
            const int l0 = (int)((k >> 12) & 0xfffu), l1 = (int)(k & 0xfffu);
            for (int l = l0; l <= l1; l++) { sx += px[l]; sy += py[l]; sz += pz[l]; si += pi[l]; }
            npt += l1 - l0 + 1;
            if (++u >= nrun) break;
            k = K2[u];
            if ((unsigned)(k >> 32) != vid) break;
          }
          const float cnt = (float)npt;
          out[running + __popcll(hm[it] & ((1ull << lane) - 1ull))] = make_float4(sx / cnt, sy / cnt, sz / cnt, si / cnt);
        }
        running += __popcll(hm[it]);
      }
    }
  }
  if (tid == 0) S->ring_ds_cnt[r] = nvox;
  SR_STAMP();
  if (dbg_cyc && tid == 0) {
    for (int q = 0; q < 7; q++) dbg_cyc[r * 8 + q] = q + 1 < nstamp ? tstamp[q + 1] - tstamp[q] : 0;
    dbg_cyc[r * 8 + 7] = (long long)nrun | ((long long)nvox << 16) | ((long long)len << 32) | ((long long)ncand << 48);
  }
#undef SR_STAMP
  };
  // one workgroup per ring — or the catch-all: which rings are oversized (one load per lane, the same answer in every wavefront; normally
  // none, and the launch ends here).  ONE call site: a second copy of the ring body costs the small tier 50 VGPRs, i.e. its second
  // workgroup per CU.
  static_assert(kMaxRings == 64, "one ring per lane");
  if constexpr (!BIG_TIER) {
    one_ring((int)blockIdx.x);   // (straight-line: a loop around the body costs this tier 50 VGPRs, i.e. its second workgroup per CU)
  } else {
    unsigned long long todo = 1ull << (blockIdx.x & 63);
    if (gridDim.x != kMaxRings) todo = __ballot(S->ring_count[lane] > kRingCapSmall);
    while (todo != 0ull) {
      const int r = __ffsll((long long)todo) - 1;
      todo &= todo - 1ull;
      one_ring(r);
      if (todo != 0ull) __syncthreads();   // the ring's LDS is no longer read
    }
  }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sr_compact(const float4* __restrict__ cloud, FrameScalars* S, const int* __restrict__ sharp_idx,
                                                    const int* __restrict__ less_sharp_idx, const int* __restrict__ flat_idx,
                                                    const float4* __restrict__ ring_ds, float4* __restrict__ sharp,
                                                    float4* __restrict__ less_sharp, float4* __restrict__ flat,
                                                    float4* __restrict__ less_flat, int* __restrict__ dbg_feat_idx /* [3][kMaxLessSharp] */,
                                                    int* sticky_err, size_t ss) {
  VL_SESSION(ss); RB(cloud); RB(S); RB(sharp_idx); RB(less_sharp_idx); RB(flat_idx); RB(ring_ds); RB(sharp); RB(less_sharp); RB(flat); RB(less_flat);
  RB(dbg_feat_idx); RB(sticky_err);
  __shared__ int base[4];
  __shared__ int soff[kSectors][3];
  __shared__ float s_box[4][12];
  const int r = blockIdx.x, tid = threadIdx.x;
  float bmn[2][3], bmx[2][3];   // this thread's share of the line's lessSharp [0] / lessFlat [1] bounding boxes
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int a = 0; a < 3; a++) { bmn[c][a] = FLT_MAX; bmx[c][a] = -FLT_MAX; }
  // last launch of the sweep's scan registration: its error bits (S->error is per buffer set and rewritten every sweep) go into the
  // handle's sticky word, so that a burst of vloam_process_scan calls cannot lose them
  if (r == 0 && tid == 0 && sticky_err && S->error) atomicOr(sticky_err, S->error);
  {
    // 4 wavefronts: kinds 0..2 = sharp / lessSharp / flat picks per ring, kind 3 = per-ring VoxelGrid output size
    const int kind = tid >> 6, q = tid & 63;
    int v = 0;
    if (kind < 3) { for (int s = 0; s < kSectors; s++) v += S->sect_cnt[q][s][kind]; }
    else v = S->ring_ds_cnt[q];
    int inc = v;
    for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(inc, d); if (q >= d) inc += t; }
    if (q == r) base[kind] = inc - v;
    if (q == 63 && r == kMaxRings - 1) {
      if (kind == 0) S->n_sharp = inc; else if (kind == 1) S->n_less_sharp = inc; else if (kind == 2) S->n_flat = inc; else S->n_less_flat = inc;
    }
  }
  __syncthreads();
  if (tid < 3) {
    int run = base[tid];
    for (int s = 0; s < kSectors; s++) { soff[s][tid] = run; run += S->sect_cnt[r][s][tid]; }
  }
  __syncthreads();
  // sector lists: tiny — one thread per (sector, kind, slot)
  for (int k = tid; k < kSectors * (kMaxSharpPerSect + kMaxLessSharpPerSect + kMaxFlatPerSect); k += 256) {
    const int per = kMaxSharpPerSect + kMaxLessSharpPerSect + kMaxFlatPerSect;
    const int s = k / per, q = k % per;
    if (q < kMaxSharpPerSect) {
      if (q < S->sect_cnt[r][s][0]) {
        int src = sharp_idx[(r * kSectors + s) * kMaxSharpPerSect + q];
        sharp[soff[s][0] + q] = cloud[src];
        if (dbg_feat_idx) dbg_feat_idx[soff[s][0] + q] = src;
      }
    } else if (q < kMaxSharpPerSect + kMaxLessSharpPerSect) {
      const int qq = q - kMaxSharpPerSect;
      if (qq < S->sect_cnt[r][s][1]) {
        int src = less_sharp_idx[(r * kSectors + s) * kMaxLessSharpPerSect + qq];
        const float4 p = cloud[src];
        less_sharp[soff[s][1] + qq] = p;
        bmn[0][0] = fminf(bmn[0][0], p.x); bmn[0][1] = fminf(bmn[0][1], p.y); bmn[0][2] = fminf(bmn[0][2], p.z);
        bmx[0][0] = fmaxf(bmx[0][0], p.x); bmx[0][1] = fmaxf(bmx[0][1], p.y); bmx[0][2] = fmaxf(bmx[0][2], p.z);
        if (dbg_feat_idx) dbg_feat_idx[kMaxLessSharp + soff[s][1] + qq] = src;
      }
    } else {
      const int qq = q - kMaxSharpPerSect - kMaxLessSharpPerSect;
      if (qq < S->sect_cnt[r][s][2]) {
        int src = flat_idx[(r * kSectors + s) * kMaxFlatPerSect + qq];
        flat[soff[s][2] + qq] = cloud[src];
        if (dbg_feat_idx) dbg_feat_idx[2 * kMaxLessSharp + soff[s][2] + qq] = src;
      }
    }
  }
  const int nds = S->ring_ds_cnt[r];
  const float4* src = ring_ds + (size_t)r * kMaxRingLen;
  for (int k = tid; k < nds; k += 256) {
    const float4 p = src[k];
    less_flat[base[3] + k] = p;
    bmn[1][0] = fminf(bmn[1][0], p.x); bmn[1][1] = fminf(bmn[1][1], p.y); bmn[1][2] = fminf(bmn[1][2], p.z);
    bmx[1][0] = fmaxf(bmx[1][0], p.x); bmx[1][1] = fmaxf(bmx[1][1], p.y); bmx[1][2] = fmaxf(bmx[1][2], p.z);
  }
  // the line's two boxes: wavefront reductions, then the four wavefronts' results by twelve lanes
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float lo = wave_fminmax<false>(bmn[c][a]), hi = wave_fminmax<true>(bmx[c][a]);
      if ((tid & 63) == 0) { s_box[tid >> 6][c * 6 + a] = lo; s_box[tid >> 6][c * 6 + 3 + a] = hi; }
    }
  __syncthreads();
  if (tid < 12) {
    const bool is_max = (tid % 6) >= 3;
    float v = s_box[0][tid];
    for (int w = 1; w < 4; w++) v = is_max ? fmaxf(v, s_box[w][tid]) : fminf(v, s_box[w][tid]);
    S->less_bbox[tid / 6][r][tid % 6] = v;
  }
}

// ---- clouds handed in by the caller instead of produced by the kernels above (LaserOdometry::input / LaserMapping::input deep-copy whatever
// they are given, laser_odometry.cpp:141-145, laser_mapping.cpp:172-181): the counts and the two VoxelGrid bounding boxes of a buffer set are
// re-derived from the uploaded clouds.  One workgroup; a count < 0 keeps what the set holds.  (The boxes are only ever min / max-reduced over
// all 64 slots, so any partition of the points over the slots will do.)
__global__ __launch_bounds__(256) void k_sr_adopt(FrameScalars* S, const float4* __restrict__ less_sharp, const float4* __restrict__ less_flat, int n_full,
                                                  int n_sharp, int n_less_sharp, int n_flat, int n_less_flat) {
  __shared__ float s_box[4][64][6];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) {
    if (n_full >= 0) S->N2 = n_full;
    if (n_sharp >= 0) S->n_sharp = n_sharp;
    if (n_flat >= 0) S->n_flat = n_flat;
    if (n_less_sharp >= 0) S->n_less_sharp = n_less_sharp;
    if (n_less_flat >= 0) S->n_less_flat = n_less_flat;
  }
  for (int kind = 0; kind < 2; kind++) {
    const int n = kind ? n_less_flat : n_less_sharp;
    if (n < 0) continue;   // (uniform)
    const float4* pts = kind ? less_flat : less_sharp;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = tid; i < n; i += 256) {
      const float4 p = pts[i];
      mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
      mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
    for (int a = 0; a < 3; a++) { s_box[w][lane][a] = mn[a]; s_box[w][lane][3 + a] = mx[a]; }
    __syncthreads();
    if (tid < 64)
      for (int a = 0; a < 6; a++) {
        float v = s_box[0][tid][a];
        for (int ww = 1; ww < 4; ww++) v = a < 3 ? fminf(v, s_box[ww][tid][a]) : fmaxf(v, s_box[ww][tid][a]);
        S->less_bbox[kind][tid][a] = v;
      }
    __syncthreads();
  }
}
void sr_adopt_launch(hipStream_t st, const SRBuffers& b, int n_full, int n_sharp, int n_less_sharp, int n_flat, int n_less_flat) {
  VL_RAW_LAUNCH(k_sr_adopt, dim3(1, 1, 1), dim3(256), 0, st, b.S, b.less_sharp, b.less_flat, n_full, n_sharp, n_less_sharp, n_flat, n_less_flat);
}

// ------------------------------------------------------------------------------------------------
template <int CAP, int SECT>
static size_t sr_ring_smem_bytes() {
  return sizeof(float) * 4 * CAP + sr_ring_keys_bytes<CAP, SECT>() + sizeof(int) * (CAP + kRingThreads) + 4 * CAP + 64 * sizeof(int);
}
static_assert(kRingCapSmall % 4 == 0 && 2 * (sizeof(float) * 4 * kRingCapSmall + (size_t)12 * kRingCapSmall + sizeof(int) * (kRingCapSmall + kRingThreads) +
              4 * kRingCapSmall + 64 * sizeof(int)) <= 160 * 1024, "two small-tier rings must fit the 160 KB of LDS of one CU");

hipError_t sr_init() {
  hipError_t e = hipFuncSetAttribute((const void*)k_sr_ring<kRingCapSmall, kSectCapSmall, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)sr_ring_smem_bytes<kRingCapSmall, kSectCapSmall>());
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)k_sr_ring<kMaxRingLen, kSectCap, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)sr_ring_smem_bytes<kMaxRingLen, kSectCap>());
}

hipError_t sr_launch(hipStream_t st, const SRBuffers& b, const BatchIn& bi, Sess se, int N_SCANS, float min_range, int debug_level, ProfHook* ph, hipEvent_t done,
                     int* ring_watch, bool big_tier) {
  const bool debug = debug_level == 1, stamps = debug_level != 0;   // debug = 2: only the ring kernel's phase stamps (no reference-order debug sort)
  int n = 0;
  for (int k = 0; k < se.B; k++) n = bi.n[k] > n ? bi.n[k] : n;   // launch geometry for the largest sweep of the batch (blocks beyond a session's n idle)
  const unsigned Z = (unsigned)se.B;
  const int nblk = (n + kLabelBlock - 1) / kLabelBlock;
  const int nslice = 2 * kFLWalkers;   // (first, last) from the two ends, kFLWalkers strided walkers each
  int2* slice = (int2*)b.blockoff;         // [nslice] records of 8 B
  int* blk = b.blockoff + 2 * nslice;      // [2][nblk], behind the slice records (blockoff holds 64 ints per label workgroup: 32 + 2 nblk <= 64 nblk)
  VLOAM_LAUNCH(ph, kKSrFirstLast, st, k_sr_first_last, dim3(nslice, 1, Z), dim3(kFLThreads), 0, st, bi, min_range, slice, se.ss);
  static const int wide_env = getenv("VLOAM_SR_WIDE") ? atoi(getenv("VLOAM_SR_WIDE")) : -1;   // A/B and tests: 1 = always 1024 lanes, 0 = always 256
  if (wide_env >= 0 ? wide_env != 0 : se.B <= 2) {   // one or two sessions: one point per lane (latency); a batch: four points per lane (wave slots)
    VLOAM_LAUNCH(ph, kKSrLabel, st, k_sr_label<1024>, dim3(nblk, 1, Z), dim3(1024), 0, st, bi, min_range, N_SCANS, b.S, b.sid, b.ori, b.blockhist,
                 slice, nslice, blk, se.ss);
    VLOAM_LAUNCH(ph, kKSrScatter, st, k_sr_scatter<1024>, dim3(nblk, 1, Z), dim3(1024), 0, st, bi, b.S, b.sid, b.ori, b.blockhist, nblk, b.cloud, blk, se.ss);
  } else {
    VLOAM_LAUNCH(ph, kKSrLabel, st, k_sr_label<256>, dim3(nblk, 1, Z), dim3(256), 0, st, bi, min_range, N_SCANS, b.S, b.sid, b.ori, b.blockhist,
                 slice, nslice, blk, se.ss);
    VLOAM_LAUNCH(ph, kKSrScatter, st, k_sr_scatter<256>, dim3(nblk, 1, Z), dim3(256), 0, st, bi, b.S, b.sid, b.ori, b.blockhist, nblk, b.cloud, blk, se.ss);
  }
  // Behind the small tier ALWAYS comes the big tier: its full grid while the host has seen long rings (watch word), otherwise ONE catch-all
  // workgroup per session that looks at the ring lengths and works on the (normally zero) rings the small tier had to leave.  Any ring of up
  // to kMaxRingLen points is therefore processed on any sweep, like the reference's 400 000-point scratch (scan_registration.h:90) takes any
  // ring; the catch-all costs 13 - 16 us of single-sweep latency (a 146 KB LDS request) and 1.4 % of the B = 8 throughput on ordinary sweeps.
  // (Rounds 2 - 3 made it optional and REPORTED a ring that outgrew the small tier without the watch word's warning: a results gap on
  // reachable input.)
  // VLOAM_SR_CATCHALL=0: a host that KNOWS its rings stay below kRingCapSmall points (any 10 Hz sensor of the reference's three scan_line
  // settings) may drop the catch-all launch: a ring that outgrows the small tier without the watch word's warning is then REPORTED
  // (kErrRingTooLong -> VLOAM_ERR_CAPACITY) instead of processed; with the warning the full big tier runs as always.
  static const int catchall = getenv("VLOAM_SR_CATCHALL") ? atoi(getenv("VLOAM_SR_CATCHALL")) : 1;
  const bool big_launch = big_tier || catchall != 0;
  VLOAM_LAUNCH(ph, kKSrRing, st, (k_sr_ring<kRingCapSmall, kSectCapSmall, false>), dim3(kMaxRings, 1, Z), dim3(kRingThreads),
               (sr_ring_smem_bytes<kRingCapSmall, kSectCapSmall>()), st, b.cloud, b.S, b.sharp_idx, b.less_sharp_idx,
               b.flat_idx, b.ring_ds, debug ? b.dbg_curv : nullptr, debug ? b.dbg_sort : nullptr, debug ? b.dbg_picked : nullptr,
               debug ? b.dbg_label : nullptr, stamps ? b.dbg_cyc : nullptr, ring_watch, big_launch ? 1 : 0, se.ss);
  // the full big tier while rings near the small tier's capacity are around, else its one-workgroup catch-all
  if (big_launch)
    VLOAM_LAUNCH(ph, kKSrRingBig, st, (k_sr_ring<kMaxRingLen, kSectCap, true>), dim3(big_tier ? kMaxRings : 1, 1, Z), dim3(kRingThreads),
                 (sr_ring_smem_bytes<kMaxRingLen, kSectCap>()), st, b.cloud, b.S, b.sharp_idx, b.less_sharp_idx,
                 b.flat_idx, b.ring_ds, debug ? b.dbg_curv : nullptr, debug ? b.dbg_sort : nullptr, debug ? b.dbg_picked : nullptr,
                 debug ? b.dbg_label : nullptr, stamps ? b.dbg_cyc : nullptr, ring_watch, 1, se.ss);
  VLOAM_LAUNCH_EV(ph, kKSrCompact, st, done, k_sr_compact, dim3(kMaxRings, 1, Z), dim3(256), 0, st, b.cloud, b.S, b.sharp_idx, b.less_sharp_idx, b.flat_idx, b.ring_ds,
                     b.sharp, b.less_sharp, b.flat, b.less_flat, debug ? b.dbg_feat_idx : nullptr, b.sticky_err, se.ss);
  return hipGetLastError();
}

}  // namespace vloam
