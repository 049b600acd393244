// laserOdometry data association on gfx950: for every sharp / flat feature of the current sweep,
// TransformToStart, exact 1-NN in the previous sweep's lessSharp / lessFlat cloud, the adjacent-scan-line
// searches for the 2nd (and 3rd) point, and emission of the Ceres residual block as a FactorTable slot.
// Restates LaserOdometry::solveLO, /root/reference/src/lidar_odometry_mapping/src/laser_odometry.cpp:207-444
// ("LO:<line>").  The kd-trees of the reference become a two-level hash grid built once per sweep
// (k_lo_grid_count / scan / scatter); one wavefront per feature scans grid blocks with all 64 lanes and keeps
// (f32 distance bits << 32 | index or visit order) keys, which reproduces the reference's
// first-strictly-smaller-wins walks and "lowest index wins" kNN ties exactly.
// A batched handle splits the association in two launches: k_lo_assoc_fast (a 16-lane group per feature: the common query, settled inside the
// radius-1 block of the 1 m level or — blocks too full pruned to the cells near the query — on the 5 m level's ring groups) queues what it
// cannot settle, and k_lo_assoc_dense (this file's wave-per-feature body at a 128-register budget) takes the queue from scratch.
#include <hip/hip_runtime.h>
#include <math.h>
#include "lo_kernels.h"
#include "subwave.h"
#include <stdlib.h>

namespace vloam {

typedef unsigned long long u64;

__device__ __forceinline__ u64 wave_min_u64(u64 v) {
  for (int d = 32; d > 0; d >>= 1) {
    u64 o = __shfl_xor(v, d);
    v = o < v ? o : v;
  }
  return v;
}

// LO:149-167 with DISTORTION == false: Identity.slerp(1.0, q) is +-q, which rotates identically.
__device__ __forceinline__ float3 transform_to_start(float4 pi, const double* q, const double* t) {
  const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
  const double vx = pi.x, vy = pi.y, vz = pi.z;
  double cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;  // u x v
  cx = cx + cx; cy = cy + cy; cz = cz + cz;
  const double dx = uy * cz - uz * cy, dy = uz * cx - ux * cz, dz = ux * cy - uy * cx;  // u x (2 u x v)
  const double rx = (vx + w * cx) + dx, ry = (vy + w * cy) + dy, rz = (vz + w * cz) + dz;
  float3 o;
  o.x = (float)(rx + t[0]); o.y = (float)(ry + t[1]); o.z = (float)(rz + t[2]);
  return o;
}

__device__ __forceinline__ float sqdist(float4 c, float3 s) {
  const float d0 = c.x - s.x, d1 = c.y - s.y, d2 = c.z - s.z;
  return d0 * d0 + d1 * d1 + d2 * d2;
}

__device__ __forceinline__ unsigned grid_hash(int ix, int iy, int iz) {
  unsigned h = (unsigned)ix * 73856093u ^ (unsigned)iy * 19349663u ^ (unsigned)iz * 83492791u;
  h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
  return h;
}
// f64 quotient: two points closer than 5 m must never end up two cells apart through rounding of the division
__device__ __forceinline__ int coarse_cell(float v) { return (int)floor((double)v / 5.0); }
// The 5 m level is keyed by (cell, group of 4 scan lines): the second / third neighbour searches only want points within two
// scan lines of the closest point, i.e. at most two groups, instead of every line crossing those 15 m.
constexpr int kRingGroupShift = 2;
// The kRingGroups buckets of one cell are NEIGHBOURS in the table (bucket = cell slot * kRingGroups + group), so a cell's points of the groups
// [g0, g0 + ng) are ONE contiguous range of the bucket-ordered copy: every search of the 5 m level — all groups for the closest point, the
// one or two groups around its scan line for the second / third point — fetches 27 ranges, one per lane, whatever the group count
// (keyed by a hash of (cell, group) the closest-point search walked 27 x 16 buckets: seven per lane, staged through LDS).
constexpr int kRingGroups = kMaxRings >> kRingGroupShift;
static_assert((kRingGroups & (kRingGroups - 1)) == 0, "group count is a power of two");
__device__ __forceinline__ unsigned coarse_slot(int ix, int iy, int iz, unsigned mask) { return (grid_hash(ix, iy, iz) & (mask / kRingGroups)) * kRingGroups; }
__device__ __forceinline__ unsigned coarse_bucket(int ix, int iy, int iz, int grp, unsigned mask) { return coarse_slot(ix, iy, iz, mask) + (unsigned)grp; }
// bucket-ordered copies carry (point index, ring id) in .w: one fetch per candidate instead of index -> point
__device__ __forceinline__ unsigned pack_tag(int j, int ring) { return (unsigned)j | ((unsigned)ring << 24); }
__device__ __forceinline__ int tag_index(unsigned t) { return (int)(t & 0xffffffu); }
__device__ __forceinline__ int tag_ring(unsigned t) { return (int)(t >> 24); }

// Ring-ordered clouds put long runs of consecutive points into the same cell: one atomic per run of equal buckets inside a
// wavefront instead of one per point.  *off = position inside the run, *len = run length (valid on every lane of the run),
// *head_lane = first lane of the run.  All 64 lanes must be active.
__device__ __forceinline__ void wave_runs(unsigned b, int lane, int* head_lane, int* off, int* len) {
  const unsigned prev = __shfl_up(b, 1);
  const u64 H = __ballot(lane == 0 || prev != b);
  const int hl = 63 - __clzll((long long)(H & ((2ull << lane) - 1ull)));
  const u64 rest = hl == 63 ? 0ull : (H >> (hl + 1));
  const int next = rest ? hl + __ffsll((long long)rest) : 64;
  *head_lane = hl; *off = lane - hl; *len = next - hl;
}

__global__ __launch_bounds__(256) void k_lo_grid_count(const float4* __restrict__ less_sharp, const float4* __restrict__ less_flat,
                                                       const FrameScalars* __restrict__ S, LoGrid G, size_t ss) {
  const int kind = blockIdx.y, lane = threadIdx.x & 63;
  // pick the pointers out of the kernel-argument struct FIRST, then rebase them: a dynamically indexed, modified copy of the struct
  // would cost the compiler the global-address-space inference (FLAT instead of GLOBAL memory instructions)
  int* cnt_f = G.cnt[kind]; int* cnt_c = G.cnt[kind + 2]; int* occ_ = G.occ;
  VL_SESSION(ss); RB(less_sharp); RB(less_flat); RB(S); RB(cnt_f); RB(cnt_c); RB(occ_);
  const float4* pts = kind ? less_flat : less_sharp;
  const int n = kind ? S->n_less_flat : S->n_less_sharp;
  for (int base = blockIdx.x * 256; base < n; base += gridDim.x * 256) {
    const int i = base + threadIdx.x;
    unsigned bf = 0xffffffffu, bc = 0xffffffffu;
    int line = -1;
    if (i < n) {
      const float4 p = pts[i];
      bf = grid_hash((int)floorf(p.x), (int)floorf(p.y), (int)floorf(p.z)) & (unsigned)G.mask[kind];
      bc = coarse_bucket(coarse_cell(p.x), coarse_cell(p.y), coarse_cell(p.z), (int)p.w >> kRingGroupShift, (unsigned)G.mask[kind + 2]);
      line = (int)p.w;
    }
    int hl, off, len;
    wave_runs(bf, lane, &hl, &off, &len);
    if (off == 0 && i < n) atomicAdd(&cnt_f[bf], len);
    wave_runs(bc, lane, &hl, &off, &len);
    if (off == 0 && i < n) atomicAdd(&cnt_c[bc], len);
    // first / last index of every stored scan line (one atomic per run of equal lines)
    wave_runs((unsigned)line, lane, &hl, &off, &len);
    if (i < n && line >= 0 && line < kMaxRings) {
      if (off == 0) atomicMin(&occ_[(kind * 2 + 0) * kMaxRings + line], i);
      if (off == len - 1) atomicMax(&occ_[(kind * 2 + 1) * kMaxRings + line], i);
    }
  }
}

// Exclusive scan of the bucket counters, one workgroup per grid.  Every thread owns `per` consecutive counters (whole 16-byte
// vectors, registers only), wavefront scans + one LDS hop join them.  The counters themselves are left alone: the scatter
// pass counts them back down to zero.
// TPB = 1024 for a single sequence (shortest chain); batches use 256-lane workgroups: a 1024-lane workgroup needs a quarter of a
// CU's wave slots at once and waits for them on a busy chip (measured at B = 8: 131 us for this 15 us kernel).
template <int TPB>
__global__ __launch_bounds__(TPB) void k_lo_grid_scan(LoGrid G, size_t ss) {
  __shared__ int wsum[16];
  constexpr int kV = kGridMaxBuckets / TPB / 4;   // int4 vectors a lane may own
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* cnt_g = G.cnt[g]; int* start_g = G.start[g]; int* occ_ = G.occ; int* stops_ = G.stops;   // select, then rebase (see k_lo_grid_count)
  VL_SESSION(ss); RB(cnt_g); RB(start_g); RB(occ_); RB(stops_);
  const int nb = G.mask[g] + 1;
  const int per = nb / TPB;  // 4, 8 or 32 (kGridBuckets) at 1024 lanes
  const int4* src = (const int4*)(cnt_g + tid * per);
  int4 v[kV];
  int s = 0;
#pragma unroll
  for (int q = 0; q < kV; q++) {
    v[q] = make_int4(0, 0, 0, 0);
    if (q * 4 < per) v[q] = src[q];
    s += v[q].x + v[q].y + v[q].z + v[q].w;
  }
  int inc = s;
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int run = inc - s;
  for (int w = 0; w < wave; w++) run += wsum[w];
  int4* dst = (int4*)(start_g + tid * per);
#pragma unroll
  for (int q = 0; q < kV; q++)
    if (q * 4 < per) {
      int4 o;
      o.x = run; run += v[q].x; o.y = run; run += v[q].y; o.z = run; run += v[q].z; o.w = run; run += v[q].w;
      dst[q] = o;
    }
  if (tid == TPB - 1) start_g[nb] = run;
  // walk stops of the corner (g == 0) / surf (g == 1) cloud: stops[v] = first index with line >= v (v = 0 .. kStopLen - 1),
  // stops[kStopLen + v] = last index with line <= v - 3; then the occurrence table is re-armed for the next sweep
  if (g < 2 && wave == 0) {
    int* first = occ_ + (g * 2 + 0) * kMaxRings;
    int* last = occ_ + (g * 2 + 1) * kMaxRings;
    int f = first[lane], l = last[lane];  // kMaxRings == 64 lanes
    first[lane] = INT_MAX; last[lane] = -1;
    for (int d = 1; d < 64; d <<= 1) {  // suffix min of f, prefix max of l
      const int of = __shfl_down(f, d), ol = __shfl_up(l, d);
      if (lane + d < 64) f = min(f, of);
      if (lane >= d) l = max(l, ol);
    }
    int* stops = stops_ + g * 2 * kStopLen;
    stops[lane] = f;
    stops[kStopLen + 3 + lane] = l;
    if (lane < kStopLen - 64) stops[64 + lane] = INT_MAX;
    if (lane < 3) stops[kStopLen + lane] = -1;
  }
}

__global__ __launch_bounds__(256) void k_lo_grid_scatter(const float4* __restrict__ less_sharp, const float4* __restrict__ less_flat,
                                                         const FrameScalars* __restrict__ S, LoGrid G, size_t ss) {
  const int kind = blockIdx.y, lane = threadIdx.x & 63;
  int* cnt_f = G.cnt[kind]; int* cnt_c = G.cnt[kind + 2]; int* start_f = G.start[kind]; int* start_c = G.start[kind + 2];
  float4* pts_f = G.pts[kind]; float4* pts_c = G.pts[kind + 2];   // select, then rebase (see k_lo_grid_count)
  VL_SESSION(ss); RB(less_sharp); RB(less_flat); RB(S); RB(cnt_f); RB(cnt_c); RB(start_f); RB(start_c); RB(pts_f); RB(pts_c);
  const float4* pts = kind ? less_flat : less_sharp;
  const int n = kind ? S->n_less_flat : S->n_less_sharp;
  for (int base = blockIdx.x * 256; base < n; base += gridDim.x * 256) {
    const int i = base + threadIdx.x;
    unsigned bf = 0xffffffffu, bc = 0xffffffffu;
    float4 packed = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
      const float4 p = pts[i];
      bf = grid_hash((int)floorf(p.x), (int)floorf(p.y), (int)floorf(p.z)) & (unsigned)G.mask[kind];
      bc = coarse_bucket(coarse_cell(p.x), coarse_cell(p.y), coarse_cell(p.z), (int)p.w >> kRingGroupShift, (unsigned)G.mask[kind + 2]);
      packed = make_float4(p.x, p.y, p.z, __uint_as_float(pack_tag(i, (int)p.w)));
    }
    int hl, off, len, pos = 0;
    wave_runs(bf, lane, &hl, &off, &len);
    if (off == 0 && i < n) pos = start_f[bf] + atomicSub(&cnt_f[bf], len) - len;
    pos = __shfl(pos, hl);
    if (i < n) pts_f[pos + off] = packed;
    wave_runs(bc, lane, &hl, &off, &len);
    if (off == 0 && i < n) pos = start_c[bc] + atomicSub(&cnt_c[bc], len) - len;
    pos = __shfl(pos, hl);
    if (i < n) pts_c[pos + off] = packed;
  }
}

constexpr unsigned kBack = 0x40000000u;

// Visit every point stored in the (2R+1)^3 block of cells around (cx, cy, cz) minus the inner block of radius Rin (-1: none),
// times `ng` ring groups on the 5 m level, with the whole wavefront.  Lane l owns cells l, l + 64, ... (KC per lane): all bucket ranges
// are fetched in one trip, concatenated by a wavefront prefix sum, and the concatenated list is consumed 4 x 64 items per
// trip (each lane finds its item's owner by a 6-step search over the prefix sums), so a block costs two dependent memory
// round trips plus one per 256 points, however the points are spread over the cells.
// The visitor is a small value type (taken and returned by value, so that it stays in registers): v.visit(point).
// A block holding more than max_total points is not visited at all (*skipped = true).
// KC: cells per lane; UB: points per lane and trip (all of a trip's loads are in flight together — a cold neighbourhood costs
// a full HBM round trip per trip); R / Rin are compile-time so the cell decoding divides by constants.
template <int KC, int UB, int R, int Rin, bool GROUPS = false, class V = void>
__device__ __forceinline__ V for_each_candidate(const int* __restrict__ gstart, const float4* __restrict__ gpts, unsigned gmask, int cx,
                                                int cy, int cz, int g0, int ng, int lane, V v, int max_total, bool* skipped,
                                                int* s_inc, int* s_rel, long long* tm = nullptr) {
  if (tm) tm[0] = clock64();
  constexpr int w = 2 * R + 1, w3 = w * w * w;
  // GROUPS: the 5 m level — one range per cell covers its ring groups [g0, g0 + ng) (coarse_bucket); otherwise the 1 m level (g0 = 0, ng = 1)
  const int ncell = w3;
  int bs[KC], cnt[KC], mine = 0;
#pragma unroll
  for (int q = 0; q < KC; q++) {
    const int cc = q * 64 + lane;
    bs[q] = 0; cnt[q] = 0;
    if (cc < ncell) {
      const int ci = cc;
      const int ox = ci % w - R, oy = (ci / w) % w - R, oz = ci / (w * w) - R;
      if (!(abs(ox) <= Rin && abs(oy) <= Rin && abs(oz) <= Rin)) {  // the block of radius Rin was scanned by an earlier stage
        const unsigned b = GROUPS ? coarse_slot(cx + ox, cy + oy, cz + oz, gmask) + (unsigned)g0 : grid_hash(cx + ox, cy + oy, cz + oz) & gmask;
        bs[q] = gstart[b];
        cnt[q] = gstart[b + (GROUPS ? ng : 1)] - bs[q];
      }
    }
    mine += cnt[q];
  }
  int inc = mine;
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  const int total = __shfl(inc, 63);
  const int exc = inc - mine;
  if (tm) tm[1] = clock64();
  *skipped = total > max_total;  // the caller prefers a tighter block first (wavefront-uniform)
  if (total > max_total) return v;
  if constexpr (KC > 2) {
    static_assert(KC * 64 <= 512, "LDS slice holds 512 cells");
    int run = exc;
#pragma unroll
    for (int q = 0; q < KC; q++) {
      s_rel[lane * KC + q] = bs[q] - run;
      run += cnt[q];
      s_inc[lane * KC + q] = run;
    }
    for (int e = KC * 64 + lane; e < 512; e += 64) s_inc[e] = 0x7fffffff;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  for (int i0 = 0; i0 < total; i0 += UB * 64) {
    // owner lane of every item by a 6-step search over the inclusive sums, then the owner's cell; the UB searches advance in
    // lock step so that each step's cross-lane reads are issued back to back (every lane takes part in every exchange)
    int t[UB];
#pragma unroll
    for (int u = 0; u < UB; u++) t[u] = -1;
    if constexpr (KC > 2) {
      // many cells per lane: the cell-level prefix sums (lane-major order) were staged in this wavefront's LDS slice; every
      // item finds its cell by a 9-step binary search there (plain LDS reads instead of 21 cross-lane exchanges per item)
#pragma unroll
      for (int ug = 0; ug < UB; ug += 4)
        if (i0 + ug * 64 < total) {
          int pos[4];
#pragma unroll
          for (int u = 0; u < 4; u++) pos[u] = 0;  // number of cells whose inclusive sum is <= i
#pragma unroll
          for (int step = 256; step > 0; step >>= 1) {
            int vv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) vv[u] = s_inc[pos[u] + step - 1];
#pragma unroll
            for (int u = 0; u < 4; u++) if (vv[u] <= i0 + (ug + u) * 64 + lane) pos[u] += step;
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int i = i0 + (ug + u) * 64 + lane;
            t[ug + u] = i < total ? s_rel[pos[u] & 511] + i : -1;
          }
        }
    } else {
#pragma unroll
    for (int ug = 0; ug < UB; ug += 4)
      if (i0 + ug * 64 < total) {  // wavefront-uniform: groups past the end of the list cost nothing
        int lo[4], r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) lo[u] = 0;  // number of lanes whose inclusive sum is <= i == the lane owning item i
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
          int vv[4];
#pragma unroll
          for (int u = 0; u < 4; u++) vv[u] = __shfl(inc, lo[u] + step - 1);
#pragma unroll
          for (int u = 0; u < 4; u++) if (vv[u] <= i0 + (ug + u) * 64 + lane) lo[u] += step;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) r[u] = __shfl(exc, lo[u]);
#pragma unroll
        for (int u = 0; u < 4; u++) r[u] = i0 + (ug + u) * 64 + lane - r[u];
        int tt[4] = {-1, -1, -1, -1};
#pragma unroll
        for (int q = 0; q < KC; q++) {
          int cq[4], bq[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { cq[u] = __shfl(cnt[q], lo[u]); bq[u] = __shfl(bs[q], lo[u]); }
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (tt[u] < 0) { if (r[u] < cq[u]) tt[u] = bq[u] + r[u]; else r[u] -= cq[u]; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) t[ug + u] = i0 + (ug + u) * 64 + lane < total ? tt[u] : -1;
      }
    }
    if (tm) tm[2] = clock64();
    float4 c[UB];
#pragma unroll
    for (int u = 0; u < UB; u++) c[u] = gpts[t[u] >= 0 ? t[u] : 0];  // unconditional: all UB loads go out back to back
#pragma unroll
    for (int u = 0; u < UB; u++) if (t[u] >= 0) v.visit(c[u]);
    if (tm) { tm[3] = clock64(); tm[4] = total; }
  }
  return v;
}

// The radius-1 block of the 1 m level around (cx, cy, cz), fetched ONCE for both searches of a query: one trip for the 27 bucket
// ranges, one for up to 4 x 64 points, which then stay in registers (c[u], valid when t[u] >= 0) — the closest-point pass and
// the second / third point pass of the common query both run over them without touching memory again.  Returns false (nothing
// fetched) when the block holds more than 256 points; the caller then goes through for_each_candidate.
__device__ __forceinline__ bool fetch_block1(const int* __restrict__ gstart, const float4* __restrict__ gpts, unsigned gmask, int cx, int cy,
                                             int cz, int lane, float4 (&c)[4], int (&t)[4]) {
  int bs = 0, cnt = 0;
  if (lane < 27) {
    const int ox = lane % 3 - 1, oy = (lane / 3) % 3 - 1, oz = lane / 9 - 1;
    const unsigned b = grid_hash(cx + ox, cy + oy, cz + oz) & gmask;
    bs = gstart[b];
    cnt = gstart[b + 1] - bs;
  }
  int inc = cnt;
  for (int d = 1; d < 32; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }   // lanes >= 27 add nothing
  const int total = __shfl(inc, 31);
  const int exc = inc - cnt;
  if (total > 256) return false;
  int lo[4];
#pragma unroll
  for (int u = 0; u < 4; u++) lo[u] = 0;  // number of lanes whose inclusive sum is <= i == the lane owning item i (< 27)
#pragma unroll
  for (int step = 16; step > 0; step >>= 1) {
    int vv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) vv[u] = __shfl(inc, lo[u] + step - 1);
#pragma unroll
    for (int u = 0; u < 4; u++) if (vv[u] <= u * 64 + lane) lo[u] += step;
  }
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int i = u * 64 + lane;
    const int r = i - __shfl(exc, lo[u]), b0 = __shfl(bs, lo[u]);
    t[u] = i < total ? b0 + r : -1;
  }
#pragma unroll
  for (int u = 0; u < 4; u++) c[u] = gpts[t[u] >= 0 ? t[u] : 0];  // unconditional: the four loads go out back to back
  return true;
}

struct VisitNearest {  // LO:269 / LO:356: nearest candidate, ties to the lowest index
  float3 sel;
  u64 loc;
  int ring;   // stored scan line of the best candidate so far (== int(intensity) of that point)
  __device__ __forceinline__ void visit(float4 c) {
    const float d = sqdist(c, sel);
    const unsigned tag = __float_as_uint(c.w);
    const u64 key = ((u64)__float_as_uint(d) << 32) | (unsigned)tag_index(tag);
    const bool better = key < loc;
    loc = better ? key : loc;
    ring = better ? tag_ring(tag) : ring;
  }
};
struct VisitAdjacent {  // LO:279-324 / LO:368-417 as a class filter (see k_lo_assoc)
  float3 sel;
  int idx, ringA;
  int stop_f, stop_b;  // the indices at which the reference's upward / downward walk breaks (exclusive bounds)
  bool is_corner;
  u64 l2, l3;
  int visited;
  __device__ __forceinline__ void visit(float4 c) {
    const unsigned tag = __float_as_uint(c.w);
    const int j = tag_index(tag), rj = tag_ring(tag);
    visited++;
    const float d = sqdist(c, sel);
    const bool fwd = j > idx;
    const u64 key = ((u64)__float_as_uint(d) << 32) | (fwd ? (unsigned)(j - idx) : 0x40000000u + (unsigned)(idx - j));
    const bool ok = d < 25.0f && j != idx && j < stop_f && j > stop_b;
    // corner: upward walk skips scan lines <= ringA (LO:287-289), downward walk skips >= ringA (LO:311-313);
    // plane: upward, scan line <= ringA feeds the 2nd point, above it the 3rd (LO:376-389); downward mirrored (LO:402-415)
    const bool to2 = is_corner ? (fwd ? rj > ringA : rj < ringA) : (fwd ? rj <= ringA : rj >= ringA);
    const bool to3 = !is_corner && !to2;
    if (ok && to2) l2 = key < l2 ? key : l2;
    if (ok && to3) l3 = key < l3 ? key : l3;
  }
};


#define LO_ASSOC_ARGS const float4* __restrict__ sharp, const float4* __restrict__ flat, const FrameScalars* __restrict__ Sc, const float4* __restrict__ CL, \
                      const float4* __restrict__ SL, const FrameScalars* __restrict__ Sp, LoGrid G, const LOState* __restrict__ lo, FactorTable F,              \
                      int* __restrict__ corr, long long* __restrict__ dbg_cyc /* [slots][4] or null */,                                                         \
                      const int* __restrict__ queue /* null: every slot; else the slots k_lo_assoc_fast left over */,                                          \
                      int* __restrict__ queue_n /* [2]: [parity] entries of `queue`, [parity ^ 1] re-armed here */, int parity, size_t ss
__device__ __forceinline__ void lo_assoc_body(LO_ASSOC_ARGS, int (*s_inc_all)[512], int (*s_rel_all)[512]) {
  VL_SESSION(ss); RB(sharp); RB(flat); RB(Sc); RB(CL); RB(SL); RB(Sp); G.rebase(so_); RB(lo); F.rebase(so_); RB(corr); RB(dbg_cyc); RB(queue); RB(queue_n);
  const int lane = threadIdx.x & 63;
  int* s_inc = s_inc_all[threadIdx.x >> 6];
  int* s_rel = s_rel_all[threadIdx.x >> 6];
  // XCD-aware remap (workgroup b runs on XCD b % 8, each with its own L2): the features are ring / sector ordered, so neighbouring
  // slots scan overlapping grid blocks — one XCD takes a contiguous eighth of the corner slots and a contiguous eighth of the plane
  // slots (both kinds everywhere: a plane query costs more than a corner query) instead of every eighth workgroup
  constexpr int kWc = kMaxSharp / 4 / 8, kWp = kMaxFlat / 4 / 8;   // corner / plane workgroups per XCD
  static_assert(kMaxSharp % 32 == 0 && kMaxFlat % 32 == 0 && (kMaxLoFactors + 3) / 4 == 8 * (kWc + kWp), "bijective remap");
  const int bq_ = (int)(blockIdx.x >> 3), xcd_ = (int)(blockIdx.x & 7);
  int slot = (bq_ < kWc ? (xcd_ * kWc + bq_) * 4 : kMaxSharp + (xcd_ * kWp + (bq_ - kWc)) * 4) + (int)(threadIdx.x >> 6);
  if (queue) {   // second pass behind k_lo_assoc_fast: one wavefront per left-over query, in queue order
    const int qn = queue_n[parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) queue_n[parity ^ 1] = 0;   // the next launch pair's counter (its last readers finished a launch ago)
    const int qi_ = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (qi_ >= qn) return;
    slot = queue[qi_] & 0xffff;
  }
  if (slot >= kMaxLoFactors) return;
  const bool is_corner = slot < kMaxSharp;
  const int i = is_corner ? slot : slot - kMaxSharp;
  const int nfeat = is_corner ? Sc->n_sharp : Sc->n_flat;
  int type = 0, ia = -1, ib = -1, ic = -1;
  long long t0 = 0, t1 = 0, t2 = 0, tdbg = 0;
  int exact_dbg = -1, stage2_dbg = -1, cand_dbg = 0;
  if (dbg_cyc) t0 = clock64();
  if (i < nfeat) {
    const float4 pf = is_corner ? sharp[i] : flat[i];
    const float3 sel = transform_to_start(pf, lo->para_q, lo->para_t);  // LO:268 / LO:355
    const float4* cand = is_corner ? CL : SL;
    const int n = is_corner ? Sp->n_less_sharp : Sp->n_less_flat;
    // ---- exact nearest neighbour (pcl::KdTreeFLANN::nearestKSearch k = 1, flann::L2_Simple<float>), LO:269 / LO:356.
    // Expanding search over the 1 m hash grid: after every cell within Chebyshev radius R of the query's cell has been
    // scanned, any unseen point is farther than R metres; the search stops as soon as the best distance is inside that
    // bound.  Queries without a neighbour inside 2 m sweep the 27 cells of the 5 m grid, which hold every point within 5 m
    // (DISTANCE_SQ_THRESHOLD = 25 rejects anything farther).  Key = (f32 d2 bits, index): ties resolve to the lowest index
    // irrespective of visiting order.
    const int fcx = (int)floorf(sel.x), fcy = (int)floorf(sel.y), fcz = (int)floorf(sel.z);
    const int ccx = coarse_cell(sel.x), ccy = coarse_cell(sel.y), ccz = coarse_cell(sel.z);
    // stage 0: fine R = 1 block, 1: fine R = 2 shell, 2: coarse 27 cells.  bound = squared radius fully covered so far.
    // kernel-argument arrays are only ever indexed with constants (a runtime index would move the struct into scratch)
    const int* fstart = is_corner ? G.start[0] : G.start[1];
    const int* cstart = is_corner ? G.start[2] : G.start[3];
    const float4* fpts = is_corner ? G.pts[0] : G.pts[1];
    const float4* cpts = is_corner ? G.pts[2] : G.pts[3];
    const unsigned fmask = (unsigned)(is_corner ? G.mask[0] : G.mask[1]), cmask = (unsigned)(is_corner ? G.mask[2] : G.mask[3]);
    // stage 0: 1 m level, R = 1 block; 1: R = 2 shell; 2: the 27 cells of the 5 m level, ring groups [g0, g0 + ng)
    // Search plan, for the closest point and again for the second / third point: (0) the radius-1 block of the 1 m level
    // (after radius R every unseen point is farther than R metres); (1) if that is not conclusive, the 27 cells of the 5 m
    // level, which hold every point within DISTANCE_SQ_THRESHOLD — final, but only taken when the neighbourhood is sparse
    // (<= kSparse points, fetched in ONE trip: features far from everything); otherwise (2, 3) the radius-2 and radius-3 shells of the 1 m level,
    // which are conclusive in any dense neighbourhood, and (4) the 5 m level unconditionally.
    constexpr int kSparse = 1024, kAll = 0x7fffffff;
    // points per lane and trip on the 5 m level: 8 (two trips for the largest sparse neighbourhoods) keeps the kernel at 158
    // VGPRs = 3 wavefronts per SIMD, so that all ~1 800 queries stay resident in one round even while the scan-registration
    // ring kernel (one 146 KB-LDS workgroup per scan line) holds 51 of the 256 compute units
    constexpr int kSparseUB = 8;
    constexpr int kGroups = kRingGroups;
    auto stage_bound = [](int stage) { return stage == 0 ? 1.0f * 0.999999f : (stage == 2 ? 4.0f * 0.999999f : (stage == 3 ? 9.0f * 0.999999f : 3.0e38f)); };
    // walk stops of this kind's cloud, one table entry per lane (ringA indexes them through a cross-lane read below)
    const int* stops = is_corner ? G.stops : G.stops + 2 * kStopLen;
    const int sf_l = stops[lane + 3], sb_l = stops[kStopLen + lane];  // first index with line >= lane + 3, last with line <= lane - 3
    u64 best = ~0ull;
    int ringA = -1;
    // FAST PATH (most queries): the radius-1 block answers the closest point; its candidates stay in registers for the second /
    // third point
    float4 c0[4];
    int t0[4];
    bool fast = false, adj_done = false;
    int stop_f = 0, stop_b = 0;
    u64 b2 = ~0ull, b3 = ~0ull;
    const bool kept = n > 0 && fetch_block1(fstart, fpts, fmask, fcx, fcy, fcz, lane, c0, t0);
    if (kept) {
      VisitNearest vn;
      vn.sel = sel; vn.loc = ~0ull; vn.ring = 0;
#pragma unroll
      for (int u = 0; u < 4; u++) if (t0[u] >= 0) vn.visit(c0[u]);
      best = wave_min_u64(vn.loc);
      if (best != ~0ull && __uint_as_float((unsigned)(best >> 32)) <= stage_bound(0)) {
        fast = true;
        exact_dbg = 0;
        ringA = __shfl(vn.ring, __ffsll((long long)__ballot(vn.loc == best)) - 1);
        stop_f = __shfl(sf_l, ringA); stop_b = __shfl(sb_l, ringA);
        // stage 0 of the second / third point plan (see below) on the same registers; c0 is not needed after this block
        VisitAdjacent va;
        va.sel = sel; va.idx = (int)(best & 0xffffffffu); va.ringA = ringA; va.stop_f = stop_f; va.stop_b = stop_b; va.is_corner = is_corner;
        va.l2 = ~0ull; va.l3 = ~0ull; va.visited = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) if (t0[u] >= 0) va.visit(c0[u]);
        cand_dbg += va.visited;
        b2 = wave_min_u64(va.l2);
        if (!is_corner) b3 = wave_min_u64(va.l3);
        const bool done2 = b2 != ~0ull && __uint_as_float((unsigned)(b2 >> 32)) <= stage_bound(0);
        const bool done3 = is_corner || (b3 != ~0ull && __uint_as_float((unsigned)(b3 >> 32)) <= stage_bound(0));
        stage2_dbg = 0;
        adj_done = done2 && done3;
      }
    }
    if (!fast)
    for (int stage = kept ? 1 : 0; stage < 5 && n > 0; stage++) {
      VisitNearest vn;
      vn.sel = sel; vn.loc = ~0ull; vn.ring = 0;
      bool skipped = false;
      if (stage == 0) vn = for_each_candidate<1, 4, 1, -1>(fstart, fpts, fmask, fcx, fcy, fcz, 0, 1, lane, vn, kAll, &skipped, s_inc, s_rel);
      else if (stage == 1) {
        long long tm[5] = {0, 0, 0, 0, 0};
        vn = for_each_candidate<1, kSparseUB, 1, -1, true>(cstart, cpts, cmask, ccx, ccy, ccz, 0, kGroups, lane, vn, kSparse, &skipped, s_inc, s_rel, dbg_cyc ? tm : nullptr);
        if (dbg_cyc) tdbg = ((tm[1] - tm[0]) & 0xffff) | (((tm[2] - tm[1]) & 0xffff) << 16) | (((tm[3] - tm[2]) & 0xffff) << 32) | ((tm[4] & 0xffff) << 48);
      }
      else if (stage == 2) continue;  // (the closest point: both shells in one pass, the few queries that get here are the kernel's tail)
      else if (stage == 3) vn = for_each_candidate<6, 4, 3, 1>(fstart, fpts, fmask, fcx, fcy, fcz, 0, 1, lane, vn, kAll, &skipped, s_inc, s_rel);
      else vn = for_each_candidate<1, kSparseUB, 1, -1, true>(cstart, cpts, cmask, ccx, ccy, ccz, 0, kGroups, lane, vn, kAll, &skipped, s_inc, s_rel);
      if (skipped) continue;
      const u64 loc = wave_min_u64(vn.loc);
      best = loc < best ? loc : best;
      if (stage == 1 || (best != ~0ull && __uint_as_float((unsigned)(best >> 32)) <= stage_bound(stage))) { exact_dbg = stage; break; }
    }
    const float dmin = __uint_as_float((unsigned)(best >> 32));
    if (dbg_cyc) t1 = clock64();
    if (best != ~0ull && dmin < 25.0f) {  // DISTANCE_SQ_THRESHOLD, LO:272 / LO:359
      const int idx = (int)(best & 0xffffffffu);
      if (!fast) ringA = (int)cand[idx].w;  // closestPointScanID (the fast path read it off the winning candidate's tag)
      // ---- second (and third) point: the reference walks the ring-sorted cloud upwards from idx + 1 until the scan line
      // exceeds ringA + NEARBY_SCAN and downwards from idx - 1 until it drops below ringA - NEARBY_SCAN, keeping the nearest
      // point with d2 < 25 per class, first strictly smaller wins (LO:279-324 / LO:368-417).  On a ring-sorted cloud that is
      // the minimum of (d2, visiting order) over the index interval between the two break points, split into classes by
      // (j > idx, ring_j), so it is answered by the same expanding grid search with the class filter applied to every candidate.
      // Where the walks break.  The stored scan line int(intensity) of a point of scan line r is r or r - 1 (the fractional part
      // 0.1 * relTime lies in (-0.05, 0.15): SR:237-265), so the clouds are only ALMOST sorted by it: the upward walk breaks at
      // the first point anywhere with a line > ringA + NEARBY_SCAN, the downward walk at the last one with a line below
      // ringA - NEARBY_SCAN, and points beyond a break are never looked at even if their own line is in range.
      if (!fast) { stop_f = __shfl(sf_l, ringA); stop_b = __shfl(sb_l, ringA); }  // first index with line >= ringA + 3, last with line <= ringA - 3
      const int glo = max(ringA - 2, 0) >> kRingGroupShift, ghi = min(ringA + 2, kMaxRings - 1) >> kRingGroupShift;
      if (!adj_done)
      for (int stage = fast ? 1 : 0; stage < 5; stage++) {
        VisitAdjacent va;
        va.sel = sel; va.idx = idx; va.ringA = ringA; va.stop_f = stop_f; va.stop_b = stop_b; va.is_corner = is_corner;
        va.l2 = ~0ull; va.l3 = ~0ull; va.visited = 0;
        bool skipped = false;
        if (stage == 0) va = for_each_candidate<1, 4, 1, -1>(fstart, fpts, fmask, fcx, fcy, fcz, 0, 1, lane, va, kAll, &skipped, s_inc, s_rel);
        else if (stage == 1) va = for_each_candidate<1, kSparseUB, 1, -1, true>(cstart, cpts, cmask, ccx, ccy, ccz, glo, ghi - glo + 1, lane, va, kSparse, &skipped, s_inc, s_rel);  // <= 2 groups of 27 cells
        else if (stage == 2) va = for_each_candidate<2, 4, 2, 1>(fstart, fpts, fmask, fcx, fcy, fcz, 0, 1, lane, va, kAll, &skipped, s_inc, s_rel);
        else if (stage == 3) va = for_each_candidate<6, 4, 3, 2>(fstart, fpts, fmask, fcx, fcy, fcz, 0, 1, lane, va, kAll, &skipped, s_inc, s_rel);
        else va = for_each_candidate<1, kSparseUB, 1, -1, true>(cstart, cpts, cmask, ccx, ccy, ccz, glo, ghi - glo + 1, lane, va, kAll, &skipped, s_inc, s_rel);
        if (skipped) continue;
        cand_dbg += va.visited;
        const u64 l2 = wave_min_u64(va.l2);
        b2 = l2 < b2 ? l2 : b2;
        if (!is_corner) { const u64 l3 = wave_min_u64(va.l3); b3 = l3 < b3 ? l3 : b3; }
        const bool done2 = b2 != ~0ull && __uint_as_float((unsigned)(b2 >> 32)) <= stage_bound(stage);
        const bool done3 = is_corner || (b3 != ~0ull && __uint_as_float((unsigned)(b3 >> 32)) <= stage_bound(stage));
        stage2_dbg = stage;
        if (stage == 1 || (done2 && done3)) break;
      }
      if (dbg_cyc) t2 = clock64();
      auto decode = [&](u64 k) {
        const unsigned o = (unsigned)(k & 0xffffffffu);
        return o >= kBack ? idx - (int)(o - kBack) : idx + (int)o;
      };
      if (is_corner) {
        if (b2 != ~0ull) {  // LO:326-349
          ia = idx; ib = decode(b2);
          type = 1;
          if (lane == 0) {
            const float4 a = cand[ia], b = cand[ib];
            const int cap = F.cap;
            F.p[slot] = pf.x; F.p[cap + slot] = pf.y; F.p[2 * cap + slot] = pf.z;
            F.A[slot] = a.x; F.A[cap + slot] = a.y; F.A[2 * cap + slot] = a.z;
            F.B[slot] = b.x; F.B[cap + slot] = b.y; F.B[2 * cap + slot] = b.z;
            const double Ad[3] = {(double)a.x, (double)a.y, (double)a.z}, Bd[3] = {(double)b.x, (double)b.y, (double)b.z};
            factor_digest(F, slot, 1, Ad, Bd);
          }
        }
      } else if (b2 != ~0ull && b3 != ~0ull) {  // LO:419-442
        ia = idx; ib = decode(b2); ic = decode(b3);
        type = 2;
        if (lane == 0) {
          const float4 pj = cand[ia], pl = cand[ib], pm = cand[ic];
          // LidarPlaneFactor ctor (lidarFactor.hpp:62-70): ljm_norm = normalize((j - l) x (j - m))
          const double ax = (double)pj.x - (double)pl.x, ay = (double)pj.y - (double)pl.y, az = (double)pj.z - (double)pl.z;
          const double bx = (double)pj.x - (double)pm.x, by = (double)pj.y - (double)pm.y, bz = (double)pj.z - (double)pm.z;
          double nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
          const double nn = sqrt(nx * nx + ny * ny + nz * nz);
          nx = nx / nn; ny = ny / nn; nz = nz / nn;
          const int cap = F.cap;
          F.p[slot] = pf.x; F.p[cap + slot] = pf.y; F.p[2 * cap + slot] = pf.z;
          F.A[slot] = pj.x; F.A[cap + slot] = pj.y; F.A[2 * cap + slot] = pj.z;
          F.B[slot] = nx; F.B[cap + slot] = ny; F.B[2 * cap + slot] = nz;
          const double Ad[3] = {(double)pj.x, (double)pj.y, (double)pj.z}, Bd[3] = {nx, ny, nz};
          factor_digest(F, slot, 2, Ad, Bd);
        }
      }
    }
  }
  if (lane == 0) {
    F.type[slot] = type;
    if (type) atomicAdd(&F.rowcnt[slot >> 6], 1);
    if (dbg_cyc) { dbg_cyc[slot * 4] = t1 - t0; dbg_cyc[slot * 4 + 1] = t2 - t1; dbg_cyc[slot * 4 + 2] = tdbg; dbg_cyc[slot * 4 + 3] = (exact_dbg & 0xff) | ((stage2_dbg & 0xff) << 8) | ((long long)cand_dbg << 16); }
    corr[slot * 4 + 0] = type ? i : -1;
    corr[slot * 4 + 1] = ia; corr[slot * 4 + 2] = ib; corr[slot * 4 + 3] = ic;
  }
}

// The wave-per-query kernel in two register budgets.  One sequence: 158 VGPRs, three wavefronts per SIMD — every query of the launch is resident at
// once and none of its state spills.  Behind k_lo_assoc_fast in a batch (thousands of left-over queries from B sessions, on a chip the other
// stages keep full): four wavefronts per SIMD at 128 VGPRs and a few spilled values finish the pass sooner (B = 8: 123 -> 102 us; the same
// budget costs one sequence 36 -> 40 us).
#define LO_ASSOC_PASS sharp, flat, Sc, CL, SL, Sp, G, lo, F, corr, dbg_cyc, queue, queue_n, parity, ss
__global__ __launch_bounds__(256) void k_lo_assoc(LO_ASSOC_ARGS) {
  __shared__ int s_inc_all[4][512], s_rel_all[4][512];  // per-wavefront staging of cell prefix sums (for_each_candidate, KC > 2)
  lo_assoc_body(LO_ASSOC_PASS, s_inc_all, s_rel_all);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_lo_assoc_dense(LO_ASSOC_ARGS) {
  __shared__ int s_inc_all[4][512], s_rel_all[4][512];
  lo_assoc_body(LO_ASSOC_PASS, s_inc_all, s_rel_all);
}

// ---- k_lo_assoc_fast: the common query with G = 16 lanes (four queries per wavefront, sixteen per workgroup).
// ~85 % of the queries are answered by the radius-1 block of the 1 m level alone (closest point within 1 m, second / third point
// within 1 m: stage 0 of both plans of k_lo_assoc).  A full wavefront per such query spends its time in cross-lane round trips
// (prefix sums, item -> cell searches, three 64-bit reductions through ds_bpermute) around ~200 candidate points; here a 16-lane
// group does the same work for its own query with DPP row operations, its 27 bucket ranges two per lane, its candidates
// (<= 256) sixteen per lane in chunks of four loads, (d2, tag) of every candidate parked in LDS for the second / third point pass.
// A query the block cannot answer conclusively is NOT continued here: its slot goes onto a queue and k_lo_assoc (one wavefront per
// query, all stages) takes it from scratch in the launch right behind — the long tail runs with full wavefronts and balanced, the
// short queries no longer wait in line behind it.  Same keys, same bounds, same emission as k_lo_assoc: results are identical.
#ifndef VLOAM_LO_FAST_CAP
#define VLOAM_LO_FAST_CAP 256
#endif
constexpr int kFastCap = VLOAM_LO_FAST_CAP;   // candidates of the radius-1 block a 16-lane group parks in LDS (8 B each)
template <int G>
__global__ __launch_bounds__(256) void k_lo_assoc_fast(const float4* __restrict__ sharp, const float4* __restrict__ flat,
                                                       const FrameScalars* __restrict__ Sc, const float4* __restrict__ CL,
                                                       const float4* __restrict__ SL, const FrameScalars* __restrict__ Sp, LoGrid Gd,
                                                       const LOState* __restrict__ lo, FactorTable F, int* __restrict__ corr,
                                                       long long* __restrict__ dbg_cyc, int* __restrict__ queue, int* __restrict__ queue_n,
                                                       int parity, size_t ss) {
  VL_SESSION(ss); RB(sharp); RB(flat); RB(Sc); RB(CL); RB(SL); RB(Sp); Gd.rebase(so_); RB(lo); F.rebase(so_); RB(corr); RB(dbg_cyc); RB(queue); RB(queue_n);
  static_assert(G == 16, "27 bucket ranges two per lane, 256 candidates sixteen per lane");
  constexpr int Q = 64 / G, QW = 4 * Q, CHK = 4;   // queries per wavefront / workgroup, candidate loads per lane and trip
  constexpr int kWc = kMaxSharp / QW / 8, kWp = kMaxFlat / QW / 8;   // corner / plane workgroups per XCD (XCD-aware remap as in k_lo_assoc)
  static_assert(kMaxSharp % (8 * QW) == 0 && kMaxFlat % (8 * QW) == 0, "bijective remap");
  __shared__ int s_stops[2 * kStopLen];
  __shared__ int s_inc[QW][32], s_rel[QW][32];
  __shared__ int s_ring[QW];
  const int tid = threadIdx.x, lane = tid & 63, gl = lane & (G - 1), qi = (tid >> 6) * Q + lane / G;
  const int bq = (int)(blockIdx.x >> 3), xcd = (int)(blockIdx.x & 7);
  const bool is_corner = bq < kWc;                                    // a workgroup serves one kind
  const int slot = (is_corner ? (xcd * kWc + bq) * QW : kMaxSharp + (xcd * kWp + (bq - kWc)) * QW) + qi;
  {
    const int* stops = is_corner ? Gd.stops : Gd.stops + 2 * kStopLen;  // where the reference's walks break (see k_lo_assoc)
    if (tid < 2 * kStopLen) s_stops[tid] = stops[tid];
  }
  __syncthreads();
  const int i = is_corner ? slot : slot - kMaxSharp;
  const int nfeat = is_corner ? Sc->n_sharp : Sc->n_flat;
  const int n = is_corner ? Sp->n_less_sharp : Sp->n_less_flat;
  const bool live = i < nfeat;
  const bool act = live && n > 0;
  float4 pf = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) pf = is_corner ? sharp[i] : flat[i];
  const float3 sel = transform_to_start(pf, lo->para_q, lo->para_t);  // LO:268 / LO:355
  const float4* cand = is_corner ? CL : SL;
  const int* fstart = is_corner ? Gd.start[0] : Gd.start[1];
  const float4* fpts = is_corner ? Gd.pts[0] : Gd.pts[1];
  const unsigned fmask = (unsigned)(is_corner ? Gd.mask[0] : Gd.mask[1]);
  const int fcx = (int)floorf(sel.x), fcy = (int)floorf(sel.y), fcz = (int)floorf(sel.z);
  // the 27 bucket ranges of the radius-1 block: cells gl and gl + 16
  int bs[2] = {0, 0}, cnt[2] = {0, 0};
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int c = gl + 16 * k;
    if (act && c < 27) {
      const int ox = c % 3 - 1, oy = (c / 3) % 3 - 1, oz = c / 9 - 1;
      const unsigned b = grid_hash(fcx + ox, fcy + oy, fcz + oz) & fmask;
      bs[k] = fstart[b];
      cnt[k] = fstart[b + 1] - bs[k];
    }
  }
  int inc0 = grp_scan_incl<G>(cnt[0]), tot0 = grp_sum<G>(cnt[0]);
  int inc1 = tot0 + grp_scan_incl<G>(cnt[1]);
  int total = tot0 + grp_sum<G>(cnt[1]);
  // A block too full for the group's LDS list (> kFastCap points: the rule for plane features on the dense ground near the sensor — 90 % of
  // what used to be left to the wave-per-query pass) is cut down to the cells that can hold a point within HALF a metre of the query: a
  // cell at offset o is at least dmin(o) away (the query's position inside its own cell decides), cells with dmin^2 > 0.25 are dropped,
  // and every result is then only final within that smaller bound — in a neighbourhood this dense it practically always is.
  const bool crowded = act && total > kFastCap;
  float bound0 = 1.0f * 0.999999f;   // every point outside the radius-1 block is farther than 1 m
  if (__ballot(crowded) != 0ull) {
    const float fx = sel.x - (float)fcx, fy = sel.y - (float)fcy, fz = sel.z - (float)fcz;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int c = gl + 16 * k;
      const int ox = c % 3 - 1, oy = (c / 3) % 3 - 1, oz = c / 9 - 1;
      const float ax = ox < 0 ? fx : (ox > 0 ? 1.0f - fx : 0.f), ay = oy < 0 ? fy : (oy > 0 ? 1.0f - fy : 0.f), az = oz < 0 ? fz : (oz > 0 ? 1.0f - fz : 0.f);
      if (crowded && (ax * ax + ay * ay) + az * az > 0.25f) cnt[k] = 0;
    }
    inc0 = grp_scan_incl<G>(cnt[0]); tot0 = grp_sum<G>(cnt[0]);
    inc1 = tot0 + grp_scan_incl<G>(cnt[1]);
    total = tot0 + grp_sum<G>(cnt[1]);
    if (crowded) bound0 = 0.25f * 0.9999f;
    // ... and once more at a quarter of a metre where even those cells hold too many
    const bool crowded2 = crowded && total > kFastCap;
    if (__ballot(crowded2) != 0ull) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int c = gl + 16 * k;
        const int ox = c % 3 - 1, oy = (c / 3) % 3 - 1, oz = c / 9 - 1;
        const float ax = ox < 0 ? fx : (ox > 0 ? 1.0f - fx : 0.f), ay = oy < 0 ? fy : (oy > 0 ? 1.0f - fy : 0.f), az = oz < 0 ? fz : (oz > 0 ? 1.0f - fz : 0.f);
        if (crowded2 && (ax * ax + ay * ay) + az * az > 0.0625f) cnt[k] = 0;
      }
      inc0 = grp_scan_incl<G>(cnt[0]); tot0 = grp_sum<G>(cnt[0]);
      inc1 = tot0 + grp_scan_incl<G>(cnt[1]);
      total = tot0 + grp_sum<G>(cnt[1]);
      if (crowded2) bound0 = 0.0625f * 0.9999f;
    }
  }
  s_inc[qi][gl] = inc0; s_inc[qi][16 + gl] = inc1;                      // inclusive sums in (k, lane) order
  s_rel[qi][gl] = bs[0] - (inc0 - cnt[0]); s_rel[qi][16 + gl] = bs[1] - (inc1 - cnt[1]);   // item i of a cell lives at rel + i
  sw_lds_sync();
  const bool kept = act && total <= kFastCap;   // (more: k_lo_assoc walks the block through for_each_candidate)
  // ---- closest point (LO:269 / LO:356): every candidate once; (d2, tag) parked for the second / third point
  u64 loc = ~0ull;
  int ring = 0;
  for (int u0 = 0; __ballot(kept && u0 * G < total) != 0ull; u0 += CHK) {
    int addr[CHK];
    float4 c4[CHK];
#pragma unroll
    for (int u = 0; u < CHK; u++) {
      const int item = (u0 + u) * G + gl;
      addr[u] = -1;
      if (kept && item < total) {
        int pos = 0;   // number of cells whose inclusive sum is <= item == the cell holding the item
#pragma unroll
        for (int step = 16; step > 0; step >>= 1) if (s_inc[qi][pos + step - 1] <= item) pos += step;
        addr[u] = s_rel[qi][pos & 31] + item;
      }
    }
#pragma unroll
    for (int u = 0; u < CHK; u++) c4[u] = fpts[addr[u] >= 0 ? addr[u] : 0];   // unconditional: the loads go out back to back
#pragma unroll
    for (int u = 0; u < CHK; u++) {
      if (addr[u] >= 0) {
        const float d = sqdist(c4[u], sel);
        const unsigned tag = __float_as_uint(c4[u].w);
        const u64 key = ((u64)__float_as_uint(d) << 32) | (unsigned)tag_index(tag);
        const bool better = key < loc;
        loc = better ? key : loc;
        ring = better ? tag_ring(tag) : ring;
      }
    }
  }
  const u64 best = grp_min_u64<G>(loc);
  const float kBound0 = bound0;
  const bool fastq = kept && best != ~0ull && __uint_as_float((unsigned)(best >> 32)) <= kBound0;
  if (fastq && loc == best) s_ring[qi] = ring;   // the index part makes the key unique: one owner
  sw_lds_sync();
  // ---- second / third point on the same candidates (stage 0 of the plan in k_lo_assoc; class filter of VisitAdjacent)
  const int idx = (int)(best & 0xffffffffu);
  const int ringA = fastq ? s_ring[qi] : 0;
  const int stop_f = s_stops[ringA + 3], stop_b = s_stops[kStopLen + ringA];
  // (the candidates are fetched a second time — they sit in this CU's cache — instead of being parked in LDS between the passes: 32 KB of
  // LDS per workgroup held the kernel at four workgroups per CU, and this kernel's duration is rounds x per-workgroup latency)
  u64 l2 = ~0ull, l3 = ~0ull;
  for (int u0 = 0; __ballot(fastq && u0 * G < total) != 0ull; u0 += CHK) {
    int addr[CHK];
    float4 c4[CHK];
#pragma unroll
    for (int u = 0; u < CHK; u++) {
      const int item = (u0 + u) * G + gl;
      addr[u] = -1;
      if (fastq && item < total) {
        int pos = 0;
#pragma unroll
        for (int step = 16; step > 0; step >>= 1) if (s_inc[qi][pos + step - 1] <= item) pos += step;
        addr[u] = s_rel[qi][pos & 31] + item;
      }
    }
#pragma unroll
    for (int u = 0; u < CHK; u++) c4[u] = fpts[addr[u] >= 0 ? addr[u] : 0];
#pragma unroll
    for (int u = 0; u < CHK; u++) {
      if (addr[u] >= 0) {
        const float d = sqdist(c4[u], sel);
        const unsigned tag = __float_as_uint(c4[u].w);
        const int j = tag_index(tag), rj = tag_ring(tag);
        const bool fwd = j > idx;
        const u64 key = ((u64)__float_as_uint(d) << 32) | (fwd ? (unsigned)(j - idx) : 0x40000000u + (unsigned)(idx - j));
        const bool ok = d < 25.0f && j != idx && j < stop_f && j > stop_b;
        const bool to2 = is_corner ? (fwd ? rj > ringA : rj < ringA) : (fwd ? rj <= ringA : rj >= ringA);
        const bool to3 = !is_corner && !to2;
        if (ok && to2) l2 = key < l2 ? key : l2;
        if (ok && to3) l3 = key < l3 ? key : l3;
      }
    }
  }
  u64 b2 = grp_min_u64<G>(l2);
  u64 b3 = grp_min_u64<G>(l3);
  const bool done2 = b2 != ~0ull && __uint_as_float((unsigned)(b2 >> 32)) <= kBound0;
  const bool done3 = is_corner || (b3 != ~0ull && __uint_as_float((unsigned)(b3 >> 32)) <= kBound0);
  // ---- stage 1 of the second / third point plan (k_lo_assoc), also by the group: the closest point is settled (within 1 m) but its partner
  // on a neighbouring scan line is farther than the radius-1 block reaches — the rule on the ground beyond ~15 m, where the lines lie
  // metres apart (44 % of all queries of a 64 x 2048 sweep went to the wave-per-query pass for this, at eleven times the instructions).
  // The 27 cells of the 5 m level hold every point within DISTANCE_SQ_THRESHOLD; restricted to the one or two groups of four scan lines
  // around the closest point's line they are the search's FINAL stage whenever they hold no more than kSparse2 points (the same rule and
  // the same bound as stage 1 of k_lo_assoc, so the answers are the same); a denser neighbourhood goes to the queue as before.
  constexpr int kSparse2 = 1024;
  const bool need2 = act && fastq && !(done2 && done3);
  bool final2 = false;
  if (__ballot(need2) != 0ull) {
    const int* cstart = is_corner ? Gd.start[2] : Gd.start[3];
    const float4* cpts = is_corner ? Gd.pts[2] : Gd.pts[3];
    const unsigned cmask = (unsigned)(is_corner ? Gd.mask[2] : Gd.mask[3]);
    const int ccx = coarse_cell(sel.x), ccy = coarse_cell(sel.y), ccz = coarse_cell(sel.z);
    const int glo = max(ringA - 2, 0) >> kRingGroupShift, ghi = min(ringA + 2, kMaxRings - 1) >> kRingGroupShift;
    const int ng = ghi - glo + 1;
    int bs2[2] = {0, 0}, cn2[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int c = gl + 16 * k;
      if (need2 && c < 27) {
        const int ox = c % 3 - 1, oy = (c / 3) % 3 - 1, oz = c / 9 - 1;
        const unsigned b = coarse_slot(ccx + ox, ccy + oy, ccz + oz, cmask) + (unsigned)glo;
        bs2[k] = cstart[b];
        cn2[k] = cstart[b + ng] - bs2[k];
      }
    }
    const int i0 = grp_scan_incl<G>(cn2[0]), t0 = grp_sum<G>(cn2[0]);
    const int i1 = t0 + grp_scan_incl<G>(cn2[1]);
    const int total2 = t0 + grp_sum<G>(cn2[1]);
    sw_lds_sync();   // (the radius-1 block's prefix sums are no longer read)
    s_inc[qi][gl] = i0; s_inc[qi][16 + gl] = i1;
    s_rel[qi][gl] = bs2[0] - (i0 - cn2[0]); s_rel[qi][16 + gl] = bs2[1] - (i1 - cn2[1]);
    sw_lds_sync();
    final2 = need2 && total2 <= kSparse2;
    u64 m2 = ~0ull, m3 = ~0ull;
    constexpr int CHK2 = 8;   // (up to 64 candidates per lane here: eight loads in flight per trip)
    for (int u0 = 0; __ballot(final2 && u0 * G < total2) != 0ull; u0 += CHK2) {
      int addr[CHK2];
      float4 c4[CHK2];
#pragma unroll
      for (int u = 0; u < CHK2; u++) {
        const int item = (u0 + u) * G + gl;
        addr[u] = -1;
        if (final2 && item < total2) {
          int pos = 0;
#pragma unroll
          for (int step = 16; step > 0; step >>= 1) if (s_inc[qi][pos + step - 1] <= item) pos += step;
          addr[u] = s_rel[qi][pos & 31] + item;
        }
      }
#pragma unroll
      for (int u = 0; u < CHK2; u++) c4[u] = cpts[addr[u] >= 0 ? addr[u] : 0];
#pragma unroll
      for (int u = 0; u < CHK2; u++) {
        if (addr[u] >= 0) {
          const float d = sqdist(c4[u], sel);
          const unsigned tag = __float_as_uint(c4[u].w);
          const int j = tag_index(tag), rj = tag_ring(tag);
          const bool fwd = j > idx;
          const u64 key = ((u64)__float_as_uint(d) << 32) | (fwd ? (unsigned)(j - idx) : 0x40000000u + (unsigned)(idx - j));
          const bool ok = d < 25.0f && j != idx && j < stop_f && j > stop_b;
          const bool to2 = is_corner ? (fwd ? rj > ringA : rj < ringA) : (fwd ? rj <= ringA : rj >= ringA);
          const bool to3 = !is_corner && !to2;
          if (ok && to2) m2 = key < m2 ? key : m2;
          if (ok && to3) m3 = key < m3 ? key : m3;
        }
      }
    }
    const u64 g2 = grp_min_u64<G>(m2), g3 = grp_min_u64<G>(m3);
    if (final2) { b2 = g2 < b2 ? g2 : b2; b3 = g3 < b3 ? g3 : b3; }
  }
  const bool resolved = !act || (fastq && done2 && done3) || final2;
  if (gl == 0) {
    if (!resolved) {
      queue[atomicAdd(&queue_n[parity], 1)] = slot | ((!kept ? 1 : (!fastq ? 2 : 3)) << 16);   // k_lo_assoc takes it from scratch (bits 16+: why — a diagnostic, vloam_debug_get(1, 6))
    } else {
      int type = 0, ia = -1, ib = -1, ic = -1;
      if (act && (is_corner ? b2 != ~0ull : (b2 != ~0ull && b3 != ~0ull))) {   // (after the final stage a partner may simply not exist: no factor, LO:326 / LO:419)
        auto decode = [&](u64 k) {
          const unsigned o = (unsigned)(k & 0xffffffffu);
          return o >= kBack ? idx - (int)(o - kBack) : idx + (int)o;
        };
        const int cap = F.cap;
        if (is_corner) {  // LO:326-349
          ia = idx; ib = decode(b2);
          type = 1;
          const float4 a = cand[ia], b = cand[ib];
          F.p[slot] = pf.x; F.p[cap + slot] = pf.y; F.p[2 * cap + slot] = pf.z;
          F.A[slot] = a.x; F.A[cap + slot] = a.y; F.A[2 * cap + slot] = a.z;
          F.B[slot] = b.x; F.B[cap + slot] = b.y; F.B[2 * cap + slot] = b.z;
          const double Ad[3] = {(double)a.x, (double)a.y, (double)a.z}, Bd[3] = {(double)b.x, (double)b.y, (double)b.z};
          factor_digest(F, slot, 1, Ad, Bd);
        } else {          // LO:419-442
          ia = idx; ib = decode(b2); ic = decode(b3);
          type = 2;
          const float4 pj = cand[ia], pl = cand[ib], pm = cand[ic];
          // LidarPlaneFactor ctor (lidarFactor.hpp:62-70): ljm_norm = normalize((j - l) x (j - m))
          const double ax = (double)pj.x - (double)pl.x, ay = (double)pj.y - (double)pl.y, az = (double)pj.z - (double)pl.z;
          const double bx = (double)pj.x - (double)pm.x, by = (double)pj.y - (double)pm.y, bz = (double)pj.z - (double)pm.z;
          double nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
          const double nn = sqrt(nx * nx + ny * ny + nz * nz);
          nx = nx / nn; ny = ny / nn; nz = nz / nn;
          F.p[slot] = pf.x; F.p[cap + slot] = pf.y; F.p[2 * cap + slot] = pf.z;
          F.A[slot] = pj.x; F.A[cap + slot] = pj.y; F.A[2 * cap + slot] = pj.z;
          F.B[slot] = nx; F.B[cap + slot] = ny; F.B[2 * cap + slot] = nz;
          const double Ad[3] = {(double)pj.x, (double)pj.y, (double)pj.z}, Bd[3] = {nx, ny, nz};
          factor_digest(F, slot, 2, Ad, Bd);
        }
      }
      F.type[slot] = type;
      if (type) atomicAdd(&F.rowcnt[slot >> 6], 1);
      if (dbg_cyc) { dbg_cyc[slot * 4] = 0; dbg_cyc[slot * 4 + 1] = 0; dbg_cyc[slot * 4 + 2] = 0; dbg_cyc[slot * 4 + 3] = act ? 0 : 0xffff; }
      corr[slot * 4 + 0] = type ? i : -1;
      corr[slot * 4 + 1] = ia; corr[slot * 4 + 2] = ib; corr[slot * 4 + 3] = ic;
    }
  }
}

// LO:223-236 — combined mode overwrites the warm start with the VO prior at the top of each outer round.
// With vo_row7 != nullptr this launch is also where the frame's visual odometry is PUBLISHED (MAIN/src/vloam_main_node.cpp:158-162):
//   solveNlsAll's tail (VO:425-430): cam0_curr_T_cam0_last from (angle-axis, t) — only when a solve ran this frame (count > 0)
//   VloamTF::VO2VeloAndBase (vloam_tf.cpp:59-75): velo_last_VOT_velo_curr -> prior_q / prior_t, world_VOT_base_last *= base_last_VOT_base_curr
//   the VO row of the trajectory log: world_VOT_base_last as (q xyzw, t)
__global__ void k_lo_set_prior(LOState* lo, int copy_to_para, const double* vo_x, int vo_solved, double* vo_row7, int* err, size_t ss) {
  if (threadIdx.x != 0) return;
  VL_SESSION(ss); RB(lo); RB(vo_x); RB(vo_row7); RB(err);
  if (vo_row7) {
    VloamTfState& tf = lo->tf;
    if (vo_solved) {
      const double a0 = vo_x[0], a1 = vo_x[1], a2 = vo_x[2];
      const double angle = sqrt((a0 * a0 + a1 * a1) + a2 * a2);                  // VO:427
      const double ax = a0 / angle, ay = a1 / angle, az = a2 / angle;            // NaN for a zero angle, exactly like the reference
      const double d = sqrt((ax * ax + ay * ay) + az * az);                      // Quaternion::setRotation(axis, angle)
      const double sn = sin(angle * 0.5) / d;
      const double q[4] = {ax * sn, ay * sn, az * sn, cos(angle * 0.5)};
      tf_set_rotation(&tf.cam0_curr_T_cam0_last, q);
      tf.cam0_curr_T_cam0_last.o[0] = vo_x[3]; tf.cam0_curr_T_cam0_last.o[1] = vo_x[4]; tf.cam0_curr_T_cam0_last.o[2] = vo_x[5];
      if (!(angle > 0.0)) { tf.vo_nan_frames++; if (err) atomicOr(err, kErrVoDegenerate); }
    }
    TfDev inv, vi, bi, velo_last, base_last;
    tf_inverse(tf.cam0_curr_T_cam0_last, &inv);
    tf_inverse(tf.velo_T_cam0, &vi);
    tf_inverse(tf.base_T_cam0, &bi);
    tf_mul(tf.velo_T_cam0, inv, &velo_last); tf_mul(velo_last, vi, &velo_last);  // vloam_tf.cpp:62-63
    tf_mul(tf.base_T_cam0, inv, &base_last); tf_mul(base_last, bi, &base_last);  // vloam_tf.cpp:65
    double qb[4];
    tf_get_rotation(base_last, qb);
    const bool nan = isnan(base_last.o[0]) || isnan(base_last.o[1]) || isnan(base_last.o[2]) || isnan(qb[0]) || isnan(qb[1]) || isnan(qb[2]) || isnan(qb[3]);
    if (!nan) tf_mul(tf.world_VOT_base_last, base_last, &tf.world_VOT_base_last);  // vloam_tf.cpp:68-72
    tf_get_rotation(velo_last, lo->prior_q);                                        // what LO:225-232 reads back through getRotation() / getOrigin()
    for (int k = 0; k < 3; k++) lo->prior_t[k] = velo_last.o[k];
    tf_get_rotation(tf.world_VOT_base_last, vo_row7);
    for (int k = 0; k < 3; k++) vo_row7[4 + k] = tf.world_VOT_base_last.o[k];
  }
  if (copy_to_para) {
    for (int k = 0; k < 4; k++) lo->para_q[k] = lo->prior_q[k];
    for (int k = 0; k < 3; k++) lo->para_t[k] = lo->prior_t[k];
  }
}

// LO:477-478 pose integration (+ trajectory log row); q_w_curr is not renormalised, as in the reference.
__global__ void k_lo_finish(LOState* lo, double* traj_row14, int integrate, size_t ss) {
  if (threadIdx.x != 0) return;
  VL_SESSION(ss); RB(lo); RB(traj_row14);
  if (integrate) {
    const double* q = lo->q_w_curr;
    const double* ql = lo->para_q;
    const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
    const double vx = lo->para_t[0], vy = lo->para_t[1], vz = lo->para_t[2];
    double cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;
    cx = cx + cx; cy = cy + cy; cz = cz + cz;
    const double dx = uy * cz - uz * cy, dy = uz * cx - ux * cz, dz = ux * cy - uy * cx;
    lo->t_w_curr[0] = lo->t_w_curr[0] + ((vx + w * cx) + dx);
    lo->t_w_curr[1] = lo->t_w_curr[1] + ((vy + w * cy) + dy);
    lo->t_w_curr[2] = lo->t_w_curr[2] + ((vz + w * cz) + dz);
    double r[4];
    r[0] = q[3] * ql[0] + q[0] * ql[3] + q[1] * ql[2] - q[2] * ql[1];
    r[1] = q[3] * ql[1] + q[1] * ql[3] + q[2] * ql[0] - q[0] * ql[2];
    r[2] = q[3] * ql[2] + q[2] * ql[3] + q[0] * ql[1] - q[1] * ql[0];
    r[3] = q[3] * ql[3] - q[0] * ql[0] - q[1] * ql[1] - q[2] * ql[2];
    for (int k = 0; k < 4; k++) lo->q_w_curr[k] = r[k];
  }
  tf_lo_publish(lo, lo->para_q);  // LO:563-567 (para_q, para_t are contiguous: q_last_curr, t_last_curr)
  if (traj_row14) {
    for (int k = 0; k < 4; k++) traj_row14[k] = lo->q_w_curr[k];
    for (int k = 0; k < 3; k++) traj_row14[4 + k] = lo->t_w_curr[k];
    // mapping overwrites [7..13] when it runs; until then the map pose equals the odometry pose
    for (int k = 0; k < 7; k++) traj_row14[7 + k] = traj_row14[k];
  }
}

void lo_assoc_launch(hipStream_t st, Sess se, const float4* sharp, const float4* flat, const FrameScalars* Sc, const float4* CL, const float4* SL,
                     const FrameScalars* Sp, const LoGrid& G, const LOState* lo, const FactorTable& F, int* corr, long long* dbg_cyc,
                     int* queue, int* queue_n, int launch_no, ProfHook* ph) {
  // VLOAM_LO_ASSOC_LANES = 16: the common queries by 16-lane groups (k_lo_assoc_fast), the left-over ones by one wavefront each in a second
  // launch; 0: every query by one wavefront (the round-2 form).  Default: 16 for batches (the chip is full of short queries), 0 for one sequence
  static const int g_env = getenv("VLOAM_LO_ASSOC_LANES") ? atoi(getenv("VLOAM_LO_ASSOC_LANES")) : -1;
  const int lanes = g_env >= 0 ? g_env : (se.B > 1 ? 16 : 0);
  if (lanes == 16 && queue) {
    const int parity = launch_no & 1;
    VLOAM_LAUNCH(ph, kKLoAssocFast, st, k_lo_assoc_fast<16>, dim3(kMaxLoFactors / 16, 1, se.B), dim3(256), 0, st, sharp, flat, Sc, CL, SL, Sp, G, lo, F, corr, dbg_cyc,
                 queue, queue_n, parity, se.ss);
    VLOAM_LAUNCH(ph, kKLoAssoc, st, k_lo_assoc_dense, dim3((kMaxLoFactors + 3) / 4, 1, se.B), dim3(256), 0, st, sharp, flat, Sc, CL, SL, Sp, G, lo, F, corr, dbg_cyc,
                 (const int*)queue, queue_n, parity, se.ss);
  } else {
    VLOAM_LAUNCH(ph, kKLoAssoc, st, k_lo_assoc, dim3((kMaxLoFactors + 3) / 4, 1, se.B), dim3(256), 0, st, sharp, flat, Sc, CL, SL, Sp, G, lo, F, corr, dbg_cyc,
                 (const int*)nullptr, (int*)nullptr, 0, se.ss);
  }
}
void lo_grid_build_launch(hipStream_t st, Sess se, const float4* less_sharp, const float4* less_flat, const FrameScalars* S, const LoGrid& G, ProfHook* ph) {
  VLOAM_LAUNCH(ph, kKLoGridCount, st, k_lo_grid_count, dim3(64, 2, se.B), dim3(256), 0, st, less_sharp, less_flat, S, G, se.ss);
  if (se.B > 1) VLOAM_LAUNCH(ph, kKLoGridScan, st, k_lo_grid_scan<256>, dim3(4, 1, se.B), dim3(256), 0, st, G, se.ss);
  else VLOAM_LAUNCH(ph, kKLoGridScan, st, k_lo_grid_scan<1024>, dim3(4, 1, se.B), dim3(1024), 0, st, G, se.ss);
  VLOAM_LAUNCH(ph, kKLoGridScatter, st, k_lo_grid_scatter, dim3(64, 2, se.B), dim3(256), 0, st, less_sharp, less_flat, S, G, se.ss);
}
void lo_set_prior_launch(hipStream_t st, Sess se, LOState* lo, bool copy_to_para, const double* vo_x, bool vo_solved, double* vo_row7, int* err) {
  VL_RAW_LAUNCH(k_lo_set_prior, dim3(1, 1, se.B), dim3(64), 0, st, lo, copy_to_para ? 1 : 0, vo_x, vo_solved ? 1 : 0, vo_row7, err, se.ss);
}
void lo_finish_launch(hipStream_t st, Sess se, LOState* lo, double* traj_row14, bool integrate, ProfHook* ph) {
  VLOAM_LAUNCH(ph, kKLoFinish, st, k_lo_finish, dim3(1, 1, se.B), dim3(64), 0, st, lo, traj_row14, integrate ? 1 : 0, se.ss);
}

}  // namespace vloam
