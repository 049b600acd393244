// laserMapping on gfx950: scan-to-map ICP against a persistent voxel hash.
// Restates LaserMapping::input / solveMapping, /root/reference/src/lidar_odometry_mapping/src/laser_mapping.cpp:167-708
// ("LM:<line>").  One sweep = 11 launches incl. 2 Levenberg–Marquardt solves (which compact on their own), no host synchronisation; the two
// k_map_ds_* launches only need the sweep's feature clouds and run on a stream of their own.  Every kernel carries the session
// index of a batched handle in blockIdx.z and rebases its pointer arguments by blockIdx.z * ss (vloam_device.h).
//   k_map_prepare   1 WG      initial guess (LM:193-194), centre cube + grid roll (LM:207-402), gate (LM:448); stops taking sweeps
//                             when the voxel table is full
//   k_map_ds_bin    grid      pcl::VoxelGrid of the scan features (LM:432-440) without a global hash, pass 1: PCL's cell index from the cloud's
//                             bounding box (incl. its index-overflow guard: output = input), sample splitters, (cell << 24 | point) keys binned
//   k_map_ds_reduce 64 WGs    pass 2: bins drawn by ticket, sorted in LDS, output slots by a decoupled look-back over the bins' cell counts,
//                             per cell the input-ordered f32 centroid, output in PCL's order (cells ascending)
//   k_map_assoc     1 wave/pt pointAssociateToMap, exact 5-NN through the block-occupancy index of the voxel hash (one 32-byte
//                             record per candidate); the second outer round re-ranks the first round's candidates             x2
//   k_map_fit       1 thread/pt 3x3 eigen / 5x3 least squares, emission of LidarEdgeFactor / LidarPlaneNormFactor (LM:472-581) x2
//   (k_lm_solve)                                                                                                x2
//   k_map_insert    grid      transformUpdate (LM:140-144,636) + trajectory row; scan voxels -> map frame -> cube -> hash
//                             find-or-insert, queue on the voxel (LM:639-683)
//   k_map_finalize  grid      per touched voxel: stack-ordered f32 accumulation == per-cube VoxelGrid re-filter (LM:689-702);
//                             raw voxels of cubes that became valid; after a grid roll drops the voxels whose cube left the
//                             21x21x11 window (tombstones); table health -> host-mapped rebuild flag
//   k_map_rebuild_* grid      (rare, between sweeps) gather live records, clear, reinsert: tombstone reclamation
//   k_map_export    grid      /laser_cloud_map (LM:778-793) for vloam_get_map
#include <hip/hip_runtime.h>
#include <float.h>
#include <limits.h>
#include <math.h>
#include <string.h>
#include <algorithm>
#include <type_traits>
#include "lm_solve.h"
#include "map_kernels.h"
#include "subwave.h"
#include "bitonic.h"

namespace vloam {

typedef unsigned long long u64;

// ---------------------------------------------------------------------------------------------- helpers
__device__ __forceinline__ u64 mix64(u64 x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

// LM:207-216 / LM:643-652: C int() truncation plus the "< 0" correction; absolute cube coordinate (no centre offset)
__device__ __forceinline__ int cube_abs(double v) {
  int c = int((v + 25.0) / 50.0);
  if (v + 25.0 < 0) c--;
  return c;
}
__device__ __forceinline__ int cube_lo(double v) { return (int)floor((v - 1e-3 + 25.0) * 0.02); }
__device__ __forceinline__ int cube_hi(double v) { return (int)floor((v + 1e-3 + 25.0) * 0.02); }
// first global voxel index that can belong to cube A (minus one cell of slack so the local index is never negative)
__device__ __forceinline__ int cube_voxel_base(int A, float inv) { return (int)floor((50.0 * (double)A - 25.0) * (double)inv) - 1; }

// Voxel key: | seq 8 (56-63) | - | cube i + 512 (10: 45-54) | cube j + 512 (10: 35-44) | cube k + 128 (8: 27-34) | voxel lx (18-26), ly (9-17),
// lz (0-8) inside the cube, 9 bits each |.  Nine bits per axis take any leaf down to 50 m / 508 (the reference's launch files use 0.2 / 0.4 and
// 0.4 / 0.8, laser_mapping.cpp:95-101 takes any value; vloam_create's bound of 0.132 m comes from the 32-bit tie rank of k_map_assoc, not
// from the key).  seq = 0: the voxel's record (centroid, or the running sum of a raw voxel); seq = 1..255: one RAW point of a voxel whose
// cube lies outside the valid 5 x 5 x 3 block (see k_map_finalize) — the reference keeps such points un-merged in their cube until the
// cube is next re-filtered, and its kd-tree sees them one by one.  Cubes are absolute (no window offset): +-25.6 km horizontally,
// +-6.4 km vertically around the start (k_map_insert reports anything beyond).
constexpr int kCubeOffXY = 512, kCubeOffZ = 128;
constexpr int kVoxBits = 9, kVoxMax = (1 << kVoxBits) - 1;
__device__ __forceinline__ u64 pack_key(int Ai, int Aj, int Ak, int lx, int ly, int lz) {
  return ((u64)(unsigned)(Ai + kCubeOffXY) << 45) | ((u64)(unsigned)(Aj + kCubeOffXY) << 35) | ((u64)(unsigned)(Ak + kCubeOffZ) << 27) |
         ((u64)(unsigned)lx << (2 * kVoxBits)) | ((u64)(unsigned)ly << kVoxBits) | (u64)(unsigned)lz;
}
__device__ __forceinline__ void unpack_cube(u64 k, int* Ai, int* Aj, int* Ak) {
  *Ai = (int)((k >> 45) & 0x3ff) - kCubeOffXY; *Aj = (int)((k >> 35) & 0x3ff) - kCubeOffXY; *Ak = (int)((k >> 27) & 0xff) - kCubeOffZ;
}
__device__ __forceinline__ int key_lx(u64 k) { return (int)((k >> (2 * kVoxBits)) & kVoxMax); }
__device__ __forceinline__ int key_ly(u64 k) { return (int)((k >> kVoxBits) & kVoxMax); }
__device__ __forceinline__ int key_lz(u64 k) { return (int)(k & kVoxMax); }
__device__ __forceinline__ bool cube_in_key_range(int Ai, int Aj, int Ak) {
  return Ai >= -kCubeOffXY && Ai < kCubeOffXY && Aj >= -kCubeOffXY && Aj < kCubeOffXY && Ak >= -kCubeOffZ && Ak < kCubeOffZ;
}
__device__ __forceinline__ int key_seq(u64 k) { return (int)(k >> 56); }
__device__ __forceinline__ u64 key_with_seq(u64 k, int seq) { return (k & 0x00ffffffffffffffull) | ((u64)(unsigned)seq << 56); }
// VoxelRec::count of a seq-0 record: points in the sum (low 16 bits) | kRecRaw when the voxel holds raw points (its cube was outside the
// valid block when they arrived): then records seq = 1..n hold the points themselves
constexpr int kRecRaw = 1 << 30;
__device__ __forceinline__ int rec_n(int count) { return count & 0xffff; }
__device__ __forceinline__ bool rec_raw(int count) { return (count & kRecRaw) != 0; }

// A voxel record as two 16-byte loads of one 32-byte line
struct RecVal { u64 key; float4 sum; int count, pend_cnt; };
__device__ __forceinline__ RecVal rec_load(const VoxelRec* r) {
  const uint4 a = reinterpret_cast<const uint4*>(r)[0], b = reinterpret_cast<const uint4*>(r)[1];
  RecVal v;
  v.key = (u64)a.x | ((u64)a.y << 32);
  v.sum = make_float4(__uint_as_float(a.z), __uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y));
  v.count = (int)b.z; v.pend_cnt = (int)b.w;
  return v;
}
// everything but the key (which only find-or-insert writes)
__device__ __forceinline__ void rec_store_value(VoxelRec* r, float4 sum, int count, int pend_cnt) {
  reinterpret_cast<float2*>(r)[1] = make_float2(sum.x, sum.y);
  reinterpret_cast<uint4*>(r)[1] = make_uint4(__float_as_uint(sum.z), __float_as_uint(sum.w), (unsigned)count, (unsigned)pend_cnt);
}

// Eigen q * v (see lm_solve.hip / lo_kernels.hip) followed by + t, rounded to f32: pointAssociateToMap, LM:146-155
__device__ __forceinline__ float4 associate_to_map(float4 pi, const double* q, const double* t) {
  const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
  const double vx = pi.x, vy = pi.y, vz = pi.z;
  double cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;
  cx = cx + cx; cy = cy + cy; cz = cz + cz;
  const double dx = uy * cz - uz * cy, dy = uz * cx - ux * cz, dz = ux * cy - uy * cx;
  float4 o;
  o.x = (float)(((vx + w * cx) + dx) + t[0]);
  o.y = (float)(((vy + w * cy) + dy) + t[1]);
  o.z = (float)(((vz + w * cz) + dz) + t[2]);
  o.w = pi.w;
  return o;
}

__device__ __forceinline__ void dquat_mul(const double* a, const double* b, double* r) {
  r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
__device__ __forceinline__ void dquat_rot(const double* q, const double* v, double* o) {
  const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
  double cx = uy * v[2] - uz * v[1], cy = uz * v[0] - ux * v[2], cz = ux * v[1] - uy * v[0];
  cx = cx + cx; cy = cy + cy; cz = cz + cz;
  const double dx = uy * cz - uz * cy, dy = uz * cx - ux * cz, dz = ux * cy - uy * cx;
  o[0] = (v[0] + w * cx) + dx; o[1] = (v[1] + w * cy) + dy; o[2] = (v[2] + w * cz) + dz;
}

// ---------------------------------------------------------------------------------------------- prepare
__global__ __launch_bounds__(256) void k_map_prepare(MapState* ms, MapFrame* fr, const LOState* lo, int* cube_cnt, int skip_frame,
                                                     double* traj_row14, StackInfo* si, int* deferred0, int* deferred1, const int* newraw0,
                                                     const int* newraw1, long long* ts_log, size_t ss) {
  VL_SESSION(ss); RB(ms); RB(fr); RB(lo); RB(cube_cnt); RB(traj_row14); RB(si); RB(deferred0); RB(deferred1); RB(newraw0); RB(newraw1); RB(ts_log);
  const long long ts_begin = ts_log ? (long long)wall_clock64() : 0;
  __shared__ int shift[3], s_cen[3];
  const int tid = threadIdx.x;
  // Every load that does not depend on another one is issued up front (one memory round trip for the lot): this launch is a single
  // workgroup at the head of the stream that bounds the throughput, and each dependent trip costs ~1 us there.
  const int nn0 = min(fr->n_newraw[0], kStackCapCorner), nn1 = min(fr->n_newraw[1], kStackCapSurf);
  const int nd0 = fr->n_deferred[0], nd1 = fr->n_deferred[1];
  double row[7], qmw[4], tmw[3];
  int cen[3] = {0, 0, 0}, nst[2] = {0, 0}, si_err = 0, fr_err = 0, sweep_no = 0;
  if (tid == 0) {
    // the odometry pose of THIS sweep as k_lo_finish logged it (the live LOState may already belong to the next sweep: the
    // odometry stream runs ahead of the mapping stream)
    (void)lo;
    for (int k = 0; k < 7; k++) row[k] = traj_row14[k];
    for (int k = 0; k < 4; k++) qmw[k] = ms->q_wmap_wodom[k];
    for (int k = 0; k < 3; k++) tmw[k] = ms->t_wmap_wodom[k];
    cen[0] = ms->cenW; cen[1] = ms->cenH; cen[2] = ms->cenD;
    nst[0] = si->n_stack[0]; nst[1] = si->n_stack[1];
    si_err = si->error; fr_err = fr->error; sweep_no = ms->sweep_no;
  }
  // voxels that turned raw in the previous sweep join the list of raw voxels (k_map_finalize could not append to the list it compacts)
  if (nn0 > 0) for (int e = tid; e < nn0; e += 256) { if (nd0 + e < kStackCapCorner) deferred0[nd0 + e] = newraw0[e]; }
  if (nn1 > 0) for (int e = tid; e < nn1; e += 256) { if (nd1 + e < kStackCapSurf) deferred1[nd1 + e] = newraw1[e]; }
  // every wavefront holds its copy of the four counters (and has issued its share of the merge) before thread 0 rewrites them below:
  // the barrier's fence completes the loads above, so a wavefront that starts late can neither see the zeroed n_newraw nor the bumped n_deferred
  __syncthreads();
  if (tid == 0) {
    if (nd0 + nn0 > kStackCapCorner || nd1 + nn1 > kStackCapSurf) { atomicOr(&fr->error, kErrMapFull); fr_err |= kErrMapFull; }
    if (nn0 | nn1) {
      fr->n_deferred[0] = min(nd0 + nn0, kStackCapCorner); fr->n_deferred[1] = min(nd1 + nn1, kStackCapSurf);
      fr->n_newraw[0] = 0; fr->n_newraw[1] = 0;
    }
    // LaserMapping::input LM:182-195: q_w_curr = q_wmap_wodom * q_wodom_curr, t_w_curr = q_wmap_wodom * t_wodom_curr + t_wmap_wodom
    for (int k = 0; k < 4; k++) ms->q_wodom_curr[k] = row[k];
    for (int k = 0; k < 3; k++) ms->t_wodom_curr[k] = row[4 + k];
    double q[4], t[3];
    dquat_mul(qmw, row, q);
    dquat_rot(qmw, row + 4, t);
    for (int k = 0; k < 3; k++) t[k] = t[k] + tmw[k];
    shift[0] = shift[1] = shift[2] = 0;
    int rolled = 0;
    if (skip_frame) {  // only the high-frequency pose is produced (LM:186-190)
      if (traj_row14) { for (int k = 0; k < 4; k++) traj_row14[7 + k] = q[k]; for (int k = 0; k < 3; k++) traj_row14[11 + k] = t[k]; }
    } else {
      for (int k = 0; k < 4; k++) ms->parameters[k] = q[k];
      for (int k = 0; k < 3; k++) ms->parameters[4 + k] = t[k];
      // LM:207-216
      int cI = cube_abs(t[0]) + cen[0], cJ = cube_abs(t[1]) + cen[1], cK = cube_abs(t[2]) + cen[2];
      // LM:218-402: the six while loops only move cube pointers and the centre offsets
      while (cI < 3) { cI++; cen[0]++; shift[0]++; }
      while (cI >= kCubeW - 3) { cI--; cen[0]--; shift[0]--; }
      while (cJ < 3) { cJ++; cen[1]++; shift[1]++; }
      while (cJ >= kCubeH - 3) { cJ--; cen[1]--; shift[1]--; }
      while (cK < 3) { cK++; cen[2]++; shift[2]++; }
      while (cK >= kCubeD - 3) { cK--; cen[2]--; shift[2]--; }
      ms->centerCube[0] = cI; ms->centerCube[1] = cJ; ms->centerCube[2] = cK;
      s_cen[0] = cI; s_cen[1] = cJ; s_cen[2] = cK;
      if (shift[0] | shift[1] | shift[2]) { rolled = 1; ms->cenW = cen[0]; ms->cenH = cen[1]; ms->cenD = cen[2]; }
      // the scan features were voxelised on the scan-registration stream (k_map_ds_*): adopt this sweep's stack
      if (si_err) { atomicOr(&fr->error, si_err); fr_err |= si_err; si->error = 0; }
      if (fr_err & (kErrMapFull | kErrSolverSync)) nst[0] = nst[1] = 0;  // the map cannot take this sweep (table full), or its stack has holes (a bin of the scan-feature VoxelGrid timed out): no association, no insert; the pose stays the odometry guess (vloam_sync reports it)
      for (int k = 0; k < 2; k++) { fr->n_stack[k] = nst[k]; fr->n_touched[k] = 0; }
      ms->n_corner_stack = nst[0]; ms->n_surf_stack = nst[1];
      for (int k = 0; k < 4; k++) (&fr->n_factors[0][0])[k] = 0;
      ms->sweep_no = sweep_no + 1;
      if (ts_log) ts_log[2 * (sweep_no & 1023)] = ts_begin;
    }
    fr->rolled = rolled;
  }
  __syncthreads();
  if (skip_frame) return;
  // shift the per-cube point counters exactly like the reference shifts its cube arrays (cleared slabs -> 0)
  if (shift[0] | shift[1] | shift[2]) {
    for (int kind = 0; kind < 2; kind++) {
      int* cnt = cube_cnt + kind * kCubeNum;
      // gather-with-offset through registers: new[i][j][k] = old[i - sx][j - sy][k - sz] or 0
      int vals[(kCubeNum + 255) / 256];
      int n = 0;
      for (int c = tid; c < kCubeNum; c += 256, n++) {
        const int i = c % kCubeW, j = (c / kCubeW) % kCubeH, k = c / (kCubeW * kCubeH);
        const int si = i - shift[0], sj = j - shift[1], sk = k - shift[2];
        vals[n] = (si >= 0 && si < kCubeW && sj >= 0 && sj < kCubeH && sk >= 0 && sk < kCubeD) ? cnt[si + kCubeW * sj + kCubeW * kCubeH * sk] : 0;
      }
      __syncthreads();
      n = 0;
      for (int c = tid; c < kCubeNum; c += 256, n++) cnt[c] = vals[n];
      __syncthreads();
    }
  }
  // LM:404-430,448: points in the valid 5x5x3 block decide whether the optimisation runs
  if (tid < 64) {
    int s0 = 0, s1 = 0;
    for (int c = tid; c < 75; c += 64) {
      const int i = s_cen[0] - 2 + c / 15, j = s_cen[1] - 2 + (c / 3) % 5, k = s_cen[2] - 1 + c % 3;
      if (i >= 0 && i < kCubeW && j >= 0 && j < kCubeH && k >= 0 && k < kCubeD) {
        const int ci = i + kCubeW * j + kCubeW * kCubeH * k;
        s0 += cube_cnt[ci]; s1 += cube_cnt[kCubeNum + ci];
      }
    }
    for (int d = 32; d > 0; d >>= 1) { s0 += __shfl_xor(s0, d); s1 += __shfl_xor(s1, d); }
    if (tid == 0) {
      ms->n_map_corner = s0; ms->n_map_surf = s1;
      ms->do_optimize = (s0 > 10 && s1 > 50) ? 1 : 0;
    }
  }
}

// Drop voxels whose cube left the window (the reference clears the slab that wraps, LM:240-241 etc.).  Only after a roll.
// The window is +-500 m around the sensor while queries and inserts stay within the +-125 m valid block, so nothing of the
// sweep that rolled can touch such a voxel: the sweep's last launch (k_map_finalize) does the purge on its way out, instead
// of a launch of its own in front of the association.
__device__ void map_purge(const VoxelTable& T, const MapState* ms, int first, int stride) {
  const int cW = ms->cenW, cH = ms->cenH, cD = ms->cenD;
  int dead = 0;
  for (unsigned s = (unsigned)first; s <= T.mask; s += (unsigned)stride) {
    const RecVal v = rec_load(&T.rec[s]);
    if (v.key == 0 || v.count == 0) continue;
    int Ai, Aj, Ak;
    unpack_cube(v.key, &Ai, &Aj, &Ak);
    const int i = Ai + cW, j = Aj + cH, kk = Ak + cD;
    if (i < 0 || i >= kCubeW || j < 0 || j >= kCubeH || kk < 0 || kk >= kCubeD) { rec_store_value(&T.rec[s], make_float4(0.f, 0.f, 0.f, 0.f), 0, 0); dead++; }
  }
  if (dead) atomicAdd(&T.stats[1], dead);  // tombstones: the key stays (probe chains run through it) until the table is rebuilt
}

// ---------------------------------------------------------------------------------------------- scan VoxelGrid
// pcl::VoxelGrid<PointXYZI> (PCL 1.10 filters/impl/voxel_grid.hpp applyFilter) of laserCloudCornerLast / laserCloudSurfLast, LM:432-440:
// getMinMax3D -> the overflow guard (more than INT_MAX cells in the bounding box: a warning and output = input) -> cell index
// idx = ijk0 + ijk1 * div0 + ijk2 * div0 * div1 with ijk = floor(p * inverse_leaf) - min_b -> sort by idx -> one centroid per cell, f32
// sums in the order of the sorted index vector (std::sort leaves the order inside a cell open; canonical here and in the oracle: input order).
//
// Two launches, no global hash, no whole-chip ranking (rounds 1 - 4 hashed the points into a table, ranked the occupied cells by counting
// — u^2 compares — and folded one wavefront per cell: 80 k of the 227 k wavefront-microseconds a sweep cost at B = 8, 14 MB of traffic for
// 0.6 MB of points):
//   k_map_ds_bin     cuts the key space into P = ceil(n / 512) bins at the quantiles of a sample of the cloud (every workgroup sorts the same
//                    <= 2 048 sampled keys in LDS: same splitters everywhere, no exchange), and appends every point's sort key
//                    (cell index << 24 | point index) to its bin's region — one device-scope atomic per run of equal bins in a wavefront;
//   k_map_ds_reduce  one workgroup per bin: bitonic sort of the bin's keys in LDS, cell heads, the bin's cell count published for the bins
//                    behind it (look-back over the lower bins, in ticket order: a workgroup only ever waits for workgroups that started
//                    before it), centroids in input order, written at their final place in VoxelGrid's output order.
// A bin is a contiguous range of cell indices, so bin order == output order.  Correctness never rests on the sample: a bin that outgrows
// its region (kDsBinCap keys, e.g. thousands of points in ONE cell with a coarse leaf) spills into an overflow list and takes the slow path
// (gather, rank by counting in global memory); test_scan_voxel_bins_overflow drives it.
struct DsGrid { float mnb[3]; int div0, div1; bool overflow; };

// getMinMax3D + the guard + min_b / div_b (voxel_grid.hpp), from the per-scan-line boxes k_sr_compact left; every lane gets the result
__device__ __forceinline__ DsGrid ds_grid(const FrameScalars* __restrict__ S, int kind, float inv) {
  const int lane = threadIdx.x & 63;
  float mn[3], mx[3];
#pragma unroll
  for (int a = 0; a < 3; a++) { mn[a] = S->less_bbox[kind][lane][a]; mx[a] = S->less_bbox[kind][lane][3 + a]; }
#pragma unroll
  for (int a = 0; a < 3; a++)
    for (int d = 32; d > 0; d >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], d)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d)); }
  DsGrid g;
  const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
  g.overflow = dx * dy * dz > (long long)INT_MAX;
  int minb[3], maxb[3];
#pragma unroll
  for (int a = 0; a < 3; a++) { minb[a] = (int)floorf(mn[a] * inv); maxb[a] = (int)floorf(mx[a] * inv); g.mnb[a] = (float)minb[a]; }
  g.div0 = maxb[0] - minb[0] + 1; g.div1 = maxb[1] - minb[1] + 1;
  return g;
}
__device__ __forceinline__ u64 ds_cell(float4 p, float inv, const DsGrid& g) {
  const int i0 = (int)(floorf(p.x * inv) - g.mnb[0]), i1 = (int)(floorf(p.y * inv) - g.mnb[1]), i2 = (int)(floorf(p.z * inv) - g.mnb[2]);
  return (u64)(unsigned)i0 + (u64)(unsigned)g.div0 * ((u64)(unsigned)i1 + (u64)(unsigned)g.div1 * (u64)(unsigned)i2);
}
constexpr int kDsIdxBits = 24;   // point index in the low bits of a sort key (max_points <= 2^24, vloam_create); the cell index (< 2^33) above
__device__ __forceinline__ int ds_num_bins(int n) { const int p = (n + kDsBinTarget - 1) / kDsBinTarget; return p < 1 ? 1 : (p > kDsMaxBins ? kDsMaxBins : p); }
__device__ __forceinline__ int ds_num_samples(int P) { return P <= 32 ? 256 : (P <= 64 ? 512 : (P <= 128 ? 1024 : 2048)); }
// bin of a cell: the number of splitters <= cell (upper bound; equal cells always share a bin)
__device__ __forceinline__ int ds_bin_of(const u64* split, int P, u64 cell) {
  int lo = 0, hi = P - 1;   // answer in [0, P - 1]
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (split[mid] <= cell) lo = mid + 1; else hi = mid; }
  return lo;
}

constexpr int kDsTile = 2048;   // points a workgroup of the binning pass takes per trip
constexpr int kDsReduceGrid = 64;   // workgroups of the reduce pass per cloud (each keeps drawing bins)
__global__ __launch_bounds__(256) void k_map_ds_bin(const float4* __restrict__ corner_last, const float4* __restrict__ surf_last,
                                                    const FrameScalars* __restrict__ S, DsScratch D0, DsScratch D1, float inv0, float inv1,
                                                    float4* __restrict__ stack0, float4* __restrict__ stack1, StackInfo* fr, size_t ss) {
  VL_SESSION(ss); RB(corner_last); RB(surf_last); RB(S); D0.rebase(so_); D1.rebase(so_); RB(stack0); RB(stack1); RB(fr);
  __shared__ __attribute__((aligned(16))) u64 s_key[2048];
  __shared__ __attribute__((aligned(16))) u64 s_srt[512];
  __shared__ u64 s_split[kDsMaxBins];
  __shared__ int s_cnt[kDsMaxBins], s_base[kDsMaxBins];
  const int kind = blockIdx.y, tid = threadIdx.x;
  const DsScratch D = kind ? D1 : D0;
  const float inv = kind ? inv1 : inv0;
  const float4* pts = kind ? surf_last : corner_last;
  float4* stack = kind ? stack1 : stack0;
  const int n = kind ? S->n_less_flat : S->n_less_sharp;
  if (blockIdx.x == 0 && tid == 0) D.cursor[kDsMaxBins + 1] = 0;   // the reduce pass's ticket counter (nothing of the previous sweep draws from it any more)
  if (n <= 0) { if (blockIdx.x == 0 && tid == 0) fr->n_stack[kind] = 0; return; }
  if ((int)blockIdx.x * kDsTile >= n) return;
  const DsGrid g = ds_grid(S, kind, inv);
  if (g.overflow) {   // "Leaf size is too small for the input dataset. Integer indices would overflow." — output = *input_
    for (int t0 = blockIdx.x * kDsTile; t0 < n; t0 += gridDim.x * kDsTile)   // (only the workgroups whose first tile exists are here)
      for (int i = t0 + tid; i < min(t0 + kDsTile, n); i += 256) if (i < D.stack_cap) stack[i] = pts[i];
    if (blockIdx.x == 0 && tid == 0) { fr->n_stack[kind] = min(n, D.stack_cap); if (n > D.stack_cap) atomicOr(&fr->error, kErrStackFull); }
    return;
  }
  const int P = ds_num_bins(n);
  if (P > 1) {
    // the splitters: P - 1 quantiles of Sn evenly spaced sample points; every workgroup computes the same ones
    const int Sn = ds_num_samples(P);
    for (int j = tid; j < Sn; j += 256) s_key[j] = ds_cell(pts[(int)(((long long)j * n) / Sn)], inv, g);
    __syncthreads();
    const u64* srt = s_key;
    if (Sn <= 512) {
      // few samples: rank by counting (every lane reads every key once, broadcast LDS reads, no dependent exchange stages — a bitonic
      // network over 512 keys is ~45 dependent stages; this is the prologue of EVERY workgroup of the pass)
      for (int j = tid; j < Sn; j += 256) {
        const u64 mine = s_key[j];
        int rank = 0;
#pragma unroll 8
        for (int e = 0; e < Sn; e += 2) {   // (unrolled: eight LDS reads in flight, or every iteration waits out the LDS latency)
          const ulonglong2 k2 = *(const ulonglong2*)&s_key[e];
          rank += (k2.x < mine || (k2.x == mine && e < j)) ? 1 : 0;
          rank += (k2.y < mine || (k2.y == mine && e + 1 < j)) ? 1 : 0;
        }
        s_srt[rank] = mine;
      }
      __syncthreads();
      srt = s_srt;
    } else {
      block_bitonic_sort_u64(s_key, Sn, tid, 256);
    }
    for (int j = tid; j < P - 1; j += 256) {
      const u64 sp = srt[(int)(((long long)(j + 1) * Sn) / P)];
      s_split[j] = sp;
      if (blockIdx.x == 0) D.splitters[j] = sp;   // the reduce pass's slow path tells the bins of the overflow list by them
    }
    __syncthreads();
  }
  for (int t0 = blockIdx.x * kDsTile; t0 < n; t0 += gridDim.x * kDsTile) {
    // the tile's points are counted per bin in LDS first: ONE device-scope cursor update per (workgroup, bin) and one memory round trip for
    // all of them, instead of a dependent atomic round trip per 64 points
    constexpr int kPer = kDsTile / 256;
    __syncthreads();
    for (int b = tid; b < P; b += 256) s_cnt[b] = 0;
    __syncthreads();
    float4 p[kPer];
#pragma unroll
    for (int e = 0; e < kPer; e++) { const int i = t0 + e * 256 + tid; p[e] = i < n ? pts[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
    u64 cell[kPer];
    int bin[kPer], pos[kPer];
#pragma unroll
    for (int e = 0; e < kPer; e++) {
      const int i = t0 + e * 256 + tid;
      bin[e] = -1; pos[e] = 0; cell[e] = 0ull;
      if (i < n) {
        cell[e] = ds_cell(p[e], inv, g);
        bin[e] = P > 1 ? ds_bin_of(s_split, P, cell[e]) : 0;
        pos[e] = atomicAdd(&s_cnt[bin[e]], 1);
      }
    }
    __syncthreads();
    for (int b = tid; b < P; b += 256) { const int c = s_cnt[b]; s_base[b] = c > 0 ? atomicAdd(&D.cursor[b], c) : 0; }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kPer; e++) {
      const int i = t0 + e * 256 + tid;
      if (i < n) {
        const u64 key = (cell[e] << kDsIdxBits) | (u64)(unsigned)i;
        const int at = s_base[bin[e]] + pos[e];
        if (at < kDsBinCap) D.region[(size_t)bin[e] * kDsBinCap + at] = key;
        else { const int o = atomicAdd(&D.cursor[kDsMaxBins], 1); D.over[o] = key; }   // (o < n <= max_points: every point is written exactly once)
      }
    }
  }
}

__device__ __forceinline__ float rl(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

// heads / look-back / centroids of one bin whose keys lie sorted in `key` (LDS on the fast path, global memory on the slow one)
template <bool LDS_KEYS>
__device__ __forceinline__ void ds_reduce_sorted(const u64* key, int m, int bin, int P, const float4* __restrict__ pts, float4* __restrict__ stack,
                                                 const DsScratch& D, StackInfo* fr, int kind, unsigned gen, int* s_cnt /* [256 + 8] */, int* s_big /* [3 * kDsBigCap + 1] */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // contiguous chunks: thread t owns keys [t * per, t * per + per)
  const int per = (m + 255) / 256;
  const int c0 = min(tid * per, m), c1 = min(c0 + per, m);
  int heads = 0;
  for (int i = c0; i < c1; i++) heads += (i == 0 || (key[i] >> kDsIdxBits) != (key[i - 1] >> kDsIdxBits)) ? 1 : 0;
  // block exclusive scan of the head counts
  int inc = heads;
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  if (lane == 63) s_cnt[256 + wave] = inc;
  if (tid == 0) s_big[0] = 0;
  __syncthreads();
  int wbase = 0, u = 0;
  for (int w = 0; w < 4; w++) { const int t = s_cnt[256 + w]; wbase += w < wave ? t : 0; u += t; }
  const int rank0 = wbase + inc - heads;   // output rank (inside the bin) of this thread's first head
  // publish this bin's cell count, then add up the bins in front (look-back; every one of them started before this workgroup took its ticket)
  u64* done = D.done;
  if (tid == 0) __hip_atomic_store(&done[bin], ((u64)gen << 32) | (u64)(unsigned)u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  int before = 0;
  if (wave == 0) {
    bool bad = false;
    for (int b = lane; b < bin; b += 64) {
      u64 v = 0ull;
      int spins = 0;
      for (;;) {
        v = __hip_atomic_load(&done[b], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == gen) break;
        if (++spins > (1 << 22)) { bad = true; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      before += (int)(unsigned)(v & 0xffffffffull);
    }
    for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d);
    // (never seen: a bounded wait instead of a hang.)  The offset would come from a partial sum: this bin stores NOTHING (its centroids would
    // land on other bins' slots) and the sweep's stack is declared empty, so the mapping of this sweep associates nothing instead of
    // consuming misplaced centroids; vloam_sync reports the sticky bit
    const bool any_bad = __ballot(bad) != 0ull;
    if (any_bad && lane == 0) { atomicOr(&fr->error, kErrSolverSync); }
    if (lane == 0) s_cnt[0] = any_bad ? -1 : before;
  }
  __syncthreads();
  const bool timed_out = s_cnt[0] < 0;
  before = timed_out ? D.stack_cap : s_cnt[0];   // (every output slot then counts as beyond the stack: nothing is stored)
  // centroids: the thread that owns a head walks the cell (its points in input order: the sort key ends in the point index); cells of more
  // than kDsBigCell points are left to a whole wavefront each
  int r = rank0;
  for (int i = c0; i < c1; i++) {
    const u64 ci = key[i] >> kDsIdxBits;
    if (!(i == 0 || ci != (key[i - 1] >> kDsIdxBits))) continue;
    int e = i + 1;
    while (e < m && (key[e] >> kDsIdxBits) == ci) e++;
    const int out = before + r;
    r++;
    if (out >= D.stack_cap) continue;   // reported by the last bin
    if (e - i > kDsBigCell) {
      const int q = atomicAdd(&s_big[0], 1);
      if (q < kDsBigCap) { s_big[1 + 3 * q] = i; s_big[2 + 3 * q] = e; s_big[3 + 3 * q] = out; }
      else {   // (more big cells than the list holds: this thread folds it after all)
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        for (int j = i; j < e; j++) { const float4 p = pts[(int)(key[j] & ((1ull << kDsIdxBits) - 1ull))]; sx += p.x; sy += p.y; sz += p.z; si += p.w; }
        const float nn = (float)(e - i);
        stack[out] = make_float4(sx / nn, sy / nn, sz / nn, si / nn);
      }
      continue;
    }
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    for (int j = i; j < e; j += 4) {   // four loads in flight, added in order
      float4 p[4];
#pragma unroll
      for (int q = 0; q < 4; q++) if (j + q < e) p[q] = pts[(int)(key[j + q] & ((1ull << kDsIdxBits) - 1ull))];
#pragma unroll
      for (int q = 0; q < 4; q++) if (j + q < e) { sx += p[q].x; sy += p[q].y; sz += p[q].z; si += p[q].w; }
    }
    const float nn = (float)(e - i);
    stack[out] = make_float4(sx / nn, sy / nn, sz / nn, si / nn);
  }
  __syncthreads();
  const int nbig = min(s_big[0], kDsBigCap);
  for (int q = wave; q < nbig; q += 4) {   // a wavefront per big cell: 64 points fetched at once, folded one by one (f32, CentroidPoint's order)
    const int i = s_big[1 + 3 * q], e = s_big[2 + 3 * q], out = s_big[3 + 3 * q];
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    for (int j0 = i; j0 < e; j0 += 64) {
      const int mm = min(64, e - j0);
      const float4 p = lane < mm ? pts[(int)(key[j0 + lane] & ((1ull << kDsIdxBits) - 1ull))] : make_float4(0.f, 0.f, 0.f, 0.f);
      for (int t = 0; t < mm; t++) { sx += rl(p.x, t); sy += rl(p.y, t); sz += rl(p.z, t); si += rl(p.w, t); }
    }
    if (lane == 0) { const float nn = (float)(e - i); stack[out] = make_float4(sx / nn, sy / nn, sz / nn, si / nn); }
  }
  if (bin == P - 1 && tid == 0) {
    // the last bin has seen every other bin's count, i.e. every workgroup of this pass is done with the region counters, the overflow list
    // and the ticket: hand them back clean for the next sweep
    const int total = timed_out ? 0 : before + u;
    fr->n_stack[kind] = min(total, D.stack_cap);
    if (total > D.stack_cap) atomicOr(&fr->error, kErrStackFull);
    D.cursor[kDsMaxBins] = 0; D.cursor[kDsMaxBins + 2] = 0;   // (the ticket counter is reset by the next sweep's binning pass: workgroups still draw from it)
  }
}

__global__ __launch_bounds__(256) void k_map_ds_reduce(const float4* __restrict__ corner_last, const float4* __restrict__ surf_last,
                                                       const FrameScalars* __restrict__ S, DsScratch D0, DsScratch D1, float inv0, float inv1,
                                                       float4* __restrict__ stack0, float4* __restrict__ stack1, StackInfo* __restrict__ fr, unsigned gen,
                                                       size_t ss) {
  VL_SESSION(ss); RB(corner_last); RB(surf_last); RB(S); D0.rebase(so_); D1.rebase(so_); RB(stack0); RB(stack1); RB(fr);
  __shared__ __attribute__((aligned(16))) u64 s_key[kDsBinCap];
  __shared__ u64 s_split[kDsMaxBins];
  __shared__ int s_cnt[256 + 8];
  __shared__ int s_big[3 * kDsBigCap + 1];
  __shared__ int s_bin;
  const int kind = blockIdx.y, tid = threadIdx.x;
  const DsScratch D = kind ? D1 : D0;
  const float inv = kind ? inv1 : inv0;
  const float4* pts = kind ? surf_last : corner_last;
  float4* stack = kind ? stack1 : stack0;
  const int n = kind ? S->n_less_flat : S->n_less_sharp;
  if (n <= 0) return;
  const int P = ds_num_bins(n);
  if ((int)blockIdx.x >= P) return;
  if (ds_grid(S, kind, inv).overflow) return;   // the binning pass copied the cloud (voxel_grid.hpp's guard)
  // Bins are taken in ticket order, not in blockIdx order: the look-back then only waits for workgroups that are already running.  The
  // grid is a fraction of kDsMaxBins (a capacity): a workgroup keeps taking tickets until the bins are gone — workgroups that only
  // find out that there is nothing for them cost a wave slot for a memory round trip each, and on a chip full of sessions that adds up
  // (8 192 such workgroups per launch at B = 16 were 85 % of this kernel's wave-microseconds).
  for (;;) {
  __syncthreads();   // (the previous bin's LDS is no longer read)
  if (tid == 0) s_bin = atomicAdd(&D.cursor[kDsMaxBins + 1], 1);
  __syncthreads();
  const int bin = s_bin;
  if (bin >= P) {
    return;   // every ticket is out (the counter is reset by the next sweep's binning pass)
  }
  const int m_raw = D.cursor[bin];
  __syncthreads();
  if (tid == 0) D.cursor[bin] = 0;   // this bin's region counter, clean for the next sweep
  if (m_raw <= kDsBinCap) {
    int P2 = 256;
    while (P2 < m_raw) P2 <<= 1;
    const u64* reg = D.region + (size_t)bin * kDsBinCap;
    for (int i = tid; i < P2; i += 256) s_key[i] = i < m_raw ? reg[i] : ~0ull;
    __syncthreads();
    block_bitonic_sort_u64(s_key, P2, tid, 256);
    ds_reduce_sorted<true>(s_key, m_raw, bin, P, pts, stack, D, fr, kind, gen, s_cnt, s_big);
    continue;
  }
  // ---- slow path: the bin outgrew its region.  Gather region + this bin's share of the overflow list, rank by counting (keys are unique:
  // they end in the point index), continue from global memory.
  const int n_over = D.cursor[kDsMaxBins];
  for (int j = tid; j < P - 1; j += 256) s_split[j] = D.splitters[j];
  __syncthreads();
  int mine = 0;
  for (int j = tid; j < n_over; j += 256) mine += (P > 1 ? ds_bin_of(s_split, P, D.over[j] >> kDsIdxBits) : 0) == bin ? 1 : 0;
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d);
  if ((tid & 63) == 0) s_cnt[256 + (tid >> 6)] = mine;
  __syncthreads();
  const int m = kDsBinCap + s_cnt[256] + s_cnt[257] + s_cnt[258] + s_cnt[259];
  __syncthreads();
  if (tid == 0) { s_cnt[1] = atomicAdd(&D.cursor[kDsMaxBins + 2], m); s_cnt[2] = kDsBinCap; }
  __syncthreads();
  u64* raw = D.tmp + s_cnt[1];
  u64* srt = D.sorted + s_cnt[1];
  const u64* reg = D.region + (size_t)bin * kDsBinCap;
  for (int i = tid; i < kDsBinCap; i += 256) raw[i] = reg[i];
  for (int j = tid; j < n_over; j += 256) {
    const u64 k = D.over[j];
    if ((P > 1 ? ds_bin_of(s_split, P, k >> kDsIdxBits) : 0) == bin) raw[atomicAdd(&s_cnt[2], 1)] = k;
  }
  __threadfence();
  __syncthreads();
  for (int i = tid; i < m; i += 256) {
    const u64 k = raw[i];
    int rank = 0;
    for (int j = 0; j < m; j++) rank += raw[j] < k ? 1 : 0;
    srt[rank] = k;
  }
  __threadfence();
  __syncthreads();
  ds_reduce_sorted<false>(srt, m, bin, P, pts, stack, D, fr, kind, gen, s_cnt, s_big);
  }   // next ticket
}

// ---------------------------------------------------------------------------------------------- data association
__device__ __forceinline__ void lds_sync_wave() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// Largest two eigenvalues + unit eigenvector of the largest, of a symmetric 3x3 (the only outputs LM:500-506 uses of
// Eigen::SelfAdjointEigenSolver).  Closed form (trigonometric solution of the characteristic cubic; eigenvector from the
// best-conditioned cross product of two rows of A - lambda I): ~2 us of dependent f64 latency instead of ~25 us for the
// iterative sweeps the CPU oracle runs.  Agrees with the oracle's Jacobi iteration to ~1e-13 for the well-separated
// spectra of line features; the eigenvector's sign is arbitrary in both.
__device__ void sym_eig3_top(const double A[3][3], double* e_mid, double* e_max, double v[3]) {
  const double q = (A[0][0] + A[1][1] + A[2][2]) / 3.0;
  const double p1 = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
  const double d0 = A[0][0] - q, d1 = A[1][1] - q, d2 = A[2][2] - q;
  const double p2 = d0 * d0 + d1 * d1 + d2 * d2 + 2.0 * p1;
  if (!(p2 > 0.0)) { *e_mid = q; *e_max = q; v[0] = 1; v[1] = 0; v[2] = 0; return; }
  const double p = sqrt(p2 / 6.0), ip = 1.0 / p;
  const double b00 = d0 * ip, b11 = d1 * ip, b22 = d2 * ip, b01 = A[0][1] * ip, b02 = A[0][2] * ip, b12 = A[1][2] * ip;
  double r = 0.5 * (b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02));
  r = fmin(1.0, fmax(-1.0, r));
  const double phi = acos(r) / 3.0;
  const double emax = q + 2.0 * p * cos(phi);
  const double emin = q + 2.0 * p * cos(phi + 2.0943951023931954923);  // + 2 pi / 3
  *e_max = emax;
  *e_mid = 3.0 * q - emax - emin;
  const double r0[3] = {A[0][0] - emax, A[0][1], A[0][2]}, r1[3] = {A[0][1], A[1][1] - emax, A[1][2]}, r2[3] = {A[0][2], A[1][2], A[2][2] - emax};
  const double c0[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
  const double c1[3] = {r0[1] * r2[2] - r0[2] * r2[1], r0[2] * r2[0] - r0[0] * r2[2], r0[0] * r2[1] - r0[1] * r2[0]};
  const double c2[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
  const double n0 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2], n1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2],
               n2 = c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2];
  const bool u0 = n0 >= n1 && n0 >= n2, u1 = !u0 && n1 >= n2;
  const double nn = u0 ? n0 : (u1 ? n1 : n2);
  const double inv = 1.0 / sqrt(nn);
#pragma unroll
  for (int k = 0; k < 3; k++) v[k] = (u0 ? c0[k] : (u1 ? c1[k] : c2[k])) * inv;
}

// Householder least squares, identical operation order to oracle/orc_math.h householder_ls (m = 5, n = 3; LM:557)
__device__ bool householder_ls_5x3(double* A, double* b, double* x) {
  const int m = 5, n = 3;
  for (int k = 0; k < n; k++) {
    double nrm = 0;
    for (int i = k; i < m; i++) nrm += A[i * n + k] * A[i * n + k];
    nrm = sqrt(nrm);
    if (nrm == 0.0) return false;
    const double alpha = (A[k * n + k] > 0) ? -nrm : nrm;
    const double v0 = A[k * n + k] - alpha;
    double vtv = v0 * v0;
    for (int i = k + 1; i < m; i++) vtv += A[i * n + k] * A[i * n + k];
    if (vtv != 0.0) {
      for (int j = k + 1; j < n; j++) {
        double s = v0 * A[k * n + j];
        for (int i = k + 1; i < m; i++) s += A[i * n + k] * A[i * n + j];
        s = 2.0 * s / vtv;
        A[k * n + j] -= s * v0;
        for (int i = k + 1; i < m; i++) A[i * n + j] -= s * A[i * n + k];
      }
      double s = v0 * b[k];
      for (int i = k + 1; i < m; i++) s += A[i * n + k] * b[i];
      s = 2.0 * s / vtv;
      b[k] -= s * v0;
      for (int i = k + 1; i < m; i++) b[i] -= s * A[i * n + k];
    }
    A[k * n + k] = alpha;
    for (int i = k + 1; i < m; i++) A[i * n + k] = 0.0;
  }
  for (int k = n - 1; k >= 0; k--) {
    double s = b[k];
    for (int j = k + 1; j < n; j++) s -= A[k * n + j] * x[j];
    if (A[k * n + k] == 0.0) return false;
    x[k] = s / A[k * n + k];
  }
  return true;
}

// ---- k_map_assoc: the 5-NN search with G lanes per stack point (G = 16 / 32 / 64; 64 / G queries share a wavefront).
// One query's work is ~27 block probes and ~30 candidate records along a chain of three dependent memory trips: a full wavefront per
// query leaves most lanes idle most of the time and the chip runs out of wave slots, not of bandwidth (round 2: 38 % active, 5 120
// resident waves per 14 us).  Here a group of G lanes walks the same chain for its own query — piece tables by three lanes, one block
// per lane and trip, up to four candidate records per lane in flight — and the best five are taken by five group arg-min rounds over
// the (d2, tie) keys through DPP row operations (keys are unique: distinct voxels), so no candidate list is ranked in LDS any more.
// Results do not depend on G: the keys, not the visiting order, decide.
// amdgpu_waves_per_eu(5): the 16-lane form would take 108 VGPRs (4 wavefronts per SIMD); held to 96 it spills nothing and runs 5
// (B = 8: 101 -> 85 us per launch; 6 wavefronts / 80 VGPRs spills 16 values and loses again: 106 us).  The 32 / 64-lane forms use 80 anyway.
template <int G, int KB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void k_map_assoc(const float4* __restrict__ stack0, const float4* __restrict__ stack1, VoxelTable T0,
                                                   VoxelTable T1, float inv0, float inv1, const MapState* __restrict__ ms, MapFrame* fr,
                                                   float4* __restrict__ nbr, int outer, int4* __restrict__ cbox, float4* __restrict__ ccand,
                                                   long long* __restrict__ dbg_cyc /* [16] phase cycle sums + wavefront count, or null */, size_t ss) {
  VL_SESSION(ss); RB(stack0); RB(stack1); T0.rebase(so_); T1.rebase(so_); RB(ms); RB(fr); RB(nbr); RB(cbox); RB(ccand); RB(dbg_cyc);
  long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (dbg_cyc) tp[0] = clock64();
  constexpr int Q = 64 / G, QW = 4 * Q;     // queries per wavefront / per workgroup
  constexpr int U = G == 16 ? 4 : 2, CH = U * G;   // candidates per lane and pass / per query and pass (64 / 64 / 128: the typical query has ~30; four per lane in the
  // 32-lane form — one pass for the ~100 candidates of a corner query in a dense map — measured: 96 VGPRs with 13 spills, 23.3 instead of 20.3 us)
  // KB: blocks per lane and trip
  static_assert(CH <= kCandCache, "the second round's candidate cache holds kCandCache entries per slot");   // (rows of CH entries: a query's candidates sit in 1 - 2 KB, not in a 4 KB page of their own)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = lane & (G - 1), qi = wave * Q + lane / G;
  const int nc = ms->n_corner_stack, nsf = ms->n_surf_stack;
  // XCD-aware assignment as before, in units of QW queries: workgroup b runs on XCD b % 8 and every XCD takes a contiguous eighth of
  // the corner points and a contiguous eighth of the surf points (VoxelGrid output is spatially sorted: neighbours share an L2)
  const int wc = (nc + QW - 1) / QW, ws = (nsf + QW - 1) / QW, cpc = (wc + 7) >> 3, cps = (ws + 7) >> 3;
  const int bq = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
  if (bq >= cpc + cps || !ms->do_optimize) return;
  const int kind = bq < cpc ? 0 : 1;        // a workgroup serves one kind
  const int i = (kind ? xcd * cps + (bq - cpc) : xcd * cpc + bq) * QW + qi;
  const bool live = i < (kind ? nsf : nc);
  if (__ballot(live) == 0ull) return;
  const int slot = kind ? kStackCapCorner + i : i;
  const VoxelTable T = kind ? T1 : T0;
  const float inv = kind ? inv1 : inv0;
  const unsigned nv = (unsigned)vox_radix(inv);
  __shared__ u64 s_cand[QW][CH + 8];        // voxel keys of the pass, flattened per query
  __shared__ u64 s_bestk[QW][8];            // the best five so far: key ...
  __shared__ float4 s_bestp[QW][8];         // ... and centroid
  __shared__ int s_piece[QW][3][8];         // per axis: (cube, 4-voxel block, 4-bit mask) pieces of the search range
  __shared__ int s_np[QW][4];               // pieces per axis | overflow flag
  constexpr int kRawCap = 64;
  __shared__ u64 s_rawk[QW][kRawCap];       // raw voxels among the candidates (key, point count): expanded in extra passes
  __shared__ int s_rawn[QW][kRawCap];
  __shared__ int s_nraw[QW];
  float4 pointOri = make_float4(0.f, 0.f, 0.f, 0.f);
  // second outer round: the box of the first round is requested WITH the point and the pose (its address depends on the slot only).  The
  // cache entries are not: fetching the first 2 G of them up front as well was measured (A/B builds) — no change in the launch's duration
  // (10.1 us either way), 4 MB more traffic per sweep for entries beyond the lists' ends.
  int4 box0 = make_int4(0, 0, 0, 0), box1 = make_int4(0, 0, -1, 0);
  if (live) {
    pointOri = kind ? stack1[i] : stack0[i];
    if (outer > 0) { box0 = cbox[2 * slot]; box1 = cbox[2 * slot + 1]; }
  }
  const float4 sel = associate_to_map(pointOri, ms->parameters, ms->parameters + 4);  // LM:476 / LM:542
  const float q0 = sel.x, q1 = sel.y, q2 = sel.z;
  const int cen0 = ms->cenW, cen1 = ms->cenH, cen2 = ms->cenD;
  const int ctr0 = ms->centerCube[0] - cen0, ctr1 = ms->centerCube[1] - cen1, ctr2 = ms->centerCube[2] - cen2;  // absolute centre cube
  // voxel-index box that can hold a point within 1 m (pointSearchSqDis[4] < 1.0 gates everything, LM:479 / LM:547)
  // The FIRST round searches 5 cm more than the 1.001 m it needs: the second round's query — the same point under a pose that moved by
  // millimetres to centimetres — then still finds its box INSIDE the cached one and re-ranks the cached candidates (a superset of its own:
  // the five nearest within 1 m are the same).  With the exact box 15 % of the second round's queries saw a voxel boundary move and
  // searched again, and the launch lasted as long as the slowest of them.
  const float mg = outer == 0 ? 1.051f : 1.001f;
  int blo0 = (int)floorf((q0 - mg) * inv), bhi0 = (int)floorf((q0 + mg) * inv);
  int blo1 = (int)floorf((q1 - mg) * inv), bhi1 = (int)floorf((q1 + mg) * inv);
  int blo2 = (int)floorf((q2 - mg) * inv), bhi2 = (int)floorf((q2 + mg) * inv);
  // second outer round: same box as the first round => same candidates, re-ranked from the cache without a hash probe
  bool from_cache = false;
  int total = 0;
  if (outer > 0 && live) {
    const int4 b0 = box0, b1 = box1;
    from_cache = b1.z >= 0 && b1.z <= kCandCache && b0.x <= blo0 && b0.y >= bhi0 && b0.z <= blo1 && b0.w >= bhi1 && b1.x <= blo2 && b1.y >= bhi2;
    if (from_cache) total = b1.z;
  }
  const bool search = live && !from_cache;
  bool chain_too_long = false;
  if (dbg_cyc) tp[1] = clock64() + (__float_as_int(q0) & 0) + (total & 0);   // (the stamp waits for the loads above)
  if (gl < 8) s_bestk[qi][gl] = ~0ull;
  if (gl < 4) s_np[qi][gl] = 0;
  if (gl == 4) s_nraw[qi] = 0;
  sw_lds_sync();
  if (__ballot(search) != 0ull) {
    // per axis the index range [lo, hi] is cut into (cube, 4-voxel block) pieces — a voxel that straddles a 50 m cube face exists once
    // per cube — each with the 4-bit mask of its voxels inside the range; lane a of the group lists axis a
    if (search && gl < 3) {
      const int lo = gl == 0 ? blo0 : (gl == 1 ? blo1 : blo2), hi = gl == 0 ? bhi0 : (gl == 1 ? bhi1 : bhi2);
      const int ctr = gl == 0 ? ctr0 : (gl == 1 ? ctr1 : ctr2), cen = gl == 0 ? cen0 : (gl == 1 ? cen1 : cen2);
      const int halfw = gl == 2 ? 1 : 2, wdim = gl == 0 ? kCubeW : (gl == 1 ? kCubeH : kCubeD);   // valid block: 5 x 5 x 3 cubes (LM:404-420)
      const double leaf = 1.0 / (double)inv;
      const int Amin = cube_lo((double)lo * leaf), Amax = cube_hi((double)(hi + 1) * leaf);
      int n = 0;
      bool ovf = false;
      for (int A = Amin; A <= Amax; A++) {
        if (abs(A - ctr) > halfw) continue;
        const int wv = A + cen;
        if (wv < 0 || wv >= wdim) continue;
        // voxels that can hold points of cube A: from the voxel containing its lower face to the one containing its upper face
        const int base = cube_voxel_base(A, inv);
        const int ia = max(lo, base + 1), ib = min(hi, cube_voxel_base(A + 1, inv) + 1);
        if (ia > ib) continue;
        for (int blk = (ia - base) >> 2; blk <= ((ib - base) >> 2); blk++) {
          if ((unsigned)blk > (unsigned)(kVoxMax >> 2) || n >= 8) { ovf = true; break; }  // not reachable for the leaves vloam_create accepts (>= 0.132 m: <= 17 voxels = 6 blocks + a cube face)
          int m4 = 0;
#pragma unroll
          for (int t = 0; t < 4; t++) { const int iv = base + (blk << 2) + t; if (iv >= ia && iv <= ib) m4 |= 1 << t; }
          s_piece[qi][gl][n] = ((A + 8192) << 11) | (blk << 4) | m4;
          n++;
        }
      }
      s_np[qi][gl] = n;
      if (ovf) s_np[qi][3] = 1;
    }
    sw_lds_sync();
  }
  const int np0 = s_np[qi][0], np1 = s_np[qi][1], np2 = s_np[qi][2];
  const bool overflow = s_np[qi][3] != 0;
  if (dbg_cyc) tp[2] = clock64() + (np0 & 0);
  const int nblocks = search ? np0 * np1 * np2 : 0;
  // ---- the pieces every pass shares
  u64 key_u[U + 1];                       // (d2, tie) of this lane's candidates of the pass (+ one carried best)
  float px[U + 1], py[U + 1], pz[U + 1];
  int c0 = 0, ncand = 0;                  // pass window [c0, c0 + CH) of the query's candidate ordinals; candidates of this pass
  bool cache_pass = false;                // first-round pass whose candidates are left for the second round
  // phase 3: up to U voxel records per lane, two in flight per trip (the second pair is only touched when some query of the wavefront has
  // more than 2 G candidates); centroid out of the record, squared distance, key = (f32 d2 bits, position of the voxel in the
  // reference's gathered map cloud): laserCloud*FromMap concatenates the valid cubes in (i, j, k) loop order (LM:404-430) and every cube
  // cloud is VoxelGrid output, i.e. sorted by (iz, iy, ix) — so equal distances resolve to the lowest index of that cloud, the oracle's
  // canonical kNN tie rule.  A RAW voxel (points that arrived while its cube was outside the valid block, k_map_finalize) is no
  // candidate itself: it goes onto the query's raw list and its points are looked at one by one in extra passes below.
  auto probe_pair = [&](auto UB) {
    constexpr int u0 = decltype(UB)::value;
    u64 ck[2];
    unsigned sl[2];
    RecVal rv[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int w = gl + (u0 + h) * G;
      ck[h] = w < ncand ? s_cand[qi][w] : 0ull;
      sl[h] = (unsigned)mix64(ck[h]) & T.mask;
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
      rv[h].key = 0ull; rv[h].sum = make_float4(0.f, 0.f, 0.f, 0.f); rv[h].count = 0; rv[h].pend_cnt = 0;
      if (gl + (u0 + h) * G < ncand) rv[h] = rec_load(&T.rec[sl[h]]);
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int u = u0 + h;
      if (gl + u * G < ncand) {
        const u64 key = ck[h];
#pragma nounroll
        for (int probe = 0;;) {
          if (rv[h].key == 0ull) break;
          if (rv[h].key == key) {
            const int n = rec_n(rv[h].count);
            const bool raw = rec_raw(rv[h].count);
            float4 p = rv[h].sum;
            if (raw && n >= 2) {   // (a raw voxel of one point IS that point)
              const int pos = atomicAdd(&s_nraw[qi], 1);
              if (pos < kRawCap) { s_rawk[qi][pos] = key; s_rawn[qi][pos] = min(n, 255); }
            } else if (n > 0) {
              if (n > 1) { const float nn = (float)n; p.x = p.x / nn; p.y = p.y / nn; p.z = p.z / nn; }
              const float d0 = q0 - p.x, d1 = q1 - p.y, d2 = q2 - p.z;
              int Ai, Aj, Ak;
              unpack_cube(key, &Ai, &Aj, &Ak);
              // position in the gathered cloud as a mixed-radix number: cube (i, j, k loop order of the 5 x 5 x 3 block) then (lz, ly, lx);
              // every partial result stays below 2^24 until the last step (three full-rate 24-bit multiply-adds)
              const unsigned cube = (unsigned)((Ai - ctr0 + 2) * 15 + (Aj - ctr1 + 2) * 3 + (Ak - ctr2 + 1));
              unsigned tie = ((cube * nv + (unsigned)key_lz(key)) * nv + (unsigned)key_ly(key)) * nv + (unsigned)key_lx(key);
              const int seq = key_seq(key);
              if (seq) { const unsigned nv3 = nv * nv * nv; tie = cube * nv3 + (tie * 2654435761u + (unsigned)seq * 40503u) % nv3; }   // raw points of one voxel: distinct ties inside the cube
              key_u[u] = ((u64)__float_as_uint(d0 * d0 + d1 * d1 + d2 * d2) << 32) | tie;
              px[u] = p.x; py[u] = p.y; pz[u] = p.z;
            }
            break;
          }
          if (++probe == kMaxProbe) { chain_too_long = true; break; }
          sl[h] = (sl[h] + 1) & T.mask;
          rv[h] = rec_load(&T.rec[sl[h]]);
        }
        if (cache_pass)   // leave the candidate for the second round
          ccand[(size_t)slot * kCandCache + c0 + gl + u * G] = make_float4(px[u], py[u], pz[u], __uint_as_float(key_u[u] == ~0ull ? 0xffffffffu : (unsigned)key_u[u]));
      }
    }
  };
  // phase 4: five group arg-min rounds over this pass's candidates (and, with `carry`, the best five of the earlier passes); the
  // winner's lane publishes its centroid and retires the candidate
  auto select_rounds = [&](bool act, bool carry) {
    if (carry && act && gl < 5) {
      key_u[U] = s_bestk[qi][gl];
      const float4 bp = s_bestp[qi][gl];
      px[U] = bp.x; py[U] = bp.y; pz[U] = bp.z;
    }
    sw_lds_sync();
#pragma unroll
    for (int r = 0; r < 5; r++) {
      u64 m = key_u[0];
#pragma unroll
      for (int u = 1; u <= U; u++) m = key_u[u] < m ? key_u[u] : m;
      const u64 gm = grp_min_u64<G>(m);
      if (gm != ~0ull) {
#pragma unroll
        for (int u = 0; u <= U; u++)
          if (key_u[u] == gm) { s_bestk[qi][r] = gm; s_bestp[qi][r] = make_float4(px[u], py[u], pz[u], 0.f); key_u[u] = ~0ull; }
      } else if (act && gl == 0) s_bestk[qi][r] = ~0ull;
    }
    sw_lds_sync();
  };
  auto reset_candidates = [&]() {
#pragma unroll
    for (int u = 0; u <= U; u++) { key_u[u] = ~0ull; px[u] = 0.f; py[u] = 0.f; pz[u] = 0.f; }
  };
  // Candidate lists longer than CH (a dense map at a fine leaf: up to 9^3 voxels in the box) take several passes: every pass regenerates
  // the work list, keeps ordinals [c0, c0 + CH), and the best five so far compete again.
  for (c0 = 0;; c0 += CH) {
    const bool act = live && (c0 == 0 || c0 < total);
    if (__ballot(act) == 0ull) break;
    reset_candidates();
    const bool gen = act && search;
    ncand = 0;
    if (__ballot(gen) != 0ull) {
      // phase 1 + 2: one block per lane and trip fetches its occupancy mask; the existing voxels are flattened into the query's list
      int produced = 0;
      for (int bb0 = 0; __ballot(gen && bb0 < nblocks) != 0ull; bb0 += KB * G) {
        u64 occ[KB], bkey[KB];
        unsigned bs[KB];
        ulonglong2 e[KB];
        int pA[KB][3];   // cube of the block per axis
        int pb[KB][3];   // block index per axis
        int pm[KB][3];   // voxel mask per axis
        bool has[KB];
#pragma unroll
        for (int k = 0; k < KB; k++) {
          const int bb = bb0 + k * G + gl;
          has[k] = gen && bb < nblocks;
          occ[k] = 0ull; bkey[k] = 0ull; bs[k] = 0u;
#pragma unroll
          for (int a = 0; a < 3; a++) { pA[k][a] = 0; pb[k][a] = 0; pm[k][a] = 0; }
          if (has[k]) {
            // bb -> (ex, ey, ez), bb < 512 and np <= 8: quotients through an f32 reciprocal ((bb + 0.5) / n is at least 0.5 / 64 away from
            // an integer, the reciprocal is good to 1 ulp) instead of two ~35-instruction integer divisions
            const int n01 = np0 * np1;
            const int ez = (int)(((float)bb + 0.5f) * __builtin_amdgcn_rcpf((float)n01)), rem = bb - ez * n01;
            const int ey = (int)(((float)rem + 0.5f) * __builtin_amdgcn_rcpf((float)np0)), ex = rem - ey * np0;
            const int w0 = s_piece[qi][0][ex], w1 = s_piece[qi][1][ey], w2 = s_piece[qi][2][ez];
            pA[k][0] = (w0 >> 11) - 8192; pb[k][0] = (w0 >> 4) & 127; pm[k][0] = w0 & 15;
            pA[k][1] = (w1 >> 11) - 8192; pb[k][1] = (w1 >> 4) & 127; pm[k][1] = w1 & 15;
            pA[k][2] = (w2 >> 11) - 8192; pb[k][2] = (w2 >> 4) & 127; pm[k][2] = w2 & 15;
            bkey[k] = pack_key(pA[k][0], pA[k][1], pA[k][2], pb[k][0], pb[k][1], pb[k][2]) | (1ull << 63);
            bs[k] = (unsigned)mix64(bkey[k]) & T.bslots_mask;
          }
        }
#pragma unroll
        for (int k = 0; k < KB; k++) if (has[k]) e[k] = T.blk[bs[k]];   // all first probes in flight together
#pragma unroll
        for (int k = 0; k < KB; k++) {
          if (has[k]) {
#pragma nounroll
            for (int probe = 0;;) {   // (collisions are rare: keep the chain walk a compact loop)
              if (e[k].x == 0ull) break;
              if (e[k].x == bkey[k]) { occ[k] = e[k].y; break; }
              if (++probe == kMaxProbe) { chain_too_long = true; break; }
              bs[k] = (bs[k] + 1) & T.bslots_mask;
              e[k] = T.blk[bs[k]];
            }
            // voxels of the block inside the search box: bit = z * 16 + y * 4 + x
            const int mx = pm[k][0], my = pm[k][1], mz = pm[k][2];
            const u64 ex4 = (u64)mx * 0x1111111111111111ull;
            const u64 ey4 = ((u64)((my & 1) * 0xF) | ((u64)(((my >> 1) & 1) * 0xF) << 4) | ((u64)(((my >> 2) & 1) * 0xF) << 8) | ((u64)(((my >> 3) & 1) * 0xF) << 12)) * 0x0001000100010001ull;
            const u64 ez4 = ((mz & 1) ? 0xFFFFull : 0ull) | ((mz & 2) ? 0xFFFFull << 16 : 0ull) | ((mz & 4) ? 0xFFFFull << 32 : 0ull) | ((mz & 8) ? 0xFFFFull << 48 : 0ull);
            occ[k] &= ex4 & ey4 & ez4;
          }
        }
#pragma unroll
        for (int k = 0; k < KB; k++) {
          // flatten: exclusive prefix of the per-lane voxel counts inside the group, then every lane appends its voxel keys
          const int mine = __popcll(occ[k]);
          const int inc = grp_scan_incl<G>(mine);
          const int tot = grp_sum<G>(mine);
          int o = produced + inc - mine;
          u64 oc = occ[k];
          while (oc) {
            const int bit = __ffsll((long long)oc) - 1;
            oc &= oc - 1;
            if (o >= c0 && o < c0 + CH)
              s_cand[qi][o - c0] = pack_key(pA[k][0], pA[k][1], pA[k][2], (pb[k][0] << 2) | (bit & 3), (pb[k][1] << 2) | ((bit >> 2) & 3), (pb[k][2] << 2) | (bit >> 4));
            o++;
          }
          produced += tot;
        }
      }
      if (gen) total = produced;
      sw_lds_sync();
      ncand = gen ? min(total - c0, CH) : 0;
      if (dbg_cyc && c0 == 0) tp[3] = clock64() + (ncand & 0);
      cache_pass = outer == 0 && total <= kCandCache;   // (lists of up to kCandCache candidates: every pass leaves its part)
      probe_pair(std::integral_constant<int, 0>{});
      if constexpr (U > 2) { if (__ballot(2 * G < ncand) != 0ull) probe_pair(std::integral_constant<int, 2>{}); }
      cache_pass = false;
    }
    if (__ballot(from_cache && act) != 0ull) {
      // all of a lane's cache entries of the pass in flight together, complete 16-byte entries (a load that is conditional on the entry's own
      // tie field turns into two dependent trips per entry)
      float4 cc[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int w = c0 + gl + u * G;
        cc[u] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xffffffffu));
        if (from_cache && act && w < total) cc[u] = ccand[(size_t)slot * kCandCache + w];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const unsigned tie = __float_as_uint(cc[u].w);
        const float d0 = q0 - cc[u].x, d1 = q1 - cc[u].y, d2 = q2 - cc[u].z;
        const u64 k = ((u64)__float_as_uint(d0 * d0 + d1 * d1 + d2 * d2) << 32) | tie;
        if (from_cache && tie != 0xffffffffu) { key_u[u] = k; px[u] = cc[u].x; py[u] = cc[u].y; pz[u] = cc[u].z; }
      }
    }
    if (dbg_cyc && c0 == 0) tp[4] = clock64() + ((int)key_u[0] & 0);
    select_rounds(act, c0 > 0);
    if (dbg_cyc && c0 == 0) tp[5] = clock64();
  }
  // ---- raw voxels among the candidates (only with returns beyond the valid block, i.e. ranges over 100 m): their points, one record each
  // (seq 1..n), go through the same probe + arg-min passes, the best five so far carried along
  const int nraw = min(s_nraw[qi], kRawCap);
  const bool raw_overflow = s_nraw[qi] > kRawCap;
  if (__ballot(live && nraw > 0) != 0ull) {
    int total_raw = 0;
    for (c0 = 0;; c0 += CH) {
      const bool act = live && nraw > 0 && (c0 == 0 || c0 < total_raw);
      if (__ballot(act) == 0ull) break;
      reset_candidates();
      int produced = 0;
      for (int r0 = 0; __ballot(act && r0 < nraw) != 0ull; r0 += G) {
        const int r = r0 + gl;
        const int nr = (act && r < nraw) ? s_rawn[qi][r] : 0;
        const u64 vk = (act && r < nraw) ? s_rawk[qi][r] : 0ull;
        const int inc = grp_scan_incl<G>(nr);
        const int tot = grp_sum<G>(nr);
        int o = produced + inc - nr;
        for (int j = 1; j <= nr; j++, o++)
          if (o >= c0 && o < c0 + CH) s_cand[qi][o - c0] = key_with_seq(vk, j);
        produced += tot;
      }
      if (act) total_raw = produced;
      sw_lds_sync();
      ncand = act ? min(total_raw - c0, CH) : 0;
      probe_pair(std::integral_constant<int, 0>{});
      if constexpr (U > 2) { if (__ballot(2 * G < ncand) != 0ull) probe_pair(std::integral_constant<int, 2>{}); }
      select_rounds(act, true);
    }
  }
  if (outer == 0 && search && gl == 0) {
    cbox[2 * slot] = make_int4(blo0, bhi0, blo1, bhi1);
    cbox[2 * slot + 1] = make_int4(blo2, bhi2, (total <= kCandCache && !overflow && s_nraw[qi] == 0) ? total : -1, total);   // (.w: the count itself, tools/assoc_phases.py)
  }
  // hand the five neighbours to k_map_fit (one THREAD per query there: the 3x3 eigen / 5x3 least-squares fits are heavy in registers
  // and pure per-query math) — as points, so that the fit does not have to go back to the table
  if (live && gl < 5) {
    const u64 k4 = s_bestk[qi][4];
    const bool ok = k4 != ~0ull && __uint_as_float((unsigned)(k4 >> 32)) < 1.0f;  // LM:479 / LM:547
    float4 p = s_bestp[qi][gl];
    p.w = ok ? 1.0f : 0.0f;
    nbr[slot * 5 + gl] = p;
  }
  if (live && gl == 0 && total > kCandChunk) atomicMax(&fr->max_candidates, total);
  if (__ballot(chain_too_long || (live && overflow)) && lane == 0) atomicOr(&fr->error, kErrMapFull);
  if (__ballot(live && raw_overflow) && lane == 0) atomicOr(&fr->error, kErrMapDeferred);
  if (dbg_cyc && lane == 0) {   // debug builds of the handle: where a wavefront's time goes (setup | pieces | blocks + flatten | records | arg-min rounds)
    tp[6] = clock64();
    if (tp[3] == 0) tp[3] = tp[2];
    if (tp[4] == 0) tp[4] = tp[3];
    for (int k = 0; k < 6; k++) atomicAdd((unsigned long long*)&dbg_cyc[outer * 8 + k], (unsigned long long)(tp[k + 1] - tp[k]));
    atomicAdd((unsigned long long*)&dbg_cyc[outer * 8 + 7], 1ull);
  }
}

__global__ __launch_bounds__(256) void k_map_fit(const float4* __restrict__ stack0, const float4* __restrict__ stack1, VoxelTable T0,
                                                 VoxelTable T1, const MapState* __restrict__ ms, MapFrame* fr, const float4* __restrict__ nbr,
                                                 FactorTable F, int outer, size_t ss) {
  VL_SESSION(ss); RB(stack0); RB(stack1); RB(ms); RB(fr); RB(nbr); F.rebase(so_);
  const int slot = blockIdx.x * 256 + threadIdx.x;
  if (slot >= kMapFactorCap) return;
  const int kind = slot < kStackCapCorner ? 0 : 1;
  const int i = kind ? slot - kStackCapCorner : slot;
  const int nst = kind ? ms->n_surf_stack : ms->n_corner_stack;
  int type = 0;
  (void)T0; (void)T1;
  if (ms->do_optimize && i < nst && nbr[slot * 5].w != 0.0f) {
    const float4 pointOri = kind ? stack1[i] : stack0[i];
    double P[5][3];
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const float4 p = nbr[slot * 5 + j];
      P[j][0] = p.x; P[j][1] = p.y; P[j][2] = p.z;
    }
    double A3[3] = {0, 0, 0}, B3[3] = {0, 0, 0};
    if (kind == 0) {  // LM:481-517
      double center[3] = {0, 0, 0};
#pragma unroll
      for (int j = 0; j < 5; j++) for (int a = 0; a < 3; a++) center[a] = center[a] + P[j][a];
      for (int a = 0; a < 3; a++) center[a] = center[a] / 5.0;
      double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
      for (int j = 0; j < 5; j++) {
        const double z[3] = {P[j][0] - center[0], P[j][1] - center[1], P[j][2] - center[2]};
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) cov[a][b] = cov[a][b] + z[a] * z[b];
      }
      double e_mid, e_max, dir[3];
      sym_eig3_top(cov, &e_mid, &e_max, dir);
      if (e_max > 3 * e_mid) {
        for (int a = 0; a < 3; a++) { A3[a] = 0.1 * dir[a] + center[a]; B3[a] = -0.1 * dir[a] + center[a]; }
        type = 1;
      }
    } else {          // LM:545-581
      double matA0[15], matB0[5], nrm[3];
#pragma unroll
      for (int j = 0; j < 5; j++) { matA0[j * 3] = P[j][0]; matA0[j * 3 + 1] = P[j][1]; matA0[j * 3 + 2] = P[j][2]; matB0[j] = -1.0; }
      if (householder_ls_5x3(matA0, matB0, nrm)) {
        const double nn_ = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
        const double negative_OA_dot_norm = 1 / nn_;
        nrm[0] = nrm[0] / nn_; nrm[1] = nrm[1] / nn_; nrm[2] = nrm[2] / nn_;
        bool planeValid = true;
#pragma unroll
        for (int j = 0; j < 5; j++)
          if (fabs(nrm[0] * P[j][0] + nrm[1] * P[j][1] + nrm[2] * P[j][2] + negative_OA_dot_norm) > 0.2) planeValid = false;
        if (planeValid) { A3[0] = nrm[0]; A3[1] = nrm[1]; A3[2] = nrm[2]; B3[0] = negative_OA_dot_norm; type = 3; }
      }
    }
    if (type) {
      const int cap = F.cap;
      F.p[slot] = pointOri.x; F.p[cap + slot] = pointOri.y; F.p[2 * cap + slot] = pointOri.z;
      F.A[slot] = A3[0]; F.A[cap + slot] = A3[1]; F.A[2 * cap + slot] = A3[2];
      F.B[slot] = B3[0]; F.B[cap + slot] = B3[1]; F.B[2 * cap + slot] = B3[2];
      factor_digest(F, slot, type, A3, B3);   // the solve's form of the factor, ready when the solve starts (lm_solve.hip)
    }
  }
  F.type[slot] = type;
  // a wavefront == one 64-slot row of the table: its accepted slots as one mask (every row, every launch: nothing to clear); the solve
  // compacts from the masks on its own
  const unsigned long long m = __ballot(type != 0);
  if ((threadIdx.x & 63) == 0) {
    F.rowmask[slot >> 6] = m;
    if (m) atomicAdd(&fr->n_factors[outer][kind], __popcll(m));
  }
}

// ---------------------------------------------------------------------------------------------- update / insert / finalize
// transformUpdate + the map half of the trajectory row; done by one lane of k_map_insert (nothing else in that launch reads
// q_wmap_wodom / t_wmap_wodom)
__device__ void map_update(MapState* ms, double* traj_row14) {
  // transformUpdate LM:140-144: q_wmap_wodom = q_w_curr * q_wodom_curr^-1 ; t_wmap_wodom = t_w_curr - q_wmap_wodom * t_wodom_curr
  const double* q = ms->q_wodom_curr;
  const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  double qi[4] = {0, 0, 0, 0};
  if (n2 > 0.0) { qi[0] = -q[0] / n2; qi[1] = -q[1] / n2; qi[2] = -q[2] / n2; qi[3] = q[3] / n2; }
  double r[4], t[3];
  dquat_mul(ms->parameters, qi, r);
  for (int k = 0; k < 4; k++) ms->q_wmap_wodom[k] = r[k];
  dquat_rot(ms->q_wmap_wodom, ms->t_wodom_curr, t);
  for (int k = 0; k < 3; k++) ms->t_wmap_wodom[k] = ms->parameters[4 + k] - t[k];
  if (traj_row14) for (int k = 0; k < 7; k++) traj_row14[7 + k] = ms->parameters[k];
}

// set the voxel's bit in its 4 x 4 x 4 block's occupancy mask (find-or-insert of the block entry)
__device__ bool map_publish_block(const VoxelTable& T, int Ai, int Aj, int Ak, int lx, int ly, int lz) {
  const u64 bkey = pack_key(Ai, Aj, Ak, lx >> 2, ly >> 2, lz >> 2) | (1ull << 63);
  unsigned bs = (unsigned)mix64(bkey) & T.bslots_mask;
  for (int bp = 0; bp < kMaxProbe; bp++, bs = (bs + 1) & T.bslots_mask) {
    const u64 bold = atomicCAS(&T.blk[bs].x, 0ull, bkey);
    if (bold == 0ull || bold == bkey) {
      if (bold == 0ull) atomicAdd(&T.stats[2], 1);
      atomicOr(&T.blk[bs].y, 1ull << (((lz & 3) << 4) | ((ly & 3) << 2) | (lx & 3)));
      return true;
    }
  }
  return false;
}

// One raw point of a voxel as a record of its own: key | seq, sum = the point, count = 1, pend_cnt = arrival stamp (sweep << 14 | stack
// index; 0 for a centroid that became raw point number one).  find-or-insert: a purged record of the same key is reused.
__device__ bool map_put_raw_point(const VoxelTable& T, u64 voxel_key, int seq, float4 p, int stamp) {
  const u64 key = key_with_seq(voxel_key, seq);
  unsigned s = (unsigned)mix64(key) & T.mask;
  for (int probe = 0; probe < kMaxProbe; probe++, s = (s + 1) & T.mask) {
    const u64 old = atomicCAS(&T.rec[s].key, 0ull, key);
    if (old == 0ull || old == key) {
      if (old == 0ull) atomicAdd(&T.stats[0], 1);
      rec_store_value(&T.rec[s], p, 1, stamp);
      return true;
    }
  }
  return false;
}
// the cube is valid again: the raw points of the voxel are merged into its centroid and their records retire (key kept, count 0: purged)
__device__ void map_drop_raw_points(const VoxelTable& T, u64 voxel_key, int n) {
  int dead = 0;
  for (int seq = 1; seq <= min(n, 255); seq++) {
    const u64 key = key_with_seq(voxel_key, seq);
    unsigned s = (unsigned)mix64(key) & T.mask;
    for (int probe = 0; probe < kMaxProbe; probe++, s = (s + 1) & T.mask) {
      const u64 k = T.rec[s].key;
      if (k == 0ull) break;
      if (k == key) { rec_store_value(&T.rec[s], make_float4(0.f, 0.f, 0.f, 0.f), 0, 0); dead++; break; }
    }
  }
  if (dead) atomicAdd(&T.stats[1], dead);
}

__global__ __launch_bounds__(256) void k_map_insert(const float4* __restrict__ stack0, const float4* __restrict__ stack1,
                                                    float4* __restrict__ smap0, float4* __restrict__ smap1, VoxelTable T0, VoxelTable T1,
                                                    float inv0, float inv1, MapState* ms, MapFrame* fr, int* __restrict__ touched0,
                                                    int* __restrict__ touched1, int* __restrict__ deferred0, int* __restrict__ deferred1,
                                                    double* traj_row14, size_t ss) {
  VL_SESSION(ss); RB(stack0); RB(stack1); RB(smap0); RB(smap1); T0.rebase(so_); T1.rebase(so_); RB(ms); RB(fr); RB(touched0); RB(touched1);
  RB(deferred0); RB(deferred1); RB(traj_row14);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) map_update(ms, traj_row14);  // LM:636
  const int kind = blockIdx.y;
  const VoxelTable T = kind ? T1 : T0;
  const float inv = kind ? inv1 : inv0;
  const float4* stack = kind ? stack1 : stack0;
  float4* smap = kind ? smap1 : smap0;
  int* touched = kind ? touched1 : touched0;
  (void)deferred0; (void)deferred1;
  const int n = kind ? ms->n_surf_stack : ms->n_corner_stack;
  const int cap = kind ? kStackCapSurf : kStackCapCorner;
  const int lane = threadIdx.x & 63;
  // the thread's first stack point is requested together with the stack size, not behind it (the stale tail of the array is valid memory)
  const int i_first = blockIdx.x * 256 + threadIdx.x;
  const float4 p_first = stack[i_first < cap ? i_first : 0];
  for (int i0 = blockIdx.x * 256 + (threadIdx.x & ~63); i0 < n; i0 += gridDim.x * 256) {   // wavefront-uniform
    const int i = i0 + lane;
    // returns the slot if this point is the first of the sweep in its voxel (the voxel joins the touched list), else -1
    auto insert_point = [&]() -> int {
      if (i >= n) return -1;
      const float4 p = associate_to_map(i == i_first ? p_first : stack[i], ms->parameters, ms->parameters + 4);  // LM:641 / LM:664
      smap[i] = p;
      const int Ai = cube_abs((double)p.x), Aj = cube_abs((double)p.y), Ak = cube_abs((double)p.z);
      const int wi = Ai + ms->cenW, wj = Aj + ms->cenH, wk = Ak + ms->cenD;  // == cubeI, cubeJ, cubeK of LM:643-652
      if (wi < 0 || wi >= kCubeW || wj < 0 || wj >= kCubeH || wk < 0 || wk >= kCubeD) return -1;  // LM:654-655: outside the grid -> dropped
      const int lx = (int)floorf(p.x * inv) - cube_voxel_base(Ai, inv), ly = (int)floorf(p.y * inv) - cube_voxel_base(Aj, inv),
                lz = (int)floorf(p.z * inv) - cube_voxel_base(Ak, inv);
      if ((unsigned)lx > (unsigned)kVoxMax || (unsigned)ly > (unsigned)kVoxMax || (unsigned)lz > (unsigned)kVoxMax || !cube_in_key_range(Ai, Aj, Ak)) { atomicOr(&fr->error, kErrMapFull); return -1; }
      const u64 key = pack_key(Ai, Aj, Ak, lx, ly, lz);
      unsigned s = (unsigned)mix64(key) & T.mask;
      for (int probe = 0; probe < kMaxProbe; probe++, s = (s + 1) & T.mask) {
        const u64 old = atomicCAS(&T.rec[s].key, 0ull, key);
        if (old == 0ull || old == key) {
          if (old == 0ull) {  // new voxel: publish it in its block's occupancy mask
            atomicAdd(&T.stats[0], 1);
            if (!map_publish_block(T, Ai, Aj, Ak, lx, ly, lz)) atomicOr(&fr->error, kErrMapFull);
          }
          const int pos = atomicAdd(&T.rec[s].pend_cnt, 1);
          if (pos < kPendCap) T.pend[(size_t)s * kPendCap + pos] = i; else atomicOr(&fr->error, kErrMapFull);
          return pos == 0 ? (int)s : -1;   // (whether the cube is inside the valid block is k_map_finalize's business: raw points, see there)
        }
      }
      atomicOr(&fr->error, kErrMapFull);
      return -1;
    };
    const int first = insert_point();
    // touched list: one counter update per wavefront instead of one per voxel on the same word
    const u64 tm = __ballot(first >= 0);
    if (tm != 0ull) {
      const int leader = __ffsll((long long)tm) - 1;
      int base = 0;
      if (lane == leader) base = atomicAdd(&fr->n_touched[kind], __popcll(tm));
      base = __shfl(base, leader);
      if (first >= 0) {
        const int tpos = base + __popcll(tm & ((1ull << lane) - 1ull));
        if (tpos < cap) touched[tpos] = first; else atomicOr(&fr->error, kErrMapFull);
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_map_finalize(const float4* __restrict__ smap0, const float4* __restrict__ smap1, VoxelTable T0,
                                                      VoxelTable T1, MapState* ms, MapFrame* fr,
                                                      const int* __restrict__ touched0, const int* __restrict__ touched1,
                                                      int* __restrict__ deferred0, int* __restrict__ deferred1, int* __restrict__ newraw0,
                                                      int* __restrict__ newraw1, int* __restrict__ cube_cnt, int* host_flags, long long* ts_log, size_t ss) {
  VL_SESSION(ss); RB(smap0); RB(smap1); T0.rebase(so_); T1.rebase(so_); RB(ms); RB(fr); RB(touched0); RB(touched1); RB(deferred0); RB(deferred1);
  RB(newraw0); RB(newraw1); RB(cube_cnt); RB(ts_log);
  if (ts_log && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ts_log[2 * ((ms->sweep_no - 1) & 1023) + 1] = (long long)wall_clock64();   // START of the sweep's last launch
  if (host_flags) host_flags += 2 * blockIdx.z;   // host-mapped, one pair per session (not part of the arenas)
  const int kind = blockIdx.y;
  const VoxelTable T = kind ? T1 : T0;
  const float4* smap = kind ? smap1 : smap0;
  const int* touched = kind ? touched1 : touched0;
  const int cap = kind ? kStackCapSurf : kStackCapCorner;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int s_spec = touched[t < cap ? t : 0];   // requested together with the list's length, not behind it (stale entries are valid slots)
  const int nt = min(fr->n_touched[kind], cap);
  const int cI = ms->centerCube[0], cJ = ms->centerCube[1], cK = ms->centerCube[2];
  int* deferred = kind ? deferred1 : deferred0;
  int* newraw = kind ? newraw1 : newraw0;
  if (t < nt) {
    const int s = s_spec;
    int idx[kPendCap];
    const RecVal rv = rec_load(&T.rec[s]);
    const int np = min(rv.pend_cnt, kPendCap);
    for (int a = 0; a < np; a++) idx[a] = T.pend[(size_t)s * kPendCap + a];
    for (int a = 1; a < np; a++) {  // stack order == the order the reference push_back()s into the cube cloud
      const int v = idx[a];
      int c = a - 1;
      while (c >= 0 && idx[c] > v) { idx[c + 1] = idx[c]; c--; }
      idx[c + 1] = v;
    }
    const int n_old = rec_n(rv.count);
    const bool was_raw = rec_raw(rv.count);
    float4 acc = n_old > 0 ? rv.sum : make_float4(0.f, 0.f, 0.f, 0.f);
    int Ai, Aj, Ak;
    unpack_cube(rv.key, &Ai, &Aj, &Ak);
    const int wi = Ai + ms->cenW, wj = Aj + ms->cenH, wk = Ak + ms->cenD;
    int* cube_n = &cube_cnt[kind * kCubeNum + wi + kCubeW * wj + kCubeW * kCubeH * wk];   // points of the cube's cloud (the LM:448 gate counts them)
    const bool valid = abs(wi - cI) <= 2 && abs(wj - cJ) <= 2 && abs(wk - cK) <= 1;
    if (valid) {
      // VoxelGrid re-filter of a valid cube (LM:689-702): centroid of (what the cube held of this voxel — its centroid, or raw points in
      // arrival order —, new points...): the running f32 sum already is that sum in that order
      if (was_raw) map_drop_raw_points(T, rv.key, n_old);
      for (int a = 0; a < np; a++) { const float4 p = smap[idx[a]]; acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
      const float nn = (float)(n_old + np);
      acc.x = acc.x / nn; acc.y = acc.y / nn; acc.z = acc.z / nn; acc.w = acc.w / nn;
      const int held = was_raw ? n_old : (n_old > 0 ? 1 : 0);
      if (held != 1) atomicAdd(cube_n, 1 - held);
      rec_store_value(&T.rec[s], acc, 1, 0);
    } else {
      // The cube is outside the valid block: the reference appends the points to the cube's cloud and leaves them there un-merged until
      // the cube is next valid (LM:654-659 push_back, LM:689 only filters valid cubes) — its kd-tree then sees them one by one.  The
      // seq-0 record keeps the running sum (the eventual centroid), every point also gets a record of its own (seq 1..n: k_map_assoc
      // expands a raw voxel into them, vloam_get_map emits them); an existing centroid becomes raw point number one.
      int seq = n_old;
      bool overflow = false;
      if (!was_raw) {
        if (n_old == 1 && !map_put_raw_point(T, rv.key, 1, rv.sum, 0)) overflow = true;   // stamp 0: belongs to the filtered (voxel-ordered) part
        // (onto a list of its own: workgroup 0 compacts the raw-voxel list in this very launch; k_map_prepare of the next sweep merges)
        const int dpos = atomicAdd(&fr->n_newraw[kind], 1);
        if (dpos < cap) newraw[dpos] = s; else atomicOr(&fr->error, kErrMapFull);   // the list of raw voxels is full: say so
        atomicAdd(&ms->deferred, 1);
      }
      for (int a = 0; a < np; a++) {
        const float4 p = smap[idx[a]];
        acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        seq++;
        if (seq > 255 || !map_put_raw_point(T, rv.key, seq, p, ((ms->sweep_no & 0x3ffff) << 14) | idx[a])) overflow = true;
      }
      if (overflow) atomicOr(&fr->error, kErrMapDeferred);   // more than 255 raw points in one voxel, or no slot for one
      atomicAdd(cube_n, np);
      rec_store_value(&T.rec[s], acc, min(seq, 0xffff) | kRecRaw, 0);
    }
  }
  // Raw voxels of earlier sweeps whose cube is valid now (only with ranges beyond the 5x5x3 block): the reference's re-filter of the
  // valid cubes (LM:689-702) merges them in the first sweep their cube is valid, touched or not.  One workgroup walks the list,
  // resolves what became valid and COMPACTS the rest in place (chunks of 256, left to right: an entry only ever moves left), so the
  // list holds the currently-raw voxels only.  A slot that also received points this sweep (pend_cnt > 0, or no longer raw) is
  // finalised by its own thread above — with the validity rule applied to the whole sum — and only leaves the list here.
  if (blockIdx.x == 0) {
    __shared__ int s_wcnt[4], s_out;
    const int nd = min(fr->n_deferred[kind], cap);
    if (threadIdx.x == 0) s_out = 0;
    __syncthreads();
    for (int base = 0; base < nd; base += 256) {
      const int d = base + threadIdx.x;
      int s = d < nd ? deferred[d] : -1;
      if (s >= 0) {
        const RecVal dv = rec_load(&T.rec[s]);
        int Ai, Aj, Ak;
        unpack_cube(dv.key, &Ai, &Aj, &Ak);
        const int wi = Ai + ms->cenW, wj = Aj + ms->cenH, wk = Ak + ms->cenD;
        const bool in_window = wi >= 0 && wi < kCubeW && wj >= 0 && wj < kCubeH && wk >= 0 && wk < kCubeD;
        if (!in_window || dv.count == 0) s = -1;   // purged with its cube
        else if (abs(wi - cI) <= 2 && abs(wj - cJ) <= 2 && abs(wk - cK) <= 1) {
          if (dv.pend_cnt == 0 && rec_raw(dv.count)) {
            const int n = rec_n(dv.count);
            map_drop_raw_points(T, dv.key, n);
            float4 a = dv.sum; const float nn = (float)n;
            a.x /= nn; a.y /= nn; a.z /= nn; a.w /= nn;
            rec_store_value(&T.rec[s], a, 1, 0);
            if (n != 1) atomicAdd(&cube_cnt[kind * kCubeNum + wi + kCubeW * wj + kCubeW * kCubeH * wk], 1 - n);
          }
          s = -1;
        }
      }
      const unsigned long long keep = __ballot(s >= 0);
      if ((threadIdx.x & 63) == 0) s_wcnt[threadIdx.x >> 6] = __popcll(keep);
      __syncthreads();   // every entry of the chunk has been read
      int o = s_out;
      for (int w = 0; w < (int)(threadIdx.x >> 6); w++) o += s_wcnt[w];
      if (s >= 0) deferred[o + __popcll(keep & ((1ull << (threadIdx.x & 63)) - 1ull))] = s;
      __syncthreads();
      if (threadIdx.x == 0) s_out += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
      __syncthreads();
    }
    if (threadIdx.x == 0) fr->n_deferred[kind] = s_out;
  }
  if (fr->rolled) map_purge(T, ms, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
  // table health, once per sweep and kind: the error surfaces well before probe chains degrade, and the host learns (through a
  // host-mapped word it polls without synchronising) when enough purged entries have piled up for a rebuild to pay
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const long long slots = (long long)T.mask + 1;
    const int keys = T.stats[0], dead = T.stats[1];
    if ((long long)(keys - dead) * 10 > slots * 6 || (long long)T.stats[2] * 10 > ((long long)T.bslots_mask + 1) * 6) atomicOr(&fr->error, kErrMapFull);
    if (host_flags) {
      const int want = ((long long)dead * 8 > slots || (long long)keys * 2 > slots) && dead > 0 ? 1 : 0;
      __hip_atomic_store(&host_flags[kind], want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---------------------------------------------------------------------------------------------- table rebuild (tombstone reclamation)
// Purged entries keep their key so that probe chains stay intact; on a long drive they would fill the table.  When the host sees
// the flag k_map_finalize raises, it enqueues — between two sweeps, on the mapping stream — gather (live records -> list), two
// memsets, reinsert.  Voxel contents are untouched: only slot positions change (nothing keeps slot ids across sweeps except the
// deferred list, which is rebuilt here).
__global__ __launch_bounds__(256) void k_map_rebuild_gather(VoxelTable T, VoxelRec* __restrict__ tmp, int cap, int* n_tmp, MapFrame* fr, int kind) {
  if (blockIdx.x == 0 && threadIdx.x == 0) { fr->n_deferred[kind] = 0; fr->n_newraw[kind] = 0; }   // slot ids change: both lists are rebuilt from the raw flags
  for (unsigned s = blockIdx.x * 256 + threadIdx.x; s <= T.mask; s += gridDim.x * 256) {
    const RecVal v = rec_load(&T.rec[s]);
    if (v.key == 0ull || v.count == 0) continue;
    const int o = atomicAdd(n_tmp, 1);
    if (o < cap) { VoxelRec r; r.key = v.key; r.sx = v.sum.x; r.sy = v.sum.y; r.sz = v.sum.z; r.si = v.sum.w; r.count = v.count; r.pend_cnt = key_seq(v.key) ? v.pend_cnt : 0; tmp[o] = r; }   // (a raw point's pend_cnt is its arrival stamp)
    else atomicOr(&fr->error, kErrMapFull);
  }
}
__global__ __launch_bounds__(256) void k_map_rebuild_insert(VoxelTable T, const VoxelRec* __restrict__ tmp, int cap, int* n_tmp, MapFrame* fr, int kind,
                                                            int* __restrict__ deferred, int deferred_cap, float inv, int* host_flags) {
  const int n = min(*n_tmp, cap);
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const VoxelRec r = tmp[e];
    unsigned s = (unsigned)mix64(r.key) & T.mask;
    bool done = false;
    for (int probe = 0; probe < kMaxProbe && !done; probe++, s = (s + 1) & T.mask) {
      if (atomicCAS(&T.rec[s].key, 0ull, r.key) != 0ull) continue;   // keys are unique in the list
      rec_store_value(&T.rec[s], make_float4(r.sx, r.sy, r.sz, r.si), r.count, r.pend_cnt);
      int Ai, Aj, Ak;
      unpack_cube(r.key, &Ai, &Aj, &Ak);
      if (key_seq(r.key) == 0 && !map_publish_block(T, Ai, Aj, Ak, key_lx(r.key), key_ly(r.key), key_lz(r.key))) atomicOr(&fr->error, kErrMapFull);
      if (key_seq(r.key) == 0 && rec_raw(r.count)) {  // raw voxel of a cube outside the valid block: still owed a centroid (see k_map_finalize)
        const int dpos = atomicAdd(&fr->n_deferred[kind], 1);
        if (dpos < deferred_cap) deferred[dpos] = (int)s;
      }
      done = true;
    }
    if (!done) atomicOr(&fr->error, kErrMapFull);
  }
  (void)inv;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    T.stats[0] = n; T.stats[1] = 0;
    if (host_flags) __hip_atomic_store(&host_flags[kind], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void k_map_register(const float4* __restrict__ cloud, const FrameScalars* __restrict__ S, const MapState* __restrict__ ms,
                               float4* __restrict__ out) {
  const int n = S->N2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    out[i] = associate_to_map(cloud[i], ms->parameters, ms->parameters + 4);  // LM:795-799
}

// ---------------------------------------------------------------------------------------------- host side
vloam_status map_layout(MapContext* m, const vloam_config& cfg, Arena& A) {
  bool ok = true;
  ok = ok && A.take(&m->state, 1) && A.take(&m->frame, 1) && A.take(&m->cube_cnt, 2 * (size_t)kCubeNum);
  const int lg = cfg.map_capacity_log2 < 10 ? 10 : (cfg.map_capacity_log2 > 28 ? 28 : cfg.map_capacity_log2);
  const size_t slots = (size_t)1 << lg;
  for (int k = 0; k < 2 && ok; k++) {
    VoxelTable& T = m->tab[k];
    ok = ok && A.take(&T.rec, slots) && A.take(&T.pend, slots * kPendCap) && A.take(&T.stats, 4);
    T.mask = (unsigned)(slots - 1);
    const size_t bslots = slots / 2;
    ok = ok && A.take(&T.blk, bslots);
    T.bslots_mask = (unsigned)(bslots - 1);
    DsScratch& D = m->ds[k];
    if (k == 0) for (int c = 0; c < MapContext::kSets; c++) ok = ok && A.take(&m->stack_info[c], 1);
    D.stack_cap = k ? kStackCapSurf : kStackCapCorner;
    ok = ok && A.take(&D.region, (size_t)kDsMaxBins * kDsBinCap) && A.take(&D.over, (size_t)cfg.max_points) && A.take(&D.tmp, (size_t)cfg.max_points) &&
         A.take(&D.sorted, (size_t)cfg.max_points) && A.take(&D.splitters, (size_t)kDsMaxBins) && A.take(&D.cursor, (size_t)kDsMaxBins + 4) &&
         A.take(&D.done, (size_t)kDsMaxBins);
    for (int c = 0; c < MapContext::kSets; c++) ok = ok && A.take(&m->stack_sets[c][k], (size_t)D.stack_cap);
    m->stack[k] = m->stack_sets[0][k];
    ok = ok && A.take(&m->stack_map[k], (size_t)D.stack_cap) && A.take(&m->touched[k], (size_t)D.stack_cap) && A.take(&m->deferred[k], (size_t)D.stack_cap) &&
         A.take(&m->newraw[k], (size_t)D.stack_cap);
    FactorTable& F = m->F[k];
    F.cap = kMapFactorCap;
    ok = ok && A.take(&F.type, (size_t)F.cap) && A.take(&F.p, 3 * (size_t)F.cap) && A.take(&F.A, 3 * (size_t)F.cap) && A.take(&F.B, 3 * (size_t)F.cap) &&
         A.take(&F.resid, 3 * (size_t)F.cap) && A.take(&F.ctype, (size_t)F.cap) && A.take(&F.cslot, (size_t)F.cap) && A.take(&F.cpack, 11 * (size_t)F.cap) && A.take(&F.dg, 8 * (size_t)F.cap) &&
         A.take(&F.rowcnt, (size_t)F.cap / 64 + 1) && A.take(&F.rowmask, (size_t)F.cap / 64 + 2);
    F.gsync = nullptr;  // the handle places the sync words (lm_sync_calibrate)
    F.err = ok ? &m->frame->error : nullptr;
    F.fallbacks = ok ? &m->frame->fallback_solves : nullptr;
    F.host_degraded = nullptr;   // (the handle points it at its host-mapped word)
    F.gen = 0; F.spin_limit = 1 << 18;
  }
  ok = ok && A.take(&m->rec, 2) && A.take(&m->nbr, 5 * (size_t)kMapFactorCap);
  ok = ok && A.take(&m->cbox, 2 * (size_t)kMapFactorCap) && A.take(&m->ccand, (size_t)kMapFactorCap * kCandCache);
  m->rebuild_cap = (int)(slots / 2);
  ok = ok && A.take(&m->rebuild_tmp, (size_t)m->rebuild_cap) && A.take(&m->rebuild_n, 2);
  ok = ok && A.take(&m->registered, (size_t)cfg.max_points) && A.take(&m->assoc_cyc, 16) && A.take(&m->ts_log, 2048);
  if (!ok) return VLOAM_ERR_HIP;
  m->max_points = cfg.max_points;
  m->inv_leaf[0] = 1.0f / cfg.mapping_line_resolution;   // inverse_leaf_size_ of downSizeFilterCorner (LM:100)
  m->inv_leaf[1] = 1.0f / cfg.mapping_plane_resolution;  // downSizeFilterSurf (LM:101)
  return VLOAM_OK;
}

vloam_status map_init(MapContext* m, hipStream_t st) {
  if (!m->host_flags) {
    if (hipHostMalloc((void**)&m->host_flags, sizeof(int) * 2 * kMaxBatch, hipHostMallocMapped) != hipSuccess) { m->host_flags = nullptr; return VLOAM_ERR_HIP; }
    for (int k = 0; k < 2 * kMaxBatch; k++) m->host_flags[k] = 0;
  }
  MapState init;
  memset(&init, 0, sizeof(init));
  init.parameters[3] = 1.0; init.q_wmap_wodom[3] = 1.0; init.q_wodom_curr[3] = 1.0;  // LM:74-91
  init.cenW = 10; init.cenH = 10; init.cenD = 5;                                      // laser_mapping.h:76-78
  if (hipMemcpyAsync(m->state, &init, sizeof(init), hipMemcpyHostToDevice, st) != hipSuccess) return VLOAM_ERR_HIP;
  return hipStreamSynchronize(st) == hipSuccess ? VLOAM_OK : VLOAM_ERR_HIP;
}

// pcl::VoxelGrid of this sweep's lessSharp / lessFlat clouds (LM:432-440) -> stack set `set`.  Needs nothing but the scan
// registration output, so it is enqueued on the scan-registration stream, ahead of the mapping stage that consumes it.
vloam_status map_stack_enqueue(MapContext* m, hipStream_t st, const SRBuffers& cur, int set, ProfHook* ph, hipEvent_t done) {
  StackInfo* si = m->stack_info[set];
  const unsigned Z = (unsigned)m->se.B;
  const size_t ss = m->se.ss;
  m->ds_gen++;   // tags the per-bin cell counts of this sweep's reduce pass (look-back words are never reset)
  VLOAM_LAUNCH(ph, kKMapStack, st, k_map_ds_bin, dim3(32, 2, Z), dim3(256), 0, st, cur.less_sharp, cur.less_flat, cur.S, m->ds[0], m->ds[1],
               m->inv_leaf[0], m->inv_leaf[1], m->stack_sets[set][0], m->stack_sets[set][1], si, ss);
  // `done` (the stack of this sweep is complete) is bound to the last dispatch instead of a marker packet behind it
  VLOAM_LAUNCH_EV(ph, kKMapDsReduce, st, done, k_map_ds_reduce, dim3(kDsReduceGrid, 2, Z), dim3(256), 0, st, cur.less_sharp, cur.less_flat, cur.S, m->ds[0], m->ds[1],
                  m->inv_leaf[0], m->inv_leaf[1], m->stack_sets[set][0], m->stack_sets[set][1], si, m->ds_gen, ss);
  return hipGetLastError() == hipSuccess ? VLOAM_OK : VLOAM_ERR_HIP;
}

void map_destroy(MapContext* m) {
  if (m->host_flags) { (void)hipHostFree(m->host_flags); m->host_flags = nullptr; }
}

// gather the live records, clear the table, reinsert (see k_map_rebuild_*); between two sweeps on the mapping stream; one session
static vloam_status map_rebuild_enqueue(MapContext* m0, hipStream_t st, int session, int kind) {
  MapContext ms_ = m0->for_session(session);
  MapContext* m = &ms_;
  VoxelTable& T = m->tab[kind];
  const size_t slots = (size_t)T.mask + 1, bslots = (size_t)T.bslots_mask + 1;
  const int cap = kind ? kStackCapSurf : kStackCapCorner;
  if (hipMemsetAsync(m->rebuild_n, 0, sizeof(int), st) != hipSuccess) return VLOAM_ERR_HIP;
  VL_RAW_LAUNCH(k_map_rebuild_gather, dim3(1024), dim3(256), 0, st, T, m->rebuild_tmp, m->rebuild_cap, m->rebuild_n, m->frame, kind);
  if (hipMemsetAsync(T.rec, 0, slots * sizeof(VoxelRec), st) != hipSuccess) return VLOAM_ERR_HIP;
  if (hipMemsetAsync(T.blk, 0, bslots * sizeof(ulonglong2), st) != hipSuccess) return VLOAM_ERR_HIP;
  if (hipMemsetAsync(T.stats, 0, 4 * sizeof(int), st) != hipSuccess) return VLOAM_ERR_HIP;
  VL_RAW_LAUNCH(k_map_rebuild_insert, dim3(1024), dim3(256), 0, st, T, m->rebuild_tmp, m->rebuild_cap, m->rebuild_n, m->frame, kind,
                     m->deferred[kind], cap, m->inv_leaf[kind], m->host_flags);
  m0->rebuilds++;
  return hipGetLastError() == hipSuccess ? VLOAM_OK : VLOAM_ERR_HIP;
}

vloam_status map_force_rebuild(MapContext* m, hipStream_t st) {
  for (int b = 0; b < m->se.B; b++)
    for (int k = 0; k < 2; k++) { vloam_status s = map_rebuild_enqueue(m, st, b, k); if (s != VLOAM_OK) return s; }
  return VLOAM_OK;
}

static const bool g_ts_log = getenv("VLOAM_TS_LOG") != nullptr;   // debug: device-side time stamps of the mapping stage (tools/map_stream_gaps.py)
vloam_status map_enqueue(MapContext* m, const vloam_config& cfg, hipStream_t st, const SRBuffers& cur, LOState* lo, double* traj_row14,
                         bool skip_frame, int set, ProfHook* ph, hipEvent_t done) {
  (void)cur;
  MapState* ms = m->state;
  MapFrame* fr = m->frame;
  const unsigned Z = (unsigned)m->se.B;
  const size_t ss = m->se.ss;
  m->stack[0] = m->stack_sets[set][0]; m->stack[1] = m->stack_sets[set][1];
  if (!skip_frame) {
    // the flag is written by k_map_finalize of an EARLIER sweep (plain read of host-mapped memory, no synchronisation): a rebuild
    // a few sweeps late is as good; the cool-down covers the sweeps already in flight that still report the old state
    for (int b = 0; b < m->se.B; b++)
      for (int k = 0; k < 2; k++) {
        if (m->rebuild_cooldown[b][k] > 0) { m->rebuild_cooldown[b][k]--; continue; }
        if (__atomic_load_n(&m->host_flags[2 * b + k], __ATOMIC_RELAXED)) {
          if (map_rebuild_enqueue(m, st, b, k) != VLOAM_OK) return VLOAM_ERR_HIP;
          m->rebuild_cooldown[b][k] = 8;
        }
      }
  }
  // `done` (mapping of this sweep finished) rides on the sweep's last dispatch: a marker packet behind it costs ~5 us of idle stream
  VLOAM_LAUNCH_EV(ph, kKMapPrepare, st, skip_frame ? done : nullptr, k_map_prepare, dim3(1, 1, Z), dim3(256), 0, st, ms, fr, lo, m->cube_cnt,
                  skip_frame ? 1 : 0, traj_row14, m->stack_info[set], m->deferred[0], m->deferred[1], m->newraw[0], m->newraw[1], g_ts_log ? m->ts_log : (long long*)nullptr, ss);
  if (skip_frame) return hipGetLastError() == hipSuccess ? VLOAM_OK : VLOAM_ERR_HIP;
  for (int outer = 0; outer < 2; outer++) {  // LM:458
    // lanes per query of the 5-NN search: a batch fills the chip with 16-lane groups (four queries per wavefront); a single sequence
    // (9 000 queries on 5 120 wave slots) is quickest with two per wavefront.  VLOAM_MAP_ASSOC_LANES = 16 / 32 / 64 overrides (A/B runs;
    // the round-2 form — one wavefront per query, rank counting in LDS — measured 27.8 / 184 us at B = 1 / 8 against 16 / 105 here).
    static const int g_env = getenv("VLOAM_MAP_ASSOC_LANES") ? atoi(getenv("VLOAM_MAP_ASSOC_LANES")) : -1;
    const int G = g_env >= 0 ? g_env : (m->se.B > 1 ? 16 : 32);
    auto grid_for = [](int g) { const int qw = 256 / g; return dim3(8 * ((kStackCapCorner / qw + 7) / 8 + (kStackCapSurf / qw + 7) / 8), 1, 1); };
#define VL_MAP_ASSOC(KERN, GRID)                                                                                                      \
    VLOAM_LAUNCH(ph, kKMapAssoc, st, KERN, dim3((GRID).x, 1, Z), dim3(256), 0, st, m->stack[0], m->stack[1], m->tab[0], m->tab[1], \
                 m->inv_leaf[0], m->inv_leaf[1], ms, fr, m->nbr, outer, m->cbox, m->ccand, cfg.debug ? m->assoc_cyc : (long long*)nullptr, ss)
    static const int kb_env = getenv("VLOAM_MAP_ASSOC_KB") ? atoi(getenv("VLOAM_MAP_ASSOC_KB")) : 2;   // blocks per lane and trip of the 16-lane groups
    if (G == 16 && kb_env == 1) VL_MAP_ASSOC((k_map_assoc<16, 1>), grid_for(16));
    else if (G == 16) VL_MAP_ASSOC((k_map_assoc<16, 2>), grid_for(16));
    else if (G == 64) VL_MAP_ASSOC((k_map_assoc<64, 1>), grid_for(64));
    else VL_MAP_ASSOC((k_map_assoc<32, 1>), grid_for(32));
#undef VL_MAP_ASSOC
    VLOAM_LAUNCH(ph, kKMapFit, st, k_map_fit, dim3(kMapFactorCap / 256, 1, Z), dim3(256), 0, st, m->stack[0], m->stack[1], m->tab[0], m->tab[1], ms, fr, m->nbr,
                 m->F[outer], outer, ss);
    lm_launch(st, m->se, m->F[outer], kStackCapCorner, ms->parameters, m->rec + outer, 4, 0.1, true, &ms->do_optimize, ph);
  }
  VLOAM_LAUNCH(ph, kKMapInsert, st, k_map_insert, dim3(64, 2, Z), dim3(256), 0, st, m->stack[0], m->stack[1], m->stack_map[0], m->stack_map[1],
               m->tab[0], m->tab[1], m->inv_leaf[0], m->inv_leaf[1], ms, fr, m->touched[0], m->touched[1], m->deferred[0], m->deferred[1], traj_row14, ss);
  VLOAM_LAUNCH_EV(ph, kKMapFinalize, st, done, k_map_finalize, dim3(kStackCapSurf / 256, 2, Z), dim3(256), 0, st, m->stack_map[0], m->stack_map[1],
                  m->tab[0], m->tab[1], ms, fr, m->touched[0], m->touched[1], m->deferred[0], m->deferred[1], m->newraw[0], m->newraw[1], m->cube_cnt, m->host_flags, g_ts_log ? m->ts_log : (long long*)nullptr, ss);
  return hipGetLastError() == hipSuccess ? VLOAM_OK : VLOAM_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------- map export
// /laser_cloud_map (LM:778-793): for cube index 0..4850 the corner cloud then the surf cloud of the cube.  A cube cloud that has been
// through its VoxelGrid re-filter (every cube that was ever valid) is ordered by voxel (iz, iy, ix); points that arrived while the cube
// was outside the valid block follow un-merged, in arrival order (sweep, then stack order) — the raw point records of k_map_finalize; a
// centroid that became raw point number one (stamp 0) keeps its place in the voxel-ordered part.  The device compacts the live records
// with their order key; the final ordering of this OUTPUT path is a host sort of the compacted list.
struct ExportRow { u64 okey; float x, y, z, w; };
__global__ __launch_bounds__(256) void k_map_export(VoxelTable T, const MapState* __restrict__ ms, int kind, ExportRow* __restrict__ out, long long cap,
                                                    unsigned long long* n_out) {
  const int cW = ms->cenW, cH = ms->cenH, cD = ms->cenD;
  for (unsigned s = blockIdx.x * 256 + threadIdx.x; s <= T.mask; s += gridDim.x * 256) {
    const RecVal v = rec_load(&T.rec[s]);
    if (v.key == 0ull || v.count == 0) continue;
    const int seq = key_seq(v.key);
    if (seq == 0 && rec_raw(v.count)) continue;   // a raw voxel is published through its point records
    int Ai, Aj, Ak;
    unpack_cube(v.key, &Ai, &Aj, &Ak);
    const int i = Ai + cW, j = Aj + cH, k = Ak + cD;
    if (i < 0 || i >= kCubeW || j < 0 || j >= kCubeH || k < 0 || k >= kCubeD) continue;
    const u64 cube = (u64)(i + kCubeW * j + kCubeW * kCubeH * k);
    const u64 lx = (u64)key_lx(v.key), ly = (u64)key_ly(v.key), lz = (u64)key_lz(v.key);
    const bool tail = seq != 0 && v.pend_cnt != 0;   // an un-merged arrival: behind the voxel-ordered part, by arrival stamp
    ExportRow r;
    r.okey = (cube << 50) | ((u64)kind << 49) | ((u64)(tail ? 1 : 0) << 48) | (tail ? (u64)(unsigned)v.pend_cnt : ((lz << (2 * kVoxBits)) | (ly << kVoxBits) | lx));
    const int n = seq ? 1 : rec_n(v.count);
    const float nn = (float)n;
    r.x = n > 1 ? v.sum.x / nn : v.sum.x; r.y = n > 1 ? v.sum.y / nn : v.sum.y; r.z = n > 1 ? v.sum.z / nn : v.sum.z;
    r.w = n > 1 ? v.sum.w / nn : v.sum.w;
    const unsigned long long o = atomicAdd(n_out, 1ull);
    if ((long long)o < cap) out[o] = r;
  }
}

vloam_status map_export(MapContext* m0, hipStream_t st, float* xyzi4, long long cap, long long* n) {
  const MapContext msel = m0->for_session(m0->sel);
  const MapContext* m = &msel;
  if (hipStreamSynchronize(st) != hipSuccess) return VLOAM_ERR_HIP;
  int stats[2][4];
  for (int k = 0; k < 2; k++) if (hipMemcpy(stats[k], m->tab[k].stats, sizeof(stats[k]), hipMemcpyDeviceToHost) != hipSuccess) return VLOAM_ERR_HIP;
  const long long bound = (long long)stats[0][0] + stats[1][0] + 1;  // keys ever inserted >= live voxels
  ExportRow* d_rows = nullptr;
  unsigned long long* d_n = nullptr;
  if (hipMalloc((void**)&d_rows, (size_t)bound * sizeof(ExportRow)) != hipSuccess || hipMalloc((void**)&d_n, sizeof(*d_n)) != hipSuccess) return VLOAM_ERR_HIP;
  vloam_status rc = VLOAM_OK;
  unsigned long long cnt = 0;
  if (hipMemsetAsync(d_n, 0, sizeof(*d_n), st) != hipSuccess) rc = VLOAM_ERR_HIP;
  for (int k = 0; k < 2 && rc == VLOAM_OK; k++) VL_RAW_LAUNCH(k_map_export, dim3(1024), dim3(256), 0, st, m->tab[k], m->state, k, d_rows, bound, d_n);
  if (rc == VLOAM_OK && (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(&cnt, d_n, sizeof(cnt), hipMemcpyDeviceToHost) != hipSuccess)) rc = VLOAM_ERR_HIP;
  std::vector<ExportRow> rows((size_t)((long long)cnt < bound ? (long long)cnt : bound));
  if (rc == VLOAM_OK && !rows.empty() && hipMemcpy(rows.data(), d_rows, rows.size() * sizeof(ExportRow), hipMemcpyDeviceToHost) != hipSuccess) rc = VLOAM_ERR_HIP;
  (void)hipFree(d_rows); (void)hipFree(d_n);
  if (rc != VLOAM_OK) return rc;
  std::sort(rows.begin(), rows.end(), [](const ExportRow& a, const ExportRow& b) { return a.okey < b.okey; });
  if (n) *n = (long long)rows.size();
  const long long c = (long long)rows.size() < cap ? (long long)rows.size() : cap;
  for (long long i = 0; xyzi4 && i < c; i++) { xyzi4[4 * i] = rows[(size_t)i].x; xyzi4[4 * i + 1] = rows[(size_t)i].y; xyzi4[4 * i + 2] = rows[(size_t)i].z; xyzi4[4 * i + 3] = rows[(size_t)i].w; }
  return VLOAM_OK;
}

vloam_status map_get_cloud(MapContext* m0, hipStream_t st, int which, const SRBuffers& cur, float* xyzi4, int cap, int* n) {
  // `cur` is already the selected session's buffer set (the caller rebased it)
  const MapContext msel = m0->for_session(m0->sel);
  const MapContext* m = &msel;
  if (hipStreamSynchronize(st) != hipSuccess) return VLOAM_ERR_HIP;
  MapState ms;
  if (hipMemcpy(&ms, m->state, sizeof(ms), hipMemcpyDeviceToHost) != hipSuccess) return VLOAM_ERR_HIP;
  const float4* src = nullptr;
  int cnt = 0;
  if (which == 7) { src = m->stack[0]; cnt = ms.n_corner_stack; }
  else if (which == 8) { src = m->stack[1]; cnt = ms.n_surf_stack; }
  else if (which == 11) {
    FrameScalars S;
    if (hipMemcpy(&S, cur.S, sizeof(S), hipMemcpyDeviceToHost) != hipSuccess) return VLOAM_ERR_HIP;
    VL_RAW_LAUNCH(k_map_register, dim3(256), dim3(256), 0, st, cur.cloud, cur.S, m->state, m->registered);
    if (hipStreamSynchronize(st) != hipSuccess) return VLOAM_ERR_HIP;
    src = m->registered; cnt = S.N2;
  } else return VLOAM_ERR_INVALID;
  *n = cnt;
  const int c = cnt < cap ? cnt : cap;
  if (xyzi4 && c > 0 && hipMemcpy(xyzi4, src, (size_t)c * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess) return VLOAM_ERR_HIP;
  return VLOAM_OK;
}

__global__ void k_map_error_fetch(MapFrame* fr, int clear_mask, int* out) { out[0] = atomicAnd(&fr->error, ~clear_mask); out[1] = fr->fallback_solves; }

// the sticky error words of all sessions OR-ed; the bits of clear_mask are reported once and cleared (transient per-sweep conditions)
vloam_status map_error(MapContext* m, int* e, int clear_mask, long long* fallback_solves) {
  *e = 0;
  if (fallback_solves) *fallback_solves = 0;
  for (int b = 0; b < m->se.B; b++) {
    const MapContext mb = m->for_session(b);
    int* d_out = mb.rebuild_n;  // scratch ints (no rebuild can be in flight: the caller has synchronised the streams)
    VL_RAW_LAUNCH(k_map_error_fetch, dim3(1), dim3(1), 0, 0, mb.frame, clear_mask, d_out);
    int v[2] = {0, 0};
    if (hipMemcpy(v, d_out, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return VLOAM_ERR_HIP;
    *e |= v[0];
    if (fallback_solves) *fallback_solves += v[1];
  }
  return VLOAM_OK;
}

static vloam_status copy_dev(const void* src, size_t bytes, void* buf, long long cap, long long* n) {
  if (n) *n = (long long)bytes;
  const size_t c = bytes < (size_t)cap ? bytes : (size_t)cap;
  if (buf && c && hipMemcpy(buf, src, c, hipMemcpyDeviceToHost) != hipSuccess) return VLOAM_ERR_HIP;
  return VLOAM_OK;
}

// item = outer * 16 + k:  k = 0 factor types i32[kMapFactorCap], 1 A f64[3][cap], 2 B f64[3][cap], 3 LM record, 4 residuals f64[3][cap],
//                         5 p f64[3][cap].  item 64: MapState.  item 65: MapFrame.  item 66: cube_cnt i32[2][4851].
//                         item 67/68: dump of the corner / surf table as rows {key lo, key hi, count, x, y, z, w} (7 x 4 bytes) for live slots.
vloam_status map_debug_get(MapContext* m0, int item, void* buf, long long cap, long long* n) {
  const MapContext msel = m0->for_session(m0->sel);
  const MapContext* m = &msel;
  if (item == 64) return copy_dev(m->state, sizeof(MapState), buf, cap, n);
  if (item == 65) return copy_dev(m->frame, sizeof(MapFrame), buf, cap, n);
  if (item == 66) return copy_dev(m->cube_cnt, sizeof(int) * 2 * kCubeNum, buf, cap, n);
  if (item == 67 || item == 68) {
    const VoxelTable& T = m->tab[item - 67];
    const size_t slots = (size_t)T.mask + 1;
    std::vector<VoxelRec> recs(slots);
    if (hipMemcpy(recs.data(), T.rec, slots * sizeof(VoxelRec), hipMemcpyDeviceToHost) != hipSuccess) return VLOAM_ERR_HIP;
    std::vector<unsigned> rows;
    for (size_t s = 0; s < slots; s++) {
      if (recs[s].key == 0 || recs[s].count == 0) continue;
      if ((recs[s].key >> 56) == 0 && (recs[s].count & (1 << 30))) continue;   // a raw voxel shows up as its point records (seq != 0)
      unsigned r[7];
      r[0] = (unsigned)(recs[s].key & 0xffffffffu); r[1] = (unsigned)(recs[s].key >> 32); r[2] = (unsigned)recs[s].count;
      memcpy(r + 3, &recs[s].sx, 16);
      rows.insert(rows.end(), r, r + 7);
    }
    if (n) *n = (long long)(rows.size() * 4);
    const size_t c = rows.size() * 4 < (size_t)cap ? rows.size() * 4 : (size_t)cap;
    if (buf && c) memcpy(buf, rows.data(), c);
    return VLOAM_OK;
  }
  if (item == 72) return copy_dev(m->ts_log, sizeof(long long) * 2048, buf, cap, n);   // VLOAM_TS_LOG=1: [sweep % 1024][prepare start, finalize start], 100 MHz ticks
  if (item == 73) return copy_dev(m->cbox, sizeof(int4) * 2 * (size_t)kMapFactorCap, buf, cap, n);   // per stack slot: the first round's search box + its candidate count (-1: not cached)
  if (item == 71) return copy_dev(m->assoc_cyc, sizeof(long long) * 16, buf, cap, n);   // k_map_assoc phase cycles (debug handles): [outer][6 phases, spare, wavefronts]
  if (item == 69) {  // table health: {keys, purged, block keys, spare} x {corner, surf}, rebuilds, largest candidate list
    int out[12] = {0};
    for (int k = 0; k < 2; k++) if (hipMemcpy(out + 4 * k, m->tab[k].stats, 4 * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return VLOAM_ERR_HIP;
    MapFrame fr;
    if (hipMemcpy(&fr, m->frame, sizeof(fr), hipMemcpyDeviceToHost) != hipSuccess) return VLOAM_ERR_HIP;
    out[8] = (int)m0->rebuilds; out[9] = fr.max_candidates; out[10] = fr.n_deferred[0] + fr.n_newraw[0]; out[11] = fr.n_deferred[1] + fr.n_newraw[1];
    if (n) *n = sizeof(out);
    if (buf) memcpy(buf, out, (size_t)cap < sizeof(out) ? (size_t)cap : sizeof(out));
    return VLOAM_OK;
  }
  const int outer = item / 16, k = item % 16;
  if (outer < 0 || outer > 1) return VLOAM_ERR_INVALID;
  const FactorTable& F = m->F[outer];
  switch (k) {
    case 0: return copy_dev(F.type, sizeof(int) * F.cap, buf, cap, n);
    case 1: return copy_dev(F.A, sizeof(double) * 3 * F.cap, buf, cap, n);
    case 2: return copy_dev(F.B, sizeof(double) * 3 * F.cap, buf, cap, n);
    case 3: return copy_dev(m->rec + outer, sizeof(LMRecord), buf, cap, n);
    case 4: return copy_dev(F.resid, sizeof(double) * 3 * F.cap, buf, cap, n);
    case 5: return copy_dev(F.p, sizeof(double) * 3 * F.cap, buf, cap, n);
  }
  return VLOAM_ERR_INVALID;
}

vloam_status map_counts(MapContext* m0, long long c[16]) {
  const MapContext msel = m0->for_session(m0->sel);
  const MapContext* m = &msel;
  MapState ms;
  MapFrame fr;
  LMRecord rec[2];
  if (hipMemcpy(&ms, m->state, sizeof(ms), hipMemcpyDeviceToHost) != hipSuccess) return VLOAM_ERR_HIP;
  if (hipMemcpy(&fr, m->frame, sizeof(fr), hipMemcpyDeviceToHost) != hipSuccess) return VLOAM_ERR_HIP;
  if (hipMemcpy(rec, m->rec, sizeof(rec), hipMemcpyDeviceToHost) != hipSuccess) return VLOAM_ERR_HIP;
  c[11] = ms.n_corner_stack; c[12] = ms.n_surf_stack;
  c[13] = fr.n_factors[1][0] + fr.n_factors[1][1];
  c[14] = ms.do_optimize ? (long long)(rec[0].n_evals + rec[1].n_evals) : 0;
  c[15] = ms.n_map_corner + ms.n_map_surf;
  return VLOAM_OK;
}

}  // namespace vloam
