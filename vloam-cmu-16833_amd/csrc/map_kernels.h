// laserMapping on gfx950 — host-visible interface (map_kernels.hip).
//
// The reference keeps 21x21x11 cubes of pcl clouds, gathers the 5x5x3 block around the sensor, builds
// two kd-trees per sweep and re-voxelises every valid cube (laser_mapping.cpp:198-708).  Here the map
// is a PERSISTENT voxel hash in HBM keyed by (absolute cube, voxel inside the cube), one slot per
// per-cube VoxelGrid cell, holding the cell's f32 running sum + count — which reproduces the
// reference's "centroid of (old centroid, new points...)" update exactly (SURVEY.md §8a map notes)
// with no per-sweep gather / tree build / re-sort.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "../../include/vloam_hip/c_api.h"
#include "sr_kernels.h"
#include "vloam_device.h"

namespace vloam {

constexpr int kCubeW = 21, kCubeH = 21, kCubeD = 11, kCubeNum = kCubeW * kCubeH * kCubeD;  // laser_mapping.h:110-117
constexpr int kStackCapCorner = 8192;   // >= kMaxLessSharp
constexpr int kStackCapSurf = 24576;    // voxels of one sweep's lessFlat cloud at the plane resolution (with the corner stack: 32 768 slots = the 512 mask rows of a solve)
constexpr int kMapFactorCap = kStackCapCorner + kStackCapSurf;
constexpr int kPendCap = 16;            // stack points that may land in one map voxel in one sweep
// Voxel indices one axis of a 50 m cube can take at a leaf of 1 / inv (+ slack: the first index of a cube is taken one cell early, and the
// cube test runs in f64 on the point while the voxel index is an f32 product): the radix of k_map_assoc's 32-bit tie rank — the position of
// a voxel in the reference's gathered map cloud, 75 cubes x radix^3.  vloam_create keeps 75 radix^3 below 2^32 (leaf >= 0.132 m); the
// voxel key itself has 9 bits per axis (map_kernels.hip).
__host__ __device__ inline int vox_radix(float inv) { return (int)(50.0f * inv) + 4; }

// One map voxel == one 32-byte record, so that a candidate of the 5-NN search, an insert and a finalize each touch ONE cache line
// (round 1 kept keys / sums / counts in three arrays: three scattered lines per candidate, 27x the algorithmic traffic).
struct __attribute__((aligned(32))) VoxelRec {
  unsigned long long key;    // 0 = empty
  float sx, sy, sz, si;      // f32 running sum (x, y, z, intensity) in arrival order
  int count;                 // points in the sum (1 after a valid-cube finalize; 0 = purged, or created this sweep and not finalized yet)
  int pend_cnt;              // stack points queued this sweep
};
static_assert(sizeof(VoxelRec) == 32, "one voxel, one half cache line");

struct VoxelTable {       // open addressing, linear probing, probe chains bounded by kMaxProbe
  VoxelRec* rec;
  int* pend;                 // [slots][kPendCap] stack indices
  unsigned mask;             // slots - 1
  // occupancy index: one entry per 4 x 4 x 4 block of voxels of a cube -> 64-bit mask of the voxels that exist.
  // The 5-NN search probes ~27 blocks instead of ~343 mostly empty voxels.
  ulonglong2* blk;           // {key (0 = empty), mask}: one 16-byte load per probe
  unsigned bslots_mask;      // block slots - 1
  int* stats;                // [4] keys in the table (live + purged) | entries purged since the last rebuild | block keys | spare
  __host__ __device__ void rebase(size_t off) { rbp(rec, off); rbp(pend, off); rbp(blk, off); rbp(stats, off); }
};
constexpr int kMaxProbe = 128;  // a longer chain means the table is overloaded: the lookup reports kErrMapFull instead of spinning
constexpr int kCandChunk = 256; // candidate lists longer than this are reported in MapFrame::max_candidates (the 5-NN search takes them in several passes, exactly)
constexpr int kCandCache = 128; // entries per stack point in the second outer round's candidate cache (>= the pass size of every lane count of k_map_assoc)

// scan-feature VoxelGrid (k_map_ds_bin / k_map_ds_reduce): the cell-index space of a sweep's cloud is cut into bins of ~kDsBinTarget points
constexpr int kDsMaxBins = 256;      // bins per cloud (a cloud of more than 256 x 512 points gets fuller bins)
constexpr int kDsBinTarget = 512;    // points per bin aimed at
constexpr int kDsBinCap = 4096;      // keys a bin's region holds == keys the reduce pass sorts in LDS; a fuller bin spills into the overflow list
constexpr int kDsBigCell = 32;       // cells with more points than this are folded by a whole wavefront
constexpr int kDsBigCap = 128;       // such cells per bin (4 096 / 33)

struct DsScratch {        // per-sweep VoxelGrid of the scan features (laser_mapping.cpp:432-440)
  unsigned long long* region;     // [kDsMaxBins][kDsBinCap] sort keys (cell index << 24 | point index) by bin, arrival order inside a bin
  unsigned long long* over;       // [max_points] keys that found their bin's region full
  unsigned long long* tmp;        // [max_points] slow path: an over-full bin gathered from its region + the overflow list ...
  unsigned long long* sorted;     // [max_points] ... and ranked
  unsigned long long* splitters;  // [kDsMaxBins] this sweep's bin boundaries (cell indices)
  int* cursor;                    // [kDsMaxBins] keys per bin | [kDsMaxBins] overflow entries | [+1] ticket of the reduce pass | [+2] slow-path scratch in use
  unsigned long long* done;       // [kDsMaxBins] look-back words: sweep generation << 32 | cells of the bin
  int stack_cap;
  __host__ __device__ void rebase(size_t off) {
    rbp(region, off); rbp(over, off); rbp(tmp, off); rbp(sorted, off); rbp(splitters, off); rbp(cursor, off); rbp(done, off);
  }
};

struct StackInfo {        // per stack set: counters of the scan-feature VoxelGrid (written on its own stream)
  int n_stack[2];         // laserCloudCornerStackNum / laserCloudSurfStackNum
  int error;
};

struct MapFrame {         // per-sweep device counters
  int n_uniq[2];
  int n_stack[2];
  int n_touched[2];
  int n_deferred[2];
  int n_newraw[2];        // voxels that turned raw this sweep (merged into the deferred list by the next k_map_prepare)
  int rolled;
  int error;
  int n_factors[2][2];    // [outer][corner, surf] accepted factors
  int max_candidates;     // largest 5-NN candidate list seen (diagnostic; lists beyond kCandChunk take extra passes)
  int fallback_solves;    // cooperative Levenberg-Marquardt solves of this session that degraded to one workgroup (lm_solve.hip)
};

struct MapContext {
  MapState* state = nullptr;
  MapFrame* frame = nullptr;
  VoxelTable tab[2];       // 0 corner, 1 surf
  int* cube_cnt = nullptr; // [2][kCubeNum] points per cube, window-relative index (== the reference's array index)
  DsScratch ds[2];
  unsigned ds_gen = 0;      // scan-feature VoxelGrids enqueued so far (tags the look-back words of k_map_ds_reduce)
  static constexpr int kSets = kBufferSets;   // == the SR buffer sets of the handle (same rotation)
  float4* stack_sets[kSets][2] = {};          // laserCloudCornerStack / laserCloudSurfStack (sensor frame), one pair per set
  StackInfo* stack_info[kSets] = {};
  float4* stack[2] = {nullptr, nullptr};      // the pair of the sweep mapping is working on (host-side alias)
  float4* stack_map[2] = {nullptr, nullptr};  // the same points in the map frame (at insert time)
  int* touched[2] = {nullptr, nullptr};       // table slots that received points this sweep
  int* deferred[2] = {nullptr, nullptr};      // slots of the raw voxels (points that arrived while their cube was outside the valid block)
  int* newraw[2] = {nullptr, nullptr};        // ... the ones that turned raw in the sweep being finalized
  FactorTable F[2];        // one per outer round (kept for the parity hooks)
  LMRecord* rec = nullptr; // [2]
  float4* nbr = nullptr;   // [kMapFactorCap][5] the 5 nearest map points of every stack point (.w of the first: 1 = accepted, LM:479 / LM:547)
  int4* cbox = nullptr;    // [kMapFactorCap][2] voxel-index search box + candidate count of the first outer round's 5-NN search
  float4* ccand = nullptr; // [kMapFactorCap][pass size <= kCandCache] its candidates (centroid, tie rank): the second round re-ranks them without a hash probe
  VoxelRec* rebuild_tmp = nullptr;  // live records while a table is being rebuilt (tombstone reclamation after grid rolls)
  int rebuild_cap = 0;
  int* rebuild_n = nullptr;
  int* host_flags = nullptr;        // host-mapped: [kind] 1 = the table of that kind wants a rebuild (written by k_map_finalize)
  int rebuild_cooldown[kMaxBatch][2] = {};
  long long rebuilds = 0;
  Sess se;                 // sessions of the handle (launch geometry .z and arena stride)
  int sel = 0;             // session the host-side getters read (vloam_select_session)
  // a copy whose device pointers address session b (host-side getters, per-session rebuilds)
  MapContext for_session(int b) const {
    MapContext m = *this;
    const size_t off = (size_t)b * se.ss;
    rbp(m.state, off); rbp(m.frame, off); m.tab[0].rebase(off); m.tab[1].rebase(off); rbp(m.cube_cnt, off); m.ds[0].rebase(off); m.ds[1].rebase(off);
    for (int c = 0; c < kSets; c++) { rbp(m.stack_sets[c][0], off); rbp(m.stack_sets[c][1], off); rbp(m.stack_info[c], off); }
    for (int k = 0; k < 2; k++) { rbp(m.stack[k], off); rbp(m.stack_map[k], off); rbp(m.touched[k], off); rbp(m.deferred[k], off); rbp(m.newraw[k], off); m.F[k].rebase(off); }
    rbp(m.rec, off); rbp(m.nbr, off); rbp(m.cbox, off); rbp(m.ccand, off); rbp(m.registered, off); rbp(m.assoc_cyc, off); rbp(m.ts_log, off); rbp(m.rebuild_tmp, off); rbp(m.rebuild_n, off);
    if (m.host_flags) m.host_flags += 2 * b;
    m.se.B = 1; m.sel = 0;
    return m;
  }
  float4* registered = nullptr;  // full-resolution cloud in the map frame, on request
  long long* assoc_cyc = nullptr;  // [2][8] debug: phase cycle sums of k_map_assoc per outer round
  long long* ts_log = nullptr;     // [1024][2] VLOAM_TS_LOG=1: constant-rate (100 MHz) clock at the start of k_map_prepare / end of k_map_finalize of sweep k % 1024
  int max_points = 0;
  float inv_leaf[2] = {0, 0};
};

// layout: carve the session arena (called twice: dry to measure, then for real); init: the initial device state of session 0
vloam_status map_layout(MapContext* m, const vloam_config& cfg, Arena& A);
vloam_status map_init(MapContext* m, hipStream_t st);
vloam_status map_stack_enqueue(MapContext* m, hipStream_t st, const SRBuffers& cur, int set, ProfHook* ph, hipEvent_t done = nullptr);
vloam_status map_enqueue(MapContext* m, const vloam_config& cfg, hipStream_t st, const SRBuffers& cur, LOState* lo, double* traj_row14,
                         bool skip_frame, int set, ProfHook* ph, hipEvent_t done = nullptr);
vloam_status map_get_cloud(MapContext* m, hipStream_t st, int which, const SRBuffers& cur, float* xyzi4, int cap, int* n);
vloam_status map_error(MapContext* m, int* err_bits, int clear_mask = 0, long long* fallback_solves = nullptr);
vloam_status map_debug_get(MapContext* m, int item, void* buf, long long cap, long long* n);
// == /laser_cloud_map (laser_mapping.cpp:778-793): every cube's corner cloud then surf cloud, cube index ascending
vloam_status map_export(MapContext* m, hipStream_t st, float* xyzi4, long long cap, long long* n);
vloam_status map_force_rebuild(MapContext* m, hipStream_t st);  // every session
void map_destroy(MapContext* m);
vloam_status map_counts(MapContext* m, long long c[16]);

}  // namespace vloam
