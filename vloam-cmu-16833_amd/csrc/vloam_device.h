// Device-side data layout shared by the HIP kernels and the host side of libvloam_hip.so.
// gfx950 / CDNA4 only.  All per-frame counts live in HBM (FrameScalars) so that a whole sweep is
// enqueued without a single host round trip.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdint.h>

namespace vloam {

// ---------------------------------------------------------------- sessions of a handle (batched execution)
// One handle drives B independent sequences in lock step: every launch of the sweep chain carries a session index in blockIdx.z.
// All device state of a session lives in ONE arena; session b's arena is the byte-for-byte layout of session 0's, `ss` bytes
// further on.  Hosts and kernels therefore hold session 0's pointers only and kernels rebase them by blockIdx.z * ss on entry —
// one extra kernel argument instead of B copies of every argument.
constexpr int kMaxBatch = 24;   // 10 cooperating solver workgroups per session must be co-resident: 10 * 24 <= 256 compute units (c_api.h)
// Rotating buffer sets of everything one stage hands to a later stage (set = sweep mod kBufferSets): scan-registration output, NN grids,
// mapping stack clouds, VO depth maps.  3 suffice for correctness; the host may enqueue scan registration kBufferSets - 1 sweeps ahead of
// the odometry it waits for and kBufferSets ahead of the mapping.  With 4 sets the host blocked (hipEventSynchronize, ~50-100 us to wake up)
// with only one mapping stage queued behind the running one, and every stream idled 60-75 us per 214 us period
// (profiles/r03_critical_path_4_sets.txt); 8 sets keep several sweeps queued on every stream, so the period is the longest stage again.
constexpr int kBufferSets = 8;
struct Sess { int B = 1; size_t ss = 0; int crowd = 0; int no_coop = 0; };   // crowd: single-sequence handles alive in this process when the handle was created (lm_launch: one-XCD placement only for the first two); no_coop: a cooperative solve of this handle had to degrade once — one-workgroup solves from now on (vloam_sync)
struct BatchIn { const float4* in[kMaxBatch]; int n[kMaxBatch]; };   // the one thing that is not in the arenas: the callers' sweeps
template <class T>
__host__ __device__ inline void rbp(T*& p, size_t off) { if (p) p = (T*)((char*)p + off); }   // pointer arithmetic, NOT an integer round trip:
// the compiler must keep seeing a kernel-argument-derived (global address space) pointer, or every access turns into a FLAT instruction
// (measured: flat atomics made the atomic-heavy kernels 7-9x slower)
#define VL_SESSION(ss) const size_t so_ = (size_t)blockIdx.z * (size_t)(ss)
#define RB(p) ((p) = (p) ? (decltype(p))((char*)(p) + so_) : (p))   // works on __restrict__-qualified kernel parameters too

// Bump allocator over a session arena.  dry = true only measures (pointers are offsets, never dereferenced).
struct Arena {
  char* base = nullptr;
  size_t off = 256, cap = 0;
  bool dry = true;
  template <class T>
  bool take(T** p, size_t count) {
    const size_t bytes = (count * sizeof(T) + 256 + 255) & ~(size_t)255;  // 256 B of slack behind every buffer (vector loads may run over)
    if (!dry && off + bytes > cap) return false;
    *p = (T*)(base + off);
    off += bytes;
    return true;
  }
};

constexpr int kMaxRings = 64;        // N_SCANS upper bound (scan_registration.cpp:195-226)
constexpr int kSectors = 6;          // scan_registration.cpp:317
constexpr int kMaxRingLen = 4096;    // points of one ring kept in LDS by k_sr_ring (HDL-64E: <= ~2100)
constexpr int kSectCap = 1024;       // padded sector length for the LDS bitonic sort
constexpr int kLabelBlock = 1024;    // points per workgroup in the label / scatter kernels
constexpr int kMaxSharpPerSect = 2, kMaxLessSharpPerSect = 20, kMaxFlatPerSect = 4;  // scan_registration.cpp:335-345,391
constexpr int kMaxSharp = kMaxRings * kSectors * kMaxSharpPerSect;          // 768
constexpr int kMaxLessSharp = kMaxRings * kSectors * kMaxLessSharpPerSect;  // 7680
constexpr int kMaxFlat = kMaxRings * kSectors * kMaxFlatPerSect;            // 1536
constexpr int kMaxLoFactors = kMaxSharp + kMaxFlat;                         // 2304

// Kernel ids for the per-kernel HIP-event timer (vloam_profile_kernel); names = the __global__ symbols.
enum KernelId : int {
  kKNone = 0, kKSrFirstLast, kKSrLabel, kKSrScan, kKSrScatter, kKSrRing, kKSrCompact, kKLoAssoc, kKLmSolve, kKLoFinish,
  kKMapPrepare, kKMapStack, kKMapAssoc, kKMapInsert, kKMapFinalize, kKVoProject, kKVoMatch,
  kKLoGridCount, kKLoGridScan, kKLoGridScatter, kKMapDsReduce, kKMapFit, kKLmCompact, kKVoFold, kKSrRingBig,
  kKImgSobel, kKImgEig, kKImgLocalMax, kKImgNeighbours, kKImgSelect, kKImgPyrDown, kKImgScharr, kKImgLk, kKLoAssocFast, kKCount
};
static const char* const kKernelNames[kKCount] = {"", "k_sr_first_last", "k_sr_label", "k_sr_scan", "k_sr_scatter", "k_sr_ring",
  "k_sr_compact", "k_lo_assoc", "k_lm_solve", "k_lo_finish", "k_map_prepare", "k_map_ds_bin", "k_map_assoc", "k_map_insert",
  "k_map_finalize", "k_vo_project", "k_vo_match", "k_lo_grid_count", "k_lo_grid_scan", "k_lo_grid_scatter",
  "k_map_ds_reduce", "k_map_fit", "k_lm_compact", "k_vo_fold", "k_sr_ring_big_tier",
  "k_img_sobel", "k_img_eig", "k_img_localmax", "k_img_neighbours", "k_img_select", "k_img_pyrdown", "k_img_scharr", "k_img_lk", "k_lo_assoc_fast"};
constexpr int kKAll = -1;  // ProfHook::id: bracket every launch, whichever kernel

// Records a HIP-event pair around every launch of one selected kernel (or of all kernels), on the stream it is launched on.
struct ProfHook {
  int id = kKNone;
  hipEvent_t* ev = nullptr;  // [2 * cap] start/stop pairs
  int* kid_of = nullptr;     // [cap] kernel id of every recorded pair
  int cap = 0, used = 0;
  inline bool begin(int kid, hipStream_t st) {
    if ((kid != id && id != kKAll) || used >= cap) return false;
    kid_of[used] = kid;
    (void)hipEventRecord(ev[2 * used], st);
    return true;
  }
  inline void end(hipStream_t st) { (void)hipEventRecord(ev[2 * used + 1], st); used++; }
};
// ---- session-major launch geometry.  Every kernel is WRITTEN against a logical grid (x, y, z = session) but LAUNCHED as the hardware
// grid (session, x, y): the dispatcher hands workgroups to the 8 XCDs in linear order (block b -> XCD b % 8, x fastest), so with the
// session in the fastest dimension all workgroups of session s of a batch of 8 run on XCD s (B = 16: two sessions per XCD; B = 2 / 4:
// a session owns every 2nd / 4th XCD; B = 1: the plain round robin).  A session's working set — its NN grids, the hot part of its voxel
// map, its solver's barrier words — then lives in ONE 4 MB L2 instead of competing with seven other sessions in every L2.  Inside the
// kernels `blockIdx` / `gridDim` are redirected to the logical coordinates (the two accessors below are defined BEFORE the macros, so
// they read the hardware values); host code passes logical grids to the launch macros, which transpose them.
struct VlDim3 { unsigned x, y, z; };
#ifdef VLOAM_SPREAD_GEOMETRY   // A/B build (make spread): the plain grid (x, y, session) — every session's workgroups dealt over all eight XCDs (profiles/r05_batch_scaling.txt)
__device__ __forceinline__ VlDim3 vl_block_idx() { VlDim3 r; r.x = blockIdx.x; r.y = blockIdx.y; r.z = blockIdx.z; return r; }
__device__ __forceinline__ VlDim3 vl_grid_dim() { VlDim3 r; r.x = gridDim.x; r.y = gridDim.y; r.z = gridDim.z; return r; }
inline dim3 vl_hw_grid(dim3 g) { return g; }
#else
__device__ __forceinline__ VlDim3 vl_block_idx() { VlDim3 r; r.x = blockIdx.y; r.y = blockIdx.z; r.z = blockIdx.x; return r; }
__device__ __forceinline__ VlDim3 vl_grid_dim() { VlDim3 r; r.x = gridDim.y; r.y = gridDim.z; r.z = gridDim.x; return r; }
inline dim3 vl_hw_grid(dim3 g) { return dim3(g.z, g.x, g.y); }
#endif
#define blockIdx (::vloam::vl_block_idx())
#define gridDim (::vloam::vl_grid_dim())
#define VL_RAW_LAUNCH(kern, grid, ...) hipLaunchKernelGGL(kern, ::vloam::vl_hw_grid(grid), __VA_ARGS__)

#define VLOAM_LAUNCH(ph, kid, st, kern, grid, ...)                            \
  do {                                                                        \
    bool prof_ = (ph) && (ph)->begin((kid), (st));                            \
    hipLaunchKernelGGL(kern, ::vloam::vl_hw_grid(grid), __VA_ARGS__);         \
    if (prof_) (ph)->end((st));                                               \
  } while (0)

// Same, with the stage's "finished" event bound to the dispatch itself (its completion signal) instead of a marker packet
// behind it: a separate hipEventRecord costs ~5 us of idle stream on MI355X, the bound event costs nothing.
extern int g_vl_plain_events;   // VLOAM_PLAIN_EVENTS=1 (A/B switch): a marker packet behind the dispatch instead of the dispatch's own completion signal
#define VLOAM_LAUNCH_EV(ph, kid, st, stop_ev, kern, grid, block, shmem, stream, ...)                        \
  do {                                                                                                      \
    bool prof_ = (ph) && (ph)->begin((kid), (st));                                                          \
    if ((stop_ev) && !::vloam::g_vl_plain_events) hipExtLaunchKernelGGL(kern, ::vloam::vl_hw_grid(grid), block, shmem, stream, nullptr, (stop_ev), 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern, ::vloam::vl_hw_grid(grid), block, shmem, stream, __VA_ARGS__);            \
    if ((stop_ev) && ::vloam::g_vl_plain_events) (void)hipEventRecord((stop_ev), (stream));                 \
    if (prof_) (ph)->end((st));                                                                             \
  } while (0)

enum ErrorBits : int {
  kErrEmpty = 1,       // no point survived S1
  kErrRingTooLong = 2, // a ring exceeded kMaxRingLen
  kErrMapFull = 4,     // voxel hash out of slots
  kErrMapDeferred = 8, // raw-point capacity: more than 255 un-merged points in one voxel of a cube outside the valid block, or more than 64 raw voxels around one query
  kErrStackFull = 16,
  kErrVoDegenerate = 64, // the VO solve returned a zero rotation angle: the reference divides by it (visual_odometry.cpp:427-430) -> NaN poses
  kErrSolverSync = 32,  // k_map_ds_reduce: a workgroup of the scan-feature VoxelGrid gave up waiting for the cell counts of the bins in front of it (that sweep's stack is left EMPTY).  (Until round 4 also the cooperative LM solves' timeout: they degrade to one workgroup instead since round 5.)
};

// Per-frame scalars of scan registration (one per sequence).
struct FrameScalars {
  int first_valid, last_valid;  // raw input indices of the first / last point surviving S1
  int istar;                    // raw index of the kept point that flips halfPassed (INT_MAX: none)
  int n_after_s1;               // debug
  int N2;                       // points kept (== laserCloud->size())
  float startOri, endOri;
  int error;
  int ring_count[kMaxRings];
  int ring_off[kMaxRings + 1];
  int scanStartInd[kMaxRings], scanEndInd[kMaxRings];
  int sect_cnt[kMaxRings][kSectors][3];  // sharp, lessSharp, flat picks per sector
  int ring_ds_cnt[kMaxRings];            // per-ring VoxelGrid(0.2) output size
  int n_sharp, n_less_sharp, n_flat, n_less_flat;
  // bounding box (min x, y, z, max x, y, z) of the points every scan line contributes to cornerPointsLessSharp [0] / surfPointsLessFlat [1]
  // (k_sr_compact; an empty line holds +FLT_MAX / -FLT_MAX): getMinMax3D of the clouds pcl::VoxelGrid is run on in the mapping stage
  float less_bbox[2][kMaxRings][6];
};

// ---------------------------------------------------------------- Levenberg–Marquardt
// One record per solve, written by k_lm_solve.  Plain doubles so tests can read it verbatim.
constexpr int kLmMaxTrace = 104;
struct LMRecord {
  double x_in[7], x_out[7];
  double H0[36], g0[6];          // J^T J and J^T r at the initial point (tangent space, after the loss corrector)
  double initial_cost, final_cost;
  double n_iterations;           // rows of trace that are valid
  double termination;            // 0 iteration cap, 1 convergence, 2 failure
  double n_factors;              // residual blocks in the problem
  double n_evals;                // residual/Jacobian evaluations performed (E_o / E_m of SURVEY §8d)
  double cyc[4];                 // shader-clock cycles: prologue (compaction), evaluations, thread-0 LM bookkeeping, whole kernel
  double trace[kLmMaxTrace][8];  // cost, cost_change, gradient_max_norm, step_norm, relative_decrease, radius, valid, successful
};

// Factor table consumed by k_lm_solve (structure of arrays, one slot per candidate feature).
//   type 0: none
//   type 1: LidarEdgeFactor       p = curr, A = last_point_a, B = last_point_b     (lidarFactor.hpp:14-56)
//   type 2: LidarPlaneFactor      p = curr, A = last_point_j, B = ljm_norm         (lidarFactor.hpp:58-106)
//   type 3: LidarPlaneNormFactor  p = curr, A = plane_unit_norm, B.x = negative_OA_dot_norm (lidarFactor.hpp:108-139)
//   type 4: CostFunctor32         p = X0, A.x,A.y = x1_bar,y1_bar                  (ceres_cost_function.h:54-96)
//   type 5: CostFunctor22         p.x,p.y = x0_bar,y0_bar, A.x,A.y = x1_bar,y1_bar (ceres_cost_function.h:147-185)
struct FactorTable {
  int* type;      // [cap]
  double* p;      // [3][cap]
  double* A;      // [3][cap]
  double* B;      // [3][cap]
  double* resid;  // [3][cap] raw residuals at the initial point (parity hook)
  int* ctype;     // [cap]    k_lm_solve's compacted copy (accepted factors only, slot order)
  int* cslot;     // [cap]    original slot of every compacted factor
  double* cpack;  // [11][cap] compacted factors: p, then (e1, e2, d1, d2) of an edge / (n, d) of a plane / A, B otherwise
  double* dg;     // optional [8][cap], by SLOT: what the solve's evaluation consumes of a LiDAR factor — (e1, e2, d1, d2) of an edge, (n, d) of a plane —
                  // written by the kernel that emits the factor (k_lo_assoc*, k_map_fit: factor_digest below), so that the solve's first evaluation is one
                  // memory trip instead of type -> raw points -> two square roots and six divisions on the bounding chain
  int* rowcnt;    // [cap / 64] accepted factors per 64-slot row (atomicAdd by the association kernels, zeroed by k_lm_solve)
  unsigned long long* rowmask;  // optional [cap / 64 (+ 2)]: accepted slots of every 64-slot row as a bit mask, rewritten in full by the producer
                               // (k_map_fit) for every solve: the solve then compacts on its own and k_lm_compact is not launched (lm_solve.hip, kLmRowMask)
  int cap;
  double* gsync;  // optional [kLmSyncDoubles]: poison word + the tagged partial sums the workgroups of a cooperative solve exchange (null: one workgroup)
  int* err;       // optional sticky error word (ErrorBits) the host polls in vloam_sync
  int* fallbacks; // optional counter: cooperative solves that degraded to one workgroup (k_lm_solve; vloam_get_health)
  int* host_degraded;  // optional HOST-MAPPED word (not in the arenas, never rebased): set when a cooperative solve degrades; the host polls it before every enqueue
  unsigned gen;   // generation of the launch (lm_launch): the tag of everything the solve's workgroups exchange
  int spin_limit; // polls a workgroup of a cooperative solve waits for its partners before it gives up (lm_launch; VLOAM_LM_SPIN_LIMIT)
  __host__ __device__ void rebase(size_t off) {
    rbp(type, off); rbp(p, off); rbp(A, off); rbp(B, off); rbp(resid, off); rbp(ctype, off); rbp(cslot, off); rbp(cpack, off); rbp(dg, off);
    rbp(rowcnt, off); rbp(rowmask, off); rbp(gsync, off); rbp(err, off); rbp(fallbacks, off);
  }
};
// Line through a, b -> what the solver's edge evaluation consumes: an orthonormal pair (e1, e2) with e1 x e2 = v = (b - a) / |a - b|
// (lidarFactor.hpp:37-42 divides by de.norm()), e1 = normalize(v x axis of the smallest |v| component), e2 = v x e1, and the
// offsets d_i = -(e_i . a).  out = e1, e2, d1, d2.  (r = ((lp - a) x (lp - b)) / |a - b| = c1 e2 - c2 e1 with c_i = e_i . lp + d_i: lm_solve.hip.)
__device__ __forceinline__ void edge_frame(double ax, double ay, double az, double bx, double by, double bz, double (&out)[8]) {
  const double dx = bx - ax, dy = by - ay, dz = bz - az;
  const double dn = sqrt(dx * dx + dy * dy + dz * dz);
  const double vx = dx / dn, vy = dy / dn, vz = dz / dn;
  const double fx = fabs(vx), fy = fabs(vy), fz = fabs(vz);
  double e1x, e1y, e1z;
  if (fx <= fy && fx <= fz) { e1x = 0.0; e1y = vz; e1z = -vy; }        // v x (1, 0, 0)
  else if (fy <= fz) { e1x = -vz; e1y = 0.0; e1z = vx; }              // v x (0, 1, 0)
  else { e1x = vy; e1y = -vx; e1z = 0.0; }                             // v x (0, 0, 1)
  const double en = sqrt(e1x * e1x + e1y * e1y + e1z * e1z);
  e1x /= en; e1y /= en; e1z /= en;
  const double e2x = vy * e1z - vz * e1y, e2y = vz * e1x - vx * e1z, e2z = vx * e1y - vy * e1x;
  out[0] = e1x; out[1] = e1y; out[2] = e1z; out[3] = e2x; out[4] = e2y; out[5] = e2z;
  out[6] = -(e1x * ax + e1y * ay + e1z * az);
  out[7] = -(e2x * ax + e2y * ay + e2z * az);
}
// The digest of the factor just written to slot `slot` of F (type 1: A = last_point_a, B = last_point_b; type 2: A = last_point_j, B = ljm_norm;
// type 3: A = plane_unit_norm, B.x = negative_OA_dot_norm) -> F.dg.  One lane.
__device__ __forceinline__ void factor_digest(const FactorTable& F, int slot, int type, const double (&A)[3], const double (&B)[3]) {
  if (!F.dg) return;
  const int cap = F.cap;
  if (type == 1) {
    double fr[8];
    edge_frame(A[0], A[1], A[2], B[0], B[1], B[2], fr);
#pragma unroll
    for (int q = 0; q < 8; q++) F.dg[q * cap + slot] = fr[q];
  } else if (type == 2) {   // (lp - j) . n  ->  n . lp + d with d = -(n . j)
    F.dg[slot] = B[0]; F.dg[cap + slot] = B[1]; F.dg[2 * cap + slot] = B[2]; F.dg[3 * cap + slot] = -(B[0] * A[0] + B[1] * A[1] + B[2] * A[2]);
  } else if (type == 3) {   // n . lp + negative_OA_dot_norm
    F.dg[slot] = A[0]; F.dg[cap + slot] = A[1]; F.dg[2 * cap + slot] = A[2]; F.dg[3 * cap + slot] = B[0];
  }
}

constexpr int kLmMaxBlocks = 8;                          // workgroups a cooperative solve may use
constexpr int kLmSyncDoubles = 8 + 2 * kLmMaxBlocks * 64;  // generation, poison word (+ spare) | [parity][workgroup][2][32] tagged 8-byte granules of the partial accumulators

// ---------------------------------------------------------------- vloam_tf blackboard (coupled VO <-> LiDAR odometry loop)
// tf2::Transform restated: row-major 3x3 basis + origin, double precision, the same operation order as tf2/LinearMath
// (Transform::operator*, inverse(), Matrix3x3::setRotation / getRotation, Quaternion::setRotation(axis, angle) / getAxis / getAngle).
struct TfDev { double m[9]; double o[3]; };
__host__ __device__ inline void tf_identity(TfDev* t) { for (int k = 0; k < 9; k++) t->m[k] = (k % 4 == 0) ? 1.0 : 0.0; t->o[0] = t->o[1] = t->o[2] = 0.0; }
__host__ __device__ inline void tf_mul(const TfDev& a, const TfDev& b, TfDev* r) {  // (a.m b.m, a.m b.o + a.o)
  TfDev t;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t.m[i * 3 + j] = (a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j]) + a.m[i * 3 + 2] * b.m[6 + j];
  for (int i = 0; i < 3; i++) t.o[i] = ((a.m[i * 3] * b.o[0] + a.m[i * 3 + 1] * b.o[1]) + a.m[i * 3 + 2] * b.o[2]) + a.o[i];
  *r = t;
}
__host__ __device__ inline void tf_inverse(const TfDev& a, TfDev* r) {              // (m^T, m^T * -o)
  TfDev t;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t.m[i * 3 + j] = a.m[j * 3 + i];
  const double n[3] = {-a.o[0], -a.o[1], -a.o[2]};
  for (int i = 0; i < 3; i++) t.o[i] = (t.m[i * 3] * n[0] + t.m[i * 3 + 1] * n[1]) + t.m[i * 3 + 2] * n[2];
  *r = t;
}
__host__ __device__ inline void tf_set_rotation(TfDev* t, const double q[4]) {      // Matrix3x3::setRotation
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double d = ((x * x + y * y) + z * z) + w * w;
  const double s = 2.0 / d;
  const double xs = x * s, ys = y * s, zs = z * s;
  const double wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
  t->m[0] = 1.0 - (yy + zz); t->m[1] = xy - wz; t->m[2] = xz + wy;
  t->m[3] = xy + wz; t->m[4] = 1.0 - (xx + zz); t->m[5] = yz - wx;
  t->m[6] = xz - wy; t->m[7] = yz + wx; t->m[8] = 1.0 - (xx + yy);
}
__host__ __device__ inline void tf_get_rotation(const TfDev& t, double q[4]) {      // Matrix3x3::getRotation
  const double* m = t.m;
  const double trace = (m[0] + m[4]) + m[8];
  if (trace > 0.0) {
    double s = sqrt(trace + 1.0);
    q[3] = s * 0.5;
    s = 0.5 / s;
    q[0] = (m[7] - m[5]) * s; q[1] = (m[2] - m[6]) * s; q[2] = (m[3] - m[1]) * s;
  } else {
    const int i = m[0] < m[4] ? (m[4] < m[8] ? 2 : 1) : (m[0] < m[8] ? 2 : 0);
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    double s = sqrt(((m[i * 4] - m[j * 4]) - m[k * 4]) + 1.0);
    q[i] = s * 0.5;
    s = 0.5 / s;
    q[3] = (m[k * 3 + j] - m[j * 3 + k]) * s;
    q[j] = (m[j * 3 + i] + m[i * 3 + j]) * s;
    q[k] = (m[k * 3 + i] + m[i * 3 + k]) * s;
  }
}

struct VloamTfState {            // the slice of vloam::VloamTF the per-frame loop reads and writes (vloam_tf.h:34-49)
  TfDev base_T_cam0, velo_T_cam0;          // static extrinsics (vloam_tf.cpp:55-56), vloam_set_extrinsics
  TfDev cam0_curr_T_cam0_last;             // VisualOdometry's result as a transform (visual_odometry.cpp:425-430); identity before the first solve
  TfDev cam0_curr_LOT_cam0_prev;           // LaserOdometry::publish (laser_odometry.cpp:563-567) -> solveNlsAll's initial guess
  TfDev world_VOT_base_last;               // VO2VeloAndBase accumulates it (vloam_tf.cpp:67-72)
  int coupled;                             // 1 once the extrinsics are set: LaserOdometry::publish maintains cam0_curr_LOT_cam0_prev
  int vo_nan_frames;                       // frames whose VO result had a zero rotation angle (NaN transform in the reference, :427-430)
};

struct LOState {           // laser odometry state carried across frames (laser_odometry.h:106-146)
  double para_q[4], para_t[3];  // q_last_curr (x,y,z,w), t_last_curr
  double q_w_curr[4], t_w_curr[3];
  double prior_q[4], prior_t[3];  // vloam_tf->velo_last_VOT_velo_curr
  VloamTfState tf;
};

// LaserOdometry::publish, laser_odometry.cpp:560-567: base_prev_LOT_base_curr = (q_last_curr, t_last_curr);
// cam0_curr_LOT_cam0_prev = base_T_cam0^-1 * base_prev_LOT_base_curr^-1 * base_T_cam0.  x = (q xyzw, t) as just solved.
__device__ inline void tf_lo_publish(LOState* lo, const double* x) {
  if (!lo->tf.coupled) return;
  TfDev b, bi, ci, r;
  tf_set_rotation(&b, x);
  b.o[0] = x[4]; b.o[1] = x[5]; b.o[2] = x[6];
  tf_inverse(b, &bi);
  tf_inverse(lo->tf.base_T_cam0, &ci);
  tf_mul(ci, bi, &r);
  tf_mul(r, lo->tf.base_T_cam0, &r);
  lo->tf.cam0_curr_LOT_cam0_prev = r;
}

struct MapState {          // laser mapping state (laser_mapping.h:141-155)
  double parameters[7];    // q_w_curr (x,y,z,w), t_w_curr
  double q_wmap_wodom[4], t_wmap_wodom[3];
  double q_wodom_curr[4], t_wodom_curr[3];
  int cenW, cenH, cenD;    // laserCloudCen{Width,Height,Depth}
  int centerCube[3];
  int n_corner_stack, n_surf_stack;
  int do_optimize;         // laserCloudCornerFromMapNum > 10 && laserCloudSurfFromMapNum > 50
  int n_map_corner, n_map_surf;  // points in the valid 5x5x3 block at gather time
  int deferred;            // voxels that turned raw (points arriving in a cube outside the valid block) since the start
  int sweep_no;            // mapped sweeps so far (arrival stamps of raw points)
};

}  // namespace vloam
