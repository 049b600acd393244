// Host side of libvloam_hip.so: the C ABI declared in include/vloam_hip/c_api.h.
// One handle = one HIP device, B sequences advanced in lock step (B = 1 unless vloam_create_batch), one in-order stream per façade stage
// (scan registration, odometry, mapping) plus one for the mapping stage's scan-feature VoxelGrid;
// every per-frame count stays in HBM, so a sweep is a fixed chain of kernel launches with no host synchronisation until
// the caller asks for results (the only host waits are the back-pressure on buffer sets that are still being read).
// There is NO CPU fallback: without a usable HIP device vloam_create fails.
#include "../../include/vloam_hip/c_api.h"

#include <hip/hip_runtime.h>
#include <atomic>
#include <limits.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "lm_solve.h"
#include "lo_kernels.h"
#include "map_kernels.h"
#include "sr_kernels.h"
#include "vloam_device.h"
#include "vo_kernels.h"
#include "img_kernels.h"

using namespace vloam;

static thread_local std::string g_err;
static const bool g_host_prof = getenv("VLOAM_HOST_PROF") != nullptr;
namespace vloam { int g_vl_plain_events = getenv("VLOAM_PLAIN_EVENTS") ? atoi(getenv("VLOAM_PLAIN_EVENTS")) : 0; }
static const int g_enqueue_order = getenv("VLOAM_ENQUEUE_ORDER") ? atoi(getenv("VLOAM_ENQUEUE_ORDER")) : 0;
static const int g_stage_inline = getenv("VLOAM_STAGE_INLINE") ? atoi(getenv("VLOAM_STAGE_INLINE")) : 0;   // 1: vloam_process_scan / vloam_batch_process_scan stage inline too (no deferred ring)
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void set_err(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}
#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);     \
      return VLOAM_ERR_HIP;                                                                   \
    }                                                                                         \
  } while (0)

struct vloam_handle {
  vloam_config cfg;
  int device = 0;
  // Three in-order streams, one per stage of the façade: scan registration of sweep k + 1 overlaps laser odometry of sweep k
  // and laser mapping of sweep k - 1 (each stage only needs the previous stage's result of the SAME sweep plus its own state
  // of the previous sweep).  Cross-stage edges are HIP events; SR output lives in kSets rotating buffer sets so that a stage
  // running ahead never overwrites what a slower stage still reads.
  hipStream_t stream = nullptr;   // scan registration (+ NN grid build); also creation / VO work
  hipStream_t s_lo = nullptr;     // laser odometry
  hipStream_t s_map = nullptr;    // laser mapping
  hipStream_t s_img = nullptr;    // image front-end of the coupled frame loop (needs the image only: next to the scan-registration stream)
  hipStream_t s_ds = nullptr;     // VoxelGrid of the scan features for mapping (needs the sweep's feature clouds only: off the SR stream's chain)
  static constexpr int kSets = kBufferSets;   // 3 suffice for correctness; the rest is run-ahead for the host (vloam_device.h)
  hipEvent_t ev_sr[kSets] = {}, ev_lo[kSets] = {}, ev_map[kSets] = {}, ev_stack[kSets] = {};  // "stage finished for the sweep in set c"
  static_assert(kSets == MapContext::kSets, "the stack sets rotate with the SR buffer sets");
  // Device memory: B session arenas of identical layout, `se.ss` bytes apart, in ONE allocation; every device pointer below is
  // session 0's (kernels add blockIdx.z * se.ss, host-side getters add sel * se.ss)
  char* arena = nullptr;
  size_t arena_bytes = 0;
  Sess se;
  int sel = 0;          // session the getters read (vloam_select_session)
  double* sync_pool = nullptr;
  bool counted_single = false; // this handle is in g_single_handles
  int* ring_watch = nullptr;   // host-mapped [kMaxBatch]: a ring of that session came near the small ring tier's capacity (k_sr_ring)
  int* coop_flag = nullptr;    // host-mapped [1]: a cooperative solve of this handle degraded to one workgroup (k_lm_solve) — polled before every enqueue
  long long fallback_solves = 0;   // cooperative solves that degraded to one workgroup, as of the last vloam_sync
  int frame = 0;        // sweeps accepted (scan registration enqueued)
  int lo_done = 0;      // sweeps whose laser odometry has been enqueued (vloam_process_scan defers it, see drain_deferred)
  int map_done = 0;     // sweeps whose laser mapping has been enqueued
  int stage = 0;        // façade order inside a sweep: 0 idle, 1 after SR, 2 after LO
  int nblk_max = 0;
  // scan registration
  // Host sweeps.  vloam_process_scan / vloam_batch_process_scan (the asynchronous whole-sweep calls): a copy stream + a ring of kInRing device
  // input buffers, and the sweep itself is ENQUEUED BY THE NEXT CALL (like its odometry and mapping: drain_deferred) — by then its copy has
  // landed, so nothing on the device ever waits for the host link and no stream waits for another: the copy of sweep k + 1 runs beside the scan
  // registration of sweep k.  ev_in_copied: the copy into the slot has landed (checked by the HOST); in_reader: the event behind the slot's last
  // reader (the sweep's own "scan registration finished" event; checked by the HOST before the slot is copied into again, three calls later).
  // Every other host-pointer entry point (stage-wise calls, frames) stages INLINE: hipMemcpyAsync on the scan-registration stream into a
  // buffer of its own (slot kInRing), the stream's order being the dependency.
  static constexpr int kInRing = 4;
  float4* d_in = nullptr;         // [kInRing + 1][max_points]
  hipStream_t s_copy = nullptr;   // created by the first deferred host sweep
  hipEvent_t ev_in_copied[kInRing] = {};
  hipEvent_t in_reader[kInRing] = {};
  int in_next = 0;                // next ring slot
  struct { bool valid = false; BatchIn bi; int slot = -1; } pend;   // the host sweep whose copy is in flight: enqueued by the next call / any drain
  SRBuffers sr[kSets];  // rotating sets: S, cloud and the feature clouds are per set, the scratch arrays are shared (SR stream only)
  // clouds / odometry pose the caller handed to LaserMapping::input instead of the odometry's own (vloam_set_mapping_input): the mapping of
  // that one sweep reads them from here, the odometry keeps its CornerLast / SurfLast (the reference's stages hold separate copies)
  SRBuffers sub{};             // S, cloud, less_sharp, less_flat only
  double* sub_row = nullptr;   // [14] trajectory row of the substituted sweep: odometry pose in, map pose out
  int sub_cloud_frame = -1, sub_pose_frame = -1;
  // laser odometry
  LOState* lo = nullptr;
  FactorTable lo_F{};
  int* lo_corr[2] = {nullptr, nullptr};
  int* lo_queue = nullptr;     // [kMaxLoFactors] queries k_lo_assoc_fast left to the wave-per-query pass
  int* lo_queue_n = nullptr;   // [2] entries of the queue, by launch parity
  int lo_launches = 0;
  long long* lo_cyc[2] = {nullptr, nullptr};  // debug: per-slot shader-clock cycles of k_lo_assoc (NN, walks, emit, exact radius)
  LMRecord* lo_rec = nullptr;  // [2]
  double* lo_resid[2] = {nullptr, nullptr};
  double* traj = nullptr;      // [max_frames][14]
  double* vo_traj = nullptr;   // [max_frames][7] world_VOT_base_last per frame (coupled frame loop)
  hipEvent_t ev_vo[kSets] = {};     // depth map + matches of the frame in set c are in HBM
  bool vo_frame[kSets] = {};        // the sweep in set c came through vloam_process_frame (its odometry is preceded by the VO solve)
  bool have_extrinsics = false;
  ImgContext img;
  std::vector<unsigned char> img_pack;   // host scratch: a padded image packed to width-stride rows
  hipEvent_t ev_img[kSets] = {};    // the image-derived matches of the frame in set c are in HBM
  bool img_frame[kSets] = {};
  LoGrid grid[kSets];          // per set: NN grid over that sweep's lessSharp / lessFlat
  // mapping + vo
  MapContext map;
  VOContext vo;
  // timing
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // cfg.timing: begin / end of SR, LO, mapping
  double stage_ms[4] = {0, 0, 0, 0};
  int timed_scans = 0;
  int last_n_in = 0;
  ProfHook prof;
  // VLOAM_HOST_PROF=1: host seconds spent inside the enqueue helpers (printed by vloam_destroy): throttle waits | SR (+ grids, scan-feature VoxelGrid) | LO | mapping
  double host_s[4] = {0, 0, 0, 0};
  long long host_calls = 0;
  std::vector<hipEvent_t> prof_events;
  std::vector<int> prof_kids;
};

#define TAKE(p, n) do { if (!A.take(&(p), (size_t)(n))) { set_err("session arena too small (internal)"); return VLOAM_ERR_HIP; } } while (0)

static vloam_status take_factor_table(Arena& A, FactorTable* F, int cap) {
  F->cap = cap;
  TAKE(F->type, cap);
  TAKE(F->p, 3 * (size_t)cap);
  TAKE(F->A, 3 * (size_t)cap);
  TAKE(F->B, 3 * (size_t)cap);
  TAKE(F->resid, 3 * (size_t)cap);
  TAKE(F->ctype, cap);
  TAKE(F->cslot, cap);
  TAKE(F->cpack, 11 * (size_t)cap);
  TAKE(F->dg, 8 * (size_t)cap);
  TAKE(F->rowcnt, (size_t)cap / 64 + 1);
  F->rowmask = nullptr;  // the odometry / VO tables use the per-row counters
  F->gsync = nullptr;  // placed by lm_sync_calibrate once everything is allocated
  F->err = nullptr;    // set once the mapping context (owner of the sticky error word) exists
  F->fallbacks = nullptr; F->host_degraded = nullptr; F->gen = 0; F->spin_limit = 1 << 18;
  return VLOAM_OK;
}

constexpr int kSyncCand = 48;          // candidate cache lines for the sync words of the cooperative solves
constexpr size_t kSyncStride = 8448;   // 8 KB + 256 B: walks page and sub-page address bits; >= one slot
static_assert(kSyncStride >= kLmSyncDoubles * sizeof(double), "slots must not overlap");

// Carve one session's device state out of its arena.  Runs twice: dry (A.base == nullptr) to measure, then for real.
static vloam_status handle_layout(vloam_handle* h, Arena& A) {
  const vloam_config* cfg = &h->cfg;
  const int P = cfg->max_points;
  h->nblk_max = (P + kLabelBlock - 1) / kLabelBlock;
  TAKE(h->d_in, (size_t)(vloam_handle::kInRing + 1) * (size_t)P);   // the ring of the deferred host sweeps + the inline staging buffer
  SRBuffers& a = h->sr[0];
  TAKE(a.sid, (size_t)P);
  TAKE(a.ori, (size_t)P);
  TAKE(a.blockhist, (size_t)h->nblk_max * kMaxRings);
  TAKE(a.blockoff, (size_t)h->nblk_max * kMaxRings);
  TAKE(a.sharp_idx, kMaxSharp);
  TAKE(a.less_sharp_idx, kMaxLessSharp);
  TAKE(a.flat_idx, kMaxFlat);
  TAKE(a.ring_ds, (size_t)kMaxRings * kMaxRingLen);
  TAKE(a.dbg_curv, (size_t)P);
  TAKE(a.dbg_sort, (size_t)P);
  TAKE(a.dbg_picked, (size_t)P);
  TAKE(a.dbg_label, (size_t)P);
  TAKE(a.dbg_cyc, (size_t)kMaxRings * 8);
  TAKE(a.dbg_feat_idx, 3 * kMaxLessSharp);
  for (int k = 1; k < vloam_handle::kSets; k++) h->sr[k] = a;
  for (int k = 0; k < vloam_handle::kSets; k++) {
    TAKE(h->sr[k].S, 1);
    TAKE(h->sr[k].cloud, (size_t)P);
    TAKE(h->sr[k].sharp, kMaxSharp);
    TAKE(h->sr[k].flat, kMaxFlat);
    TAKE(h->sr[k].less_sharp, kMaxLessSharp);
    TAKE(h->sr[k].less_flat, (size_t)P);
  }
  for (int k = 0; k < vloam_handle::kSets; k++) {
    TAKE(h->grid[k].occ, 4 * kMaxRings);
    TAKE(h->grid[k].stops, 4 * kStopLen);
    for (int g = 0; g < 4; g++) {
      h->grid[k].mask[g] = kGridBuckets[g] - 1;
      TAKE(h->grid[k].cnt[g], kGridBuckets[g]);
      TAKE(h->grid[k].start[g], kGridBuckets[g] + 2);
      TAKE(h->grid[k].pts[g], (g & 1) ? (size_t)P : (size_t)kMaxLessSharp);
    }
  }
  TAKE(h->sub.S, 1);
  TAKE(h->sub.cloud, (size_t)P);
  TAKE(h->sub.less_sharp, kMaxLessSharp);
  TAKE(h->sub.less_flat, (size_t)P);
  TAKE(h->sub_row, 14);
  TAKE(h->lo, 1);
  vloam_status s = take_factor_table(A, &h->lo_F, kMaxLoFactors);
  if (s != VLOAM_OK) return s;
  for (int k = 0; k < 2; k++) { TAKE(h->lo_corr[k], kMaxLoFactors * 4); TAKE(h->lo_resid[k], 3 * kMaxLoFactors); if (h->cfg.debug) TAKE(h->lo_cyc[k], 4 * kMaxLoFactors); }
  TAKE(h->lo_rec, 2);
  TAKE(h->lo_queue, kMaxLoFactors);
  TAKE(h->lo_queue_n, 2);
  TAKE(h->traj, (size_t)cfg->max_frames * 14);
  TAKE(h->vo_traj, (size_t)cfg->max_frames * 7);
  if (map_layout(&h->map, h->cfg, A) != VLOAM_OK) { set_err("map_layout failed"); return VLOAM_ERR_HIP; }
  TAKE(h->sync_pool, kSyncCand * kSyncStride / sizeof(double));
  if (vo_layout(&h->vo, h->cfg, A) != VLOAM_OK) { set_err("vo_layout failed"); return VLOAM_ERR_HIP; }
  if (img_layout(&h->img, h->cfg, A) != VLOAM_OK) { set_err("img_layout failed"); return VLOAM_ERR_HIP; }
  h->lo_F.err = &h->map.frame->error;
  h->lo_F.fallbacks = &h->map.frame->fallback_solves;
  for (int k = 0; k < vloam_handle::kSets; k++) h->sr[k].sticky_err = &h->map.frame->error;
  return VLOAM_OK;
}

static std::atomic<int> g_single_handles{0};   // single-sequence handles alive in this process (Sess::crowd)
#define SINGLE_SESSION_ONLY(h) do { if ((h)->se.B != 1) { set_err("this entry point drives one sequence: the handle has %d sessions (use the vloam_batch_* calls)", (h)->se.B); return VLOAM_ERR_INVALID; } } while (0)

// session-relative pointer for the host-side getters
template <class T>
static inline T* SEL(const vloam_handle* h, T* p) { return p ? (T*)((char*)p + (size_t)h->sel * h->se.ss) : p; }

extern "C" {

void vloam_default_config(vloam_config* c) {
  memset(c, 0, sizeof(*c));
  c->scan_line = 64;
  c->minimum_range = 5.0;
  c->mapping_skip_frame = 1;
  c->mapping_line_resolution = 0.4f;
  c->mapping_plane_resolution = 0.8f;
  c->detach_VO_LO = 1;
  c->reset_VO_to_identity = 0;
  c->remove_VO_outlier = 100;
  c->with_mapping = 1;
  c->max_points = 262144;
  c->max_frames = 8192;
  c->map_capacity_log2 = 22;
  c->debug = 0;
  c->timing = 0;
  c->image_width = 0;
  c->image_height = 0;
  c->CLAHE = 0;
}

const char* vloam_last_error(void) { return g_err.c_str(); }
const char* vloam_version(void) { return "vloam_hip 0.1 (gfx950)"; }

vloam_status vloam_create_batch(const vloam_config* cfg, int device, int n_sessions, vloam_handle** out) {
  if (!cfg || !out) { set_err("null argument"); return VLOAM_ERR_INVALID; }
  if (n_sessions < 1 || n_sessions > kMaxBatch) { set_err("n_sessions must be 1..%d", kMaxBatch); return VLOAM_ERR_INVALID; }
  if (cfg->scan_line != 16 && cfg->scan_line != 32 && cfg->scan_line != 64) {
    set_err("only support velodyne with 16, 32 or 64 scan line!");  // scan_registration.cpp:54-58
    return VLOAM_ERR_INVALID;
  }
  if (cfg->max_points < 64 || cfg->max_points > (1 << 24) || cfg->max_frames < 1 || cfg->mapping_skip_frame < 1) {  // 24-bit point tags
    set_err("bad capacity"); return VLOAM_ERR_INVALID;
  }
  // laser_mapping.cpp:95-101 takes any leaf; the reference's launch files use 0.2 / 0.4 (VLP-16, HDL-32) and 0.4 / 0.8 (KITTI).  Here the
  // position of a voxel in the gathered map cloud (the 5-NN tie rank) is a 32-bit mixed-radix number: 75 cubes x radix^3 voxels.
  for (const float leaf : {cfg->mapping_line_resolution, cfg->mapping_plane_resolution}) {
    const double nv = leaf > 0.f ? (double)vox_radix(1.0f / leaf) : 1e9;
    if (!(leaf > 0.f) || 75.0 * nv * nv * nv >= 4294967295.0) { set_err("mapping resolutions below 0.132 m are not supported"); return VLOAM_ERR_INVALID; }
  }
  if (cfg->image_width < 0 || cfg->image_height < 0 || (long long)cfg->image_width * cfg->image_height > (1ll << 24) ||
      ((cfg->image_width > 0) != (cfg->image_height > 0)) || (cfg->image_width > 0 && (cfg->image_width < 2 * kImgWin || cfg->image_height < 2 * kImgWin))) {
    set_err("image_width x image_height must be 0 x 0 (no image front-end) or between %d x %d and 2^24 pixels", 2 * kImgWin, 2 * kImgWin); return VLOAM_ERR_INVALID;
  }
  // A handle drives four to six HIP streams that must run side by side (scan registration | scan-feature VoxelGrid | odometry | mapping
  // [| images] [| host-input copies]); the runtime maps ALL streams of the process onto GPU_MAX_HW_QUEUES hardware queues (default 4) and two
  // stages sharing a queue serialise — measured: 214 us per sweep instead of 164 with 4 queues; with 8 queues the same happens as soon as
  // the host process keeps four streams of its own alive (6 400 -> 4 700 scans/s, 3 800 with five; 16 queues: 6 300 with any number,
  // profiles/r05_hw_queues.txt).  The variable is read when the HIP runtime initialises, so it belongs to the HOST's environment
  // (the Python package and bench.py export 16 before they load the runtime): a library must not setenv() behind a
  // multi-threaded host's back (not thread-safe against a concurrent getenv, and silently without effect once HIP is up).
  // Said once per process instead.
  {
    static std::atomic<bool> warned{false};   // (handles may be created from several threads)
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    if ((!q || atoi(q) < 8) && !warned.exchange(true)) {
      fprintf(stderr, "libvloam_hip: GPU_MAX_HW_QUEUES is %s: the stage streams of a handle will share hardware queues and a sweep takes ~30 %% longer; "
                      "export GPU_MAX_HW_QUEUES=16 before the process initialises HIP\n", q ? q : "unset (runtime default 4)");
    }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_err("no HIP device visible: libvloam_hip has no CPU fallback");
    return VLOAM_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) { set_err("device %d out of range (%d visible)", device, ndev); return VLOAM_ERR_NO_DEVICE; }
  HIPCHK(hipSetDevice(device));
  {
    // co-residency of the cooperative solves (c_api.h): 4 + 6 workgroups per session may have to be resident at once
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (10 * n_sessions > prop.multiProcessorCount) {
      set_err("%d sessions need %d co-resident solver workgroups, the device has %d compute units", n_sessions, 10 * n_sessions, prop.multiProcessorCount);
      return VLOAM_ERR_CAPACITY;
    }
  }
  vloam_handle* h = new vloam_handle;
  h->cfg = *cfg;
  h->device = device;
  *out = nullptr;
  vloam_status st = VLOAM_OK;
  do {
    {
      // VLOAM_STREAM_PRIO = "sr,lo,map,ds" (0 = default priority, 1 = the device's highest, -1 = its lowest): which stage's workgroups the
      // dispatcher places first when several stages have work pending (batched handles fill the chip; a single sequence does not)
      int lo_p = 0, hi_p = 0, pr[5] = {0, 0, 0, 0, 0};   // (a fifth field: the image front-end's stream)
      if (const char* e = getenv("VLOAM_STREAM_PRIO")) sscanf(e, "%d,%d,%d,%d,%d", &pr[0], &pr[1], &pr[2], &pr[3], &pr[4]);
      if (hipDeviceGetStreamPriorityRange(&lo_p, &hi_p) != hipSuccess) { lo_p = hi_p = 0; }   // lo_p = numerically greatest = lowest priority
      // VLOAM_RESERVE_CUS = "n[,stride]" (experiment): the scan-registration, odometry and VoxelGrid streams are created with a CU mask that
      // leaves n compute units (every stride-th bit from 0) to the mapping stream alone
      int rsv = 0, rstride = 1;
      if (const char* e = getenv("VLOAM_RESERVE_CUS")) sscanf(e, "%d,%d", &rsv, &rstride);
      hipDeviceProp_t prop;
      const int ncu = (rsv > 0 && hipGetDeviceProperties(&prop, device) == hipSuccess) ? prop.multiProcessorCount : 0;
      auto mk = [&](hipStream_t* s, int which) {
        if (ncu > 0 && which != 2) {
          uint32_t mask[16];
          for (int w = 0; w < 16; w++) mask[w] = 0;
          for (int c = 0; c < ncu && c < 512; c++) mask[c >> 5] |= 1u << (c & 31);
          for (int k = 0, c = 0; k < rsv && c < ncu; k++, c += rstride) mask[c >> 5] &= ~(1u << (c & 31));
          return hipExtStreamCreateWithCUMask(s, (uint32_t)((ncu + 31) / 32), mask) == hipSuccess;
        }
        const int p = pr[which] > 0 ? hi_p : (pr[which] < 0 ? lo_p : 0);
        return hipStreamCreateWithPriority(s, hipStreamNonBlocking, p) == hipSuccess;
      };
      // The copy stream of the deferred host-sweep ring comes FIRST and is used once before any other stream of the handle has work: measured
      // (tools/host_input_probe.py, extring / extring_late) a copy stream that gets its hardware queue after the compute streams runs host-fed
      // sequences at 3 900 - 4 600 scans/s, one that got it before them at 5 650.
      if (!g_stage_inline && !cfg->timing) {
        if (hipStreamCreateWithFlags(&h->s_copy, hipStreamNonBlocking) != hipSuccess) { set_err("hipStreamCreate failed"); st = VLOAM_ERR_HIP; break; }
        static int warm_src = 0;
        int* warm_dst = nullptr;
        if (hipMalloc(&warm_dst, sizeof(int)) == hipSuccess) {
          (void)hipMemcpyAsync(warm_dst, &warm_src, sizeof(int), hipMemcpyHostToDevice, h->s_copy);
          (void)hipStreamSynchronize(h->s_copy);
          (void)hipFree(warm_dst);
        }
      }
      if (!mk(&h->stream, 0) || !mk(&h->s_lo, 1) || (cfg->with_mapping && (!mk(&h->s_map, 2) || !mk(&h->s_ds, 3))) ||   // no mapping: no further hardware queues
          (cfg->image_width > 0 && !mk(&h->s_img, 4))) {
        set_err("hipStreamCreate failed"); st = VLOAM_ERR_HIP; break;
      }
    }
    if (sr_init() != hipSuccess) { set_err("sr_init failed (no gfx950 code object for this device?)"); st = VLOAM_ERR_HIP; break; }
    if (cfg->image_width > 0 && img_init() != hipSuccess) { set_err("img_init failed"); st = VLOAM_ERR_HIP; break; }
    auto body = [&]() -> vloam_status {
      // 1. measure one session, 2. one allocation for all sessions (zeroed), 3. lay session 0 out for real
      Arena dry;
      vloam_status s = handle_layout(h, dry);
      if (s != VLOAM_OK) return s;
      // 2 MB granules + a skew: with arenas exactly 2 MB-aligned every session's hot words (bucket counters, cursors, table heads) share
      // their low address bits, i.e. ALL sessions of a batch hit the same memory channels at the same time (k_lo_grid_count's atomics:
      // 2 150 cycles of vector-memory latency alone, 8 160 at B = 16, profiles/r05_batch_pmc.txt); the skew walks the sessions over the channels
      static const size_t skew = getenv("VLOAM_ARENA_SKEW") ? (size_t)atol(getenv("VLOAM_ARENA_SKEW")) & ~(size_t)255 : 0;
      const size_t ss = ((dry.off + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1)) + (n_sessions > 1 ? skew : 0);
      h->se.B = n_sessions;
      h->se.ss = ss;
      // The cooperative solves of a single sequence are placed on ONE XCD each (lm_solve.hip: lm_coop_block): 8 + 8 compute units of XCDs 2 and
      // 6.  Several single-sequence handles in one process would crowd those two XCDs (and their solves wait for each other's compute
      // units): only the first two alive get the placement, the others launch spread over the XCDs like before round 4.  Same arithmetic.
      if (n_sessions == 1) { h->se.crowd = g_single_handles.fetch_add(1); h->counted_single = true; }
      {
        // ... and only on the device the placement was measured on: 256 compute units dealt round-robin to 8 XCDs (SPX mode).  Anything else
        // (a CPX / NPS partition, a CU-masked context, another part) keeps the plain spread launch; and whatever the placement, a solve
        // whose workgroups do not end up resident together degrades to one workgroup instead of failing (lm_solve.hip).
        hipDeviceProp_t prop2;
        if (hipGetDeviceProperties(&prop2, device) != hipSuccess || prop2.multiProcessorCount != 256) h->se.crowd = 1 << 20;
      }
      h->arena_bytes = ss * (size_t)n_sessions;
      if (hipMalloc((void**)&h->arena, h->arena_bytes) != hipSuccess) {
        set_err("hipMalloc of %zu MB for %d session(s) failed", h->arena_bytes >> 20, n_sessions); h->arena = nullptr; return VLOAM_ERR_HIP;
      }
      HIPCHK(hipMemsetAsync(h->arena, 0, h->arena_bytes, h->stream));
      Arena A;
      A.base = h->arena; A.cap = ss; A.dry = false;
      s = handle_layout(h, A);
      if (s != VLOAM_OK) return s;
      h->map.se = h->se;
      h->vo.se = h->se;
      // ---- initial state of session 0
      for (int k = 0; k < vloam_handle::kSets; k++) {
        int arm[4 * kMaxRings];
        for (int q = 0; q < 4 * kMaxRings; q++) arm[q] = ((q / kMaxRings) & 1) ? -1 : INT_MAX;
        HIPCHK(hipMemcpyAsync(h->grid[k].occ, arm, sizeof(arm), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
      }
      LOState init;
      memset(&init, 0, sizeof(init));
      init.para_q[3] = 1.0; init.q_w_curr[3] = 1.0; init.prior_q[3] = 1.0;  // laser_odometry.cpp:80-90
      tf_identity(&init.tf.base_T_cam0); tf_identity(&init.tf.velo_T_cam0); tf_identity(&init.tf.cam0_curr_T_cam0_last);  // visual_odometry.cpp:73-74
      tf_identity(&init.tf.cam0_curr_LOT_cam0_prev); tf_identity(&init.tf.world_VOT_base_last);                            // vloam_tf.cpp:10-11
      HIPCHK(hipMemcpyAsync(h->lo, &init, sizeof(init), hipMemcpyHostToDevice, h->stream));
      if (hipHostMalloc((void**)&h->ring_watch, sizeof(int) * kMaxBatch, hipHostMallocMapped) != hipSuccess) { h->ring_watch = nullptr; set_err("hipHostMalloc failed"); return VLOAM_ERR_HIP; }
      for (int b = 0; b < kMaxBatch; b++) h->ring_watch[b] = 0;
      if (hipHostMalloc((void**)&h->coop_flag, sizeof(int), hipHostMallocMapped) != hipSuccess) { h->coop_flag = nullptr; set_err("hipHostMalloc failed"); return VLOAM_ERR_HIP; }
      *h->coop_flag = 0;
      h->lo_F.host_degraded = h->coop_flag; h->map.F[0].host_degraded = h->coop_flag; h->map.F[1].host_degraded = h->coop_flag;
      {
        // patience of the workgroups of a cooperative solve (polls before one gives up on its partners, ~0.2 s by default): read ONCE per handle,
        // here — not on the enqueue path, where a getenv per launch would also race a host thread's setenv.  Tests force the degraded path
        // with VLOAM_LM_SPIN_LIMIT=1; anything below 1 (a typo, an empty string) would degrade every solve for good and means "default".
        const char* e = getenv("VLOAM_LM_SPIN_LIMIT");
        const int v = e ? atoi(e) : 0;
        const int limit = v >= 1 ? v : (1 << 18);
        h->lo_F.spin_limit = limit; h->map.F[0].spin_limit = limit; h->map.F[1].spin_limit = limit; h->vo.F.spin_limit = limit;
      }
      s = map_init(&h->map, h->stream);
      if (s != VLOAM_OK) { set_err("map_init failed: %s", hipGetErrorString(hipGetLastError())); return VLOAM_ERR_HIP; }
      {
        // sync words of the three cooperative solves (odometry, mapping outer rounds): the fastest of 48 candidate lines, probed
        // on session 0 (the other sessions' arenas start on 2 MB boundaries: same low address bits)
        int order[kSyncCand];
        if (lm_sync_calibrate(h->stream, h->sync_pool, kSyncCand - 1, kSyncStride, order) != 0) { set_err("lm_sync_calibrate failed"); return VLOAM_ERR_HIP; }
        auto slot = [&](int r) { return reinterpret_cast<double*>(reinterpret_cast<char*>(h->sync_pool) + kSyncStride * (size_t)order[r]); };
        h->lo_F.gsync = slot(0);
        h->map.F[0].gsync = slot(1);
        h->map.F[1].gsync = slot(2);
      }
      for (int k = 0; k < 6; k++) HIPCHK(hipEventCreate(&h->ev[k]));
      for (int k = 0; k < vloam_handle::kSets; k++) {
        HIPCHK(hipEventCreateWithFlags(&h->ev_sr[k], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_lo[k], hipEventDisableTiming | hipEventBlockingSync));   // the host throttle sleeps on these
        HIPCHK(hipEventCreateWithFlags(&h->ev_map[k], hipEventDisableTiming | hipEventBlockingSync));
        HIPCHK(hipEventCreateWithFlags(&h->ev_stack[k], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_vo[k], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_img[k], hipEventDisableTiming));
      }
      HIPCHK(hipStreamSynchronize(h->stream));
      // ---- the other sessions start as byte-for-byte copies of session 0
      for (int b = 1; b < n_sessions; b++)
        HIPCHK(hipMemcpyAsync(h->arena + (size_t)b * ss, h->arena, dry.off, hipMemcpyDeviceToDevice, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      return VLOAM_OK;
    };
    st = body();
  } while (0);
  if (st != VLOAM_OK) { vloam_destroy(h); return st; }
  *out = h;
  return VLOAM_OK;
}

vloam_status vloam_create(const vloam_config* cfg, int device, vloam_handle** out) { return vloam_create_batch(cfg, device, 1, out); }

vloam_status vloam_batch_size(vloam_handle* h, int* n_sessions) {
  if (!h || !n_sessions) return VLOAM_ERR_INVALID;
  *n_sessions = h->se.B;
  return VLOAM_OK;
}

// which session the getters (trajectory, features, counts, map, parity hooks) read; 0 after creation
vloam_status vloam_select_session(vloam_handle* h, int session) {
  if (!h || session < 0 || session >= h->se.B) return VLOAM_ERR_INVALID;
  h->sel = session;
  h->map.sel = session;
  return VLOAM_OK;
}

vloam_status vloam_destroy(vloam_handle* h) {
  if (!h) return VLOAM_OK;
  if (h->counted_single) g_single_handles.fetch_sub(1);
  if (g_host_prof && h->host_calls > 0)
    fprintf(stderr, "[vloam host prof] %lld sweeps: per sweep %.1f us in the buffer-set throttle, %.1f us enqueueing SR (incl. throttle), %.1f us LO, %.1f us mapping\n",
            h->host_calls, 1e6 * h->host_s[0] / h->host_calls, 1e6 * h->host_s[1] / h->host_calls, 1e6 * h->host_s[2] / h->host_calls, 1e6 * h->host_s[3] / h->host_calls);
  (void)hipSetDevice(h->device);
  for (hipStream_t st : {h->s_copy, h->stream, h->s_lo, h->s_map, h->s_ds, h->s_img}) if (st) (void)hipStreamSynchronize(st);
  if (h->arena) (void)hipFree(h->arena);
  for (int k = 0; k < vloam_handle::kInRing; k++) if (h->ev_in_copied[k]) (void)hipEventDestroy(h->ev_in_copied[k]);
  for (int k = 0; k < 6; k++) if (h->ev[k]) (void)hipEventDestroy(h->ev[k]);
  for (int k = 0; k < vloam_handle::kSets; k++)
    for (hipEvent_t e : {h->ev_sr[k], h->ev_lo[k], h->ev_map[k], h->ev_stack[k], h->ev_vo[k], h->ev_img[k]}) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
  for (hipStream_t st : {h->s_copy, h->stream, h->s_lo, h->s_map, h->s_ds, h->s_img}) if (st) (void)hipStreamDestroy(st);
  map_destroy(&h->map);
  if (h->ring_watch) (void)hipHostFree(h->ring_watch);
  if (h->coop_flag) (void)hipHostFree(h->coop_flag);
  delete h;
  return VLOAM_OK;
}

vloam_status vloam_reset_frame(vloam_handle* h) {
  if (!h) return VLOAM_ERR_INVALID;
  h->stage = 0;  // scan_registration.reset() clears the per-frame clouds; laser_mapping.reset() zeroes the valid-cube counters
  return VLOAM_OK;
}

// ------------------------------------------------------------------ stage enqueue helpers
static inline int set_of(int frame) { return frame % vloam_handle::kSets; }
static vloam_status drain_deferred(vloam_handle* h, int lag_lo, int lag_map);
static vloam_status flush_pending(vloam_handle* h);
static vloam_status sync_all(vloam_handle* h) {
  { vloam_status s_ = drain_deferred(h, 0, 0); if (s_ != VLOAM_OK) return s_; }
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->s_lo));
  if (h->s_ds) HIPCHK(hipStreamSynchronize(h->s_ds));
  if (h->s_map) HIPCHK(hipStreamSynchronize(h->s_map));
  if (h->s_img) HIPCHK(hipStreamSynchronize(h->s_img));
  return VLOAM_OK;
}

static BatchIn one_sweep(const void* d_xyz_pad4, int n) {
  BatchIn bi;
  memset(&bi, 0, sizeof(bi));
  bi.in[0] = (const float4*)d_xyz_pad4; bi.n[0] = n;
  return bi;
}

static vloam_status enqueue_sr(vloam_handle* h, const BatchIn& bi) {
  int n = 0;
  for (int b = 0; b < h->se.B; b++) {
    if (!bi.in[b]) { set_err("null sweep pointer for session %d", b); return VLOAM_ERR_INVALID; }
    if (bi.n[b] <= 0) { set_err("empty cloud"); return VLOAM_ERR_EMPTY; }
    if (bi.n[b] > h->cfg.max_points) { set_err("cloud of %d points exceeds max_points=%d", bi.n[b], h->cfg.max_points); return VLOAM_ERR_CAPACITY; }
    n = bi.n[b] > n ? bi.n[b] : n;
  }
  if (h->frame >= h->cfg.max_frames) { set_err("trajectory log full (max_frames=%d)", h->cfg.max_frames); return VLOAM_ERR_CAPACITY; }
  const int k = h->frame, cur = set_of(k);
  // Set `cur` still holds sweep k - 3: read by odometry of sweeps k - 3 (current) and k - 2 (previous), mapping of sweep k - 3.
  // The HOST waits for those before enqueueing (it may run at most two sweeps ahead of the odometry, three ahead of the
  // mapping): a cross-stream barrier packet in front of every sweep costs ~12 us on the stream that bounds the throughput,
  // a host-side check of an event that has almost always fired costs nothing on the device.
  constexpr int kS = vloam_handle::kSets;  // set `cur` holds sweep k - kS: odometry of sweeps k - kS and k - kS + 1, mapping of k - kS
  const double tw0 = g_host_prof ? now_s() : 0.0;
  if (k >= kS - 1) HIPCHK(hipEventSynchronize(h->ev_lo[set_of(k - (kS - 1))]));
  if (k >= kS && h->cfg.with_mapping) HIPCHK(hipEventSynchronize(h->ev_map[set_of(k - kS)]));
  if (g_host_prof) h->host_s[0] += now_s() - tw0;
  if (h->cfg.timing) HIPCHK(hipEventRecord(h->ev[0], h->stream));
  // the full grid of the big ring tier only while long rings are around (watch word of an earlier sweep: plain read of host-mapped memory);
  // whatever the host knows or does not know, ONE catch-all workgroup of the big tier follows the small tier on every sweep, so any ring of up
  // to kMaxRingLen points is processed (sr_launch)
  bool big_tier = false;
  for (int b = 0; b < h->se.B; b++) big_tier = big_tier || __atomic_load_n(&h->ring_watch[b], __ATOMIC_RELAXED) != 0;
  HIPCHK(sr_launch(h->stream, h->sr[cur], bi, h->se, h->cfg.scan_line, (float)h->cfg.minimum_range, h->cfg.debug, &h->prof,
                   h->ev_sr[cur], h->ring_watch, big_tier));  // the odometry of THIS sweep needs the feature clouds only (its NN grids were built with the previous sweep)
  // == kdtreeCornerLast / kdtreeSurfLast->setInputCloud (laser_odometry.cpp:525-526): index this sweep's clouds for the next one
  // (the next sweep's ev_sr is recorded behind this on the same stream, so its odometry sees the finished grids)
  lo_grid_build_launch(h->stream, h->se, h->sr[cur].less_sharp, h->sr[cur].less_flat, h->sr[cur].S, h->grid[cur], &h->prof);
  HIPCHK(hipGetLastError());
  if (h->cfg.timing) HIPCHK(hipEventRecord(h->ev[1], h->stream));
  // the mapping stage's VoxelGrid of the scan features only needs this sweep's clouds: run it here, off the mapping stream
  if (h->cfg.with_mapping && ((k + 1) % h->cfg.mapping_skip_frame) == 0) {
    HIPCHK(hipStreamWaitEvent(h->s_ds, h->ev_sr[cur], 0));   // the feature clouds (ev_sr is bound to k_sr_compact); a live wait, on a stream that has the time
    if (map_stack_enqueue(&h->map, h->s_ds, h->sr[cur], cur, &h->prof, h->ev_stack[cur]) != VLOAM_OK) { set_err("map_stack_enqueue failed"); return VLOAM_ERR_HIP; }
  }
  h->last_n_in = n;
  h->stage = 1;
  h->vo_frame[cur] = false;
  h->img_frame[cur] = false;
  return VLOAM_OK;
}

// A cooperative solve that found its partners missing finished on one workgroup and said so in a host-mapped word: from the next enqueue on
// this handle launches one-workgroup solves (no host synchronisation needed: a host that streams thousands of sweeps between two vloam_sync
// calls pays the ~0.2 s wait once, not per solve).
static inline void poll_coop_flag(vloam_handle* h) {
  if (!h->se.no_coop && h->coop_flag && __atomic_load_n(h->coop_flag, __ATOMIC_RELAXED)) { h->se.no_coop = 1; h->map.se.no_coop = 1; h->vo.se.no_coop = 1; }
}

static vloam_status enqueue_lo(vloam_handle* h, int frame) {
  poll_coop_flag(h);
  const int cur = set_of(frame), prev = set_of(frame + vloam_handle::kSets - 1);
  HIPCHK(hipStreamWaitEvent(h->s_lo, h->ev_sr[cur], 0));
  if (h->cfg.timing) HIPCHK(hipEventRecord(h->ev[2], h->s_lo));
  const bool coupled = h->vo_frame[cur];  // MAIN/src/vloam_main_node.cpp:125-180: this frame's VO runs in front of its laser odometry
  const bool use_prior = !h->cfg.detach_VO_LO;
  if (coupled) {
    HIPCHK(hipStreamWaitEvent(h->s_lo, h->ev_vo[cur], 0));
    if (h->img_frame[cur]) HIPCHK(hipStreamWaitEvent(h->s_lo, h->ev_img[cur], 0));
    if (frame > 0) {  // Section 4: if (count > 0) VO->solveNlsAll()
      vloam_status s = vo_solve_enqueue(&h->vo, h->cfg, h->s_lo, frame, h->lo, &h->prof);
      if (s != VLOAM_OK) { set_err("vo_solve_enqueue failed"); return s; }
    }
    // vloam_tf->VO2VeloAndBase(VO->cam0_curr_T_cam0_last) + (combined mode) the first outer round's para_q / para_t overwrite
    lo_set_prior_launch(h->s_lo, h->se, h->lo, use_prior && frame > 0, h->vo.x, frame > 0, h->vo_traj + (size_t)frame * 7, &h->map.frame->error);
  }
  if (frame > 0) {  // first sweep only initialises (laser_odometry.cpp:196-204)
    for (int outer = 0; outer < 2; outer++) {  // laser_odometry.cpp:211
      if (use_prior && !(coupled && outer == 0)) lo_set_prior_launch(h->s_lo, h->se, h->lo);  // laser_odometry.cpp:223-236, both rounds (quirk A.8-4)
      FactorTable F = h->lo_F;
      F.resid = h->lo_resid[outer];
      lo_assoc_launch(h->s_lo, h->se, h->sr[cur].sharp, h->sr[cur].flat, h->sr[cur].S, h->sr[prev].less_sharp, h->sr[prev].less_flat,
                      h->sr[prev].S, h->grid[prev], h->lo, F, h->lo_corr[outer], h->lo_cyc[outer], h->lo_queue, h->lo_queue_n, h->lo_launches++, &h->prof);
      // the second solve also integrates the pose and writes the trajectory row (laser_odometry.cpp:530-531)
      lm_launch(h->s_lo, h->se, F, kMaxSharp, h->lo->para_q, h->lo_rec + outer, 4, 0.1, true, nullptr, &h->prof, outer == 1 ? h->lo : nullptr,
                outer == 1 ? h->traj + (size_t)frame * 14 : nullptr, outer == 1 ? h->ev_lo[cur] : nullptr);
    }
  } else {
    lo_finish_launch(h->s_lo, h->se, h->lo, h->traj + (size_t)frame * 14, false, &h->prof);
    HIPCHK(hipEventRecord(h->ev_lo[cur], h->s_lo));
  }
  HIPCHK(hipGetLastError());
  if (h->cfg.timing) HIPCHK(hipEventRecord(h->ev[3], h->s_lo));
  return VLOAM_OK;
}

static vloam_status enqueue_map(vloam_handle* h, int frame) {
  poll_coop_flag(h);
  const int cur = set_of(frame);
  HIPCHK(hipStreamWaitEvent(h->s_map, h->ev_lo[cur], 0));
  HIPCHK(hipStreamWaitEvent(h->s_map, h->ev_stack[cur], 0));
  if (h->cfg.timing) HIPCHK(hipEventRecord(h->ev[4], h->s_map));
  // LaserOdometry::output: skip_frame = (frameCount % mapping_skip_frame != 0), frameCount already incremented (laser_odometry.cpp:535,618)
  const bool skip = ((frame + 1) % h->cfg.mapping_skip_frame) != 0;
  const bool sub_pose = h->sub_pose_frame == frame;   // LaserMapping::input was handed another odometry pose (vloam_set_mapping_input)
  vloam_status s = map_enqueue(&h->map, h->cfg, h->s_map, h->sr[cur], h->lo, sub_pose ? h->sub_row : h->traj + (size_t)frame * 14, skip, cur, &h->prof, h->ev_map[cur]);
  if (s != VLOAM_OK) { set_err("map_enqueue failed: %s", hipGetErrorString(hipGetLastError())); return VLOAM_ERR_HIP; }
  if (sub_pose) HIPCHK(hipMemcpyAsync(h->traj + (size_t)frame * 14 + 7, h->sub_row + 7, 7 * sizeof(double), hipMemcpyDeviceToDevice, h->s_map));   // the map half of the log
  if (h->cfg.timing) HIPCHK(hipEventRecord(h->ev[5], h->s_map));
  return VLOAM_OK;
}

// vloam_process_scan leaves the odometry of its sweep (and the mapping of the sweep before) to the NEXT call: by then the
// producing stage has normally finished, hipStreamWaitEvent on a completed event inserts nothing, and the ~11 us cross-stream
// barrier packet disappears from the stream that bounds the throughput.  Everything that reads results drains first (sync_all).
static vloam_status drain_deferred(vloam_handle* h, int lag_lo, int lag_map) {
  if (lag_lo == 0 && lag_map == 0) { vloam_status s0 = flush_pending(h); if (s0 != VLOAM_OK) return s0; }   // a full drain: the host sweep still in flight first
  while (h->lo_done < h->frame - lag_lo) {
    const double t0 = g_host_prof ? now_s() : 0.0;
    vloam_status s = enqueue_lo(h, h->lo_done);
    if (g_host_prof) h->host_s[2] += now_s() - t0;
    if (s != VLOAM_OK) return s;
    h->lo_done++;
  }
  if (!h->cfg.with_mapping) { h->map_done = h->lo_done; return VLOAM_OK; }
  const int lim = h->frame - lag_map < h->lo_done ? h->frame - lag_map : h->lo_done;
  while (h->map_done < lim) {
    const double t0 = g_host_prof ? now_s() : 0.0;
    vloam_status s = enqueue_map(h, h->map_done);
    if (g_host_prof) h->host_s[3] += now_s() - t0;
    if (s != VLOAM_OK) return s;
    h->map_done++;
  }
  return VLOAM_OK;
}

static vloam_status finish_frame(vloam_handle* h) {
  if (h->cfg.timing) {  // per-stage times need the sweep drained: timing mode gives up the overlap between sweeps
    vloam_status s = sync_all(h);
    if (s != VLOAM_OK) return s;
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, h->ev[0], h->ev[1])); h->stage_ms[0] += ms;
    HIPCHK(hipEventElapsedTime(&ms, h->ev[2], h->ev[3])); h->stage_ms[1] += ms;
    if (h->cfg.with_mapping) { HIPCHK(hipEventElapsedTime(&ms, h->ev[4], h->ev[5])); h->stage_ms[2] += ms; }
    h->timed_scans++;
  }
  h->frame++;
  h->stage = 0;
  if (h->lo_done < h->frame) h->lo_done = h->frame;    // stage-wise callers may leave stages out: nothing is owed for this sweep
  if (h->map_done < h->frame) h->map_done = h->frame;
  return VLOAM_OK;
}

// ------------------------------------------------------------------ host sweeps in
// Two forms (vloam_handle: "Host sweeps").
//   INLINE: hipMemcpyAsync on the scan-registration stream itself into the inline buffer, no events — the stream's order IS the dependency.
//     Every host-pointer entry point except the two below; with VLOAM_STAGE_INLINE=1 those two as well.  The 2 MB cross the link IN FRONT of the
//     sweep's scan registration (24 GB/s inside hipMemcpyAsync: ~87 us on a stream with ~30 us to spare per period): 0.88 - 0.92 x the
//     device-resident rate.
//   DEFERRED RING (vloam_process_scan, vloam_batch_process_scan): copy on a stream of its own into ring slot k % kInRing, the sweep enqueued
//     by the NEXT call once the HOST has seen the copy's event — no stream ever waits for another one, the copy of sweep k + 1 runs beside
//     the scan registration of sweep k: 0.95 - 0.98 x (tools/host_input_probe.py).  Rounds 4 - 5 had a ring WITHOUT the deferral (two live
//     cross-stream waits per sweep): 2 200 - 4 400 scans/s depending on what else the process had created (profiles/r05_host_input.txt); a copy
//     KERNEL reading the pinned buffer (49 GB/s) slows every kernel beside it by ~45 us (profiles/r06_host_input.txt).  Both are gone.
// Pageable source memory: hipMemcpyAsync has taken its copy when it returns (the caller may reuse the buffer at once).  Pinned source
// memory (hipHostMalloc / hipHostRegister) is read by DMA later: it must stay unchanged until the next vloam_sync() (c_api.h).
static vloam_status stage_sweep(vloam_handle* h, int b, const float* xyz_pad4, int n, const float4** d_out) {
  float4* dst = (float4*)((char*)(h->d_in + (size_t)vloam_handle::kInRing * (size_t)h->cfg.max_points) + (size_t)b * h->se.ss);
  HIPCHK(hipMemcpyAsync(dst, xyz_pad4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, h->stream));
  *d_out = dst;
  return VLOAM_OK;
}
static vloam_status process_scan_batch(vloam_handle* h, const BatchIn& bi);
// the host sweep whose copy is in flight: enqueue it (every drain, every new sweep of whatever kind, vloam_sync)
static vloam_status flush_pending(vloam_handle* h) {
  if (!h->pend.valid) return VLOAM_OK;
  h->pend.valid = false;
  const int slot = h->pend.slot;
  HIPCHK(hipEventSynchronize(h->ev_in_copied[slot]));   // host side; the copy was enqueued a whole call ago
  const vloam_status st = process_scan_batch(h, h->pend.bi);
  // the slot's last reader is this sweep's scan registration (its event belongs to the sweep until the buffer set comes round again: kInRing < kSets)
  h->in_reader[slot] = st == VLOAM_OK ? h->ev_sr[set_of(h->frame - 1)] : nullptr;
  if (st != VLOAM_OK) HIPCHK(hipStreamSynchronize(h->stream));   // a refused sweep: whatever was enqueued before the refusal may still read the slot
  return st;
}
static vloam_status host_scan_deferred(vloam_handle* h, const float* const* xyz_pad4, const int* n) {
  if (h->frame + (h->pend.valid ? 1 : 0) >= h->cfg.max_frames) { set_err("trajectory log full (max_frames=%d)", h->cfg.max_frames); return VLOAM_ERR_CAPACITY; }
  if (!h->s_copy) HIPCHK(hipStreamCreateWithFlags(&h->s_copy, hipStreamNonBlocking));   // (normally created first of all streams: vloam_create)
  if (!h->ev_in_copied[0]) for (int k = 0; k < vloam_handle::kInRing; k++) HIPCHK(hipEventCreateWithFlags(&h->ev_in_copied[k], hipEventDisableTiming));
  const int slot = h->in_next % vloam_handle::kInRing;
  h->in_next++;
  // host side.  The slot's reader was enqueued kInRing - 1 calls ago; it can still be running while the host sprints ahead at the start of a
  // burst (measured: without this wait the last pose of a 300-sweep run changes from run to run)
  if (h->in_reader[slot]) HIPCHK(hipEventSynchronize(h->in_reader[slot]));
  BatchIn bi;
  memset(&bi, 0, sizeof(bi));
  for (int b = 0; b < h->se.B; b++) {
    float4* dst = (float4*)((char*)(h->d_in + (size_t)slot * (size_t)h->cfg.max_points) + (size_t)b * h->se.ss);
    HIPCHK(hipMemcpyAsync(dst, xyz_pad4[b], (size_t)n[b] * sizeof(float4), hipMemcpyHostToDevice, h->s_copy));
    bi.in[b] = dst; bi.n[b] = n[b];
  }
  HIPCHK(hipEventRecord(h->ev_in_copied[slot], h->s_copy));
  const vloam_status st = flush_pending(h);   // the sweep before this one
  h->pend.bi = bi; h->pend.slot = slot; h->pend.valid = true;
  return st;
}
static_assert(vloam_handle::kInRing < vloam_handle::kSets, "a slot's reader event (per buffer set) must not be re-recorded before the slot is reused");
// the inline form's brackets around a call's copies: whatever host sweep is still pending goes first (it is older than this call's sweep)
static inline vloam_status stage_begin(vloam_handle* h) { return flush_pending(h); }
static inline vloam_status stage_end(vloam_handle*) { return VLOAM_OK; }
static inline vloam_status stage_release(vloam_handle*, vloam_status call_status) { return call_status; }
#define STAGE_ONE(h, xyz, n, dptr)                                                                              \
  const float4* dptr = nullptr;                                                                                 \
  { vloam_status s_ = flush_pending(h); if (s_ == VLOAM_OK) s_ = stage_sweep(h, 0, xyz, n, &dptr); if (s_ != VLOAM_OK) return s_; }

// ------------------------------------------------------------------ stage-wise API (façade order)
vloam_status vloam_scan_registration_device(vloam_handle* h, const void* d_xyz_pad4, int n) {
  if (!h || !d_xyz_pad4) return VLOAM_ERR_INVALID;
  SINGLE_SESSION_ONLY(h);
  HIPCHK(hipSetDevice(h->device));
  if (h->stage == 2) { vloam_status s = finish_frame(h); if (s != VLOAM_OK) return s; }  // previous sweep ended after LO (no mapping call)
  { vloam_status s = drain_deferred(h, 0, 0); if (s != VLOAM_OK) return s; }               // stages owed by earlier vloam_process_scan calls
  return enqueue_sr(h, one_sweep(d_xyz_pad4, n));
}

vloam_status vloam_scan_registration(vloam_handle* h, const float* xyz_pad4, int n) {
  if (!h || !xyz_pad4) return VLOAM_ERR_INVALID;
  if (n > h->cfg.max_points) { set_err("cloud of %d points exceeds max_points=%d", n, h->cfg.max_points); return VLOAM_ERR_CAPACITY; }
  if (n <= 0) { set_err("empty cloud"); return VLOAM_ERR_EMPTY; }
  SINGLE_SESSION_ONLY(h);
  HIPCHK(hipSetDevice(h->device));
  STAGE_ONE(h, xyz_pad4, n, d);
  return stage_release(h, vloam_scan_registration_device(h, d, n));
}

static vloam_status read_sr_error(vloam_handle* h, int cur) {
  int err = 0;
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  HIPCHK(hipMemcpy(&err, &SEL(h, h->sr[cur].S)->error, sizeof(int), hipMemcpyDeviceToHost));
  if (err & kErrEmpty) { set_err("no point survived NaN / minimum_range removal"); return VLOAM_ERR_EMPTY; }
  if (err & kErrRingTooLong) { set_err("a ring holds more than %d points", kMaxRingLen); return VLOAM_ERR_CAPACITY; }
  return VLOAM_OK;
}

vloam_status vloam_get_features(vloam_handle* h, int which, float* xyzi4, int cap, int* n) {
  if (!h || !n) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }   // (first: a host sweep still in flight is enqueued by this and counts)
  // after finish_frame() the sweep just processed is frame-1
  const int f = (h->stage == 0 && h->frame > 0) ? h->frame - 1 : h->frame;
  const int cur = set_of(f);
  SRBuffers bsel = h->sr[cur];
  if (which == 11 && h->sub_cloud_frame == f) { bsel.cloud = h->sub.cloud; bsel.S = h->sub.S; }   // laserCloudFullRes as handed to LaserMapping::input
  bsel.rebase((size_t)h->sel * h->se.ss);
  FrameScalars S;
  HIPCHK(hipMemcpy(&S, bsel.S, sizeof(S), hipMemcpyDeviceToHost));
  const float4* src = nullptr;
  int cnt = 0;
  switch (which) {
    case 0: src = bsel.cloud; cnt = S.N2; break;
    case 1: src = bsel.sharp; cnt = S.n_sharp; break;
    case 2: case 5: src = bsel.less_sharp; cnt = S.n_less_sharp; break;
    case 3: src = bsel.flat; cnt = S.n_flat; break;
    case 4: case 6: src = bsel.less_flat; cnt = S.n_less_flat; break;
    default: return map_get_cloud(&h->map, h->stream, which, bsel, xyzi4, cap, n);
  }
  *n = cnt;
  const int m = cnt < cap ? cnt : cap;
  if (xyzi4 && m > 0) HIPCHK(hipMemcpy(xyzi4, src, (size_t)m * sizeof(float4), hipMemcpyDeviceToHost));
  return VLOAM_OK;
}

// == the /laser_cloud_map product of LaserMapping::publish (laser_mapping.cpp:778-793)
vloam_status vloam_get_map(vloam_handle* h, float* xyzi4, long long cap, long long* n) {
  if (!h || !n) return VLOAM_ERR_INVALID;
  if (!h->cfg.with_mapping) { *n = 0; return VLOAM_OK; }
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  return map_export(&h->map, h->s_map, xyzi4, cap, n);
}

vloam_status vloam_set_lo_prior(vloam_handle* h, const double q[4], const double t[3]) {
  if (!h || !q || !t) return VLOAM_ERR_INVALID;
  SINGLE_SESSION_ONLY(h);
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = drain_deferred(h, 0, 0); if (s_ != VLOAM_OK) return s_; }  // the prior belongs to the NEXT sweep's odometry
  double buf[7] = {q[0], q[1], q[2], q[3], t[0], t[1], t[2]};
  HIPCHK(hipMemcpyAsync(h->lo->prior_q, buf, sizeof(buf), hipMemcpyHostToDevice, h->s_lo));
  HIPCHK(hipStreamSynchronize(h->s_lo));
  return VLOAM_OK;
}

// LaserOdometry::input with clouds that are NOT what scan registration left on the device (the reference deep-copies whatever it is handed,
// laser_odometry.cpp:141-145).  Between vloam_scan_registration and vloam_laser_odometry; a null cloud keeps the device's.  The clouds go
// into the sweep's buffer set, and what scan registration had derived from them is rebuilt: counts, the NN grids over the two less-clouds
// (the NEXT sweep's CornerLast / SurfLast, laser_odometry.cpp:506-526) and the mapping stage's VoxelGrid of them.
vloam_status vloam_set_odometry_input(vloam_handle* h, const float* laserCloud, int n_full, const float* cornerPointsSharp, int n_sharp,
                                      const float* cornerPointsLessSharp, int n_less_sharp, const float* surfPointsFlat, int n_flat,
                                      const float* surfPointsLessFlat, int n_less_flat) {
  if (!h) return VLOAM_ERR_INVALID;
  SINGLE_SESSION_ONLY(h);
  HIPCHK(hipSetDevice(h->device));
  if (h->stage != 1) { set_err("vloam_set_odometry_input belongs between scan registration and laser odometry"); return VLOAM_ERR_ORDER; }
  const float* src[5] = {laserCloud, cornerPointsSharp, cornerPointsLessSharp, surfPointsFlat, surfPointsLessFlat};
  int n[5] = {n_full, n_sharp, n_less_sharp, n_flat, n_less_flat};
  const int cap[5] = {h->cfg.max_points, kMaxSharp, kMaxLessSharp, kMaxFlat, h->cfg.max_points};
  for (int k = 0; k < 5; k++) {
    if (!src[k]) { n[k] = -1; continue; }
    if (n[k] < 0) { set_err("negative cloud size"); return VLOAM_ERR_INVALID; }
    if (n[k] > cap[k]) { set_err("substituted cloud %d holds %d points, the buffer takes %d", k, n[k], cap[k]); return VLOAM_ERR_CAPACITY; }
  }
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  const int cur = set_of(h->frame);
  const SRBuffers& b = h->sr[cur];
  float4* dst[5] = {b.cloud, b.sharp, b.less_sharp, b.flat, b.less_flat};
  for (int k = 0; k < 5; k++) if (src[k] && n[k] > 0) HIPCHK(hipMemcpy(dst[k], src[k], (size_t)n[k] * sizeof(float4), hipMemcpyHostToDevice));
  sr_adopt_launch(h->stream, b, n[0], n[1], n[2], n[3], n[4]);
  if (src[2] || src[4]) {
    lo_grid_build_launch(h->stream, h->se, b.less_sharp, b.less_flat, b.S, h->grid[cur], &h->prof);
    HIPCHK(hipEventRecord(h->ev_sr[cur], h->stream));
    if (h->cfg.with_mapping && ((h->frame + 1) % h->cfg.mapping_skip_frame) == 0) {
      HIPCHK(hipStreamWaitEvent(h->s_ds, h->ev_sr[cur], 0));
      if (map_stack_enqueue(&h->map, h->s_ds, b, cur, &h->prof, h->ev_stack[cur]) != VLOAM_OK) { set_err("map_stack_enqueue failed"); return VLOAM_ERR_HIP; }
    }
  } else HIPCHK(hipEventRecord(h->ev_sr[cur], h->stream));
  HIPCHK(hipGetLastError());
  return VLOAM_OK;
}

// The laser-odometry pose of the sweep in progress (after vloam_laser_odometry) or of the last finished sweep: q_w_curr / t_w_curr as
// LaserOdometry::output hands them on (laser_odometry.cpp:610-616).
vloam_status vloam_get_odometry_pose(vloam_handle* h, double q_w[4], double t_w[3]) {
  if (!h) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  const int f = (h->stage == 0 && h->frame > 0) ? h->frame - 1 : h->frame;
  double row[7] = {0, 0, 0, 1, 0, 0, 0};
  if (f < h->cfg.max_frames && (h->stage == 2 || h->frame > 0)) HIPCHK(hipMemcpy(row, SEL(h, h->traj) + (size_t)f * 14, sizeof(row), hipMemcpyDeviceToHost));
  if (q_w) memcpy(q_w, row, sizeof(double) * 4);
  if (t_w) memcpy(t_w, row + 4, sizeof(double) * 3);
  return VLOAM_OK;
}

// LaserMapping::input with clouds / an odometry pose that are NOT LaserOdometry::output's (laser_mapping.cpp:167-196 copies what it is handed;
// on a skipped sweep only the pose, :172-181).  Between vloam_laser_odometry and vloam_laser_mapping; null = keep the device's.  Only this
// sweep's mapping sees them: the odometry's CornerLast / SurfLast stay what they were.
vloam_status vloam_set_mapping_input(vloam_handle* h, const float* laserCloudCornerLast, int n_corner, const float* laserCloudSurfLast, int n_surf,
                                     const float* laserCloudFullRes, int n_full, const double q_wodom_curr[4], const double t_wodom_curr[3]) {
  if (!h) return VLOAM_ERR_INVALID;
  SINGLE_SESSION_ONLY(h);
  HIPCHK(hipSetDevice(h->device));
  if (h->stage != 2 || !h->cfg.with_mapping) { set_err("vloam_set_mapping_input belongs between laser odometry and laser mapping of a handle with mapping"); return VLOAM_ERR_ORDER; }
  if ((laserCloudCornerLast && (n_corner < 0 || n_corner > kMaxLessSharp)) || (laserCloudSurfLast && (n_surf < 0 || n_surf > h->cfg.max_points)) ||
      (laserCloudFullRes && (n_full < 0 || n_full > h->cfg.max_points))) { set_err("substituted cloud does not fit its buffer"); return VLOAM_ERR_CAPACITY; }
  if ((q_wodom_curr == nullptr) != (t_wodom_curr == nullptr)) { set_err("the odometry pose is q AND t"); return VLOAM_ERR_INVALID; }
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  const int frame = h->frame, cur = set_of(frame);
  const bool skip = ((frame + 1) % h->cfg.mapping_skip_frame) != 0;
  if (q_wodom_curr) {
    double row[14];
    HIPCHK(hipMemcpy(row, h->traj + (size_t)frame * 14, sizeof(row), hipMemcpyDeviceToHost));
    memcpy(row, q_wodom_curr, 4 * sizeof(double)); memcpy(row + 4, t_wodom_curr, 3 * sizeof(double));
    HIPCHK(hipMemcpy(h->sub_row, row, sizeof(row), hipMemcpyHostToDevice));
    h->sub_pose_frame = frame;
  }
  if (!skip && (laserCloudCornerLast || laserCloudSurfLast || laserCloudFullRes)) {
    const SRBuffers& b = h->sr[cur];
    FrameScalars S;
    HIPCHK(hipMemcpy(&S, b.S, sizeof(S), hipMemcpyDeviceToHost));
    struct { const float* src; int n; float4* own; int n_own; float4* dst; } c[3] = {
      {laserCloudCornerLast, n_corner, b.less_sharp, S.n_less_sharp, h->sub.less_sharp}, {laserCloudSurfLast, n_surf, b.less_flat, S.n_less_flat, h->sub.less_flat},
      {laserCloudFullRes, n_full, b.cloud, S.N2, h->sub.cloud}};
    int n[3];
    for (int k = 0; k < 3; k++) {
      n[k] = c[k].src ? c[k].n : c[k].n_own;
      if (n[k] > 0) HIPCHK(hipMemcpy(c[k].dst, c[k].src ? (const void*)c[k].src : (const void*)c[k].own, (size_t)n[k] * sizeof(float4),
                                     c[k].src ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice));
    }
    HIPCHK(hipMemcpy(h->sub.S, &S, sizeof(S), hipMemcpyHostToDevice));
    sr_adopt_launch(h->stream, h->sub, n[2], -1, n[0], -1, n[1]);
    HIPCHK(hipStreamSynchronize(h->stream));
    h->sub_cloud_frame = frame;
    if (laserCloudCornerLast || laserCloudSurfLast)
      if (map_stack_enqueue(&h->map, h->s_ds, h->sub, cur, &h->prof, h->ev_stack[cur]) != VLOAM_OK) { set_err("map_stack_enqueue failed"); return VLOAM_ERR_HIP; }
  }
  HIPCHK(hipGetLastError());
  return VLOAM_OK;
}

vloam_status vloam_laser_odometry(vloam_handle* h, double q_w[4], double t_w[3], double q_lc[4], double t_lc[3]) {
  if (!h) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  if (h->stage != 1) { set_err("laser odometry called before scan registration"); return VLOAM_ERR_ORDER; }
  vloam_status s = enqueue_lo(h, h->frame);
  if (s != VLOAM_OK) return s;
  h->lo_done = h->frame + 1;
  h->stage = 2;
  s = read_sr_error(h, set_of(h->frame));
  if (s != VLOAM_OK) return s;
  LOState lo;
  HIPCHK(hipMemcpy(&lo, SEL(h, h->lo), sizeof(lo), hipMemcpyDeviceToHost));
  if (q_w) memcpy(q_w, lo.q_w_curr, sizeof(double) * 4);
  if (t_w) memcpy(t_w, lo.t_w_curr, sizeof(double) * 3);
  if (q_lc) memcpy(q_lc, lo.para_q, sizeof(double) * 4);
  if (t_lc) memcpy(t_lc, lo.para_t, sizeof(double) * 3);
  if (!h->cfg.with_mapping) return finish_frame(h);
  return VLOAM_OK;
}

vloam_status vloam_laser_mapping(vloam_handle* h, double q_map[4], double t_map[3]) {
  if (!h) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  if (h->stage != 2) { set_err("laser mapping called before laser odometry"); return VLOAM_ERR_ORDER; }
  vloam_status s = enqueue_map(h, h->frame);
  if (s != VLOAM_OK) return s;
  h->map_done = h->frame + 1;
  s = sync_all(h);
  if (s != VLOAM_OK) return s;
  // what LaserMapping::publish reports (laser_mapping.cpp:718-757): q_w_curr / t_w_curr after a mapped sweep, the
  // high-frequency pose q_wmap_wodom * q_wodom_curr after a skipped one — the map half of this sweep's trajectory row
  double row[14];
  HIPCHK(hipMemcpy(row, SEL(h, h->traj) + (size_t)h->frame * 14, sizeof(row), hipMemcpyDeviceToHost));
  if (q_map) memcpy(q_map, row + 7, sizeof(double) * 4);
  if (t_map) memcpy(t_map, row + 11, sizeof(double) * 3);
  return finish_frame(h);
}

// ------------------------------------------------------------------ whole façade, asynchronous
// sweeps by which the odometry / mapping enqueue trails the scan registration enqueue (see drain_deferred); the buffer-reuse
// waits of enqueue_sr (odometry of sweep k - 3, mapping of sweep k - 4) stay behind what is enqueued here
static constexpr int kLagLO = 1, kLagMap = 2;
static_assert(kLagLO <= vloam_handle::kSets - 2 && kLagMap <= vloam_handle::kSets - 1, "deferred stages must be enqueued before enqueue_sr waits for them");
static vloam_status process_scan_batch(vloam_handle* h, const BatchIn& bi) {
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s0 = flush_pending(h); if (s0 != VLOAM_OK) return s0; }   // (a device-pointer sweep behind a host sweep: the older one first)
  if (h->stage == 2) { vloam_status s0 = finish_frame(h); if (s0 != VLOAM_OK) return s0; }
  if (g_enqueue_order == 1 && !h->cfg.timing) {   // A/B: the deferred odometry / mapping of earlier sweeps first, then this sweep's scan registration
    vloam_status s0 = drain_deferred(h, kLagLO - 1, kLagMap - 1);
    if (s0 != VLOAM_OK) return s0;
  }
  const double ts0 = g_host_prof ? now_s() : 0.0;
  vloam_status s = enqueue_sr(h, bi);
  if (g_host_prof) { h->host_s[1] += now_s() - ts0; h->host_calls++; }
  if (s != VLOAM_OK) return s;
  if (h->cfg.timing) {  // per-stage times: nothing deferred, the sweep is drained in finish_frame
    s = enqueue_lo(h, h->frame);
    if (s != VLOAM_OK) return s;
    if (h->cfg.with_mapping) { s = enqueue_map(h, h->frame); if (s != VLOAM_OK) return s; }
    return finish_frame(h);
  }
  h->frame++;
  h->stage = 0;
  return drain_deferred(h, kLagLO, kLagMap);
}

vloam_status vloam_process_scan_device(vloam_handle* h, const void* d_xyz_pad4, int n) {
  if (!h || !d_xyz_pad4) return VLOAM_ERR_INVALID;
  SINGLE_SESSION_ONLY(h);
  return process_scan_batch(h, one_sweep(d_xyz_pad4, n));
}

// Batched execution: one call advances ALL sessions of the handle by one sweep (session b gets d_xyz_pad4[b], n[b] points); every kernel
// of the sweep chain is launched ONCE with the session index in blockIdx.z.  The sessions are independent sequences that share nothing
// but the launch chain; results per session through vloam_select_session + the getters.
vloam_status vloam_batch_process_scan_device(vloam_handle* h, const void* const* d_xyz_pad4, const int* n) {
  if (!h || !d_xyz_pad4 || !n) return VLOAM_ERR_INVALID;
  BatchIn bi;
  memset(&bi, 0, sizeof(bi));
  for (int b = 0; b < h->se.B; b++) { bi.in[b] = (const float4*)d_xyz_pad4[b]; bi.n[b] = n[b]; }
  return process_scan_batch(h, bi);
}

vloam_status vloam_batch_process_scan(vloam_handle* h, const float* const* xyz_pad4, const int* n) {
  if (!h || !xyz_pad4 || !n) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  BatchIn bi;
  memset(&bi, 0, sizeof(bi));
  for (int b = 0; b < h->se.B; b++) {
    if (!xyz_pad4[b]) return VLOAM_ERR_INVALID;
    if (n[b] > h->cfg.max_points) { set_err("cloud of %d points exceeds max_points=%d", n[b], h->cfg.max_points); return VLOAM_ERR_CAPACITY; }
    if (n[b] <= 0) { set_err("empty cloud"); return VLOAM_ERR_EMPTY; }
  }
  if (!g_stage_inline && !h->cfg.timing) return host_scan_deferred(h, xyz_pad4, n);
  { vloam_status s_ = stage_begin(h); if (s_ != VLOAM_OK) return s_; }
  for (int b = 0; b < h->se.B; b++) {
    vloam_status s_ = stage_sweep(h, b, xyz_pad4[b], n[b], &bi.in[b]);
    if (s_ != VLOAM_OK) return stage_release(h, s_);
    bi.n[b] = n[b];
  }
  { vloam_status s_ = stage_end(h); if (s_ != VLOAM_OK) return stage_release(h, s_); }
  return stage_release(h, process_scan_batch(h, bi));
}

vloam_status vloam_process_scan(vloam_handle* h, const float* xyz_pad4, int n) {
  if (!h || !xyz_pad4) return VLOAM_ERR_INVALID;
  if (n > h->cfg.max_points) { set_err("cloud of %d points exceeds max_points=%d", n, h->cfg.max_points); return VLOAM_ERR_CAPACITY; }
  if (n <= 0) { set_err("empty cloud"); return VLOAM_ERR_EMPTY; }
  SINGLE_SESSION_ONLY(h);
  HIPCHK(hipSetDevice(h->device));
  if (!g_stage_inline && !h->cfg.timing) return host_scan_deferred(h, &xyz_pad4, &n);
  STAGE_ONE(h, xyz_pad4, n, d);
  return vloam_process_scan_device(h, d, n);
}

// ------------------------------------------------------------------ coupled VLOAM frame (configs[3])
// == vloam_tf->processStaticTransform()'s products base_T_cam0 / velo_T_cam0 (vloam_tf.cpp:55-56), row-major 4x4
vloam_status vloam_set_extrinsics(vloam_handle* h, const double base_T_cam0[16], const double velo_T_cam0[16]) {
  if (!h || !base_T_cam0 || !velo_T_cam0) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  for (int b = 0; b < h->se.B; b++) {   // one sensor rig for all sessions of a batched handle
    LOState* lo = (LOState*)((char*)h->lo + (size_t)b * h->se.ss);
    VloamTfState tf;
    HIPCHK(hipMemcpy(&tf, &lo->tf, sizeof(tf), hipMemcpyDeviceToHost));
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) { tf.base_T_cam0.m[r * 3 + c] = base_T_cam0[r * 4 + c]; tf.velo_T_cam0.m[r * 3 + c] = velo_T_cam0[r * 4 + c]; }
      tf.base_T_cam0.o[r] = base_T_cam0[r * 4 + 3]; tf.velo_T_cam0.o[r] = velo_T_cam0[r * 4 + 3];
    }
    tf.coupled = 1;
    HIPCHK(hipMemcpy(&lo->tf, &tf, sizeof(tf), hipMemcpyHostToDevice));
  }
  h->have_extrinsics = true;
  return VLOAM_OK;
}

// One callback() of MAIN/src/vloam_main_node.cpp:125-180 without a host round trip: VO->reset / LOAM->reset, processPointCloud (depth map
// from the SAME device-resident sweep scan registration reads), solveNlsAll for count > 0 (initial guess = the previous frame's
// cam0_curr_LOT_cam0_prev, on the device), VO2VeloAndBase (-> velo_last_VOT_velo_curr, read by solveLO when detach_VO_LO == 0),
// scanRegistrationIO, laserOdometryIO (publish() refreshes cam0_curr_LOT_cam0_prev), laserMappingIO.
// prev_uv / curr_uv: n_match integer pixel pairs in HOST memory (previous frame -> this frame; ignored for the first frame).
// Host image -> the device staging buffer, on the image stream.  hipMemcpyAsync from pageable memory has taken its copy of the source when
// it returns; hipMemcpy2DAsync has NOT (measured: tools/microbench/pageable_async_copy.py) — so a padded image is packed on the host
// first, and the caller may reuse its buffer as soon as the call is back either way.
static vloam_status upload_image(vloam_handle* h, const unsigned char* gray, int width, int height, int stride, int session = 0) {
  const unsigned char* src = gray;
  if (stride != width) {
    h->img_pack.resize((size_t)width * height);
    for (int y = 0; y < height; y++) memcpy(h->img_pack.data() + (size_t)y * width, gray + (size_t)y * stride, (size_t)width);
    src = h->img_pack.data();
  }
  HIPCHK(hipMemcpyAsync(h->img.staging + (size_t)session * h->se.ss, src, (size_t)width * height, hipMemcpyHostToDevice, h->s_img));
  return VLOAM_OK;
}

static vloam_status process_frame_common(vloam_handle* h, const BatchIn& bi, const int* const* prev_uv, const int* const* curr_uv, const int* n_match,
                                         const unsigned char* const* d_gray /* one image per session, or null */, int width, int height, int stride) {
  if (!h->vo.have_calib || !h->have_extrinsics) { set_err("vloam_process_frame needs vloam_vo_set_calib and vloam_set_extrinsics first"); return VLOAM_ERR_ORDER; }
  for (int b = 0; b < h->se.B; b++) {
    if (n_match[b] < 0 || (n_match[b] > 0 && (!prev_uv[b] || !curr_uv[b]))) { set_err("bad match arrays for session %d", b); return VLOAM_ERR_INVALID; }
    if (n_match[b] > kVoMaxMatches) { set_err("%d matches exceed the capacity of %d", n_match[b], kVoMaxMatches); return VLOAM_ERR_CAPACITY; }
  }
  if (d_gray) for (int b = 0; b < h->se.B; b++) if (!d_gray[b]) { set_err("null image pointer for session %d", b); return VLOAM_ERR_INVALID; }
  if (d_gray && h->img.max_w == 0) { set_err("the handle was created without an image front-end (cfg.image_width / image_height)"); return VLOAM_ERR_ORDER; }
  if (d_gray && img_check(&h->img, width, height, stride) != VLOAM_OK) {   // before anything of this frame is enqueued
    set_err("image front-end: bad image size (%d x %d, stride %d; capacity %d x %d, one size per sequence)", width, height, stride, h->img.max_w, h->img.max_h);
    return VLOAM_ERR_INVALID;
  }
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s0 = flush_pending(h); if (s0 != VLOAM_OK) return s0; }   // a host sweep of vloam_process_scan still in flight: the older one first
  if (h->stage == 2) { vloam_status s0 = finish_frame(h); if (s0 != VLOAM_OK) return s0; }
  vloam_status s = enqueue_sr(h, bi);
  if (s != VLOAM_OK) return s;
  const int k = h->frame, cur = set_of(k);
  // depth map + matches ride on the scan-registration stream (they only need the sweep); the solve itself belongs to the odometry stream
  int no_match[kMaxBatch] = {0};
  s = vo_depth_enqueue(&h->vo, h->stream, bi, k, prev_uv, curr_uv, d_gray ? no_match : n_match, &h->prof);
  if (s != VLOAM_OK) { set_err("vo_depth_enqueue failed"); return s; }
  HIPCHK(hipEventRecord(h->ev_vo[cur], h->stream));
  h->vo_frame[cur] = true;
  if (d_gray) {
    // processImage on its own stream: corners + flow straight into this frame's match arrays (one entry per corner slot, untracked
    // slots marked), consumed by the VO solve in front of this frame's laser odometry
    // (a batched handle runs its sessions' images one after the other on that stream: each session has the image buffers of its own arena)
    const int vset = k % VOContext::kSets;
    ImgContext after = h->img;
    for (int b = 0; b < h->se.B; b++) {
      const size_t so = (size_t)b * h->se.ss;
      ImgContext cb = h->img.rebased(so);
      s = img_process(&cb, h->s_img, d_gray[b], width, height, stride, (int*)((char*)h->vo.d_prev_set[vset] + so), (int*)((char*)h->vo.d_curr_set[vset] + so), &h->prof);
      if (s != VLOAM_OK) { set_err("image front-end: bad image size (%d x %d, stride %d; capacity %d x %d, one size per sequence)", width, height, stride, h->img.max_w, h->img.max_h); return s; }
      h->vo.n_match_set[vset].n[b] = k > 0 ? kImgMaxCorners : 0;
      after = cb;
    }
    h->img.adopt_host_state(after);
    HIPCHK(hipEventRecord(h->ev_img[cur], h->s_img));
    h->img_frame[cur] = true;
  }
  if (h->cfg.timing) {
    s = enqueue_lo(h, h->frame);
    if (s != VLOAM_OK) return s;
    if (h->cfg.with_mapping) { s = enqueue_map(h, h->frame); if (s != VLOAM_OK) return s; }
    return finish_frame(h);
  }
  h->frame++;
  h->stage = 0;
  return drain_deferred(h, kLagLO, kLagMap);
}

vloam_status vloam_process_frame_device(vloam_handle* h, const void* d_xyz_pad4, int n, const int* prev_uv, const int* curr_uv, int n_match) {
  if (!h || !d_xyz_pad4 || n_match < 0 || (n_match > 0 && (!prev_uv || !curr_uv))) return VLOAM_ERR_INVALID;
  SINGLE_SESSION_ONLY(h);
  return process_frame_common(h, one_sweep(d_xyz_pad4, n), &prev_uv, &curr_uv, &n_match, nullptr, 0, 0, 0);
}

// Batched coupled frames: one call advances ALL sessions of the handle by one VLOAM frame (session b: sweep d_xyz_pad4[b] with n[b] points,
// n_match[b] pixel matches prev_uv[b] -> curr_uv[b] in HOST memory); every kernel of the frame chain — depth map, match + VO solve,
// VO2VeloAndBase, scan registration, odometry in combined mode, mapping — is launched once with the session index in blockIdx.z.
vloam_status vloam_batch_process_frame_device(vloam_handle* h, const void* const* d_xyz_pad4, const int* n, const int* const* prev_uv,
                                              const int* const* curr_uv, const int* n_match) {
  if (!h || !d_xyz_pad4 || !n || !prev_uv || !curr_uv || !n_match) return VLOAM_ERR_INVALID;
  BatchIn bi;
  memset(&bi, 0, sizeof(bi));
  for (int b = 0; b < h->se.B; b++) { bi.in[b] = (const float4*)d_xyz_pad4[b]; bi.n[b] = n[b]; }
  return process_frame_common(h, bi, prev_uv, curr_uv, n_match, nullptr, 0, 0, 0);
}

vloam_status vloam_batch_process_frame(vloam_handle* h, const float* const* xyz_pad4, const int* n, const int* const* prev_uv,
                                       const int* const* curr_uv, const int* n_match) {
  if (!h || !xyz_pad4 || !n || !prev_uv || !curr_uv || !n_match) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  BatchIn bi;
  memset(&bi, 0, sizeof(bi));
  for (int b = 0; b < h->se.B; b++) {
    if (!xyz_pad4[b]) return VLOAM_ERR_INVALID;
    if (n[b] > h->cfg.max_points) { set_err("cloud of %d points exceeds max_points=%d", n[b], h->cfg.max_points); return VLOAM_ERR_CAPACITY; }
    if (n[b] <= 0) { set_err("empty cloud"); return VLOAM_ERR_EMPTY; }
  }
  { vloam_status s_ = stage_begin(h); if (s_ != VLOAM_OK) return s_; }
  for (int b = 0; b < h->se.B; b++) {
    vloam_status s_ = stage_sweep(h, b, xyz_pad4[b], n[b], &bi.in[b]);
    if (s_ != VLOAM_OK) return stage_release(h, s_);
    bi.n[b] = n[b];
  }
  { vloam_status s_ = stage_end(h); if (s_ != VLOAM_OK) return stage_release(h, s_); }
  return stage_release(h, process_frame_common(h, bi, prev_uv, curr_uv, n_match, nullptr, 0, 0, 0));
}

vloam_status vloam_process_frame(vloam_handle* h, const float* xyz_pad4, int n, const int* prev_uv, const int* curr_uv, int n_match) {
  if (!h || !xyz_pad4) return VLOAM_ERR_INVALID;
  if (n > h->cfg.max_points) { set_err("cloud of %d points exceeds max_points=%d", n, h->cfg.max_points); return VLOAM_ERR_CAPACITY; }
  if (n <= 0) { set_err("empty cloud"); return VLOAM_ERR_EMPTY; }
  SINGLE_SESSION_ONLY(h);
  if (n_match < 0 || (n_match > 0 && (!prev_uv || !curr_uv))) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  STAGE_ONE(h, xyz_pad4, n, d);
  return stage_release(h, vloam_process_frame_device(h, d, n, prev_uv, curr_uv, n_match));
}

vloam_status vloam_process_frame_image_device(vloam_handle* h, const void* d_xyz_pad4, int n, const void* d_gray, int width, int height, int stride) {
  if (!h || !d_xyz_pad4 || !d_gray) return VLOAM_ERR_INVALID;
  SINGLE_SESSION_ONLY(h);
  { const int* none = nullptr; const int zero = 0;
    const unsigned char* g = (const unsigned char*)d_gray;
    return process_frame_common(h, one_sweep(d_xyz_pad4, n), &none, &none, &zero, &g, width, height, stride); }
}

vloam_status vloam_process_frame_image(vloam_handle* h, const float* xyz_pad4, int n, const unsigned char* gray, int width, int height, int stride) {
  if (!h || !xyz_pad4 || !gray) return VLOAM_ERR_INVALID;
  SINGLE_SESSION_ONLY(h);   // before anything is enqueued: VLOAM_ERR_INVALID has no side effects (c_api.h)
  if (n > h->cfg.max_points) { set_err("cloud of %d points exceeds max_points=%d", n, h->cfg.max_points); return VLOAM_ERR_CAPACITY; }
  if (n <= 0) { set_err("empty cloud"); return VLOAM_ERR_EMPTY; }
  if (h->img.max_w == 0) { set_err("the handle was created without an image front-end (cfg.image_width / image_height)"); return VLOAM_ERR_ORDER; }
  if (width <= 0 || height <= 0 || stride < width || width > h->img.max_w || height > h->img.max_h) { set_err("bad image size (%d x %d; the handle was created for at most %d x %d)", width, height, h->img.max_w, h->img.max_h); return VLOAM_ERR_INVALID; }
  HIPCHK(hipSetDevice(h->device));
  STAGE_ONE(h, xyz_pad4, n, d);
  { vloam_status s_ = upload_image(h, gray, width, height, stride); if (s_ != VLOAM_OK) return stage_release(h, s_); }
  { const int* none = nullptr; const int zero = 0;
    const unsigned char* g = h->img.staging;
    return stage_release(h, process_frame_common(h, one_sweep(d, n), &none, &none, &zero, &g, width, height, width)); }
}

// Batched coupled frames from raw inputs: session b gets sweep d_xyz_pad4[b] and the 8-bit grey image d_gray[b] (all images of one size).
vloam_status vloam_batch_process_frame_image_device(vloam_handle* h, const void* const* d_xyz_pad4, const int* n, const void* const* d_gray, int width, int height,
                                                    int stride) {
  if (!h || !d_xyz_pad4 || !n || !d_gray) return VLOAM_ERR_INVALID;
  BatchIn bi;
  memset(&bi, 0, sizeof(bi));
  const int* none[kMaxBatch];
  int zero[kMaxBatch];
  const unsigned char* g[kMaxBatch];
  for (int b = 0; b < h->se.B; b++) { bi.in[b] = (const float4*)d_xyz_pad4[b]; bi.n[b] = n[b]; none[b] = nullptr; zero[b] = 0; g[b] = (const unsigned char*)d_gray[b]; }
  return process_frame_common(h, bi, none, none, zero, g, width, height, stride);
}

vloam_status vloam_batch_process_frame_image(vloam_handle* h, const float* const* xyz_pad4, const int* n, const unsigned char* const* gray, int width, int height,
                                             int stride) {
  if (!h || !xyz_pad4 || !n || !gray) return VLOAM_ERR_INVALID;
  if (h->img.max_w == 0) { set_err("the handle was created without an image front-end (cfg.image_width / image_height)"); return VLOAM_ERR_ORDER; }
  if (width <= 0 || height <= 0 || stride < width || width > h->img.max_w || height > h->img.max_h) { set_err("bad image size (%d x %d; the handle was created for at most %d x %d)", width, height, h->img.max_w, h->img.max_h); return VLOAM_ERR_INVALID; }
  HIPCHK(hipSetDevice(h->device));
  const void* d_in[kMaxBatch];
  const void* d_img[kMaxBatch];
  for (int b = 0; b < h->se.B; b++) {
    if (!xyz_pad4[b] || !gray[b]) return VLOAM_ERR_INVALID;
    if (n[b] > h->cfg.max_points) { set_err("cloud of %d points exceeds max_points=%d", n[b], h->cfg.max_points); return VLOAM_ERR_CAPACITY; }
    if (n[b] <= 0) { set_err("empty cloud"); return VLOAM_ERR_EMPTY; }
  }
  { vloam_status s_ = stage_begin(h); if (s_ != VLOAM_OK) return s_; }
  for (int b = 0; b < h->se.B; b++) {
    const float4* dst = nullptr;
    { vloam_status s_ = stage_sweep(h, b, xyz_pad4[b], n[b], &dst); if (s_ != VLOAM_OK) return stage_release(h, s_); }
    { vloam_status s_ = upload_image(h, gray[b], width, height, stride, b); if (s_ != VLOAM_OK) return stage_release(h, s_); }
    d_in[b] = dst; d_img[b] = h->img.staging + (size_t)b * h->se.ss;
  }
  { vloam_status s_ = stage_end(h); if (s_ != VLOAM_OK) return stage_release(h, s_); }
  return stage_release(h, vloam_batch_process_frame_image_device(h, d_in, n, d_img, width, height, width));
}

// ---- the image front-end on its own (VisualOdometry::processImage, optical_flow_match = true)
vloam_status vloam_vo_process_image_device(vloam_handle* h, const void* d_gray, int width, int height, int stride) {
  if (!h || !d_gray) return VLOAM_ERR_INVALID;
  SINGLE_SESSION_ONLY(h);
  if (h->img.max_w == 0) { set_err("the handle was created without an image front-end (cfg.image_width / image_height)"); return VLOAM_ERR_ORDER; }
  HIPCHK(hipSetDevice(h->device));
  // (no match outputs here: vloam_vo_solve uploads host matches into d_prev / d_curr on another stream; nothing consumes device-side matches in standalone mode)
  vloam_status s = img_process(&h->img, h->s_img, (const unsigned char*)d_gray, width, height, stride, nullptr, nullptr, &h->prof);
  if (s != VLOAM_OK) set_err("image front-end: bad image size (%d x %d, stride %d; capacity %d x %d, one size per sequence)", width, height, stride, h->img.max_w, h->img.max_h);
  return s;
}

vloam_status vloam_vo_process_image(vloam_handle* h, const unsigned char* gray, int width, int height, int stride) {
  if (!h || !gray) return VLOAM_ERR_INVALID;
  if (h->img.max_w == 0) { set_err("the handle was created without an image front-end (cfg.image_width / image_height)"); return VLOAM_ERR_ORDER; }
  if (width <= 0 || height <= 0 || stride < width || width > h->img.max_w || height > h->img.max_h) { set_err("bad image size (%d x %d; the handle was created for at most %d x %d)", width, height, h->img.max_w, h->img.max_h); return VLOAM_ERR_INVALID; }
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = upload_image(h, gray, width, height, stride); if (s_ != VLOAM_OK) return s_; }
  return vloam_vo_process_image_device(h, h->img.staging, width, height, width);
}

// ImageUtil::matchDescriptors (image_util.cpp:221-296) for binary descriptors: BFMatcher(NORM_HAMMING), knnMatch k = 2 + ratio 0.8 (select_knn != 0,
// the reference's SelectType::KNN) or match with crossCheck (select_knn == 0, SelectType::NN).  Descriptors in host memory, n x bytes_per_desc.
vloam_status vloam_vo_match_descriptors(vloam_handle* h, const unsigned char* desc_prev, int n_prev, const unsigned char* desc_curr, int n_curr, int bytes_per_desc,
                                        int select_knn, int* query_idx, int* train_idx, int cap, int* n_matches) {
  if (!h || !n_matches || cap < 0 || (n_prev > 0 && !desc_prev) || (n_curr > 0 && !desc_curr) || (cap > 0 && (!query_idx || !train_idx))) return VLOAM_ERR_INVALID;
  SINGLE_SESSION_ONLY(h);
  if (h->img.max_w == 0) { set_err("the handle was created without an image front-end (cfg.image_width / image_height)"); return VLOAM_ERR_ORDER; }
  HIPCHK(hipSetDevice(h->device));
  vloam_status s = img_match_descriptors(&h->img, h->s_img, desc_prev, n_prev, desc_curr, n_curr, bytes_per_desc, select_knn != 0, query_idx, train_idx, cap, n_matches);
  if (s == VLOAM_ERR_CAPACITY) set_err("more than %d descriptors", kImgMaxDesc);
  if (s == VLOAM_ERR_INVALID) set_err("bytes_per_desc must be a multiple of 4 up to %d", kImgMaxDescBytes);
  return s;
}

static vloam_status img_results(vloam_handle* h, std::vector<float2>* corners, std::vector<float2>* tracked, std::vector<unsigned char>* status, int* n_corners,
                                bool* have_flow) {
  if (h->img.max_w == 0 || h->img.count < 0) { set_err("no image processed yet"); return VLOAM_ERR_ORDER; }
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  const ImgContext img = h->img.rebased((size_t)h->sel * h->se.ss);   // the session vloam_select_session chose
  int ierr = 0;
  HIPCHK(hipMemcpy(&ierr, img.error, sizeof(int), hipMemcpyDeviceToHost));
  if (ierr) { set_err("image front-end capacity exceeded (bits %d: 1 candidates > %d, 2 neighbours > %d, 4 corners > %d)", ierr, kImgCandCap, kImgNbrCap, kImgAccCap); return VLOAM_ERR_CAPACITY; }
  const int cur = h->img.count % 2;
  HIPCHK(hipMemcpy(n_corners, img.n_corners[cur], sizeof(int), hipMemcpyDeviceToHost));
  corners->resize((size_t)*n_corners + 1);
  if (*n_corners) HIPCHK(hipMemcpy(corners->data(), img.corners[cur], sizeof(float2) * (size_t)*n_corners, hipMemcpyDeviceToHost));
  *have_flow = h->img.count > 0 && !h->img.orb;   // ORB + brute-force tracks nothing: matches come from vloam_vo_get_flow_matches
  if (tracked && *have_flow && *n_corners) {
    tracked->resize((size_t)*n_corners); status->resize((size_t)*n_corners);
    HIPCHK(hipMemcpy(tracked->data(), img.tracked, sizeof(float2) * (size_t)*n_corners, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(status->data(), img.status, (size_t)*n_corners, hipMemcpyDeviceToHost));
  }
  return VLOAM_OK;
}

vloam_status vloam_vo_get_keypoints(vloam_handle* h, float* xy, int cap, int* n) {
  if (!h || !n || cap < 0 || (cap > 0 && !xy)) return VLOAM_ERR_INVALID;
  std::vector<float2> c;
  int nc = 0;
  bool flow = false;
  vloam_status s = img_results(h, &c, nullptr, nullptr, &nc, &flow);
  if (s != VLOAM_OK) return s;
  *n = nc;
  for (int k = 0; k < nc && k < cap; k++) { xy[2 * k] = c[(size_t)k].x; xy[2 * k + 1] = c[(size_t)k].y; }
  return VLOAM_OK;
}

vloam_status vloam_vo_get_flow(vloam_handle* h, float* prev_xy, float* curr_xy, unsigned char* status, int cap, int* n) {
  if (!h || !n || cap < 0 || (cap > 0 && (!prev_xy || !curr_xy || !status))) return VLOAM_ERR_INVALID;
  std::vector<float2> c, t;
  std::vector<unsigned char> st;
  int nc = 0;
  bool flow = false;
  vloam_status s = img_results(h, &c, &t, &st, &nc, &flow);
  if (s != VLOAM_OK) return s;
  *n = flow ? nc : 0;
  for (int k = 0; k < *n && k < cap; k++) {
    prev_xy[2 * k] = c[(size_t)k].x; prev_xy[2 * k + 1] = c[(size_t)k].y;
    curr_xy[2 * k] = t[(size_t)k].x; curr_xy[2 * k + 1] = t[(size_t)k].y;
    status[k] = st[(size_t)k];
  }
  return VLOAM_OK;
}

// ORB + brute-force configuration: the keypoints that survive ORB's border filter (what match indices refer to) and their descriptors
static vloam_status orb_results(vloam_handle* h, int which /* 0: the latest image, 1: the one before */, std::vector<float2>* kp, std::vector<unsigned char>* desc) {
  if (h->img.max_w == 0 || h->img.count < 0) { set_err("no image processed yet"); return VLOAM_ERR_ORDER; }
  if (!h->img.orb) { set_err("the handle runs the optical-flow configuration (no ORB pattern set: vloam_vo_set_orb_pattern)"); return VLOAM_ERR_ORDER; }
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  const ImgContext img = h->img.rebased((size_t)h->sel * h->se.ss);
  const int slot = (h->img.count + (which ? 1 : 0)) % 2;
  int n = 0;
  if (!(which && h->img.count == 0)) HIPCHK(hipMemcpy(&n, img.n_okp[slot], sizeof(int), hipMemcpyDeviceToHost));
  kp->resize((size_t)n);
  if (n) HIPCHK(hipMemcpy(kp->data(), img.okp[slot], sizeof(float2) * (size_t)n, hipMemcpyDeviceToHost));
  if (desc) {
    std::vector<unsigned> rows((size_t)n * (kImgMaxDescBytes / 4));
    if (n) HIPCHK(hipMemcpy(rows.data(), img.desc[slot], rows.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    desc->resize((size_t)n * 32);
    for (int k = 0; k < n; k++) memcpy(desc->data() + (size_t)k * 32, rows.data() + (size_t)k * (kImgMaxDescBytes / 4), 32);
  }
  return VLOAM_OK;
}

vloam_status vloam_vo_set_orb_pattern(vloam_handle* h, const signed char* pattern_256x4) {
  if (!h) return VLOAM_ERR_INVALID;
  if (h->img.max_w == 0) { set_err("the handle was created without an image front-end (cfg.image_width / image_height)"); return VLOAM_ERR_ORDER; }
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  if (h->img.count >= 0 && (pattern_256x4 != nullptr) != h->img.orb) { set_err("the image configuration cannot change in the middle of a sequence"); return VLOAM_ERR_ORDER; }
  const vloam_status s = img_set_orb_pattern(&h->img, h->stream, pattern_256x4, h->se.B, h->se.ss);
  if (s == VLOAM_ERR_INVALID) set_err("ORB pattern: a steered offset leaves the 31-pixel border the keypoints keep (|x|, |y| <= 30)");
  return s;
}

vloam_status vloam_vo_get_descriptors(vloam_handle* h, float* xy, unsigned char* desc32, int cap, int* n) {
  if (!h || !n || cap < 0) return VLOAM_ERR_INVALID;
  std::vector<float2> kp;
  std::vector<unsigned char> d;
  vloam_status s = orb_results(h, 0, &kp, &d);
  if (s != VLOAM_OK) return s;
  *n = (int)kp.size();
  const int m = *n < cap ? *n : cap;
  for (int k = 0; xy && k < m; k++) { xy[2 * k] = kp[(size_t)k].x; xy[2 * k + 1] = kp[(size_t)k].y; }
  if (desc32 && m > 0) memcpy(desc32, d.data(), (size_t)m * 32);
  return VLOAM_OK;
}

vloam_status vloam_vo_get_flow_matches(vloam_handle* h, int* prev_uv, int* curr_uv, int cap, int* n) {
  if (!h || !n || cap < 0 || (cap > 0 && (!prev_uv || !curr_uv))) return VLOAM_ERR_INVALID;
  if (h->img.max_w != 0 && h->img.orb) {
    // ORB + brute force: matchDescriptors(previous, latest) + the match loop's reads (visual_odometry.cpp:113-116,296-303), from what the device left
    std::vector<float2> kc, kp;
    vloam_status s = orb_results(h, 0, &kc, nullptr);
    if (s == VLOAM_OK) s = orb_results(h, 1, &kp, nullptr);
    if (s != VLOAM_OK) return s;
    *n = 0;
    if (h->img.count == 0 || kp.empty() || kc.size() < 2) return VLOAM_OK;
    const ImgContext img = h->img.rebased((size_t)h->sel * h->se.ss);
    std::vector<uint2> best(kp.size());
    HIPCHK(hipMemcpy(best.data(), img.best2[0], sizeof(uint2) * best.size(), hipMemcpyDeviceToHost));
    int m = 0;
    for (size_t q = 0; q < kp.size(); q++) {
      const uint2 b = best[q];
      if (b.y == 0xffffffffu || !((double)(float)(b.x >> 16) < 0.8 * (double)(float)(b.y >> 16))) continue;
      const float2 a = kp[q], c = kc[(size_t)(b.x & 0xffffu)];
      if (m < cap) { prev_uv[2 * m] = (int)a.x; prev_uv[2 * m + 1] = (int)a.y; curr_uv[2 * m] = (int)c.x; curr_uv[2 * m + 1] = (int)c.y; }
      m++;
    }
    *n = m;
    return VLOAM_OK;
  }
  std::vector<float2> c, t;
  std::vector<unsigned char> st;
  int nc = 0, m = 0;
  bool flow = false;
  vloam_status s = img_results(h, &c, &t, &st, &nc, &flow);
  if (s != VLOAM_OK) return s;
  for (int k = 0; flow && k < nc; k++) {
    if (st[(size_t)k] != 1) continue;   // visual_odometry.cpp:298-308
    if (m < cap) {
      prev_uv[2 * m] = (int)c[(size_t)k].x; prev_uv[2 * m + 1] = (int)c[(size_t)k].y;
      curr_uv[2 * m] = (int)t[(size_t)k].x; curr_uv[2 * m + 1] = (int)t[(size_t)k].y;
    }
    m++;
  }
  *n = m;
  return VLOAM_OK;
}

// world_VOT_base_last of frames first..first+count-1 as {q xyzw, t} (what VO2Cam0StartFrame turns into VO rows, vloam_tf.cpp:77-101)
vloam_status vloam_get_vo_trajectory(vloam_handle* h, int first, int count, double* poses7) {
  if (!h || !poses7 || first < 0 || count < 0 || first + count > h->frame + (h->pend.valid ? 1 : 0)) return VLOAM_ERR_INVALID;   // (a host sweep in flight is enqueued by the sync below)
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  if (count) HIPCHK(hipMemcpy(poses7, SEL(h, h->vo_traj) + (size_t)first * 7, sizeof(double) * 7 * (size_t)count, hipMemcpyDeviceToHost));
  return VLOAM_OK;
}

// the last frame's VO estimate (angles_0to1, t_0to1), its counter32 / counter22 and the LiDAR-odometry prior derived from it
vloam_status vloam_get_vo_result(vloam_handle* h, double angle_axis[3], double t[3], int counters32_22[2], double prior_q[4], double prior_t[3]) {
  if (!h) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  double x[6];
  int cnt[2];
  LOState lo;
  HIPCHK(hipMemcpy(x, SEL(h, h->vo.x), sizeof(x), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cnt, SEL(h, h->vo.counters), sizeof(cnt), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&lo, SEL(h, h->lo), sizeof(lo), hipMemcpyDeviceToHost));
  for (int k = 0; k < 3; k++) { if (angle_axis) angle_axis[k] = x[k]; if (t) t[k] = x[3 + k]; if (prior_t) prior_t[k] = lo.prior_t[k]; }
  if (counters32_22) { counters32_22[0] = cnt[0]; counters32_22[1] = cnt[1]; }
  if (prior_q) for (int k = 0; k < 4; k++) prior_q[k] = lo.prior_q[k];
  return VLOAM_OK;
}

vloam_status vloam_sync(vloam_handle* h) {
  if (!h) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  if (h->frame > 0) {
    // surface sticky device-side errors: scan-registration bits of ANY sweep since the last vloam_sync (k_sr_compact folds every
    // sweep's word into the handle's sticky word; reported once, then cleared), map / solver bits for good
    int merr = 0;
    long long fb = 0;
    vloam_status s = map_error(&h->map, &merr, kErrEmpty | kErrRingTooLong, &fb);
    if (s != VLOAM_OK) return s;
    if (fb > h->fallback_solves) {
      // a cooperative solve found its partner workgroups missing and finished on one workgroup (lm_solve.hip): same answer, ~0.5 s late.
      // Whatever kept them apart (a CU-masked or partitioned device, another process' solves holding the compute units) is likely to last:
      // this handle launches one-workgroup solves from now on.
      h->se.no_coop = 1; h->map.se.no_coop = 1; h->vo.se.no_coop = 1;
    }
    h->fallback_solves = fb;
    if (merr & kErrEmpty) { set_err("no point survived NaN / minimum_range removal in at least one sweep since the last vloam_sync"); return VLOAM_ERR_EMPTY; }
    if (merr & kErrRingTooLong) { set_err("a ring held more than %d points (dropped) in at least one sweep since the last vloam_sync", kMaxRingLen); return VLOAM_ERR_CAPACITY; }
    if (merr & kErrMapFull) { set_err("voxel hash full (map_capacity_log2=%d)", h->cfg.map_capacity_log2); return VLOAM_ERR_CAPACITY; }
    if (merr & kErrStackFull) { set_err("mapping factor table full"); return VLOAM_ERR_CAPACITY; }
    if (merr & kErrMapDeferred) { set_err("raw-point capacity of the map exceeded (more than 255 un-merged points in a voxel of a cube outside the valid block, or more than 64 raw voxels around one query)"); return VLOAM_ERR_CAPACITY; }
    if (merr & kErrSolverSync) { set_err("a workgroup of the scan-feature VoxelGrid gave up waiting for the bins in front of it"); return VLOAM_ERR_HIP; }
    if (merr & kErrVoDegenerate) { set_err("a VO solve returned a zero rotation angle: poses are NaN from that frame on, as in the reference (visual_odometry.cpp:427-430)"); return VLOAM_ERR_INVALID; }
  }
  if (h->img.max_w != 0 && h->img.count >= 0) {
    // the image front-end's own sticky word: a corner / match set cut by one of its capacities differs from goodFeaturesToTrack's and was
    // fed into the VO solve of the coupled loop — say so here too, not only in the vloam_vo_get_* getters
    for (int b = 0; b < h->se.B; b++) {
      int ierr = 0;
      HIPCHK(hipMemcpy(&ierr, (const char*)h->img.error + (size_t)b * h->se.ss, sizeof(int), hipMemcpyDeviceToHost));
      if (ierr) { set_err("image front-end capacity exceeded in session %d (bits %d: 1 candidates > %d, 2 neighbours > %d, 4 corners > %d)", b, ierr, kImgCandCap, kImgNbrCap, kImgAccCap); return VLOAM_ERR_CAPACITY; }
    }
  }
  return VLOAM_OK;
}

vloam_status vloam_get_trajectory(vloam_handle* h, int first, int count, double* poses14) {
  if (!h || !poses14 || first < 0 || count < 0 || first + count > h->frame + (h->pend.valid ? 1 : 0)) return VLOAM_ERR_INVALID;   // (a host sweep in flight is enqueued by the sync below)
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  if (count) HIPCHK(hipMemcpy(poses14, SEL(h, h->traj) + (size_t)first * 14, sizeof(double) * 14 * (size_t)count, hipMemcpyDeviceToHost));
  return VLOAM_OK;
}
vloam_status vloam_frame_count(vloam_handle* h, int* frames) {
  if (!h || !frames) return VLOAM_ERR_INVALID;
  *frames = h->frame + (h->pend.valid ? 1 : 0);   // sweeps handed over (the last host sweep may still be in flight: deferred ring)
  return VLOAM_OK;
}
vloam_status vloam_trajectory_device_ptr(vloam_handle* h, void** d_ptr, long long* bytes) {
  if (!h || !d_ptr || !bytes) return VLOAM_ERR_INVALID;
  *d_ptr = SEL(h, h->traj);
  *bytes = (long long)h->cfg.max_frames * 14 * (long long)sizeof(double);
  return VLOAM_OK;
}

// ------------------------------------------------------------------ VO
vloam_status vloam_vo_set_calib(vloam_handle* h, const vloam_calib* c) {
  if (!h || !c) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  return vo_set_calib(&h->vo, h->stream, c) == VLOAM_OK ? VLOAM_OK : VLOAM_ERR_HIP;
}
vloam_status vloam_vo_process_point_cloud(vloam_handle* h, const float* xyz_pad4, int n) {
  if (!h || !xyz_pad4 || n <= 0) return VLOAM_ERR_INVALID;
  SINGLE_SESSION_ONLY(h);
  if (n > h->cfg.max_points) return VLOAM_ERR_CAPACITY;
  HIPCHK(hipSetDevice(h->device));
  STAGE_ONE(h, xyz_pad4, n, d);
  return stage_release(h, vo_process_point_cloud(&h->vo, h->stream, d, n) == VLOAM_OK ? VLOAM_OK : VLOAM_ERR_HIP);
}
vloam_status vloam_vo_solve(vloam_handle* h, const int* prev_uv, const int* curr_uv, int n_match, double aa[3], double t[3], int counters[2]) {
  if (!h || !prev_uv || !curr_uv || !aa || !t || n_match < 0) return VLOAM_ERR_INVALID;
  SINGLE_SESSION_ONLY(h);
  HIPCHK(hipSetDevice(h->device));
  vloam_status s = vo_solve(&h->vo, h->cfg, h->stream, prev_uv, curr_uv, n_match, aa, t, counters);
  if (s != VLOAM_OK) set_err("vo_solve failed: %s", hipGetErrorString(hipGetLastError()));
  return s;
}

// ------------------------------------------------------------------ parity hooks
static vloam_status copy_out(const void* d_src, size_t bytes, void* buf, long long cap, long long* n) {
  if (n) *n = (long long)bytes;
  size_t m = bytes < (size_t)cap ? bytes : (size_t)cap;
  if (buf && m) HIPCHK(hipMemcpy(buf, d_src, m, hipMemcpyDeviceToHost));
  return VLOAM_OK;
}

vloam_status vloam_debug_get(vloam_handle* h, int stage, int item, void* buf, long long cap, long long* n) {
  if (!h) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  const int f = (h->stage == 0 && h->frame > 0) ? h->frame - 1 : h->frame;
  const int cur = set_of(f);
  if (stage == 0) {
    SRBuffers b = h->sr[cur];
    b.rebase((size_t)h->sel * h->se.ss);
    FrameScalars S;
    HIPCHK(hipMemcpy(&S, b.S, sizeof(S), hipMemcpyDeviceToHost));
    switch (item) {
      case 0: return copy_out(b.dbg_curv, sizeof(float) * S.N2, buf, cap, n);
      case 1: return copy_out(b.dbg_sort, sizeof(int) * S.N2, buf, cap, n);
      case 2: return copy_out(b.dbg_picked, sizeof(int) * S.N2, buf, cap, n);
      case 3: return copy_out(b.dbg_label, sizeof(int) * S.N2, buf, cap, n);
      case 11: return copy_out(b.dbg_cyc, sizeof(long long) * kMaxRings * 8, buf, cap, n);
      case 4: return copy_out(&b.S->scanStartInd[0], sizeof(int) * kMaxRings, buf, cap, n);
      case 5: return copy_out(&b.S->scanEndInd[0], sizeof(int) * kMaxRings, buf, cap, n);
      case 6: return copy_out(b.dbg_feat_idx, sizeof(int) * S.n_sharp, buf, cap, n);
      case 7: return copy_out(b.dbg_feat_idx + kMaxLessSharp, sizeof(int) * S.n_less_sharp, buf, cap, n);
      case 8: return copy_out(b.dbg_feat_idx + 2 * kMaxLessSharp, sizeof(int) * S.n_flat, buf, cap, n);
      case 9: {
        float sc[5] = {S.startOri, S.endOri, (float)S.istar, (float)S.n_after_s1, (float)S.N2};
        if (n) *n = sizeof(sc);
        if (buf) memcpy(buf, sc, (size_t)cap < sizeof(sc) ? (size_t)cap : sizeof(sc));
        return VLOAM_OK;
      }
      case 10: return copy_out(b.S, sizeof(FrameScalars), buf, cap, n);
    }
    return VLOAM_ERR_INVALID;
  }
  if (stage == 1) {
    const int outer = item / 16, k = item % 16;
    if (outer < 0 || outer > 1) return VLOAM_ERR_INVALID;
    switch (k) {
      case 0: return copy_out(SEL(h, h->lo_corr[outer]), sizeof(int) * 4 * kMaxSharp, buf, cap, n);
      case 1: return copy_out(SEL(h, h->lo_corr[outer]) + 4 * kMaxSharp, sizeof(int) * 4 * kMaxFlat, buf, cap, n);
      case 2: return copy_out(SEL(h, h->lo_rec) + outer, sizeof(LMRecord), buf, cap, n);
      case 3: return copy_out(SEL(h, h->lo_resid[outer]), sizeof(double) * 3 * kMaxLoFactors, buf, cap, n);
      case 6: return copy_out(SEL(h, h->lo_queue), sizeof(int) * kMaxLoFactors, buf, cap, n);   // the queue itself: slot | reason << 16 (1 block over capacity, 2 closest point beyond 1 m, 3 dense 5 m neighbourhood)
      case 5: return copy_out(SEL(h, h->lo_queue_n), sizeof(int) * 2, buf, cap, n);   // queries k_lo_assoc_fast left to the wave-per-query pass (last launch pair of a batch)
      case 4: if (!h->lo_cyc[outer]) return VLOAM_ERR_INVALID;
              return copy_out(SEL(h, h->lo_cyc[outer]), sizeof(long long) * 4 * kMaxLoFactors, buf, cap, n);
    }
    return VLOAM_ERR_INVALID;
  }
  if (stage == 2 && item == 70) {  // test hook: rebuild both voxel tables now (the production trigger is k_map_finalize's host-mapped flag)
    if (n) *n = 0;
    return h->cfg.with_mapping ? map_force_rebuild(&h->map, h->s_map) : VLOAM_OK;
  }
  if (stage == 2) return map_debug_get(&h->map, item, buf, cap, n);
  if (stage == 3) return vo_debug_get(&h->vo, item, buf, cap, n);
  if (stage == 4) return img_debug_get(&h->img, item, buf, cap, n);
  return VLOAM_ERR_INVALID;
}

// Per-kernel HIP-event timer: every launch of the named kernel (its __global__ symbol, e.g. "k_lo_assoc") is
// bracketed by an event pair on the handle's stream; max_launches bounds the event pool.  name == "" disables.
vloam_status vloam_profile_kernel(vloam_handle* h, const char* name, int max_launches) {
  if (!h || !name) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  int id = kKNone;
  for (int k = 1; k < kKCount; k++) if (strcmp(name, kKernelNames[k]) == 0) id = k;
  if (strcmp(name, "*") == 0) id = kKAll;  // every launch of every kernel (vloam_profile_read_table)
  if (id == kKNone && name[0] != 0) { set_err("unknown kernel %s", name); return VLOAM_ERR_INVALID; }
  while ((int)h->prof_events.size() < 2 * max_launches) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    h->prof_events.push_back(e);
  }
  h->prof_kids.assign((size_t)max_launches + 1, 0);
  h->prof.id = id;
  h->prof.kid_of = h->prof_kids.data();
  h->prof.ev = h->prof_events.data();
  h->prof.cap = max_launches;
  h->prof.used = 0;
  return VLOAM_OK;
}
vloam_status vloam_profile_read(vloam_handle* h, double* total_ms, int* launches) {
  if (!h || !total_ms || !launches) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  double tot = 0;
  for (int k = 0; k < h->prof.used; k++) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, h->prof.ev[2 * k], h->prof.ev[2 * k + 1]));
    tot += ms;
  }
  *total_ms = tot;
  *launches = h->prof.used;
  h->prof.used = 0;
  return VLOAM_OK;
}

// Per-kernel totals of the recorded launches (all-kernel mode "*", or the one selected kernel): ms[k], launches[k] for kernel id k
// < n_kernels; names through vloam_profile_kernel_name.  Resets the recorder like vloam_profile_read.
vloam_status vloam_profile_read_table(vloam_handle* h, int n_kernels, double* ms, int* launches) {
  if (!h || !ms || !launches || n_kernels < 0) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  for (int k = 0; k < n_kernels; k++) { ms[k] = 0; launches[k] = 0; }
  for (int k = 0; k < h->prof.used; k++) {
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, h->prof.ev[2 * k], h->prof.ev[2 * k + 1]));
    const int kid = h->prof.kid_of[k];
    if (kid >= 0 && kid < n_kernels) { ms[kid] += t; launches[kid]++; }
  }
  h->prof.used = 0;
  return VLOAM_OK;
}
int vloam_profile_kernel_count(void) { return kKCount; }
const char* vloam_profile_kernel_name(int k) { return (k >= 0 && k < kKCount) ? kKernelNames[k] : ""; }

// {cooperative solves that degraded to one workgroup (all sessions), 1 if the handle has switched to one-workgroup solves, voxel-table
// rebuilds, 0 ...}
vloam_status vloam_get_health(vloam_handle* h, long long out8[8]) {
  if (!h || !out8) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  for (int k = 0; k < 8; k++) out8[k] = 0;
  int merr = 0;
  long long fb = 0;
  if (h->frame > 0) { vloam_status s = map_error(&h->map, &merr, 0, &fb); if (s != VLOAM_OK) return s; }
  out8[0] = fb; out8[1] = h->se.no_coop; out8[2] = h->map.rebuilds;
  return VLOAM_OK;
}

vloam_status vloam_get_stage_ms(vloam_handle* h, double ms4[4], int* scans) {
  if (!h || !ms4) return VLOAM_ERR_INVALID;
  for (int k = 0; k < 4; k++) ms4[k] = h->stage_ms[k];
  if (scans) *scans = h->timed_scans;
  return VLOAM_OK;
}

vloam_status vloam_get_counts(vloam_handle* h, long long c[16]) {
  if (!h || !c) return VLOAM_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  { vloam_status s_ = sync_all(h); if (s_ != VLOAM_OK) return s_; }
  memset(c, 0, sizeof(long long) * 16);
  if (h->frame == 0) return VLOAM_OK;
  const int f = (h->stage == 0) ? h->frame - 1 : h->frame;
  const int cur = set_of(f), prev = set_of(f + vloam_handle::kSets - 1);
  FrameScalars S, Sp;
  HIPCHK(hipMemcpy(&S, SEL(h, h->sr[cur].S), sizeof(S), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&Sp, SEL(h, h->sr[prev].S), sizeof(Sp), hipMemcpyDeviceToHost));
  LMRecord rec[2];
  HIPCHK(hipMemcpy(rec, SEL(h, h->lo_rec), sizeof(rec), hipMemcpyDeviceToHost));
  c[0] = h->last_n_in; c[1] = S.N2; c[2] = S.n_sharp; c[3] = S.n_less_sharp; c[4] = S.n_flat; c[5] = S.n_less_flat;
  c[6] = Sp.n_less_sharp; c[7] = Sp.n_less_flat;
  if (f > 0) {
    // factor counts of the 2nd outer round; evaluations summed over both rounds
    int corr[4 * kMaxLoFactors];
    HIPCHK(hipMemcpy(corr, SEL(h, h->lo_corr[1]), sizeof(corr), hipMemcpyDeviceToHost));
    for (int k = 0; k < kMaxLoFactors; k++) if (corr[4 * k] >= 0) c[k < kMaxSharp ? 8 : 9]++;
    c[10] = (long long)(rec[0].n_evals + rec[1].n_evals);
  }
  return map_counts(&h->map, c);
}

}  // extern "C"
