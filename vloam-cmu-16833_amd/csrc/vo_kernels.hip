// Depth-enhanced visual-odometry residual stack on gfx950 (config 4).
//   k_vo_project  grid   N x 4 f32 homogeneous points through cam_T_velo^T, rect0_T_cam^T, P_rect0^T (PCU:148-174), bucket id
//   k_vo_scan     1 WG   bucket counts -> segment offsets
//   k_vo_scatter  grid   group point indices by 5-px bucket
//   k_vo_fold     grid   per bucket: input-ordered "incremental average" of (u, v, depth) exactly as written (PCU:205-260)
//   k_vo_match    grid   per match: outlier gate, queryDepth 3-NN inverse-distance lookup (PCU:302-387), K^-1 by f32
//                        column-pivoted QR, CostFunctor32 / CostFunctor22 factor (VO:283-416)
//   (k_lm_solve)         Huber(0.1), DENSE_QR-equivalent LM, <= 100 iterations on (angle-axis, t) (VO:67-68,423)
// "PCU" = /root/reference/src/visual_odometry/src/point_cloud_util.cpp, "VO" = .../visual_odometry.cpp.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <string.h>
#include "lm_solve.h"
#include "vo_kernels.h"

namespace vloam {

// (every kernel carries the session index of a batched handle in blockIdx.z and rebases its arena pointers by blockIdx.z * ss; the
// sweeps themselves are the callers' buffers: BatchIn)
__global__ __launch_bounds__(256) void k_vo_project(BatchIn bi, const vloam_calib* __restrict__ c,
                                                    float4* __restrict__ uvd, int* __restrict__ bcount, size_t ss) {
  VL_SESSION(ss); RB(c); RB(uvd); RB(bcount);
  const float4* __restrict__ in = bi.in[blockIdx.z];
  const int n = bi.n[blockIdx.z];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float4 q = in[i];
    const float t[4] = {q.x, q.y, q.z, 1.0f};
    float a[4], b[4], p[3];
    // three f32 GEMMs evaluated left to right, K = 4 inner products accumulated in k order
    for (int r = 0; r < 4; r++) a[r] = ((t[0] * c->cam_T_velo[r * 4 + 0] + t[1] * c->cam_T_velo[r * 4 + 1]) + t[2] * c->cam_T_velo[r * 4 + 2]) + t[3] * c->cam_T_velo[r * 4 + 3];
    for (int r = 0; r < 4; r++) b[r] = ((a[0] * c->rect0_T_cam[r * 4 + 0] + a[1] * c->rect0_T_cam[r * 4 + 1]) + a[2] * c->rect0_T_cam[r * 4 + 2]) + a[3] * c->rect0_T_cam[r * 4 + 3];
    for (int r = 0; r < 3; r++) p[r] = ((b[0] * c->P_rect0[r * 4 + 0] + b[1] * c->P_rect0[r * 4 + 1]) + b[2] * c->P_rect0[r * 4 + 2]) + b[3] * c->P_rect0[r * 4 + 3];
    int bucket = -1;
    float u = 0.f, v = 0.f;
    if (p[2] > 0.1f) {                  // PCU:156-158 (Eigen: f32 array > Scalar(0.1))
      const float inv = 1.0f / p[2];    // PCU:171-173
      u = p[0] * inv; v = p[1] * inv;
      const int ix = (int)(u / kGrid), iy = (int)(v / kGrid);  // PCU:218-219 (truncation toward zero)
      if (ix >= 0 && ix < kBW && iy >= 0 && iy < kBH) { bucket = ix * kBH + iy; atomicAdd(&bcount[bucket], 1); }
    }
    uvd[i] = make_float4(u, v, p[2], __int_as_float(bucket));
  }
}

__global__ __launch_bounds__(1024) void k_vo_scan(int* bcount, int* bfill, size_t ss) {
  VL_SESSION(ss); RB(bcount); RB(bfill);
  constexpr int kPer = (kBuckets + 1023) / 1024;  // 19: odd, so a lane stride of kPer words is LDS-bank-conflict free
  __shared__ int buf[1024 * kPer];                 // the counters pass through LDS: coalesced in, coalesced out
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < 1024 * kPer; k += 1024) buf[k] = k < kBuckets ? bcount[k] : 0;
  __syncthreads();
  const int lo = tid * kPer;
  int v[kPer], s = 0;
#pragma unroll
  for (int k = 0; k < kPer; k++) { v[k] = buf[lo + k]; s += v[k]; }
  int inc = s;
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int run = inc - s;
  for (int w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
  for (int k = 0; k < kPer; k++) { buf[lo + k] = run; run += v[k]; }
  __syncthreads();
  for (int k = tid; k <= kBuckets; k += 1024) { bcount[k] = buf[k]; if (k < kBuckets) bfill[k] = 0; }  // buf[kBuckets] == total (the tail counters are 0)
}

__global__ __launch_bounds__(256) void k_vo_scatter(const float4* __restrict__ uvd, BatchIn bi, const int* __restrict__ boff, int* bfill,
                                                    int* __restrict__ seg, size_t ss) {
  VL_SESSION(ss); RB(uvd); RB(boff); RB(bfill); RB(seg);
  const int n = bi.n[blockIdx.z];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int b = __float_as_int(uvd[i].w);
    if (b < 0) continue;
    seg[boff[b] + atomicAdd(&bfill[b], 1)] = i;
  }
}

__global__ __launch_bounds__(256) void k_vo_fold(const float4* __restrict__ uvd, int* __restrict__ boff, int* __restrict__ seg, DepthMapDev M, size_t ss) {
  VL_SESSION(ss); RB(uvd); RB(boff); RB(seg); M.rebase(so_);
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= kBuckets) return;
  const int b0 = boff[b], cnt = boff[b + 1] - b0;
  int* s = seg + b0;
  for (int a = 1; a < cnt; a++) {  // input order
    const int v = s[a];
    int c = a - 1;
    while (c >= 0 && s[c] > v) { s[c + 1] = s[c]; c--; }
    s[c + 1] = v;
  }
  float x = 0.f, y = 0.f, d = 0.f;
  int count = 0;
  for (int a = 0; a < cnt; a++) {
    const float4 p = uvd[s[a]];
    if (count == 0) { x = p.x; y = p.y; d = p.z; }
    else {  // "incremental averaging" with the count BEFORE the increment, verbatim (PCU:230-235)
      x += (p.x - x) / count; y += (p.y - y) / count; d += (p.z - d) / count;
    }
    ++count;
  }
  M.bx[b] = x; M.by[b] = y; M.bd[b] = d; M.bc[b] = count;
}

// PCU:302-387.  The reference collects the occupied buckets of the 5 x 5 neighbourhood, std::sorts them by distance and
// interpolates over the three nearest (if at least 10 are occupied).  Only the three smallest distances are ever used, so
// they are kept in registers by stable insertion (ties keep scan order == the canonical tie order of the sort).
__device__ __forceinline__ float query_depth(const DepthMapDev& M, float x, float y) {
  const int searching_radius = 2;
  const int index_x = (int)(x / kGrid), index_y = (int)(y / kGrid);
  float d0 = 3.0e38f, d1 = 3.0e38f, d2 = 3.0e38f, z0 = 0.f, z1 = 0.f, z2 = 0.f;  // distances ascending, depths alongside
  int cnt = 0;
  // two fully unrolled passes so that the 25 occupancy loads, then the loads of the occupied buckets, are each in flight
  // together (a loop over the neighbourhood would be 25 dependent round trips)
  int occ[25];
#pragma unroll
  for (int q = 0; q < 25; q++) {
    const int ix = index_x - searching_radius + q / 5, iy = index_y - searching_radius + q % 5;  // same order as the reference's loops
    occ[q] = (ix >= 0 && ix < kBW && iy >= 0 && iy < kBH) ? M.bc[ix * kBH + iy] : 0;
  }
  float bxs[25], bys[25], bds[25];
#pragma unroll
  for (int q = 0; q < 25; q++) {
    const int b = (index_x - searching_radius + q / 5) * kBH + (index_y - searching_radius + q % 5);
    bxs[q] = 0.f; bys[q] = 0.f; bds[q] = 0.f;
    if (occ[q] > 0) { bxs[q] = M.bx[b]; bys[q] = M.by[b]; bds[q] = M.bd[b]; }
  }
#pragma unroll
  for (int q = 0; q < 25; q++)
    if (occ[q] > 0) {
      const double dx = (double)(x - bxs[q]), dy = (double)(y - bys[q]);
      const float dist = (float)sqrt(dx * dx + dy * dy);  // std::sqrt(std::pow(float, 2) + std::pow(float, 2)) in double
      const float bd = bds[q];
      if (dist < d2) {
        if (dist < d1) {
          d2 = d1; z2 = z1;
          if (dist < d0) { d1 = d0; z1 = z0; d0 = dist; z0 = bd; }
          else { d1 = dist; z1 = bd; }
        } else { d2 = dist; z2 = bd; }
      }
      cnt++;
    }
  if (cnt < 10) return -1.0f;
  return (z0 * d1 * d2 + z1 * d0 * d2 + z2 * d0 * d1) / (0.0001f + d1 * d2 + d0 * d2 + d0 * d1);
}

// op-for-op twin of oracle/orc_vo.cpp solve3x3_colpiv_qr_f32 (P_rect0.leftCols(3).colPivHouseholderQr().solve, VO:350-355)
__device__ void solve3x3_colpiv_qr_f32(const float* A_, const float* b_, float* x) {
  float A[3][3], b[3] = {b_[0], b_[1], b_[2]};
  int perm[3] = {0, 1, 2};
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) A[r][c] = A_[r * 3 + c];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    int best = k; float bestn = -1.0f;
    for (int c = k; c < 3; c++) {
      float s = 0.0f;
      for (int r = k; r < 3; r++) s += A[r][c] * A[r][c];
      if (s > bestn) { bestn = s; best = c; }
    }
    if (best != k) {
      for (int r = 0; r < 3; r++) { const float t = A[r][k]; A[r][k] = A[r][best]; A[r][best] = t; }
      const int t = perm[k]; perm[k] = perm[best]; perm[best] = t;
    }
    const float nrm = sqrtf(bestn);
    if (nrm == 0.0f) continue;
    const float alpha = A[k][k] > 0.0f ? -nrm : nrm;
    float v[3] = {0, 0, 0};
    v[k] = A[k][k] - alpha;
    float vtv = v[k] * v[k];
    for (int r = k + 1; r < 3; r++) { v[r] = A[r][k]; vtv += v[r] * v[r]; }
    if (vtv != 0.0f) {
      for (int c = k + 1; c < 3; c++) {
        float s = 0.0f;
        for (int r = k; r < 3; r++) s += v[r] * A[r][c];
        s = 2.0f * s / vtv;
        for (int r = k; r < 3; r++) A[r][c] -= s * v[r];
      }
      float s = 0.0f;
      for (int r = k; r < 3; r++) s += v[r] * b[r];
      s = 2.0f * s / vtv;
      for (int r = k; r < 3; r++) b[r] -= s * v[r];
    }
    A[k][k] = alpha;
    for (int r = k + 1; r < 3; r++) A[r][k] = 0.0f;
  }
  float y[3];
  for (int k = 2; k >= 0; k--) {
    float s = b[k];
    for (int c = k + 1; c < 3; c++) s -= A[k][c] * y[c];
    y[k] = s / A[k][k];
  }
  for (int k = 0; k < 3; k++) x[perm[k]] = y[k];
}

__global__ __launch_bounds__(256) void k_vo_match(const int* __restrict__ prev_uv, const int* __restrict__ curr_uv, VoMatchCounts nm,
                                                  const vloam_calib* __restrict__ c, DepthMapDev Mprev, int remove_outlier, FactorTable F,
                                                  double* __restrict__ dbg, int* counters, const LOState* __restrict__ lo, double* x_init,
                                                  int reset_to_identity, size_t ss) {
  VL_SESSION(ss); RB(prev_uv); RB(curr_uv); RB(c); Mprev.rebase(so_); F.rebase(so_); RB(dbg); RB(counters); RB(lo); RB(x_init);
  const int n_match = nm.n[blockIdx.z];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j == 0 && x_init) {  // solveNlsAll's initial guess (VO:258-281); the solve behind this launch reads it
    double a[3] = {0, 0, 0}, t[3] = {0, 0, 0};
    if (!reset_to_identity && lo) {
      double q[4];
      tf_get_rotation(lo->tf.cam0_curr_LOT_cam0_prev, q);
      const double w = q[3] < -1.0 ? -1.0 : (q[3] > 1.0 ? 1.0 : q[3]);
      const double angle = 2.0 * acos(w);                          // Quaternion::getAngle
      const double s_squared = 1.0 - q[3] * q[3];                  // Quaternion::getAxis
      double ax = 1.0, ay = 0.0, az = 0.0;
      if (!(s_squared < 10.0 * 2.220446049250313e-16)) { const double s = sqrt(s_squared); ax = q[0] / s; ay = q[1] / s; az = q[2] / s; }
      a[0] = ax * angle; a[1] = ay * angle; a[2] = az * angle;
      for (int k = 0; k < 3; k++) t[k] = lo->tf.cam0_curr_LOT_cam0_prev.o[k];
    }
    for (int k = 0; k < 3; k++) { x_init[k] = a[k]; x_init[3 + k] = t[k]; }
  }
  if (j >= F.cap) return;
  int type = 0;
  double obs[5] = {0, 0, 0, 0, 0};
  float depth0 = 0.f;
  if (j < n_match) {
    const int px = prev_uv[2 * j], py = prev_uv[2 * j + 1], cx = curr_uv[2 * j], cy = curr_uv[2 * j + 1];
    const long long d2 = (long long)(px - cx) * (px - cx) + (long long)(py - cy) * (py - cy);
    // px == INT_MIN: an entry of the image front-end without a tracked corner (optical_flow_status != 1: `continue`, VO:308)
    if (px != INT_MIN && !(remove_outlier > 0 && d2 > (long long)remove_outlier * remove_outlier)) {  // VO:309-314
      float K[9];
      for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++) K[r * 3 + q] = c->P_rect0[r * 4 + q];
      depth0 = query_depth(Mprev, (float)px, (float)py);  // VO:316 (depth1 is computed but unused in the reference)
      float p0[3], p1[3], r0[3], r1[3];
      if (depth0 > 0) {  // VO:345-368
        p0[0] = px * depth0; p0[1] = py * depth0; p0[2] = depth0;
        p1[0] = (float)cx; p1[1] = (float)cy; p1[2] = 1.0f;
        solve3x3_colpiv_qr_f32(K, p0, r0);
        solve3x3_colpiv_qr_f32(K, p1, r1);
        type = 4;
        obs[0] = r0[0]; obs[1] = r0[1]; obs[2] = r0[2];
        obs[3] = (double)r1[0] / (double)r1[2]; obs[4] = (double)r1[1] / (double)r1[2];
      } else {           // VO:393-415
        p0[0] = (float)px; p0[1] = (float)py; p0[2] = 1.0f;
        p1[0] = (float)cx; p1[1] = (float)cy; p1[2] = 1.0f;
        solve3x3_colpiv_qr_f32(K, p0, r0);
        solve3x3_colpiv_qr_f32(K, p1, r1);
        type = 5;
        obs[0] = (double)r0[0] / (double)r0[2]; obs[1] = (double)r0[1] / (double)r0[2];
        obs[2] = (double)r1[0] / (double)r1[2]; obs[3] = (double)r1[1] / (double)r1[2];
      }
    }
  }
  const int cap = F.cap;
  F.type[j] = type;
  {  // one atomic per wavefront and counter (a wavefront is exactly one 64-slot row of the factor table)
    const unsigned long long m32 = __ballot(type == 4), m22 = __ballot(type == 5);
    if ((threadIdx.x & 63) == 0) {
      if (m32) atomicAdd(&counters[0], __popcll(m32));
      if (m22) atomicAdd(&counters[1], __popcll(m22));
      if (m32 | m22) atomicAdd(&F.rowcnt[j >> 6], __popcll(m32 | m22));
    }
  }
  if (type == 4) {
    F.p[j] = obs[0]; F.p[cap + j] = obs[1]; F.p[2 * cap + j] = obs[2];
    F.A[j] = obs[3]; F.A[cap + j] = obs[4]; F.A[2 * cap + j] = 0;
  } else if (type == 5) {
    F.p[j] = obs[0]; F.p[cap + j] = obs[1]; F.p[2 * cap + j] = 0;
    F.A[j] = obs[2]; F.A[cap + j] = obs[3]; F.A[2 * cap + j] = 0;
  }
  if (j < n_match) {
    dbg[7 * j] = type == 4 ? 32 : (type == 5 ? 22 : 0);
    dbg[7 * j + 1] = depth0;
    for (int k = 0; k < 5; k++) dbg[7 * j + 2 + k] = obs[k];
  }
}

// ---------------------------------------------------------------------------------------------- host side
// carve the VO buffers out of the session arena (dry run to measure, then for real); everything starts zeroed with the arena
vloam_status vo_layout(VOContext* v, const vloam_config& cfg, Arena& A) {
  bool ok = A.take(&v->d_calib, 1);
  for (int k = 0; k < VOContext::kSets && ok; k++)
    ok = A.take(&v->d_prev_set[k], 2 * kVoMaxMatches) && A.take(&v->d_curr_set[k], 2 * kVoMaxMatches) &&
         A.take(&v->maps[k].bx, kBuckets) && A.take(&v->maps[k].by, kBuckets) &&
         A.take(&v->maps[k].bd, kBuckets) && A.take(&v->maps[k].bc, kBuckets);
  ok = ok && A.take(&v->uvd, (size_t)cfg.max_points) && A.take(&v->bcount, kBuckets + 1) &&
       A.take(&v->bfill, kBuckets) && A.take(&v->seg, (size_t)cfg.max_points) &&
       A.take(&v->d_prev, 2 * kVoMaxMatches) && A.take(&v->d_curr, 2 * kVoMaxMatches);
  v->F.cap = kVoMaxMatches;
  ok = ok && A.take(&v->F.type, kVoMaxMatches) && A.take(&v->F.p, 3 * kVoMaxMatches) &&
       A.take(&v->F.A, 3 * kVoMaxMatches) && A.take(&v->F.B, 3 * kVoMaxMatches) &&
       A.take(&v->F.resid, 3 * kVoMaxMatches) && A.take(&v->F.ctype, kVoMaxMatches) &&
       A.take(&v->F.cslot, kVoMaxMatches) && A.take(&v->F.cpack, 11 * kVoMaxMatches) &&
       A.take(&v->F.rowcnt, kVoMaxMatches / 64 + 1);
  v->F.gsync = nullptr;
  v->F.dg = nullptr;   // (the VO factors are evaluated from their raw form)
  v->F.host_degraded = nullptr;
  v->F.err = nullptr;
  ok = ok && A.take(&v->rec, 1) && A.take(&v->x, 8) && A.take(&v->match_dbg, 7 * kVoMaxMatches) && A.take(&v->counters, 4);
  v->max_points = cfg.max_points;
  return ok ? VLOAM_OK : VLOAM_ERR_HIP;
}

vloam_status vo_set_calib(VOContext* v, hipStream_t st, const vloam_calib* c) {
  for (int b = 0; b < v->se.B; b++)   // one camera for all sessions of a batched handle
    if (hipMemcpyAsync((char*)v->d_calib + (size_t)b * v->se.ss, c, sizeof(*c), hipMemcpyHostToDevice, st) != hipSuccess) return VLOAM_ERR_HIP;
  if (hipStreamSynchronize(st) != hipSuccess) return VLOAM_ERR_HIP;
  v->have_calib = true;
  return VLOAM_OK;
}

// zero `bytes` at the same offset of every session's arena (one strided fill instead of B small ones)
static hipError_t memset_sessions(void* p, size_t bytes, Sess se, hipStream_t st) {
  return se.B > 1 ? hipMemset2DAsync(p, se.ss, 0, bytes, (size_t)se.B, st) : hipMemsetAsync(p, 0, bytes, st);
}

static vloam_status vo_depth_launch(VOContext* v, hipStream_t st, const BatchIn& bi, int set, ProfHook* ph) {
  const Sess se = v->se;
  const unsigned Z = (unsigned)se.B;
  if (memset_sessions(v->bcount, sizeof(int) * (kBuckets + 1), se, st) != hipSuccess) return VLOAM_ERR_HIP;
  VLOAM_LAUNCH(ph, kKVoProject, st, k_vo_project, dim3(256, 1, Z), dim3(256), 0, st, bi, v->d_calib, v->uvd, v->bcount, se.ss);
  VL_RAW_LAUNCH(k_vo_scan, dim3(1, 1, Z), dim3(1024), 0, st, v->bcount, v->bfill, se.ss);
  VL_RAW_LAUNCH(k_vo_scatter, dim3(256, 1, Z), dim3(256), 0, st, v->uvd, bi, v->bcount, v->bfill, v->seg, se.ss);
  VLOAM_LAUNCH(ph, kKVoFold, st, k_vo_fold, dim3((kBuckets + 255) / 256, 1, Z), dim3(256), 0, st, v->uvd, v->bcount, v->seg, v->maps[set], se.ss);
  return hipGetLastError() == hipSuccess ? VLOAM_OK : VLOAM_ERR_HIP;
}

static BatchIn vo_one(const float4* d_in, int n) {
  BatchIn bi;
  memset(&bi, 0, sizeof(bi));
  bi.in[0] = d_in; bi.n[0] = n;
  return bi;
}

vloam_status vo_process_point_cloud(VOContext* v, hipStream_t st, const float4* d_in, int n) {
  if (!v->have_calib) return VLOAM_ERR_ORDER;
  ++v->count;                // VisualOdometry::reset(), VO:86-90
  v->i = v->count % VOContext::kSets;
  return vo_depth_launch(v, st, vo_one(d_in, n), v->i, nullptr);
}

// coupled frame loop: the depth map of frame `frame` goes to maps[frame % kSets]; the frame's pixel matches (per session: host arrays
// prev_uv[b] / curr_uv[b] of n_match[b] pairs) are staged into the same set
vloam_status vo_depth_enqueue(VOContext* v, hipStream_t st, const BatchIn& bi, int frame, const int* const* prev_uv, const int* const* curr_uv,
                              const int* n_match, ProfHook* ph) {
  if (!v->have_calib) return VLOAM_ERR_ORDER;
  const int set = frame % VOContext::kSets;
  for (int b = 0; b < v->se.B; b++) if (n_match[b] > kVoMaxMatches || n_match[b] < 0) return VLOAM_ERR_CAPACITY;
  v->count = frame;
  v->i = set;
  for (int b = 0; b < v->se.B; b++) {
    v->n_match_set[set].n[b] = n_match[b];
    if (n_match[b] > 0 && frame > 0) {
      const size_t off = (size_t)b * v->se.ss;
      if (hipMemcpyAsync((char*)v->d_prev_set[set] + off, prev_uv[b], sizeof(int) * 2 * n_match[b], hipMemcpyHostToDevice, st) != hipSuccess) return VLOAM_ERR_HIP;
      if (hipMemcpyAsync((char*)v->d_curr_set[set] + off, curr_uv[b], sizeof(int) * 2 * n_match[b], hipMemcpyHostToDevice, st) != hipSuccess) return VLOAM_ERR_HIP;
    }
  }
  return vo_depth_launch(v, st, bi, set, ph);
}

vloam_status vo_solve_enqueue(VOContext* v, const vloam_config& cfg, hipStream_t st, int frame, LOState* lo, ProfHook* ph) {
  const int set = frame % VOContext::kSets, prev = (frame + VOContext::kSets - 1) % VOContext::kSets;
  const Sess se = v->se;
  if (memset_sessions(v->counters, sizeof(int) * 2, se, st) != hipSuccess) return VLOAM_ERR_HIP;
  VLOAM_LAUNCH(ph, kKVoMatch, st, k_vo_match, dim3(kVoMaxMatches / 256, 1, (unsigned)se.B), dim3(256), 0, st, v->d_prev_set[set], v->d_curr_set[set], v->n_match_set[set],
               v->d_calib, v->maps[prev], cfg.remove_VO_outlier, v->F, v->match_dbg, v->counters, lo, v->x, cfg.reset_VO_to_identity, se.ss);
  lm_launch(st, se, v->F, 0, v->x, v->rec, 100, 0.1, false, nullptr, ph);
  return hipGetLastError() == hipSuccess ? VLOAM_OK : VLOAM_ERR_HIP;
}

// the stage-wise VO entry points drive one sequence (session 0 of the handle)
vloam_status vo_solve(VOContext* v, const vloam_config& cfg, hipStream_t st, const int* prev_uv, const int* curr_uv, int n_match,
                      double aa[3], double t[3], int counters[2]) {
  if (!v->have_calib || v->count < 1) return VLOAM_ERR_ORDER;  // needs the previous frame's depth map
  if (n_match > kVoMaxMatches) return VLOAM_ERR_CAPACITY;
  double x[6] = {0, 0, 0, 0, 0, 0};
  if (!cfg.reset_VO_to_identity) { for (int k = 0; k < 3; k++) { x[k] = aa[k]; x[3 + k] = t[k]; } }  // VO:260-281
  if (hipMemcpyAsync(v->d_prev, prev_uv, sizeof(int) * 2 * n_match, hipMemcpyHostToDevice, st) != hipSuccess) return VLOAM_ERR_HIP;
  if (hipMemcpyAsync(v->d_curr, curr_uv, sizeof(int) * 2 * n_match, hipMemcpyHostToDevice, st) != hipSuccess) return VLOAM_ERR_HIP;
  if (hipMemcpyAsync(v->x, x, sizeof(x), hipMemcpyHostToDevice, st) != hipSuccess) return VLOAM_ERR_HIP;
  if (hipMemsetAsync(v->counters, 0, sizeof(int) * 2, st) != hipSuccess) return VLOAM_ERR_HIP;
  VoMatchCounts nm;
  memset(&nm, 0, sizeof(nm));
  nm.n[0] = n_match;
  VL_RAW_LAUNCH(k_vo_match, dim3(kVoMaxMatches / 256), dim3(256), 0, st, v->d_prev, v->d_curr, nm, v->d_calib,
                     v->maps[(v->i + VOContext::kSets - 1) % VOContext::kSets], cfg.remove_VO_outlier, v->F, v->match_dbg, v->counters,
                     (const LOState*)nullptr, (double*)nullptr, 0, (size_t)0);
  lm_launch(st, Sess(), v->F, 0, v->x, v->rec, 100, 0.1, false, nullptr);
  if (hipMemcpyAsync(x, v->x, sizeof(x), hipMemcpyDeviceToHost, st) != hipSuccess) return VLOAM_ERR_HIP;
  int cnt[2];
  if (hipMemcpyAsync(cnt, v->counters, sizeof(cnt), hipMemcpyDeviceToHost, st) != hipSuccess) return VLOAM_ERR_HIP;
  if (hipStreamSynchronize(st) != hipSuccess) return VLOAM_ERR_HIP;
  for (int k = 0; k < 3; k++) { aa[k] = x[k]; t[k] = x[3 + k]; }
  if (counters) { counters[0] = cnt[0]; counters[1] = cnt[1]; }
  return VLOAM_OK;
}

static vloam_status copy_dev(const void* src, size_t bytes, void* buf, long long cap, long long* n) {
  if (n) *n = (long long)bytes;
  const size_t c = bytes < (size_t)cap ? bytes : (size_t)cap;
  if (buf && c && hipMemcpy(buf, src, c, hipMemcpyDeviceToHost) != hipSuccess) return VLOAM_ERR_HIP;
  return VLOAM_OK;
}

// item 0..3: bucket_x / bucket_y / bucket_depth / bucket_count of the CURRENT map; 4..7 the same of the PREVIOUS map;
// 8: per-match rows f64[n][7]; 9: LM record; 10: projected points f32[n][4] (u, v, depth, bucket id bits)
vloam_status vo_debug_get(VOContext* v, int item, void* buf, long long cap, long long* n) {
  if (item >= 0 && item < 8) {
    const DepthMapDev& M = v->maps[item < 4 ? v->i : (v->i + VOContext::kSets - 1) % VOContext::kSets];
    switch (item & 3) {
      case 0: return copy_dev(M.bx, sizeof(float) * kBuckets, buf, cap, n);
      case 1: return copy_dev(M.by, sizeof(float) * kBuckets, buf, cap, n);
      case 2: return copy_dev(M.bd, sizeof(float) * kBuckets, buf, cap, n);
      case 3: return copy_dev(M.bc, sizeof(int) * kBuckets, buf, cap, n);
    }
  }
  if (item == 8) return copy_dev(v->match_dbg, sizeof(double) * 7 * kVoMaxMatches, buf, cap, n);
  if (item == 9) return copy_dev(v->rec, sizeof(LMRecord), buf, cap, n);
  if (item == 10) return copy_dev(v->uvd, sizeof(float4) * (size_t)v->max_points, buf, cap, n);
  return VLOAM_ERR_INVALID;
}

}  // namespace vloam
