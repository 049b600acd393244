// laserOdometry data association on gfx950: for every sharp / flat feature of the current sweep,
// TransformToStart, exact 1-NN in the previous sweep's lessSharp / lessFlat cloud, the adjacent-ring
// walks for the 2nd (and 3rd) point, and emission of the Ceres residual block as a FactorTable slot.
// Restates LaserOdometry::solveLO, /root/reference/src/lidar_odometry_mapping/src/laser_odometry.cpp:207-444
// ("LO:<line>").  One wavefront per feature: 64 lanes sweep the candidate array, keep
// (f32 distance bits << 32 | visit order) keys and reduce them with wavefront shuffles, which
// reproduces the reference's first-strictly-smaller-wins scans and "lowest index wins" kNN ties.
#include <hip/hip_runtime.h>
#include <math.h>
#include "lo_kernels.h"

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

constexpr unsigned kBack = 0x40000000u;

__global__ __launch_bounds__(256) void k_lo_assoc(const float4* __restrict__ sharp, const float4* __restrict__ flat,
                                                  const FrameScalars* __restrict__ Sc, const float4* __restrict__ CL,
                                                  const float4* __restrict__ SL, const FrameScalars* __restrict__ Sp,
                                                  const LOState* __restrict__ lo, FactorTable F, int* __restrict__ corr) {
  const int lane = threadIdx.x & 63;
  const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slot >= kMaxLoFactors) return;
  const bool is_corner = slot < kMaxSharp;
  const int i = is_corner ? slot : slot - kMaxSharp;
  const int nfeat = is_corner ? Sc->n_sharp : Sc->n_flat;
  int type = 0, ia = -1, ib = -1, ic = -1;
  if (i < nfeat) {
    const float4 pf = is_corner ? sharp[i] : flat[i];
    const float3 sel = transform_to_start(pf, lo->para_q, lo->para_t);  // LO:268 / LO:355
    const float4* cand = is_corner ? CL : SL;
    const int n = is_corner ? Sp->n_less_sharp : Sp->n_less_flat;
    // ---- exact nearest neighbour (pcl::KdTreeFLANN::nearestKSearch k = 1, flann::L2_Simple<float>), LO:269 / LO:356
    u64 best = ~0ull;
    for (int j = lane; j < n; j += 64) {
      const float4 c = cand[j];
      const float d0 = sel.x - c.x, d1 = sel.y - c.y, d2 = sel.z - c.z;
      const float d = d0 * d0 + d1 * d1 + d2 * d2;
      const u64 key = ((u64)__float_as_uint(d) << 32) | (unsigned)j;
      best = key < best ? key : best;
    }
    best = wave_min_u64(best);
    const float dmin = __uint_as_float((unsigned)(best >> 32));
    if (best != ~0ull && dmin < 25.0f) {  // DISTANCE_SQ_THRESHOLD, LO:272 / LO:359
      const int idx = (int)(best & 0xffffffffu);
      const int ringA = (int)cand[idx].w;  // closestPointScanID
      u64 b2 = ~0ull, b3 = ~0ull;
      // ---- increasing scan line, LO:279-300 / LO:368-391
      bool stopped = false;
      for (int base = idx + 1; base < n && !stopped; base += 64) {
        const int j = base + lane;
        const bool in = j < n;
        const float4 c = in ? cand[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        const int rj = (int)c.w;
        const bool stop = in && ((double)rj > (double)ringA + 2.5);  // NEARBY_SCAN
        const u64 sm = __ballot(stop);
        const int first_stop = sm ? __ffsll((long long)sm) - 1 : 64;
        if (in && lane < first_stop) {
          const float d = sqdist(c, sel);
          const u64 key = ((u64)__float_as_uint(d) << 32) | (unsigned)(j - idx);
          if (d < 25.0f) {
            if (is_corner) { if (!(rj <= ringA)) b2 = key < b2 ? key : b2; }
            else if (rj <= ringA) b2 = key < b2 ? key : b2;
            else b3 = key < b3 ? key : b3;
          }
        }
        stopped = sm != 0;
      }
      // ---- decreasing scan line, LO:303-324 / LO:394-417
      stopped = false;
      for (int base = idx - 1; base >= 0 && !stopped; base -= 64) {
        const int j = base - lane;
        const bool in = j >= 0;
        const float4 c = in ? cand[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        const int rj = (int)c.w;
        const bool stop = in && ((double)rj < (double)ringA - 2.5);
        const u64 sm = __ballot(stop);
        const int first_stop = sm ? __ffsll((long long)sm) - 1 : 64;
        if (in && lane < first_stop) {
          const float d = sqdist(c, sel);
          const u64 key = ((u64)__float_as_uint(d) << 32) | (kBack + (unsigned)(idx - j));
          if (d < 25.0f) {
            if (is_corner) { if (!(rj >= ringA)) b2 = key < b2 ? key : b2; }
            else if (rj >= ringA) b2 = key < b2 ? key : b2;
            else b3 = key < b3 ? key : b3;
          }
        }
        stopped = sm != 0;
      }
      b2 = wave_min_u64(b2);
      b3 = wave_min_u64(b3);
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
        }
      }
    }
  }
  if (lane == 0) {
    F.type[slot] = type;
    if (type) atomicAdd(&F.rowcnt[slot >> 6], 1);
    corr[slot * 4 + 0] = type ? i : -1;
    corr[slot * 4 + 1] = ia; corr[slot * 4 + 2] = ib; corr[slot * 4 + 3] = ic;
  }
}

// LO:223-236 — combined mode overwrites the warm start with the VO prior at the top of each outer round
__global__ void k_lo_set_prior(LOState* lo) {
  const int t = threadIdx.x;
  if (t < 4) lo->para_q[t] = lo->prior_q[t];
  else if (t < 7) lo->para_t[t - 4] = lo->prior_t[t - 4];
}

// LO:477-478 pose integration (+ trajectory log row); q_w_curr is not renormalised, as in the reference.
__global__ void k_lo_finish(LOState* lo, double* traj_row14, int integrate) {
  if (threadIdx.x != 0) return;
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
  if (traj_row14) {
    for (int k = 0; k < 4; k++) traj_row14[k] = lo->q_w_curr[k];
    for (int k = 0; k < 3; k++) traj_row14[4 + k] = lo->t_w_curr[k];
    // mapping overwrites [7..13] when it runs; until then the map pose equals the odometry pose
    for (int k = 0; k < 7; k++) traj_row14[7 + k] = traj_row14[k];
  }
}

void lo_assoc_launch(hipStream_t st, const float4* sharp, const float4* flat, const FrameScalars* Sc, const float4* CL, const float4* SL,
                     const FrameScalars* Sp, const LOState* lo, const FactorTable& F, int* corr, ProfHook* ph) {
  VLOAM_LAUNCH(ph, kKLoAssoc, st, k_lo_assoc, dim3((kMaxLoFactors + 3) / 4), dim3(256), 0, st, sharp, flat, Sc, CL, SL, Sp, lo, F, corr);
}
void lo_set_prior_launch(hipStream_t st, LOState* lo) { hipLaunchKernelGGL(k_lo_set_prior, dim3(1), dim3(64), 0, st, lo); }
void lo_finish_launch(hipStream_t st, LOState* lo, double* traj_row14, bool integrate, ProfHook* ph) {
  VLOAM_LAUNCH(ph, kKLoFinish, st, k_lo_finish, dim3(1), dim3(64), 0, st, lo, traj_row14, integrate ? 1 : 0);
}

}  // namespace vloam
