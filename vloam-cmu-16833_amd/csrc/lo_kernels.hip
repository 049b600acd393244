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

__device__ __forceinline__ unsigned grid_hash(int ix, int iy, int iz) {
  unsigned h = (unsigned)ix * 73856093u ^ (unsigned)iy * 19349663u ^ (unsigned)iz * 83492791u;
  h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
  return h;
}
__device__ __forceinline__ int coarse_cell(float v) { return (int)floorf(v / 5.0f); }

__global__ __launch_bounds__(256) void k_lo_grid_count(const float4* __restrict__ less_sharp, const float4* __restrict__ less_flat,
                                                       const FrameScalars* __restrict__ S, LoGrid G) {
  const int kind = blockIdx.y;
  const float4* pts = kind ? less_flat : less_sharp;
  const int n = kind ? S->n_less_flat : S->n_less_sharp;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float4 p = pts[i];
    atomicAdd(&G.cnt[kind][grid_hash((int)floorf(p.x), (int)floorf(p.y), (int)floorf(p.z)) & (unsigned)G.mask[kind]], 1);
    atomicAdd(&G.cnt[kind + 2][grid_hash(coarse_cell(p.x), coarse_cell(p.y), coarse_cell(p.z)) & (unsigned)G.mask[kind + 2]], 1);
  }
}
__global__ __launch_bounds__(1024) void k_lo_grid_scan(LoGrid G) {
  __shared__ int buf[kGridMaxBuckets];  // the whole counter array goes through LDS: coalesced in, coalesced out
  __shared__ int sums[1024];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int nb = G.mask[g] + 1;
  int* cnt = G.cnt[g];
  for (int k = tid; k < nb; k += 1024) { buf[k] = cnt[k]; cnt[k] = 0; G.fill[g][k] = 0; }
  __syncthreads();
  const int per = nb / 1024;
  const int lo = tid * per;
  int s = 0;
  for (int k = 0; k < per; k++) s += buf[lo + k];
  sums[tid] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) { const int v = tid >= d ? sums[tid - d] : 0; __syncthreads(); sums[tid] += v; __syncthreads(); }
  int run = tid ? sums[tid - 1] : 0;
  for (int k = 0; k < per; k++) { const int c = buf[lo + k]; buf[lo + k] = run; run += c; }
  __syncthreads();
  for (int k = tid; k < nb; k += 1024) G.start[g][k] = buf[k];
  if (tid == 1023) G.start[g][nb] = sums[1023];
}
__global__ __launch_bounds__(256) void k_lo_grid_scatter(const float4* __restrict__ less_sharp, const float4* __restrict__ less_flat,
                                                         const FrameScalars* __restrict__ S, LoGrid G) {
  const int kind = blockIdx.y;
  const float4* pts = kind ? less_flat : less_sharp;
  const int n = kind ? S->n_less_flat : S->n_less_sharp;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float4 p = pts[i];
    const unsigned b = grid_hash((int)floorf(p.x), (int)floorf(p.y), (int)floorf(p.z)) & (unsigned)G.mask[kind];
    G.items[kind][G.start[kind][b] + atomicAdd(&G.fill[kind][b], 1)] = i;
    const unsigned c = grid_hash(coarse_cell(p.x), coarse_cell(p.y), coarse_cell(p.z)) & (unsigned)G.mask[kind + 2];
    G.items[kind + 2][G.start[kind + 2][c] + atomicAdd(&G.fill[kind + 2][c], 1)] = i;
  }
}

constexpr unsigned kBack = 0x40000000u;

__global__ __launch_bounds__(256) void k_lo_assoc(const float4* __restrict__ sharp, const float4* __restrict__ flat,
                                                  const FrameScalars* __restrict__ Sc, const float4* __restrict__ CL,
                                                  const float4* __restrict__ SL, const FrameScalars* __restrict__ Sp, LoGrid G,
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
    // ---- exact nearest neighbour (pcl::KdTreeFLANN::nearestKSearch k = 1, flann::L2_Simple<float>), LO:269 / LO:356.
    // Expanding search over the 1 m hash grid: after every cell within Chebyshev radius R of the query's cell has been
    // scanned, any unseen point is farther than R metres; the search stops as soon as the best distance is inside that
    // bound (or at R = 5, beyond which DISTANCE_SQ_THRESHOLD = 25 rejects the match anyway).  Key = (f32 d2 bits, index)
    // so ties resolve to the lowest index irrespective of visiting order.
    const int kind = is_corner ? 0 : 1;
    const int* gstart = G.start[kind];
    const int* gitems = G.items[kind];
    const int gmask = G.mask[kind];
    const int cx = (int)floorf(sel.x), cy = (int)floorf(sel.y), cz = (int)floorf(sel.z);
    u64 best = ~0ull;
    bool exact = false;
    for (int R = 1; R <= 2 && n > 0; R++) {
      const int w = 2 * R + 1, ncell = w * w * w;
      u64 loc = ~0ull;
      // four cells per lane and trip: their bucket bounds are fetched together, then only non-empty buckets are walked
      for (int cc0 = lane; cc0 < ncell; cc0 += 4 * 64) {
        int bs[4], be[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int cc = cc0 + u * 64;
          bs[u] = 0; be[u] = 0;
          if (cc < ncell) {
            const int ox = cc % w - R, oy = (cc / w) % w - R, oz = cc / (w * w) - R;
            if (!(R > 1 && abs(ox) < R && abs(oy) < R && abs(oz) < R)) {  // interior was scanned at the previous radius
              const unsigned b = grid_hash(cx + ox, cy + oy, cz + oz) & (unsigned)gmask;
              bs[u] = gstart[b]; be[u] = gstart[b + 1];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          for (int t = bs[u]; t < be[u]; t++) {
            const int j = gitems[t];
            const float4 c = cand[j];
            const float d0 = sel.x - c.x, d1 = sel.y - c.y, d2 = sel.z - c.z;
            const float d = d0 * d0 + d1 * d1 + d2 * d2;
            const u64 key = ((u64)__float_as_uint(d) << 32) | (unsigned)j;
            loc = key < loc ? key : loc;
          }
        }
      }
      loc = wave_min_u64(loc);
      best = loc < best ? loc : best;
      const float bd = __uint_as_float((unsigned)(best >> 32));
      const float bound = (float)(R * R) * 0.999999f;
      if (best != ~0ull && bd <= bound) { exact = true; break; }
    }
    if (!exact && n > 0) {
      // no neighbour within 2 m: sweep the 27 cells of the 5 m grid around the query — together they contain every point
      // within 5 m, and anything farther is rejected by DISTANCE_SQ_THRESHOLD below.  Lanes stride over each cell's points.
      const int* cstart = G.start[kind + 2];
      const int* citems = G.items[kind + 2];
      const int cmask = G.mask[kind + 2];
      const int ccx = coarse_cell(sel.x), ccy = coarse_cell(sel.y), ccz = coarse_cell(sel.z);
      u64 loc = ~0ull;
      for (int cc = 0; cc < 27; cc++) {
        const unsigned b = grid_hash(ccx + cc % 3 - 1, ccy + (cc / 3) % 3 - 1, ccz + cc / 9 - 1) & (unsigned)cmask;
        const int e = cstart[b + 1];
        for (int t0 = cstart[b] + lane; t0 < e; t0 += 4 * 64) {
          int js[4];
          float4 cs[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { const int t = t0 + u * 64; js[u] = t < e ? citems[t] : -1; }
#pragma unroll
          for (int u = 0; u < 4; u++) cs[u] = js[u] >= 0 ? cand[js[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int u = 0; u < 4; u++) {
            if (js[u] < 0) continue;
            const float d0 = sel.x - cs[u].x, d1 = sel.y - cs[u].y, d2 = sel.z - cs[u].z;
            const float d = d0 * d0 + d1 * d1 + d2 * d2;
            const u64 key = ((u64)__float_as_uint(d) << 32) | (unsigned)js[u];
            loc = key < loc ? key : loc;
          }
        }
      }
      loc = wave_min_u64(loc);
      best = loc < best ? loc : best;
    }
    const float dmin = __uint_as_float((unsigned)(best >> 32));
    if (best != ~0ull && dmin < 25.0f) {  // DISTANCE_SQ_THRESHOLD, LO:272 / LO:359
      const int idx = (int)(best & 0xffffffffu);
      const int ringA = (int)cand[idx].w;  // closestPointScanID
      u64 b2 = ~0ull, b3 = ~0ull;
      // Both walks stream the candidate array in trips of kU x 64 points whose loads are all issued before the first is
      // examined (the stop test is applied afterwards, chunk by chunk, exactly in visiting order; points fetched beyond the
      // stop are simply ignored) — otherwise every 64-point chunk would cost a full dependent memory round trip.
      constexpr int kU = 8;
      // ---- increasing scan line, LO:279-300 / LO:368-391
      bool stopped = false;
      for (int base = idx + 1; base < n && !stopped; base += 64 * kU) {
        float4 c[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) { const int j = base + u * 64 + lane; c[u] = j < n ? cand[j] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int u = 0; u < kU; u++) {
          const int j = base + u * 64 + lane;
          const bool in = j < n && !stopped;
          const int rj = (int)c[u].w;
          const bool stop = in && ((double)rj > (double)ringA + 2.5);  // NEARBY_SCAN
          const u64 sm = __ballot(stop);
          const int first_stop = sm ? __ffsll((long long)sm) - 1 : 64;
          if (in && lane < first_stop) {
            const float d = sqdist(c[u], sel);
            const u64 key = ((u64)__float_as_uint(d) << 32) | (unsigned)(j - idx);
            if (d < 25.0f) {
              if (is_corner) { if (!(rj <= ringA)) b2 = key < b2 ? key : b2; }
              else if (rj <= ringA) b2 = key < b2 ? key : b2;
              else b3 = key < b3 ? key : b3;
            }
          }
          stopped = stopped || sm != 0;
        }
      }
      // ---- decreasing scan line, LO:303-324 / LO:394-417
      stopped = false;
      for (int base = idx - 1; base >= 0 && !stopped; base -= 64 * kU) {
        float4 c[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) { const int j = base - u * 64 - lane; c[u] = j >= 0 ? cand[j] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int u = 0; u < kU; u++) {
          const int j = base - u * 64 - lane;
          const bool in = j >= 0 && !stopped;
          const int rj = (int)c[u].w;
          const bool stop = in && ((double)rj < (double)ringA - 2.5);
          const u64 sm = __ballot(stop);
          const int first_stop = sm ? __ffsll((long long)sm) - 1 : 64;
          if (in && lane < first_stop) {
            const float d = sqdist(c[u], sel);
            const u64 key = ((u64)__float_as_uint(d) << 32) | (kBack + (unsigned)(idx - j));
            if (d < 25.0f) {
              if (is_corner) { if (!(rj >= ringA)) b2 = key < b2 ? key : b2; }
              else if (rj >= ringA) b2 = key < b2 ? key : b2;
              else b3 = key < b3 ? key : b3;
            }
          }
          stopped = stopped || sm != 0;
        }
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
                     const FrameScalars* Sp, const LoGrid& G, const LOState* lo, const FactorTable& F, int* corr, ProfHook* ph) {
  VLOAM_LAUNCH(ph, kKLoAssoc, st, k_lo_assoc, dim3((kMaxLoFactors + 3) / 4), dim3(256), 0, st, sharp, flat, Sc, CL, SL, Sp, G, lo, F, corr);
}
void lo_grid_build_launch(hipStream_t st, const float4* less_sharp, const float4* less_flat, const FrameScalars* S, const LoGrid& G, ProfHook* ph) {
  (void)ph;
  hipLaunchKernelGGL(k_lo_grid_count, dim3(64, 2), dim3(256), 0, st, less_sharp, less_flat, S, G);
  hipLaunchKernelGGL(k_lo_grid_scan, dim3(4), dim3(1024), 0, st, G);
  hipLaunchKernelGGL(k_lo_grid_scatter, dim3(64, 2), dim3(256), 0, st, less_sharp, less_flat, S, G);
}
void lo_set_prior_launch(hipStream_t st, LOState* lo) { hipLaunchKernelGGL(k_lo_set_prior, dim3(1), dim3(64), 0, st, lo); }
void lo_finish_launch(hipStream_t st, LOState* lo, double* traj_row14, bool integrate, ProfHook* ph) {
  VLOAM_LAUNCH(ph, kKLoFinish, st, k_lo_finish, dim3(1), dim3(64), 0, st, lo, traj_row14, integrate ? 1 : 0);
}

}  // namespace vloam
