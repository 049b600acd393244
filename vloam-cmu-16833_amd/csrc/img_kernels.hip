// Image front-end of the visual odometry (optical-flow configuration) as hand-written HIP for gfx950.  See img_kernels.h for the
// reference call sites; the arithmetic follows oracle/orc_img.h's statement of cv::goodFeaturesToTrack / cv::calcOpticalFlowPyrLK
// (integer-exact structure tensor and Lucas-Kanade sums, everything else in OpenCV's own f32 / fixed-point order).
//
//   k_img_clahe_*     1 WG / tile, 1 thread / pixel   optional cv::CLAHE(2.0, 8 x 8) in front of everything (cfg.CLAHE)
//   k_img_sobel       1 thread / pixel     Sobel 3x3 (ints) of the 8-bit image + contiguous copy = pyramid level 0
//   k_img_eig         32 x 8 tiles         5 x 5 box sums of the gradient products through LDS, min eigenvalue (f64 -> f32), global max
//   k_img_localmax    1 thread / pixel     quality threshold + 3 x 3 non-maximum suppression -> candidate list + pixel -> candidate map
//   k_img_neighbours  1 wavefront / cand.  the stronger candidates closer than minDistance (what the greedy pass can be blocked by)
//   k_img_select      1 workgroup          the greedy minDistance pass as a fixed point over "blocked by an accepted stronger
//                                          neighbour" (order-free: the result equals the sorted sequential pass), then the
//                                          (strength, address)-sorted cut at maxCorners through an LDS bitonic network
//   k_img_pyrdown     1 thread / pixel     cv::pyrDown
//   k_img_scharr      1 thread / pixel     calcSharrDeriv of all pyramid levels in one launch
//   k_img_bf_knn      1 wavefront / query  brute-force Hamming 2-NN over the other image's descriptors (image_util.cpp:221-296)
//   k_img_lk          1 wavefront / corner pyramidal Lucas-Kanade, all levels in one launch; the 15 x 15 window lives in registers
//                                          (4 pixels per lane), the search patch of the next image in LDS, the 2 x 2 system in exact
//                                          integer sums (DPP row reductions)
#include <hip/hip_runtime.h>
#include <limits.h>
#include <float.h>
#include <math.h>
#include <vector>
#include "img_kernels.h"

namespace vloam {

typedef unsigned long long u64;

__device__ __forceinline__ int reflect101(int i, int n) {  // cv::borderInterpolate(i, n, BORDER_REFLECT_101)
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}

// the same for indices at most one image size outside (|overshoot| < n): two folds, no loop — the Lucas-Kanade window hangs over a
// border by at most winSize + 1 pixels and every pyramid level is larger than the window
__device__ __forceinline__ int reflect101_near(int i, int n) {
  i = i < 0 ? -i : i;
  i = i >= n ? 2 * n - 2 - i : i;
  return i < 0 ? -i : i;
}

__global__ __launch_bounds__(256) void k_img_sobel(const unsigned char* __restrict__ img, int w, int h, int stride, short2* __restrict__ out,
                                                   unsigned char* __restrict__ level0, unsigned* maxbits, int* n_cand) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx == 0) { *maxbits = 0u; *n_cand = 0; }
  if (idx >= w * h) return;
  const int y = idx / w, x = idx - y * w;
  const unsigned char* r0 = img + (size_t)reflect101(y - 1, h) * stride;
  const unsigned char* r1 = img + (size_t)y * stride;
  const unsigned char* r2 = img + (size_t)reflect101(y + 1, h) * stride;
  const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
  const int dx = ((int)r0[xp] - (int)r0[xm]) + 2 * ((int)r1[xp] - (int)r1[xm]) + ((int)r2[xp] - (int)r2[xm]);
  const int dy = ((int)r2[xm] - (int)r0[xm]) + 2 * ((int)r2[x] - (int)r0[x]) + ((int)r2[xp] - (int)r0[xp]);
  out[idx] = make_short2((short)dx, (short)dy);
  level0[idx] = r1[x];
}

// cv::CLAHE (imgproc/src/clahe.cpp), 8-bit: one workgroup per tile builds the tile's histogram in LDS (the image padded REFLECT_101 to a
// multiple of the tile grid), clips it, spreads the excess (uniform batch + residual at a fixed stride) and writes the cumulative LUT.
__global__ __launch_bounds__(256) void k_img_clahe_lut(const unsigned char* __restrict__ img, int w, int h, int stride, int tiles, int tw, int th, int clip,
                                                       float lut_scale, unsigned char* __restrict__ lut) {
  __shared__ int hist[256], scan[256], s_clipped;
  const int tid = threadIdx.x, tx = blockIdx.x % tiles, ty = blockIdx.x / tiles;
  hist[tid] = 0;
  if (tid == 0) s_clipped = 0;
  __syncthreads();
  for (int e = tid; e < tw * th; e += 256) {
    const int y = ty * th + e / tw, x = tx * tw + e % tw;
    atomicAdd(&hist[img[(size_t)reflect101(y, h) * stride + reflect101(x, w)]], 1);   // integer counts: order-free
  }
  __syncthreads();
  int v = hist[tid];
  if (clip > 0) {
    if (v > clip) { atomicAdd(&s_clipped, v - clip); v = clip; }
    __syncthreads();
    const int clipped = s_clipped, batch = clipped / 256, residual = clipped - batch * 256;
    v += batch;
    if (residual != 0) {
      const int step = max(256 / residual, 1);
      if (tid % step == 0 && tid / step < residual) v++;   // for (i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++
    }
  }
  scan[tid] = v;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {   // inclusive scan (exact integers)
    const int o = tid >= d ? scan[tid - d] : 0;
    __syncthreads();
    scan[tid] += o;
    __syncthreads();
  }
  const int r = (int)rintf((float)scan[tid] * lut_scale);   // saturate_cast<uchar>(float)
  lut[(size_t)blockIdx.x * 256 + tid] = (unsigned char)min(max(r, 0), 255);
}

__global__ __launch_bounds__(256) void k_img_clahe_apply(const unsigned char* __restrict__ img, int w, int h, int stride, int tiles, float inv_tw, float inv_th,
                                                         const unsigned char* __restrict__ lut, unsigned char* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= w * h) return;
  const int y = idx / w, x = idx - y * w;
  const float tyf = (float)y * inv_th - 0.5f, txf = (float)x * inv_tw - 0.5f;
  int ty1 = (int)floorf(tyf), tx1 = (int)floorf(txf);
  const float ya = tyf - (float)ty1, ya1 = 1.0f - ya, xa = txf - (float)tx1, xa1 = 1.0f - xa;
  const int ty2 = min(ty1 + 1, tiles - 1), tx2 = min(tx1 + 1, tiles - 1);
  ty1 = max(ty1, 0); tx1 = max(tx1, 0);
  const int v = img[(size_t)y * stride + x];
  const float l11 = lut[(size_t)(ty1 * tiles + tx1) * 256 + v], l12 = lut[(size_t)(ty1 * tiles + tx2) * 256 + v];
  const float l21 = lut[(size_t)(ty2 * tiles + tx1) * 256 + v], l22 = lut[(size_t)(ty2 * tiles + tx2) * 256 + v];
  const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
  out[idx] = (unsigned char)min(max((int)rintf(res), 0), 255);
}

constexpr int kTileW = 32, kTileH = 8, kHalo = kImgBlock / 2;
__global__ __launch_bounds__(kTileW * kTileH) void k_img_eig(const short2* __restrict__ D, int w, int h, double hs2, float* __restrict__ eig,
                                                             unsigned* maxbits) {
  __shared__ short2 tile[kTileH + 2 * kHalo][kTileW + 2 * kHalo + 1];
  __shared__ unsigned wmax[kTileW * kTileH / 64];
  const int tid = threadIdx.x, tx = tid % kTileW, ty = tid / kTileW;
  const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
  constexpr int TW = kTileW + 2 * kHalo, TH = kTileH + 2 * kHalo;
  for (int e = tid; e < TW * TH; e += kTileW * kTileH) {
    const int ly = e / TW, lx = e - ly * TW;
    // the box filter reflects the PRODUCT images (BORDER_REFLECT_101 on the filter's own input), i.e. the gradient of the reflected pixel
    const int gx = reflect101(x0 + lx - kHalo, w), gy = reflect101(y0 + ly - kHalo, h);
    tile[ly][lx] = D[(size_t)gy * w + gx];
  }
  __syncthreads();
  const int x = x0 + tx, y = y0 + ty;
  float e = 0.f;
  if (x < w && y < h) {
    int sxx = 0, sxy = 0, syy = 0;  // <= 25 * 1020^2: exact in 32 bits
#pragma unroll
    for (int j = 0; j < kImgBlock; j++)
#pragma unroll
      for (int i = 0; i < kImgBlock; i++) {
        const short2 g = tile[ty + j][tx + i];
        sxx += (int)g.x * g.x; sxy += (int)g.x * g.y; syy += (int)g.y * g.y;
      }
    const long long d = (long long)sxx - syy;
    const double root = sqrt((double)(d * d + 4ll * sxy * sxy));   // < 2^53: the radicand is exact
    e = (float)(((double)(sxx + syy) - root) * hs2);
    eig[(size_t)y * w + x] = e;
  }
  unsigned b = e > 0.f ? __float_as_uint(e) : 0u;  // positive floats order like their bit patterns
  for (int d = 32; d > 0; d >>= 1) { const unsigned o = __shfl_xor(b, d); b = o > b ? o : b; }
  if ((tid & 63) == 0) wmax[tid >> 6] = b;
  __syncthreads();
  if (tid == 0) {
    unsigned m = 0;
    for (int k = 0; k < kTileW * kTileH / 64; k++) m = wmax[k] > m ? wmax[k] : m;
    if (m) atomicMax(maxbits, m);
  }
}

__global__ __launch_bounds__(256) void k_img_localmax(const float* __restrict__ eig, int w, int h, const unsigned* __restrict__ maxbits,
                                                      double quality, int* __restrict__ cmap, int* __restrict__ clist, int* n_cand, int* err) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const bool inside = idx < w * h;
  const int y = inside ? idx / w : 0, x = inside ? idx - y * w : 0;
  bool is_cand = false;
  if (inside && x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
    const float thr = (float)((double)__uint_as_float(*maxbits) * quality);   // threshold(eig, maxVal * qualityLevel, THRESH_TOZERO)
    const float v = eig[idx];
    if (v > thr) {
      // v == dilate3x3(thresholded)  <=>  no neighbour is larger (thr >= 0: a neighbour the threshold zeroed is below v anyway)
      bool is_max = true;
#pragma unroll
      for (int j = -1; j <= 1; j++)
#pragma unroll
        for (int i = -1; i <= 1; i++) is_max = is_max && !(eig[idx + j * w + i] > v);
      is_cand = is_max;
    }
  }
  // one counter update per wavefront (thousands of single atomics on one word serialise for tens of microseconds)
  const unsigned long long m = __ballot(is_cand);
  int c = -1;
  if (m) {
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(n_cand, __popcll(m));
    base = __shfl(base, leader);
    if (is_cand) {
      c = base + __popcll(m & ((1ull << lane) - 1ull));
      if (c < kImgCandCap) clist[c] = idx;
      else { c = -1; atomicOr(err, kErrImgCandidates); }
    }
  }
  if (inside) cmap[idx] = c;
}

// one wavefront per candidate: the (2R + 1)^2 positions of its neighbourhood spread over the lanes (one round trip for the candidate
// map, one for the strengths), the stronger candidates appended through a ballot prefix
__global__ __launch_bounds__(256) void k_img_neighbours(const float* __restrict__ eig, int w, int h, const int* __restrict__ cmap,
                                                        const int* __restrict__ clist, const int* __restrict__ n_cand, float md2, int R,
                                                        int* __restrict__ nbr, unsigned char* __restrict__ nbr_cnt, int* err) {
  const int n = min(*n_cand, kImgCandCap);
  const int lane = threadIdx.x & 63;
  const int side = 2 * R + 1, npos = side * side;
  for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < n; c += gridDim.x * 4) {
    const int p = clist[c];
    const int y = p / w, x = p - y * w;
    const float v = eig[p];
    int cnt = 0;
    for (int base = 0; base < npos; base += 64) {
      const int pos = base + lane;
      int cq = -1, q = 0;
      if (pos < npos) {
        const int dy = pos / side - R, dx = pos - (pos / side) * side - R;
        const int yy = y + dy, xx = x + dx;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w && (dx != 0 || dy != 0) && (float)(dx * dx + dy * dy) < md2) {   // featureselect.cpp: dx*dx + dy*dy < minDistance^2
          q = yy * w + xx;
          cq = cmap[q];
        }
      }
      bool stronger = false;
      if (cq >= 0) {
        const float vq = eig[q];
        stronger = vq > v || (vq == v && q > p);   // sorted ahead of this candidate (greaterThanPtr: value, then the larger address)
      }
      const unsigned long long m = __ballot(stronger);
      if (stronger) {
        const int k = cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (k < kImgNbrCap) nbr[(size_t)c * kImgNbrCap + k] = cq;
      }
      cnt += __popcll(m);
    }
    if (lane == 0) {
      if (cnt > kImgNbrCap) { atomicOr(err, kErrImgNeighbours); cnt = kImgNbrCap; }
      nbr_cnt[c] = (unsigned char)cnt;
    }
  }
}

constexpr int kSelThreads = 1024;
static_assert(kImgCandCap <= 65536, "candidate indices travel as 16-bit values in k_img_select");
static_assert(kImgNbrCap % 4 == 0 && kImgNbrCap <= 255, "neighbour rows are read as int4, their lengths stored in a byte");
__global__ __launch_bounds__(kSelThreads) void k_img_select(const float* __restrict__ eig, int w, const int* __restrict__ clist,
                                                            const int* __restrict__ n_cand, const int* __restrict__ nbr,
                                                            const unsigned char* __restrict__ nbr_cnt, u64* acc, int max_corners,
                                                            float2* __restrict__ corners, int* n_corners, int* err) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* status = smem;          // [kImgCandCap]: 0 undecided, 1 accepted, 2 dropped;   later reused as u64 keys[kImgAccCap]
  __shared__ int s_changed, s_nacc;
  const int tid = threadIdx.x;
  long long tst[8]; int nst = 0, nsweep = 0;
  tst[nst++] = clock64();
  const int n = min(*n_cand, kImgCandCap);
  for (int c = tid; c < n; c += kSelThreads) status[c] = 0;
  if (tid == 0) s_nacc = 0;
  __syncthreads();
  // The sequential pass accepts a candidate iff no ACCEPTED candidate sorted ahead of it lies closer than minDistance.  A candidate is
  // therefore decided as soon as one such neighbour is accepted (dropped) or all of them are dropped (accepted): decisions never
  // change, every sweep decides at least the strongest undecided candidate, and any update order reaches the same fixed point.
  // The lists of a lane's first kFast candidates (tid, tid + 1024, ...) are fetched ONCE, in one round trip, and kept in registers as
  // 16-bit indices when they have at most kFastN entries: a sweep is then LDS reads between two barriers.  Longer lists and
  // candidates beyond kFast x 1024 are read from memory in every sweep.
  constexpr int kFast = 3, kFastN = 16;
  int f_cnt[kFast];
  unsigned f_pk[kFast][kFastN / 2];
#pragma unroll
  for (int m = 0; m < kFast; m++) {
    const int c = tid + m * kSelThreads;
    f_cnt[m] = -1;
#pragma unroll
    for (int k = 0; k < kFastN / 2; k++) f_pk[m][k] = 0u;
    if (c < n) {
      const int cnt = nbr_cnt[c];
      const int4* row = reinterpret_cast<const int4*>(nbr + (size_t)c * kImgNbrCap);   // rows of kImgNbrCap ints, 16-byte aligned
      int4 v[kFastN / 4];
#pragma unroll
      for (int k = 0; k < kFastN / 4; k++) v[k] = row[k];   // entries past cnt are stale or zero candidate indices: never looked at
      if (cnt <= kFastN) {
        f_cnt[m] = cnt;
#pragma unroll
        for (int k = 0; k < kFastN / 4; k++) {
          f_pk[m][2 * k] = ((unsigned)v[k].x & 0xffffu) | ((unsigned)v[k].y << 16);
          f_pk[m][2 * k + 1] = ((unsigned)v[k].z & 0xffffu) | ((unsigned)v[k].w << 16);
        }
      }
    }
  }
  tst[nst++] = clock64();
  for (int sweep = 0; sweep <= n; sweep++) {
    nsweep++;
    if (tid == 0) s_changed = 0;
    __syncthreads();
    bool changed = false;
#pragma unroll
    for (int m = 0; m < kFast; m++) {
      const int c = tid + m * kSelThreads;
      const bool open = c < n && f_cnt[m] >= 0 && status[c] == 0;
      if (!__any(open)) continue;           // after the first sweeps most wavefronts have nothing left to decide
      bool any_acc = false, all_drop = true;
#pragma unroll
      for (int k0 = 0; k0 < kFastN; k0 += 4) {
        if (!__any(open && k0 < f_cnt[m])) break;
#pragma unroll
        for (int k = k0; k < k0 + 4; k++)
          if (open && k < f_cnt[m]) {
            const int s = status[(f_pk[m][k >> 1] >> ((k & 1) * 16)) & 0xffffu];
            any_acc = any_acc || s == 1;
            all_drop = all_drop && s == 2;
          }
      }
      if (open && any_acc) { status[c] = 2; changed = true; }
      else if (open && all_drop) { status[c] = 1; changed = true; }
    }
    bool any_long = false;
#pragma unroll
    for (int q = 0; q < kFast; q++) any_long = any_long || (tid + q * kSelThreads < n && f_cnt[q] < 0);
    for (int c = tid; (any_long || n > kFast * kSelThreads) && c < n; c += kSelThreads) {
      if (status[c] != 0) continue;
      const int m = c / kSelThreads;
      bool fast = false;
#pragma unroll
      for (int q = 0; q < kFast; q++) fast = fast || (m == q && f_cnt[q] >= 0);   // handled above
      if (fast) continue;
      const int cnt = nbr_cnt[c];
      bool any_acc = false, all_drop = true;
      for (int k = 0; k < cnt; k++) {
        const int s = status[nbr[(size_t)c * kImgNbrCap + k]];
        any_acc = any_acc || s == 1;
        all_drop = all_drop && s == 2;
      }
      if (any_acc) { status[c] = 2; changed = true; }
      else if (all_drop) { status[c] = 1; changed = true; }
    }
    if (changed) s_changed = 1;
    __syncthreads();
    if (!s_changed) break;
    __syncthreads();
  }
  tst[nst++] = clock64();
  for (int c = tid; c < n; c += kSelThreads)
    if (status[c] == 1) {
      const int k = atomicAdd(&s_nacc, 1);
      const int p = clist[c];
      if (k < kImgAccCap) acc[k] = ((u64)__float_as_uint(eig[p]) << 32) | (unsigned)p;
    }
  __syncthreads();
  int nacc = s_nacc;
  if (nacc > kImgAccCap) { if (tid == 0) atomicOr(err, kErrImgAccepted); nacc = kImgAccCap; }
  // corners come out in sorted order (descending strength, ties: larger address first), cut at maxCorners
  u64* keys = reinterpret_cast<u64*>(smem);
  int P = 2;
  while (P < nacc) P <<= 1;
  __syncthreads();  // everyone is done with `status`
  for (int t = tid; t < P; t += kSelThreads) keys[t] = t < nacc ? acc[t] : 0ull;
  __syncthreads();
  tst[nst++] = clock64();
  // Bitonic network, descending.  A wavefront owns 128-element blocks (lane l: elements l and l + 64 of the block) and runs every
  // stage of stride <= 64 on them in REGISTERS (cross-lane exchange for strides < 64, the lane's own pair for stride 64); only the
  // strides >= 128 go through LDS with a workgroup barrier — 10 of the 66 stages at 2 048 keys.
  const int lane = tid & 63, wv = tid >> 6;
  auto reg_stages = [&](int k_lo, int k_hi) {   // for k = k_lo .. k_hi (doubling): strides min(k / 2, 64) .. 1 of merge level k
    for (int blk = wv; blk * 128 < P; blk += kSelThreads / 64) {
      const int base = blk * 128;
      u64 a0 = keys[base + lane], a1 = keys[base + 64 + lane];
      for (int k = k_lo; k <= k_hi; k <<= 1) {
        const bool desc0 = ((base + lane) & k) == 0, desc1 = ((base + 64 + lane) & k) == 0;
        if (k > 64) {   // stride 64: the lane's own pair (same direction: bit k is above bit 6)
          const u64 hi = a0 > a1 ? a0 : a1, lo = a0 > a1 ? a1 : a0;
          a0 = desc0 ? hi : lo; a1 = desc0 ? lo : hi;
        }
        for (int j = (k > 64 ? 32 : k >> 1); j > 0; j >>= 1) {
          const u64 b0 = __shfl_xor(a0, j), b1 = __shfl_xor(a1, j);
          const bool lower = (lane & j) == 0;
          const u64 mx0 = a0 > b0 ? a0 : b0, mn0 = a0 > b0 ? b0 : a0, mx1 = a1 > b1 ? a1 : b1, mn1 = a1 > b1 ? b1 : a1;
          a0 = (desc0 == lower) ? mx0 : mn0;
          a1 = (desc1 == lower) ? mx1 : mn1;
        }
      }
      keys[base + lane] = a0; keys[base + 64 + lane] = a1;
    }
  };
  if (P < 128) {   // tiny lists: the plain network
    for (int k = 2; k <= P; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < (P >> 1); i += kSelThreads) {
          const int t = ((i & ~(j - 1)) << 1) | (i & (j - 1)), l = t | j;
          const u64 a = keys[t], b = keys[l];
          const bool desc = (t & k) == 0;
          if (desc ? a < b : a > b) { keys[t] = b; keys[l] = a; }
        }
        __syncthreads();
      }
  } else {
    reg_stages(2, 128);        // merge levels 2 .. 128 entirely in registers
    __syncthreads();
    for (int k = 256; k <= P; k <<= 1) {
      for (int j = k >> 1; j >= 128; j >>= 1) {
        for (int i = tid; i < (P >> 1); i += kSelThreads) {   // one compare-exchange per lane: pair i = (t, t | j)
          const int t = ((i & ~(j - 1)) << 1) | (i & (j - 1)), l = t | j;
          const u64 a = keys[t], b = keys[l];
          const bool desc = (t & k) == 0;   // descending blocks first: the whole array ends up descending
          if (desc ? a < b : a > b) { keys[t] = b; keys[l] = a; }
        }
        __syncthreads();
      }
      reg_stages(k, k);        // strides 64 .. 1 of this level
      __syncthreads();
    }
  }
  __syncthreads();
  tst[nst++] = clock64();
  const int nout = max_corners > 0 ? min(nacc, max_corners) : nacc;
  for (int t = tid; t < nout && t < kImgMaxCorners; t += kSelThreads) {
    const int p = (int)(keys[t] & 0xffffffffu);
    const int y = p / w, x = p - y * w;
    corners[t] = make_float2((float)x, (float)y);
  }
  if (tid == 0) { *n_corners = min(nout, kImgMaxCorners); tst[nst++] = clock64(); for (int q = 0; q + 1 < nst; q++) acc[kImgAccCap - 8 + q] = (u64)(tst[q + 1] - tst[q]); acc[kImgAccCap - 2] = (u64)nsweep; acc[kImgAccCap - 1] = (u64)nacc; }
}

__global__ __launch_bounds__(256) void k_img_pyrdown(const unsigned char* __restrict__ src, int sw, int sh, unsigned char* __restrict__ dst, int dw,
                                                     int dh) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= dw * dh) return;
  const int y = idx / dw, x = idx - y * dw;
  int xs[5];
#pragma unroll
  for (int i = 0; i < 5; i++) xs[i] = reflect101(2 * x + i - 2, sw);
  int s = 0;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const unsigned char* row = src + (size_t)reflect101(2 * y + j - 2, sh) * sw;
    const int rs = (int)row[xs[0]] + 4 * (int)row[xs[1]] + 6 * (int)row[xs[2]] + 4 * (int)row[xs[3]] + (int)row[xs[4]];
    s += (j == 0 || j == 4) ? rs : ((j == 2) ? 6 * rs : 4 * rs);
  }
  dst[idx] = (unsigned char)((s + 128) >> 8);
}

// calcSharrDeriv of every pyramid level in one launch (workgroup -> level by the levels' block counts)
__global__ __launch_bounds__(256) void k_img_scharr(ImgPyrDev P) {
  int b = blockIdx.x, level = 0;
  for (; level < P.levels - 1; level++) {
    const int nb = (P.w[level] * P.h[level] + 255) / 256;
    if (b < nb) break;
    b -= nb;
  }
  const int w = P.w[level], h = P.h[level];
  const unsigned char* __restrict__ img = P.img[level];
  short2* __restrict__ deriv = P.deriv[level];
  const int idx = b * 256 + threadIdx.x;
  if (idx >= w * h) return;
  const int y = idx / w, x = idx - y * w;
  const unsigned char* s0 = img + (size_t)reflect101(y - 1, h) * w;
  const unsigned char* s1 = img + (size_t)y * w;
  const unsigned char* s2 = img + (size_t)reflect101(y + 1, h) * w;
  const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
  const int t0m = ((int)s0[xm] + (int)s2[xm]) * 3 + (int)s1[xm] * 10, t0p = ((int)s0[xp] + (int)s2[xp]) * 3 + (int)s1[xp] * 10;
  const int t1m = (int)s2[xm] - (int)s0[xm], t1c = (int)s2[x] - (int)s0[x], t1p = (int)s2[xp] - (int)s0[xp];
  deriv[idx] = make_short2((short)(t0p - t0m), (short)((t1p + t1m) * 3 + t1c * 10));
}

// Wavefront-wide integer sum through DPP row operations (no LDS crossbar): quad swaps, row rotations, then the row_bcast15 /
// row_bcast31 carries of gfx9; the total lands in lane 63 and is broadcast.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add_step(int v) {
  return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);  // lanes without a source add 0
}
__device__ __forceinline__ int wave_sum_i32(int v) {
  v = dpp_add_step<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add_step<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_add_step<0x124, 0xf>(v);  // row_ror:4
  v = dpp_add_step<0x128, 0xf>(v);  // row_ror:8   -> every lane holds the sum of its row of 16
  v = dpp_add_step<0x142, 0xa>(v);  // row_bcast15 -> rows 1 and 3 take in the row below
  v = dpp_add_step<0x143, 0xc>(v);  // row_bcast31 -> rows 2 and 3 take in lane 31
  return __builtin_amdgcn_readlane(v, 63);
}
// exact wavefront sum of per-lane integers below 2^31 in magnitude: low 16 bits and the (signed) rest are summed separately
__device__ __forceinline__ long long wave_sum_exact(long long v) {
  const int lo = (int)(v & 0xffff), hi = (int)(v >> 16);
  return ((long long)wave_sum_i32(hi) << 16) + (long long)wave_sum_i32(lo);
}
__device__ __forceinline__ int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }   // CV_DESCALE

// LKTrackerInvoker::operator() for one point, levels from the top (lkpyramid.cpp).  Pyramid levels: REFLECT_101 border, derivative
// images: zero border.  Every lane runs the same scalar control flow on wavefront-uniform values; the window sums are the only
// cross-lane step.
__global__ __launch_bounds__(256) void k_img_lk(ImgPyrDev P, ImgPyrDev N, const float2* __restrict__ pts, const int* __restrict__ n_pts,
                                                float2* __restrict__ out, unsigned char* __restrict__ status, int* __restrict__ prev_uv,
                                                int* __restrict__ curr_uv, double eps2) {
  // The search region of the NEXT image, staged once per level in LDS: the window moves by fractions of a pixel per iteration, so
  // the ten iterations read the same 32 x 32 patch (window + 8 pixels of slack all round) instead of making ten dependent trips to
  // L2; a window that walks out of the patch reloads it.  Border pixels are reflected when the patch is loaded.
  constexpr int kPS = 32, kSlack = (kPS - (kImgWin + 1)) / 2;
  __shared__ unsigned char s_patch[4][kPS][kPS + 4];
  unsigned char (*patch)[kPS + 4] = s_patch[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63, p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= kImgMaxCorners) return;
  if (p >= *n_pts) {
    if (lane == 0) {
      status[p] = 0;
      out[p] = make_float2(0.f, 0.f);
      if (prev_uv) { prev_uv[2 * p] = INT_MIN; prev_uv[2 * p + 1] = 0; curr_uv[2 * p] = INT_MIN; curr_uv[2 * p + 1] = 0; }
    }
    return;
  }
  constexpr int win = kImgWin, W_BITS = 14, kQ = (win * win + 63) / 64;
  const float FLT_SCALE = 1.f / (1 << 20);
  const float half = (win - 1) * 0.5f;
  const float2 pt = pts[p];
  int wxq[kQ], wyq[kQ];      // this lane's window pixels (wi = q * 64 + lane < win * win)
#pragma unroll
  for (int q = 0; q < kQ; q++) { const int wi = q * 64 + lane; wyq[q] = wi / win; wxq[q] = wi - wyq[q] * win; }
  float ox = 0.f, oy = 0.f;  // nextPts[ptidx]
  bool st = true;
  const int top = min(P.levels, N.levels) - 1;
  for (int level = top; level >= 0; level--) {
    const int cw = P.w[level], ch = P.h[level];
    const unsigned char* I = P.img[level];
    const unsigned char* J = N.img[level];
    const short2* dI = P.deriv[level];
    float px = pt.x * (float)(1. / (1 << level)), py = pt.y * (float)(1. / (1 << level));
    float nx, ny;
    if (level == top) { nx = px; ny = py; }
    else { nx = ox * 2.f; ny = oy * 2.f; }
    ox = nx; oy = ny;
    px -= half; py -= half;
    const int ipx = (int)floorf(px), ipy = (int)floorf(py);
    if (ipx < -win || ipx >= cw || ipy < -win || ipy >= ch) {
      if (level == 0) st = false;
      continue;
    }
    float a = px - ipx, b = py - ipy;
    int iw00 = (int)rintf((1.f - a) * (1.f - b) * (1 << W_BITS));
    int iw01 = (int)rintf(a * (1.f - b) * (1 << W_BITS));
    int iw10 = (int)rintf((1.f - a) * b * (1 << W_BITS));
    int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
    int Iw[kQ], Ix[kQ], Iy[kQ];
    long long sA11 = 0, sA12 = 0, sA22 = 0;   // per lane: 4 products of two 13-bit values
#pragma unroll
    for (int q = 0; q < kQ; q++) {
      const int wi = q * 64 + lane;
      Iw[q] = 0; Ix[q] = 0; Iy[q] = 0;
      if (wi < win * win) {
        const int X = ipx + wxq[q], Y = ipy + wyq[q];
        const int x0 = reflect101_near(X, cw), x1 = reflect101_near(X + 1, cw), y0 = reflect101_near(Y, ch), y1 = reflect101_near(Y + 1, ch);
        Iw[q] = descale((int)I[(size_t)y0 * cw + x0] * iw00 + (int)I[(size_t)y0 * cw + x1] * iw01 + (int)I[(size_t)y1 * cw + x0] * iw10 +
                        (int)I[(size_t)y1 * cw + x1] * iw11, W_BITS - 5);
        const bool bx0 = X >= 0 && X < cw, bx1 = X + 1 >= 0 && X + 1 < cw, by0 = Y >= 0 && Y < ch, by1 = Y + 1 >= 0 && Y + 1 < ch;
        const short2 z = make_short2(0, 0);
        const short2 d00 = (bx0 && by0) ? dI[(size_t)Y * cw + X] : z, d01 = (bx1 && by0) ? dI[(size_t)Y * cw + X + 1] : z;
        const short2 d10 = (bx0 && by1) ? dI[(size_t)(Y + 1) * cw + X] : z, d11 = (bx1 && by1) ? dI[(size_t)(Y + 1) * cw + X + 1] : z;
        Ix[q] = descale((int)d00.x * iw00 + (int)d01.x * iw01 + (int)d10.x * iw10 + (int)d11.x * iw11, W_BITS);
        Iy[q] = descale((int)d00.y * iw00 + (int)d01.y * iw01 + (int)d10.y * iw10 + (int)d11.y * iw11, W_BITS);
        sA11 += (long long)(Ix[q] * Ix[q]); sA12 += (long long)(Ix[q] * Iy[q]); sA22 += (long long)(Iy[q] * Iy[q]);
      }
    }
    sA11 = wave_sum_exact(sA11); sA12 = wave_sum_exact(sA12); sA22 = wave_sum_exact(sA22);
    const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float min_eig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
    if ((double)min_eig < 1e-4 || D < FLT_EPSILON) {   // minEigThreshold
      if (level == 0) st = false;
      continue;
    }
    D = 1.f / D;
    nx -= half; ny -= half;
    float pdx = 0.f, pdy = 0.f;
    int pox = INT_MIN, poy = 0;   // top-left corner of the staged patch (this level's image)
    for (int j = 0; j < kImgLkIters; j++) {
      const int inx = (int)floorf(nx), iny = (int)floorf(ny);
      if (inx < -win || inx >= cw || iny < -win || iny >= ch) {
        if (level == 0) st = false;
        break;
      }
      if (pox == INT_MIN || inx < pox || inx > pox + kPS - (win + 1) || iny < poy || iny > poy + kPS - (win + 1)) {   // wavefront-uniform
        pox = inx - kSlack; poy = iny - kSlack;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();          // earlier reads of the old patch are done
        const int row = lane >> 1, c0 = (lane & 1) * (kPS / 2);
        const unsigned char* src = J + (size_t)reflect101_near(poy + row, ch) * cw;
        int v[kPS / 2];
#pragma unroll
        for (int i = 0; i < kPS / 2; i++) v[i] = src[reflect101_near(pox + c0 + i, cw)];   // all 16 loads in flight together
#pragma unroll
        for (int i = 0; i < kPS / 2; i++) patch[row][c0 + i] = (unsigned char)v[i];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
      }
      a = nx - inx; b = ny - iny;
      iw00 = (int)rintf((1.f - a) * (1.f - b) * (1 << W_BITS));
      iw01 = (int)rintf(a * (1.f - b) * (1 << W_BITS));
      iw10 = (int)rintf((1.f - a) * b * (1 << W_BITS));
      iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      long long sb1 = 0, sb2 = 0;
#pragma unroll
      for (int q = 0; q < kQ; q++) {
        const int wi = q * 64 + lane;
        if (wi < win * win) {
          const int lx = inx - pox + wxq[q], ly = iny - poy + wyq[q];
          const int diff = descale((int)patch[ly][lx] * iw00 + (int)patch[ly][lx + 1] * iw01 + (int)patch[ly + 1][lx] * iw10 +
                                   (int)patch[ly + 1][lx + 1] * iw11, W_BITS - 5) - Iw[q];
          sb1 += (long long)(diff * Ix[q]);
          sb2 += (long long)(diff * Iy[q]);
        }
      }
      sb1 = wave_sum_exact(sb1); sb2 = wave_sum_exact(sb2);
      const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
      const float ddx = (A12 * b2 - A22 * b1) * D, ddy = (A12 * b1 - A11 * b2) * D;
      nx += ddx; ny += ddy;
      ox = nx + half; oy = ny + half;
      if ((double)ddx * (double)ddx + (double)ddy * (double)ddy <= eps2) break;
      if (j > 0 && fabs((double)(ddx + pdx)) < 0.01 && fabs((double)(ddy + pdy)) < 0.01) {
        ox -= ddx * 0.5f;
        oy -= ddy * 0.5f;
        break;
      }
      pdx = ddx; pdy = ddy;
    }
    if (st && level == 0) {  // the error measure's window check (the reference asks for `err`)
      const int ix = (int)floorf(ox - half), iy = (int)floorf(oy - half);
      if (ix < -win || ix >= cw || iy < -win || iy >= ch) st = false;
    }
  }
  if (lane == 0) {
    out[p] = make_float2(ox, oy);
    status[p] = st ? 1 : 0;
    if (prev_uv) {  // visual_odometry.cpp:303-306: int = float (truncation); untracked corners are skipped by the match loop (:308)
      prev_uv[2 * p] = st ? (int)pt.x : INT_MIN; prev_uv[2 * p + 1] = (int)pt.y;
      curr_uv[2 * p] = st ? (int)ox : INT_MIN; curr_uv[2 * p + 1] = st ? (int)oy : 0;
    }
  }
}


// Brute-force Hamming matcher (cv::BFMatcher(NORM_HAMMING)::knnMatch with k = 2): one wavefront per query descriptor; every lane walks
// the train descriptors with stride 64 and keeps its two smallest keys (distance << 16 | train index): unique, and ordered the way
// cv::batchDistance resolves ties (strictly smaller distance wins, i.e. the lower index among equals).
__device__ __forceinline__ void bf_knn_body(const unsigned* __restrict__ q_desc, int nq, const unsigned* __restrict__ t_desc, int nt, int words, int stride_words,
                                            uint2* __restrict__ best2) {
  const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;
  unsigned qw[kImgMaxDescBytes / 4];
#pragma unroll
  for (int w = 0; w < kImgMaxDescBytes / 4; w++) qw[w] = w < words ? q_desc[(size_t)q * stride_words + w] : 0u;
  unsigned k0 = 0xffffffffu, k1 = 0xffffffffu;
  for (int t = lane; t < nt; t += 64) {
    const unsigned* td = t_desc + (size_t)t * stride_words;
    int d = 0;
#pragma unroll
    for (int w = 0; w < kImgMaxDescBytes / 4; w++) if (w < words) d += __popc(qw[w] ^ td[w]);
    const unsigned key = ((unsigned)d << 16) | (unsigned)t;
    if (key < k1) { if (key < k0) { k1 = k0; k0 = key; } else k1 = key; }
  }
  unsigned g0 = k0;
  for (int s = 32; s > 0; s >>= 1) { const unsigned o = __shfl_xor(g0, s); g0 = o < g0 ? o : g0; }
  unsigned c1 = (k0 == g0 && g0 != 0xffffffffu) ? k1 : k0;   // the winner's lane offers its runner-up
  for (int s = 32; s > 0; s >>= 1) { const unsigned o = __shfl_xor(c1, s); c1 = o < c1 ? o : c1; }
  if (lane == 0) best2[q] = make_uint2(g0, c1);
}
__global__ __launch_bounds__(256) void k_img_bf_knn(const unsigned* __restrict__ q_desc, int nq, const unsigned* __restrict__ t_desc, int nt, int words,
                                                    uint2* __restrict__ best2) {
  bf_knn_body(q_desc, nq, t_desc, nt, words, words, best2);
}

// ---- ORB + brute-force configuration of processImage (optical_flow_match = false: vloam_main.launch:10, visual_odometry.cpp:106-116)
// cv::GaussianBlur(level, Size(7, 7), 2, 2, BORDER_REFLECT_101) as OpenCV's bit-exact fixed-point path computes it for 8-bit images
// (oracle/orc_img.h: Q8 kernel {18, 34, 48, 56, 48, 34, 18}, exact horizontal pass, vertical pass rounded once): integers only.
__constant__ int kGaussQ8[7] = {18, 34, 48, 56, 48, 34, 18};
__global__ __launch_bounds__(256) void k_img_gauss_h(const unsigned char* __restrict__ img, int w, int h, int stride, unsigned short* __restrict__ row) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= w * h) return;
  const int y = p / w, x = p - y * w;
  const unsigned char* src = img + (size_t)y * stride;
  int s = 0;
#pragma unroll
  for (int k = -3; k <= 3; k++) s += kGaussQ8[k + 3] * (int)src[reflect101_near(x + k, w)];
  row[p] = (unsigned short)s;
}
__global__ __launch_bounds__(256) void k_img_gauss_v(const unsigned short* __restrict__ row, int w, int h, unsigned char* __restrict__ out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= w * h) return;
  const int y = p / w, x = p - y * w;
  unsigned s = 0;
#pragma unroll
  for (int k = -3; k <= 3; k++) s += (unsigned)kGaussQ8[k + 3] * (unsigned)row[(size_t)reflect101_near(y + k, h) * w + x];
  out[p] = (unsigned char)((s + (1u << 15)) >> 16);
}
// KeyPointsFilter::runByImageBorder(edgeThreshold 31) + computeOrbDescriptors on the surviving corners, order kept (image_util.cpp:203 hands the
// keypoint vector in by reference: the matcher's indices refer to the FILTERED list).  One workgroup: the filter is a 1 024-wide ordered
// compaction; a thread then computes one 32-bit word (32 tests, 64 pixel reads of the blurred image) of one descriptor at a time.
__global__ __launch_bounds__(1024) void k_img_orb(const float2* __restrict__ corners, const int* __restrict__ n_corners, const unsigned char* __restrict__ blur,
                                                  int w, int h, const short2* __restrict__ off, float2* __restrict__ okp, int* __restrict__ n_okp,
                                                  unsigned* __restrict__ desc) {
  __shared__ int s_wave[16], s_total;
  __shared__ short2 s_off[512];
  __shared__ int s_cx[kImgMaxCorners], s_cy[kImgMaxCorners];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 512) s_off[tid] = off[tid];
  const int n = min(*n_corners, kImgMaxCorners);
  float2 pt = make_float2(0.f, 0.f);
  bool keep = false;
  if (tid < n) {
    pt = corners[tid];
    const float edge = 31.f;   // cv::ORB::create(): edgeThreshold 31
    keep = pt.x >= edge && pt.x < (float)(w - 31) && pt.y >= edge && pt.y < (float)(h - 31);
  }
  const unsigned long long m = __ballot(keep);
  if (lane == 0) s_wave[wave] = __popcll(m);
  __syncthreads();
  int base = 0;
  for (int q = 0; q < wave; q++) base += s_wave[q];
  if (tid == 0) { int t = 0; for (int q = 0; q < 16; q++) t += s_wave[q]; s_total = t; *n_okp = t; }
  if (keep) {
    const int o = base + __popcll(m & ((1ull << lane) - 1ull));
    okp[o] = pt;
    s_cx[o] = (int)rintf(pt.x); s_cy[o] = (int)rintf(pt.y);   // cvRound(kpt.pt * scale), scale = 1
  }
  __syncthreads();
  const int total = s_total;
  for (int e = tid; e < total * 8; e += 1024) {
    const int kp = e >> 3, word = e & 7;
    const unsigned char* c = blur + (size_t)s_cy[kp] * w + s_cx[kp];
    unsigned val = 0;
#pragma unroll 8
    for (int bit = 0; bit < 32; bit++) {
      const int t = 32 * word + bit;
      const short2 o0 = s_off[2 * t], o1 = s_off[2 * t + 1];
      const int v0 = c[o0.y * w + o0.x], v1 = c[o1.y * w + o1.x];
      val |= (unsigned)(v0 < v1) << bit;   // byte j of the descriptor = tests 8 j .. 8 j + 7, LSB first: word = four bytes, little endian
    }
    desc[(size_t)kp * (kImgMaxDescBytes / 4) + word] = val;
  }
}
// matchDescriptors(descriptors[1 - i], descriptors[i]) with BF / NORM_HAMMING / KNN (image_util.cpp:221-296): the previous image's descriptors
// are the queries; counts come from the device (no host round trip in the frame loop)
__global__ __launch_bounds__(256) void k_img_bf_knn_dev(const unsigned* __restrict__ q_desc, const int* __restrict__ nq, const unsigned* __restrict__ t_desc,
                                                        const int* __restrict__ nt, uint2* __restrict__ best2) {
  bf_knn_body(q_desc, *nq, t_desc, *nt, 8, kImgMaxDescBytes / 4, best2);
}
// ... the ratio test (knn_match[0].distance < 0.8 * knn_match[1].distance, float distances, double product) and the match loop's reads
// (visual_odometry.cpp:296-303: keypoints[1 - i][queryIdx].pt / keypoints[i][trainIdx].pt truncated to int) into the frame's match slots,
// query order; a slot without a match is marked like an untracked corner
__global__ void k_img_orb_matches(const uint2* __restrict__ best2, const int* __restrict__ nq, const int* __restrict__ nt, const float2* __restrict__ kp_prev,
                                  const float2* __restrict__ kp_curr, int* __restrict__ prev_uv, int* __restrict__ curr_uv) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= kImgMaxCorners) return;
  bool ok = false;
  int t = 0;
  if (q < *nq && *nt >= 2) {   // (a single train descriptor: the reference would read knn_match[1] out of range)
    const uint2 b = best2[q];
    ok = b.y != 0xffffffffu && (double)(float)(b.x >> 16) < 0.8 * (double)(float)(b.y >> 16);
    t = (int)(b.x & 0xffffu);
  }
  if (prev_uv) {
    const float2 a = ok ? kp_prev[q] : make_float2(0.f, 0.f), c = ok ? kp_curr[t] : make_float2(0.f, 0.f);
    prev_uv[2 * q] = ok ? (int)a.x : INT_MIN; prev_uv[2 * q + 1] = ok ? (int)a.y : 0;
    curr_uv[2 * q] = ok ? (int)c.x : INT_MIN; curr_uv[2 * q + 1] = ok ? (int)c.y : 0;
  }
}

__global__ void k_img_no_matches(int* prev_uv, int* curr_uv) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p < kImgMaxCorners) { prev_uv[2 * p] = INT_MIN; prev_uv[2 * p + 1] = 0; curr_uv[2 * p] = INT_MIN; curr_uv[2 * p + 1] = 0; }
}

// ---- host side
static void pyr_dims(int w, int h, ImgPyrDev* P) {   // buildOpticalFlowPyramid's early stop: the next level must be larger than the window
  P->w[0] = w; P->h[0] = h; P->levels = 1;
  for (int l = 0; l < kImgMaxLevel; l++) {
    const int nw = (P->w[l] + 1) / 2, nh = (P->h[l] + 1) / 2;
    if (nw <= kImgWin || nh <= kImgWin) break;
    P->w[l + 1] = nw; P->h[l + 1] = nh; P->levels = l + 2;
  }
}

// per device (function attributes belong to the device's code object): k_img_select sorts up to kImgAccCap keys in 128 KB of dynamic LDS
hipError_t img_init() {
  return hipFuncSetAttribute((const void*)k_img_select, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(u64) * kImgAccCap));
}

vloam_status img_layout(ImgContext* c, const vloam_config& cfg, Arena& A) {
  c->max_w = cfg.image_width; c->max_h = cfg.image_height;
  if (c->max_w <= 0 || c->max_h <= 0) { c->max_w = c->max_h = 0; return VLOAM_OK; }
  const size_t npx = (size_t)c->max_w * c->max_h;
  bool ok = true;
  for (int s = 0; s < 2; s++) {
    ImgPyrDev tmp;
    pyr_dims(c->max_w, c->max_h, &tmp);
    for (int l = 0; l < kImgLevels; l++) {
      const size_t n = l < tmp.levels ? (size_t)tmp.w[l] * tmp.h[l] : 1;
      ok = ok && A.take(&c->pyr[s].img[l], n) && A.take(&c->pyr[s].deriv[l], n);
    }
    ok = ok && A.take(&c->corners[s], kImgMaxCorners) && A.take(&c->n_corners[s], 1);
  }
  ok = ok && A.take(&c->sobel, npx) && A.take(&c->eig, npx) && A.take(&c->maxbits, 1) && A.take(&c->cmap, npx) && A.take(&c->clist, kImgCandCap) &&
       A.take(&c->n_cand, 1) && A.take(&c->nbr, (size_t)kImgCandCap * kImgNbrCap) && A.take(&c->nbr_cnt, kImgCandCap) && A.take(&c->acc, kImgAccCap) &&
       A.take(&c->tracked, kImgMaxCorners) && A.take(&c->status, kImgMaxCorners) && A.take(&c->error, 1) && A.take(&c->staging, npx);
  c->clahe = cfg.CLAHE != 0;
  ok = ok && A.take(&c->clahe_img, npx) && A.take(&c->clahe_lut, (size_t)kImgClaheTiles * kImgClaheTiles * 256);
  for (int k = 0; k < 2; k++) ok = ok && A.take(&c->desc[k], (size_t)kImgMaxDesc * (kImgMaxDescBytes / 4)) && A.take(&c->best2[k], kImgMaxDesc);
  ok = ok && A.take(&c->orb_off, 512) && A.take(&c->blur, npx);
  for (int k = 0; k < 2; k++) ok = ok && A.take(&c->okp[k], kImgMaxCorners) && A.take(&c->n_okp[k], 1);
  return ok ? VLOAM_OK : VLOAM_ERR_CAPACITY;
}

vloam_status img_check(const ImgContext* c, int width, int height, int stride) {
  if (c->max_w == 0) return VLOAM_ERR_ORDER;
  // per DIMENSION, not per area: the pyramid and derivative buffers are sized from the level dimensions of max_w x max_h, and an image of the
  // same area but another aspect ratio has larger (and possibly more) levels than were allocated
  if (width < 2 * kImgWin || height < 2 * kImgWin || width > c->max_w || height > c->max_h || stride < width) return VLOAM_ERR_INVALID;
  if (c->count >= 0 && (width != c->w || height != c->h)) return VLOAM_ERR_INVALID;   // one image size per sequence
  return VLOAM_OK;
}

vloam_status img_process(ImgContext* c, hipStream_t st, const unsigned char* d_gray, int width, int height, int stride, int* prev_uv, int* curr_uv,
                         ProfHook* ph) {
  { const vloam_status chk = img_check(c, width, height, stride); if (chk != VLOAM_OK) return chk; }
  c->w = width; c->h = height;
  c->count++;
  const int cur = c->count % 2;
  ImgPyrDev& P = c->pyr[cur];
  pyr_dims(width, height, &P);
  const int npx = width * height, gpx = (npx + 255) / 256;
  if (c->clahe) {   // visual_odometry.cpp:97-98: clahe->apply(img00, images[i]) — everything below then sees the equalised image
    const int T = kImgClaheTiles;
    const bool fits = width % T == 0 && height % T == 0;
    const int ew = fits ? width : width + (T - width % T), eh = fits ? height : height + (T - height % T);
    const int tw = ew / T, th = eh / T;
    const int clip = max((int)(2.0 * (tw * th) / 256), 1);   // clipLimit 2.0 (visual_odometry.cpp:31)
    VL_RAW_LAUNCH(k_img_clahe_lut, dim3(T * T), dim3(256), 0, st, d_gray, width, height, stride, T, tw, th, clip, 255.0f / (float)(tw * th), c->clahe_lut);
    VL_RAW_LAUNCH(k_img_clahe_apply, dim3(gpx), dim3(256), 0, st, d_gray, width, height, stride, T, 1.0f / (float)tw, 1.0f / (float)th, c->clahe_lut, c->clahe_img);
    d_gray = c->clahe_img;
    stride = width;
  }
  // image_util.cpp:17-31: block_size 5, min_distance 7.5, maxCorners 1024, quality_level 0.03
  const double quality = 0.03, min_distance = kImgBlock * 1.5;
  const double scale = 1.0 / (4.0 * (double)kImgBlock * 255.0);
  const double hs2 = 0.5 * scale * scale;
  VLOAM_LAUNCH(ph, kKImgSobel, st, k_img_sobel, dim3(gpx), dim3(256), 0, st, d_gray, width, height, stride, c->sobel, P.img[0], c->maxbits, c->n_cand);
  VLOAM_LAUNCH(ph, kKImgEig, st, k_img_eig, dim3((width + kTileW - 1) / kTileW, (height + kTileH - 1) / kTileH), dim3(kTileW * kTileH), 0, st, c->sobel, width,
               height, hs2, c->eig, c->maxbits);
  VLOAM_LAUNCH(ph, kKImgLocalMax, st, k_img_localmax, dim3(gpx), dim3(256), 0, st, c->eig, width, height, c->maxbits, quality, c->cmap, c->clist, c->n_cand,
               c->error);
  VLOAM_LAUNCH(ph, kKImgNeighbours, st, k_img_neighbours, dim3(1024), dim3(256), 0, st, c->eig, width, height, c->cmap, c->clist, c->n_cand,
               (float)(min_distance * min_distance), (int)min_distance, c->nbr, c->nbr_cnt, c->error);
  VLOAM_LAUNCH(ph, kKImgSelect, st, k_img_select, dim3(1), dim3(kSelThreads), sizeof(u64) * kImgAccCap, st, c->eig, width, c->clist, c->n_cand, c->nbr,
               c->nbr_cnt, c->acc, kImgMaxCorners, c->corners[cur], c->n_corners[cur], c->error);
  if (c->orb) {
    // visual_odometry.cpp:106-116 with optical_flow_match = false: descKeypoints on this image's corners, then — from the second image on —
    // matchDescriptors(previous, this).  No pyramid, no flow.
    unsigned short* rows = reinterpret_cast<unsigned short*>(c->sobel);   // (the Sobel pairs have been consumed by k_img_eig: same stream)
    VL_RAW_LAUNCH(k_img_gauss_h, dim3(gpx), dim3(256), 0, st, d_gray, width, height, stride, rows);
    VL_RAW_LAUNCH(k_img_gauss_v, dim3(gpx), dim3(256), 0, st, rows, width, height, c->blur);
    VL_RAW_LAUNCH(k_img_orb, dim3(1), dim3(1024), 0, st, c->corners[cur], c->n_corners[cur], c->blur, width, height, c->orb_off, c->okp[cur], c->n_okp[cur], c->desc[cur]);
    if (c->count > 0) {
      VL_RAW_LAUNCH(k_img_bf_knn_dev, dim3(kImgMaxCorners / 4), dim3(256), 0, st, c->desc[1 - cur], c->n_okp[1 - cur], c->desc[cur], c->n_okp[cur], c->best2[0]);
      VL_RAW_LAUNCH(k_img_orb_matches, dim3(kImgMaxCorners / 256), dim3(256), 0, st, c->best2[0], c->n_okp[1 - cur], c->n_okp[cur], c->okp[1 - cur], c->okp[cur],
                    prev_uv, curr_uv);
    } else if (prev_uv) {
      VL_RAW_LAUNCH(k_img_no_matches, dim3(kImgMaxCorners / 256), dim3(256), 0, st, prev_uv, curr_uv);
    }
    return hipGetLastError() == hipSuccess ? VLOAM_OK : VLOAM_ERR_HIP;
  }
  for (int l = 1; l < P.levels; l++)
    VLOAM_LAUNCH(ph, kKImgPyrDown, st, k_img_pyrdown, dim3((P.w[l] * P.h[l] + 255) / 256), dim3(256), 0, st, P.img[l - 1], P.w[l - 1], P.h[l - 1], P.img[l],
                 P.w[l], P.h[l]);
  int scharr_blocks = 0;
  for (int l = 0; l < P.levels; l++) scharr_blocks += (P.w[l] * P.h[l] + 255) / 256;
  VLOAM_LAUNCH(ph, kKImgScharr, st, k_img_scharr, dim3(scharr_blocks), dim3(256), 0, st, P);
  if (c->count > 0) {
    const double eps = 0.03;   // image_util.cpp:362 TermCriteria(COUNT + EPS, 10, 0.03); calcOpticalFlowPyrLK squares epsilon
    VLOAM_LAUNCH(ph, kKImgLk, st, k_img_lk, dim3(kImgMaxCorners / 4), dim3(256), 0, st, c->pyr[1 - cur], P, c->corners[cur], c->n_corners[cur], c->tracked,
                 c->status, prev_uv, curr_uv, eps * eps);
  } else if (prev_uv) {
    VL_RAW_LAUNCH(k_img_no_matches, dim3(kImgMaxCorners / 256), dim3(256), 0, st, prev_uv, curr_uv);
  }
  return hipGetLastError() == hipSuccess ? VLOAM_OK : VLOAM_ERR_HIP;
}


vloam_status img_set_orb_pattern(ImgContext* c, hipStream_t st, const signed char* pattern, int n_sessions, size_t ss) {
  if (c->max_w == 0) return VLOAM_ERR_ORDER;
  if (!pattern) { c->orb = false; return VLOAM_OK; }
  // computeOrbDescriptors (orb.cpp) for a keypoint with angle = -1 (goodFeaturesToTrack keypoints carry no orientation, image_util.cpp:30-34):
  //   float angle = kpt.angle; angle *= (float)(CV_PI / 180.f); float a = (float)cos(angle), b = (float)sin(angle);
  //   x = pattern[idx].x * a - pattern[idx].y * b;  y = pattern[idx].x * b + pattern[idx].y * a;  ix = cvRound(x), iy = cvRound(y)
  // — the same for every keypoint, so the steered offsets are computed once, here (f32 products and sums, no contraction: csrc/Makefile)
  float angle = -1.0f;
  angle *= (float)(3.14159265358979323846 / 180.0);
  const float a = (float)cos((double)angle), b = (float)sin((double)angle);
  short2 off[512];
  for (int i = 0; i < 512; i++) {
    const float px = (float)pattern[2 * i], py = (float)pattern[2 * i + 1];
    const float x = px * a - py * b, y = px * b + py * a;
    off[i] = make_short2((short)lrintf(x), (short)lrintf(y));
    if (abs((int)off[i].x) > 30 || abs((int)off[i].y) > 30) return VLOAM_ERR_INVALID;   // must stay inside the 31-pixel border the keypoints keep
  }
  for (int b_ = 0; b_ < n_sessions; b_++)
    if (hipMemcpyAsync((char*)c->orb_off + (size_t)b_ * ss, off, sizeof(off), hipMemcpyHostToDevice, st) != hipSuccess) return VLOAM_ERR_HIP;
  if (hipStreamSynchronize(st) != hipSuccess) return VLOAM_ERR_HIP;
  c->orb = true;
  return VLOAM_OK;
}

vloam_status img_match_descriptors(ImgContext* c, hipStream_t st, const unsigned char* desc0, int n0, const unsigned char* desc1, int n1, int bytes, bool knn,
                                   int* query_idx, int* train_idx, int cap, int* n_matches) {
  if (c->max_w == 0) return VLOAM_ERR_ORDER;
  if (n0 < 0 || n1 < 0 || n0 > kImgMaxDesc || n1 > kImgMaxDesc) return VLOAM_ERR_CAPACITY;
  if (bytes <= 0 || bytes > kImgMaxDescBytes || (bytes & 3)) return VLOAM_ERR_INVALID;
  *n_matches = 0;
  if (n0 == 0 || n1 == 0) return VLOAM_OK;
  const int words = bytes / 4;
  // (hipMemcpyAsync from pageable memory has read its source when it returns)
  if (hipMemcpyAsync(c->desc[0], desc0, (size_t)n0 * bytes, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(c->desc[1], desc1, (size_t)n1 * bytes, hipMemcpyHostToDevice, st) != hipSuccess) return VLOAM_ERR_HIP;
  VL_RAW_LAUNCH(k_img_bf_knn, dim3((n0 + 3) / 4), dim3(256), 0, st, c->desc[0], n0, c->desc[1], n1, words, c->best2[0]);
  if (!knn) VL_RAW_LAUNCH(k_img_bf_knn, dim3((n1 + 3) / 4), dim3(256), 0, st, c->desc[1], n1, c->desc[0], n0, words, c->best2[1]);
  std::vector<uint2> fwd((size_t)n0), bwd((size_t)(knn ? 0 : n1));
  if (hipMemcpyAsync(fwd.data(), c->best2[0], sizeof(uint2) * (size_t)n0, hipMemcpyDeviceToHost, st) != hipSuccess) return VLOAM_ERR_HIP;
  if (!knn && hipMemcpyAsync(bwd.data(), c->best2[1], sizeof(uint2) * (size_t)n1, hipMemcpyDeviceToHost, st) != hipSuccess) return VLOAM_ERR_HIP;
  if (hipStreamSynchronize(st) != hipSuccess) return VLOAM_ERR_HIP;
  int m = 0;
  for (int q = 0; q < n0; q++) {
    const unsigned g0 = fwd[(size_t)q].x, g1 = fwd[(size_t)q].y;
    bool keep;
    if (knn) {   // image_util.cpp:262-271: knn_match[0].distance < 0.8 * knn_match[1].distance (float distance, double product)
      if (g1 == 0xffffffffu) continue;   // a single train descriptor: the reference would read knn_match[1] out of range
      keep = (double)(float)(g0 >> 16) < 0.8 * (double)(float)(g1 >> 16);
    } else {     // crossCheck: q is also the best query of its best train descriptor
      keep = (bwd[(size_t)(g0 & 0xffffu)].x & 0xffffu) == (unsigned)q;
    }
    if (!keep) continue;
    if (m < cap) { query_idx[m] = q; train_idx[m] = (int)(g0 & 0xffffu); }
    m++;
  }
  *n_matches = m;
  return VLOAM_OK;
}

static vloam_status img_copy_out(const void* d, size_t bytes, void* buf, long long cap, long long* n) {
  if (n) *n = (long long)bytes;
  if (!buf) return VLOAM_OK;
  const size_t m = (size_t)cap < bytes ? (size_t)cap : bytes;
  return hipMemcpy(buf, d, m, hipMemcpyDeviceToHost) == hipSuccess ? VLOAM_OK : VLOAM_ERR_HIP;
}

// items: 0 eig map, 1 + l pyramid level l of the last image, 4 + l its derivative level l, 8 candidate count, 9 error bits,
// 10 k_img_select's shader cycles per phase {init + list prefetch, decision sweeps, collect, sort, output, -, number of sweeps, accepted}
vloam_status img_debug_get(ImgContext* c, int item, void* buf, long long cap, long long* n) {
  if (c->max_w == 0 || c->count < 0) return VLOAM_ERR_ORDER;
  const ImgPyrDev& P = c->pyr[c->count % 2];
  if (item == 0) return img_copy_out(c->eig, sizeof(float) * (size_t)c->w * c->h, buf, cap, n);
  if (item >= 1 && item <= 3) { const int l = item - 1; if (l >= P.levels) return VLOAM_ERR_INVALID; return img_copy_out(P.img[l], (size_t)P.w[l] * P.h[l], buf, cap, n); }
  if (item >= 4 && item <= 6) { const int l = item - 4; if (l >= P.levels) return VLOAM_ERR_INVALID; return img_copy_out(P.deriv[l], sizeof(short2) * (size_t)P.w[l] * P.h[l], buf, cap, n); }
  if (item == 8) return img_copy_out(c->n_cand, sizeof(int), buf, cap, n);
  if (item == 9) return img_copy_out(c->error, sizeof(int), buf, cap, n);
  if (item == 10) return img_copy_out(c->acc + kImgAccCap - 8, 8 * sizeof(u64), buf, cap, n);
  if (item == 11) { if (!c->clahe) return VLOAM_ERR_ORDER; return img_copy_out(c->clahe_img, (size_t)c->w * c->h, buf, cap, n); }   // the equalised image
  return VLOAM_ERR_INVALID;
}

}  // namespace vloam
