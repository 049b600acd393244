// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.  See orc_img.h for what is restated and from where.
#include "orc_img.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstdint>

namespace orc {

static inline int reflect101(int i, int n) {  // cv::borderInterpolate(i, n, BORDER_REFLECT_101)
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}
static inline int cv_round(float v) { return (int)lrintf(v); }   // cvRound: round half to even
static inline int cv_floor(float v) { return (int)floorf(v); }

// ---------------------------------------------------------------------------------------------------------------------------
// cv::goodFeaturesToTrack (imgproc/src/featureselect.cpp) with useHarrisDetector = false, gradientSize = 3, no mask:
//   cornerMinEigenVal(image, eig, blockSize, 3)   -> Sobel 3x3 (scale 1 / (2^(3-1) * blockSize * 255)), products, blockSize x blockSize
//                                                    unnormalised box sum (BORDER_REFLECT_101 on the source for the Sobel pass and on
//                                                    the product images for the box pass), eig = (a + c) - sqrt((a - c)^2 + b^2) with
//                                                    a = Sxx / 2, b = Sxy, c = Syy / 2                      (corner.cpp calcMinEigenVal)
//   minMaxLoc; threshold(eig, maxVal * qualityLevel, THRESH_TOZERO); dilate 3x3
//   candidates: 1 <= x <= w - 2, 1 <= y <= h - 2 with eig != 0 and eig == dilated
//   sort descending (ties: larger address first — greaterThanPtr)
//   greedy pass with a cell grid of cvRound(minDistance): a candidate is dropped when an accepted corner lies closer than minDistance
std::vector<ImgCorner> good_features_to_track(const uint8_t* img, int w, int h, int max_corners, double quality_level, double min_distance,
                                              int block_size, std::vector<float>* eig_out) {
  std::vector<int> dx((size_t)w * h), dy((size_t)w * h);
  for (int y = 0; y < h; y++) {
    const uint8_t* r0 = img + (size_t)reflect101(y - 1, h) * w;
    const uint8_t* r1 = img + (size_t)y * w;
    const uint8_t* r2 = img + (size_t)reflect101(y + 1, h) * w;
    for (int x = 0; x < w; x++) {
      const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
      dx[(size_t)y * w + x] = (r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
      dy[(size_t)y * w + x] = (r2[xm] - r0[xm]) + 2 * (r2[x] - r0[x]) + (r2[xp] - r0[xp]);
    }
  }
  // EXACT integer structure tensor (see the header: order-independent stand-in for boxFilter's f32 running sums)
  const double scale = 1.0 / (4.0 * (double)block_size * 255.0);
  const double hs2 = 0.5 * scale * scale;
  const int r = block_size / 2;  // anchor at the centre (odd block sizes; the reference uses 5)
  std::vector<float> eig((size_t)w * h);
  float max_val = 0.f;  // eig >= 0 up to rounding; an image without any positive response yields no corners
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      long long sxx = 0, sxy = 0, syy = 0;
      for (int j = -r; j < block_size - r; j++) {
        const int yy = reflect101(y + j, h);
        for (int i = -r; i < block_size - r; i++) {
          const int xx = reflect101(x + i, w);
          const long long gx = dx[(size_t)yy * w + xx], gy = dy[(size_t)yy * w + xx];
          sxx += gx * gx; sxy += gx * gy; syy += gy * gy;
        }
      }
      const long long d = sxx - syy;
      const double root = std::sqrt((double)(d * d + 4 * sxy * sxy));
      const float e = (float)(((double)(sxx + syy) - root) * hs2);
      eig[(size_t)y * w + x] = e;
      if (e > max_val) max_val = e;
    }
  if (eig_out) *eig_out = eig;
  const float thr = (float)((double)max_val * quality_level);
  auto tz = [&](int x, int y) { const float v = eig[(size_t)y * w + x]; return v > thr ? v : 0.f; };  // THRESH_TOZERO
  std::vector<int> cand;  // addresses y * w + x
  for (int y = 1; y < h - 1; y++)
    for (int x = 1; x < w - 1; x++) {
      const float v = tz(x, y);
      if (v == 0.f) continue;
      float m = v;  // dilate 3x3 (pixels outside the image do not take part)
      for (int j = -1; j <= 1; j++)
        for (int i = -1; i <= 1; i++) m = std::max(m, tz(x + i, y + j));   // inside: 1 <= x <= w - 2
      if (v == m) cand.push_back(y * w + x);
    }
  std::sort(cand.begin(), cand.end(), [&](int a, int b) {
    const float va = eig[a], vb = eig[b];
    return va > vb ? true : (va < vb ? false : a > b);
  });
  std::vector<ImgCorner> corners;
  if (min_distance >= 1) {
    const int cell = cv_round((float)min_distance);  // cvRound(double) in OpenCV; 7.5 -> 8 either way
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    std::vector<std::vector<ImgCorner>> grid((size_t)gw * gh);
    const float md2 = (float)(min_distance * min_distance);
    for (int a : cand) {
      const int y = a / w, x = a - y * w;
      const int xc = x / cell, yc = y / cell;
      const int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1), x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
      bool good = true;
      for (int yy = y1; yy <= y2 && good; yy++)
        for (int xx = x1; xx <= x2 && good; xx++)
          for (const ImgCorner& m : grid[(size_t)yy * gw + xx]) {
            const float ddx = (float)x - m.x, ddy = (float)y - m.y;
            if (ddx * ddx + ddy * ddy < md2) { good = false; break; }
          }
      if (good) {
        grid[(size_t)yc * gw + xc].push_back(ImgCorner{(float)x, (float)y});
        corners.push_back(ImgCorner{(float)x, (float)y});
        if (max_corners > 0 && (int)corners.size() == max_corners) break;
      }
    }
  } else {
    for (int a : cand) {
      corners.push_back(ImgCorner{(float)(a % w), (float)(a / w)});
      if (max_corners > 0 && (int)corners.size() == max_corners) break;
    }
  }
  return corners;
}

// ---------------------------------------------------------------------------------------------------------------------------
// cv::buildOpticalFlowPyramid (video/src/lkpyramid.cpp): level 0 is the image, level l + 1 = cv::pyrDown(level l) (5-tap binomial
// [1 4 6 4 1] / 16 in both directions, exact integers, (sum + 128) >> 8, BORDER_REFLECT_101, size ((w + 1) / 2, (h + 1) / 2)); the
// pyramid stops early when the next level would not be larger than the window.  calcSharrDeriv per level.
void Pyramid::build(const uint8_t* src, int width, int height, int win, int max_level) {
  img.clear(); deriv.clear(); w.clear(); h.clear();
  img.emplace_back(src, src + (size_t)width * height);
  w.push_back(width); h.push_back(height);
  for (int level = 0;; level++) {
    const int cw = w[level], ch = h[level];
    const std::vector<uint8_t>& I = img[level];
    // calcSharrDeriv: t0 = 3 (s0 + s2) + 10 s1, t1 = s2 - s0 over rows y - 1, y, y + 1;  Ix = t0[x + 1] - t0[x - 1],
    // Iy = 3 (t1[x + 1] + t1[x - 1]) + 10 t1[x]; rows / columns beyond the image: reflect-101
    std::vector<int16_t> D((size_t)cw * ch * 2);
    for (int y = 0; y < ch; y++) {
      const uint8_t* s0 = I.data() + (size_t)reflect101(y - 1, ch) * cw;
      const uint8_t* s1 = I.data() + (size_t)y * cw;
      const uint8_t* s2 = I.data() + (size_t)reflect101(y + 1, ch) * cw;
      for (int x = 0; x < cw; x++) {
        const int xm = reflect101(x - 1, cw), xp = reflect101(x + 1, cw);
        const int t0m = (s0[xm] + s2[xm]) * 3 + s1[xm] * 10, t0p = (s0[xp] + s2[xp]) * 3 + s1[xp] * 10;
        const int t1m = s2[xm] - s0[xm], t1c = s2[x] - s0[x], t1p = s2[xp] - s0[xp];
        D[2 * ((size_t)y * cw + x)] = (int16_t)(t0p - t0m);
        D[2 * ((size_t)y * cw + x) + 1] = (int16_t)((t1p + t1m) * 3 + t1c * 10);
      }
    }
    deriv.push_back(std::move(D));
    const int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
    if (level >= max_level || nw <= win || nh <= win) break;
    std::vector<uint8_t> N((size_t)nw * nh);
    static const int k[5] = {1, 4, 6, 4, 1};
    for (int y = 0; y < nh; y++)
      for (int x = 0; x < nw; x++) {
        int s = 0;
        for (int j = 0; j < 5; j++) {
          const uint8_t* row = I.data() + (size_t)reflect101(2 * y + j - 2, ch) * cw;
          int rs = 0;
          for (int i = 0; i < 5; i++) rs += k[i] * row[reflect101(2 * x + i - 2, cw)];
          s += k[j] * rs;
        }
        N[(size_t)y * nw + x] = (uint8_t)((s + 128) >> 8);
      }
    img.push_back(std::move(N));
    w.push_back(nw); h.push_back(nh);
  }
}

// cv::calcOpticalFlowPyrLK -> LKTrackerInvoker::operator() (video/src/lkpyramid.cpp), one point at a time, levels from the top.
// The pyramid levels carry a REFLECT_101 border of winSize pixels, the derivative images a ZERO border of the same width.
void calc_optical_flow_pyr_lk(const Pyramid& P, const Pyramid& N, const std::vector<ImgCorner>& prev_pts, std::vector<ImgCorner>* next_pts,
                              std::vector<uint8_t>* status, int win, int max_count, double epsilon) {
  const int npt = (int)prev_pts.size();
  next_pts->assign((size_t)npt, ImgCorner{0.f, 0.f});
  status->assign((size_t)npt, 1);
  max_count = std::min(std::max(max_count, 0), 100);
  epsilon = std::min(std::max(epsilon, 0.), 10.);
  const double eps2 = epsilon * epsilon;
  const int max_level = std::min(P.levels(), N.levels()) - 1;
  const int W_BITS = 14;
  const float FLT_SCALE = 1.f / (1 << 20);
  const double min_eig_threshold = 1e-4;
  const float half = (win - 1) * 0.5f;
  std::vector<int> Iw((size_t)win * win), Ix((size_t)win * win), Iy((size_t)win * win);
  auto descale = [](int v, int n) { return (v + (1 << (n - 1))) >> n; };
  for (int level = max_level; level >= 0; level--) {
    const int cw = P.w[level], ch = P.h[level];
    const uint8_t* I = P.img[level].data();
    const uint8_t* J = N.img[level].data();
    const int16_t* dI = P.deriv[level].data();
    auto pix = [&](const uint8_t* im, int x, int y) { return (int)im[(size_t)reflect101(y, ch) * cw + reflect101(x, cw)]; };
    auto der = [&](int x, int y, int c) { return (x < 0 || x >= cw || y < 0 || y >= ch) ? 0 : (int)dI[2 * ((size_t)y * cw + x) + c]; };
    for (int p = 0; p < npt; p++) {
      float px = prev_pts[p].x * (float)(1. / (1 << level)), py = prev_pts[p].y * (float)(1. / (1 << level));
      float nx, ny;
      if (level == max_level) { nx = px; ny = py; }
      else { nx = (*next_pts)[p].x * 2.f; ny = (*next_pts)[p].y * 2.f; }
      (*next_pts)[p] = ImgCorner{nx, ny};
      px -= half; py -= half;
      const int ipx = cv_floor(px), ipy = cv_floor(py);
      if (ipx < -win || ipx >= cw || ipy < -win || ipy >= ch) {
        if (level == 0) (*status)[p] = 0;
        continue;
      }
      float a = px - ipx, b = py - ipy;
      int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
      int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
      int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
      int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      long long sA11 = 0, sA12 = 0, sA22 = 0;  // EXACT (see the header: OpenCV accumulates these integer products in f32)
      for (int y = 0; y < win; y++)
        for (int x = 0; x < win; x++) {
          const int X = ipx + x, Y = ipy + y;
          const int ival = descale(pix(I, X, Y) * iw00 + pix(I, X + 1, Y) * iw01 + pix(I, X, Y + 1) * iw10 + pix(I, X + 1, Y + 1) * iw11, W_BITS - 5);
          const int ixv = descale(der(X, Y, 0) * iw00 + der(X + 1, Y, 0) * iw01 + der(X, Y + 1, 0) * iw10 + der(X + 1, Y + 1, 0) * iw11, W_BITS);
          const int iyv = descale(der(X, Y, 1) * iw00 + der(X + 1, Y, 1) * iw01 + der(X, Y + 1, 1) * iw10 + der(X + 1, Y + 1, 1) * iw11, W_BITS);
          Iw[(size_t)y * win + x] = ival; Ix[(size_t)y * win + x] = ixv; Iy[(size_t)y * win + x] = iyv;
          sA11 += (long long)ixv * ixv; sA12 += (long long)ixv * iyv; sA22 += (long long)iyv * iyv;
        }
      const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
      float D = A11 * A22 - A12 * A12;
      const float min_eig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
      if ((double)min_eig < min_eig_threshold || D < FLT_EPSILON) {
        if (level == 0) (*status)[p] = 0;
        continue;
      }
      D = 1.f / D;
      nx -= half; ny -= half;
      float pdx = 0.f, pdy = 0.f;
      for (int j = 0; j < max_count; j++) {
        const int inx = cv_floor(nx), iny = cv_floor(ny);
        if (inx < -win || inx >= cw || iny < -win || iny >= ch) {
          if (level == 0) (*status)[p] = 0;
          break;
        }
        a = nx - inx; b = ny - iny;
        iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
        iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
        iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        long long sb1 = 0, sb2 = 0;
        for (int y = 0; y < win; y++)
          for (int x = 0; x < win; x++) {
            const int X = inx + x, Y = iny + y;
            const int diff = descale(pix(J, X, Y) * iw00 + pix(J, X + 1, Y) * iw01 + pix(J, X, Y + 1) * iw10 + pix(J, X + 1, Y + 1) * iw11, W_BITS - 5) -
                             Iw[(size_t)y * win + x];
            sb1 += (long long)diff * Ix[(size_t)y * win + x];
            sb2 += (long long)diff * Iy[(size_t)y * win + x];
          }
        const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
        const float ddx = (A12 * b2 - A22 * b1) * D, ddy = (A12 * b1 - A11 * b2) * D;
        nx += ddx; ny += ddy;
        (*next_pts)[p] = ImgCorner{nx + half, ny + half};
        if ((double)ddx * ddx + (double)ddy * ddy <= eps2) break;
        if (j > 0 && std::abs(ddx + pdx) < 0.01 && std::abs(ddy + pdy) < 0.01) {
          (*next_pts)[p].x -= ddx * 0.5f;
          (*next_pts)[p].y -= ddy * 0.5f;
          break;
        }
        pdx = ddx; pdy = ddy;
      }
      // the error measure is requested by the reference (it passes `err`): its window check clears the status of a point that left
      // the image on its last update
      if ((*status)[p] && level == 0) {
        const int ix = cv_floor((*next_pts)[p].x - half), iy = cv_floor((*next_pts)[p].y - half);
        if (ix < -win || ix >= cw || iy < -win || iy >= ch) (*status)[p] = 0;
      }
    }
  }
}

void clahe_apply(const uint8_t* img, int w, int h, double clip_limit, int tiles, uint8_t* out) {
  const int hist_size = 256;
  // the LUTs are computed on an image padded to a multiple of the tile grid (bottom / right, REFLECT_101): note OpenCV pads by
  // tiles - (size % tiles), i.e. only when the size does not divide
  const int ew = (w % tiles == 0 && h % tiles == 0) ? w : w + (tiles - w % tiles);
  const int eh = (w % tiles == 0 && h % tiles == 0) ? h : h + (tiles - h % tiles);
  const int tw = ew / tiles, th = eh / tiles;
  const int tile_area = tw * th;
  const float lut_scale = (float)(hist_size - 1) / (float)tile_area;
  int clip = 0;
  if (clip_limit > 0.0) { clip = (int)(clip_limit * tile_area / hist_size); clip = std::max(clip, 1); }
  std::vector<uint8_t> lut((size_t)tiles * tiles * hist_size);
  for (int ty = 0; ty < tiles; ty++)
    for (int tx = 0; tx < tiles; tx++) {
      int hist[256] = {0};
      for (int y = ty * th; y < (ty + 1) * th; y++)
        for (int x = tx * tw; x < (tx + 1) * tw; x++) hist[img[(size_t)reflect101(y, h) * w + reflect101(x, w)]]++;
      if (clip > 0) {
        int clipped = 0;
        for (int i = 0; i < hist_size; i++)
          if (hist[i] > clip) { clipped += hist[i] - clip; hist[i] = clip; }
        const int batch = clipped / hist_size;
        int residual = clipped - batch * hist_size;
        for (int i = 0; i < hist_size; i++) hist[i] += batch;
        if (residual != 0) {
          const int step = std::max(hist_size / residual, 1);
          for (int i = 0; i < hist_size && residual > 0; i += step, residual--) hist[i]++;
        }
      }
      int sum = 0;
      uint8_t* L = lut.data() + (size_t)(ty * tiles + tx) * hist_size;
      for (int i = 0; i < hist_size; i++) {
        sum += hist[i];
        const int v = cv_round((float)sum * lut_scale);   // saturate_cast<uchar>(float)
        L[i] = (uint8_t)std::min(std::max(v, 0), 255);
      }
    }
  const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
  for (int y = 0; y < h; y++) {
    const float tyf = (float)y * inv_th - 0.5f;
    int ty1 = cv_floor(tyf), ty2 = ty1 + 1;
    const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
    ty1 = std::max(ty1, 0); ty2 = std::min(ty2, tiles - 1);
    for (int x = 0; x < w; x++) {
      const float txf = (float)x * inv_tw - 0.5f;
      int tx1 = cv_floor(txf), tx2 = tx1 + 1;
      const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
      tx1 = std::max(tx1, 0); tx2 = std::min(tx2, tiles - 1);
      const int v = img[(size_t)y * w + x];
      const float l11 = lut[(size_t)(ty1 * tiles + tx1) * hist_size + v], l12 = lut[(size_t)(ty1 * tiles + tx2) * hist_size + v];
      const float l21 = lut[(size_t)(ty2 * tiles + tx1) * hist_size + v], l22 = lut[(size_t)(ty2 * tiles + tx2) * hist_size + v];
      const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
      out[(size_t)y * w + x] = (uint8_t)std::min(std::max(cv_round(res), 0), 255);
    }
  }
}

static inline int hamming(const uint8_t* a, const uint8_t* b, int bytes) {
  int d = 0;
  for (int k = 0; k < bytes; k++) d += __builtin_popcount((unsigned)(a[k] ^ b[k]));
  return d;
}

// ---------------------------------------------------------------- ORB descriptors on provided keypoints (see orc_img.h)
void gaussian_blur_7x7_s2(const uint8_t* img, int w, int h, uint8_t* out) {
  static const int K[7] = {18, 34, 48, 56, 48, 34, 18};   // Q8, sums to 256
  std::vector<uint16_t> row((size_t)w * h);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int s = 0;
      for (int k = -3; k <= 3; k++) s += K[k + 3] * (int)img[(size_t)y * w + reflect101(x + k, w)];
      row[(size_t)y * w + x] = (uint16_t)s;   // <= 255 * 256
    }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      uint32_t s = 0;
      for (int k = -3; k <= 3; k++) s += (uint32_t)K[k + 3] * (uint32_t)row[(size_t)reflect101(y + k, h) * w + x];
      out[(size_t)y * w + x] = (uint8_t)((s + (1u << 15)) >> 16);
    }
}
static inline int cv_round_f(float v) { return (int)std::nearbyintf(v); }   // cvRound: round half to even (default rounding mode)
void orb_descriptors(const uint8_t* img, int w, int h, const std::vector<ImgCorner>& kps, const int8_t* pattern, std::vector<int>* kept,
                     std::vector<uint8_t>* desc) {
  kept->clear(); desc->clear();
  const int edge = 31;
  std::vector<uint8_t> blur((size_t)w * h);
  gaussian_blur_7x7_s2(img, w, h, blur.data());
  const float angle = -1.0f * (float)(3.14159265358979323846 / 180.0);   // kpt.angle (-1) * (float)(CV_PI / 180.f)
  const float a = (float)std::cos(angle), b = (float)std::sin(angle);
  int ox[512], oy[512];
  for (int i = 0; i < 512; i++) {
    const float px = (float)pattern[2 * i], py = (float)pattern[2 * i + 1];
    ox[i] = cv_round_f(px * a - py * b);
    oy[i] = cv_round_f(px * b + py * a);
  }
  for (int i = 0; i < (int)kps.size(); i++) {
    const float x = kps[i].x, y = kps[i].y;
    if (!(x >= (float)edge && x < (float)(w - edge) && y >= (float)edge && y < (float)(h - edge))) continue;   // Rect(edge, edge, w - 2 edge, h - 2 edge).contains(pt)
    kept->push_back(i);
    const int cx = cv_round_f(x), cy = cv_round_f(y);
    for (int j = 0; j < 32; j++) {
      int val = 0;
      for (int bit = 0; bit < 8; bit++) {
        const int t = 8 * j + bit;
        const int v0 = blur[(size_t)(cy + oy[2 * t]) * w + (cx + ox[2 * t])], v1 = blur[(size_t)(cy + oy[2 * t + 1]) * w + (cx + ox[2 * t + 1])];
        val |= (v0 < v1) << bit;
      }
      desc->push_back((uint8_t)val);
    }
  }
}

std::vector<std::pair<int, int>> bf_match_hamming(const uint8_t* desc0, int n0, const uint8_t* desc1, int n1, int bytes, bool knn) {
  std::vector<std::pair<int, int>> out;
  if (knn) {
    for (int q = 0; q < n0; q++) {
      int d0 = INT32_MAX, t0 = -1, d1 = INT32_MAX, t1 = -1;   // the two smallest, ties to the lower train index
      for (int t = 0; t < n1; t++) {
        const int d = hamming(desc0 + (size_t)q * bytes, desc1 + (size_t)t * bytes, bytes);
        if (d < d1) {
          if (d < d0) { d1 = d0; t1 = t0; d0 = d; t0 = t; }
          else { d1 = d; t1 = t; }
        }
      }
      if (t1 < 0) continue;   // fewer than two train descriptors: the reference would index knn_match[1] out of range
      if ((double)(float)d0 < 0.8 * (double)(float)d1) out.emplace_back(q, t0);
    }
  } else {
    std::vector<int> bq((size_t)n0, -1), bt((size_t)n1, -1), dq((size_t)n0, INT32_MAX), dt((size_t)n1, INT32_MAX);
    for (int q = 0; q < n0; q++)
      for (int t = 0; t < n1; t++) {
        const int d = hamming(desc0 + (size_t)q * bytes, desc1 + (size_t)t * bytes, bytes);
        if (d < dq[q]) { dq[q] = d; bq[q] = t; }
        if (d < dt[t]) { dt[t] = d; bt[t] = q; }
      }
    for (int q = 0; q < n0; q++)
      if (bq[q] >= 0 && bt[bq[q]] == q) out.emplace_back(q, bq[q]);
  }
  return out;
}

}  // namespace orc
