// Image front-end of the visual odometry on gfx950 — host-visible interface (img_kernels.hip).
// Restates VisualOdometry::processImage in its optical-flow configuration (optical_flow_match = true):
//   ImageUtil::detKeypoints, ShiTomasi   /root/reference/src/visual_odometry/src/image_util.cpp:13-36   cv::goodFeaturesToTrack
//   ImageUtil::calculateOpticalFlow      /root/reference/src/visual_odometry/src/image_util.cpp:351-372 cv::calcOpticalFlowPyrLK
//   VisualOdometry::processImage         /root/reference/src/visual_odometry/src/visual_odometry.cpp:91-132
//   the match loop's float -> int reads  /root/reference/src/visual_odometry/src/visual_odometry.cpp:296-308
// OpenCV 4 itself is not part of the reference tree: the two algorithms follow its published sources (featureselect.cpp, corner.cpp,
// lkpyramid.cpp, pyramids.cpp) with the order-independent sums stated in oracle/orc_img.h.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/vloam_hip/c_api.h"
#include "vloam_device.h"

namespace vloam {

constexpr int kImgMaxCorners = 1024;   // image_util.cpp:23   maxCorners
constexpr int kImgBlock = 5;           // image_util.cpp:17   block_size
constexpr int kImgWin = 15;            // image_util.cpp:364  winSize
constexpr int kImgMaxLevel = 2;        // image_util.cpp:364  maxLevel
constexpr int kImgLkIters = 10;        // image_util.cpp:362  TermCriteria count
constexpr int kImgLevels = kImgMaxLevel + 1;
constexpr int kImgCandCap = 65536;     // local maxima above the quality threshold (one status byte each in the selection kernel's LDS)
constexpr int kImgNbrCap = 176;        // stronger candidates within minDistance of a candidate = EVERY pixel offset inside the 7.5 px circle (OpenCV's local-maximum
                                       // test is val == dilate(val): on a plateau of the eigenvalue map adjacent pixels are all candidates; rounds 2 - 5 held 64)
constexpr int kImgAccCap = 16384;      // corners before the maxCorners cut (a 1242 x 375 image holds < 10 600 at minDistance 7.5)
constexpr int kImgMaxRadius = 8;       // floor(minDistance) the neighbourhood scan supports
constexpr int kImgClaheTiles = 8;       // cv::createCLAHE default tileGridSize (8, 8); clipLimit 2.0 (visual_odometry.cpp:31)
constexpr int kImgMaxDesc = 8192;      // descriptors per image the brute-force matcher takes
constexpr int kImgMaxDescBytes = 64;   // ORB / BRISK: 32 / 64 bytes

struct ImgPyrDev {
  unsigned char* img[kImgLevels];
  short2* deriv[kImgLevels];   // Scharr (Ix, Iy), calcSharrDeriv
  int w[kImgLevels], h[kImgLevels];
  int levels;
};

struct ImgContext {
  int max_w = 0, max_h = 0;    // 0: no image front-end in this handle
  int w = 0, h = 0;            // size of the images seen so far (all images of a sequence share it)
  int count = -1;              // VisualOdometry::reset(): ++count; i = count % 2
  ImgPyrDev pyr[2];
  short2* sobel = nullptr;     // [w * h] Sobel (dx, dy)
  float* eig = nullptr;        // [w * h] cornerMinEigenVal
  unsigned* maxbits = nullptr; // [1] bits of the largest eigenvalue
  int* cmap = nullptr;         // [w * h] candidate index of a pixel or -1
  int* clist = nullptr;        // [kImgCandCap] pixel address of candidate c
  int* n_cand = nullptr;       // [1]
  int* nbr = nullptr;          // [kImgCandCap][kImgNbrCap] stronger candidates closer than minDistance
  unsigned char* nbr_cnt = nullptr;
  unsigned long long* acc = nullptr;  // [kImgAccCap] accepted corners, (eig bits, address) keys
  float2* corners[2] = {nullptr, nullptr};   // [kImgMaxCorners] per image, acceptance order
  int* n_corners[2] = {nullptr, nullptr};
  float2* tracked = nullptr;   // [kImgMaxCorners] calcOpticalFlowPyrLK's nextPts
  unsigned char* status = nullptr;
  int* error = nullptr;        // sticky capacity bits
  unsigned char* staging = nullptr;   // [max_w * max_h] upload buffer of the host-pointer entry
  bool clahe = false;          // cfg.CLAHE
  unsigned char* clahe_img = nullptr;   // [max_w * max_h] equalised image
  unsigned char* clahe_lut = nullptr;   // [tiles^2][256]
  // ---- ORB + brute-force configuration (optical_flow_match = false): active once the caller has handed in OpenCV's sampling pattern
  bool orb = false;
  short2* orb_off = nullptr;   // [512] the pattern steered by the provided keypoints' angle (-1 degree) and rounded: pixel offsets of tests' two points
  unsigned char* blur = nullptr;   // [max_w * max_h] GaussianBlur(7 x 7, sigma 2) of the current image
  float2* okp[2] = {nullptr, nullptr};   // [kImgMaxCorners] per image: the corners that survive ORB's border filter, order kept (what the match indices refer to)
  int* n_okp[2] = {nullptr, nullptr};
  unsigned* desc[2] = {nullptr, nullptr};   // [kImgMaxDesc][kImgMaxDescBytes / 4] descriptors of the two images (brute-force matcher)
  uint2* best2[2] = {nullptr, nullptr};     // [kImgMaxDesc] per descriptor: the two smallest (distance << 16 | index) keys against the other set

  // The same buffers of another session of a batched handle: every session's arena has the layout of session 0, `off` bytes further on.
  // The host-side fields (image size, image count) are common to all sessions of a handle: they advance together.
  ImgContext rebased(size_t off) const {
    ImgContext c = *this;
    if (off == 0) return c;
    auto mv = [off](auto*& p) { if (p) p = (typename std::remove_reference<decltype(p)>::type)((char*)p + off); };
    for (int k = 0; k < 2; k++) {
      for (int l = 0; l < kImgLevels; l++) { mv(c.pyr[k].img[l]); mv(c.pyr[k].deriv[l]); }
      mv(c.corners[k]); mv(c.n_corners[k]); mv(c.desc[k]); mv(c.best2[k]); mv(c.okp[k]); mv(c.n_okp[k]);
    }
    mv(c.sobel); mv(c.eig); mv(c.maxbits); mv(c.cmap); mv(c.clist); mv(c.n_cand); mv(c.nbr); mv(c.nbr_cnt); mv(c.acc); mv(c.tracked); mv(c.status);
    mv(c.error); mv(c.staging); mv(c.clahe_img); mv(c.clahe_lut); mv(c.orb_off); mv(c.blur);
    return c;
  }
  // img_process advanced a rebased copy: take over its host-side state (image size / count, pyramid level dimensions), not its pointers
  void adopt_host_state(const ImgContext& o) {
    w = o.w; h = o.h; count = o.count;
    for (int k = 0; k < 2; k++) {
      pyr[k].levels = o.pyr[k].levels;
      for (int l = 0; l < kImgLevels; l++) { pyr[k].w[l] = o.pyr[k].w[l]; pyr[k].h[l] = o.pyr[k].h[l]; }
    }
  }
};

constexpr int kErrImgCandidates = 1, kErrImgNeighbours = 2, kErrImgAccepted = 4;

hipError_t img_init();   // once per device a handle with an image front-end is created on
vloam_status img_layout(ImgContext* c, const vloam_config& cfg, Arena& A);   // session 0's pointers; ImgContext::rebased() gives the other sessions'
// processImage for the image in d_gray (device, row stride in bytes): pyramid + derivatives, corners, and — from the second image
// on — the flow of the new corners from the previous image into this one.  prev_uv / curr_uv (device, [kImgMaxCorners][2] ints, may be
// null): the match loop's integer pixel pairs, x = INT_MIN in entries without a tracked corner.
vloam_status img_check(const ImgContext* c, int width, int height, int stride);   // what img_process would refuse, without touching anything
vloam_status img_process(ImgContext* c, hipStream_t st, const unsigned char* d_gray, int width, int height, int stride, int* prev_uv, int* curr_uv,
                         ProfHook* ph);
vloam_status img_debug_get(ImgContext* c, int item, void* buf, long long cap, long long* n);
// ImageUtil::descKeypoints with ORB (image_util.cpp:162-212) needs OpenCV's learned sampling pattern (orb.cpp: bit_pattern_31_, 256 tests x (x0, y0,
// x1, y1)): third-party DATA that is not in the reference tree.  Handing it in switches the handle's image front-end to the ORB + brute-force
// configuration (optical_flow_match = false); null switches back to optical flow.  n_sessions: every session's arena gets the steered offsets.
vloam_status img_set_orb_pattern(ImgContext* c, hipStream_t st, const signed char* pattern_256x4, int n_sessions, size_t ss);
// ImageUtil::matchDescriptors, MatcherType::BF + NORM_HAMMING (image_util.cpp:221-296): descriptors in HOST memory (n x bytes), matches
// (queryIdx into desc0, trainIdx into desc1) in query order; knn: 2-NN + ratio 0.8 (SelectType::KNN), else NN with cross check.
vloam_status img_match_descriptors(ImgContext* c, hipStream_t st, const unsigned char* desc0, int n0, const unsigned char* desc1, int n1, int bytes, bool knn,
                                   int* query_idx, int* train_idx, int cap, int* n_matches);

}  // namespace vloam
