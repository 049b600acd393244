// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  PARITY UNPINNED.
//
// CPU restatement of the image front-end of the visual odometry in its optical-flow configuration
// (vloam_main.launch:10 optical_flow_match = true):
//   ImageUtil::detKeypoints, DetectorType::ShiTomasi   src/visual_odometry/src/image_util.cpp:13-36
//        cv::goodFeaturesToTrack(img, corners, 1024, 0.03, 7.5, Mat(), 5, false, 0.04)
//   ImageUtil::calculateOpticalFlow                    src/visual_odometry/src/image_util.cpp:351-372
//        cv::calcOpticalFlowPyrLK(image0, image1, pts0, pts1, status, err, Size(15, 15), 2, TermCriteria(COUNT + EPS, 10, 0.03))
//   VisualOdometry::processImage                       src/visual_odometry/src/visual_odometry.cpp:91-132
//        (the corners detected in the CURRENT image are handed to calcOpticalFlowPyrLK as points of the PREVIOUS image, :121-122)
//   the match loop's float -> int truncation           src/visual_odometry/src/visual_odometry.cpp:296-308
//
// The two algorithms live in a third-party dependency that is absent from the reference tree and from this image: OpenCV 4
// (the reference includes <opencv4/opencv2/opencv.hpp>; Ubuntu 20.04 / ROS Noetic ships 4.2.0).  They are restated here from
// OpenCV's published algorithm (modules/imgproc/src/featureselect.cpp + corner.cpp, modules/video/src/lkpyramid.cpp,
// modules/imgproc/src/pyramids.cpp), anchored on the reference's call sites and parameters above.  No OpenCV output is
// available to check against: PARITY UNPINNED.  Where OpenCV's result depends on the build (SIMD paths add floats in a
// different order than the scalar path) the restatement takes an order-independent definition, stated at each place:
//   * cornerMinEigenVal: the structure tensor is summed EXACTLY in integers (Sobel responses of an 8-bit image are integers)
//     and the smaller eigenvalue is evaluated once in f64 and rounded to f32 — within a few f32 ulps of any OpenCV build's
//     f32 running sums (boxFilter's RowSum / ColumnSum), identical on every machine;
//   * Lucas-Kanade: the 2x2 gradient matrix and the mismatch vector are sums of integer products (OpenCV's fixed-point
//     window, W_BITS = 14); they are summed exactly and rounded to f32 once (OpenCV: f32 accumulation, order build-dependent).
// The ORB + brute-force configuration (optical_flow_match = false, the launch file's default: vloam_main.launch:10) is restated too:
// orb_descriptors below (ImageUtil::descKeypoints = cv::ORB::create()->compute on the Shi-Tomasi corners, image_util.cpp:162-212) and
// bf_match_hamming (image_util.cpp:221-296).  ONE input stays outside: OpenCV's learned sampling table (orb.cpp, bit_pattern_31_: 256 x 4
// integers) is data of the library, not an algorithm — it is handed in by the caller (vloam_vo_set_orb_pattern; tests use a seeded one).
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

namespace orc {

struct ImgCorner { float x, y; };

// cv::goodFeaturesToTrack on an 8-bit image (row stride = width).  eig_out (optional): the min-eigenvalue map (f32, w*h).
std::vector<ImgCorner> good_features_to_track(const uint8_t* img, int w, int h, int max_corners, double quality_level, double min_distance,
                                              int block_size, std::vector<float>* eig_out = nullptr);

struct Pyramid {                      // cv::buildOpticalFlowPyramid without the padded borders (border pixels are computed on access)
  std::vector<std::vector<uint8_t>> img;
  std::vector<std::vector<int16_t>> deriv;   // [level][2 * (y * w + x)]: Scharr Ix, Iy (calcSharrDeriv); zero outside the image
  std::vector<int> w, h;
  void build(const uint8_t* src, int width, int height, int win, int max_level);
  int levels() const { return (int)img.size(); }
};

// cv::calcOpticalFlowPyrLK(prev, next, prev_pts -> next_pts, status), winSize win x win, maxLevel from the pyramids, criteria
// (COUNT + EPS, max_count, epsilon), flags = 0, minEigThreshold = 1e-4, err requested (its bounds check clears status too).
void calc_optical_flow_pyr_lk(const Pyramid& prev, const Pyramid& next, const std::vector<ImgCorner>& prev_pts, std::vector<ImgCorner>* next_pts,
                              std::vector<uint8_t>* status, int win, int max_count, double epsilon);

// cv::CLAHE::apply of cv::createCLAHE(clip_limit, Size(tiles, tiles)) on an 8-bit image (imgproc/src/clahe.cpp) — the optional first step of
// VisualOdometry::processImage (visual_odometry.cpp:31,97-100; vloam_main.launch:8 CLAHE = false by default): per-tile histograms (the image
// padded REFLECT_101 to a multiple of the tile grid when it does not divide), clipping with uniform redistribution of the excess and
// the residual spread at a fixed stride, cumulative LUTs scaled by 255 / tile area, bilinear interpolation of the four surrounding
// tiles' LUT values in f32.  out: w * h bytes.
void clahe_apply(const uint8_t* img, int w, int h, double clip_limit, int tiles, uint8_t* out);

// ImageUtil::matchDescriptors with MatcherType::BF on binary descriptors (NORM_HAMMING), image_util.cpp:221-296:
//   knn != 0 (SelectType::KNN, the reference's setting, visual_odometry.cpp:37): BFMatcher::knnMatch(desc0, desc1, 2) and the ratio
//            test  d_best < 0.8 * d_second  (:262-271);
//   knn == 0 (SelectType::NN): BFMatcher(NORM_HAMMING, crossCheck = true)::match (:225,:250).
// cv::batchDistance keeps the K smallest distances per query, admitting a train descriptor only on a strictly smaller distance: equal
// distances resolve to the lower train index (and, for the cross check, to the lower query index).  Returns (queryIdx, trainIdx) pairs
// in query order.  desc: n x bytes, row major.
// cv::GaussianBlur(img, Size(7, 7), 2, 2, BORDER_REFLECT_101) on an 8-bit image — what ORB_Impl::detectAndCompute applies to every pyramid
// level before the descriptors (orb.cpp).  OpenCV >= 4.1 runs 8-bit images through its bit-exact fixed-point path (smooth.dispatch.cpp):
// the kernel in Q8 by error diffusion from the taps outward-in, the centre taking what is left of 256 — {18, 34, 48, 56, 48, 34, 18} for
// sigma 2 (restated from the published source as recalled; PARITY UNPINNED) —, a horizontal pass in Q8.8 (exact: the weights sum to 1), a
// vertical pass in Q16.16, and one rounding (+ 2^15) >> 16.  All integers: no build dependence.
void gaussian_blur_7x7_s2(const uint8_t* img, int w, int h, uint8_t* out);

// ImageUtil::descKeypoints with DescriptorType::ORB (image_util.cpp:178-180,203: cv::ORB::create()->compute(img, keypoints, descriptors)) on
// keypoints that come from goodFeaturesToTrack (octave 0, angle -1 — image_util.cpp:30-34 sets only pt and size):
//   * KeyPointsFilter::runByImageBorder(keypoints, size, edgeThreshold = 31): keypoints outside [31, w - 31) x [31, h - 31) are REMOVED from the
//     caller's vector (descKeypoints takes it by reference: the match indices refer to the filtered list), order kept;
//   * every keypoint has octave 0 -> one pyramid level, the image itself, blurred as above;
//   * the orientation is NOT recomputed for provided keypoints: computeOrbDescriptors steers the pattern by kpt.angle = -1 DEGREE for every
//     keypoint (a = cos, b = sin of -pi / 180 in f32; x' = cvRound(x a - y b), y' = cvRound(x b + y a));
//   * WTA_K = 2: bit k of the 256 = blurred[p0_k] < blurred[p1_k], 32 bytes, bit i of byte j = test 8 j + i.
// pattern: [256][4] = (x0, y0, x1, y1) of test k (OpenCV: bit_pattern_31_).  kept (out): indices into kps of the keypoints that survive the
// border filter; desc (out): 32 bytes each.
void orb_descriptors(const uint8_t* img, int w, int h, const std::vector<ImgCorner>& kps, const int8_t* pattern, std::vector<int>* kept,
                     std::vector<uint8_t>* desc);

std::vector<std::pair<int, int>> bf_match_hamming(const uint8_t* desc0, int n0, const uint8_t* desc1, int n1, int bytes, bool knn);

}  // namespace orc
