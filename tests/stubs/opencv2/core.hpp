// STAND-IN, NOT THE LIBRARY.  Minimal declarations with the member names / signatures the real header gives the types vloam_hip/compat.hpp and
// vloam_hip/factors.hpp are templated over, so that tests/test_cpp_compat_types.py and tests/test_gpu_cpp_boundary.py can instantiate every adapter
// overload (a syntax / overload-resolution check of OUR headers).  It has no numerical role, is not an oracle, and is never used to build the reference.
#pragma once
#include <cstddef>
namespace cv {
struct MatStep { std::size_t v = 0; operator std::size_t() const { return v; } };   // cv::Mat::step converts to size_t
class Mat {
 public:
  unsigned char* data = nullptr;
  int rows = 0, cols = 0;
  MatStep step;
};
}  // namespace cv
