// STAND-IN, NOT THE LIBRARY.  Minimal declarations with the member names / signatures the real header gives the types vloam_hip/compat.hpp and
// vloam_hip/factors.hpp are templated over, so that tests/test_cpp_compat_types.py and tests/test_gpu_cpp_boundary.py can instantiate every adapter
// overload (a syntax / overload-resolution check of OUR headers).  It has no numerical role, is not an oracle, and is never used to build the reference.
#pragma once
#include <cstdint>
#include <vector>
#include <boost/shared_ptr.hpp>
namespace pcl {
template <class PointT> class PointCloud {
 public:
  typedef boost::shared_ptr<PointCloud<PointT> > Ptr;
  typedef boost::shared_ptr<const PointCloud<PointT> > ConstPtr;
  std::vector<PointT> points;
  std::uint32_t width = 0, height = 0;
  bool is_dense = true;
  std::size_t size() const { return points.size(); }
};
}  // namespace pcl
