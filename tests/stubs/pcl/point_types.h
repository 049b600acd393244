// STAND-IN, NOT THE LIBRARY.  Minimal declarations with the member names / signatures the real header gives the types vloam_hip/compat.hpp and
// vloam_hip/factors.hpp are templated over, so that tests/test_cpp_compat_types.py and tests/test_gpu_cpp_boundary.py can instantiate every adapter
// overload (a syntax / overload-resolution check of OUR headers).  It has no numerical role, is not an oracle, and is never used to build the reference.
#pragma once
namespace pcl {
struct PointXYZ { float x, y, z, data_pad; };                                   // 16 bytes like PCL's (x, y, z, padding)
struct PointXYZI { float x, y, z, data_pad; float intensity, pad_[3]; };        // 32 bytes like PCL's
}  // namespace pcl
