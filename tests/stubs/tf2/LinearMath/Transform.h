// STAND-IN, NOT THE LIBRARY.  Minimal declarations with the member names / signatures the real header gives the types vloam_hip/compat.hpp and
// vloam_hip/factors.hpp are templated over, so that tests/test_cpp_compat_types.py and tests/test_gpu_cpp_boundary.py can instantiate every adapter
// overload (a syntax / overload-resolution check of OUR headers).  It has no numerical role, is not an oracle, and is never used to build the reference.
#pragma once
namespace tf2 {
class Vector3 {
 public:
  Vector3() {}
  Vector3(double x, double y, double z) { v_[0] = x; v_[1] = y; v_[2] = z; }
  void setValue(double x, double y, double z) { v_[0] = x; v_[1] = y; v_[2] = z; }
  double getX() const { return v_[0]; } double getY() const { return v_[1]; } double getZ() const { return v_[2]; }
 private:
  double v_[3] = {0, 0, 0};
};
class Quaternion {
 public:
  Quaternion() {}
  void setValue(double x, double y, double z, double w) { v_[0] = x; v_[1] = y; v_[2] = z; v_[3] = w; }
  double getX() const { return v_[0]; } double getY() const { return v_[1]; } double getZ() const { return v_[2]; } double getW() const { return v_[3]; }
 private:
  double v_[4] = {0, 0, 0, 1};
};
class Transform {   // (stores the rotation as the quaternion it was given: enough to read it back)
 public:
  Vector3 getOrigin() const { return o_; }
  void setOrigin(const Vector3& o) { o_ = o; }
  Quaternion getRotation() const { return q_; }
  void setRotation(const Quaternion& q) { q_ = q; }
 private:
  Vector3 o_; Quaternion q_;
};
}  // namespace tf2
