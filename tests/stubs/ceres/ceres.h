// STAND-IN, NOT THE LIBRARY.  Minimal declarations with the member names / signatures the real header gives the types vloam_hip/compat.hpp and
// vloam_hip/factors.hpp are templated over, so that tests/test_cpp_compat_types.py and tests/test_gpu_cpp_boundary.py can instantiate every adapter
// overload (a syntax / overload-resolution check of OUR headers).  It has no numerical role, is not an oracle, and is never used to build the reference.
#pragma once
#include <memory>
namespace ceres {
class CostFunction {
 public:
  virtual ~CostFunction() {}
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
};
template <class Functor, int kNumResiduals, int N0, int N1>
class AutoDiffCostFunction : public CostFunction {
 public:
  explicit AutoDiffCostFunction(Functor* f) : f_(f) {}   // takes ownership, like Ceres
  bool Evaluate(double const* const* parameters, double* residuals, double**) const override { return (*f_)(parameters[0], parameters[1], residuals); }   // residuals only
  static constexpr int num_residuals = kNumResiduals, n0 = N0, n1 = N1;
 private:
  std::unique_ptr<Functor> f_;
};
}  // namespace ceres
