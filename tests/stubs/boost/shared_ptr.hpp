// STAND-IN, NOT THE LIBRARY.  Minimal declarations with the member names / signatures the real header gives the types vloam_hip/compat.hpp and
// vloam_hip/factors.hpp are templated over, so that tests/test_cpp_compat_types.py and tests/test_gpu_cpp_boundary.py can instantiate every adapter
// overload (a syntax / overload-resolution check of OUR headers).  It has no numerical role, is not an oracle, and is never used to build the reference.
#pragma once
#include <memory>
#include <utility>
namespace boost {
template <class T> using shared_ptr = std::shared_ptr<T>;
template <class T, class... A> inline shared_ptr<T> make_shared(A&&... a) { return std::make_shared<T>(std::forward<A>(a)...); }
}  // namespace boost
