// Host check of csrc/fdlibm_f32.h against the C library's atanf / atan2f, bit for bit (tests/test_fdlibm_f32.py).
// argv[1] = arguments per function.  Prints "atanf <tested> <mismatches>" / "atan2f <tested> <mismatches>" and the first mismatches.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "fdlibm_f32.h"

static uint64_t s = 0x9e3779b97f4a7c15ull;
static inline uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline bool same(float a, float b) { return (std::isnan(a) && std::isnan(b)) || vloam::fd_bits(a) == vloam::fd_bits(b); }

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : (1l << 22);
  long bad1 = 0, n1 = 0, bad2 = 0, n2 = 0;
  auto t1 = [&](float x) {
    volatile float xv = x;   // (keep the library call a call)
    const float a = atanf(xv), b = vloam::fd_atanf(x);
    n1++;
    if (!same(a, b)) { if (bad1 < 8) printf("  atanf(%a) libm %a restated %a\n", x, a, b); bad1++; }
  };
  auto t2 = [&](float y, float x) {
    volatile float yv = y, xv = x;
    const float a = atan2f(yv, xv), b = vloam::fd_atan2f(y, x);
    n2++;
    if (!same(a, b)) { if (bad2 < 8) printf("  atan2f(%a, %a) libm %a restated %a\n", y, x, a, b); bad2++; }
  };
  const float special[] = {0.0f, -0.0f, 1.0f, -1.0f, 0.4375f, 0.6875f, 1.1875f, 2.4375f, 1.5f, 0.5f, INFINITY, -INFINITY, NAN, 1e-30f, -1e-30f, 3.4e38f, -3.4e38f,
                           1.17549435e-38f, 1e-45f, 33554432.0f, 33554430.0f, 1.8626451e-9f, 1.862645e-9f};
  for (float a : special) { t1(a); t1(std::nextafterf(a, 10.f)); t1(std::nextafterf(a, -10.f)); for (float b : special) t2(a, b); }
  // every binade, random mantissas, both signs
  for (long i = 0; i < n; i++) { const uint32_t u = (uint32_t)rnd(); float x; memcpy(&x, &u, 4); t1(x); }
  // the arguments scan registration produces: z / sqrt(x^2 + y^2) of LiDAR returns (|elevation| < 30 deg) and full-circle azimuths
  for (long i = 0; i < n; i++) {
    const float v = (float)((double)(rnd() >> 11) / 9007199254740992.0 * 1.2 - 0.6);
    t1(v);
  }
  for (long i = 0; i < n; i++) {
    const uint32_t u = (uint32_t)rnd(), v = (uint32_t)rnd();
    float y, x; memcpy(&y, &u, 4); memcpy(&x, &v, 4); t2(y, x);
  }
  for (long i = 0; i < n; i++) {
    const float y = (float)((double)(rnd() >> 11) / 9007199254740992.0 * 240.0 - 120.0), x = (float)((double)(rnd() >> 11) / 9007199254740992.0 * 240.0 - 120.0);
    t2(y, x);
  }
  printf("atanf %ld %ld\natan2f %ld %ld\n", n1, bad1, n2, bad2);
  return (bad1 || bad2) ? 1 : 0;
}
