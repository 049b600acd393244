// atanf / atan2f as the reference's platform computes them.
//
// scan_registration.cpp:166-167,192,234 call the float overloads of atan / atan2 (scan_registration.h:57 `using std::atan2`, <cmath>), i.e.
// glibc's atanf / atan2f.  The scan line of a return is a TRUNCATION of that elevation (scan_registration.cpp:195-226) and the relative time
// of a return goes through +-pi comparisons of that azimuth (:237-261), so a library whose result differs in the last bit moves a return
// that sits on a bin edge into the neighbouring scan line — every later index of the sweep shifts.  OCML's atanf / atan2f are a different
// algorithm (1 - 2 ulp apart from glibc on ~10 % of the arguments): rounds 1 - 5 carried that as a stated tolerance and only ever tested
// elevations in the middle of a bin; round 6's random range images (tests/test_gpu_fuzz.py) put returns on the edges.
//
// glibc <= 2.40 (the reference's ROS Noetic / Ubuntu 20.04 platform has 2.31, this image 2.35) builds both functions from the FreeBSD /
// Sun fdlibm float sources, sysdeps/ieee754/flt-32/s_atanf.c and e_atan2f.c: argument reduction to |x| < 7/16 with four break points, an
// 11-term odd polynomial split into two Horner chains, high / low parts of the break-point arctangents — plain f32 operations in a fixed
// order, no fused multiply-adds on the x86-64 baseline.  Restated below operation for operation (constants by their bit patterns); the
// translation unit is built with -ffp-contract=off, so the device rounds every product and sum exactly like that build.
// tests/test_fdlibm_f32.py compiles this header for the host and compares it with the C library's atanf / atan2f bit for bit on 2^26
// arguments per function (all binades, the break points, signed zeros, infinities, NaN).
//
//   Original notice of the restated sources:  Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.  Developed at SunPro, a
//   Sun Microsystems, Inc. business.  Permission to use, copy, modify, and distribute this software is freely granted, provided that this
//   notice is preserved.  (Conversion to float by Ian Lance Taylor, Cygnus Support.)
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define VL_FD_HD __host__ __device__ __forceinline__
#else
#define VL_FD_HD inline
#endif

namespace vloam {

VL_FD_HD float fd_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
VL_FD_HD uint32_t fd_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

VL_FD_HD float fd_atanf(float x) {
  // atan(0.5), atan(1), atan(1.5), atan(inf): high and low parts
  const uint32_t hi_[4] = {0x3eed6338u, 0x3f490fdau, 0x3f7b985eu, 0x3fc90fdau};
  const uint32_t lo_[4] = {0x31ac3769u, 0x33222168u, 0x33140fb4u, 0x33a22168u};
  const float aT0 = fd_from_bits(0x3eaaaaabu), aT1 = fd_from_bits(0xbe4ccccdu), aT2 = fd_from_bits(0x3e124925u), aT3 = fd_from_bits(0xbde38e38u),
              aT4 = fd_from_bits(0x3dba2e6eu), aT5 = fd_from_bits(0xbd9d8795u), aT6 = fd_from_bits(0x3d886b35u), aT7 = fd_from_bits(0xbd6ef16bu),
              aT8 = fd_from_bits(0x3d4bda59u), aT9 = fd_from_bits(0xbd15a221u), aT10 = fd_from_bits(0x3c8569d7u);
  const int32_t hx = (int32_t)fd_bits(x);
  const int32_t ix = hx & 0x7fffffff;
  int id;
  if (ix >= 0x4c000000) {  // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;  // NaN
    const float r = fd_from_bits(hi_[3]) + fd_from_bits(lo_[3]);
    return hx > 0 ? r : -fd_from_bits(hi_[3]) - fd_from_bits(lo_[3]);
  }
  if (ix < 0x3ee00000) {   // |x| < 0.4375
    if (ix < 0x31000000) return x;  // |x| < 2^-29
    id = -1;
  } else {
    x = fd_from_bits((uint32_t)ix);   // fabsf
    if (ix < 0x3f980000) {            // |x| < 1.1875
      if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }   // 7/16 <= |x| < 11/16
      else { id = 1; x = (x - 1.0f) / (x + 1.0f); }                           // 11/16 <= |x| < 19/16
    } else {
      if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }   // |x| < 2.4375
      else { id = 3; x = -1.0f / x; }                                        // 2.4375 <= |x| < 2^25
    }
  }
  const float z = x * x;
  const float w = z * z;
  // the sum over aT[i] z^(i+1) as an odd and an even Horner chain
  const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (id < 0) return x - x * (s1 + s2);
  const float hi = fd_from_bits(id == 0 ? hi_[0] : (id == 1 ? hi_[1] : (id == 2 ? hi_[2] : hi_[3])));
  const float lo = fd_from_bits(id == 0 ? lo_[0] : (id == 1 ? lo_[1] : (id == 2 ? lo_[2] : lo_[3])));
  const float r = hi - ((x * (s1 + s2) - lo) - x);
  return hx < 0 ? -r : r;
}

VL_FD_HD float fd_atan2f(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = fd_from_bits(0x3f490fdbu), pi_o_2 = fd_from_bits(0x3fc90fdbu), pi = fd_from_bits(0x40490fdbu),
              pi_lo = fd_from_bits(0xb3bbbd2eu);
  const int32_t hx = (int32_t)fd_bits(x), hy = (int32_t)fd_bits(y);
  const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;  // NaN
  if (hx == 0x3f800000) return fd_atanf(y);               // x = 1
  const int m = (int)(((uint32_t)hy >> 31) & 1u) | (int)(((uint32_t)hx >> 30) & 2u);  // 2 sign(x) + sign(y)
  if (iy == 0) {
    if (m < 2) return y;
    return m == 2 ? pi + tiny : -pi - tiny;
  }
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) {
      if (m == 0) return pi_o_4 + tiny;
      if (m == 1) return -pi_o_4 - tiny;
      if (m == 2) return 3.0f * pi_o_4 + tiny;
      return -3.0f * pi_o_4 - tiny;
    }
    if (m == 0) return 0.0f;
    if (m == 1) return -0.0f;
    return m == 2 ? pi + tiny : -pi - tiny;
  }
  if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int32_t k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;          // |y / x| > 2^60
  else if (hx < 0 && k < -60) z = 0.0f;            // |y| / x < -2^60
  else z = fd_atanf(fd_from_bits(fd_bits(y / x) & 0x7fffffffu));
  if (m == 0) return z;
  if (m == 1) return fd_from_bits(fd_bits(z) ^ 0x80000000u);
  if (m == 2) return pi - (z - pi_lo);
  return (z - pi_lo) - pi;
}

}  // namespace vloam
