// Bit-faithful device versions of the three libm float functions the reference's tick path calls:
//   std::cos / std::sin on float  (Velocity(angle, speed), agario/core/types.hpp:158-159, used by
//                                  Engine::disrupt, Engine.hpp:1283)
//   std::atan on float            (Velocity::direction, core/types.hpp:167-174, Engine.hpp:1279-1281)
//
// These live in a third-party dependency that is not in /root/reference: GNU libc 2.35
// (Ubuntu GLIBC 2.35-0ubuntu3.11, the libm the reference build links).  Its published algorithms are
// restated here:
//   sinf/cosf : sysdeps/ieee754/flt-32/{s_sinf.c,s_cosf.c,sincosf.h,sincosf_data.c} -- the
//               "optimized routines" implementation: range reduction by 2/pi in double, 4-quadrant
//               sign table, degree-7/8 polynomials evaluated in double, one final rounding to float;
//   atanf     : sysdeps/ieee754/flt-32/s_atanf.c -- fdlibm: 4-interval argument reduction and an
//               11-term odd/even split polynomial, all in float.
// The device evaluates the same operations in IEEE double / float; the build disables implicit
// contraction (-ffp-contract=off) and fma() is written out exactly where glibc's FMA build fuses.  tests/test_libm_restatement.py compares them against the host libm
// (exhaustively over all 2^32 floats when run with AGAR_EXHAUSTIVE=1: 0 mismatches on an FMA-capable
// x86-64 host; on a CPU without FMA glibc's own sinf/cosf differ from this in ~1e-8 of inputs).
#pragma once
#include <stdint.h>
#include <math.h>

#ifndef AG_DEV
#define AG_DEV static inline
#endif

AG_DEV uint32_t ag_asuint(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
AG_DEV float ag_asfloat(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }
AG_DEV uint32_t ag_abstop12(float x) { return (ag_asuint(x) >> 20) & 0x7ff; }

// polynomial sets: [0] quadrants 0/1, [1] quadrants 2/3 (cosine coefficients negated)
struct AgSinCos { double c0, c1, c2, c3, c4, s1, s2, s3; };
AG_DEV AgSinCos ag_sincos_tab(int k) {
  AgSinCos p;
  p.c0 = 0x1p0; p.c1 = -0x1.ffffffd0c621cp-2; p.c2 = 0x1.55553e1068f19p-5; p.c3 = -0x1.6c087e89a359dp-10; p.c4 = 0x1.99343027bf8c3p-16;
  p.s1 = -0x1.555545995a603p-3; p.s2 = 0x1.1107605230bc4p-7; p.s3 = -0x1.994eb3774cf24p-13;
  if (k) { p.c0 = -p.c0; p.c1 = -p.c1; p.c2 = -p.c2; p.c3 = -p.c3; p.c4 = -p.c4; }
  return p;
}
AG_DEV double ag_quadrant_sign(int n) { return ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0; }  // {1,-1,-1,1}

AG_DEV float ag_sinf_poly(double x, double x2, const AgSinCos &p, int n) {
  // a*b+c is evaluated with ONE rounding: x86-64 glibc selects its FMA build (s_sinf-fma /
  // s_cosf-fma, sysdeps/x86_64/fpu/multiarch) on every FMA-capable CPU, where GCC contracts these.
  if ((n & 1) == 0) {
    double x3 = x * x2;
    double s1 = fma(x2, p.s3, p.s2);
    double x7 = x3 * x2;
    double s = fma(x3, p.s1, x);
    return (float)fma(x7, s1, s);
  } else {
    double x4 = x2 * x2;
    double c2 = fma(x2, p.c4, p.c3);
    double c1 = fma(x2, p.c1, p.c0);
    double x6 = x4 * x2;
    double c = fma(x4, p.c2, c1);
    return (float)fma(x6, c2, c);
  }
}
// |x| < 120: one multiply by 2/pi * 2^24, quadrant from bits 24..31
AG_DEV double ag_reduce_fast(double x, int *np) {
  double r = x * 0x1.45F306DC9C883p+23;
  int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return fma(-(double)n, 0x1.921FB54442D18p0, x);
}
// 120 <= |x| < inf: 4/pi to 192 bits, 32x96 -> 128-bit product approximated by three 64-bit products
AG_DEV double ag_reduce_large(uint32_t xi, int *np) {
  const uint32_t inv_pio4[24] = {0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
                                 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,
                                 0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
  const uint32_t *arr = &inv_pio4[(xi >> 26) & 15];
  int shift = (xi >> 23) & 7;
  uint64_t n, res0, res1, res2;
  xi = (xi & 0xffffff) | 0x800000;
  xi <<= shift;
  res0 = xi * arr[0];
  res1 = (uint64_t)xi * arr[4];
  res2 = (uint64_t)xi * arr[8];
  res0 = (res2 >> 32) | (res0 << 32);
  res0 += res1;
  n = (res0 + (1ULL << 61)) >> 62;
  res0 -= n << 62;
  double x = (double)(int64_t)res0;
  *np = (int)n;
  return x * 0x1.921FB54442D18p-62;
}

AG_DEV float ag_sinf(float y) {
  double x = y, s; int n;
  if (ag_abstop12(y) < ag_abstop12(0x1.921FB6p-1f)) {
    s = x * x;
    if (ag_abstop12(y) < ag_abstop12(0x1p-12f)) return y;
    return ag_sinf_poly(x, s, ag_sincos_tab(0), 0);
  } else if (ag_abstop12(y) < ag_abstop12(120.0f)) {
    x = ag_reduce_fast(x, &n);
    s = ag_quadrant_sign(n);
    return ag_sinf_poly(x * s, x * x, ag_sincos_tab((n & 2) ? 1 : 0), n);
  } else if (ag_abstop12(y) < 0x7f8) {
    uint32_t xi = ag_asuint(y); int sign = (int)(xi >> 31);
    x = ag_reduce_large(xi, &n);
    s = ag_quadrant_sign(n + sign);
    return ag_sinf_poly(x * s, x * x, ag_sincos_tab(((n + sign) & 2) ? 1 : 0), n);
  }
  return (y - y) / (y - y);  // inf / nan -> nan
}
AG_DEV float ag_cosf(float y) {
  double x = y, s; int n;
  if (ag_abstop12(y) < ag_abstop12(0x1.921FB6p-1f)) {
    if (ag_abstop12(y) < ag_abstop12(0x1p-12f)) return 1.0f;
    return ag_sinf_poly(x, x * x, ag_sincos_tab(0), 1);
  } else if (ag_abstop12(y) < ag_abstop12(120.0f)) {
    x = ag_reduce_fast(x, &n);
    s = ag_quadrant_sign(n);
    return ag_sinf_poly(x * s, x * x, ag_sincos_tab((n & 2) ? 1 : 0), n ^ 1);
  } else if (ag_abstop12(y) < 0x7f8) {
    uint32_t xi = ag_asuint(y); int sign = (int)(xi >> 31);
    x = ag_reduce_large(xi, &n);
    s = ag_quadrant_sign(n + sign);
    return ag_sinf_poly(x * s, x * x, ag_sincos_tab(((n + sign) & 2) ? 1 : 0), n ^ 1);
  }
  return (y - y) / (y - y);
}

AG_DEV float ag_atanf(float x) {
  const float atanhi[4] = {ag_asfloat(0x3eed6338u), ag_asfloat(0x3f490fdau), ag_asfloat(0x3f7b985eu), ag_asfloat(0x3fc90fdau)};
  const float atanlo[4] = {ag_asfloat(0x31ac3769u), ag_asfloat(0x33222168u), ag_asfloat(0x33140fb4u), ag_asfloat(0x33a22168u)};
  const float aT[11] = {ag_asfloat(0x3eaaaaabu), ag_asfloat(0xbe4ccccdu), ag_asfloat(0x3e124925u), ag_asfloat(0xbde38e38u),
                        ag_asfloat(0x3dba2e6eu), ag_asfloat(0xbd9d8795u), ag_asfloat(0x3d886b35u), ag_asfloat(0xbd6ef16bu),
                        ag_asfloat(0x3d4bda59u), ag_asfloat(0xbd15a221u), ag_asfloat(0x3c8569d7u)};
  float w, s1, s2, z;
  int32_t hx = (int32_t)ag_asuint(x), ix = hx & 0x7fffffff, id;
  if (ix >= 0x4c000000) {  // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;
    if (hx > 0) return atanhi[3] + atanlo[3];
    return -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) {  // |x| < 0.4375
    if (ix < 0x31000000) return x;  // |x| < 2^-29
    id = -1;
  } else {
    x = ag_asfloat((uint32_t)ix);
    if (ix < 0x3f980000) {
      if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
      else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
    } else {
      if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
      else { id = 3; x = -1.0f / x; }
    }
  }
  z = x * x;
  w = z * z;
  s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
  return (hx < 0) ? -z : z;
}
