// sinf/cosf with the exact results of the host libm the reference links against.
//
// The reference computes the rBRIEF steering terms with libm: `a = cos(angle), b = sin(angle)`
// on floats (reference src/ORBextractor.cc:109-110) and rounds rotated pattern coordinates
// with cvRound, so descriptor bits depend on the last bit of sinf/cosf.  glibc >= 2.28 ships
// the ARM "optimized routines" single-precision algorithm (S. Nagy, W. Dijkstra; published in
// github.com/ARM-software/optimized-routines math/sincosf.h): range-reduce by pi/2 in double,
// evaluate a degree-7/8 polynomial in double, round once to float.  It is NOT correctly
// rounded (0.04-0.09 % of inputs differ from the correctly rounded value), so it has to be
// restated operation by operation.  The fused multiply-adds below are placed exactly where
// glibc 2.35's x86-64 FMA ifunc variant (the one selected on every FMA-capable host,
// including this image's Xeon and the GPU box's host CPU) places them; gfx950 v_fma_f64 is
// an IEEE fused op, so device results are bit-identical.  tools/check_sincosf.cc verifies
// this file against the host libm for EVERY float in [0, 2*pi] (1.09e9 inputs, 0 mismatches).
//
// Domain handled: |y| < 120 (the extractor only passes angles in [0, 2*pi]); larger
// arguments fall back to the double-precision libm call.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SD_HD __host__ __device__ inline
#else
#define SD_HD inline
#endif

namespace sdsc {

SD_HD uint32_t f2u(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
SD_HD uint32_t abstop12(float f) { return (f2u(f) >> 20) & 0x7ff; }

// table[q][..]: q = 1 is the sign-flipped copy used when (n & 2)
struct Poly { double c0, c1, s1, c2, s2, c3, s3, c4; };

SD_HD Poly poly(int neg) {
  const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5,
               c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
  const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
  Poly p;
  if (!neg) { p.c0 = c0; p.c1 = c1; p.c2 = c2; p.c3 = c3; p.c4 = c4; }
  else { p.c0 = -c0; p.c1 = -c1; p.c2 = -c2; p.c3 = -c3; p.c4 = -c4; }
  p.s1 = s1; p.s2 = s2; p.s3 = s3;
  return p;
}

SD_HD float sin_poly(double x, double x2, const Poly& p) {
  double x3 = x2 * x;
  double s1 = __builtin_fma(x2, p.s3, p.s2);
  double x7 = x2 * x3;
  double s = __builtin_fma(x3, p.s1, x);
  return (float)__builtin_fma(s1, x7, s);
}
SD_HD float cos_poly(double x2, const Poly& p) {
  double x4 = x2 * x2;
  double a = __builtin_fma(x2, p.c1, p.c0);
  double b = __builtin_fma(x2, p.c4, p.c3);
  double x6 = x2 * x4;
  double c = __builtin_fma(x4, p.c2, a);
  return (float)__builtin_fma(b, x6, c);
}

// x -> x - n*pi/2, n = round(x * 2/pi)
SD_HD double reduce_fast(double x, int* np) {
  const double hpi_inv = 0x1.45f306dc9c883p+23, hpi = 0x1.921fb54442d18p+0;
  double r = x * hpi_inv;
  int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return __builtin_fma(-(double)n, hpi, x);
}

SD_HD double sign_of(int n) { return ((n + 1) & 2) ? -1.0 : 1.0; }  // {1,-1,-1,1}[n&3]

SD_HD float sinf_glibc(float y) {
  double x = y;
  if (abstop12(y) < 0x3f4) {          // |y| < pi/4
    if (abstop12(y) < 0x398) return y;  // |y| < 2^-12
    return sin_poly(x, x * x, poly(0));
  }
  int n;
  x = reduce_fast(x, &n);
  Poly p = poly((n & 2) != 0);
  if ((n & 1) == 0) return sin_poly(x * sign_of(n), x * x, p);
  return cos_poly(x * x, p);
}

SD_HD float cosf_glibc(float y) {
  double x = y;
  if (abstop12(y) < 0x3f4) {
    if (abstop12(y) < 0x398) return 1.0f;
    return cos_poly(x * x, poly(0));
  }
  int n;
  x = reduce_fast(x, &n);
  Poly p = poly((n & 2) != 0);
  if ((n & 1) != 0) return sin_poly(x * sign_of(n), x * x, p);
  return cos_poly(x * x, p);
}

}  // namespace sdsc
