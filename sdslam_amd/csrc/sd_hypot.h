// hypot() with the exact results of the host libm (glibc 2.35, x86-64: C. F. Borges,
// "An Improved Algorithm for hypot(a,b)", corrected-sqrt kernel without FMA -- the variant
// sysdeps/ieee754/dbl-64/e_hypot.c builds when __FP_FAST_FMA is undefined).  glibc's hypot is
// NOT correctly rounded (0.2 % of inputs differ from the correctly rounded value), and OpenCV's
// one-sided Jacobi SVD calls it inside every rotation; for EPnP's 4-point minimal sets the
// 12x12 Gram matrix has a 4-dimensional null space whose basis is decided by last-bit
// rounding, so the device restates the same operation sequence.  tools/check_hypot.cc compares
// this file with the host libm on 2e8 random inputs (0 mismatches).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define SD_HD __host__ __device__ inline
#else
#define SD_HD inline
#endif

namespace sdsc {

SD_HD double hypot_kernel(double ax, double ay) {
  double t1, t2;
  double h = sqrt(ax * ax + ay * ay);
  if (h <= 2.0 * ay) {
    double delta = h - ay;
    t1 = ax * (2.0 * delta - ax);
    t2 = (delta - 2.0 * (ax - ay)) * delta;
  } else {
    double delta = h - ax;
    t1 = 2.0 * delta * (ax - 2.0 * ay);
    t2 = (4.0 * delta - ay) * ay + delta * delta;
  }
  h -= (t1 + t2) / (2.0 * h);
  return h;
}

SD_HD double hypot_glibc(double x, double y) {
  const double SCALE = 0x1p-600, LARGE_VAL = 0x1p+511, TINY_VAL = 0x1p-459, EPS = 0x1p-54;
  x = fabs(x);
  y = fabs(y);
  double ax = x < y ? y : x;
  double ay = x < y ? x : y;
  if (ax > LARGE_VAL) {
    if (ay <= ax * EPS) return ax + ay;
    return hypot_kernel(ax * SCALE, ay * SCALE) / SCALE;
  }
  if (ay < TINY_VAL) {
    if (ax >= ay / EPS) return ax + ay;
    return hypot_kernel(ax / SCALE, ay / SCALE) * SCALE;
  }
  if (ax >= ay / EPS) return ax + ay;
  return hypot_kernel(ax, ay);
}

}  // namespace sdsc
